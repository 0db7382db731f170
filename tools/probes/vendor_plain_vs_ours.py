"""Plain bf16 GEMM, store epilogue only: this library's big-tile kernel vs the vendor library (torch.mm) on the same operands,
three layouts, the step's two FFN shapes and 8192^3; operands N(0,1) x N(0,0.05^2) as in tools/bench_gemm_ab.py.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import make, timed
ops = Ops.default()
for lay, ta, tb in (("NT", 0, 0), ("NN", 0, 1), ("TN", 1, 1)):
    for (M, N, K) in ((8192, 2048, 8192), (8192, 8192, 2048), (8192, 8192, 8192)):
        A, B, C, kw = make(M, N, K, ta, tb, abi.EPI_STORE_BF16)
        At = A.t() if ta else A
        Bt = B if tb else B.t()
        fns = {"ours": lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb)), "vendor": lambda: torch.mm(At, Bt, out=C)}
        best = {k: 1e9 for k in fns}
        for fn in fns.values():
            for _ in range(3): fn()
        torch.cuda.synchronize()
        for _ in range(4):
            for k, fn in fns.items():
                best[k] = min(best[k], timed(fn, 10))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(layout=lay, MNK=[M, N, K], ours_us=round(best["ours"] * 1e3, 1), vendor_us=round(best["vendor"] * 1e3, 1),
                              ours_tflops=round(fl / best["ours"] / 1e9), vendor_tflops=round(fl / best["vendor"] / 1e9))), flush=True)
