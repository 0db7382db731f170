import json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import make, timed
ops = Ops.default()
E = abi
for name, ta, tb, epi in (("NN store", 0, 1, E.EPI_STORE_BF16), ("NT gate_res", 0, 0, E.EPI_GATE_RESID)):
    for (M, N, K) in ((2048, 4096, 16384), (2048, 4096, 8192), (4096, 4096, 8192), (4096, 4096, 16384), (2048, 4096, 4096)):
        A, B, C, kw = make(M, N, K, ta, tb, epi)
        fn = lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, **kw)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ms = min(timed(fn, 10) for _ in range(4))
        print(json.dumps(dict(case=name, MNK=[M, N, K], tiles=(M // 256) * (N // 256), us=round(ms * 1e3, 1), tflops=round(2.0 * M * N * K / ms / 1e9, 1))), flush=True)
