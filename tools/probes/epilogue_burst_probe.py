"""Is a tile's epilogue bound by its own CU or by the chip-wide burst?  The 4-wave kernel (safe = 7) on grids of 32 and 256
tiles (one per CU on 32 / 256 CUs), K sweep: the intercept at K -> 0 is prologue + epilogue of one tile; if it is the same at
32 tiles (HBM nearly idle) as at 256 (every CU storing at once), de-phasing the CUs cannot hide it.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools_lib import routed_ops      # product library; kernel-forcing selectors (safe >= 2) -> tools/libofhip_tools.so
from bench_gemm_ab import make, timed

ops = routed_ops()
E = abi
for name, ta, tb, epi in (("NT store_bf16", 0, 0, E.EPI_STORE_BF16), ("NT gelu two outputs", 0, 0, E.EPI_GELU), ("NN dgelu_dot", 0, 1, E.EPI_DGELU_DOT),
                          ("TN acc_f32", 1, 1, E.EPI_ACC_F32)):
    for (M, N) in ((1024, 2048), (2048, 2048), (4096, 2048), (8192, 2048)):
        rec = dict(epilogue=name, M=M, N=N, tiles=(M // 256) * (N // 256))
        ts = {}
        for K in (256, 512, 1024, 2048, 4096):
            A, B, C, kw = make(M, N, K, ta, tb, epi)
            fn = lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=7, **kw)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts[K] = min(timed(fn, 20) for _ in range(4)) * 1e3
            rec["K%d_us" % K] = round(ts[K], 2)
        slope = (ts[4096] - ts[1024]) / 3072.0            # us per unit of K
        rec["us_per_64_of_K"] = round(slope * 64, 3)
        rec["intercept_us"] = round(ts[1024] - slope * 1024, 2)
        print(json.dumps(rec), flush=True)
