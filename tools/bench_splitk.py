"""Split-K factor sweep for the small-output weight-gradient GEMMs (dW = dY^T X, K = tokens)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from tools.bench_kernels import timeit
ops = Ops.default()
for (M, N, K) in [(512, 2048, 8192), (2048, 512, 8192), (1024, 1024, 4096), (1024, 1024, 20480), (512, 1024, 4096), (1024, 4096, 4096)]:
    A = torch.randn(K, M, device="cuda").to(torch.bfloat16)
    B = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    r = dict(MNK=[M, N, K])
    for ls in range(0, 5):
        ms = timeit(lambda: ops.gemm(A, B, C, ta=True, tb=True, epi=abi.EPI_ACC_F32, safe=8 + ls))
        r[f"split{1 << ls}"] = round(ms * 1e3, 1)
    ms = timeit(lambda: ops.gemm(A, B, C, ta=True, tb=True, epi=abi.EPI_ACC_F32))
    r["auto_us"] = round(ms * 1e3, 1)
    ms = timeit(lambda: torch.matmul(A.t(), B))
    r["torch_us"] = round(ms * 1e3, 1)
    print(json.dumps(r), flush=True)
