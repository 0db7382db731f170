"""Launch the two big-tile kernels and hipBLASLt (torch.matmul) on the same random operands for rocprofv3 passes.
argv: M N K [layout NT|NN|TN]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
ops = Ops.default()
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 2048, 8192)
lay = sys.argv[4] if len(sys.argv) > 4 else "NT"
fill = sys.argv[5] if len(sys.argv) > 5 else "random"
ta, tb = lay[0] == "T", lay[1] == "N"
mk = torch.randn if fill == "random" else torch.zeros
A = mk((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
B = mk((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
Am, Bm = (A.t() if ta else A), (B if tb else B.t())
for _ in range(4):
    ops.gemm(A, B, C, ta=ta, tb=tb, safe=4)
    ops.gemm(A, B, C, ta=ta, tb=tb, safe=6)
    ops.gemm(A, B, C, ta=ta, tb=tb, safe=7)
    torch.matmul(Am, Bm)
torch.cuda.synchronize()
