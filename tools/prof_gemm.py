"""Launch the ping-pong GEMM (and ablations) a few times for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
from tools.tools_lib import tools_ops
ops = tools_ops()   # tools/libofhip_tools.so: the ablation / A-B variants are not in the product library
M, N, K = 8192, 2048, 8192
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
masks = [int(x) for x in sys.argv[1:]] or [0]
for m in masks:
    for _ in range(3):
        ops.gemm(A, B, C, safe=16 + m)
torch.cuda.synchronize()
