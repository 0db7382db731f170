"""Launch the step's dominant GEMM shapes a few times each (for rocprofv3 --pmc traffic passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
ops = Ops.default()
def r(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
# dW1 = da^T u : (TN) M=8192 N=2048 K=8192 tokens, fp32 out
A, B, C = r(8192, 8192), r(8192, 2048), torch.empty(8192, 2048, device="cuda")
for _ in range(2): ops.gemm(A, B, C, ta=True, tb=True, epi=abi.EPI_ACC_F32)      # beta = 0: what the training step runs
# since the step epilogue leaves the weight gradients for the backward to overwrite (launches 1 and 2 of this symbol)
# the same launch accumulating into an existing gradient (beta = 1: a second backward of the same optimizer step,
# e.g. LAION + MMC4) -- launches 3 and 4 of this kernel symbol
for _ in range(2): ops.gemm(A, B, C, ta=True, tb=True, epi=abi.EPI_ACC_F32, beta=1.0)
# ffn down + gate + residual : (NT) M=8192 N=2048 K=8192
A, W, res, out = r(8192, 8192), r(2048, 8192), torch.randn(8192, 2048, device="cuda"), torch.empty(8192, 2048, device="cuda")
g = torch.tensor([0.5], device="cuda")
for _ in range(2): ops.gemm(A, W, out, epi=abi.EPI_GATE_RESID, aux=res, gate=g)
# ffn up + gelu : (NT) M=8192 N=8192 K=2048
A, W, b, a = r(8192, 2048), r(8192, 2048), torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16), torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)
for _ in range(2): ops.gemm(A, W, b, epi=abi.EPI_GELU, out2=a)
# ffn dX with the gate-gradient dot : (NN) M=8192 N=8192 K=2048, erf-GELU derivative -- of_gemm's selection: the two-workgroups-per-CU kernel
dy, W2, x = r(8192, 2048), r(2048, 8192), r(8192, 8192)
da, dot = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16), torch.zeros(1, device="cuda")
for _ in range(2): ops.gemm(dy, W2, da, tb=True, epi=abi.EPI_DGELU_DOT, aux=x, gate=g, dot=dot)
torch.cuda.synchronize()
