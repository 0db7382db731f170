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
# the fused attention branch of a gated block (csrc/xattn_fused.hip) at config 2's shape: 8192 rows x 2048, 2 x 64 media tokens per sequence
from open_flamingo_amd.hip import path
from open_flamingo_amd.hip.ops import BF16
B_, L_, T_, n_, d_ = 32, 256, 2, 64, 2048
P = {"attn.norm.weight": torch.rand(d_, device="cuda") + 0.5, "attn.norm.bias": torch.zeros(d_, device="cuda"),
     "ff.0.weight": torch.rand(d_, device="cuda") + 0.5, "ff.0.bias": torch.zeros(d_, device="cuda"), "attn_gate": g}
Wd = path.WeightDict({"attn.to_q.weight": r(512, d_) * d_ ** -0.5, "attn.to_out.weight": r(d_, 512) * 512 ** -0.5})
x, kv = torch.randn(B_ * L_, d_, device="cuda"), r(B_ * T_ * n_, 1024)
ml = torch.zeros(B_, L_, dtype=torch.bool, device="cuda")
ml[:, 0] = True
ml[:, L_ // 2] = True
tt = torch.empty(B_, L_, dtype=torch.int32, device="cuda")
ops.text_time(ml.to(torch.uint8).contiguous(), tt, L_, False)
for _ in range(2):
    y, S = path.masked_cross_attention_fwd(ops, P, Wd, x, None, tt, B=B_, L=L_, T=T_, n=n_, heads=8, only_immediate=True, gate=g, residual=True,
                                           kv=kv, next_ln=(P["ff.0.weight"], P["ff.0.bias"]))
assert "next_ln" in S
torch.cuda.synchronize()
