"""Cost of each fused epilogue on the two N=8192, K=2048 FFN shapes (same main loop, different epilogue)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from tools.bench_kernels import timeit
ops = Ops.default()
M, N, K = 8192, 8192, 2048
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W_nt = torch.randn(N, K, device="cuda").to(torch.bfloat16)
W_nn = torch.randn(K, N, device="cuda").to(torch.bfloat16)
Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
C2 = torch.empty_like(Cb)
aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
gate = torch.tensor([0.5], device="cuda")
dot = torch.zeros(1, device="cuda")
fl = 2.0 * M * N * K
cases = [("NT store_bf16", lambda: ops.gemm(A, W_nt, Cb)),
         ("NT gelu (b only)", lambda: ops.gemm(A, W_nt, Cb, epi=abi.EPI_GELU)),
         ("NT gelu (a and b)", lambda: ops.gemm(A, W_nt, Cb, epi=abi.EPI_GELU, out2=C2)),
         ("NN store_bf16", lambda: ops.gemm(A, W_nn, Cb, tb=True)),
         ("NN scale_dot (no dot_out)", lambda: ops.gemm(A, W_nn, Cb, tb=True, epi=abi.EPI_SCALE_DOT, aux=aux, gate=gate)),
         ("NN scale_dot", lambda: ops.gemm(A, W_nn, Cb, tb=True, epi=abi.EPI_SCALE_DOT, aux=aux, gate=gate, dot=dot)),
         ("NN dgelu_dot (no dot_out)", lambda: ops.gemm(A, W_nn, Cb, tb=True, epi=abi.EPI_DGELU_DOT, aux=aux, gate=gate)),
         ("NN dgelu_dot", lambda: ops.gemm(A, W_nn, Cb, tb=True, epi=abi.EPI_DGELU_DOT, aux=aux, gate=gate, dot=dot))]
for name, fn in cases:
    ms = timeit(fn)
    print(json.dumps(dict(case=name, ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1))), flush=True)
