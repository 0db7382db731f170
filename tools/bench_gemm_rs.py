"""A/B of the two operand-staging modes of the 256x256 ping-pong GEMM on the hot path's big shapes:
safe=4 LDS-DMA (global_load_lds) vs safe=5 register staging (global_load -> VGPR -> ds_write_b128).
Interleaved timing, bit-equality of the results and a repeat screen of the register-staged mode."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from tools.tools_lib import tools_ops
from tools.bench_kernels import timeit
ops = tools_ops()   # tools/libofhip_tools.so: the ablation / A-B variants are not in the product library
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
gate = torch.tensor([0.5], device="cuda")
cases = []
# (name, M, N, K, kwargs builder)
X, W1, W3 = r(8192, 2048), r(8192, 2048), r(2048, 8192)         # tokens x d, ff.1 (4d x d), ff.3 (d x 4d)
H_, dY, dA = r(8192, 8192), r(8192, 2048), r(8192, 8192)
aux = r(8192, 8192)
res = torch.randn(8192, 2048, device="cuda", generator=g)
cases.append(("ffn up + GELU            NT 8192x8192x2048", 8192, 8192, 2048,
              lambda out, safe: ops.gemm(X, W1, out, epi=abi.EPI_GELU, safe=safe), torch.bfloat16))
cases.append(("ffn down + gate + resid  NT 8192x2048x8192", 8192, 2048, 8192,
              lambda out, safe: ops.gemm(H_, W3, out, epi=abi.EPI_GATE_RESID, aux=res, gate=gate, safe=safe), torch.float32))
cases.append(("da = dy W3 (dgelu)       NN 8192x8192x2048", 8192, 8192, 2048,
              lambda out, safe: ops.gemm(dY, W3, out, tb=True, epi=abi.EPI_DGELU_DOT, aux=aux, gate=gate, safe=safe), torch.bfloat16))
cases.append(("du = da W1               NN 8192x2048x8192", 8192, 2048, 8192,
              lambda out, safe: ops.gemm(dA, W1, out, tb=True, safe=safe), torch.bfloat16))
cases.append(("dW1 = da^T u             TN 8192x2048x8192", 8192, 2048, 8192,
              lambda out, safe: ops.gemm(dA, X, out, ta=True, tb=True, epi=abi.EPI_ACC_F32, safe=safe), torch.float32))
for name, M, N, K, fn, dt in cases:
    o4, o5 = torch.empty(M, N, device="cuda", dtype=dt), torch.empty(M, N, device="cuda", dtype=dt)
    fn(o4, 4); fn(o5, 5)
    same = bool(torch.equal(o4, o5))
    stable = True
    for _ in range(6):
        o = torch.empty(M, N, device="cuda", dtype=dt)
        fn(o, 5)
        stable = stable and bool(torch.equal(o, o5))
    variants = {"dma": 4, "regs": 5}
    t = {k: [] for k in variants}
    for _ in range(3):
        for k, sf in variants.items():
            t[k].append(timeit(lambda: fn(o4, sf)))
    fl = 2.0 * M * N * K
    row = dict(case=name, bit_equal=same, repeat_stable=stable)
    for k in variants:
        row[k + "_ms"] = round(min(t[k]), 4)
        row[k + "_tflops"] = round(fl / min(t[k]) / 1e9, 1)
    print(json.dumps(row), flush=True)
