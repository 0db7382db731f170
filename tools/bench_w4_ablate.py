"""Timing-only ablations of the 4-wave GEMM kernel (tools/libofhip_tools.so; ablated launches give garbage by design)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.tools_lib import tools_ops
ops = tools_ops()
names = {0: "full", 1: "no-lds-write", 2: "no-global-load", 3: "no-staging", 4: "no-frag-read", 7: "mfma+barrier only",
         8: "no-barrier", 9: "no-barrier,no-lds-write", 15: "mfma only", 32: "no-k-advance (loads always hit cache)",
         40: "no-k-advance,no-barrier"}
dma_names = {0: "DMA full", 2: "DMA no-global-load", 8: "DMA no-barrier", 32: "DMA no-k-advance", 40: "DMA no-k-advance,no-barrier"}


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


for (M, N, K) in [(8192, 2048, 8192), (8192, 8192, 2048)]:
    for fill in ("random", "zeros"):
        A = (torch.randn(M, K, device="cuda") if fill == "random" else torch.zeros(M, K, device="cuda")).to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") if fill == "random" else torch.zeros(N, K, device="cuda")).to(torch.bfloat16)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        rows = {}
        for rnd in range(2):
            for mask, nm in names.items():
                ms = timeit(lambda: ops.gemm(A, B, C, safe=32 + mask))
                rows.setdefault(nm, []).append(ms)
            for mask, nm in dma_names.items():      # C2 != NULL selects the LDS-DMA staged variant of the ablation entry
                ms = timeit(lambda: ops.gemm(A, B, C, safe=32 + mask, out2=C))
                rows.setdefault(nm, []).append(ms)
            rows.setdefault("pp (8-wave LDS-DMA)", []).append(timeit(lambda: ops.gemm(A, B, C, safe=4)))
            rows.setdefault("torch.matmul (hipBLASLt)", []).append(timeit(lambda: torch.matmul(A, B.t())))
        for nm, v in rows.items():
            print(json.dumps(dict(shape=[M, N, K], fill=fill, variant=nm, ms=round(min(v), 4), tflops_equiv=round(fl / min(v) / 1e9, 1))), flush=True)
