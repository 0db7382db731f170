"""Timing-only ablations of the ping-pong GEMM kernel (which part of the loop costs what).  Results of the ablated
launches are garbage by design; only the duration matters."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
from tools.tools_lib import tools_ops
from tools.bench_kernels import timeit

ops = tools_ops()   # tools/libofhip_tools.so: the ablation / A-B variants are not in the product library
names = {0: "full", 1: "no-dma", 2: "no-ldsread", 3: "no-dma,no-ldsread", 4: "no-mfma", 5: "no-dma,no-mfma",
         6: "dma-only", 22: "dma-only,no-drain", 16: "full,no-drain(racy)", 18: "dma+mfma,no-drain"}
for (M, N, K) in [(8192, 2048, 8192), (8192, 8192, 2048)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for mask, nm in names.items():
        ms = timeit(lambda: ops.gemm(A, B, C, safe=16 + mask))
        print(json.dumps(dict(shape=[M, N, K], variant=nm, ms=round(ms, 4), tflops_equiv=round(fl / ms / 1e9, 1))), flush=True)
    ms = timeit(lambda: torch.matmul(A, B.t()))
    print(json.dumps(dict(shape=[M, N, K], variant="torch.matmul (hipBLASLt)", ms=round(ms, 4), tflops_equiv=round(fl / ms / 1e9, 1))), flush=True)
