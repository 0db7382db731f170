"""Does the row stride (L2/HBM channel mapping) limit the operand DMA?  Same GEMM with padded leading dimensions."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
from tools.tools_lib import tools_ops
from tools.bench_kernels import timeit
ops = tools_ops()   # tools/libofhip_tools.so: the ablation / A-B variants are not in the product library
for (M, N, K) in [(8192, 2048, 8192), (8192, 8192, 2048)]:
    for pad in (0, 64, 128, 192, 320):
        A = torch.randn(M, K + pad, device="cuda").to(torch.bfloat16)[:, :K]
        B = torch.randn(N, K + pad, device="cuda").to(torch.bfloat16)[:, :K]
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        r = dict(shape=[M, N, K], pad=pad)
        for mask, nm in ((0, "full"), (6, "dma_only"), (22, "dma_only_nodrain"), (3, "mfma_only")):
            ms = timeit(lambda: ops.gemm(A, B, C, safe=16 + mask))
            r[nm] = round(ms, 4)
        ms = timeit(lambda: torch.matmul(A, B.t()))
        r["torch"] = round(ms, 4)
        r["full_tflops"] = round(fl / r["full"] / 1e9, 1)
        print(json.dumps(r), flush=True)
