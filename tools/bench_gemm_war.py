"""Does the VGPR write-after-read interlock between a wave's queued MFMAs and its next fragment reads cost time?
ABL 128 build: identical LDS / DMA / MFMA work, but the MFMAs read constant registers (results wrong by design)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
from tools.tools_lib import tools_ops
from tools.bench_kernels import timeit
ops = tools_ops()   # tools/libofhip_tools.so: the ablation / A-B variants are not in the product library
for (M, N, K) in [(8192, 2048, 8192), (8192, 8192, 2048)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = {}
    for _ in range(3):
        for name, safe in (("full", 16 + 0), ("mfma_on_constant_regs", 16 + 128), ("no_frag_reads", 16 + 2), ("mfma_only", 16 + 3)):
            t.setdefault(name, []).append(timeit(lambda: ops.gemm(A, B, C, safe=safe)))
    # DVFS control: the constant-register MFMAs toggle fewer bits (lower power, higher clock).  Same kernels on
    # constant-filled operands: if the interlock matters, "full" stays slower than the constant-register build here too.
    A.fill_(1.0)
    B.fill_(0.5)
    for _ in range(3):
        for name, safe in (("full_const_data", 16 + 0), ("mfma_on_constant_regs_const_data", 16 + 128)):
            t.setdefault(name, []).append(timeit(lambda: ops.gemm(A, B, C, safe=safe)))
    print(json.dumps(dict(shape=[M, N, K], **{k: round(min(v), 4) for k, v in t.items()})), flush=True)
