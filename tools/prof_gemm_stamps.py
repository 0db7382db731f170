"""Where a wave of the ping-pong GEMM spends its cycles (ABL 64 build of the kernel: shader-cycle stamps around every
part of the load and compute segments, summed over the K loop, per wave).  Prints per-group averages per stage."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
from tools.tools_lib import tools_ops
from tools.bench_kernels import timeit
ops = tools_ops()   # tools/libofhip_tools.so: the ablation / A-B variants are not in the product library
NAMES = ["frag_read_issue", "dma_issue", "vmcnt_wait", "lgkm_wait", "load_seg_barrier", "mfma_issue", "compute_seg_barrier", "-"]
for (M, N, K) in [(8192, 2048, 8192), (8192, 8192, 2048)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tiles = (M // 256) * (N // 256)
    # the counters come back through the C2 pointer; the wrapper wants an (M, N) bf16 tensor there
    buf = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, B, C, safe=16 + 64, out2=buf)
    torch.cuda.synchronize()
    t = buf.view(torch.int32).flatten()[: tiles * 64].view(tiles, 8, 8).double()
    stages = K // 64
    ms_plain = timeit(lambda: ops.gemm(A, B, C, safe=4))
    ms_stamp = timeit(lambda: ops.gemm(A, B, C, safe=16 + 64, out2=buf))
    row = dict(shape=[M, N, K], plain_ms=round(ms_plain, 4), stamped_ms=round(ms_stamp, 4))
    for g, name in ((slice(0, 4), "G0"), (slice(4, 8), "G1")):
        per_stage = t[:, g, :].mean(dim=(0, 1)) / stages
        row[name] = {n: round(float(v), 1) for n, v in zip(NAMES[:7], per_stage[:7])}
        row[name]["sum_per_stage"] = round(float(per_stage[:7].sum()), 1)
    print(json.dumps(row), flush=True)
