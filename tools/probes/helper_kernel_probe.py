"""Can an erf-GELU pass (64-register elementwise kernel, no LDS) run BESIDE the 448-register big-tile GEMM on the same CUs for free?
Stream A: NT 8192 x 8192 x 2048 bf16-store GEMMs (the 256x256 kernel) back to back; stream B: of_gelu_fwd / of_gelu_bwd passes over
independent 8192 x 8192 bf16 buffers (rotating: cold).  Wall time of A alone, B alone, both at once.  What a second kernel that does
the fat epilogues' math next to the GEMM (last item) could hope for.  PROFILING TOOL."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools_lib import routed_ops      # product library; kernel-forcing selectors (safe >= 2) -> tools/libofhip_tools.so
from bench_gemm_ab import make

ops = routed_ops()
opsB = Ops(ops.lib, ops._stream_fn)
E = abi
A, B, C, kw = make(8192, 8192, 2048, 0, 0, E.EPI_STORE_BF16)
NB = 6
xs = [torch.randn(8192, 8192, device="cuda").to(torch.bfloat16) for _ in range(NB)]
ys = [torch.empty_like(x) for x in xs]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
NA, NP = 20, 40


def run_a():
    for _ in range(NA):
        ops.gemm(A, B, C, epi=E.EPI_STORE_BF16, safe=16)


def run_b(kind):
    for i in range(NP):
        if kind == "gelu_fwd":
            opsB.gelu_fwd(xs[i % NB], out=ys[i % NB])
        else:
            opsB.gelu_bwd(xs[i % NB], xs[(i + 1) % NB], out=ys[i % NB])


def wall(fa, fb):
    torch.cuda.synchronize()
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    if fa:
        with torch.cuda.stream(sA):
            ea0.record()
            fa()
            ea1.record()
    if fb:
        with torch.cuda.stream(sB):
            eb0.record()
            fb()
            eb1.record()
    torch.cuda.synchronize()
    return (ea0.elapsed_time(ea1) if fa else None, eb0.elapsed_time(eb1) if fb else None)


for kind in ("gelu_fwd", "gelu_bwd"):
    for _ in range(2):
        wall(run_a, lambda: run_b(kind))
    a_alone = min(wall(run_a, None)[0] for _ in range(3))
    b_alone = min(wall(None, lambda: run_b(kind))[1] for _ in range(3))
    both = [wall(run_a, lambda: run_b(kind)) for _ in range(3)]
    a_with, b_with = min(x[0] for x in both), min(x[1] for x in both)
    print(json.dumps(dict(helper=kind, gemm_us_alone=round(a_alone / NA * 1e3, 1), pass_us_alone=round(b_alone / NP * 1e3, 1),
                          gemm_us_beside_the_passes=round(a_with / NA * 1e3, 1), pass_us_beside_the_gemms=round(b_with / NP * 1e3, 1),
                          serial_ms=round(a_alone + b_alone, 3), concurrent_ms=round(max(a_with, b_with), 3),
                          note="A = 20 GEMMs NT 8192x8192x2048 bf16 store (w4m256), B = 40 passes over 8192x8192 bf16")), flush=True)
