"""The step epilogue's two streaming passes on COLD buffers (rotating over more bytes than the 256-MB infinity cache holds -- the step's
situation: a bucket's gradients were written tens of milliseconds earlier), next to ATen reductions / copies of the same bytes:
of_sumsq_partial (read 4 B / element) and of_adamw_clip (read 16, write 14 B / element) on a 36.7-M-element bucket (one gated block of OF-3B).
Optional: a second build of the library (python tools/probes/step_epilogue_cold_probe.py other.so [parts]) with another OF_SUMSQ_PARTS.
PROFILING TOOL."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops

ops = Ops.default()
libs = {"product": (ops.lib, abi.OF_SUMSQ_PARTS)}
if len(sys.argv) > 1:
    other = ctypes.CDLL(sys.argv[1])
    abi.declare(other, require_all=False)
    libs["variant"] = (other, int(sys.argv[2]) if len(sys.argv) > 2 else abi.OF_SUMSQ_PARTS)
n, NB = 36_708_352, 8
dev = "cuda"
g = [torch.randn(n, device=dev) for _ in range(NB)]
p = [torch.randn(n, device=dev) for _ in range(NB)]
m = [torch.zeros(n, device=dev) for _ in range(NB)]
v = [torch.zeros(n, device=dev) for _ in range(NB)]
pb = [torch.empty(n, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
parts = torch.empty(8192, device=dev)
acc = torch.ones(1, device=dev)
stream = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, rotate, reps=32):
    ev = []
    for i in range(reps + 4):
        k = i % NB if rotate else 0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn(k)
        e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[4:])
    return round(t[len(t) // 2], 1)


for name, (lib, nparts) in libs.items():
    def sumsq(k, lib=lib):
        assert lib.of_sumsq_partial(g[k].data_ptr(), n, parts.data_ptr(), stream()) == 0

    def adamw(k, lib=lib):
        assert lib.of_adamw_clip(p[k].data_ptr(), g[k].data_ptr(), m[k].data_ptr(), v[k].data_ptr(), pb[k].data_ptr(), n, acc.data_ptr(), 1.0,
                                 1e-4, 0.9, 0.999, 1e-8, 0.1, 1.0, 1, 0, None, stream()) == 0
    for lab, fn, nbytes in (("of_sumsq_partial", sumsq, 4 * n), ("of_adamw_clip (no gradient clear)", adamw, 30 * n)):
        same, cold = timed(fn, False), timed(fn, True)
        print(json.dumps({"library": name, "sumsq_parts": nparts, "kernel": lab, "same_buffer_us": same, "rotating_us": cold,
                          "TBps_rotating": round(nbytes / cold / 1e6, 2)}), flush=True)
for lab, fn, nbytes in (("ATen sum (read 4 B)", lambda k: torch.sum(g[k]), 4 * n), ("ATen fp32 copy (read 4, write 4)", lambda k: m[k].copy_(g[k]), 8 * n),
                        ("ATen mul_ (read 4, write 4)", lambda k: v[k].mul_(1.0), 8 * n)):
    same, cold = timed(fn, False), timed(fn, True)
    print(json.dumps({"kernel": lab, "same_buffer_us": same, "rotating_us": cold, "TBps_rotating": round(nbytes / cold / 1e6, 2)}), flush=True)
