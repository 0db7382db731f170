"""libofhip's HBM-bound element-wise kernels on COLD buffers (rotating over more bytes than the 256-MB infinity cache), at the step's
shapes, alone and right behind a bf16 8192 x 8192 x 2048 matrix product (the step's situation: the chip at its power-limited clock),
each next to an ATen copy that moves the same bytes.  Further libraries on the command line (name=path) are timed alongside the product:
    python tools/probes/elementwise_cold_probe.py u2=tools/ab/libofhip_ew2.so u4=tools/ab/libofhip_ew4.so
PROFILING TOOL."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops

libs = {"product": Ops.default()}
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    lib = ctypes.CDLL(path)
    abi.declare(lib, require_all=False)
    libs[name] = Ops(lib, lambda: torch.cuda.current_stream().cuda_stream)
dev, BF, F32 = "cuda", torch.bfloat16, torch.float32
A = torch.randn(8192, 2048, device=dev).to(BF)
B = torch.randn(2048, 8192, device=dev).to(BF)
C = torch.empty(8192, 8192, device=dev, dtype=BF)


def timed(fn, nb, mm, reps=24):
    ev = []
    for i in range(reps + 4):
        if mm:
            torch.mm(A, B, out=C)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn(i % nb)
        e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[4:])
    return round(t[len(t) // 2], 1)


def bufs(shape, dtype, n):
    return [torch.randn(*shape, device=dev).to(dtype) for _ in range(n)]


def case(name, nbytes, nb, fns, ref):
    rec = {"kernel": name, "MB": round(nbytes / 1e6)}
    for mm in (0, 1):
        tag = "behind_a_gemm" if mm else "alone"
        rec["aten_copy_same_bytes_" + tag + "_us"] = round(timed(ref[0], nb, mm) * ref[1], 1)
        for lab, fn in fns.items():
            rec[lab + "_" + tag + "_us"] = timed(fn, nb, mm)
    print(json.dumps(rec), flush=True)


x, y, dy = bufs((8192, 8192), BF, 6), bufs((8192, 8192), BF, 6), bufs((8192, 8192), BF, 6)
case("of_gelu_fwd 8192x8192 bf16", 4 * x[0].numel(), 6, {l: (lambda k, o=o: o.gelu_fwd(x[k], out=y[k])) for l, o in libs.items()},
     (lambda k: y[k].copy_(x[k]), 1.0))
case("of_gelu_bwd 8192x8192 bf16", 6 * x[0].numel(), 6, {l: (lambda k, o=o: o.gelu_bwd(dy[k], x[k], out=y[k])) for l, o in libs.items()},
     (lambda k: y[k].copy_(x[k]), 1.5))
del x, y, dy
x, y = bufs((16448, 4096), BF, 6), bufs((16448, 4096), BF, 6)
case("of_quick_gelu 16448x4096 bf16", 4 * x[0].numel(), 6, {l: (lambda k, o=o: o.quick_gelu(x[k], out=y[k])) for l, o in libs.items()},
     (lambda k: y[k].copy_(x[k]), 1.0))
del x, y
a, b, o32 = bufs((8192, 2048), F32, 24), bufs((8192, 2048), BF, 24), bufs((8192, 2048), F32, 24)
case("of_add_bf16 8192x2048", 10 * a[0].numel(), 24, {l: (lambda k, o=o: o.add_bf16(a[k], b[k], out=o32[k])) for l, o in libs.items()},
     (lambda k: o32[k].copy_(a[k]), 1.25))
