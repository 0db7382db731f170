"""Does the big-tile GEMM speed up when the operands draw less power?  The chip is power managed: the MFMA clock a launch gets depends on how
many operand bits toggle.  NT launches of the frozen MLP shapes in a hot loop on operands that are random normal (every probe of rounds 1-5),
all zero (the floor of the power draw), half zero, and normal with the weights' real scale (0.02) -- of_gemm's kernel against the vendor
library's.  If the vendor's time falls with the power draw and this library's does not, the K loop has a limiter right behind the power cap
that hot loops on random operands never showed.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_flamingo_amd.hip.ops import Ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools_lib import routed_ops      # product library; kernel-forcing selectors (safe >= 2) -> tools/libofhip_tools.so

ops = routed_ops()
bf = torch.bfloat16


def timed(fn, iters=20, rounds=4):
    best = 1e9
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return round(best, 1)


def data(kind, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(N, K, device="cuda", generator=g)
    if kind == "zeros":
        A.zero_(); B.zero_()
    elif kind == "half zero":
        A *= (torch.rand(M, K, device="cuda", generator=g) < 0.5)
        B *= (torch.rand(N, K, device="cuda", generator=g) < 0.5)
    elif kind == "normal, weights x 0.02":
        B *= 0.02
    elif kind == "normal x normal 0.05":
        B *= 0.05
    elif kind == "one value":
        A.fill_(1.0); B.fill_(0.5)
    return A.to(bf), B.to(bf)


for name, M, N, K in (("up_proj", 8192, 8192, 2048), ("down_proj", 8192, 2048, 8192)):
    C = torch.empty(M, N, device="cuda", dtype=bf)
    for kind in ("normal x normal 0.05", "normal, weights x 0.02", "half zero", "one value", "zeros", "normal x normal 0.05"):
        A, B = data(kind, M, N, K)
        Bt = B.t()
        for fn in (lambda: ops.gemm(A, B, C), lambda: torch.mm(A, Bt, out=C)):
            for _ in range(5):
                fn()
        torch.cuda.synchronize()
        rec = dict(case=name, operands=kind, ours_us=timed(lambda: ops.gemm(A, B, C)), vendor_us=timed(lambda: torch.mm(A, Bt, out=C)),
                   ours_stage_order_us=timed(lambda: ops.gemm(A, B, C, safe=16)), w4h_us=timed(lambda: ops.gemm(A, B, C, safe=18)))
        print(json.dumps(rec), flush=True)
