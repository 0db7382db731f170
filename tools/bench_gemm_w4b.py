"""DMA-placement A/B of the 4-wave big-tile kernel + K sweep (fixed per-tile cost vs per-stage cost), random operands,
interleaved rounds.  Needs tools/libofhip_tools.so (safe = 71..73 are tools-only variants)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.tools_lib import tools_ops
ops = tools_ops()
dev = "cuda"


def r(shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def run(name, M, N, K, ta, tb, arms_sel):
    A = r((K, M) if ta else (M, K))
    B = r((K, N) if tb else (N, K), K ** -0.5)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Am, Bm = (A.t() if ta else A), (B if tb else B.t())
    arms = {"pp": 4, "w4": 6, "dma0": 70, "dma1": 71, "dma2": 7, "dma3": 73}
    fns = {k: (lambda sf=sf: ops.gemm(A, B, out, ta=ta, tb=tb, safe=sf)) for k, sf in arms.items() if k in arms_sel}
    fns["blaslt"] = lambda: torch.matmul(Am, Bm)
    ref = torch.empty_like(out)
    ops.gemm(A, B, ref, ta=ta, tb=tb, safe=4)
    same = {}
    for k, sf in arms.items():
        if k in arms_sel and k != "pp":
            o = torch.empty_like(out)
            for _ in range(3):
                ops.gemm(A, B, o, ta=ta, tb=tb, safe=sf)
                same[k] = same.get(k, True) and bool(torch.equal(o, ref))
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(5):
        for k, f in fns.items():
            ts[k].append(timed(f))
    res = {"case": name, "MNK": [M, N, K], "bit_identical_to_pp": same}
    for k, v in ts.items():
        v.sort()
        res[k] = [round(v[len(v) // 2], 4), round(2.0 * M * N * K / v[len(v) // 2] / 1e9, 1)]
    print(json.dumps(res), flush=True)


ALL = ("pp", "w4", "dma0", "dma1", "dma2", "dma3")
run("NT", 8192, 2048, 8192, False, False, ALL)
run("NN", 8192, 2048, 8192, False, True, ALL)
run("TN", 8192, 2048, 8192, True, True, ALL)
run("NT sq", 8192, 8192, 8192, False, False, ALL)
for K in (256, 512, 1024, 2048, 4096):
    run("NT ksweep", 8192, 8192, K, False, False, ("pp", "dma0", "dma3"))
for K in (1024, 2048, 4096):
    run("NT ksweep N=2048", 8192, 2048, K, False, False, ("pp", "dma0"))
