"""A GEMM launch timed the way a train step runs it -- right behind HBM-bound kernels (LayerNorm, element-wise passes), not in a hot loop
of itself: HIP events around ONE launch, a streaming filler of a chosen length in front of every launch, median of 40.  The chip is power
managed: a hot loop of MFMA-bound launches settles at the sustained clock, a launch that follows a streaming pass starts with headroom.
Arms: of_gemm's own selection (the 256x256 kernel with the K rotation of round 5), that kernel in plain stage order (OfGemmArgs.safe = 16),
the two-workgroups-per-CU 256x128 kernel, the vendor library (torch.mm).  IL_MAPS = tile-walk / rotation knobs of the tools build
(gemm_w4m.hip: w4m_rotation_knob; 0xf000 = no rotation), IL_TOUCH = 1 reads both operands once between filler and launch.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import make, timed

MAPS = [int(x, 0) for x in os.environ["IL_MAPS"].split(",")] if os.environ.get("IL_MAPS") else None      # tile-walk knobs (tools build)
if MAPS:
    import ctypes
    from tools_lib import tools_ops
    ops = tools_ops()
    ops.lib.of_tools_set_w4m_map_knob.argtypes = [ctypes.c_int]
else:
    from tools_lib import routed_ops
    ops = routed_ops()          # product library; the forced arms (safe >= 2) run on the tools build of the same sources
E = abi
CASES = [("NT store", 8192, 8192, 2048, 0, 0, E.EPI_STORE_BF16), ("NT store K=8192", 8192, 2048, 8192, 0, 0, E.EPI_STORE_BF16),
         ("NT Wqkv", 8192, 6144, 2048, 0, 0, E.EPI_STORE_BF16), ("NN dX K=8192", 8192, 2048, 8192, 0, 1, E.EPI_STORE_BF16),
         ("NT gelu", 8192, 8192, 2048, 0, 0, E.EPI_GELU), ("TN dW", 2048, 8192, 8192, 1, 1, E.EPI_ACC_F32),
         ("NN dgelu_dot", 8192, 8192, 2048, 0, 1, E.EPI_DGELU_DOT), ("NT gate_resid K=8192", 8192, 2048, 8192, 0, 0, E.EPI_GATE_RESID),
         ("small to_q NT", 8192, 512, 2048, 0, 0, E.EPI_STORE_BF16), ("small to_out gate_resid", 8192, 2048, 512, 0, 0, E.EPI_GATE_RESID),
         ("small to_out dX scale_dot", 8192, 512, 2048, 0, 1, E.EPI_SCALE_DOT), ("small to_q dX", 8192, 2048, 512, 0, 1, E.EPI_STORE_BF16),
         ("small to_q dW", 512, 2048, 8192, 1, 1, E.EPI_ACC_F32), ("small perceiver to_kv", 20480, 1024, 1024, 0, 0, E.EPI_STORE_BF16),
         ("small perceiver ffn up gelu", 4096, 4096, 1024, 0, 0, E.EPI_GELU), ("small perceiver ffn down", 4096, 1024, 4096, 0, 0, E.EPI_GATE_RESID),
         ("small perceiver to_kv dX", 20480, 1024, 1024, 0, 1, E.EPI_STORE_BF16)]
if os.environ.get("IL_CASES"):
    CASES = [c for c in CASES if any(k in c[0] for k in os.environ["IL_CASES"].split(","))]
FILL = torch.randn(1 << 28, device="cuda").to(torch.bfloat16)          # 256 Mi elements: 512 MB in, 512 MB out at full length
FOUT = torch.empty_like(FILL)
FILLERS = {"hot loop": 0, "50 us": 1 << 26, "200 us": 1 << 28}
if os.environ.get("IL_FILLERS"):
    FILLERS = {k: v for k, v in FILLERS.items() if any(x in k for x in os.environ["IL_FILLERS"].split(","))}
SAFE = {"auto": 0, "w4m256 stage order": 16, "w4h256x128": 18}      # auto: of_gemm's own selection (the 256x256 kernel with the K rotation)
if os.environ.get("IL_SAFE"):          # extra arms: name=safe,...
    for kv in os.environ["IL_SAFE"].split(","):
        k, v = kv.split("=")
        SAFE[k] = int(v)


TOUCH = os.environ.get("IL_TOUCH") == "1"      # read both operands once between the filler and the launch (they are then in the Infinity Cache)


def median_after_filler(fn, n_fill, reps=40, touch=()):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        if n_fill:
            ops.gelu_fwd(FILL[:n_fill], out=FOUT[:n_fill])
        for t in touch:
            t.view(torch.int16).max()
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev[4:])
    return t[len(t) // 2]


for name, M, N, K, ta, tb, epi in CASES:
    A, B, C, kw = make(M, N, K, ta, tb, epi)
    fns = {}
    for label, safe in SAFE.items():
        if label == "w4h256x128" and ta:
            continue
        if name.startswith("small") and safe and not os.environ.get("IL_SMALL_FORCED"):
            continue
        fns[label] = (lambda safe=safe: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=safe, **kw))
    if MAPS:
        fns = {}
        for knob in MAPS:
            def fn(knob=knob):
                ops.lib.of_tools_set_w4m_map_knob(knob)
                ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=0, **kw)
            fns["w4m256 map 0x%x" % knob] = fn
    if epi == E.EPI_STORE_BF16:
        At, Bt = (A.t() if ta else A), (B if tb else B.t())
        fns["vendor"] = lambda: torch.mm(At, Bt, out=C)
    for fn in fns.values():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    rec = dict(case=name, MNK=[M, N, K])
    if MAPS:          # every walk / rotation gives the same product (another order of the fp32 additions: close, not equal)
        ops.lib.of_tools_set_w4m_map_knob(0xf000)
        ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=0, **kw)
        want = C.float().clone()
        worst = 0.0
        for k, fn in fns.items():
            if k.startswith("w4m256 map"):
                C.zero_()
                fn()
                worst = max(worst, float((C.float() - want).abs().max() / want.abs().max()))
        rec["worst_rel_diff_to_stage_order"] = worst
    for fname, n_fill in FILLERS.items():
        for rnd in range(2):
            for k, fn in fns.items():
                us = median_after_filler(fn, n_fill) * 1e3
                key = "%s | %s" % (k, fname)
                rec[key] = round(min(rec.get(key, 1e9), us), 1)
                if TOUCH and n_fill:
                    us = median_after_filler(fn, n_fill, touch=(A, B)) * 1e3
                    key += " + operands read once"
                    rec[key] = round(min(rec.get(key, 1e9), us), 1)
    print(json.dumps(rec), flush=True)
    del A, B, C, kw
