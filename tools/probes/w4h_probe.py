"""The two-workgroups-per-CU 256 x 128 kernel (csrc/gemm_w4h.hip, OfGemmArgs.safe = 18) against the 256 x 256 kernel (safe = 16)
on the launches of a train step: same box, interleaved rounds, random operands.  tools/libofhip_tools.so (-DOF_TOOLS_BUILD) carries
the scheduling knob of the study build (of_tools_set_w4h_knob: 0 = none, 1 = priority 1 in the second half of a K loop, 2 = priority 1
for odd rounds of workgroups) and the phase stamps.  One JSON line per (case, arm); first a parity / race screen.  PROFILING TOOL."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from tools_lib import tools_ops
from bench_gemm_ab import make, timed

ops = tools_ops()
lib = ops.lib
lib.of_tools_set_w4h_knob.argtypes = [ctypes.c_int]
lib.of_tools_set_w4h_stamp_buffer.argtypes = [ctypes.c_void_p]
E = abi
KNOBS = [int(k) for k in os.environ.get("W4H_KNOBS", "0,1,2").split(",")]
CASES = [("NT store_bf16", 8192, 8192, 2048, 0, 0, E.EPI_STORE_BF16), ("NT gelu 2 outputs", 8192, 8192, 2048, 0, 0, E.EPI_GELU),
         ("NN store_bf16", 8192, 8192, 2048, 0, 1, E.EPI_STORE_BF16), ("NN scale_dot", 8192, 8192, 2048, 0, 1, E.EPI_SCALE_DOT),
         ("NN dgelu_dot", 8192, 8192, 2048, 0, 1, E.EPI_DGELU_DOT), ("NT gate_resid fp32", 8192, 2048, 8192, 0, 0, E.EPI_GATE_RESID),
         ("NN store K=8192", 8192, 2048, 8192, 0, 1, E.EPI_STORE_BF16), ("NT Wqkv store", 8192, 6144, 2048, 0, 0, E.EPI_STORE_BF16),
         ("NT out_proj store", 8192, 2048, 2048, 0, 0, E.EPI_STORE_BF16), ("NT gate_resid K=512", 8192, 2048, 512, 0, 0, E.EPI_GATE_RESID),
         ("NT to_q K=2048 N=512", 8192, 512, 2048, 0, 0, E.EPI_STORE_BF16), ("NN scale_dot N=512", 8192, 512, 2048, 0, 1, E.EPI_SCALE_DOT),
         ("NT store 8192^3", 8192, 8192, 8192, 0, 0, E.EPI_STORE_BF16)]
if os.environ.get("W4H_CASES"):
    keep = os.environ["W4H_CASES"].split(",")
    CASES = [c for c in CASES if any(k in c[0] for k in keep)]
med = lambda t: float(t.double().median())


def outputs(C, kw):
    o = [C.clone()]
    if "out2" in kw:
        o.append(kw["out2"].clone())
    if "dot" in kw:
        o.append(kw["dot"].clone())
    return o


# ---- parity / race screen: the half-tile kernel against the 256 x 256 kernel, bit for bit, three launches each knob
for name, M, N, K, ta, tb, epi in CASES:
    if N % 256:
        continue
    A, B, C, kw = make(M, N, K, ta, tb, epi)
    if "dot" in kw:
        kw["dot"].zero_()
    ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=16, **kw)
    want = outputs(C, kw)
    ok, dot_rel = True, 0.0
    for knob in [k for k in KNOBS if (k >> 4) & 7 < 2]:      # variants 2 / 3 are wrong by design (timing only)
        lib.of_tools_set_w4h_knob(knob)
        for rep in range(3):
            C.zero_()
            if "dot" in kw:
                kw["dot"].zero_()
            if epi == E.EPI_GATE_RESID:
                pass
            ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=18, **kw)
            got = outputs(C, kw)
            nout = len(want) - (1 if "dot" in kw else 0)
            same = all(torch.equal(a, b) for a, b in zip(want[:nout], got[:nout]))
            if "dot" in kw:      # one partial per tile: twice as many, other order -> close, not equal; repeatable
                dot_rel = max(dot_rel, abs(float(got[-1]) - float(want[-1])) / (abs(float(want[-1])) + 1e-30))
            ok = ok and same
    print(json.dumps(dict(parity=name, MNK=[M, N, K], bit_equal_to_256x256=bool(ok), gate_dot_rel_diff=dot_rel)), flush=True)
    del A, B, C, kw

# ---- timing
for name, M, N, K, ta, tb, epi in CASES:
    A, B, C, kw = make(M, N, K, ta, tb, epi)
    arms = {}
    if N % 256 == 0 and (M // 256) * (N // 256) >= 1:
        arms["w4m256"] = (16, 0)
    else:
        arms["auto"] = (0, 0)
    for knob in KNOBS:
        arms["w4h knob%d" % knob] = (18, knob)
    fns = {}
    for label, (safe, knob) in arms.items():
        def fn(safe=safe, knob=knob):
            ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=safe, **kw)
        fns[label] = (fn, knob)
    best = {k: 1e9 for k in fns}
    for k, (fn, knob) in fns.items():
        lib.of_tools_set_w4h_knob(knob)
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for _ in range(4):
        for k, (fn, knob) in fns.items():
            lib.of_tools_set_w4h_knob(knob)
            torch.cuda.synchronize()
            best[k] = min(best[k], timed(fn, 10))
    rec = dict(case=name, MNK=[M, N, K])
    for k, ms in best.items():
        rec[k + "_us"] = round(ms * 1e3, 1)
        rec[k + "_tflops"] = round(2.0 * M * N * K / ms / 1e9, 1)
    print(json.dumps(rec), flush=True)
    # phase stamps of the half-tile kernel (knob of the best arm)
    if os.environ.get("W4H_STAMPS", "1") == "1" and M * N >= 8192 * 2048:
      for bk in [k for k in best if k.startswith("w4h")]:
        knob = fns[bk][1]
        lib.of_tools_set_w4h_knob(knob)
        ntile = (M // 256) * (N // 128)
        buf = torch.zeros(ntile, 8, dtype=torch.int64, device="cuda")
        lib.of_tools_set_w4h_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        fns[bk][0]()
        torch.cuda.synchronize()
        lib.of_tools_set_w4h_stamp_buffer(None)
        s = buf.cpu()
        t = (s[:, :5] - s[:, 0].min()).double() / 100.0
        hw = s[:, 7]
        cu_key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 8) & 0xf)
        # per CU: fraction of the launch's span in which at least one resident workgroup was inside its K loop
        cover, both = [], []
        span = float(t[:, 4].max())
        for key in cu_key.unique().tolist():
            idx = (cu_key == key).nonzero().flatten()
            ev = []
            for i in idx.tolist():
                ev.append((float(t[i, 1]), 1))
                ev.append((float(t[i, 2]), -1))
            ev.sort()
            depth, last, c1, c2 = 0, 0.0, 0.0, 0.0
            for x, dlt in ev:
                if depth >= 1:
                    c1 += x - last
                if depth >= 2:
                    c2 += x - last
                depth += dlt
                last = x
            cover.append(c1 / span)
            both.append(c2 / span)
        cyc = (s[:, 6] - s[:, 5]).double()
        wall = (s[:, 2] - s[:, 1]).double() / 100.0
        print(json.dumps(dict(stamps=name, arm=bk, kloop_cycles_per_stage=round(med(cyc) / (K // 64), 1), kloop_clock_ghz=round(med(cyc / wall) / 1e3, 3), tiles=ntile, cus_seen=int(cu_key.unique().numel()), span_us=round(span, 1),
                              prologue_us=round(med(t[:, 1] - t[:, 0]), 2), kloop_us=round(med(t[:, 2] - t[:, 1]), 2),
                              kloop_us_p10_p90=[round(float((t[:, 2] - t[:, 1]).kthvalue(max(1, ntile // 10)).values), 1),
                                                round(float((t[:, 2] - t[:, 1]).kthvalue(max(1, ntile * 9 // 10)).values), 1)],
                              epilogue_issue_us=round(med(t[:, 3] - t[:, 2]), 2), store_ack_us=round(med(t[:, 4] - t[:, 3]), 2),
                              cu_time_with_a_k_loop_running=round(float(torch.tensor(cover).median()), 3),
                              cu_time_with_two_k_loops_running=round(float(torch.tensor(both).median()), 3))), flush=True)
    del A, B, C, kw
