"""The frozen MPT blocks' attention backward at BASELINE config 2's shape (B 32, L 256, 16 heads x 128, causal + ALiBi; q | k | v views of
the fused Wqkv output, as train/frozen_blocks.py passes them): the two-pass kernels (of_attn_q_kernel + of_attn_dkv_kernel, safe = 2)
against the single pass (csrc/attn_bwd_res.hip, safe = 3), each timed with HIP events over ROTATING buffer sets and behind a 512-MB
copy (a train step never finds its operands in the Infinity Cache), outputs compared.  Also head 64 and a short-sequence case.
PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_flamingo_amd.hip.ops import Ops, BF16

ops = Ops.default()
dev = "cuda"
NSETS, REPS = 6, 5
g = torch.Generator(device=dev).manual_seed(2)
filler_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
filler_b = torch.empty_like(filler_a)


def timed(fn, cold):
    ts = []
    for i in range(NSETS * REPS):
        if cold:
            filler_b.copy_(filler_a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[NSETS:])
    return round(ts[len(ts) // 2], 1)


def case(name, B, L, H, dh, causal=True):
    d = H * dh
    sets = []
    for _ in range(NSETS):
        qkv = torch.randn(B * L, 3 * d, device=dev, generator=g).to(BF16)
        do = torch.randn(B * L, d, device=dev, generator=g).to(BF16)
        o, lse = torch.empty(B * L, d, device=dev, dtype=BF16), torch.empty(B, H, L, device=dev)
        sets.append((qkv, do, o, lse, torch.empty_like(qkv), torch.empty(B, H, L, device=dev)))
    kw = dict(batch=B, Lq=L, Lk=L, heads=H, head_dim=dh, causal=causal, scale=dh ** -0.5)
    if causal:
        kw["alibi_slopes"] = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device=dev)
    for qkv, do, o, lse, dqkv, delta in sets:
        ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, **kw)

    def bwd(i, safe):
        qkv, do, o, lse, dqkv, delta = sets[i % NSETS]
        ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], delta, safe=safe, **kw)

    outs = {}
    for safe in (2, 3):
        sets[0][4].fill_(float("nan"))
        bwd(0, safe)
        torch.cuda.synchronize()
        outs[safe] = sets[0][4].clone()
    a, b = outs[3].float(), outs[2].float()
    rec = {"probe": "attn_bwd_single_pass", "case": name, "shape": {"B": B, "L": L, "heads": H, "head_dim": dh, "causal": causal},
           "finite": bool(torch.isfinite(a).all()), "max_rel_diff_single_vs_two_pass": float((a - b).abs().max() / b.abs().max()),
           "dv_bit_equal": bool(torch.equal(outs[3][:, 2 * d:], outs[2][:, 2 * d:])),
           "algorithmic_MB": round(8 * B * L * d * 2 / 1e6, 1)}
    for cold in (False, True):
        tag = "behind_a_512MB_copy" if cold else "rotating_sets"
        rec[f"two_pass_us_{tag}"] = timed(lambda i: bwd(i, 2), cold)
        rec[f"single_pass_us_{tag}"] = timed(lambda i: bwd(i, 3), cold)
    rec["single_pass_TBps_cold"] = round(rec["algorithmic_MB"] / rec["single_pass_us_behind_a_512MB_copy"], 2)
    print(json.dumps(rec), flush=True)


case("frozen MPT-1B block (OF-3B)", 32, 256, 16, 128)
case("head 64, L 256", 32, 256, 16, 64)
case("head 128, L 128", 64, 128, 16, 128)
case("no mask, head 128, L 256", 32, 256, 16, 128, causal=False)

# ---- where the single pass spends its time: phase stamps of the tools build (wave 0 of every workgroup, 100-MHz wall clock)
if os.environ.get("BR_PHASES", "1") == "1":
    import ctypes
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools_lib import tools_ops
    tops = tools_ops()
    tops.lib.of_tools_set_br_stamp_buffer.argtypes = [ctypes.c_void_p]
    B, L, H, dh = 32, 256, 16, 128
    d = H * dh
    qkv = torch.randn(B * L, 3 * d, device=dev, generator=g).to(BF16)
    do = torch.randn(B * L, d, device=dev, generator=g).to(BF16)
    o, lse = torch.empty(B * L, d, device=dev, dtype=BF16), torch.empty(B, H, L, device=dev)
    dqkv, delta = torch.empty_like(qkv), torch.empty(B, H, L, device=dev)
    kw = dict(batch=B, Lq=L, Lk=L, heads=H, head_dim=dh, causal=True, scale=dh ** -0.5,
              alibi_slopes=torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device=dev))
    tops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, **kw)
    run = lambda: tops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], delta, safe=3, **kw)
    for cold in (False, True):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        buf = torch.zeros(B * H, 12, dtype=torch.int64, device=dev)
        tops.lib.of_tools_set_br_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        if cold:
            filler_b.copy_(filler_a)
        run()
        torch.cuda.synchronize()
        tops.lib.of_tools_set_br_stamp_buffer(None)
        s = buf.cpu().double() / 100.0
        t0 = s[:, 0].min()
        med = lambda v: round(float(v.median()), 2)
        first = s[:, 0] - t0 < 5.0            # workgroups of the first round
        print(json.dumps({"probe": "attn_bwd_single_pass_phases", "cold": cold, "workgroups": B * H, "first_round_workgroups": int(first.sum()),
                          "prologue_us": med(s[:, 1] - s[:, 0]), "prologue_first_round_us": med((s[:, 1] - s[:, 0])[first]),
                          "prologue_second_round_us": med((s[:, 1] - s[:, 0])[~first]) if (~first).any() else None, "sum_over_tiles_us": {"tile_load_issue": med(s[:, 2]), "s_dp_softmax": med(s[:, 3]),
                          "ds_writes_dv_dk": med(s[:, 4]), "stats_and_block_loads_issue": med(s[:, 5]), "wait_dS_barrier": med(s[:, 6]), "phase2_dq": med(s[:, 7]),
                          "wait_loads_lds_writes_stats_dq_store": med(s[:, 8]), "wait_tile_barrier": med(s[:, 9])}, "epilogue_us": med(s[:, 10] - s[:, 11]),
                          "workgroup_us": med(s[:, 10] - s[:, 0]), "second_round_entry_us": med((s[:, 0] - t0)[~first]) if (~first).any() else None,
                          "span_us": round(float((s[:, 10] - t0).max()), 1)}), flush=True)
