"""The attention section of a frozen GPT-NeoX block at BASELINE config 4's shape (OF-4B = RedPajama-INCITE-3B: B 16, L 256, 32 heads x 80,
causal, rotary) with COMPACT heads (OfAttnArgs.head_valid = 80 at the 128-wide kernels, ABI v11) against the zero-padded copies of rounds
2-5: rotary pass, attention forward, (output repack), (dO repack), attention backward, inverse rotary pass -- each timed with HIP events
over rotating buffer sets behind a 512-MB copy (a train step never finds its operands in the Infinity Cache).  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_flamingo_amd.hip.ops import Ops, BF16

ops = Ops.default()
dev = "cuda"
NSETS, REPS = 4, 5
g = torch.Generator(device=dev).manual_seed(4)
filler_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
filler_b = torch.empty_like(filler_a)


def timed(fn):
    ts = []
    for i in range(NSETS * REPS):
        filler_b.copy_(filler_a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[NSETS:])
    return round(ts[len(ts) // 2], 1)


def case(B, L, H, hs, pad):
    rows, d = B * L, H * hs
    cos = torch.randn(L, hs, device=dev, generator=g)
    sin = torch.randn(L, hs, device=dev, generator=g)
    rec = {"probe": "compact_heads", "shape": {"B": B, "L": L, "heads": H, "head_size": hs, "kernel_head_dim": pad}}
    for compact in (True, False):
        hw = hs if compact else pad
        sets = []
        for _ in range(NSETS):
            qkv = torch.randn(rows, 3 * d, device=dev, generator=g).to(BF16)
            do = torch.randn(rows, d, device=dev, generator=g).to(BF16)
            sets.append(dict(qkv=qkv, do=do, qp=torch.empty(3, rows, H * hw, device=dev, dtype=BF16), op=torch.empty(rows, H * hw, device=dev, dtype=BF16),
                             o=torch.empty(rows, d, device=dev, dtype=BF16), dop=torch.empty(rows, H * hw, device=dev, dtype=BF16),
                             dqp=torch.empty(3, rows, H * hw, device=dev, dtype=BF16), dqkv=torch.empty(rows, 3 * d, device=dev, dtype=BF16),
                             lse=torch.empty(B, H, L, device=dev), delta=torch.empty(B, H, L, device=dev)))
        kw = dict(batch=B, Lq=L, Lk=L, heads=H, scale=hs ** -0.5, head_dim=pad, head_valid=0 if hw == pad else hs, causal=True)

        def rot(i):
            s = sets[i % NSETS]
            ops.rotary_neox(s["qkv"], cos, sin, s["qp"][0], s["qp"][1], s["qp"][2], L=L, heads=H, head_size=hs, rot_dims=hs, head_pad=hw)

        def fwd(i):
            s = sets[i % NSETS]
            ops.attn_fwd(s["qp"][0], s["qp"][1], s["qp"][2], s["op"], s["lse"], **kw)

        def unpad(i):
            s = sets[i % NSETS]
            ops.head_repack(s["op"], s["o"], heads=H, src_head_size=hw, dst_head_size=hs)

        def padd(i):
            s = sets[i % NSETS]
            ops.head_repack(s["do"], s["dop"], heads=H, src_head_size=hs, dst_head_size=hw)

        def bwd(i):
            s = sets[i % NSETS]
            ops.attn_bwd(s["qp"][0], s["qp"][1], s["qp"][2], s["op"], s["lse"], s["do"] if compact else s["dop"], s["dqp"][0], s["dqp"][1], s["dqp"][2],
                         s["delta"], **kw)

        def irot(i):
            s = sets[i % NSETS]
            ops.rotary_neox(s["dqkv"], cos, sin, s["dqp"][0], s["dqp"][1], s["dqp"][2], L=L, heads=H, head_size=hs, rot_dims=hs, head_pad=hw, inverse=True)

        for i in range(NSETS):
            rot(i), fwd(i)
            if not compact:
                unpad(i), padd(i)
            bwd(i), irot(i)
        torch.cuda.synchronize()
        t = {"rotary": timed(rot), "attn_fwd": timed(fwd), "attn_bwd": timed(bwd), "rotary_inverse": timed(irot)}
        if not compact:
            t["repack_o"], t["repack_dout"] = timed(unpad), timed(padd)
        t["sum_us"] = round(sum(t.values()), 1)
        rec["compact" if compact else "padded"] = t
    print(json.dumps(rec), flush=True)


case(16, 256, 32, 80, 128)       # BASELINE config 4 (bench.py --config 4: B 16, L 256)
case(32, 256, 32, 80, 128)
