"""of_attn_fwd's two forms -- one workgroup per (batch, head) with K / V resident in LDS (safe = 3) against one per 64-query tile (safe = 2) --
and its own choice (safe = 0) at head counts that are whole and ragged rounds of the 256 CUs (causal + ALiBi, 256 x 256, head 128: the frozen
MPT blocks of OF-3B: 32 x 16 = 512 heads, of OF-9B: 10 x 32 = 320), behind a 512-MB copy.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_flamingo_amd.hip.ops import Ops, BF16

ops = Ops.default()
dev = "cuda"
NSETS = 4
filler_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
filler_b = torch.empty_like(filler_a)


def timed(fn):
    ts = []
    for i in range(NSETS * 6):
        filler_b.copy_(filler_a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[NSETS:])
    return round(ts[len(ts) // 2], 1)


for B, H, L, dh in ((32, 16, 256, 128), (10, 32, 256, 128), (12, 32, 256, 128), (6, 32, 256, 128), (20, 16, 256, 128), (64, 16, 257, 64)):
    d = H * dh
    causal = L == 256
    sets = [(torch.randn(B * L, 3 * d, device=dev).to(BF16), torch.empty(B * L, d, device=dev, dtype=BF16), torch.empty(B, H, L, device=dev)) for _ in range(NSETS)]
    kw = dict(batch=B, Lq=L, Lk=L, heads=H, head_dim=dh, causal=causal, scale=dh ** -0.5)
    if causal:
        kw["alibi_slopes"] = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device=dev)
    rec = {"probe": "attn_fwd_forms", "batch": B, "heads": H, "L": L, "head_dim": dh, "bh": B * H, "rounds_of_256": round(B * H / 256, 2)}
    for name, safe in (("auto", 0), ("tiled", 2), ("resident", 3)):
        def f(i, safe=safe):
            qkv, o, lse = sets[i % NSETS]
            ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, safe=safe, **kw)
        f(0)
        rec[name + "_us"] = timed(f)
    print(json.dumps(rec), flush=True)
