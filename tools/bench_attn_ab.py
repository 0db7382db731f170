"""Same-box A/B of two libofhip builds on the step's attention launches (HIP events, interleaved rounds, random operands):
frozen MPT block (causal + ALiBi, head 128), CLIP layer (head 64, 257 tokens), gated cross-attention (64-key windows), Perceiver.
Outputs of both builds are compared.  PROFILING TOOL.

    python tools/bench_attn_ab.py old.so"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import load, timed

old, new = load(sys.argv[1]), Ops.default()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
def r(*s): return torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)

cases = [("frozen MPT causal+alibi", dict(batch=32, Lq=256, Lk=256, heads=16, head_dim=128, causal=True, scale=128 ** -0.5), True),
         ("CLIP layer", dict(batch=64, Lq=257, Lk=257, heads=16, head_dim=64, scale=0.125), False),
         ("gated xattn (T=2, 64 latents)", dict(batch=32, Lq=256, Lk=128, heads=8, head_dim=64, scale=0.125, n_per_media=64, T_img=2, only_immediate=True), "tt"),
         ("perceiver (64 latents over 257+64)", dict(batch=64, Lq=64, Lk=321, heads=8, head_dim=64, scale=0.125), False)]
for name, kw, extra in cases:
    B, Lq, Lk, H, dh = kw["batch"], kw["Lq"], kw["Lk"], kw["heads"], kw["head_dim"]
    q, k, v, do = r(B * Lq, H * dh), r(B * Lk, H * dh), r(B * Lk, H * dh), r(B * Lq, H * dh)
    if extra is True:
        kw["alibi_slopes"] = torch.linspace(0.5, 0.01, H, device=dev)
    if extra == "tt":
        tt = torch.zeros(B, Lq, dtype=torch.int32, device=dev)
        tt[:, 16:] = 1
        tt[:, 140:] = 2
        kw["text_time"] = tt
    res = {}
    fns = {}
    for lab, ops in (("old", old), ("new", new)):
        o, lse = torch.zeros(B * Lq, H * dh, device=dev, dtype=torch.bfloat16), torch.zeros(B, H, Lq, device=dev)
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        delta = torch.zeros(B, H, Lq, device=dev)
        ops.attn_fwd(q, k, v, o, lse, **kw)
        ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, delta, **kw)
        torch.cuda.synchronize()
        res[lab] = (o.clone(), dq.clone(), dk.clone(), dv.clone())
        fns[lab + "_fwd"] = (lambda ops=ops, o=o, lse=lse: ops.attn_fwd(q, k, v, o, lse, **kw))
        fns[lab + "_bwd"] = (lambda ops=ops, o=o, lse=lse, dq=dq, dk=dk, dv=dv, delta=delta: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, delta, **kw))
    diff = max(float((a.float() - b.float()).abs().max()) for a, b in zip(res["old"], res["new"]))
    best = {k2: 1e9 for k2 in fns}
    for _ in range(4):
        for k2, fn in fns.items():
            best[k2] = min(best[k2], timed(fn, 10))
    print(json.dumps(dict(case=name, max_abs_diff_old_new=diff, **{k2 + "_us": round(v2 * 1e3, 1) for k2, v2 in best.items()})), flush=True)
