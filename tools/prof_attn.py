"""Launch the MPT-shape causal+ALiBi attention forward/backward a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
ops = Ops.default()
Bm, Hm, Lm, dh = 32, 16, 256, 128
d = Hm * dh
qkv = torch.randn(Bm * Lm, 3 * d, device="cuda").to(torch.bfloat16)
o = torch.empty(Bm * Lm, d, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(Bm, Hm, Lm, device="cuda")
slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / Hm) for i in range(Hm)], device="cuda")
kw = dict(batch=Bm, Lq=Lm, Lk=Lm, heads=Hm, scale=dh ** -0.5, head_dim=dh, causal=True, alibi_slopes=slopes)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv); delta = torch.empty(Bm, Hm, Lm, device="cuda")
for _ in range(3):
    ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, **kw)
    ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], delta, **kw)
torch.cuda.synchronize()
