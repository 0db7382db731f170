"""Launch the frozen-tower self-attention kernels at cfg-2 shapes (for rocprofv3): MPT-1B causal + ALiBi (B=32, 16 heads,
head dim 128, L=256; forward resident + tiled, backward) and CLIP ViT-L/14 (64 images, 16 heads, 257 tokens, head dim 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
ops = Ops.default(); dev = "cuda"
Bm, Hm, Lm, dh = 32, 16, 256, 128
d = Hm * dh
qkv = torch.randn(Bm * Lm, 3 * d, device=dev).to(torch.bfloat16)
o = torch.empty(Bm * Lm, d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(Bm, Hm, Lm, device=dev)
slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / Hm) for i in range(Hm)], device=dev)
kw = dict(batch=Bm, Lq=Lm, Lk=Lm, heads=Hm, scale=dh ** -0.5, head_dim=dh, causal=True, alibi_slopes=slopes)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv); delta = torch.empty(Bm, Hm, Lm, device=dev)
Nv, Hv, Sv = 64, 16, 257
qkv2 = torch.randn(Nv * Sv, 3 * 1024, device=dev).to(torch.bfloat16)
ov = torch.empty(Nv * Sv, 1024, device=dev, dtype=torch.bfloat16)
lsev = torch.empty(Nv, Hv, Sv, device=dev)
for _ in range(3):
    for safe in (0, 2):
        ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, safe=safe, **kw)
        ops.attn_fwd(qkv2[:, :1024], qkv2[:, 1024:2048], qkv2[:, 2048:], ov, lsev, batch=Nv, Lq=Sv, Lk=Sv, heads=Hv, safe=safe)
    for safe in (0, 2):       # 0: of_attn_bwd's own choice (the single pass, csrc/attn_bwd_res.hip), 2: the two-pass kernels
        ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], delta, safe=safe, **kw)
torch.cuda.synchronize()
