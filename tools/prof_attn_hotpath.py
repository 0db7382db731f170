"""Launch the hot-path attention cores at cfg-2 shapes (for rocprofv3 --pmc): gated cross-attention core
(B=32, L=256, T=2, n=64, 8 heads) and Perceiver core (64 media, 64 latents x 320 keys, 8 heads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip.ops import Ops
ops = Ops.default(); dev = "cuda"
B_, L, T, n, H = 32, 256, 2, 64, 8
q = torch.randn(B_ * L, H * 64, device=dev).to(torch.bfloat16)
kv = torch.randn(B_ * T * n, 2 * H * 64, device=dev).to(torch.bfloat16)
o = torch.empty_like(q); lse = torch.empty(B_, H, L, device=dev)
ml = torch.zeros(B_, L, dtype=torch.uint8, device=dev); ml[:, 0] = 1; ml[:, L // 2] = 1
tt = torch.empty(B_, L, dtype=torch.int32, device=dev); ops.text_time(ml, tt, L, False)
kw = dict(batch=B_, Lq=L, Lk=T * n, heads=H, text_time=tt, n_per_media=n, T_img=T)
do = torch.randn_like(q); dq, dkv = torch.empty_like(q), torch.empty_like(kv); delta = torch.empty(B_, H, L, device=dev)
for _ in range(2):
    ops.attn_fwd(q, kv[:, :512], kv[:, 512:], o, lse, **kw)
    ops.attn_bwd(q, kv[:, :512], kv[:, 512:], o, lse, do, dq, dkv[:, :512], dkv[:, 512:], delta, **kw)
N = 64
q = torch.randn(N * 64, 512, device=dev).to(torch.bfloat16)
kv = torch.randn(N * 320, 1024, device=dev).to(torch.bfloat16)
o = torch.empty_like(q); lse = torch.empty(N, H, 64, device=dev)
kw = dict(batch=N, Lq=64, Lk=320, heads=H)
do = torch.randn_like(q); dq, dkv = torch.empty_like(q), torch.empty_like(kv); delta = torch.empty(N, H, 64, device=dev)
for _ in range(2):
    ops.attn_fwd(q, kv[:, :512], kv[:, 512:], o, lse, **kw)
    ops.attn_bwd(q, kv[:, :512], kv[:, 512:], o, lse, do, dq, dkv[:, :512], dkv[:, 512:], delta, **kw)
torch.cuda.synchronize()
