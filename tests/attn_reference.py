"""fp64 dense restatement of the attention cores, used by the emulator tests and the GPU parity tests.
Follows helpers.py:55-64 (Perceiver) and helpers.py:192-231 (masked cross attention) on already-projected
q/k/v; the mask is built exactly as the reference builds it (eq / ge against media_time, -finfo.max fill,
post-softmax zeroing)."""
import torch


def dense_attention(q, k, v, heads, text_time=None, n=0, T=0, only_immediate=True, round_p=False, head_dim=64,
                    causal=False, alibi_slopes=None):
    """q (B,Lq,H*dh), k/v (B,Lk,H*dh) float64 -> o (B,Lq,H*dh).  causal / alibi_slopes: the MPT self-attention form
    (HF modeling_mpt.MptAttention: scores * scale + slope_h * (j - (Lk - 1)), causal mask, softmax)."""
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    dh = head_dim
    qh = q.reshape(B, Lq, heads, dh).transpose(1, 2) * dh ** -0.5
    kh = k.reshape(B, Lk, heads, dh).transpose(1, 2)
    vh = v.reshape(B, Lk, heads, dh).transpose(1, 2)
    sim = qh @ kh.transpose(-1, -2)
    if alibi_slopes is not None:
        pos = torch.arange(1 - Lk, 1, dtype=sim.dtype)
        sim = sim + alibi_slopes.to(sim.dtype).view(1, heads, 1, 1) * pos.view(1, 1, 1, Lk)
    if causal:
        i = torch.arange(Lq).view(Lq, 1) + (Lk - Lq)
        sim = sim.masked_fill(torch.arange(Lk).view(1, Lk) > i, -torch.finfo(torch.float32).max)
    if text_time is not None:
        key_time = (torch.arange(T) + 1).repeat_interleave(n)
        tt = text_time[:, None, :, None]
        keep = (tt == key_time) if only_immediate else (tt >= key_time)
        sim = sim.masked_fill(~keep, -torch.finfo(torch.float32).max)
    sim = sim - sim.amax(dim=-1, keepdim=True).detach()
    attn = sim.softmax(dim=-1)
    if text_time is not None and only_immediate:
        attn = attn.masked_fill((text_time == 0)[:, None, :, None], 0.0)
    if round_p:
        attn = attn + (attn.detach().to(torch.bfloat16).to(attn.dtype) - attn.detach())
    out = attn @ vh
    return out.transpose(1, 2).reshape(B, Lq, heads * dh)
