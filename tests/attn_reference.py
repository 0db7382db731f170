"""fp64 dense restatement of the attention cores, used by the emulator tests and the GPU parity tests.
Follows helpers.py:55-64 (Perceiver) and helpers.py:192-231 (masked cross attention) on already-projected
q/k/v; the mask is built exactly as the reference builds it (eq / ge against media_time, -finfo.max fill,
post-softmax zeroing)."""
import torch


def dense_attention(q, k, v, heads, text_time=None, n=0, T=0, only_immediate=True, round_p=False):
    """q (B,Lq,H*64), k/v (B,Lk,H*64) float64 -> o (B,Lq,H*64)."""
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    qh = q.reshape(B, Lq, heads, 64).transpose(1, 2) * 64 ** -0.5
    kh = k.reshape(B, Lk, heads, 64).transpose(1, 2)
    vh = v.reshape(B, Lk, heads, 64).transpose(1, 2)
    sim = qh @ kh.transpose(-1, -2)
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
    return out.transpose(1, 2).reshape(B, Lq, heads * 64)
