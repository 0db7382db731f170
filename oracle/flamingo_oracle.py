"""CPU oracle for the OpenFlamingo visual-conditioning hot path.

TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path in
``open_flamingo_amd``; it is never the thing that is shipped or measured.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path (``open_flamingo_amd.src.helpers``)
never imports anything from ``oracle/`` and raises if the HIP library is missing.

It is a plain-PyTorch (CPU, fp32 or fp64) restatement of the arithmetic in the
reference file ``open_flamingo/src/helpers.py`` (all ``file:line`` below are
relative to the reference checkout), written as functions over a flat
``{state_dict_name: tensor}`` mapping so that the reference's own
``state_dict()`` can be fed in unchanged.

Parity pin: ``tests/golden/make_golden.py`` imports the *real* reference modules
(in the authoring container, where ``/root/reference`` exists), runs them on
seeded inputs and commits inputs/outputs/gradients under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function here against those files.
The reference itself ships no tests for this path.

``quant`` hook: every function takes ``quant`` (default: identity).  With the
identity the functions are exactly the reference's fp32 semantics.  The GPU
parity tests also call them with ``quant=bf16_round`` which rounds a tensor to
bfloat16 at precisely the points where the HIP pipeline stores a bf16
intermediate (GEMM operands); that turns the oracle into a rounding-point
emulation of the HIP path so the comparison tolerance can be ~1e-3 instead of
the ~2e-2 a bf16-vs-fp32 comparison needs.
"""

from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]
Quant = Callable[[Tensor], Tensor]


def identity(t: Tensor) -> Tensor:
    return t


def bf16_round(t: Tensor) -> Tensor:
    """Round-to-nearest-even to bfloat16 and back (straight-through for autograd)."""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _sub(p: Params, prefix: str) -> Params:
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def _layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    # nn.LayerNorm(dim) default eps=1e-5, affine (helpers.py:18,33-34,105,152)
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def _linear(x: Tensor, w: Tensor, quant: Quant) -> Tensor:
    # nn.Linear(..., bias=False): y = x @ W^T (helpers.py:19,21,36-38,154-156)
    return quant(x) @ quant(w).transpose(-1, -2)


def feed_forward(x: Tensor, p: Params, quant: Quant = identity) -> Tensor:
    """FeedForward(dim, mult) = Sequential(LN, Linear, GELU(erf), Linear) -- helpers.py:15-22.

    ``p`` keys: ``0.weight 0.bias 1.weight 3.weight`` (Sequential indices).
    """
    u = _layer_norm(x, p["0.weight"], p["0.bias"])
    a = _linear(u, p["1.weight"], quant)
    h = F.gelu(a)  # nn.GELU() default = exact erf form
    return _linear(h, p["3.weight"], quant)


def _split_heads(t: Tensor, heads: int) -> Tensor:
    # "... n (h d) -> ... h n d"
    *lead, n, hd = t.shape
    return t.reshape(*lead, n, heads, hd // heads).transpose(-2, -3)


def _merge_heads(t: Tensor) -> Tensor:
    # "... h n d -> ... n (h d)"
    *lead, h, n, d = t.shape
    return t.transpose(-2, -3).reshape(*lead, n, h * d)


def perceiver_attention(x: Tensor, latents: Tensor, p: Params, heads: int = 8,
                        quant: Quant = identity) -> Tensor:
    """PerceiverAttention.forward -- helpers.py:39-65.

    x (b,T,n1,D) media features, latents (b,T,n2,D).  Keys/values are computed
    from cat(LN_media(x), LN_latents(latents)) (helpers.py:53-54); the amax
    subtraction (helpers.py:60) is detached so it is a no-op for the value and
    the gradient of softmax.
    """
    xm = _layer_norm(x, p["norm_media.weight"], p["norm_media.bias"])
    lt = _layer_norm(latents, p["norm_latents.weight"], p["norm_latents.bias"])
    q = _linear(lt, p["to_q.weight"], quant)
    kv_in = torch.cat((xm, lt), dim=-2)
    kv = _linear(kv_in, p["to_kv.weight"], quant)
    inner = kv.shape[-1] // 2
    k, v = kv[..., :inner], kv[..., inner:]
    dh = inner // heads
    # rearrange "b t n (h d) -> b h t n d"; equivalent per-(b,t) head split
    qh, kh, vh = (_split_heads(quant(t), heads) for t in (q, k, v))  # (b,T,h,n,d)
    qh = qh * dh ** -0.5
    sim = qh @ kh.transpose(-1, -2)
    sim = sim - sim.amax(dim=-1, keepdim=True).detach()
    attn = sim.softmax(dim=-1)
    out = quant(attn) @ vh
    out = _merge_heads(out)
    return _linear(out, p["to_out.weight"], quant)


def perceiver_resampler(x: Tensor, p: Params, heads: int = 8, quant: Quant = identity) -> Tensor:
    """PerceiverResampler.forward -- helpers.py:107-132.

    x (b,T,F,v,D) -> (b,T,n,D).  Optional ``frame_embs`` (helpers.py:117-119) and
    ``media_time_embs`` (helpers.py:123-124) are applied when present in ``p``.
    """
    b, T, Fr, v = x.shape[:4]
    if "frame_embs" in p:
        x = x + p["frame_embs"][:Fr].reshape(1, 1, Fr, 1, -1)
    x = x.reshape(b, T, Fr * v, x.shape[-1])
    if "media_time_embs" in p:
        x = x + p["media_time_embs"][:T]
    latents = p["latents"].unsqueeze(0).unsqueeze(0).expand(b, T, -1, -1)
    depth = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("layers."))
    for i in range(depth):
        latents = perceiver_attention(x, latents, _sub(p, f"layers.{i}.0."), heads, quant) + latents
        latents = feed_forward(latents, _sub(p, f"layers.{i}.1."), quant) + latents
    return _layer_norm(latents, p["norm.weight"], p["norm.bias"])


def text_time_from_locations(media_locations: Tensor, t_txt: int, use_cached_media: bool) -> Tensor:
    """helpers.py:199-208: running count of <image> tokens (or the total count,
    broadcast over the new tokens, in the cached-media decode branch)."""
    if use_cached_media:
        return torch.count_nonzero(media_locations, dim=1).unsqueeze(1).expand(-1, t_txt)
    return media_locations.cumsum(dim=-1)


def masked_cross_attention(x: Tensor, media: Tensor, media_locations: Optional[Tensor], p: Params,
                           heads: int = 8, only_attend_immediate_media: bool = True,
                           use_cached_media: bool = False, quant: Quant = identity) -> Tensor:
    """MaskedCrossAttention.forward -- helpers.py:160-233.

    x (B,T_txt,d); media (B,T_img,n,D); media_locations (B,T_txt) bool.
    Quirks kept on purpose: rows with text_time==0 are
    zeroed *after* softmax (helpers.py:223-229); rows whose text_time exceeds
    T_img have every key masked with -finfo.max and therefore attend uniformly
    to all T_img*n keys (helpers.py:218-221).
    """
    if not use_cached_media:
        assert media_locations.shape[1] == x.shape[1], (
            f"media_location.shape is {media_locations.shape} but x.shape is {x.shape}")
    t_txt = x.shape[1]
    _, t_img, n = media.shape[:3]
    xn = _layer_norm(x, p["norm.weight"], p["norm.bias"])
    q = _linear(xn, p["to_q.weight"], quant)
    med = media.reshape(media.shape[0], t_img * n, media.shape[-1])
    kv = _linear(med, p["to_kv.weight"], quant)
    inner = kv.shape[-1] // 2
    dh = inner // heads
    qh, kh, vh = (_split_heads(quant(t), heads) for t in (q, kv[..., :inner], kv[..., inner:]))
    qh = qh * dh ** -0.5
    sim = qh @ kh.transpose(-1, -2)  # (B,h,T_txt,T_img*n)
    text_time = None
    if media_locations is not None:
        media_time = torch.arange(t_img, device=x.device) + 1
        text_time = text_time_from_locations(media_locations, t_txt, use_cached_media)
        key_time = media_time.repeat_interleave(n)  # "j -> (j n)"
        tt = text_time[:, None, :, None]
        keep = (tt == key_time) if only_attend_immediate_media else (tt >= key_time)
        sim = sim.masked_fill(~keep, -torch.finfo(sim.dtype).max)
    sim = sim - sim.amax(dim=-1, keepdim=True).detach()
    attn = sim.softmax(dim=-1)
    if media_locations is not None and only_attend_immediate_media:
        attn = attn.masked_fill((text_time == 0)[:, None, :, None], 0.0)
    out = _merge_heads(quant(attn) @ vh)
    return _linear(out, p["to_out.weight"], quant)


def gated_cross_attention_block(x: Tensor, media: Tensor, media_locations: Optional[Tensor], p: Params,
                                heads: int = 8, only_attend_immediate_media: bool = True,
                                use_cached_media: bool = False, quant: Quant = identity) -> Tensor:
    """GatedCrossAttentionBlock.forward -- helpers.py:260-279."""
    a = masked_cross_attention(x, media, media_locations, _sub(p, "attn."), heads,
                               only_attend_immediate_media, use_cached_media, quant)
    x = a * p["attn_gate"].tanh() + x
    f = feed_forward(x, _sub(p, "ff."), quant)
    return f * p["ff_gate"].tanh() + x


# ----------------------------------------------------------------------------------------------
# nn.Module shells with the reference's parameter names (state_dict-compatible).  Used by tests to
# (a) check state-dict round trips against the HIP modules and (b) stand in for the hot-path
# modules when exercising the host-side Flamingo control flow on CPU.
# ----------------------------------------------------------------------------------------------

def _ff_module(dim: int, mult: int = 4) -> torch.nn.Sequential:
    inner = int(dim * mult)
    return torch.nn.Sequential(torch.nn.LayerNorm(dim), torch.nn.Linear(dim, inner, bias=False),
                               torch.nn.GELU(), torch.nn.Linear(inner, dim, bias=False))


class _PAttnParams(torch.nn.Module):
    def __init__(self, dim, dim_head, heads):
        super().__init__()
        inner = dim_head * heads
        self.norm_media = torch.nn.LayerNorm(dim)
        self.norm_latents = torch.nn.LayerNorm(dim)
        self.to_q = torch.nn.Linear(dim, inner, bias=False)
        self.to_kv = torch.nn.Linear(dim, inner * 2, bias=False)
        self.to_out = torch.nn.Linear(inner, dim, bias=False)


class OraclePerceiverResampler(torch.nn.Module):
    """Same constructor and state_dict as helpers.py:68-105; forward via perceiver_resampler()."""

    def __init__(self, *, dim, depth=6, dim_head=64, heads=8, num_latents=64,
                 max_num_media=None, max_num_frames=None, ff_mult=4):
        super().__init__()
        self.heads = heads
        self.latents = torch.nn.Parameter(torch.randn(num_latents, dim))
        self.frame_embs = (torch.nn.Parameter(torch.randn(max_num_frames, dim))
                           if max_num_frames is not None else None)
        self.media_time_embs = (torch.nn.Parameter(torch.randn(max_num_media, 1, dim))
                                if max_num_media is not None else None)
        self.layers = torch.nn.ModuleList([
            torch.nn.ModuleList([_PAttnParams(dim, dim_head, heads), _ff_module(dim, ff_mult)])
            for _ in range(depth)])
        self.norm = torch.nn.LayerNorm(dim)

    def forward(self, x, quant: Quant = identity):
        return perceiver_resampler(x, dict(self.named_parameters()), self.heads, quant)


class _XAttnParams(torch.nn.Module):
    def __init__(self, dim, dim_visual, dim_head, heads):
        super().__init__()
        inner = dim_head * heads
        self.norm = torch.nn.LayerNorm(dim)
        self.to_q = torch.nn.Linear(dim, inner, bias=False)
        self.to_kv = torch.nn.Linear(dim_visual, inner * 2, bias=False)
        self.to_out = torch.nn.Linear(inner, dim, bias=False)


class OracleGatedCrossAttentionBlock(torch.nn.Module):
    """Same constructor and state_dict as helpers.py:236-258; forward via gated_cross_attention_block()."""

    def __init__(self, *, dim, dim_visual, dim_head=64, heads=8, ff_mult=4,
                 only_attend_immediate_media=True):
        super().__init__()
        self.heads = heads
        self.only_attend_immediate_media = only_attend_immediate_media
        self.attn = _XAttnParams(dim, dim_visual, dim_head, heads)
        self.attn_gate = torch.nn.Parameter(torch.tensor([0.0]))
        self.ff = _ff_module(dim, ff_mult)
        self.ff_gate = torch.nn.Parameter(torch.tensor([0.0]))

    def forward(self, x, media, media_locations=None, use_cached_media=False, quant: Quant = identity):
        return gated_cross_attention_block(x, media, media_locations, dict(self.named_parameters()),
                                           self.heads, self.only_attend_immediate_media,
                                           use_cached_media, quant)


def seeded_state(shapes: Dict[str, tuple], seed: int, dtype=torch.float32) -> Params:
    """Deterministic weights shared by the golden generator and the tests: N(0,1)/sqrt(fan_in) for
    matrices, 1+0.1*N(0,1) for LayerNorm weights, 0.1*N(0,1) for biases, N(0,1) for latents, gates
    drawn in (-1,1).  Independent of nn.init so both sides of a comparison can rebuild it."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in shapes.items():
        t = torch.randn(*shp, generator=g, dtype=torch.float64)
        if name.endswith("_gate"):
            t = torch.tanh(t) * 0.9
        elif name.endswith("bias"):
            t = 0.1 * t
        elif len(shp) == 1:
            t = 1.0 + 0.1 * t
        elif name.endswith("latents") or "embs" in name:
            pass
        else:
            t = t / math.sqrt(shp[-1])
        out[name] = t.to(dtype)
    return out
