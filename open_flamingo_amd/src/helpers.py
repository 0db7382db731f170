"""Drop-in modules of the visual-conditioning hot path, computed by libofhip (gfx950) -- no PyTorch math.

Mirrors the public surface of the reference ``open_flamingo/src/helpers.py``: the same class names, constructor
signatures, parameter names/shapes (so reference checkpoints load with ``strict=True``) and forward signatures:

  PerceiverResampler(*, dim, depth=6, dim_head=64, heads=8, num_latents=64, max_num_media=None,
                     max_num_frames=None, ff_mult=4)              reference helpers.py:68-132
  GatedCrossAttentionBlock(*, dim, dim_visual, dim_head=64, heads=8, ff_mult=4,
                           only_attend_immediate_media=True)       reference helpers.py:236-279
  MaskedCrossAttention / PerceiverAttention / FeedForward          parameter containers with the reference names

The nn.Modules only own fp32 master parameters.  ``forward`` hands raw device pointers to the C ABI
(include/of_hip.h) through ``torch.autograd.Function``s whose backward is the hand-derived kernel schedule in
``open_flamingo_amd/hip/path.py``.  There is no CPU or eager fallback: CPU tensors raise, a missing
``libofhip.so`` raises at first use.
"""
import torch
from torch import nn

from ..hip import path as _path
from ..hip.ops import BF16, F32, Ops


def exists(val):
    return val is not None


def _require_hip(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: the open_flamingo_amd hot path only runs on an AMD GPU through libofhip.so "
            f"(got a {t.device} tensor); there is no CPU/PyTorch fallback.")
    if t.dtype not in (F32, BF16):
        raise TypeError(f"{what}: activations must be float32 or bfloat16 (got {t.dtype})")


# per-forward artefacts shared by all blocks that see the same media / media_locations tensors live in the model's Scope
# (hip/path.py: Scope.shared); modules used on their own share the default scope
_shared = _path.DEFAULT_SCOPE.shared


class _HipParamModule(nn.Module):
    """Common: bf16 operand copies of the fp32 master weights.

    ``overwrites_fresh_grads``: protocol flag for train/reducer.py + train/optim.py -- this module's backward writes the
    gradient of an nn.Linear weight marked ``_of_grad_fresh`` with beta = 0 (and clears the mark) instead of adding to
    it, so the step epilogue does not have to zero those buffers and the dW GEMM does not have to read them.

    Freshness: torch optimizers update parameters WITHOUT bumping ``Tensor._version`` (measured: version stays put
    across ``AdamW.step()``), so a version-keyed cache silently trains on step-0 weights.  Therefore in training mode
    the copies are re-cast on every forward (one HBM pass over the block's weights) unless the libofhip step epilogue
    (``train/optim.py``), which rewrites the bf16 copies in its AdamW pass, vouches for them; in eval mode they are
    cached and re-validated by version/pointer (``load_state_dict`` and other in-place writes do bump the version)."""

    overwrites_fresh_grads = True

    def _weights_bf16(self, ops, named):
        cache = self.__dict__.setdefault("_w_bf16_cache", {})
        provider = self.__dict__.get("_w_bf16_provider")   # train/optim.py keeps bf16 copies current in its AdamW pass
        out = _path.WeightDict()
        # fragment-major copies for the fused attention branch (hip/path.py: packed_weight): cached next to the bf16 copies they are
        # made from -- which only holds while those are immutable tensors (no provider rewriting them in place)
        out.pk_cache = self.__dict__.setdefault("_w_pk_cache", {}) if provider is None else None
        for name, p in named:
            if p.dim() != 2 or name.endswith("latents") or "embs" in name:
                continue
            if p.dtype == BF16:
                out[name] = p.detach()
                continue
            if provider is not None:
                view = provider.bf16_view(p)
                if view is not None:
                    out[name] = view
                    pk = provider.packed_view(p) if hasattr(provider, "packed_view") else None
                    if pk is not None:
                        out[name + "#pk"] = pk
                    continue
            ent = cache.get(name)
            if self.training or ent is None or ent[0] != p._version or ent[1] != p.data_ptr():
                ent = (p._version, p.data_ptr(), ops.to_bf16(p.detach().contiguous()))
                cache[name] = ent
            out[name] = ent[2]
        return out

    def invalidate_weight_cache(self):
        """Call after mutating parameters through ``.data`` (which does not bump the version counter)."""
        self.__dict__.pop("_w_bf16_cache", None)
        self.__dict__.pop("_w_pk_cache", None)
        self.__dict__.pop("_kv_cache", None)
        self.__dict__.pop("_decode_graph", None)

    def train(self, mode: bool = True):
        # the copies cast during the last training forward predate the optimizer step that followed it, and their
        # version stamp cannot tell (see above): never carry them across a train() / eval() switch
        self.invalidate_weight_cache()
        return super().train(mode)

    @staticmethod
    def _masters(named):
        return {k: (p.detach() if p.dtype == F32 else p.detach().float()).contiguous() for k, p in named}


def _grad_sinks(names, params):
    """Parameters whose owner (train/reducer.py) asked for in-place gradient accumulation: their existing fp32 ``.grad``
    (a view into a reducer bucket) is handed to the backward as the accumulation target.  The saving is autograd's
    AccumulateGrad ``grad += new`` kernel per parameter per backward (~4.8 ms / step in the benchmark)."""
    sinks, fresh = {}, set()
    for k, p in zip(names, params):
        if getattr(p, "_of_inplace_grad", False) and p.grad is not None and p.grad.dtype == F32 \
                and p.grad.is_contiguous() and p.grad.shape == p.shape:
            sinks[k] = p.grad
            if getattr(p, "_of_grad_fresh", False):      # not cleared by the step epilogue: this backward overwrites it
                fresh.add(k)
                p._of_grad_fresh = False
        elif getattr(p, "_of_grad_fresh", False):
            # marked "stale content, overwrite me" but not usable as a sink (its .grad was replaced or is not the plain fp32
            # view any more): autograd's AccumulateGrad would ADD the fresh gradient to the previous step's -- clear it first
            if p.grad is not None:
                p.grad.zero_()
            p._of_grad_fresh = False
    return sinks, fresh


def _hand_back(names, params, dtypes, g, sinks, notified=()):
    """Gradients for autograd: None where the kernels already accumulated into ``.grad`` -- for those the owner's
    post-accumulate callback is run here (unless the backward already ran it, ``notified``), since AccumulateGrad (and
    with it the registered hook) will not fire."""
    out = []
    for k, p, dt in zip(names, params, dtypes):
        if k in sinks:
            out.append(None)
        else:
            out.append(g[k] if g[k].dtype == dt else g[k].to(dt))
    for k, p in zip(names, params):
        if k in sinks and k not in notified:
            p._of_on_grad(p)
    return tuple(out)


class _FeedForwardFn(torch.autograd.Function):
    """Stand-alone FeedForward (reference helpers.py:15-22): LN -> Linear -> erf-GELU -> Linear, no residual, no gate."""
    NAMES = ("0.weight", "0.bias", "1.weight", "3.weight")

    @staticmethod
    def forward(ctx, mod, x, *params):
        ops = Ops.default()
        named = list(zip(_FeedForwardFn.NAMES, params))
        P, W = mod._masters(named), mod._weights_bf16(ops, named)
        xr = x.detach().reshape(-1, x.shape[-1])
        xr = xr if xr.is_contiguous() else xr.contiguous()
        keep = any(ctx.needs_input_grad)
        y, S = _path.feed_forward_fwd(ops, P, W, xr, keep=keep)
        ctx.S, ctx.P, ctx.W, ctx.params, ctx.param_dtypes = S, P, W, params, tuple(p.dtype for p in params)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        ops = Ops.default()
        sinks, fresh = _grad_sinks(_FeedForwardFn.NAMES, ctx.params)
        G = _path._GradOut(sinks, dy.device, fresh)
        dx, _ = _path.feed_forward_bwd(ops, ctx.P, ctx.W, ctx.S, dy.reshape(-1, dy.shape[-1]).contiguous(), G)
        ctx.S = None
        return (None, dx.view(dy.shape)) + _hand_back(_FeedForwardFn.NAMES, ctx.params, ctx.param_dtypes, G.g, sinks)


class FeedForward(nn.Sequential, _HipParamModule):
    """The reference's Sequential layout (0: LayerNorm, 1: Linear, 2: GELU, 3: Linear; helpers.py:15-22).  Inside
    PerceiverResampler / GatedCrossAttentionBlock it runs fused with the residual and the tanh gate; called directly it
    is the same libofhip kernels without them (``ff(x)`` as in the reference: no residual)."""

    def __init__(self, dim, mult=4):
        inner_dim = int(dim * mult)
        super().__init__(nn.LayerNorm(dim), nn.Linear(dim, inner_dim, bias=False), nn.GELU(),
                         nn.Linear(inner_dim, dim, bias=False))

    def forward(self, x):
        _require_hip(x, "FeedForward")
        return _FeedForwardFn.apply(self, x, self[0].weight, self[0].bias, self[1].weight, self[3].weight)


_PATTN_NAMES = ("norm_media.weight", "norm_media.bias", "norm_latents.weight", "norm_latents.bias", "to_q.weight",
                "to_kv.weight", "to_out.weight")


class _PerceiverAttentionFn(torch.autograd.Function):
    """Stand-alone PerceiverAttention.forward(x, latents) (reference helpers.py:39-65): no residual."""

    @staticmethod
    def forward(ctx, mod, x, latents, *params):
        ops = Ops.default()
        named = list(zip(_PATTN_NAMES, params))
        P, W = mod._masters(named), mod._weights_bf16(ops, named)
        b, T, n1, D = x.shape
        n2 = latents.shape[2]
        xr = x.detach().reshape(b * T * n1, D).contiguous()
        lr = latents.detach().reshape(b * T * n2, D).to(xr.dtype).contiguous()
        dims = dict(N=b * T, Fv=n1, n=n2, heads=mod.heads, prefix="", dim_head=_kernel_dim_head(mod.dim_head), scale=mod.dim_head ** -0.5)
        out, S = _path.perceiver_attention_fwd(ops, P, W, xr, lr, **dims)
        ctx.S, ctx.P, ctx.W, ctx.dims, ctx.params, ctx.param_dtypes = S, P, W, dims, params, tuple(p.dtype for p in params)
        ctx.xshape, ctx.lshape = tuple(x.shape), tuple(latents.shape)
        return out.view(b, T, n2, D)

    @staticmethod
    def backward(ctx, dout):
        ops = Ops.default()
        sinks, fresh = _grad_sinks(_PATTN_NAMES, ctx.params)
        G = _path._GradOut(sinks, dout.device, fresh)
        d2 = dout.reshape(-1, dout.shape[-1]).contiguous()
        dlat, dx = _path.perceiver_attention_bwd(ops, ctx.P, ctx.W, ctx.S, d2, ops.to_bf16(d2), G,
                                                 need_dx=ctx.needs_input_grad[1], **ctx.dims)
        ctx.S = None
        return (None, dx.view(ctx.xshape) if dx is not None else None, dlat.view(ctx.lshape)) + \
            _hand_back(_PATTN_NAMES, ctx.params, ctx.param_dtypes, G.g, sinks)


def _check_dim_head(dim_head, who):
    if not (isinstance(dim_head, int) and 1 <= dim_head <= 128):
        raise NotImplementedError(f"{who}: the libofhip attention kernels take heads of up to 128 columns; got dim_head={dim_head}")


def _kernel_dim_head(dim_head):
    """The attention kernels exist for head sizes 64 (every released OpenFlamingo model) and 128.  The reference accepts ANY
    dim_head (helpers.py:26-30,137-149): other sizes run with every head ZERO-PADDED to the next kernel size -- zero query / key
    columns add nothing to a score, zero value columns give zero output columns, which meet zero columns of to_out -- and the softmax
    scale of the TRUE size (``_pad_heads``: the padded weights are differentiable views of the parameters, so their gradients are the
    true parameters' through autograd's slicing; the same arithmetic on more columns, not a fast path)."""
    return 64 if dim_head <= 64 else 128


def _pad_heads(names, params, heads, dim_head):
    """to_q / to_kv rows and to_out columns of every attention in ``names`` padded per head from dim_head to the kernel size."""
    dhp = _kernel_dim_head(dim_head)
    if dhp == dim_head:
        return list(params)
    out = []
    for k, p in zip(names, params):
        if k.endswith(("to_q.weight", "to_kv.weight", "to_out.weight")) and getattr(p, "_of_grad_fresh", False):
            # the step epilogue left this gradient stale for a backward that OVERWRITES it (train/optim.py); the padded view's
            # gradient reaches the parameter through autograd's AccumulateGrad, which adds: clear the stale content now
            if p.grad is not None:
                p.grad.zero_()
            p._of_grad_fresh = False
        if k.endswith("to_q.weight") or k.endswith("to_kv.weight"):          # (groups * dim_head, D): groups = heads | k heads + v heads
            g = p.shape[0] // dim_head
            p = torch.nn.functional.pad(p.view(g, dim_head, p.shape[1]), (0, 0, 0, dhp - dim_head)).reshape(g * dhp, p.shape[1])
        elif k.endswith("to_out.weight"):                                     # (D, heads * dim_head)
            p = torch.nn.functional.pad(p.view(p.shape[0], heads, dim_head), (0, dhp - dim_head)).reshape(p.shape[0], heads * dhp)
        out.append(p)
    return out


class PerceiverAttention(_HipParamModule):
    """helpers.py:25-65.  Inside PerceiverResampler it runs fused with the residual; called directly it returns
    attention(x, latents) like the reference."""

    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        _check_dim_head(dim_head, "PerceiverAttention")
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        inner_dim = dim_head * heads
        self.norm_media = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    def forward(self, x, latents):
        """x (b, T, n1, D) media features, latents (b, T, n2, D) -> (b, T, n2, D)   [reference helpers.py:39-65]"""
        _require_hip(x, "PerceiverAttention")
        _require_hip(latents, "PerceiverAttention(latents)")
        params = [self.norm_media.weight, self.norm_media.bias, self.norm_latents.weight, self.norm_latents.bias,
                  self.to_q.weight, self.to_kv.weight, self.to_out.weight]
        params = _pad_heads(_PATTN_NAMES, params, self.heads, self.dim_head)
        return _PerceiverAttentionFn.apply(self, x, latents, *params)


def _perceiver_operands(mod, names, x, params):
    ops = Ops.default()
    b, T, Fr, v, D = x.shape
    named = list(zip(names, params))
    P = mod._masters(named)
    W = mod._weights_bf16(ops, named)
    xr = x.detach().reshape(b * T * Fr * v, D)
    if not xr.is_contiguous():
        xr = xr.contiguous()
    assert ("frame_embs" not in P or Fr <= P["frame_embs"].shape[0]) and \
        ("media_time_embs" not in P or T <= P["media_time_embs"].shape[0]), "more frames/media than embedding rows"
    dims = dict(N=b * T, Fv=Fr * v, n=P["latents"].shape[0], heads=mod.heads, depth=mod.depth, T=T, frames=Fr,
                dim_head=_kernel_dim_head(mod.dim_head), scale=mod.dim_head ** -0.5)
    return ops, P, W, xr, dims


class _PerceiverFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, names, x, *params):
        b, T, _, _, D = x.shape
        ops, P, W, xr, dims = _perceiver_operands(mod, names, x, params)
        out, S = _path.perceiver_fwd(ops, P, W, xr, **dims)
        ctx.mod, ctx.names, ctx.dims, ctx.S, ctx.P, ctx.W = mod, names, dims, S, P, W
        ctx.xshape = tuple(x.shape)
        ctx.params, ctx.param_dtypes = params, tuple(p.dtype for p in params)
        return out.view(b, T, dims["n"], D)

    @staticmethod
    def backward(ctx, dout):
        ops = Ops.default()
        dims = ctx.dims
        D = ctx.xshape[-1]
        need_dx = ctx.needs_input_grad[2]
        sinks, fresh = _grad_sinks(ctx.names, ctx.params)
        by_name = dict(zip(ctx.names, ctx.params))
        notified = set()

        def on_ready(names):      # a layer's gradients are final: let their owner start the exchange now
            for k in names:
                if k in sinks and k not in notified:
                    notified.add(k)
                    by_name[k]._of_on_grad(by_name[k])

        dx, g = _path.perceiver_bwd(ops, ctx.P, ctx.W, ctx.S, dout.reshape(-1, D), need_dx=need_dx, sinks=sinks,
                                    fresh=fresh, on_ready=on_ready, **dims)
        ctx.S = None
        grads = _hand_back(ctx.names, ctx.params, ctx.param_dtypes, g, sinks, notified)
        return (None, None, dx.view(ctx.xshape) if need_dx else None) + grads


class PerceiverResampler(_HipParamModule):
    def __init__(self, *, dim, depth=6, dim_head=64, heads=8, num_latents=64, max_num_media=None,
                 max_num_frames=None, ff_mult=4):
        super().__init__()
        _check_dim_head(dim_head, "PerceiverResampler")
        # creation order == reference (helpers.py:82-105) so torch.manual_seed(s) gives identical initial weights
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.frame_embs = nn.Parameter(torch.randn(max_num_frames, dim)) if exists(max_num_frames) else None
        self.media_time_embs = nn.Parameter(torch.randn(max_num_media, 1, dim)) if exists(max_num_media) else None
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                              FeedForward(dim=dim, mult=ff_mult)]))
        self.norm = nn.LayerNorm(dim)
        self.heads, self.depth, self.dim, self.dim_head = heads, depth, dim, dim_head

    def forward(self, x):
        """x (b, T, F, v, D) -> (b, T, num_latents, D)   [reference helpers.py:107-132]"""
        _require_hip(x, "PerceiverResampler")
        assert x.dim() == 5 and x.shape[-1] == self.dim, f"expected (b,T,F,v,{self.dim}), got {tuple(x.shape)}"
        named = list(self.named_parameters())
        names = tuple(k for k, _ in named)
        params = _pad_heads(names, [p for _, p in named], self.heads, self.dim_head)
        if not torch.is_grad_enabled():          # inference: nothing kept for a backward
            ops, P, W, xr, dims = _perceiver_operands(self, names, x, params)
            out, _ = _path.perceiver_fwd(ops, P, W, xr, keep=False, **dims)
            return out.view(x.shape[0], x.shape[1], dims["n"], x.shape[-1])
        return _PerceiverFn.apply(self, names, x, *params)


_MCA_NAMES = ("norm.weight", "norm.bias", "to_q.weight", "to_kv.weight", "to_out.weight")


class MaskedCrossAttention(_HipParamModule):
    """helpers.py:137-233.  Inside GatedCrossAttentionBlock it runs fused with the tanh gate and the residual; called
    directly it returns the attention output like the reference."""

    def __init__(self, *, dim, dim_visual, dim_head=64, heads=8, only_attend_immediate_media=True):
        super().__init__()
        _check_dim_head(dim_head, "MaskedCrossAttention")
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        inner_dim = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim_visual, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self.only_attend_immediate_media = only_attend_immediate_media

    def forward(self, x, media, media_locations=None, use_cached_media=False):
        """x (B, T_txt, D_txt), media (B, T_img, n, D_img), media_locations (B, T_txt) bool   [reference helpers.py:160-233]"""
        _require_hip(x, "MaskedCrossAttention")
        _require_hip(media, "MaskedCrossAttention(media)")
        if not use_cached_media:
            assert media_locations is None or media_locations.shape[1] == x.shape[1], (
                f"media_location.shape is {media_locations.shape} but x.shape is {x.shape}")
        params = [self.norm.weight, self.norm.bias, self.to_q.weight, self.to_kv.weight, self.to_out.weight]
        params = _pad_heads(_MCA_NAMES, params, self.heads, self.dim_head)
        return _MaskedCrossAttentionFn.apply(self, x, media, media_locations, use_cached_media, *params)


_capture_streams = {}


def _capture_stream(device):
    """HIP graphs are captured on a side stream (capture on the default stream is not allowed)."""
    if device not in _capture_streams:
        _capture_streams[device] = torch.cuda.Stream(device=device)
    return _capture_streams[device]


_XATTN_NAMES = ("attn_gate", "ff_gate", "attn.norm.weight", "attn.norm.bias", "attn.to_q.weight", "attn.to_kv.weight",
                "attn.to_out.weight", "ff.0.weight", "ff.0.bias", "ff.1.weight", "ff.3.weight")


def _xattn_operands(mod, x, media, media_locations, use_cached_media, params, names=None, attn=None):
    ops = Ops.default()
    B, L, d = x.shape
    _, T, n, Dv = media.shape
    named = list(zip(names or _XATTN_NAMES, params))
    attn = attn if attn is not None else mod.attn
    P = mod._masters(named)
    W = mod._weights_bf16(ops, named)
    xr = x.detach().reshape(B * L, d)
    if not xr.is_contiguous():
        xr = xr.contiguous()
    med = media.detach()
    shared = _path.scope_of(mod).shared
    media_bf = shared.get(media, "bf16", lambda: ops.to_bf16(med.reshape(B * T * n, Dv).contiguous())
                          if med.dtype == F32 else med.reshape(B * T * n, Dv).contiguous())
    tt = None
    if media_locations is not None:
        def _tt():
            out = torch.empty(B, L, dtype=torch.int32, device=x.device)
            ml = media_locations.to(torch.uint8).contiguous()
            ops.text_time(ml, out, L, bool(use_cached_media))
            return out
        tt = shared.get(media_locations, ("tt", L, bool(use_cached_media)), _tt)
    dims = dict(B=B, L=L, T=T, n=n, heads=attn.heads, only_immediate=attn.only_attend_immediate_media,
                dim_head=_kernel_dim_head(attn.dim_head), scale=attn.dim_head ** -0.5)
    return ops, P, W, xr, media_bf, tt, dims


class _MaskedCrossAttentionFn(torch.autograd.Function):
    """Stand-alone MaskedCrossAttention.forward (reference helpers.py:160-233): no gate, no residual."""

    @staticmethod
    def forward(ctx, mod, x, media, media_locations, use_cached_media, *params):
        ops, P, W, xr, media_bf, tt, dims = _xattn_operands(mod, x, media, media_locations, use_cached_media, params,
                                                            names=_MCA_NAMES, attn=mod)
        y, S = _path.masked_cross_attention_fwd(ops, P, W, xr, media_bf, tt, prefix="", **dims)
        ctx.S, ctx.P, ctx.W, ctx.dims, ctx.media_bf, ctx.tt = S, P, W, dims, media_bf, tt
        ctx.xshape, ctx.mshape, ctx.mdtype = tuple(x.shape), tuple(media.shape), media.dtype
        ctx.params, ctx.param_dtypes = params, tuple(p.dtype for p in params)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        ops = Ops.default()
        sinks, fresh = _grad_sinks(_MCA_NAMES, ctx.params)
        G = _path._GradOut(sinks, dy.device, fresh)
        d2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        dx, dmedia = _path.masked_cross_attention_bwd(ops, ctx.P, ctx.W, ctx.S, ctx.media_bf, ctx.tt, d2, ops.to_bf16(d2), G,
                                                      prefix="", need_dmedia=ctx.needs_input_grad[2], **ctx.dims)
        ctx.S = None
        if dmedia is not None:
            dmedia = (dmedia if ctx.mdtype == F32 else ops.to_bf16(dmedia)).view(ctx.mshape)
        return (None, dx.view(ctx.xshape), dmedia, None, None) + _hand_back(_MCA_NAMES, ctx.params, ctx.param_dtypes, G.g, sinks)


class MediaKVGroup:
    """All gated blocks project the SAME media tensor with their own to_kv (reference helpers.py:187-189, called from every
    FlamingoLayer with the one tensor flamingo.py:199-200 hands out): SURVEY appendix B3.  One grouped GEMM right after the
    Perceiver computes every block's keys/values (``kv_all``, block i owns columns [i*E, (i+1)*E)), every block's backward
    writes its d(k|v) into the matching columns of ``dkv_all``, and ONE K-grouped GEMM forms the media gradient of all
    blocks (instead of 24 small GEMMs whose fp32 results autograd then adds with 23 more kernels).  The weight gradients
    of to_kv stay in the blocks' own backward, so they are ready block by block for the gradient exchange."""

    def __init__(self, blocks, media):
        self.blocks = list(blocks)
        self.index = {id(b): i for i, b in enumerate(self.blocks)}
        self.media = media
        self.kv_all = self.dkv_all = self.token = None
        self.backward_probe = None      # (block index in forward order, callable): called once at the start of that block's backward

    def kv_of(self, block):
        i = self.index[id(block)]
        return self.kv_all[:, i * self.E:(i + 1) * self.E]

    def dkv_of(self, block):
        if self.dkv_all is None:        # every column block is fully written by its block's attention backward
            self.dkv_all = torch.empty_like(self.kv_all)
            self.written = set()
        i = self.index[id(block)]
        self.written.add(i)
        return self.dkv_all[:, i * self.E:(i + 1) * self.E]


_media_groups = []        # [(media tensor, MediaKVGroup)]: identity-keyed, a handful of entries (LAION + MMC4 passes)


def can_group_media(media):
    return media.is_cuda


def group_media_projections(blocks, media):
    """Called by Flamingo._encode_vision_x under autograd on the GPU; blocks that find their media tensor here use the
    grouped projection (``GatedCrossAttentionBlock.forward``)."""
    blocks = [b for b in blocks if b is not None]
    if len(blocks) < 2 or not can_group_media(media) or not torch.is_grad_enabled() \
            or not all(isinstance(b, GatedCrossAttentionBlock) for b in blocks):
        return None
    a0 = blocks[0].attn
    E = a0.to_kv.weight.shape[0]
    if any(_kernel_dim_head(b.attn.dim_head) != b.attn.dim_head for b in blocks):
        return None                       # zero-padded heads: the blocks project their padded weights on their own
    if any(b.attn.to_kv.weight.shape != a0.to_kv.weight.shape for b in blocks) or E % 256 or media.shape[-1] % 256 \
            or (media.shape[0] * media.shape[1] * media.shape[2]) % 256:
        return None                       # not big-tile eligible (forward: N = E, K = D_img; backward: N = D_img, K = E
                                          # per group; both M = B*T*n): the blocks project on their own
    grp = MediaKVGroup(blocks, media)
    grp.E = E
    grp.token = _GroupedMediaKVFn.apply(grp, media)
    _media_groups.append((media, grp))
    del _media_groups[:-4]
    return grp


def _media_group_of(block, media):
    for m, grp in _media_groups:
        if m is media and id(block) in grp.index:
            return grp
    return None


def drop_media_groups():
    _media_groups.clear()


class _GroupedMediaKVFn(torch.autograd.Function):
    _table = None        # ((weight pointers, device), device table of them): rebuilt only when a pointer changes

    @staticmethod
    def forward(ctx, grp, media):
        ops = Ops.default()
        B, T, n, Dv = media.shape
        med = media.detach()
        grp.media_bf = _path.scope_of(grp.blocks[0]).shared.get(
            media, "bf16", lambda: ops.to_bf16(med.reshape(B * T * n, Dv).contiguous()) if med.dtype == F32 else med.reshape(B * T * n, Dv).contiguous())
        grp.Ws = [b._weights_bf16(ops, [("attn.to_kv.weight", b.attn.to_kv.weight)])["attn.to_kv.weight"] for b in grp.blocks]
        ptrs = tuple(w.data_ptr() for w in grp.Ws)           # stable while the step epilogue owns the bf16 copies
        if _GroupedMediaKVFn._table is None or _GroupedMediaKVFn._table[0] != (ptrs, media.device):
            _GroupedMediaKVFn._table = ((ptrs, media.device), torch.tensor(ptrs, dtype=torch.int64, device=media.device))
        grp.table = _GroupedMediaKVFn._table[1]
        grp.kv_all = torch.empty(B * T * n, grp.E * len(grp.Ws), dtype=BF16, device=media.device)
        ops.gemm_grouped(grp.media_bf, grp.Ws, grp.table, grp.kv_all, kind=1)
        ctx.grp, ctx.mshape, ctx.mdtype = grp, tuple(media.shape), media.dtype
        return torch.zeros(1, device=media.device)          # token: ties every block's backward to this node

    @staticmethod
    def backward(ctx, _g):
        grp = ctx.grp
        if not ctx.needs_input_grad[1] or grp.dkv_all is None:
            return None, None
        for i in range(len(grp.blocks)):       # a block whose output the loss does not reach never wrote its columns: zero
            if i not in grp.written:
                grp.dkv_all[:, i * grp.E:(i + 1) * grp.E].zero_()
        ops = Ops.default()
        from ..hip.abi import EPI_ACC_F32
        dmedia = torch.empty(grp.kv_all.shape[0], ctx.mshape[-1], dtype=F32, device=grp.kv_all.device)
        ops.gemm_grouped(grp.dkv_all, grp.Ws, grp.table, dmedia, kind=2, epi=EPI_ACC_F32)
        # dkv_all is re-made by the next backward (retain_graph / a second grad-enabled forward over cached media); kv_all
        # lives as long as the group does (dropped with the conditioning or when newer groups push it out)
        grp.dkv_all = None
        return None, (dmedia if ctx.mdtype == F32 else ops.to_bf16(dmedia)).view(ctx.mshape)


class _GatedXAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, media, media_locations, use_cached_media, grp, token, *params):
        B, L, d = x.shape
        ops, P, W, xr, media_bf, tt, dims = _xattn_operands(mod, x, media, media_locations, use_cached_media, params)
        y, S = _path.xattn_block_fwd(ops, P, W, xr, media_bf, tt, kv=grp.kv_of(mod) if grp is not None else None, **dims)
        ctx.grp = grp
        ctx.mod, ctx.dims, ctx.S, ctx.P, ctx.W, ctx.media_bf, ctx.tt = mod, dims, S, P, W, media_bf, tt
        ctx.xshape, ctx.mshape, ctx.mdtype = tuple(x.shape), tuple(media.shape), media.dtype
        ctx.params, ctx.param_dtypes = params, tuple(p.dtype for p in params)
        return y.view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        ops = Ops.default()
        d = ctx.xshape[-1]
        grp = ctx.grp
        if grp is not None and grp.backward_probe is not None and grp.index[id(ctx.mod)] == grp.backward_probe[0]:
            probe, grp.backward_probe = grp.backward_probe[1], None
            probe()                        # Flamingo.schedule_vision_prefetch: the next step's tower forward starts on its side stream here
        need_dmedia = ctx.needs_input_grad[2] and grp is None      # grouped: the group's node forms the media gradient
        sinks, fresh = _grad_sinks(_XATTN_NAMES, ctx.params)
        # norm taps (train/optim.py): parameters whose owner registered slots for the sum of squares of their gradient
        taps = {k: p._of_sumsq_slots for k, p in zip(_XATTN_NAMES, ctx.params) if getattr(p, "_of_sumsq_slots", None) is not None}
        dx, dmedia, g = _path.xattn_block_bwd(ops, ctx.P, ctx.W, ctx.S, ctx.media_bf, ctx.tt, dy.reshape(-1, d),
                                              need_dmedia=need_dmedia, sinks=sinks, fresh=fresh,
                                              dkv_out=grp.dkv_of(ctx.mod) if grp is not None else None,
                                              scope=_path.scope_of(ctx.mod), sumsq=taps, **ctx.dims)
        done = g.pop("__sumsq_done__", {})
        for k, p in zip(_XATTN_NAMES, ctx.params):
            if k in taps:
                p._of_sumsq_valid = bool(done.get(k, False))
                p._of_sumsq_version = p.grad._version if p.grad is not None else None      # (step() voids the tap if the gradient is written again)
        ctx.S = None
        if dmedia is not None:
            dmedia = (dmedia if ctx.mdtype == F32 else ops.to_bf16(dmedia)).view(ctx.mshape)
        grads = _hand_back(_XATTN_NAMES, ctx.params, ctx.param_dtypes, g, sinks)
        # the token's gradient carries no value (the group's node reads dkv_all); a defined tensor from the block that
        # runs LAST in the backward (first in the forward) is enough to schedule the group's node after every block
        dtok = torch.zeros(1, device=dy.device) if (grp is not None and grp.index[id(ctx.mod)] == 0) else None
        return (None, dx.view(ctx.xshape), dmedia, None, None, None, dtok) + grads


class GatedCrossAttentionBlock(_HipParamModule):
    def __init__(self, *, dim, dim_visual, dim_head=64, heads=8, ff_mult=4, only_attend_immediate_media=True):
        super().__init__()
        _check_dim_head(dim_head, "GatedCrossAttentionBlock")
        self.attn = MaskedCrossAttention(dim=dim, dim_visual=dim_visual, dim_head=dim_head, heads=heads,
                                         only_attend_immediate_media=only_attend_immediate_media)
        self.attn_gate = nn.Parameter(torch.tensor([0.0]))
        self.ff = FeedForward(dim, mult=ff_mult)
        self.ff_gate = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x, media, media_locations=None, use_cached_media=False):
        """x (B, T_txt, D_txt), media (B, T_img, n, D_img), media_locations (B, T_txt) bool
        [reference helpers.py:260-279 + 160-233]"""
        _require_hip(x, "GatedCrossAttentionBlock")
        _require_hip(media, "GatedCrossAttentionBlock(media)")
        if not use_cached_media:
            if media_locations is None:
                raise ValueError("media_locations is required unless use_cached_media=True")
            assert media_locations.shape[1] == x.shape[1], (
                f"media_location.shape is {media_locations.shape} but x.shape is {x.shape}")
        a, f = self.attn, self.ff                        # the parameters in _XATTN_NAMES order
        params = [self.attn_gate, self.ff_gate, a.norm.weight, a.norm.bias, a.to_q.weight, a.to_kv.weight,
                  a.to_out.weight, f[0].weight, f[0].bias, f[1].weight, f[3].weight]
        params = _pad_heads(_XATTN_NAMES, params, a.heads, a.dim_head)
        if not torch.is_grad_enabled():
            return self._forward_inference(x, media, media_locations, use_cached_media, params)
        grp = _media_group_of(self, media)
        return _GatedXAttnFn.apply(self, x, media, media_locations, use_cached_media, grp,
                                   grp.token if grp is not None else None, *params)

    # A decode step (T_txt = 1) is ~10 small launches per block whose cost is host time, not GPU time: replay them as
    # one HIP graph per block.  Class-level switch; set False (on the class or an instance) to launch kernel by kernel.
    decode_graphs = True

    def release_media_cache(self):
        self.__dict__.pop("_kv_cache", None)
        drop_media_groups()

    def _forward_inference(self, x, media, media_locations, use_cached_media, params):
        """No-grad forward.  The reference recomputes ``to_kv(media)`` in every block for every generated token
        (helpers.py:189); here the projected keys/values are kept per block for as long as the caller keeps
        conditioning on the same media tensor (``Flamingo.generate`` / ``cache_media`` hold one tensor on the layers
        for the whole decode) and the block's to_kv weights are unchanged.  Nothing is saved for a backward."""
        ops, P, W, xr, media_bf, tt, dims = _xattn_operands(self, x, media, media_locations, use_cached_media, params)
        w_kv = W["attn.to_kv.weight"]
        key = (media.data_ptr(), media._version, tuple(media.shape), media.dtype, w_kv.data_ptr(), w_kv._version)
        ent = self.__dict__.get("_kv_cache")
        fresh = ent is None or ent[0] != key or ent[1] is not media
        graph = self._decode_graph_for(ops, P, W, xr, tt, dims, params) if x.shape[1] == 1 else None
        if graph is not None:
            # the graph reads the projected media from ITS buffer: project into it (new prompt) or move the prompt
            # pass's projection there once
            if fresh:
                _path.xattn_project_media(ops, W, media_bf, dims["heads"], out=graph["kv"], dim_head=dims["dim_head"])
            elif ent[3] is not graph["kv"]:
                graph["kv"].copy_(ent[3])
            if fresh or ent[3] is not graph["kv"]:
                self.__dict__["_kv_cache"] = (key, media, w_kv, graph["kv"])
            if graph.get("tt_src") is not tt:
                graph["tt"].copy_(tt)
                graph["tt_src"] = tt
            graph["x"].copy_(xr)
            graph["graph"].replay()
            return graph["y"].clone().view(x.shape)
        if fresh:
            ent = (key, media, w_kv, _path.xattn_project_media(ops, W, media_bf, dims["heads"], dim_head=dims["dim_head"]))
            self.__dict__["_kv_cache"] = ent       # holds `media` and the weight copy: their storage cannot be reused
        y, _ = _path.xattn_block_fwd(ops, P, W, xr, media_bf, tt, kv=ent[3], keep=False, **dims)
        return y.view(x.shape)

    def _decode_graph_for(self, ops, P, W, xr, tt, dims, params):
        """HIP graph of one decode step of this block for the current (batch, images) shape and weights, or None.
        The graph owns static input buffers (token, text_time, projected media) and its split-K workspace; it is
        captured the second time a shape is seen (the first call runs kernel by kernel, which also warms every lazily
        initialised piece outside the capture), and dropped with the weight copies (``invalidate_weight_cache``)."""
        if not self.decode_graphs or tt is None or not xr.is_cuda or _kernel_dim_head(self.attn.dim_head) != self.attn.dim_head:
            return None
        sig = (dims["B"], dims["T"], dims["n"], xr.dtype, tt.shape,
               tuple((w.data_ptr(), w._version) for w in W.values()),
               tuple((p.data_ptr(), p._version) for p in params))
        slots = self.__dict__.setdefault("_decode_graph", {})      # a few shapes (e.g. the last, smaller batch of an eval)
        st = slots.get(sig)
        if st is None:
            if len(slots) >= 4:
                slots.pop(next(iter(slots)))
            slots[sig] = dict(sig=sig, graph=None)
            return None
        if st["graph"] is not None or st.get("failed"):
            return st if st["graph"] is not None else None
        inner = dims["heads"] * dims["dim_head"]
        st.update(x=torch.empty_like(xr), tt=torch.empty_like(tt), tt_src=None,
                  kv=torch.empty(dims["B"] * dims["T"] * dims["n"], 2 * inner, dtype=BF16, device=xr.device),
                  keep=(P, W))                       # the operands whose addresses the graph holds
        outer_ws = ops.__dict__.pop("_gemm_ws", None)     # the shared scratch may be re-allocated later: the graph gets
        try:                                              # its own, allocated from the graph's memory pool
            g = torch.cuda.CUDAGraph()
            side = _capture_stream(xr.device)
            side.wait_stream(torch.cuda.current_stream(xr.device))
            with torch.cuda.stream(side):                 # (torch.cuda.graph() would add a device sync + gc per block)
                g.capture_begin()
                try:
                    y, _ = _path.xattn_block_fwd(ops, P, W, st["x"], None, st["tt"], kv=st["kv"], keep=False, **dims)
                finally:
                    g.capture_end()
            torch.cuda.current_stream(xr.device).wait_stream(side)
            st.update(graph=g, y=y, ws=ops.__dict__.pop("_gemm_ws", None))
        except Exception as exc:                          # same kernels, launched one by one
            import warnings
            warnings.warn(f"GatedCrossAttentionBlock: HIP graph capture of the decode step failed ({exc!r}); "
                          "continuing with per-kernel launches")
            st.update(graph=None, failed=True)
        finally:
            ops.__dict__.pop("_gemm_ws", None)
            if outer_ws is not None:
                ops._gemm_ws = outer_ws
        return st if st["graph"] is not None else None
