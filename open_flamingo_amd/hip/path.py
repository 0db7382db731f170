"""Kernel schedules of the hot path: which libofhip kernels run, in which order, on which buffers.

Pure orchestration over ``Ops`` (no arithmetic happens in Python/PyTorch here): forward and hand-derived
backward of ``GatedCrossAttentionBlock`` (reference helpers.py:236-279) and ``PerceiverResampler``
(helpers.py:68-132).  The backward formulas are SURVEY.md appendix A; each GEMM names its role.

Conventions: P = fp32 master parameters by their reference state-dict names; W = bf16 copies of the weight
matrices (GEMM operands); "stream dtype" tensors (x, y, their grads) are fp32 or bf16 as the caller provides.
All matrices are 2-D row-major views: token rows x features.
"""
import torch

from .abi import (EPI_ACC_F32, EPI_DGELU_DOT, EPI_GATE_RESID, EPI_GELU, EPI_SCALE_DOT, EPI_STORE_BF16)
from .ops import BF16, F32


def _e(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


def _z(shape, dev):
    return torch.zeros(shape, dtype=F32, device=dev)


class _GradOut:
    """Destination of the parameter gradients of one backward.  By default a fresh buffer per parameter (handed to
    autograd, which then runs one ``grad += new`` kernel per parameter when ``.grad`` already exists -- 342 kernels and
    three passes over 3.8 GB per step in the benchmark).  When the caller supplies ``sinks`` (name -> the parameter's
    existing fp32 ``.grad``, e.g. a view into a GradReducer bucket) the kernels accumulate into it directly: the dW
    GEMMs run with beta = 1, the LayerNorm / gate / table reductions already add into their output."""

    def __init__(self, sinks, dev):
        self.sinks, self.dev, self.g = (sinks or {}), dev, {}

    def mat(self, name, shape):
        """Buffer for a GEMM-produced gradient and the beta to use with it."""
        t = self.sinks.get(name)
        beta = 1.0
        if t is None:
            t, beta = _e(shape, F32, self.dev), 0.0
        self.g[name] = t
        return t, beta

    def acc(self, name, shape):
        """Buffer that the producing kernel ADDS into."""
        t = self.sinks.get(name)
        if t is None:
            t = _z(shape, self.dev)
        self.g[name] = t
        return t


# =================================================================================================
# GatedCrossAttentionBlock
# =================================================================================================
def xattn_project_media(ops, W, media_bf, heads, out=None):
    """k | v = to_kv(media) (helpers.py:189), (B*T*n, 2*inner) bf16.  Depends on the media and the block's weights only,
    so the decode loop computes it once per block and prompt instead of once per generated token."""
    kv = _e((media_bf.shape[0], 2 * heads * 64), BF16, media_bf.device) if out is None else out
    ops.gemm(media_bf, W["attn.to_kv.weight"], kv)
    return kv


def xattn_block_fwd(ops, P, W, x, media_bf, tt, *, B, L, T, n, heads, only_immediate, safe=0, kv=None, keep=True):
    """x (B*L, d) stream dtype; media_bf (B*T*n, Dv) bf16; tt (B, L) int32 or None.  Returns (y, saved).
    kv: projected media from xattn_project_media (else computed here).  keep=False (inference): nothing is saved for a
    backward -- the pre-GELU activations are not written -- and saved is None."""
    dev = x.device
    rows, d = x.shape
    inner = heads * 64
    hid = W["ff.1.weight"].shape[0]
    # --- masked cross attention (helpers.py:160-233)
    xn = _e((rows, d), BF16, dev)
    st1 = _e((rows, 2), F32, dev)
    ops.ln_fwd(x, P["attn.norm.weight"], P["attn.norm.bias"], xn, st1)
    q = _e((rows, inner), BF16, dev)
    ops.gemm(xn, W["attn.to_q.weight"], q)                                   # to_q
    if kv is None:
        kv = xattn_project_media(ops, W, media_bf, heads)                    # to_kv (k | v fused)
    o = _e((rows, inner), BF16, dev)
    lse = _e((B, heads, L), F32, dev)
    ops.attn_fwd(q, kv[:, :inner], kv[:, inner:], o, lse, batch=B, Lq=L, Lk=T * n, heads=heads, text_time=tt,
                 n_per_media=n, T_img=T, only_immediate=only_immediate, safe=safe)
    y1 = torch.empty_like(x)
    ops.gemm(o, W["attn.to_out.weight"], y1, epi=EPI_GATE_RESID, aux=x, gate=P["attn_gate"])   # to_out, *tanh(gate), +x
    # --- gated feed forward (helpers.py:15-22, 277)
    u = _e((rows, d), BF16, dev)
    st2 = _e((rows, 2), F32, dev)
    ops.ln_fwd(y1, P["ff.0.weight"], P["ff.0.bias"], u, st2)
    a = _e((rows, hid), BF16, dev) if keep else None
    b = _e((rows, hid), BF16, dev)
    ops.gemm(u, W["ff.1.weight"], b, epi=EPI_GELU, out2=a)                   # up-projection + erf GELU
    y2 = torch.empty_like(x)
    ops.gemm(b, W["ff.3.weight"], y2, epi=EPI_GATE_RESID, aux=y1, gate=P["ff_gate"])  # down, *tanh(gate), +y1
    if not keep:
        return y2, None
    saved = dict(x=x, xn=xn, st1=st1, q=q, kv=kv, o=o, lse=lse, y1=y1, u=u, st2=st2, a=a, b=b)
    return y2, saved


def xattn_block_bwd(ops, P, W, S, media_bf, tt, dy, *, B, L, T, n, heads, only_immediate, need_dmedia=True, safe=0,
                    sinks=None):
    """Returns (dx, dmedia fp32 (B*T*n, Dv) or None, grads dict keyed like P).  sinks: see _GradOut."""
    dev = dy.device
    rows, d = S["x"].shape
    inner = heads * 64
    hid = W["ff.1.weight"].shape[0]
    Dv = media_bf.shape[1]
    G = _GradOut(sinks, dev)
    g = G.g
    dy = dy.contiguous()
    dyb = ops.to_bf16(dy)
    # ---- feed forward branch: y2 = y1 + tanh(gf) * F(y1)
    da = _e((rows, hid), BF16, dev)
    ops.gemm(dyb, W["ff.3.weight"], da, tb=True, epi=EPI_DGELU_DOT, aux=S["a"], gate=P["ff_gate"],
             dot=G.acc("ff_gate", (1,)))
    t, beta = G.mat("ff.3.weight", (d, hid))
    ops.gemm(dyb, S["b"], t, ta=True, tb=True, epi=EPI_ACC_F32, gate=P["ff_gate"], beta=beta)         # dW2
    du = _e((rows, d), BF16, dev)
    ops.gemm(da, W["ff.1.weight"], du, tb=True)
    t, beta = G.mat("ff.1.weight", (hid, d))
    ops.gemm(da, S["u"], t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)                            # dW1
    dy1 = torch.empty_like(dy)
    dy1b = _e((rows, d), BF16, dev) if dy.dtype == F32 else None
    ops.ln_bwd(du, S["y1"], S["st2"], P["ff.0.weight"], resid=dy, dx=dy1, dx_bf16=dy1b, dw=G.acc("ff.0.weight", (d,)),
               db=G.acc("ff.0.bias", (d,)))
    if dy1b is None:
        dy1b = dy1
    # ---- attention branch: y1 = x + tanh(ga) * A(x, media)
    dO = _e((rows, inner), BF16, dev)
    ops.gemm(dy1b, W["attn.to_out.weight"], dO, tb=True, epi=EPI_SCALE_DOT, aux=S["o"], gate=P["attn_gate"],
             dot=G.acc("attn_gate", (1,)))
    t, beta = G.mat("attn.to_out.weight", (d, inner))
    ops.gemm(dy1b, S["o"], t, ta=True, tb=True, epi=EPI_ACC_F32, gate=P["attn_gate"], beta=beta)
    dq = _e((rows, inner), BF16, dev)
    dkv = _e((B * T * n, 2 * inner), BF16, dev)
    delta = _e((B, heads, L), F32, dev)
    kv = S["kv"]
    ops.attn_bwd(S["q"], kv[:, :inner], kv[:, inner:], S["o"], S["lse"], dO, dq, dkv[:, :inner], dkv[:, inner:], delta,
                 batch=B, Lq=L, Lk=T * n, heads=heads, text_time=tt, n_per_media=n, T_img=T,
                 only_immediate=only_immediate, safe=safe)
    dxn = _e((rows, d), BF16, dev)
    ops.gemm(dq, W["attn.to_q.weight"], dxn, tb=True)
    t, beta = G.mat("attn.to_q.weight", (inner, d))
    ops.gemm(dq, S["xn"], t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)
    dx = torch.empty_like(dy)
    ops.ln_bwd(dxn, S["x"], S["st1"], P["attn.norm.weight"], resid=dy1, dx=dx, dw=G.acc("attn.norm.weight", (d,)),
               db=G.acc("attn.norm.bias", (d,)))
    t, beta = G.mat("attn.to_kv.weight", (2 * inner, Dv))
    ops.gemm(dkv, media_bf, t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)
    dmedia = None
    if need_dmedia:
        dmedia = _e((B * T * n, Dv), F32, dev)
        ops.gemm(dkv, W["attn.to_kv.weight"], dmedia, tb=True, epi=EPI_ACC_F32)
    return dx, dmedia, g


# =================================================================================================
# PerceiverResampler
# =================================================================================================
def perceiver_fwd(ops, P, W, x, *, N, Fv, n, heads, depth, T=1, frames=1, safe=0, keep=True):
    """x (N*Fv, D) stream dtype (already flattened 'b T (F v) d' rows, N = b*T, Fv = frames*v); returns
    (out (N*n, D) stream, saved).  frame_embs / media_time_embs (helpers.py:117-119,123-124) are added when present
    in P; T and frames give the row structure they index."""
    dev = x.device
    D = x.shape[1]
    embs = ("frame_embs" in P) or ("media_time_embs" in P)
    if embs:
        v = Fv // frames
        x_in = x
        x = torch.empty_like(x_in)
        ops.add_embs(x_in, P.get("frame_embs"), v, frames, P.get("media_time_embs"), Fv, T, x)
    inner = heads * 64
    S_ = Fv + n
    hid = W["layers.0.1.1.weight"].shape[0]
    lat = _e((N * n, D), x.dtype, dev)
    ops.broadcast_rows(P["latents"], lat, N * n)                              # repeat(latents, 'n d -> b T n d')
    layers = []
    for i in range(depth):
        pa, pf = f"layers.{i}.0.", f"layers.{i}.1."
        kvin = _e((N * S_, D), BF16, dev)          # [LN_media(x) rows | LN_latents(latents) rows] per media item
        st_m = _e((N * Fv, 2), F32, dev)
        st_l = _e((N * n, 2), F32, dev)
        ltn = _e((N * n, D), BF16, dev)
        ops.ln_fwd_grouped(x, P[pa + "norm_media.weight"], P[pa + "norm_media.bias"], kvin, D, Fv, S_ * D, None, st_m)
        ops.ln_fwd_grouped(lat, P[pa + "norm_latents.weight"], P[pa + "norm_latents.bias"], kvin[Fv:], D, n, S_ * D,
                           ltn, st_l)
        q = _e((N * n, inner), BF16, dev)
        ops.gemm(ltn, W[pa + "to_q.weight"], q)
        kv = _e((N * S_, 2 * inner), BF16, dev)
        ops.gemm(kvin, W[pa + "to_kv.weight"], kv)
        o = _e((N * n, inner), BF16, dev)
        lse = _e((N, heads, n), F32, dev)
        ops.attn_fwd(q, kv[:, :inner], kv[:, inner:], o, lse, batch=N, Lq=n, Lk=S_, heads=heads, safe=safe)
        lat1 = torch.empty_like(lat)
        ops.gemm(o, W[pa + "to_out.weight"], lat1, epi=EPI_GATE_RESID, aux=lat)              # attn(x, latents) + latents
        u = _e((N * n, D), BF16, dev)
        st_f = _e((N * n, 2), F32, dev)
        ops.ln_fwd(lat1, P[pf + "0.weight"], P[pf + "0.bias"], u, st_f)
        a = _e((N * n, hid), BF16, dev) if keep else None     # pre-GELU activations: only the backward reads them
        b = _e((N * n, hid), BF16, dev)
        ops.gemm(u, W[pf + "1.weight"], b, epi=EPI_GELU, out2=a)
        lat2 = torch.empty_like(lat)
        ops.gemm(b, W[pf + "3.weight"], lat2, epi=EPI_GATE_RESID, aux=lat1)                 # ff(latents) + latents
        if keep:
            layers.append(dict(lat=lat, kvin=kvin, st_m=st_m, st_l=st_l, ltn=ltn, q=q, kv=kv, o=o, lse=lse, lat1=lat1,
                               u=u, st_f=st_f, a=a, b=b))
        lat = lat2
    out = torch.empty_like(lat)
    st_o = _e((N * n, 2), F32, dev)
    ops.ln_fwd_out(lat, P["norm.weight"], P["norm.bias"], out, st_o)
    if not keep:
        return out, None
    return out, dict(layers=layers, lat_last=lat, st_o=st_o, x=x, embs=embs, T=T, frames=frames)


def perceiver_bwd(ops, P, W, S, dout, *, N, Fv, n, heads, depth, T=1, frames=1, need_dx=False, safe=0, sinks=None,
                  on_ready=None):
    """Returns (dx (N*Fv, D) stream dtype or None, grads dict keyed like P).  sinks: see _GradOut.
    on_ready(names): called as soon as the kernels producing the FINAL value of those parameters' gradients are enqueued
    (per layer, last layer first) -- the gradient exchange of a layer can then start while the earlier layers' backward
    still runs (train/reducer.py buckets the Perceiver per layer)."""
    ready = on_ready or (lambda names: None)
    dev = dout.device
    x = S["x"]
    want_dx = need_dx
    need_dx = need_dx or S.get("embs", False)     # the position tables' gradients are reductions of dx
    D = x.shape[1]
    inner = heads * 64
    S_ = Fv + n
    hid = W["layers.0.1.1.weight"].shape[0]
    G = _GradOut(sinks, dev)
    g = G.g
    dout = dout.contiguous()
    dlat = torch.empty_like(dout)
    ops.ln_bwd(dout, S["lat_last"], S["st_o"], P["norm.weight"], dx=dlat, dw=G.acc("norm.weight", (D,)),
               db=G.acc("norm.bias", (D,)))
    ready(["norm.weight", "norm.bias"])
    dx = None
    for i in reversed(range(depth)):
        pa, pf = f"layers.{i}.0.", f"layers.{i}.1."
        Lr = S["layers"][i]
        # ---- ff(latents) + latents
        dlb = ops.to_bf16(dlat)
        da = _e((N * n, hid), BF16, dev)
        ops.gemm(dlb, W[pf + "3.weight"], da, tb=True, epi=EPI_DGELU_DOT, aux=Lr["a"])
        t, beta = G.mat(pf + "3.weight", (D, hid))
        ops.gemm(dlb, Lr["b"], t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)
        du = _e((N * n, D), BF16, dev)
        ops.gemm(da, W[pf + "1.weight"], du, tb=True)
        t, beta = G.mat(pf + "1.weight", (hid, D))
        ops.gemm(da, Lr["u"], t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)
        dlat1 = torch.empty_like(dlat)
        dlat1b = _e((N * n, D), BF16, dev) if dlat.dtype == F32 else None
        ops.ln_bwd(du, Lr["lat1"], Lr["st_f"], P[pf + "0.weight"], resid=dlat, dx=dlat1, dx_bf16=dlat1b,
                   dw=G.acc(pf + "0.weight", (D,)), db=G.acc(pf + "0.bias", (D,)))
        if dlat1b is None:
            dlat1b = dlat1
        # ---- attn(x, latents) + latents
        dO = _e((N * n, inner), BF16, dev)
        ops.gemm(dlat1b, W[pa + "to_out.weight"], dO, tb=True)
        t, beta = G.mat(pa + "to_out.weight", (D, inner))
        ops.gemm(dlat1b, Lr["o"], t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)
        dq = _e((N * n, inner), BF16, dev)
        dkv = _e((N * S_, 2 * inner), BF16, dev)
        delta = _e((N, heads, n), F32, dev)
        kv = Lr["kv"]
        ops.attn_bwd(Lr["q"], kv[:, :inner], kv[:, inner:], Lr["o"], Lr["lse"], dO, dq, dkv[:, :inner], dkv[:, inner:],
                     delta, batch=N, Lq=n, Lk=S_, heads=heads, safe=safe)
        dltn = _e((N * n, D), BF16, dev)
        ops.gemm(dq, W[pa + "to_q.weight"], dltn, tb=True)                      # through to_q
        t, beta = G.mat(pa + "to_q.weight", (inner, D))
        ops.gemm(dq, Lr["ltn"], t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)
        dkvin = _e((N * S_, D), BF16, dev)
        ops.gemm(dkv, W[pa + "to_kv.weight"], dkvin, tb=True)                    # through to_kv (media + latent rows)
        t, beta = G.mat(pa + "to_kv.weight", (2 * inner, D))
        ops.gemm(dkv, Lr["kvin"], t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta)
        # norm_media: parameter grads always; dx only if the vision features require grad (they do not in Flamingo,
        # flamingo.py:194-195 runs the ViT under no_grad)
        dx_new = torch.empty_like(x) if need_dx else None
        ops.ln_bwd(dkvin, x, Lr["st_m"], P[pa + "norm_media.weight"], lddy=D, dy_grp_rows=Fv, dy_grp_stride=S_ * D,
                   resid=dx if need_dx else None, dx=dx_new, dw=G.acc(pa + "norm_media.weight", (D,)),
                   db=G.acc(pa + "norm_media.bias", (D,)))
        dx = dx_new
        # norm_latents: two upstream gradients (k/v rows of kv_input and the to_q input) + the residual
        dlat_prev = torch.empty_like(dlat)
        ops.ln_bwd(dkvin[Fv:], Lr["lat"], Lr["st_l"], P[pa + "norm_latents.weight"], lddy=D, dy_grp_rows=n,
                   dy_grp_stride=S_ * D, dy2=dltn, resid=dlat1, dx=dlat_prev, dw=G.acc(pa + "norm_latents.weight", (D,)),
                   db=G.acc(pa + "norm_latents.bias", (D,)))
        dlat = dlat_prev
        ready([k for k in P if k.startswith(pa) or k.startswith(pf)])
    ops.reduce_rows(dlat, G.acc("latents", tuple(P["latents"].shape)))           # sum over (b, T) of the repeat
    if S.get("embs", False):
        v = Fv // frames
        if "frame_embs" in P:
            # rows >= frames keep zero gradient
            ops.reduce_rows_strided(dx, v, frames, G.acc("frame_embs", tuple(P["frame_embs"].shape)))
        if "media_time_embs" in P:
            ops.reduce_rows_strided(dx, Fv, T, G.acc("media_time_embs", tuple(P["media_time_embs"].shape)))
    ready([k for k in ("latents", "frame_embs", "media_time_embs") if k in P])
    return (dx if want_dx else None), g
