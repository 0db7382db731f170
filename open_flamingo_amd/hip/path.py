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

    def __init__(self, sinks, dev, fresh=(), sumsq=None):
        self.sinks, self.dev, self.g, self.fresh = (sinks or {}), dev, {}, fresh
        # sumsq: name -> fp32 slots that take the gradient's sum of squares from the dW GEMM's epilogue (OfGemmArgs.sumsq_out; the
        # step epilogue then skips that matrix in its global-norm pass, train/optim.py).  Only for gradients that go straight into
        # their sink; sumsq_done[name] says whether the launch honoured it.
        self.sumsq, self.sumsq_done = (sumsq or {}), {}

    def dw_gemm(self, ops, name, shape, A, B, **kw):
        """The weight gradient `name` = A^T B (TN, fp32) into its buffer, with its sum of squares into its slots where it has some."""
        t, beta = self.mat(name, shape)
        slots = self.sumsq.get(name) if name in self.sinks else None
        if slots is None:
            ops.gemm(A, B, t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta, **kw)
            if name in self.sumsq:
                self.sumsq_done[name] = False
        else:
            self.sumsq_done[name] = bool(ops.gemm(A, B, t, ta=True, tb=True, epi=EPI_ACC_F32, beta=beta, sumsq=slots, **kw))
        return t

    def mat(self, name, shape):
        """Buffer for a GEMM-produced gradient and the beta to use with it (0 for a sink in ``fresh``: its content is
        stale -- the step epilogue left it uncleared on purpose -- and this backward is the first to write it)."""
        t = self.sinks.get(name)
        beta = 0.0 if name in self.fresh else 1.0
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
# FeedForward (helpers.py:15-22) -- as a branch: out = [resid +] [tanh(gate) *] W2 gelu(W1 LN(x))
# =================================================================================================
def _branch_out(ops, A, Wt, x_like, resid, gate):
    """Final GEMM of a branch.  With a residual: resid + tanh(gate) * acc in the stream dtype, fused in the epilogue;
    stand-alone (resid None): tanh(gate) * acc (gate None: acc) in the stream dtype of ``x_like``."""
    out = torch.empty_like(x_like)
    if resid is not None:
        ops.gemm(A, Wt, out, epi=EPI_GATE_RESID, aux=resid, gate=gate)
    elif out.dtype == F32:
        ops.gemm(A, Wt, out, epi=EPI_ACC_F32, gate=gate)
    else:
        ops.gemm(A, Wt, out, epi=EPI_STORE_BF16, gate=gate)
    return out


def feed_forward_fwd(ops, P, W, x, *, prefix="", gate=None, residual=False, keep=True, pre_ln=None):
    """x (rows, d) stream dtype.  P/W keys: prefix + {0.weight, 0.bias, 1.weight, 3.weight}.  Returns (y, saved).
    pre_ln = (u, st): LN(x) in bf16 and its statistics, when the producer of x already computed them (the fused attention branch)."""
    dev = x.device
    rows, d = x.shape
    hid = W[prefix + "1.weight"].shape[0]
    if pre_ln is not None:
        u, st = pre_ln
    else:
        u = _e((rows, d), BF16, dev)
        st = _e((rows, 2), F32, dev)
        ops.ln_fwd(x, P[prefix + "0.weight"], P[prefix + "0.bias"], u, st)
    a = _e((rows, hid), BF16, dev) if keep else None      # pre-GELU activations: only the backward reads them
    b = _e((rows, hid), BF16, dev)
    ops.gemm(u, W[prefix + "1.weight"], b, epi=EPI_GELU, out2=a)               # up-projection + erf GELU
    y = _branch_out(ops, b, W[prefix + "3.weight"], x, x if residual else None, gate)   # down [, *tanh(gate), +x]
    if not keep:
        return y, None
    return y, dict(x=x, u=u, st=st, a=a, b=b)


# ---- bf16 twins of fp32 gradient-stream tensors, handed from one module's backward to the next -------------------------------
# Every backward on the fp32 stream ends in a LayerNorm backward that can emit the bf16 copy of its dx in the same pass (+2 B per
# element) and every backward starts by casting its incoming gradient to bf16 for its GEMMs (a pass of its own: 6 B per element,
# 60 launches per step at cfg-2).  The producer offers the twin, the consumer takes it: matched by storage address, element
# count and version counter of the fp32 tensor, which the entry keeps ALIVE (its address cannot be recycled while the entry
# exists); an entry is consumed by its first taker, a scope holds at most four (a gradient nobody takes ages out).
#
# The registry lives in a Scope that belongs to ONE model: Flamingo.__init__ makes one and hands it to every module of its tree
# (`_of_scope`), the autograd Functions pass their module's scope down here.  Two models in one process therefore never see each
# other's entries; modules used on their own share DEFAULT_SCOPE.
class _SharedCache:
    """per-forward artefacts shared by all blocks that see the same media / media_locations tensors (the reference recomputes them
    in each of the 24 blocks: helpers.py:187-189, 199-208); identity-keyed, a handful of entries"""

    def __init__(self, cap=8):
        self.cap, self.items = cap, []

    def get(self, key_tensor, tag, make):
        key = (key_tensor.data_ptr(), key_tensor._version, tuple(key_tensor.shape), key_tensor.dtype, tag)
        for k, src, val in self.items:
            if k == key and src is key_tensor:
                return val
        val = make()
        self.items.append((key, key_tensor, val))
        if len(self.items) > self.cap:
            self.items.pop(0)
        return val


class Scope:
    """Mutable per-model host state of the product path: the bf16-twin registry and the shared per-forward artefacts."""

    def __init__(self):
        self.twins = []
        self.shared = _SharedCache()

    def clear(self):
        self.twins.clear()
        self.shared.items.clear()


DEFAULT_SCOPE = Scope()
TWINS = True          # constant; tools/ab_bf16_twins.py clears it for the same-box A/B


def scope_of(mod):
    """the Scope a module's forward / backward works in: its model's, or DEFAULT_SCOPE for a module used on its own"""
    return getattr(mod, "_of_scope", None) or DEFAULT_SCOPE


def adopt(model, scope=None):
    """Give every module of ``model``'s tree one Scope (a new one unless given).  Called by Flamingo.__init__."""
    scope = scope or Scope()
    for m in model.modules():
        m.__dict__["_of_scope"] = scope
    return scope


def offer_bf16_twin(t, twin, scope=None):
    """t: fp32 gradient tensor about to be returned from a backward; twin: bf16 tensor with the same values."""
    if t.dtype != F32 or twin is None or twin.numel() != t.numel():
        return
    twins = (scope or DEFAULT_SCOPE).twins
    twins.append((t, t._version, twin))
    del twins[:-4]


def take_bf16_twin(t, scope=None):
    """bf16 twin of the contiguous fp32 tensor t (any view of what was offered), or None."""
    twins = (scope or DEFAULT_SCOPE).twins
    for i, (src, ver, twin) in enumerate(twins):
        if (src.data_ptr() == t.data_ptr() and src.numel() == t.numel() and src._version == ver and t._version == ver
                and t.dtype == F32 and t.is_contiguous() and twin.device == t.device):
            del twins[i]
            return twin.view(t.shape)
    return None


def bf16_of(ops, t, scope=None):
    """t as a bf16 GEMM operand: the twin a producer offered, else a cast pass."""
    twin = take_bf16_twin(t, scope) if t.dtype == F32 else None
    return twin if twin is not None else ops.to_bf16(t)


def feed_forward_bwd(ops, P, W, S, dy, G, *, prefix="", gate=None, gate_name=None, residual=False, scope=None):
    """dy (rows, d) stream dtype, contiguous.  Returns (dx, dx_bf16 or None): dx = [dy +] LN_bwd(...)."""
    dev = dy.device
    rows, d = S["x"].shape
    hid = W[prefix + "1.weight"].shape[0]
    dyb = bf16_of(ops, dy, scope)
    da = _e((rows, hid), BF16, dev)
    ops.gemm(dyb, W[prefix + "3.weight"], da, tb=True, epi=EPI_DGELU_DOT, aux=S["a"], gate=gate,
             dot=G.acc(gate_name, (1,)) if gate_name else None)
    G.dw_gemm(ops, prefix + "3.weight", (d, hid), dyb, S["b"], gate=gate)                    # dW2
    du = _e((rows, d), BF16, dev)
    ops.gemm(da, W[prefix + "1.weight"], du, tb=True)
    G.dw_gemm(ops, prefix + "1.weight", (hid, d), da, S["u"])                                # dW1
    dx = torch.empty_like(dy)
    dxb = _e((rows, d), BF16, dev) if dy.dtype == F32 else None
    ops.ln_bwd(du, S["x"], S["st"], P[prefix + "0.weight"], resid=dy if residual else None, dx=dx, dx_bf16=dxb,
               dw=G.acc(prefix + "0.weight", (d,)), db=G.acc(prefix + "0.bias", (d,)))
    return dx, (dxb if dxb is not None else dx)


# =================================================================================================
# MaskedCrossAttention (helpers.py:160-233) -- as a branch: out = [x +] [tanh(gate) *] attn(x, media)
# =================================================================================================
def xattn_project_media(ops, W, media_bf, heads, out=None, prefix="attn.", dim_head=64):
    """k | v = to_kv(media) (helpers.py:189), (B*T*n, 2*inner) bf16.  Depends on the media and the block's weights only,
    so the decode loop computes it once per block and prompt instead of once per generated token."""
    kv = _e((media_bf.shape[0], 2 * heads * dim_head), BF16, media_bf.device) if out is None else out
    ops.gemm(media_bf, W[prefix + "to_kv.weight"], kv)
    return kv


def _softmax_scale(scale, dim_head):
    """the reference's dim_head ** -0.5 (helpers.py:28,140) -- given explicitly when the heads are zero-padded to a kernel size"""
    return dim_head ** -0.5 if scale is None else float(scale)


FUSED_XATTN = True    # constant; tools/ab_fused_xattn.py clears it for the same-box A/B (the five separate launches)


class WeightDict(dict):
    """name -> bf16 GEMM operand of a module's weights (src/helpers.py: _weights_bf16); name + '#pk' -> its fragment-major copy.
    pk_cache: {name: (the bf16 tensor the copy was made from, copy)} kept by the owning module ACROSS forwards, or None when the
    bf16 tensors are rewritten in place (a provider keeps them current: their identity says nothing about their content)."""
    pk_cache = None


def packed_weight(ops, W, name):
    """fragment-major copy of the bf16 weight W[name] (of_pack_frag16): the provider's (train/optim.py re-packs once per optimizer
    step), else the owning module's cached one while the bf16 tensor it was made from is still the operand (eval mode), else made now"""
    key = name + "#pk"
    pk = W.get(key)
    if pk is None:
        src, cache = W[name], getattr(W, "pk_cache", None)
        ent = cache.get(name) if cache is not None else None
        if ent is not None and ent[0] is src:
            pk = ent[1]
        else:
            pk = ops.pack_frag16(src)
            if cache is not None:
                cache[name] = (src, pk)
        W[key] = pk
    return pk


def masked_cross_attention_fwd(ops, P, W, x, media_bf, tt, *, B, L, T, n, heads, only_immediate, prefix="attn.", gate=None,
                               residual=False, safe=0, kv=None, dim_head=64, scale=None, keep=True, next_ln=None):
    """x (B*L, d) stream dtype; media_bf (B*T*n, Dv) bf16; tt (B, L) int32 or None.  Returns (y, saved).
    next_ln = (weight, bias) of a LayerNorm the caller applies to y next (the block's FeedForward): when the branch runs as the ONE
    fused launch (of_xattn_fused_fwd) that LayerNorm leaves with it and saved["next_ln"] = (LN(y) bf16, statistics)."""
    dev = x.device
    rows, d = x.shape
    inner = heads * dim_head
    if kv is None:
        kv = xattn_project_media(ops, W, media_bf, heads, prefix=prefix, dim_head=dim_head)   # to_kv (k | v fused)
    sc = _softmax_scale(scale, dim_head)
    if FUSED_XATTN and residual and safe == 0 and dim_head == 64 and ops.xattn_fused_fwd(
            x, P[prefix + "norm.weight"], P[prefix + "norm.bias"], None, kv[:, :inner], kv[:, inner:], tt, None, gate, x, B=B, L=L,
            Lk=T * n, heads=heads, head_dim=dim_head, n_per_media=n, T_img=T, only_immediate=only_immediate, scale=sc,
            probe_only=True):
        y = torch.empty_like(x)
        xn = _e((rows, d), BF16, dev) if keep else None
        st = _e((rows, 2), F32, dev) if keep else None
        q = _e((rows, inner), BF16, dev) if keep else None
        o = _e((rows, inner), BF16, dev) if keep else None
        lse = _e((B, heads, L), F32, dev) if keep else None
        u2 = st2 = None
        if next_ln is not None:
            u2, st2 = _e((rows, d), BF16, dev), _e((rows, 2), F32, dev)
        ok = ops.xattn_fused_fwd(x, P[prefix + "norm.weight"], P[prefix + "norm.bias"], packed_weight(ops, W, prefix + "to_q.weight"),
                                 kv[:, :inner], kv[:, inner:], tt, packed_weight(ops, W, prefix + "to_out.weight"), gate, y, B=B, L=L,
                                 Lk=T * n, heads=heads, head_dim=dim_head, n_per_media=n, T_img=T, only_immediate=only_immediate,
                                 scale=sc, ln2_w=next_ln[0] if next_ln else None, ln2_b=next_ln[1] if next_ln else None, u2=u2,
                                 st2=st2, xn=xn, st=st, q=q, o=o, lse=lse)
        assert ok
        S = dict(x=x, xn=xn, st=st, q=q, kv=kv, o=o, lse=lse)
        if next_ln is not None:
            S["next_ln"] = (u2, st2)
        return y, S
    xn = _e((rows, d), BF16, dev)
    st = _e((rows, 2), F32, dev)
    ops.ln_fwd(x, P[prefix + "norm.weight"], P[prefix + "norm.bias"], xn, st)
    q = _e((rows, inner), BF16, dev)
    ops.gemm(xn, W[prefix + "to_q.weight"], q)                                   # to_q
    o = _e((rows, inner), BF16, dev)
    lse = _e((B, heads, L), F32, dev)
    ops.attn_fwd(q, kv[:, :inner], kv[:, inner:], o, lse, batch=B, Lq=L, Lk=T * n, heads=heads, text_time=tt,
                 n_per_media=n, T_img=T, only_immediate=only_immediate, safe=safe, head_dim=dim_head, scale=sc)
    y = _branch_out(ops, o, W[prefix + "to_out.weight"], x, x if residual else None, gate)   # to_out [, *tanh(gate), +x]
    return y, dict(x=x, xn=xn, st=st, q=q, kv=kv, o=o, lse=lse)


def masked_cross_attention_bwd(ops, P, W, S, media_bf, tt, dy, dyb, G, *, B, L, T, n, heads, only_immediate,
                               prefix="attn.", gate=None, gate_name=None, residual=False, need_dmedia=True, safe=0,
                               dim_head=64, dkv_out=None, offer_twin=False, scope=None, scale=None):
    """dy stream dtype + its bf16 copy dyb.  Returns (dx, dmedia fp32 or None).
    dkv_out: (B*T*n, 2*inner) bf16 destination of d(k|v) owned by the caller (a column block of the buffer shared by all
    blocks when their to_kv projections are grouped); the caller then forms the media gradient of all
    blocks in one GEMM and passes need_dmedia=False."""
    dev = dy.device
    rows, d = S["x"].shape
    inner = heads * dim_head
    Dv = media_bf.shape[1]
    dO = _e((rows, inner), BF16, dev)
    ops.gemm(dyb, W[prefix + "to_out.weight"], dO, tb=True, epi=EPI_SCALE_DOT, aux=S["o"], gate=gate,
             dot=G.acc(gate_name, (1,)) if gate_name else None)
    # The three 512-wide weight gradients of the branch (to_out, to_q, to_kv: off the critical path, 64 tiles of 128 x 128 each) go
    # out as ONE batched launch at the end (ops.gemm_batch_dw -> of_gemm_batch): separately they were three split-K launches + three
    # reduce launches, 103 us per block in-step for 43 GFLOP (profiles/r04_final_default_gemm_report.jsonl)
    dw_batch = []
    t, beta = G.mat(prefix + "to_out.weight", (d, inner))
    dw_batch.append((dyb, S["o"], t, beta, gate))
    dq = _e((rows, inner), BF16, dev)
    dkv = _e((B * T * n, 2 * inner), BF16, dev) if dkv_out is None else dkv_out
    delta = _e((B, heads, L), F32, dev)
    kv = S["kv"]
    ops.attn_bwd(S["q"], kv[:, :inner], kv[:, inner:], S["o"], S["lse"], dO, dq, dkv[:, :inner], dkv[:, inner:], delta,
                 batch=B, Lq=L, Lk=T * n, heads=heads, text_time=tt, n_per_media=n, T_img=T,
                 only_immediate=only_immediate, safe=safe, head_dim=dim_head, scale=_softmax_scale(scale, dim_head))
    dxn = _e((rows, d), BF16, dev)
    ops.gemm(dq, W[prefix + "to_q.weight"], dxn, tb=True)
    t, beta = G.mat(prefix + "to_q.weight", (inner, d))
    dw_batch.append((dq, S["xn"], t, beta, None))
    dx = torch.empty_like(dy)
    dxb = _e((rows, d), BF16, dev) if (offer_twin and TWINS and dy.dtype == F32) else None     # for the next backward down the stream
    ops.ln_bwd(dxn, S["x"], S["st"], P[prefix + "norm.weight"], resid=dy if residual else None, dx=dx, dx_bf16=dxb,
               dw=G.acc(prefix + "norm.weight", (d,)), db=G.acc(prefix + "norm.bias", (d,)))
    if dxb is not None:
        offer_bf16_twin(dx, dxb, scope)
    t, beta = G.mat(prefix + "to_kv.weight", (2 * inner, Dv))
    dw_batch.append((dkv, media_bf, t, beta, None))
    ops.gemm_batch_dw(dw_batch)
    dmedia = None
    if need_dmedia:
        dmedia = _e((B * T * n, Dv), F32, dev)
        ops.gemm(dkv, W[prefix + "to_kv.weight"], dmedia, tb=True, epi=EPI_ACC_F32)
    return dx, dmedia


# =================================================================================================
# GatedCrossAttentionBlock (helpers.py:260-279) = the two gated branches with fused residuals
# =================================================================================================
def xattn_block_fwd(ops, P, W, x, media_bf, tt, *, B, L, T, n, heads, only_immediate, safe=0, kv=None, keep=True,
                    dim_head=64, scale=None):
    """x (B*L, d) stream dtype; media_bf (B*T*n, Dv) bf16; tt (B, L) int32 or None.  Returns (y, saved).
    kv: projected media from xattn_project_media (else computed here).  keep=False (inference): nothing is saved for a
    backward -- the pre-GELU activations are not written -- and saved is None."""
    y1, Sa = masked_cross_attention_fwd(ops, P, W, x, media_bf, tt, B=B, L=L, T=T, n=n, heads=heads,
                                        only_immediate=only_immediate, gate=P["attn_gate"], residual=True, safe=safe,
                                        kv=kv, dim_head=dim_head, scale=scale, keep=keep, next_ln=(P["ff.0.weight"], P["ff.0.bias"]))
    y2, Sf = feed_forward_fwd(ops, P, W, y1, prefix="ff.", gate=P["ff_gate"], residual=True, keep=keep, pre_ln=Sa.pop("next_ln", None))
    if not keep:
        return y2, None
    return y2, dict(attn=Sa, ff=Sf)


def xattn_block_bwd(ops, P, W, S, media_bf, tt, dy, *, B, L, T, n, heads, only_immediate, need_dmedia=True, safe=0,
                    sinks=None, dim_head=64, fresh=(), dkv_out=None, scope=None, scale=None, sumsq=None):
    """Returns (dx, dmedia fp32 (B*T*n, Dv) or None, grads dict keyed like P).  sinks / fresh / sumsq: see _GradOut
    (the grads dict carries `sumsq_done` under the key "__sumsq_done__" when sumsq was given)."""
    G = _GradOut(sinks, dy.device, fresh, sumsq)
    dy = dy.contiguous()
    # ---- feed forward branch: y2 = y1 + tanh(gf) * F(y1)
    dy1, dy1b = feed_forward_bwd(ops, P, W, S["ff"], dy, G, prefix="ff.", gate=P["ff_gate"], gate_name="ff_gate",
                                 residual=True, scope=scope)
    # ---- attention branch: y1 = x + tanh(ga) * A(x, media)
    dx, dmedia = masked_cross_attention_bwd(ops, P, W, S["attn"], media_bf, tt, dy1, dy1b, G, B=B, L=L, T=T, n=n,
                                            heads=heads, only_immediate=only_immediate, gate=P["attn_gate"],
                                            gate_name="attn_gate", residual=True, need_dmedia=need_dmedia, safe=safe,
                                            dim_head=dim_head, dkv_out=dkv_out, offer_twin=True, scope=scope, scale=scale)
    if sumsq:
        G.g["__sumsq_done__"] = G.sumsq_done
    return dx, dmedia, G.g


# =================================================================================================
# PerceiverResampler
# =================================================================================================
def perceiver_attention_fwd(ops, P, W, x, lat, *, N, Fv, n, heads, prefix, residual=False, safe=0, dim_head=64, scale=None):
    """PerceiverAttention.forward (helpers.py:39-65).  x (N*Fv, D) media rows, lat (N*n, D) latents, both stream dtype.
    Returns (out (N*n, D) = attn(x, latents) [+ latents], saved)."""
    dev = x.device
    D = x.shape[1]
    inner = heads * dim_head
    S_ = Fv + n
    kvin = _e((N * S_, D), BF16, dev)          # [LN_media(x) rows | LN_latents(latents) rows] per media item
    st_m = _e((N * Fv, 2), F32, dev)
    st_l = _e((N * n, 2), F32, dev)
    ltn = _e((N * n, D), BF16, dev)
    ops.ln_fwd_grouped(x, P[prefix + "norm_media.weight"], P[prefix + "norm_media.bias"], kvin, D, Fv, S_ * D, None, st_m)
    ops.ln_fwd_grouped(lat, P[prefix + "norm_latents.weight"], P[prefix + "norm_latents.bias"], kvin[Fv:], D, n, S_ * D,
                       ltn, st_l)
    q = _e((N * n, inner), BF16, dev)
    ops.gemm(ltn, W[prefix + "to_q.weight"], q)
    kv = _e((N * S_, 2 * inner), BF16, dev)
    ops.gemm(kvin, W[prefix + "to_kv.weight"], kv)
    o = _e((N * n, inner), BF16, dev)
    lse = _e((N, heads, n), F32, dev)
    ops.attn_fwd(q, kv[:, :inner], kv[:, inner:], o, lse, batch=N, Lq=n, Lk=S_, heads=heads, safe=safe, head_dim=dim_head,
                 scale=_softmax_scale(scale, dim_head))
    out = _branch_out(ops, o, W[prefix + "to_out.weight"], lat, lat if residual else None, None)
    return out, dict(x=x, lat=lat, kvin=kvin, st_m=st_m, st_l=st_l, ltn=ltn, q=q, kv=kv, o=o, lse=lse)


def perceiver_attention_bwd(ops, P, W, S, dout, doutb, G, *, N, Fv, n, heads, prefix, residual=False, need_dx=False,
                            dx_acc=None, safe=0, dim_head=64, scale=None):
    """dout (N*n, D) stream dtype (+ bf16 copy).  Returns (dlat, dx): dlat = [dout +] gradient through norm_latents;
    dx = (dx_acc or 0) + gradient through norm_media when need_dx, else None (norm_media's dw/db are always produced)."""
    dev = dout.device
    x, lat = S["x"], S["lat"]
    D = x.shape[1]
    inner = heads * dim_head
    S_ = Fv + n
    dO = _e((N * n, inner), BF16, dev)
    ops.gemm(doutb, W[prefix + "to_out.weight"], dO, tb=True)
    # the layer's three projection weight gradients (to_out, to_q, to_kv: 32 - 64 tiles of 128 x 128 each, off the critical path) leave as
    # ONE batched launch at the end, like the gated blocks' (ops.gemm_batch_dw -> of_gemm_batch: the bits of the separate launches)
    dw_batch = []
    t, beta = G.mat(prefix + "to_out.weight", (D, inner))
    dw_batch.append((doutb, S["o"], t, beta, None))
    dq = _e((N * n, inner), BF16, dev)
    dkv = _e((N * S_, 2 * inner), BF16, dev)
    delta = _e((N, heads, n), F32, dev)
    kv = S["kv"]
    ops.attn_bwd(S["q"], kv[:, :inner], kv[:, inner:], S["o"], S["lse"], dO, dq, dkv[:, :inner], dkv[:, inner:],
                 delta, batch=N, Lq=n, Lk=S_, heads=heads, safe=safe, head_dim=dim_head, scale=_softmax_scale(scale, dim_head))
    dltn = _e((N * n, D), BF16, dev)
    ops.gemm(dq, W[prefix + "to_q.weight"], dltn, tb=True)                      # through to_q
    t, beta = G.mat(prefix + "to_q.weight", (inner, D))
    dw_batch.append((dq, S["ltn"], t, beta, None))
    dkvin = _e((N * S_, D), BF16, dev)
    ops.gemm(dkv, W[prefix + "to_kv.weight"], dkvin, tb=True)                    # through to_kv (media + latent rows)
    t, beta = G.mat(prefix + "to_kv.weight", (2 * inner, D))
    dw_batch.append((dkv, S["kvin"], t, beta, None))
    ops.gemm_batch_dw(dw_batch)
    # norm_media: parameter grads always; dx only if the vision features require grad (they do not in Flamingo,
    # flamingo.py:194-195 runs the ViT under no_grad)
    dx_new = torch.empty_like(x) if need_dx else None
    ops.ln_bwd(dkvin, x, S["st_m"], P[prefix + "norm_media.weight"], lddy=D, dy_grp_rows=Fv, dy_grp_stride=S_ * D,
               resid=dx_acc if need_dx else None, dx=dx_new, dw=G.acc(prefix + "norm_media.weight", (D,)),
               db=G.acc(prefix + "norm_media.bias", (D,)))
    # norm_latents: two upstream gradients (k/v rows of kv_input and the to_q input) [+ the residual]
    dlat = torch.empty_like(dout)
    ops.ln_bwd(dkvin[Fv:], lat, S["st_l"], P[prefix + "norm_latents.weight"], lddy=D, dy_grp_rows=n,
               dy_grp_stride=S_ * D, dy2=dltn, resid=dout if residual else None, dx=dlat,
               dw=G.acc(prefix + "norm_latents.weight", (D,)), db=G.acc(prefix + "norm_latents.bias", (D,)))
    return dlat, dx_new


def perceiver_fwd(ops, P, W, x, *, N, Fv, n, heads, depth, T=1, frames=1, safe=0, keep=True, dim_head=64, scale=None):
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
    lat = _e((N * n, D), x.dtype, dev)
    ops.broadcast_rows(P["latents"], lat, N * n)                              # repeat(latents, 'n d -> b T n d')
    layers = []
    for i in range(depth):
        lat1, Sa = perceiver_attention_fwd(ops, P, W, x, lat, N=N, Fv=Fv, n=n, heads=heads, prefix=f"layers.{i}.0.",
                                           residual=True, safe=safe, dim_head=dim_head, scale=scale)   # attn(x, latents) + latents
        lat, Sf = feed_forward_fwd(ops, P, W, lat1, prefix=f"layers.{i}.1.", residual=True, keep=keep)   # ff + latents
        if keep:
            layers.append(dict(attn=Sa, ff=Sf))
    out = torch.empty_like(lat)
    st_o = _e((N * n, 2), F32, dev)
    ops.ln_fwd_out(lat, P["norm.weight"], P["norm.bias"], out, st_o)
    if not keep:
        return out, None
    return out, dict(layers=layers, lat_last=lat, st_o=st_o, x=x, embs=embs, T=T, frames=frames)


def perceiver_bwd(ops, P, W, S, dout, *, N, Fv, n, heads, depth, T=1, frames=1, need_dx=False, safe=0, sinks=None,
                  on_ready=None, dim_head=64, fresh=(), scale=None):
    """Returns (dx (N*Fv, D) stream dtype or None, grads dict keyed like P).  sinks: see _GradOut.
    on_ready(names): called as soon as the kernels producing the FINAL value of those parameters' gradients are enqueued
    (per layer, last layer first) -- the gradient exchange of a layer can then start while the earlier layers' backward
    still runs (train/reducer.py buckets the Perceiver per layer)."""
    ready = on_ready or (lambda names: None)
    dev = dout.device
    want_dx = need_dx
    need_dx = need_dx or S.get("embs", False)     # the position tables' gradients are reductions of dx
    D = S["x"].shape[1]
    G = _GradOut(sinks, dev, fresh)
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
        dlat1, dlat1b = feed_forward_bwd(ops, P, W, Lr["ff"], dlat, G, prefix=pf, residual=True)   # ff(latents) + latents
        dlat, dx = perceiver_attention_bwd(ops, P, W, Lr["attn"], dlat1, dlat1b, G, N=N, Fv=Fv, n=n, heads=heads,
                                           prefix=pa, residual=True, need_dx=need_dx, dx_acc=dx, safe=safe,
                                           dim_head=dim_head, scale=scale)                           # attn + latents
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
