"""Whole frozen transformer blocks as ONE autograd node each (the frozen half of the train step).

The reference runs the frozen towers as stock modules (open_flamingo/src/flamingo_lm.py:63-65 calls the HF decoder layer,
flamingo.py:194-195 the CLIP tower under no_grad).  Under amp_bf16 every residual add of those blocks is a separate
mixed-dtype element-wise kernel (fp32 stream + bf16 branch), and the backward re-creates them as gradient adds and
fp32 -> bf16 casts in front of every dX GEMM.  Nothing in a frozen block has a weight gradient, so the whole block is a
fixed chain  LN -> GEMM -> attention -> GEMM -> (+, LN) -> GEMM -> GELU -> GEMM -> +  whose backward is the same chain
reversed with dX GEMMs only.  Here that chain is written out once:

  * the plain GEMMs stay on the vendor library (torch.mm -> hipBLASLt; frozen bf16 weights, no epilogue worth fusing:
    DESIGN.md 4.1 measures this repository's fused GELU / residual epilogues at break-even against hipBLASLt + one
    element-wise pass);
  * LayerNorm, residual add + LayerNorm in one pass (of_layernorm_fwd_add), causal + ALiBi attention, and in backward
    LayerNorm-backward + residual-gradient add + bf16 operand copy in one pass (of_layernorm_bwd) run on libofhip;
  * no activation that only a weight gradient would need is kept.

Arithmetic = the eager chain's: fp32 residual stream, bf16 GEMM operands and outputs, fp32 LayerNorm statistics, exact
(erf) GELU on the bf16 pre-activation.  Anything the fused form does not cover (KV cache, dropout, trainable weights, fp32
weights, no autocast, CPU tensors) takes the module's original forward.
"""
import types

import torch
from torch import nn

from ..hip import path as _path

import os

BF16 = torch.bfloat16
F32 = torch.float32
# dX GEMMs of the frozen blocks against pre-transposed weight copies (see _transposed); same-box A/B, round 2: 123.2 -> 121.8 ms
# per step (a tool that wants the other arm sets this module attribute; the product reads no environment switch)
_DX_PRETRANSPOSED = True
# Frozen MPT MLP: which of its two GEMMs run as ONE of_gemm launch with a fused epilogue instead of vendor GEMM + element-wise pass
# (constants, set from measurements: DESIGN.md 4.9; tools/ab_frozen_mlp.py flips them for the same-box A/B)
_MLP_FUSED_UP = False      # up_proj + erf-GELU (OF_EPI_GELU; the pre-activation is kept for the backward)
_MLP_FUSED_DOWN = False    # down_proj + residual add into the fp32 stream (OF_EPI_GATE_RESID without a gate)
_MLP_FUSED_DGELU = True    # backward: (dY Wdown) * gelu'(h) as ONE NN launch (OF_EPI_DGELU_DOT without a gate / dot) instead of
                           # vendor GEMM + of_gelu_bwd pass.  Round 4 (256x256 kernel): a wash at step level (profiles/r04k_*, r04_final_ab_frozen_mlp.txt).
                           # Round 5: of_gemm sends this launch to the two-workgroups-per-CU kernel (gemm_w4h.hip) -- same box, alternating:
                           # 120.13 / 119.98 -> 118.40 / 118.62 ms per step (-1.5 ms: the of_gelu_bwd pass is gone and the fused launch
                           # costs less than vendor GEMM + pass); up fused +2.0, down fused 0.0, all three +1.5 (profiles/r05o_ab_frozen_mlp.txt;
                           # with 20-step arms, r05s_*: up +0.55, down -0.1, up + down +1.15 on top of this one)
                           # -> ON; the other two stay off
# Frozen MPT block (no biases): its plain GEMMs -- Wqkv, out_proj, up_proj, down_proj forward; the four dX backward -- as of_gemm launches
# (NT forward, NN backward on the weight as it lies: no transposed copies) instead of torch.mm -> hipBLASLt.  Set from a same-box A/B
# (tools/ab_frozen_mlp.py, fourth arm digit): 106.5 / 106.6 ms per step on the vendor library, 110.1-110.2 native (+3.6 ms;
# profiles/r05s_ab_frozen_mpt_gemms_native.txt) -- in a hot loop the two libraries are within 0-6 % of each other on these shapes, behind
# the streaming passes of a step (operands not in the Infinity Cache) this library's kernel loses 7-22 %, the vendor's 3-5 % (DESIGN.md
# 4.12; the K rotation took back half) -> OFF.
_MPT_GEMMS_NATIVE = False


def _ops():
    from ..hip.ops import Ops
    return Ops.default()


def _zero_bias(norm):
    """HF MPT drops the LayerNorm bias; the kernels take one."""
    b = norm.bias
    if b is None:
        b = norm.__dict__.get("_of_zero_bias")
        if b is None or b.device != norm.weight.device:
            b = norm.__dict__["_of_zero_bias"] = torch.zeros_like(norm.weight)
    return b


def _transposed(mod, name, w):
    """W^T of a frozen nn.Linear weight as its own contiguous bf16 matrix, cached on the owning module (rebuilt if the weight
    is replaced or written).  The backward of a frozen Linear is dX = dY W: as torch.mm(dY, W) the vendor library sees an
    'NN' product whose B operand is strided along K, as torch.mm(dY, (W^T)^T) the same 'NT' product the forward layers use --
    129 vs 197 us for the 8192 x 2048 x 8192 dX of up_proj (train/tuned/).  Costs one extra bf16 copy of the frozen block
    weights (2.4 GB at MPT-1B), made once: the weights never change."""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    cache = mod.__dict__.setdefault("_of_wt", {})
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = cache[name] = (key, w.detach().t().contiguous())
    return hit[1]


def _transposed_unless_fused(mod, name, w):
    """the MLP's second matrix: with _MLP_FUSED_DGELU its backward runs as of_gemm's NN launch on W itself, so no transposed copy is
    made (and one cached by an earlier configuration is dropped)"""
    if _MLP_FUSED_DGELU:
        mod.__dict__.get("_of_wt", {}).pop(name, None)
        return None
    return _transposed(mod, name, w)


def _mm_dx(dy, w, wt):
    """dX = dY W for a frozen nn.Linear weight W (out, in); wt = its cached transpose or None."""
    return torch.mm(dy, wt.t()) if wt is not None else torch.mm(dy, w)


def _lin(ops, x, w):
    """y = x W^T for a bias-free frozen nn.Linear (MPT): of_gemm's NT launch, or the vendor library."""
    if not _MPT_GEMMS_NATIVE:
        return torch.mm(x, w.t())
    y = torch.empty(x.shape[0], w.shape[0], dtype=BF16, device=x.device)
    ops.gemm(x, w, y)
    return y


def _lin_dx(ops, dy, w, wt):
    """dX = dY W of the same layer: of_gemm's NN launch on W as it lies (no transposed copy), or the vendor library on the cached W^T."""
    if not _MPT_GEMMS_NATIVE:
        return _mm_dx(dy, w, wt)
    dx = torch.empty(dy.shape[0], w.shape[1], dtype=BF16, device=dy.device)
    ops.gemm(dy, w, dx, tb=True)
    return dx


class _FrozenMptBlockFn(torch.autograd.Function):
    """HF MptBlock (norm_1 -> Wqkv -> causal ALiBi attention -> out_proj -> + -> norm_2 -> up_proj -> GELU -> down_proj -> +)
    with frozen weights.  x: (B, L, d) fp32 residual stream; returns the new stream (fp32)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, Wqkv, Wo, Wup, Wdown, slopes, kv_len, heads, head_dim, scale, wts, scope=None):
        ops = _ops()
        B, L, d = x.shape
        rows = B * L
        x2 = x.reshape(rows, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        dev = x.device
        a = torch.empty(rows, d, dtype=BF16, device=dev)
        st1 = torch.empty(rows, 2, dtype=F32, device=dev)
        ops.ln_fwd(x2, w1, b1, a, st1)
        qkv = _lin(ops, a, Wqkv)                                     # (rows, 3d) bf16: q | k | v column blocks
        o = torch.empty(rows, d, dtype=BF16, device=dev)
        lse = torch.empty(B, heads, L, dtype=F32, device=dev)
        kw = dict(batch=B, Lq=L, Lk=L, heads=heads, scale=scale, head_dim=head_dim, causal=True, alibi_slopes=slopes,
                  kv_len=kv_len)
        ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, **kw)
        t = _lin(ops, o, Wo)
        x1 = torch.empty(rows, d, dtype=F32, device=dev)
        m = torch.empty(rows, d, dtype=BF16, device=dev)
        st2 = torch.empty(rows, 2, dtype=F32, device=dev)
        ops.ln_fwd_add(x2, t, x1, w2, b2, m, st2)                    # x1 = x + attn branch;  m = norm_2(x1)
        if _MLP_FUSED_UP:
            from ..hip import abi
            h = torch.empty(rows, Wup.shape[0], dtype=BF16, device=dev)
            g = torch.empty_like(h)
            ops.gemm(m, Wup, g, epi=abi.EPI_GELU, out2=h)            # g = gelu(m Wup^T) from the fp32 accumulator, h = pre-activation
        else:
            h = _lin(ops, m, Wup)
            g = ops.gelu_fwd(h)
        if _MLP_FUSED_DOWN:
            from ..hip import abi
            y = torch.empty(rows, d, dtype=F32, device=dev)
            ops.gemm(g, Wdown, y, epi=abi.EPI_GATE_RESID, aux=x1)    # y = x1 + g Wdown^T (no gate: plain residual), fp32 stream
        else:
            u = _lin(ops, g, Wdown)
            y = ops.add_bf16(x1, u)                                  # fp32 stream + bf16 branch -> fp32
        del g
        ctx.save_for_backward(x2, st1, qkv, o, lse, x1, st2, h, w1, w2, Wqkv, Wo, Wup, Wdown, slopes, kv_len)
        ctx.scope = scope        # the model's hip/path.py Scope (bf16-twin registry), or None = the default one
        ctx.kw, ctx.shape, ctx.wts = kw, (B, L, d), wts      # wts: (Wqkv^T, Wo^T, Wup^T, Wdown^T) or None: frozen, not autograd inputs
        return y.view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        ops = _ops()
        x2, st1, qkv, o, lse, x1, st2, h, w1, w2, Wqkv, Wo, Wup, Wdown, slopes, kv_len = ctx.saved_tensors
        tq, to, tu, td = ctx.wts if ctx.wts is not None else (None, None, None, None)
        B, L, d = ctx.shape
        rows = B * L
        dev = dy.device
        dy2 = dy.reshape(rows, d)
        if dy2.dtype != F32 or not dy2.is_contiguous():
            dy2 = dy2.to(F32).contiguous()
        if _MLP_FUSED_DGELU:
            from ..hip import abi
            dh = torch.empty(rows, Wdown.shape[1], dtype=BF16, device=dev)
            ops.gemm(_path.bf16_of(ops, dy2, ctx.scope), Wdown, dh, tb=True, epi=abi.EPI_DGELU_DOT, aux=h)     # (dY Wdown) * gelu'(h)
        else:
            dact = _lin_dx(ops, _path.bf16_of(ops, dy2, ctx.scope), Wdown, td)   # (rows, 4d); the bf16 copy the backward above left, or a cast
            dh = ops.gelu_bwd(dact, h, out=dact)                     # in place: dact * gelu'(h)
            del dact
        dm = _lin_dx(ops, dh, Wup, tu)                               # (rows, d)
        del dh
        dx1 = torch.empty(rows, d, dtype=F32, device=dev)
        dx1b = torch.empty(rows, d, dtype=BF16, device=dev)
        ops.ln_bwd(dm, x1, st2, w2, resid=dy2, dx=dx1, dx_bf16=dx1b)  # dx1 = dy + norm_2'(dm), plus its bf16 operand copy
        do = _lin_dx(ops, dx1b, Wo, to)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, ctx.kw["heads"], L, dtype=F32, device=dev)
        ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                     delta, **ctx.kw)
        da = _lin_dx(ops, dqkv, Wqkv, tq)                            # (rows, d)
        dxb = torch.empty(rows, d, dtype=BF16, device=dev) if _path.TWINS else None
        ops.ln_bwd(da, x2, st1, w1, resid=dx1, dx=dx1, dx_bf16=dxb)  # in place: dx = dx1 + norm_1'(da); + its bf16 twin for the
        _path.offer_bf16_twin(dx1, dxb, ctx.scope)                   # backward of whatever produced x
        return (dx1.view(B, L, d),) + (None,) * 15


# Debug switch (module attribute, set by the caller -- bench.py --check-right-padding, tests): verify on the host that every
# attention mask the fused blocks reduce to key counts really is a prefix of ones.  Off by default: the check synchronises.
CHECK_RIGHT_PADDING = False


class _LiteMask:
    """Stand-in for the (B, 1, L, L) boolean mask HF's MptModel.forward builds for its blocks.  transformers' eager-mask
    builder synchronises the host once per LM forward (a device scalar made with torch.tensor, an `if mask.all()`): the
    host then sits idle until the GPU has drained everything enqueued so far -- 58 of the 108 ms of host time per step
    were that wait (tools/host_profile.py), and a step whose host cannot run ahead of the GPU is at the mercy of the
    box's CPU load.  The fused blocks only need the per-sequence key counts; a block that falls back to its HF forward
    asks for the real mask (full())."""

    def __init__(self, am, B, L, device, build=None):
        self.am, self.B, self.L, self.device = am, B, L, device
        self._lens = None
        self._slopes = None
        self._full = None
        self._build, self._hf = build, None      # transformers' own mask builder for this call (GPT-NeoX fallbacks)

    def to(self, *a, **kw):          # MptModel.forward: create_causal_mask(...).to(torch.bool)
        return self

    def lens(self):
        if self.am is None:
            return None
        if self._lens is None:       # right padding (train/data.py pads on the right): real keys per sequence
            if CHECK_RIGHT_PADDING:  # debug: the per-sequence COUNT is only the mask if the ones are a prefix (one host sync)
                if not right_padded(self.am):
                    raise ValueError("fused frozen blocks: attention_mask is not right-padded (a 1 follows a 0 in some row); the "
                                     "kernels take per-sequence key counts -- left-padded / holey masks need the HF block forward")
            self._lens = self.am.ne(0).sum(-1).to(torch.int32).contiguous()
        return self._lens

    def slopes(self, position_bias):
        if self._slopes is None or self._slopes[0] is not position_bias:
            pb = position_bias[:, 0, :].float()
            self._slopes = (position_bias, (pb[:, -1] - pb[:, -2]).contiguous())
        return self._slopes[1]

    def full(self):
        """HF's boolean mask (True = masked): causal, plus the padded keys."""
        if self._full is None:
            i = torch.arange(self.L, device=self.device)
            m = (i[None, :] > i[:, None])[None, None].expand(self.B, 1, self.L, self.L)
            if self.am is not None:
                m = m | self.am.eq(0)[:, None, None, :]
            self._full = m.contiguous()
        return self._full


    def hf_mask(self):
        """What transformers' create_causal_mask would have handed the layers (built on demand: it synchronises the host)."""
        if self._hf is None:
            self._hf = (self._build(),)
        return self._hf[0]


_orig_create_causal_mask = None
_orig_neox_create_causal_mask = None


def _install_lite_mask_neox():
    """As _install_lite_mask, for GPTNeoXModel.forward (which also passes position_ids: the default arange(L) of a cache-less
    forward is what the fused blocks assume)."""
    global _orig_neox_create_causal_mask
    from transformers.models.gpt_neox import modeling_gpt_neox as mod
    if _orig_neox_create_causal_mask is not None or not hasattr(mod, "create_causal_mask"):
        return
    _orig_neox_create_causal_mask = mod.create_causal_mask

    def create_causal_mask(config=None, inputs_embeds=None, attention_mask=None, past_key_values=None, **kw):
        orig = lambda: _orig_neox_create_causal_mask(config=config, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                                                     past_key_values=past_key_values, **kw)
        lite = (getattr(config, "_of_lite_mask", False) and past_key_values is None and set(kw) <= {"position_ids"}
                and inputs_embeds is not None and inputs_embeds.dim() == 3
                and (attention_mask is None or (attention_mask.dim() == 2 and attention_mask.shape == inputs_embeds.shape[:2])))
        if not lite:
            return orig()
        return _LiteMask(attention_mask, inputs_embeds.shape[0], inputs_embeds.shape[1], inputs_embeds.device, build=orig)

    mod.create_causal_mask = create_causal_mask


def _install_lite_mask():
    """Route transformers' mask builder (the name MptModel.forward resolves in its module) through _LiteMask for language
    models that opted in (config._of_lite_mask, set by use_fused_frozen_mpt_blocks); every other call is untouched."""
    global _orig_create_causal_mask
    from transformers.models.mpt import modeling_mpt
    if _orig_create_causal_mask is not None or not hasattr(modeling_mpt, "create_causal_mask"):
        return
    _orig_create_causal_mask = modeling_mpt.create_causal_mask

    def create_causal_mask(config=None, inputs_embeds=None, attention_mask=None, past_key_values=None, **kw):
        lite = (getattr(config, "_of_lite_mask", False) and past_key_values is None and not kw and inputs_embeds is not None
                and inputs_embeds.dim() == 3
                and (attention_mask is None or (attention_mask.dim() == 2 and attention_mask.shape == inputs_embeds.shape[:2])))
        if not lite:
            return _orig_create_causal_mask(config=config, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                                            past_key_values=past_key_values, **kw)
        return _LiteMask(attention_mask, inputs_embeds.shape[0], inputs_embeds.shape[1], inputs_embeds.device)

    modeling_mpt.create_causal_mask = create_causal_mask


def _frozen_bf16(*linears):
    return all(lin.bias is None and lin.weight.dtype == BF16 and not lin.weight.requires_grad for lin in linears)


def _mpt_block_fused_forward(self, hidden_states, position_bias, attention_mask, layer_past=None, use_cache=False,
                             output_attentions=False, **kwargs):
    attn, ffn = self.attn, self.ffn
    x = hidden_states
    ok = (layer_past is None and not output_attentions and position_bias is not None and position_bias.shape[-1] >= 2
          and x.dim() == 3 and x.dtype == F32 and (x.is_cuda or getattr(self, "_of_allow_cpu", False))
          and x.shape[-1] % 8 == 0 and x.shape[-1] <= 4096 and attn.head_dim in (64, 128) and not attn.clip_qkv
          and not (self.training and (self.dropout_rate > 0.0 or attn.attn_dropout_p > 0.0 or ffn.hidden_dropout > 0.0))
          and _frozen_bf16(attn.Wqkv, attn.out_proj, ffn.up_proj, ffn.down_proj)
          and not self.norm_1.weight.requires_grad and not self.norm_2.weight.requires_grad
          and self.norm_1.weight.dtype == F32 and abs(self.norm_1.eps - 1e-5) < 1e-12 and abs(self.norm_2.eps - 1e-5) < 1e-12
          and isinstance(ffn.act, nn.GELU) and ffn.act.approximate == "none"
          and (getattr(self, "_of_allow_cpu", False)
               or (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == BF16)))
    lite = attention_mask if isinstance(attention_mask, _LiteMask) else None
    # The kernels take the NUMBER of real keys per sequence: correct for unpadded and RIGHT-padded batches only.  Training
    # batches are right-padded by contract (reference train/data.py sets tokenizer.padding_side = "right"); the reference's
    # eval wrapper LEFT-pads its prompts (eval/models/open_flamingo.py:57), so outside training a masked forward keeps HF's own
    # block forward with the real mask unless the caller vouches for right padding (use_fused_frozen_mpt_blocks(...,
    # assume_right_padding=True)).
    masked = (lite.am is not None) if lite is not None else (attention_mask is not None)
    if masked and not (self.training or getattr(self, "_of_assume_right_padding", False)):
        ok = False
    if not ok:
        return self._of_eager_forward(hidden_states, position_bias, lite.full() if lite is not None else attention_mask,
                                      layer_past=layer_past, use_cache=use_cache, output_attentions=output_attentions, **kwargs)
    if lite is not None:
        slopes, lens = lite.slopes(position_bias), lite.lens()
    else:
        from .towers import _alibi_slopes_and_lens
        slopes, lens = _alibi_slopes_and_lens(position_bias, attention_mask, x.shape[1])
    wts = None
    if _DX_PRETRANSPOSED and torch.is_grad_enabled() and x.requires_grad:
        # (the backward of down_proj is the fused DGELU launch on the weight as it lies: no transposed copy of it -- 0.75 GiB at MPT-1B)
        wts = (_transposed(attn, "Wqkv", attn.Wqkv.weight), _transposed(attn, "out_proj", attn.out_proj.weight),
               _transposed(ffn, "up_proj", ffn.up_proj.weight), _transposed_unless_fused(ffn, "down_proj", ffn.down_proj.weight))
    y = _FrozenMptBlockFn.apply(x, self.norm_1.weight, _zero_bias(self.norm_1), self.norm_2.weight, _zero_bias(self.norm_2),
                                attn.Wqkv.weight, attn.out_proj.weight, ffn.up_proj.weight, ffn.down_proj.weight,
                                slopes, lens, attn.n_heads, attn.head_dim, float(attn.softmax_scale), wts, _path.scope_of(self))
    return y, None


def right_padded(attention_mask):
    """True when every row of a (B, L) attention mask is a prefix of ones (host-synchronising debug / test helper)."""
    am = attention_mask.ne(0)
    return bool((am[:, 1:] <= am[:, :-1]).all())


def use_fused_frozen_mpt_blocks(lm, allow_cpu=False, assume_right_padding=False):
    """Route every HF MptBlock of ``lm`` through _FrozenMptBlockFn when its weights are frozen bf16 copies (see
    towers.hold_frozen_linears_in_bf16) and the call is a plain training / scoring forward; the module keeps its class,
    parameters and state-dict keys.  The kernels take the number of real keys per sequence, i.e. batches must be unpadded or
    RIGHT-padded: in training mode that is the data pipeline's contract (reference train/data.py); in eval mode a forward
    WITH an attention mask takes the fused path only with ``assume_right_padding=True`` (the reference's eval wrapper
    left-pads: eval/models/open_flamingo.py:57) and otherwise runs HF's block forward on the real mask.
    ``allow_cpu``: tests only (the host-emulator build of the kernels)."""
    n = 0
    for mod in lm.modules():
        if type(mod).__name__ == "MptBlock":
            if not hasattr(mod, "_of_eager_forward"):
                mod._of_eager_forward = mod.forward
                mod.forward = types.MethodType(_mpt_block_fused_forward, mod)
            mod._of_allow_cpu = bool(allow_cpu)
            mod._of_assume_right_padding = bool(assume_right_padding)
            n += 1
    if n and getattr(lm, "config", None) is not None:
        _install_lite_mask()
        lm.config._of_lite_mask = True
    return n


# ------------------------------------------------------------------------------------------------ frozen GPT-NeoX blocks (OF-4B)
# GPT-NeoX head sizes other than 64 / 128 (RedPajama-INCITE-3B: 80): True = compact heads (OfAttnArgs.head_valid, ABI v11: q, k, v, o and
# their gradients stay (rows, heads x head_size) in HBM, the 128-wide kernels read the missing columns as zeros); False = zero-padded copies
# + of_head_repack passes (rounds 2-5; kept as the other arm of tools/ab_neox_compact_heads.py).  A constant set from that same-box A/B.
_NEOX_COMPACT_HEADS = True


def _neox_pad(head_size):
    """Head size the attention kernels run at: 64 / 128 as they are, anything else <= 128 at the next of the two with COMPACT heads
    (OfAttnArgs.head_valid, ABI v11: RedPajama-INCITE-3B's 80 columns per head side by side in HBM, the kernels read the missing
    columns as zeros -- round 6; before: zero-padded copies of q, k, v, o and their gradients, ~0.33 GB of extra traffic and two
    repack launches per layer and step)."""
    return 64 if head_size <= 64 else 128


class _FrozenNeoXBlockFn(torch.autograd.Function):
    """HF GPTNeoXLayer with frozen weights as ONE autograd node: input_layernorm -> query_key_value
    (+ bias) -> rotary embedding -> causal attention -> dense (+ bias); post_attention_layernorm -> dense_h_to_4h (+ bias) ->
    GELU -> dense_4h_to_h (+ bias); parallel residual (x + attn + mlp, both norms on x) or sequential (x1 = x + attn, mlp on
    norm(x1)).  fp32 residual stream, bf16 GEMM operands / outputs, exact GELU on the bf16 pre-activation -- the eager chain's
    arithmetic under amp_bf16.  Rotary + head padding is one libofhip pass (of_rotary_neox), attention runs on of_attn_fwd/bwd
    (causal, per-sequence key counts), LayerNorms / residual adds as in the MPT block; plain GEMMs on the vendor library with
    the bias in the GEMM (addmm).  Backward = the chain reversed, dX GEMMs only (against pre-transposed weights)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, Wqkv, bqkv, Wd, bd, Wup, bup, Wdown, bdown, cos, sin, kv_len, heads, hs, rot, parallel, wts, scope=None):
        ops = _ops()
        B, L, d = x.shape
        rows = B * L
        dev = x.device
        x2 = x.reshape(rows, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        pad = _neox_pad(hs)
        a = torch.empty(rows, d, dtype=BF16, device=dev)
        st1 = torch.empty(rows, 2, dtype=F32, device=dev)
        ops.ln_fwd(x2, w1, b1, a, st1)
        qkv = torch.addmm(bqkv, a, Wqkv.t())                             # (rows, heads * 3 * hs): [q_h | k_h | v_h] per head
        hw = hs if _NEOX_COMPACT_HEADS else pad                          # columns a head owns in q, k, v, o and their gradients
        qp = torch.empty(3, rows, heads * hw, dtype=BF16, device=dev)    # rotated q, k, v: (rows, heads x hw), heads side by side
        ops.rotary_neox(qkv, cos, sin, qp[0], qp[1], qp[2], L=L, heads=heads, head_size=hs, rot_dims=rot, head_pad=hw)
        op = torch.empty(rows, heads * hw, dtype=BF16, device=dev)
        lse = torch.empty(B, heads, L, dtype=F32, device=dev)
        kw = dict(batch=B, Lq=L, Lk=L, heads=heads, scale=hs ** -0.5, head_dim=pad, head_valid=0 if hw == pad else hs, causal=True,
                  kv_len=kv_len)
        ops.attn_fwd(qp[0], qp[1], qp[2], op, lse, **kw)
        o = op if hw == hs else ops.head_repack(op, torch.empty(rows, d, dtype=BF16, device=dev), heads=heads, src_head_size=hw,
                                                 dst_head_size=hs)
        t = torch.addmm(bd, o, Wd.t())                                   # attention branch, bf16
        m = torch.empty(rows, d, dtype=BF16, device=dev)
        st2 = torch.empty(rows, 2, dtype=F32, device=dev)
        if parallel:
            ops.ln_fwd(x2, w2, b2, m, st2)
            x1 = x2
        else:
            x1 = torch.empty(rows, d, dtype=F32, device=dev)
            ops.ln_fwd_add(x2, t, x1, w2, b2, m, st2)                    # x1 = x + attn branch;  m = norm(x1)
        h = torch.addmm(bup, m, Wup.t())
        u = torch.addmm(bdown, ops.gelu_fwd(h), Wdown.t())
        y = ops.add_bf16(x1, u)
        if parallel:
            ops.add_bf16(y, t, out=y)                                    # x + mlp + attn
        ctx.save_for_backward(x2, st1, qp, op, lse, x1, st2, h, w1, w2, Wqkv, Wd, Wup, Wdown, cos, sin, kv_len)
        ctx.scope = scope
        ctx.kw, ctx.shape, ctx.wts, ctx.cfg = kw, (B, L, d), wts, (heads, hs, rot, hw, parallel)
        return y.view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        ops = _ops()
        x2, st1, qp, op, lse, x1, st2, h, w1, w2, Wqkv, Wd, Wup, Wdown, cos, sin, kv_len = ctx.saved_tensors
        tq, td, tu, tdn = ctx.wts if ctx.wts is not None else (None, None, None, None)
        heads, hs, rot, hw, parallel = ctx.cfg
        B, L, d = ctx.shape
        rows = B * L
        dev = dy.device
        dy2 = dy.reshape(rows, d)
        if dy2.dtype != F32 or not dy2.is_contiguous():
            dy2 = dy2.to(F32).contiguous()
        dyb = _path.bf16_of(ops, dy2, ctx.scope)
        if _MLP_FUSED_DGELU:
            from ..hip import abi
            dh = torch.empty(rows, Wdown.shape[1], dtype=BF16, device=dev)
            ops.gemm(dyb, Wdown, dh, tb=True, epi=abi.EPI_DGELU_DOT, aux=h)        # (dY Wdown) * gelu'(h): one launch (see _MLP_FUSED_DGELU)
        else:
            dact = _mm_dx(dyb, Wdown, tdn)                               # (rows, 4d)
            dh = ops.gelu_bwd(dact, h, out=dact)
            del dact
        dm = _mm_dx(dh, Wup, tu)
        del dh
        dx1 = torch.empty(rows, d, dtype=F32, device=dev)
        if parallel:
            ops.ln_bwd(dm, x2, st2, w2, resid=dy2, dx=dx1)               # dy + norm_2'(dm): the stream's and the MLP branch's share
            dtb = dyb                                                    # the attention branch sees dy itself
        else:
            dtb = torch.empty(rows, d, dtype=BF16, device=dev)
            ops.ln_bwd(dm, x1, st2, w2, resid=dy2, dx=dx1, dx_bf16=dtb)  # dx1 = dy + norm_2'(dm) reaches x AND the attention branch
        do = _mm_dx(dtb, Wd, td)                                         # (rows, d)
        dop = do if hw == hs else ops.head_repack(do, torch.empty(rows, heads * hw, dtype=BF16, device=dev), heads=heads,
                                                   src_head_size=hs, dst_head_size=hw)
        dqp = torch.empty_like(qp)
        delta = torch.empty(B, heads, L, dtype=F32, device=dev)
        ops.attn_bwd(qp[0], qp[1], qp[2], op, lse, dop, dqp[0], dqp[1], dqp[2], delta, **ctx.kw)
        dqkv = torch.empty(rows, heads * 3 * hs, dtype=BF16, device=dev)
        ops.rotary_neox(dqkv, cos, sin, dqp[0], dqp[1], dqp[2], L=L, heads=heads, head_size=hs, rot_dims=rot, head_pad=hw, inverse=True)
        da = _mm_dx(dqkv, Wqkv, tq)
        dxb = torch.empty(rows, d, dtype=BF16, device=dev) if _path.TWINS else None
        ops.ln_bwd(da, x2, st1, w1, resid=dx1, dx=dx1, dx_bf16=dxb)      # in place: dx = dx1 + norm_1'(da); + its bf16 twin
        _path.offer_bf16_twin(dx1, dxb, ctx.scope)
        return (dx1.view(B, L, d),) + (None,) * 21


def _is_exact_gelu(act):
    """nn.GELU() / transformers' GELUActivation in its default form: erf GELU through torch.nn.functional.gelu."""
    if isinstance(act, nn.GELU):
        return act.approximate == "none"
    return type(act).__name__ == "GELUActivation" and getattr(act, "act", None) is nn.functional.gelu


def _neox_layer_fused_forward(self, hidden_states, attention_mask=None, position_ids=None, use_cache=False, layer_past=None,
                              position_embeddings=None, **kwargs):
    at, mlp = self.attention, self.mlp
    x = hidden_states
    lite = attention_mask if isinstance(attention_mask, _LiteMask) else None
    lins = (at.query_key_value, at.dense, mlp.dense_h_to_4h, mlp.dense_4h_to_h)
    ok = (lite is not None and layer_past is None and not kwargs and position_embeddings is not None
          and x.dim() == 3 and x.dtype == F32 and (x.is_cuda or getattr(self, "_of_allow_cpu", False))
          and x.shape[-1] % 8 == 0 and x.shape[-1] <= 4096 and at.head_size <= 128 and at.head_size % 8 == 0
          and at.rotary_ndims % 2 == 0 and float(getattr(at, "scaling", at.head_size ** -0.5)) == at.head_size ** -0.5
          and not (self.training and (self.post_attention_dropout.p > 0.0 or self.post_mlp_dropout.p > 0.0 or at.attention_dropout > 0.0))
          and all(lin.bias is not None and lin.weight.dtype == BF16 and lin.bias.dtype == BF16 and not lin.weight.requires_grad for lin in lins)
          and all(n.weight.dtype == F32 and n.bias is not None and not n.weight.requires_grad and abs(n.eps - 1e-5) < 1e-12
                  for n in (self.input_layernorm, self.post_attention_layernorm))
          and _is_exact_gelu(mlp.act)
          and (getattr(self, "_of_allow_cpu", False)
               or (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == BF16)))
    if ok and lite.am is not None and not (self.training or getattr(self, "_of_assume_right_padding", False)):
        ok = False          # key counts = right padding only (see _mpt_block_fused_forward)
    if ok:
        cos, sin = position_embeddings
        # of_rotary_neox takes ONE [L][rot] table (position = row % L): the default position_ids = arange(L) of a cache-less
        # forward, which HF hands down as a (1, L, rot) table.  A caller-supplied per-row position_ids makes it (B, L, rot) with
        # differing rows -- that forward keeps HF's own rotary embedding (row 0's positions were applied to the batch).
        ok = cos.dim() == 3 and cos.shape[0] == 1 and cos.shape[1] == x.shape[1] and cos.shape[2] == at.rotary_ndims
    if not ok:
        mask = lite.hf_mask() if lite is not None else attention_mask
        return self._of_eager_forward(hidden_states, attention_mask=mask, position_ids=position_ids, use_cache=use_cache,
                                      layer_past=layer_past, position_embeddings=position_embeddings, **kwargs)
    cs = lite.__dict__.get("_cos_sin")
    if cs is None or cs[0] is not cos:       # one fp32 [L][rot] table per LM forward (HF repeats it per batch row)
        cs = lite.__dict__["_cos_sin"] = (cos, cos[0].float().contiguous(), sin[0].float().contiguous())
    wts = None
    if _DX_PRETRANSPOSED and torch.is_grad_enabled() and x.requires_grad:
        wts = tuple(_transposed(mod, name, lin.weight) for mod, name, lin in
                    ((at, "query_key_value", at.query_key_value), (at, "dense", at.dense),
                     (mlp, "dense_h_to_4h", mlp.dense_h_to_4h))) + (_transposed_unless_fused(mlp, "dense_4h_to_h", mlp.dense_4h_to_h.weight),)
    n1, n2 = self.input_layernorm, self.post_attention_layernorm
    return _FrozenNeoXBlockFn.apply(x, n1.weight, n1.bias, n2.weight, n2.bias, at.query_key_value.weight, at.query_key_value.bias,
                                    at.dense.weight, at.dense.bias, mlp.dense_h_to_4h.weight, mlp.dense_h_to_4h.bias,
                                    mlp.dense_4h_to_h.weight, mlp.dense_4h_to_h.bias, cs[1], cs[2], lite.lens(),
                                    at.config.num_attention_heads, at.head_size, at.rotary_ndims,
                                    bool(self.use_parallel_residual), wts, _path.scope_of(self))


def use_fused_frozen_neox_blocks(lm, allow_cpu=False, assume_right_padding=False):
    """Route every HF GPTNeoXLayer of ``lm`` (OF-4B: RedPajama-INCITE-3B) through _FrozenNeoXBlockFn when its weights are
    frozen bf16 copies and the call is a plain training / scoring forward; padding rules as use_fused_frozen_mpt_blocks."""
    n = 0
    for mod in lm.modules():
        if type(mod).__name__ == "GPTNeoXLayer":
            if not hasattr(mod, "_of_eager_forward"):
                mod._of_eager_forward = mod.forward
                mod.forward = types.MethodType(_neox_layer_fused_forward, mod)
            mod._of_allow_cpu = bool(allow_cpu)
            mod._of_assume_right_padding = bool(assume_right_padding)
            n += 1
    if n and getattr(lm, "config", None) is not None:
        _install_lite_mask_neox()
        lm.config._of_lite_mask = True
    return n


# ------------------------------------------------------------------------------------------------ CLIP vision tower
# fc2 of the CLIP MLP (K = 4096) with the rows split into whole 256-row tiles + the ragged rest: 64 images x 257 tokens = 16448 rows are
# 64.25 row tiles, every tiling of the vendor library leaves a ragged last round (tuned: 114.7 us); the first 16384 rows alone are whole
# rounds of the 256 CUs (86.3 us) and the last 64 rows a 15.5-us launch (tools/probes/vit_rows_probe.py, profiles/r06zq_*).  The same
# split does not pay for the tower's K = 1024 GEMMs (9-14 us gained, 13-14 us for the rest rows).  Rows are independent: same bits.
_FC2_WHOLE_TILES = True


def _addmm_whole_tiles(bias, a, wt):
    """a @ wt + bias, as two launches when the rows are a ragged number of 256-row tiles (see _FC2_WHOLE_TILES)"""
    rows = a.shape[0]
    main = rows // 256 * 256
    if not _FC2_WHOLE_TILES or main == rows or main < 4096:
        return torch.addmm(bias, a, wt)
    out = torch.empty(rows, wt.shape[1], dtype=a.dtype, device=a.device)
    torch.addmm(bias, a[:main], wt, out=out[:main])
    torch.addmm(bias, a[main:], wt, out=out[main:])
    return out


def _fused_qkv(attn):
    """(3D, D) weight and (3D,) bias of the three projections of one CLIPAttention as ONE GEMM operand (the three eager
    GEMMs read the same LayerNorm output); cached on the module, rebuilt when a source tensor is replaced or modified."""
    srcs = (attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight, attn.q_proj.bias, attn.k_proj.bias, attn.v_proj.bias)
    key = tuple((t.data_ptr(), t._version) for t in srcs)
    hit = attn.__dict__.get("_of_qkv")
    if hit is None or hit[0] != key:
        hit = attn.__dict__["_of_qkv"] = (key, torch.cat(srcs[:3], 0).contiguous(), torch.cat(srcs[3:], 0).contiguous())
    return hit[1], hit[2]


def _clip_layer_ok(layer):
    at, mlp = layer.self_attn, layer.mlp
    lins = (at.q_proj, at.k_proj, at.v_proj, at.out_proj, mlp.fc1, mlp.fc2)
    return (all(lin.bias is not None and lin.weight.dtype == BF16 and lin.bias.dtype == BF16
                and not lin.weight.requires_grad for lin in lins)
            and at.head_dim in (64, 128) and type(mlp.activation_fn).__name__ == "QuickGELUActivation"
            and all(n.weight.dtype == F32 and n.bias is not None and abs(n.eps - 1e-5) < 1e-12
                    for n in (layer.layer_norm1, layer.layer_norm2)))


def clip_encoder_fused(encoder, x, attention="libofhip"):
    """All CLIPEncoderLayers of a frozen tower, forward only (flamingo.py:194-195 runs the tower under no_grad), on the fp32
    stream x (N, S, D).  Returns (stream, pending): the stream BEFORE the last MLP branch is added and that bf16 branch
    output -- the caller folds the last add into whatever LayerNorm comes next (of_layernorm_fwd_add).
    Per layer: LN (or add + LN) -> ONE q|k|v GEMM with bias -> attention -> out_proj -> add + LN -> fc1 -> quick-GELU -> fc2;
    every residual add rides in a LayerNorm pass, no q/k/v/context transposes."""
    ops = _ops()
    N, S, D = x.shape
    rows = N * S
    dev = x.device
    x2 = x.reshape(rows, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    a = torch.empty(rows, D, dtype=BF16, device=dev)
    xs = None                       # our own stream buffer (x is never written)
    pending = None
    for layer in encoder.layers:
        at, mlp = layer.self_attn, layer.mlp
        n1, n2 = layer.layer_norm1, layer.layer_norm2
        if pending is None:
            ops.ln_fwd(x2, n1.weight, n1.bias, a, None)
        else:
            ops.ln_fwd_add(xs, pending, xs, n1.weight, n1.bias, a, None)
        w, b = _fused_qkv(at)
        qkv = torch.addmm(b, a, w.t())                                   # (rows, 3D)
        H, dh = at.num_heads, at.head_dim
        if attention == "libofhip":
            o = torch.empty(rows, D, dtype=BF16, device=dev)
            lse = torch.empty(N, H, S, dtype=F32, device=dev)
            ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, batch=N, Lq=S, Lk=S, heads=H,
                         scale=float(at.scale), head_dim=dh)
        else:
            q, k, v = (qkv[:, i * D:(i + 1) * D].view(N, S, H, dh).transpose(1, 2) for i in range(3))
            o = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=float(at.scale))
            o = o.transpose(1, 2).reshape(rows, D)
        t = torch.addmm(at.out_proj.bias, o, at.out_proj.weight.t())
        src = x2 if xs is None else xs
        if xs is None:
            xs = torch.empty(rows, D, dtype=F32, device=dev)
        ops.ln_fwd_add(src, t, xs, n2.weight, n2.bias, a, None)          # xs = stream + attention branch; a = LN2(xs)
        h = torch.addmm(mlp.fc1.bias, a, mlp.fc1.weight.t())
        pending = _addmm_whole_tiles(mlp.fc2.bias, ops.quick_gelu(h), mlp.fc2.weight.t())
    return xs, pending


def clip_tower_tokens_fused(vm, pixel_values, attention="libofhip"):
    """post_layernorm(encoder(pre_layrnorm(embeddings(pixels)))) for a frozen HF CLIP vision tower whose Linear weights are
    held in bf16, or None when the fused form does not apply (the caller then runs the modules)."""
    layers = vm.encoder.layers
    x = pixel_values
    if (torch.is_grad_enabled() and any(p.requires_grad for p in vm.parameters())) or not (x.is_cuda or getattr(vm, "_of_allow_cpu", False)):
        return None
    if not getattr(vm, "_of_allow_cpu", False) and not (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == BF16):
        return None
    if len(layers) == 0 or not all(_clip_layer_ok(layer) for layer in layers):
        return None
    post = vm.post_layernorm
    D = post.weight.shape[0]
    if post.weight.dtype != F32 or post.bias is None or abs(post.eps - 1e-5) > 1e-12 or D % 8 or D > 4096:
        return None
    h = vm.pre_layrnorm(vm.embeddings(pixel_values))
    if h.dtype != F32:
        h = h.float()
    N, S, _ = h.shape
    xs, pending = clip_encoder_fused(vm.encoder, h, attention=attention)
    y = torch.empty(N * S, D, dtype=F32, device=h.device)
    _ops().ln_fwd_add(xs, pending, xs, post.weight, post.bias, y, None)   # last residual add + post_layernorm, fp32 out
    return y.view(N, S, D)
