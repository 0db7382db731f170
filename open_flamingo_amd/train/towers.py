"""Random-init frozen towers with the OpenFlamingo model-family dimensions, built from HF configs.

There is no network in the build/benchmark environment, so pretrained CLIP / MPT / RedPajama weights cannot be
fetched; throughput only depends on the architecture, so the benchmark instantiates the same architectures with
random weights (stated as ``"data": "synthetic"`` in bench.py's output).  These modules are the FROZEN parts of the
step: they run as stock PyTorch-ROCm modules -- they are not part of the hot path this repo
re-implements.

LM dimensions (public model cards; consistent with the 3B/4B/9B totals in the reference README.md:106-110):
  OF-3B: MPT-1B  d=2048, 24 layers, 16 heads, vocab 50432, xattn every 1
  OF-4B: RedPajama-INCITE-3B (GPT-NeoX) d=2560, 32 layers, 32 heads, vocab 50432, xattn every 2
  OF-9B: MPT-7B  d=4096, 32 layers, 32 heads, vocab 50432, xattn every 4
Vision: CLIP ViT-L/14 @224 (width 1024, 24 layers, 16 heads, 256 patch tokens).
"""
import torch
from torch import nn

FAMILY = {
    "OF-3B": dict(lm="mpt", d=2048, layers=24, heads=16, vocab=50432, every=1),
    "OF-4B": dict(lm="neox", d=2560, layers=32, heads=32, vocab=50432, every=2),
    "OF-9B": dict(lm="mpt", d=4096, layers=32, heads=32, vocab=50432, every=4),
    # tiny variant for tests (same code paths, seconds on CPU for the frozen parts)
    "OF-tiny": dict(lm="mpt", d=256, layers=4, heads=4, vocab=1000, every=2),
}


class PatchEmbedAsGemm(nn.Module):
    """Drop-in for the ViT's non-overlapping ``Conv2d(3, width, kernel=patch, stride=patch, bias=False)`` patch
    embedding: the same weight tensor, applied as one (N*patches, 3*patch*patch) x (3*patch*patch, width) GEMM.
    MIOpen has no tuned bf16 kernel for this 14x14/stride-14 convolution on gfx950 and falls back to
    ``naive_conv_ab_nonpacked_fwd_nchw`` (9.7 ms per call at 64 images -- profiles/r01_*_kernel_stats); the GEMM
    form takes ~0.1 ms and is arithmetically the same contraction."""

    def __init__(self, conv: nn.Conv2d):
        super().__init__()
        assert conv.bias is None and conv.kernel_size == conv.stride and conv.padding == (0, 0)
        self.weight = conv.weight                     # (width, 3, p, p): same Parameter object -> same state_dict key
        self.patch = conv.kernel_size[0]

    def forward(self, x):
        n, c, hh, ww = x.shape
        p = self.patch
        gh, gw = hh // p, ww // p
        cols = x.reshape(n, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(n * gh * gw, c * p * p)
        y = torch.nn.functional.linear(cols, self.weight.reshape(self.weight.shape[0], -1))
        return y.reshape(n, gh, gw, -1).permute(0, 3, 1, 2)     # conv layout (N, width, gh, gw), as the caller expects


class ClipVisualStandIn(nn.Module):
    """``visual(x) -> (pooled, tokens)`` like open_clip's VisionTransformer with output_tokens=True
    (reference factory.py:48, flamingo.py:195): tokens = ln_post(transformer(x))[:, 1:]."""

    def __init__(self, width=1024, layers=24, heads=16, patch=14, image=224, mlp=None, patch_embed="gemm", fused=False):
        super().__init__()
        from transformers import CLIPVisionConfig, CLIPVisionModel
        cfg = CLIPVisionConfig(hidden_size=width, intermediate_size=mlp or 4 * width, num_hidden_layers=layers,
                               num_attention_heads=heads, patch_size=patch, image_size=image)
        self.model = CLIPVisionModel(cfg)
        if patch_embed == "gemm":
            emb = getattr(self.model, "vision_model", self.model).embeddings
            emb.patch_embedding = PatchEmbedAsGemm(emb.patch_embedding)
        self.width = width
        self.fused = fused          # False | "libofhip" | "sdpa": train/frozen_blocks.py::clip_tower_tokens_fused, attention kernel

    def forward(self, x):
        m = self.model
        vm = getattr(m, "vision_model", m)       # transformers < 5 nests the tower under .vision_model
        if self.fused:
            from .frozen_blocks import clip_tower_tokens_fused
            h = clip_tower_tokens_fused(vm, x, attention=self.fused)
            if h is not None:
                return h[:, 0], h[:, 1:]
        h = vm.post_layernorm(m(pixel_values=x).last_hidden_state)
        return h[:, 0], h[:, 1:]


class VisionStandIn(nn.Module):
    """Object with a ``.visual`` attribute, like an open_clip CLIP model."""

    def __init__(self, **kw):
        super().__init__()
        self.visual = ClipVisualStandIn(**kw)


_BIAS_CACHE = {}      # one entry per kind; holds the KEY TENSORS themselves and matches by identity (an address / version
                      # pair can recur across steps once the caching allocator recycles the block: stale padding lengths)


def _cache_get(kind, position_bias, attention_mask, extra):
    hit = _BIAS_CACHE.get(kind)
    if hit is not None and hit[0] is position_bias and hit[1] is attention_mask and hit[2] == extra \
            and hit[3] == (position_bias._version, None if attention_mask is None else attention_mask._version):
        return hit[4]
    return None


def _cache_put(kind, position_bias, attention_mask, extra, val):
    _BIAS_CACHE[kind] = (position_bias, attention_mask, extra,
                         (position_bias._version, None if attention_mask is None else attention_mask._version), val)
    return val


def _alibi_causal_bias(position_bias, attention_mask, q_len, dtype):
    """Additive (.., H, L, L) attention bias = ALiBi in its relative form slope_h * (j - i) (softmax is shift invariant
    per row; the relative form keeps the values small enough to survive bf16) with the masked positions of HF's boolean
    causal/padding mask set to a large negative number (HF fills finfo.min, not -inf: fully masked rows stay finite).
    Built once per LM forward (all blocks receive the same mask / alibi tensors)."""
    hit = _cache_get("bias", position_bias, attention_mask, (q_len, dtype))
    if hit is not None:
        return hit
    pb = position_bias[:, 0, -q_len:].float()                       # (H, L): slope_h * (j - (S - 1))
    bias = (pb[:, None, :] - pb[:, :, None]).unsqueeze(0)           # (1, H, L, L): slope_h * (j - i)
    if attention_mask is not None:
        bias = bias.masked_fill(attention_mask[..., -q_len:, -q_len:], -30000.0)
    return _cache_put("bias", position_bias, attention_mask, (q_len, dtype), bias.to(dtype).contiguous())


class _CausalAlibiAttention(torch.autograd.Function):
    """libofhip windowed flash attention (csrc/attention.hip) as causal self-attention with ALiBi over the fused
    Wqkv output: q/k/v are strided views of one (B*L, 3d) buffer, and the backward writes dq|dk|dv into one buffer of
    the same layout -- no chunk / transpose / cat copies.  Right padding only (kv_len = real keys per sequence)."""

    @staticmethod
    def forward(ctx, qkv, slopes, kv_len, heads, head_dim, scale):
        from ..hip.ops import Ops
        ops = Ops.default()
        b, l, three_d = qkv.shape
        d = heads * head_dim
        x = qkv.reshape(b * l, three_d)
        o = torch.empty(b * l, d, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(b, heads, l, dtype=torch.float32, device=qkv.device)
        kw = dict(batch=b, Lq=l, Lk=l, heads=heads, scale=scale, head_dim=head_dim, causal=True, alibi_slopes=slopes,
                  kv_len=kv_len)
        ops.attn_fwd(x[:, :d], x[:, d:2 * d], x[:, 2 * d:], o, lse, **kw)
        ctx.save_for_backward(x, o, lse, slopes, kv_len)
        ctx.kw, ctx.shape = kw, (b, l, d)
        return o.view(b, l, d)

    @staticmethod
    def backward(ctx, do):
        from ..hip.ops import Ops
        ops = Ops.default()
        x, o, lse, slopes, kv_len = ctx.saved_tensors
        b, l, d = ctx.shape
        do2 = do.reshape(b * l, d)
        if not do2.is_contiguous() or do2.dtype != x.dtype:
            do2 = do2.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        delta = torch.empty(b, ctx.kw["heads"], l, dtype=torch.float32, device=x.device)
        ops.attn_bwd(x[:, :d], x[:, d:2 * d], x[:, 2 * d:], o, lse, do2, dx[:, :d], dx[:, d:2 * d], dx[:, 2 * d:], delta,
                     **ctx.kw)
        return dx.view(b, l, 3 * d), None, None, None, None, None


def _alibi_slopes_and_lens(position_bias, attention_mask, q_len):
    """(heads,) fp32 ALiBi slopes and (B,) int32 real-key counts, derived on the device (no host sync) from the
    tensors HF hands every block; cached for the duration of one LM forward."""
    hit = _cache_get("slopes", position_bias, attention_mask, q_len)
    if hit is not None:
        return hit
    pb = position_bias[:, 0, :].float()
    slopes = (pb[:, -1] - pb[:, -2]).contiguous()                 # slope_h * (j - (S-1)): consecutive keys differ by slope_h
    lens = None
    if attention_mask is not None:                                 # bool (B,1,L,L), True = masked; last query row sees every real key
        lens = (~attention_mask[:, 0, -1, -q_len:]).sum(-1).to(torch.int32).contiguous()
    return _cache_put("slopes", position_bias, attention_mask, q_len, (slopes, lens))


def _mpt_attention_fused_forward(self, hidden_states, position_bias, past_key_values=None, attention_mask=None, **kwargs):
    """MptAttention.forward with the score/softmax/AV chain as ONE fused attention call (torch SDPA) instead of HF's eager
    chain (matmul, scale, + alibi, masked_fill, fp32 softmax, cast, matmul -- each materialising a (B, H, L, L) tensor).
    Same arithmetic up to the bf16 rounding of the bias.  Decode with a KV cache, attention dropout and clip_qkv keep the
    original eager path."""
    if (past_key_values is not None or position_bias is None or self.clip_qkv
            or (self.training and self.attn_dropout_p > 0.0)):
        return self._of_eager_forward(hidden_states, position_bias, past_key_values=past_key_values,
                                      attention_mask=attention_mask, **kwargs)
    b, l = hidden_states.shape[:2]
    qkv = self.Wqkv(hidden_states)
    # the libofhip kernel takes key COUNTS (unpadded / right-padded batches: the training contract, reference train/data.py);
    # outside training a masked forward may be left-padded (reference eval wrapper) and takes the SDPA form with the real mask
    if (getattr(self, "_of_attention_kernel", "sdpa") == "libofhip" and qkv.is_cuda and qkv.dtype == torch.bfloat16
            and self.head_dim in (64, 128) and position_bias.shape[-1] >= 2
            and (attention_mask is None or self.training or getattr(self, "_of_assume_right_padding", False))):
        slopes, lens = _alibi_slopes_and_lens(position_bias, attention_mask, l)
        ctx = _CausalAlibiAttention.apply(qkv.contiguous(), slopes, lens, self.n_heads, self.head_dim,
                                          float(self.softmax_scale))
        return self.out_proj(ctx), None
    q, k, v = qkv.chunk(3, dim=2)
    q, k, v = (t.reshape(b, l, self.n_heads, self.head_dim).transpose(1, 2) for t in (q, k, v))
    bias = _alibi_causal_bias(position_bias, attention_mask, l, q.dtype)
    ctx = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=bias, scale=self.softmax_scale)
    return self.out_proj(ctx.transpose(1, 2).reshape(b, l, -1)), None


class _FrozenLayerNorm(torch.autograd.Function):
    """A frozen tower's LayerNorm in front of a Linear, under amp_bf16: eager PyTorch normalises in fp32, writes fp32, and
    autocast then re-reads it to cast to bf16 (two passes forward; cast-back + LayerNorm backward going back).  libofhip's
    kernel writes the bf16 operand directly (one pass) and its backward takes the bf16 gradient as it is; no parameter
    gradients (the tower is frozen).  Same statistics (fp32), same single bf16 rounding of the output."""

    @staticmethod
    def forward(ctx, x, w, b):
        from ..hip.ops import Ops
        ops = Ops.default()
        dim = x.shape[-1]
        x2 = x.reshape(-1, dim)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y = torch.empty(x2.shape, dtype=torch.bfloat16, device=x.device)
        stats = torch.empty(x2.shape[0], 2, dtype=torch.float32, device=x.device)
        ops.ln_fwd(x2, w, b, y, stats)
        ctx.save_for_backward(x2, stats, w)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        from ..hip.ops import Ops
        ops = Ops.default()
        x2, stats, w = ctx.saved_tensors
        dy2 = dy.reshape(x2.shape)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = torch.empty_like(x2)
        ops.ln_bwd(dy2, x2, stats, w, dx=dx)
        return dx.view(ctx.shape), None, None


def _layernorm_libofhip_forward(self, x):
    w = self.weight
    if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and w is not None and not w.requires_grad
            and w.dtype == torch.float32 and abs(self.eps - 1e-5) < 1e-12 and len(self.normalized_shape) == 1
            and x.shape[-1] % 8 == 0 and x.shape[-1] <= 4096
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        b = self.bias
        if b is None:          # HF MPT drops the LayerNorm bias
            b = self.__dict__.get("_of_zero_bias")
            if b is None or b.device != w.device:
                b = self.__dict__["_of_zero_bias"] = torch.zeros_like(w)
        return _FrozenLayerNorm.apply(x, w, b)
    return self._of_eager_forward(x)


_LN_IN_FRONT_OF_LINEAR = ("norm_1", "norm_2", "norm_f",                                  # HF MPT
                          "layer_norm1", "layer_norm2",                                   # HF CLIP encoder layers
                          "input_layernorm", "post_attention_layernorm", "final_layer_norm")  # HF GPT-NeoX


def use_libofhip_layernorm(module):
    """Route the frozen towers' LayerNorms whose only consumers are Linear layers through libofhip (see
    _FrozenLayerNorm).  Modules keep their class, parameters and state-dict keys; anything the kernel does not cover
    (trainable weights, no autocast, CPU, other eps) falls through to the original forward.  CLIP's pre/post LayerNorms
    feed the fp32 residual stream / the Perceiver and are left alone."""
    import types
    n = 0
    for name, mod in module.named_modules():
        if isinstance(mod, nn.LayerNorm) and name.rsplit(".", 1)[-1] in _LN_IN_FRONT_OF_LINEAR \
                and not hasattr(mod, "_of_eager_forward"):
            mod._of_eager_forward = mod.forward
            mod.forward = types.MethodType(_layernorm_libofhip_forward, mod)
            n += 1
    return n


def _quick_gelu_libofhip_forward(self, x):
    if x.is_cuda and x.dtype == torch.bfloat16 and not (torch.is_grad_enabled() and x.requires_grad):
        from ..hip.ops import Ops
        return Ops.default().quick_gelu(x if x.is_contiguous() else x.contiguous())
    return self._of_eager_forward(x)


def use_libofhip_quick_gelu(module):
    """CLIP's MLP activation x * sigmoid(1.702 x) as one pass (libofhip) instead of HF's three element-wise kernels; only
    where no gradient is needed (the vision tower runs under no_grad), otherwise the original forward."""
    import types
    n = 0
    for mod in module.modules():
        if type(mod).__name__ == "QuickGELUActivation" and not hasattr(mod, "_of_eager_forward"):
            mod._of_eager_forward = mod.forward
            mod.forward = types.MethodType(_quick_gelu_libofhip_forward, mod)
            n += 1
    return n


class _FusedCausalLMLoss(torch.autograd.Function):
    """mean over (label != ignore) of  log-sum-exp(logits) - logits[label]  straight from the LM head's bf16 logits
    (libofhip of_ce_fwd / of_ce_bwd): what transformers' ForCausalLMLoss computes via logits.float() -> log_softmax ->
    nll_loss, in two passes over the logits instead of ~12 GB of eager traffic at OF-3B cfg-2.  Same fp32 arithmetic; the
    gradient is written in the logits' dtype directly (autograd casts the eager fp32 gradient back to bf16 the same way)."""

    @staticmethod
    def forward(ctx, logits2d, labels1d, ignore_index):
        from ..hip.ops import Ops
        ops = Ops.default()
        rows = logits2d.shape[0]
        lse = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
        loss_rows = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
        ops.ce_fwd(logits2d, labels1d, lse, loss_rows, ignore_index)
        n_valid = (labels1d != ignore_index).sum().to(torch.float32)
        ctx.save_for_backward(logits2d, labels1d, lse, n_valid)
        ctx.ignore_index = ignore_index
        return loss_rows.sum() / n_valid             # nan when every label is ignored, like F.cross_entropy

    @staticmethod
    def backward(ctx, g):
        from ..hip.ops import Ops
        ops = Ops.default()
        logits2d, labels1d, lse, n_valid = ctx.saved_tensors
        gscale = (g.to(torch.float32) / n_valid).reshape(1).contiguous()
        d = torch.empty_like(logits2d)
        ops.ce_bwd(logits2d, labels1d, lse, gscale, d, ctx.ignore_index)
        return d, None, None


def _causal_lm_loss_libofhip(logits, labels, vocab_size, num_items_in_batch=None, ignore_index=-100, shift_labels=None,
                             **kwargs):
    """Drop-in for transformers.loss.loss_utils.ForCausalLMLoss (the `loss_function` of MptForCausalLM / GPTNeoXForCausalLM)."""
    if (not logits.is_cuda or logits.dtype not in (torch.bfloat16, torch.float32) or num_items_in_batch is not None
            or kwargs or logits.shape[-1] != vocab_size):
        from transformers.loss.loss_utils import ForCausalLMLoss
        return ForCausalLMLoss(logits, labels, vocab_size, num_items_in_batch=num_items_in_batch, ignore_index=ignore_index,
                               shift_labels=shift_labels, **kwargs)
    if shift_labels is None:        # tokens < n predict n
        shift_labels = torch.nn.functional.pad(labels, (0, 1), value=ignore_index)[..., 1:]
    l2 = logits.reshape(-1, vocab_size)
    if l2.stride(1) != 1:
        l2 = l2.contiguous()
    return _FusedCausalLMLoss.apply(l2, shift_labels.reshape(-1).to(torch.int64).contiguous(), int(ignore_index))


def use_libofhip_lm_loss(lm):
    """Route the HF causal-LM loss (``self.loss_function`` in MptForCausalLM / GPTNeoXForCausalLM.forward) through the fused
    libofhip cross-entropy.  Unsupported calls (CPU tensors, num_items_in_batch, extra kwargs) fall through to HF's own."""
    lm.loss_function = _causal_lm_loss_libofhip
    return lm


def use_fused_attention_in_mpt(lm, kernel="sdpa"):
    """kernel = "sdpa": torch's fused attention with an additive bias (any mask HF builds).  kernel = "libofhip": this
    repository's windowed flash-attention kernel as causal + ALiBi self-attention (bf16 on an AMD GPU; batches must be
    unpadded or RIGHT-padded -- training batches are, train/data.py pads on the right; left-padded generation prompts
    use a KV cache and therefore the eager path anyway)."""
    import types
    assert kernel in ("sdpa", "libofhip")
    n = 0
    for mod in lm.modules():
        if type(mod).__name__ == "MptAttention":
            if not hasattr(mod, "_of_eager_forward"):
                mod._of_eager_forward = mod.forward
                mod.forward = types.MethodType(_mpt_attention_fused_forward, mod)
            mod._of_attention_kernel = kernel
            n += 1
    return n


def build_lang_encoder(family: str, extra_tokens: int = 3, max_seq_len: int = 2048, fused_attention=True):
    f = FAMILY[family]
    vocab = f["vocab"] + extra_tokens      # <|endofchunk|>, <image>, <PAD> appended by factory.py:57-63
    if f["lm"] == "mpt":
        from transformers import MptConfig, MptForCausalLM
        cfg = MptConfig(d_model=f["d"], n_heads=f["heads"], n_layers=f["layers"], vocab_size=vocab,
                        max_seq_len=max_seq_len)
        lm = MptForCausalLM(cfg)
        if fused_attention:
            use_fused_attention_in_mpt(lm, kernel=fused_attention if isinstance(fused_attention, str) else "sdpa")
        return lm, "transformer.blocks"
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    # RedPajama-INCITE-Base-3B-v1's published config (no network here to re-read it): rotary over the whole head (rotary_pct
    # 1.0), sequential residual (use_parallel_residual false), head size 2560 / 32 = 80, biases on every Linear
    cfg = GPTNeoXConfig(hidden_size=f["d"], num_hidden_layers=f["layers"], num_attention_heads=f["heads"],
                        intermediate_size=4 * f["d"], vocab_size=vocab, max_position_embeddings=max_seq_len,
                        rotary_pct=1.0, use_parallel_residual=False)
    return GPTNeoXForCausalLM(cfg), "gpt_neox.layers"


def hold_frozen_linears_in_bf16(model):
    """amp_bf16 (reference train_utils.py:34-41) re-casts every frozen fp32 nn.Linear weight to bf16 on every forward
    (autocast only caches casts of leaves that require grad); the frozen towers never change, so keep those weights in
    bf16 once.  The matmuls see bit-identical operands; LayerNorm/embedding parameters and the residual stream stay
    fp32.  The tied MPT wte/lm_head matrix stays fp32 (two of its rows are trainable, factory.py:109-113)."""
    tied = {id(p) for p in model.lang_encoder.get_input_embeddings().parameters()}
    n = 0
    for mod in list(model.vision_encoder.modules()) + list(model.lang_encoder.modules()):
        if isinstance(mod, nn.Linear) and not mod.weight.requires_grad and id(mod.weight) not in tied:
            mod.weight.data = mod.weight.data.to(torch.bfloat16)
            if mod.bias is not None:
                mod.bias.data = mod.bias.data.to(torch.bfloat16)
            n += 1
    return n


def use_tuned_vendor_gemms(table=None):
    """The frozen towers' plain GEMMs stay on the vendor libraries (hipBLASLt / rocBLAS through torch.mm); their default
    kernel heuristics are not the fastest choice for every shape of the step.  This loads a PyTorch TunableOp table --
    per-shape kernel selections measured once on an MI355X with tools/gpu_tunableop.sh and committed under train/tuned/ --
    with tuning itself switched off (no measurement at run time, shapes outside the table keep the default kernel).  The
    table is only honoured when its validator lines (PyTorch / HIP / hipBLASLt / rocBLAS versions, gfx950) match the
    running libraries.  Returns the number of table entries, 0 when TunableOp is unavailable."""
    import os
    try:
        import torch.cuda.tunable as tun
    except ImportError:
        return 0
    if table is None:
        table = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "tunableop_gfx950_of3b_cfg2.csv")
    if not os.path.exists(table) or not torch.cuda.is_available():
        return 0
    tun.enable(True)
    tun.tuning_enable(False)
    ok = tun.read_file(table)
    if not ok:
        tun.enable(False)
        return 0
    return len(tun.get_results())


def build_flamingo(family: str = "OF-3B", device="cuda", seed: int = 0, gates: float = 0.5, vision_kw=None,
                   freeze_lm_embeddings: bool = False, verbose: bool = False, frozen_bf16: bool = False,
                   fused_lm_attention=True, tower_layernorm="eager", lm_loss="hf", fused_lm_blocks=False, fused_vision=False, perceiver_depth=None):
    """Random-init Flamingo of the given family, assembled through the factory path, gates set to ``gates``
    (their init value 0 makes the hot path an exact no-op with zero weight gradients -- SURVEY.md section 7)."""
    from ..src.factory import assemble_flamingo
    f = FAMILY[family]
    torch.manual_seed(seed)
    dev = torch.device(device)
    with torch.device(dev):
        vkw = vision_kw or (dict(width=64, layers=2, heads=2, patch=14, image=224) if family == "OF-tiny" else {})
        vision = VisionStandIn(**vkw)
        lm, attr = build_lang_encoder(family, fused_attention=fused_lm_attention)
        vis_dim = vision.visual.width
        media_id, eoc_id = f["vocab"] + 1, f["vocab"]
        model = assemble_flamingo(vision, lm, eoc_id, media_id, vis_dim=vis_dim, cross_attn_every_n_layers=f["every"],
                                  decoder_layers_attr_name=attr, freeze_lm_embeddings=freeze_lm_embeddings,
                                  verbose=verbose)
        if perceiver_depth is not None:    # tests: a shallower Perceiver than the reference's fixed depth 6 (flamingo.py:74)
            from ..src.helpers import PerceiverResampler
            model.perceiver = PerceiverResampler(dim=vis_dim, depth=perceiver_depth)
            model.perceiver.requires_grad_(True)
    if frozen_bf16:
        hold_frozen_linears_in_bf16(model)
    if tower_layernorm == "libofhip":      # the element-wise pieces of the frozen towers on libofhip
        use_libofhip_layernorm(model.vision_encoder)
        use_libofhip_layernorm(model.lang_encoder)
        use_libofhip_quick_gelu(model.vision_encoder)
    if lm_loss == "libofhip":
        use_libofhip_lm_loss(model.lang_encoder)
    if fused_vision:                       # the CLIP tower's encoder as one fused forward (train/frozen_blocks.py)
        model.vision_encoder.fused = fused_vision
    if fused_lm_blocks:                    # whole frozen MPT blocks as one autograd node each (train/frozen_blocks.py)
        from .frozen_blocks import use_fused_frozen_mpt_blocks, use_fused_frozen_neox_blocks
        use_fused_frozen_mpt_blocks(model.lang_encoder)
        use_fused_frozen_neox_blocks(model.lang_encoder)     # OF-4B's GPT-NeoX layers (no-op for the MPT families)
    with torch.no_grad():
        for blk in model.lang_encoder.gated_cross_attn_layers:
            if blk is not None:
                blk.attn_gate.fill_(gates)
                blk.ff_gate.fill_(gates)
    return model, dict(f, media_token_id=media_id, eoc_token_id=eoc_id, pad_token_id=f["vocab"] + 2)
