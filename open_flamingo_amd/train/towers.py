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


class ClipVisualStandIn(nn.Module):
    """``visual(x) -> (pooled, tokens)`` like open_clip's VisionTransformer with output_tokens=True
    (reference factory.py:48, flamingo.py:195): tokens = ln_post(transformer(x))[:, 1:]."""

    def __init__(self, width=1024, layers=24, heads=16, patch=14, image=224, mlp=None):
        super().__init__()
        from transformers import CLIPVisionConfig, CLIPVisionModel
        cfg = CLIPVisionConfig(hidden_size=width, intermediate_size=mlp or 4 * width, num_hidden_layers=layers,
                               num_attention_heads=heads, patch_size=patch, image_size=image)
        self.model = CLIPVisionModel(cfg)
        self.width = width

    def forward(self, x):
        m = self.model
        vm = getattr(m, "vision_model", m)       # transformers < 5 nests the tower under .vision_model
        h = vm.post_layernorm(m(pixel_values=x).last_hidden_state)
        return h[:, 0], h[:, 1:]


class VisionStandIn(nn.Module):
    """Object with a ``.visual`` attribute, like an open_clip CLIP model."""

    def __init__(self, **kw):
        super().__init__()
        self.visual = ClipVisualStandIn(**kw)


def build_lang_encoder(family: str, extra_tokens: int = 3, max_seq_len: int = 2048):
    f = FAMILY[family]
    vocab = f["vocab"] + extra_tokens      # <|endofchunk|>, <image>, <PAD> appended by factory.py:57-63
    if f["lm"] == "mpt":
        from transformers import MptConfig, MptForCausalLM
        cfg = MptConfig(d_model=f["d"], n_heads=f["heads"], n_layers=f["layers"], vocab_size=vocab,
                        max_seq_len=max_seq_len)
        return MptForCausalLM(cfg), "transformer.blocks"
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    cfg = GPTNeoXConfig(hidden_size=f["d"], num_hidden_layers=f["layers"], num_attention_heads=f["heads"],
                        intermediate_size=4 * f["d"], vocab_size=vocab, max_position_embeddings=max_seq_len)
    return GPTNeoXForCausalLM(cfg), "gpt_neox.layers"


def build_flamingo(family: str = "OF-3B", device="cuda", seed: int = 0, gates: float = 0.5, vision_kw=None,
                   freeze_lm_embeddings: bool = False, verbose: bool = False):
    """Random-init Flamingo of the given family, assembled through the factory path, gates set to ``gates``
    (their init value 0 makes the hot path an exact no-op with zero weight gradients -- SURVEY.md section 7)."""
    from ..src.factory import assemble_flamingo
    f = FAMILY[family]
    torch.manual_seed(seed)
    dev = torch.device(device)
    with torch.device(dev):
        vkw = vision_kw or (dict(width=64, layers=2, heads=2, patch=14, image=224) if family == "OF-tiny" else {})
        vision = VisionStandIn(**vkw)
        lm, attr = build_lang_encoder(family)
        vis_dim = vision.visual.width
        media_id, eoc_id = f["vocab"] + 1, f["vocab"]
        model = assemble_flamingo(vision, lm, eoc_id, media_id, vis_dim=vis_dim, cross_attn_every_n_layers=f["every"],
                                  decoder_layers_attr_name=attr, freeze_lm_embeddings=freeze_lm_embeddings,
                                  verbose=verbose)
    with torch.no_grad():
        for blk in model.lang_encoder.gated_cross_attn_layers:
            if blk is not None:
                blk.attn_gate.fill_(gates)
                blk.ff_gate.fill_(gates)
    return model, dict(f, media_token_id=media_id, eoc_token_id=eoc_id, pad_token_id=f["vocab"] + 2)
