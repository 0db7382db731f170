"""Model assembly with the reference's entry point (open_flamingo/src/factory.py:11-119).

``create_model_and_transforms`` keeps the reference signature and flow (open_clip tower + HF tokenizer/LM +
special tokens + mixin + freeze/unfreeze).  ``assemble_flamingo`` is the offline-capable half: it takes already
constructed vision / language modules (e.g. random-init from configs, as the benchmark does -- there is no network
for checkpoints) and performs factory.py:85-113."""
from typing import Optional

from .flamingo import Flamingo
from .flamingo_lm import FlamingoLMMixin
from .utils import extend_instance

_KNOWN_DECODER_LAYERS_ATTR_NAMES = {   # reference factory.py:132-141
    "opt": "model.decoder.layers",
    "gptj": "transformer.h",
    "gpt-j": "transformer.h",
    "pythia": "gpt_neox.layers",
    "llama": "model.layers",
    "gptneoxforcausallm": "gpt_neox.layers",
    "mpt": "transformer.blocks",
    "mosaicgpt": "transformer.blocks",
}


def _infer_decoder_layers_attr_name(model):
    cls = model.__class__.__name__.lower()
    for key, attr in _KNOWN_DECODER_LAYERS_ATTR_NAMES.items():
        if key.lower() in cls:
            return attr
    raise ValueError("We require the attribute name for the nn.ModuleList in the decoder storing the transformer "
                     "block layers. Please supply this string manually.")


def assemble_flamingo(vision_encoder, lang_encoder, eoc_token_id: int, media_token_id: int, vis_dim: int,
                      cross_attn_every_n_layers: int = 1, decoder_layers_attr_name: Optional[str] = None,
                      freeze_lm_embeddings: bool = False, verbose: bool = True, **flamingo_kwargs) -> Flamingo:
    """factory.py:85-113 on pre-built modules: mix FlamingoLMMixin into the LM, build Flamingo, freeze everything,
    then unfreeze the Perceiver, the gated cross-attention blocks and (optionally) the input embeddings."""
    extend_instance(lang_encoder, FlamingoLMMixin)
    if decoder_layers_attr_name is None:
        decoder_layers_attr_name = _infer_decoder_layers_attr_name(lang_encoder)
    lang_encoder.set_decoder_layers_attr_name(decoder_layers_attr_name)
    model = Flamingo(vision_encoder, lang_encoder, eoc_token_id, media_token_id, vis_dim=vis_dim,
                     cross_attn_every_n_layers=cross_attn_every_n_layers, **flamingo_kwargs)
    model.requires_grad_(False)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 0
    model.perceiver.requires_grad_(True)
    model.lang_encoder.gated_cross_attn_layers.requires_grad_(True)
    if not freeze_lm_embeddings:
        model.lang_encoder.get_input_embeddings().requires_grad_(True)
    if verbose:
        n = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(f"Flamingo model initialized with {n} trainable parameters")
    return model


def create_model_and_transforms(clip_vision_encoder_path: str, clip_vision_encoder_pretrained: str,
                                lang_encoder_path: str, tokenizer_path: str, cross_attn_every_n_layers: int = 1,
                                use_local_files: bool = False, decoder_layers_attr_name: str = None,
                                freeze_lm_embeddings: bool = False, cache_dir: Optional[str] = None,
                                **flamingo_kwargs):
    """Returns (model, image_processor, tokenizer) exactly like the reference.  Needs ``open_clip`` and access to
    the pretrained vision/LM weights, neither of which exists in the offline build image; see assemble_flamingo."""
    try:
        import open_clip
    except ImportError as exc:
        raise ImportError("create_model_and_transforms needs the `open_clip_torch` package (reference "
                          "requirements.txt:6); offline, build the towers yourself and call "
                          "open_flamingo_amd.src.factory.assemble_flamingo") from exc
    from transformers import AutoModelForCausalLM, AutoTokenizer

    vision_encoder, _, image_processor = open_clip.create_model_and_transforms(
        clip_vision_encoder_path, pretrained=clip_vision_encoder_pretrained, cache_dir=cache_dir)
    vision_encoder.visual.output_tokens = True   # visual(x) -> (pooled, tokens)
    tok = AutoTokenizer.from_pretrained(tokenizer_path, local_files_only=use_local_files, trust_remote_code=True,
                                        cache_dir=cache_dir)
    tok.add_special_tokens({"additional_special_tokens": ["<|endofchunk|>", "<image>"]})
    if tok.pad_token is None:
        tok.add_special_tokens({"pad_token": "<PAD>"})
    lang_encoder = AutoModelForCausalLM.from_pretrained(lang_encoder_path, local_files_only=use_local_files,
                                                        trust_remote_code=True, cache_dir=cache_dir)
    if "mpt-1b-redpajama-200b" in lang_encoder_path:   # this checkpoint's remote code lacks the accessors
        class EmbeddingFnMixin:
            def get_input_embeddings(self):
                return self.transformer.wte

            def set_input_embeddings(self, new_embeddings):
                self.transformer.wte = new_embeddings
        extend_instance(lang_encoder, EmbeddingFnMixin)
    # the reference resizes after mixing in FlamingoLMMixin (factory.py:85-90); the order is immaterial
    lang_encoder.resize_token_embeddings(len(tok))
    model = assemble_flamingo(vision_encoder, lang_encoder, tok.encode("<|endofchunk|>")[-1],
                              tok.encode("<image>")[-1],
                              vis_dim=open_clip.get_model_config(clip_vision_encoder_path)["vision_cfg"]["width"],
                              cross_attn_every_n_layers=cross_attn_every_n_layers,
                              decoder_layers_attr_name=decoder_layers_attr_name,
                              freeze_lm_embeddings=freeze_lm_embeddings, **flamingo_kwargs)
    return model, image_processor, tok
