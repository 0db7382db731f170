"""Interleaving of the gated cross-attention blocks into a frozen HF causal LM.

Same public surface as the reference ``open_flamingo/src/flamingo_lm.py`` (FlamingoLayer :6-66, FlamingoLMMixin
:69-167): attribute names (``gated_cross_attn_layers``, ``old_decoder_blocks``, ``gated_cross_attn_layer``,
``decoder_layer``), the conditioning side channel (``condition_vis_x`` / ``condition_media_locations`` /
``condition_use_cached_media``), the interleave rule ``(layer_idx + 1) % cross_attn_every_n_layers == 0`` (:100) and
the cached-media decode logic (:142-151).  The cross-attention block itself is the libofhip-backed
``GatedCrossAttentionBlock``; the decoder layer stays the untouched HF module.
"""
import torch.nn as nn

from .helpers import GatedCrossAttentionBlock
from .utils import getattr_recursive, setattr_recursive


class FlamingoLayer(nn.Module):
    """(optional gated cross-attention) -> frozen decoder layer.  Conditioning is stashed on the layer because the
    HF layer loop only forwards hidden states."""

    def __init__(self, gated_cross_attn_layer, decoder_layer, gradient_checkpointing=False):
        super().__init__()
        self.gated_cross_attn_layer = gated_cross_attn_layer
        self.decoder_layer = decoder_layer
        self.vis_x = None
        self.media_locations = None
        self.use_cached_media = None
        for m in (gated_cross_attn_layer, decoder_layer):
            if m is not None:
                m._use_gradient_checkpointing = gradient_checkpointing

    def is_conditioned(self) -> bool:
        return self.vis_x is not None and self.media_locations is not None

    def condition_vis_x(self, vis_x):
        self.vis_x = vis_x

    def condition_media_locations(self, media_locations):
        self.media_locations = media_locations

    def condition_use_cached_media(self, use_cached_media):
        self.use_cached_media = use_cached_media

    def forward(self, lang_x, attention_mask=None, **decoder_layer_kwargs):
        xattn = self.gated_cross_attn_layer
        if xattn is not None:
            if self.vis_x is None:
                raise ValueError("vis_x must be conditioned before forward pass")
            if self.media_locations is None:
                raise ValueError("media_locations must be conditioned before forward pass")
            lang_x = xattn(lang_x, self.vis_x, media_locations=self.media_locations,
                           use_cached_media=self.use_cached_media)
        return self.decoder_layer(lang_x, attention_mask=attention_mask, **decoder_layer_kwargs)


class FlamingoLMMixin(nn.Module):
    """Mixed into an HF causal LM instance by ``extend_instance`` (factory)."""

    def set_decoder_layers_attr_name(self, decoder_layers_attr_name):
        self.decoder_layers_attr_name = decoder_layers_attr_name

    def _get_decoder_layers(self):
        return getattr_recursive(self, self.decoder_layers_attr_name)

    def _set_decoder_layers(self, value):
        setattr_recursive(self, self.decoder_layers_attr_name, value)

    def init_flamingo(self, media_token_id, lang_hidden_size, vis_hidden_size, cross_attn_every_n_layers,
                      gradient_checkpointing):
        self.old_decoder_blocks = self._get_decoder_layers()
        n_layers = len(self.old_decoder_blocks)
        self.gated_cross_attn_layers = nn.ModuleList([
            GatedCrossAttentionBlock(dim=lang_hidden_size, dim_visual=vis_hidden_size)
            if (idx + 1) % cross_attn_every_n_layers == 0 else None
            for idx in range(n_layers)])
        self.init_flamingo_layers(gradient_checkpointing)
        self.media_token_id = media_token_id
        self.initialized_flamingo = True
        self._use_cached_vision_x = False

    def init_flamingo_layers(self, gradient_checkpointing):
        """(Re)build the FlamingoLayer list from gated_cross_attn_layers / old_decoder_blocks."""
        pairs = zip(self.gated_cross_attn_layers, self.old_decoder_blocks)
        self._set_decoder_layers(nn.ModuleList([FlamingoLayer(x, d, gradient_checkpointing) for x, d in pairs]))

    def forward(self, input_ids, attention_mask, **kwargs):
        if not self.initialized_flamingo:
            raise ValueError("Flamingo layers are not initialized. Please call `init_flamingo` first.")
        media_locations = input_ids == self.media_token_id
        # HF generate() re-enters with one new token and no <image>: attend to the last cached media instead
        use_cached = self._use_cached_vision_x and self.is_conditioned() and not media_locations.any()
        for layer in self._get_decoder_layers():
            if not use_cached:
                layer.condition_media_locations(media_locations)
            layer.condition_use_cached_media(use_cached)
        kwargs["input_ids"] = input_ids
        kwargs["attention_mask"] = attention_mask
        return super().forward(**kwargs)

    def is_conditioned(self) -> bool:
        return all(layer.is_conditioned() for layer in self._get_decoder_layers())

    def clear_conditioned_layers(self):
        for layer in self._get_decoder_layers():
            layer.condition_vis_x(None)
            layer.condition_media_locations(None)
            layer.condition_use_cached_media(None)
