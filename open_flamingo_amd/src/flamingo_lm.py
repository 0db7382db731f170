"""Interleaving of the gated cross-attention blocks into a frozen HF causal LM.

Drop-in for the reference ``open_flamingo/src/flamingo_lm.py`` (``FlamingoLayer`` :6-66, ``FlamingoLMMixin`` :69-167).
What callers rely on, and what is therefore kept name for name:

* attributes ``gated_cross_attn_layers`` / ``old_decoder_blocks`` on the LM (factory.py:109-112, train_utils.py:299-333
  and the released checkpoints' key names go through them) and ``gated_cross_attn_layer`` / ``decoder_layer`` on a layer;
* the conditioning side channel ``condition_vis_x`` / ``condition_media_locations`` / ``condition_use_cached_media``
  (flamingo.py:199-200,303-313): the HF layer loop only forwards hidden states, so the media travel as layer state;
* the interleave rule of flamingo_lm.py:100 -- block i gets cross attention iff ``(i + 1) % every == 0``;
* the decode rule of flamingo_lm.py:142-151 -- a forward with no ``<image>`` token while media are cached attends to the
  last cached media.

The cross-attention block is the libofhip-backed ``GatedCrossAttentionBlock``; decoder layers are the untouched HF modules.
"""
from typing import Iterator, Optional

import torch.nn as nn

from .helpers import GatedCrossAttentionBlock
from .utils import getattr_recursive, setattr_recursive

_CONDITION_SLOTS = ("vis_x", "media_locations", "use_cached_media")


def _has_cross_attention(layer_index: int, every: int) -> bool:
    return (layer_index + 1) % every == 0


class FlamingoLayer(nn.Module):
    """One LM layer of the interleaved stack: optional gated cross attention, then the frozen decoder layer."""

    def __init__(self, gated_cross_attn_layer: Optional[nn.Module], decoder_layer: nn.Module,
                 gradient_checkpointing: bool = False):
        super().__init__()
        self.gated_cross_attn_layer = gated_cross_attn_layer
        self.decoder_layer = decoder_layer
        for slot in _CONDITION_SLOTS:
            setattr(self, slot, None)
        # train.py:369-381 wraps modules carrying this flag in a checkpoint wrapper
        if gated_cross_attn_layer is not None:
            gated_cross_attn_layer._use_gradient_checkpointing = gradient_checkpointing
        decoder_layer._use_gradient_checkpointing = gradient_checkpointing

    # ---- conditioning side channel -------------------------------------------------------------------------------
    def condition_vis_x(self, vis_x) -> None:
        self.vis_x = vis_x

    def condition_media_locations(self, media_locations) -> None:
        self.media_locations = media_locations

    def condition_use_cached_media(self, use_cached_media) -> None:
        self.use_cached_media = use_cached_media

    def is_conditioned(self) -> bool:
        """Media features AND their positions in the text are known."""
        return not (self.vis_x is None or self.media_locations is None)

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, lang_x, attention_mask=None, **decoder_layer_kwargs):
        block = self.gated_cross_attn_layer
        if block is not None:
            for slot in ("vis_x", "media_locations"):
                if getattr(self, slot) is None:
                    raise ValueError(f"{slot} must be conditioned before forward pass")
            lang_x = block(lang_x, self.vis_x, media_locations=self.media_locations,
                           use_cached_media=self.use_cached_media)
        # whatever the HF decoder layer returns (tensor or tuple) is handed back unchanged
        return self.decoder_layer(lang_x, attention_mask=attention_mask, **decoder_layer_kwargs)


class FlamingoLMMixin(nn.Module):
    """Grafted onto an HF causal-LM *instance* by ``utils.extend_instance`` (factory.py:85): its ``forward`` runs first
    and then defers to the LM's own ``forward`` through ``super()``."""

    # ---- where the decoder layers live in this LM ------------------------------------------------------------------
    def set_decoder_layers_attr_name(self, decoder_layers_attr_name: str) -> None:
        self.decoder_layers_attr_name = decoder_layers_attr_name

    def _get_decoder_layers(self):
        return getattr_recursive(self, self.decoder_layers_attr_name)

    def _set_decoder_layers(self, value) -> None:
        setattr_recursive(self, self.decoder_layers_attr_name, value)

    def _flamingo_layers(self) -> Iterator[FlamingoLayer]:
        return iter(self._get_decoder_layers())

    # ---- construction -----------------------------------------------------------------------------------------------
    def init_flamingo(self, media_token_id, lang_hidden_size, vis_hidden_size, cross_attn_every_n_layers,
                      gradient_checkpointing):
        """Create one gated cross-attention block per selected layer and swap the LM's layer list for FlamingoLayers."""
        self.old_decoder_blocks = self._get_decoder_layers()
        blocks = []
        for index in range(len(self.old_decoder_blocks)):
            wanted = _has_cross_attention(index, cross_attn_every_n_layers)
            blocks.append(GatedCrossAttentionBlock(dim=lang_hidden_size, dim_visual=vis_hidden_size) if wanted else None)
        self.gated_cross_attn_layers = nn.ModuleList(blocks)
        self.init_flamingo_layers(gradient_checkpointing)
        self.media_token_id = media_token_id
        self._use_cached_vision_x = False
        self.initialized_flamingo = True

    def init_flamingo_layers(self, gradient_checkpointing):
        """(Re)build the interleaved stack from ``gated_cross_attn_layers`` and ``old_decoder_blocks`` (also used after
        the blocks have been replaced or wrapped)."""
        stack = [FlamingoLayer(block, decoder, gradient_checkpointing)
                 for block, decoder in zip(self.gated_cross_attn_layers, self.old_decoder_blocks)]
        self._set_decoder_layers(nn.ModuleList(stack))

    # ---- forward ----------------------------------------------------------------------------------------------------
    def forward(self, input_ids, attention_mask, **kwargs):
        if not self.initialized_flamingo:
            raise ValueError("Flamingo layers are not initialized. Please call `init_flamingo` first.")
        media_locations = input_ids == self.media_token_id
        # During generate() HF re-enters with only the new token(s): no <image> among them while media are cached means
        # "keep attending to the last media", and the stored media_locations must survive.
        attend_cached = bool(self._use_cached_vision_x and self.is_conditioned() and not media_locations.any())
        for layer in self._flamingo_layers():
            if not attend_cached:
                layer.condition_media_locations(media_locations)
            layer.condition_use_cached_media(attend_cached)
        return super().forward(input_ids=input_ids, attention_mask=attention_mask, **kwargs)

    # ---- state queries ------------------------------------------------------------------------------------------------
    def is_conditioned(self) -> bool:
        return all(layer.is_conditioned() for layer in self._flamingo_layers())

    def clear_conditioned_layers(self) -> None:
        for layer in self._flamingo_layers():
            for slot in _CONDITION_SLOTS:
                setattr(layer, slot, None)
            release = getattr(layer.gated_cross_attn_layer, "release_media_cache", None)
            if release is not None:
                release()            # the block's projected keys/values of the media that were just dropped
