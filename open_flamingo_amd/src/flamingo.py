"""Top-level model with the reference's API (open_flamingo/src/flamingo.py).

Public methods and their reference counterparts: ``forward`` (:60-122), ``generate`` (:124-175), ``cache_media`` /
``uncache_media`` (:315-338); internal steps ``_encode_vision_x`` (:177-200) and ``_condition_media_locations``
(:303-313).  The Perceiver and the interleaved gated cross-attention blocks are the libofhip-backed modules of
``open_flamingo_amd.src.helpers``; the vision tower and the language model are whatever frozen modules the caller
injects (open_clip / HF in the reference's factory, random-init stand-ins in the benchmark).  FSDP wrapping (reference
:202-301) is outside this build: data-parallel training uses ``open_flamingo_amd.train.reducer``.
"""
import torch
from torch import nn

from .helpers import PerceiverResampler


def _lm_width(lang_encoder: nn.Module) -> int:
    cfg = lang_encoder.config
    return cfg.d_model if hasattr(cfg, "d_model") else cfg.hidden_size      # MPT says d_model, everyone else hidden_size


class Flamingo(nn.Module):
    """vision tower -> PerceiverResampler -> (gated cross attention + frozen LM layer) x N -> LM head."""

    def __init__(self, vision_encoder: nn.Module, lang_encoder: nn.Module, eoc_token_id: int, media_token_id: int,
                 vis_dim: int, cross_attn_every_n_layers: int = 1, gradient_checkpointing: bool = False):
        super().__init__()
        self.eoc_token_id, self.media_token_id = eoc_token_id, media_token_id
        self.vis_dim, self.lang_dim = vis_dim, _lm_width(lang_encoder)
        self.vision_encoder = vision_encoder.visual             # called as visual(x) -> (pooled, patch tokens)
        self.perceiver = PerceiverResampler(dim=vis_dim)
        self.lang_encoder = lang_encoder
        lang_encoder.init_flamingo(media_token_id=media_token_id, lang_hidden_size=self.lang_dim,
                                   vis_hidden_size=vis_dim, cross_attn_every_n_layers=cross_attn_every_n_layers,
                                   gradient_checkpointing=gradient_checkpointing)
        self._use_gradient_checkpointing = gradient_checkpointing
        self.perceiver._use_gradient_checkpointing = gradient_checkpointing
        # one Scope (hip/path.py: bf16-twin registry + shared per-forward artefacts) for this model's whole module tree: two
        # models in one process never see each other's host-side state
        from ..hip import path as _path
        _path.adopt(self)

    group_media_projections = True     # class-level switch (instance attribute overrides): see _encode_vision_x
    prefetch_at_block = None           # schedule_vision_prefetch: gated block (forward order) whose backward starts the tower; None = n_blocks // 12

    # ------------------------------------------------------------------------------------------------ vision prefetch
    def prefetch_vision(self, vision_x: torch.Tensor, amp_dtype=None):
        """Run the frozen vision tower (reference flamingo.py:194-195: no_grad, depends on no trainable parameter) for the
        ``vision_x`` of a FUTURE forward now, on a side HIP stream that starts behind everything enqueued so far on the current
        stream.  Called by ``train_step`` between the backward and the step epilogue: the tower's MFMA-bound GEMMs then share
        the chip with the HBM-bound clip + AdamW passes (and, under data parallelism, with the wait for the last gradient
        all-reduce) instead of opening the next step.  The forward that is later called with the SAME tensor object (unchanged:
        version counter) waits for the side stream and takes the tokens; any other input runs the tower inline as before.  The
        arithmetic is the tower's own forward under the same autocast dtype (``amp_dtype``; None = no autocast): identical bits."""
        assert vision_x.ndim == 6, "vision_x should be of shape (b, T_img, F, C, H, W)"
        import contextlib
        ac = (torch.autocast(device_type=vision_x.device.type, dtype=amp_dtype) if amp_dtype is not None else contextlib.nullcontext())
        if vision_x.is_cuda:
            main = torch.cuda.current_stream(vision_x.device)
            side = self.__dict__.get("_of_vision_stream")
            if side is None or side.device != vision_x.device:
                side = self.__dict__["_of_vision_stream"] = torch.cuda.Stream(device=vision_x.device)
            side.wait_stream(main)
            vision_x.record_stream(side)       # allocated on the main stream, read on the side one: its memory must outlive the tower
            with torch.cuda.stream(side), torch.no_grad(), ac:
                tokens = self.vision_encoder(vision_x.flatten(0, 2))[1]
            done = torch.cuda.Event()
            done.record(side)
        else:
            with torch.no_grad(), ac:
                tokens = self.vision_encoder(vision_x.flatten(0, 2))[1]
            done = None
        self.__dict__["_of_vision_prefetch"] = (vision_x, vision_x._version, tokens, done, amp_dtype)

    def schedule_vision_prefetch(self, vision_x: torch.Tensor, amp_dtype=None):
        """``prefetch_vision(vision_x)`` from INSIDE the backward of the next grad-enabled forward, so that the tower's forward on its
        side stream runs next to the END of the backward and the step epilogue and is done when the next step begins -- instead of
        starting behind the backward and running on alone for ~10 ms into the next step.  Where: at the start of the backward of gated
        block ``prefetch_at_block`` (forward order; default n_blocks // 12 -- OF-3B's 24 blocks: 2, OF-4B's 16: 1, OF-9B's 8: 0; same
        box, round 6, profiles/r06zzd_*, r06zze_*: behind the backward 105.75 ms per step, at the Perceiver's backward 104.6, at block
        2's 103.1; one block later / earlier +0.1...+0.6 ms, five blocks earlier +1 ms: two GEMM streams time-slice at a loss).  Without
        grouped media projections: when the gradient of the Perceiver's output is complete.  ``fire_vision_prefetch`` runs it at once
        if no backward did (no gradient path, a skipped step).  Same arithmetic, same bits: only the enqueue point moves."""
        self.__dict__["_of_prefetch_pending"] = (vision_x, amp_dtype)

    def fire_vision_prefetch(self):
        hit = self.__dict__.pop("_of_prefetch_pending", None)
        if hit is not None:
            self.prefetch_vision(hit[0], amp_dtype=hit[1])

    def cancel_vision_prefetch(self):
        self.__dict__.pop("_of_prefetch_pending", None)

    def _take_prefetched_vision(self, vision_x):
        hit = self.__dict__.pop("_of_vision_prefetch", None)
        if hit is None:
            return None
        # the consumer's autocast state must be the producer's (tokens of another dtype otherwise)
        amp_now = torch.get_autocast_dtype(vision_x.device.type) if torch.is_autocast_enabled(vision_x.device.type) else None
        if hit[0] is not vision_x or hit[1] != vision_x._version or hit[4] != amp_now:
            return None                        # a miss: the tokens were allocated on the side stream and go back to its pool
        tokens, done = hit[2], hit[3]
        if done is not None:
            main = torch.cuda.current_stream(vision_x.device)
            main.wait_event(done)
            tokens.record_stream(main)         # allocated on the side stream, consumed (and later freed) on this one
        return tokens

    # ------------------------------------------------------------------------------------------------ conditioning
    def _layers(self):
        return self.lang_encoder._get_decoder_layers()

    def _encode_vision_x(self, vision_x: torch.Tensor):
        """(b, T_img, F, C, H, W) pixels -> frozen vision tower (no grad) -> Perceiver -> every LM layer is handed the
        same (b, T_img, n_latents, vis_dim) tensor."""
        assert vision_x.ndim == 6, "vision_x should be of shape (b, T_img, F, C, H, W)"
        batch, n_media, n_frames = vision_x.shape[0], vision_x.shape[1], vision_x.shape[2]
        assert n_frames == 1, "Only single frame supported"
        tokens = self._take_prefetched_vision(vision_x)
        if tokens is None:
            with torch.no_grad():
                tokens = self.vision_encoder(vision_x.flatten(0, 2))[1]       # (b*T*F, patches, vis_dim)
        latents = self.perceiver(tokens.unflatten(0, (batch, n_media, n_frames)))
        if self.__dict__.get("_of_prefetch_pending") is not None and torch.is_grad_enabled() and latents.requires_grad:
            def _fire(grad, self=self):          # (tensor hook: runs in the backward, on the stream of the node that made `latents`)
                self.fire_vision_prefetch()
            latents.register_hook(_fire)
        # (not while media are being cached: several forwards may then run over the same latents, each with its own graph)
        if self.group_media_projections and torch.is_grad_enabled() and not self.lang_encoder._use_cached_vision_x:
            # every gated block applies its own to_kv to this one tensor: one grouped GEMM for all of them
            from . import helpers
            grp = helpers.group_media_projections(list(self.lang_encoder.gated_cross_attn_layers), latents)
            if grp is not None and self.__dict__.get("_of_prefetch_pending") is not None:
                k = self.prefetch_at_block if self.prefetch_at_block is not None else len(grp.blocks) // 12
                grp.backward_probe = (max(0, min(int(k), len(grp.blocks) - 1)), self.fire_vision_prefetch)
        for layer in self._layers():
            layer.condition_vis_x(latents)

    def _condition_media_locations(self, input_ids: torch.Tensor):
        is_media = input_ids == self.media_token_id
        for layer in self._layers():
            layer.condition_media_locations(is_media)

    def cache_media(self, input_ids: torch.Tensor, vision_x: torch.Tensor):
        """Pre-compute the media conditioning for a prompt; later ``forward(vision_x=None, ...)`` calls (e.g. scoring
        several continuations) attend to the LAST image of this prompt."""
        self._encode_vision_x(vision_x=vision_x)
        self._condition_media_locations(input_ids=input_ids)
        self.lang_encoder._use_cached_vision_x = True

    def uncache_media(self):
        self.lang_encoder.clear_conditioned_layers()
        self.lang_encoder._use_cached_vision_x = False

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, vision_x: torch.Tensor, lang_x: torch.Tensor, attention_mask: torch.Tensor = None,
                labels: torch.Tensor = None, clear_conditioned_layers: bool = True, past_key_values=None,
                use_cache: bool = False):
        """vision_x (B, T_img, F=1, C, H, W) or None when media were cached; lang_x (B, T_txt) token ids.
        Returns the HF causal-LM output (loss first when ``labels`` are given)."""
        lm = self.lang_encoder
        assert lm.initialized_flamingo, "Flamingo layers are not initialized. Please call `init_flamingo` first."
        cached = lm._use_cached_vision_x
        assert cached or vision_x is not None, "Must provide either vision_x or have precached media using cache_media()."
        if cached:
            assert vision_x is None, ("Expect vision_x to be None when media has been cached using cache_media(). "
                                      "Try uncache_media() first.")
            assert lm.is_conditioned()
        else:
            self._encode_vision_x(vision_x=vision_x)
            self._condition_media_locations(input_ids=lang_x)
        output = lm(input_ids=lang_x, attention_mask=attention_mask, labels=labels, past_key_values=past_key_values,
                    use_cache=use_cache)
        if clear_conditioned_layers:
            lm.clear_conditioned_layers()
        return output

    def generate(self, vision_x: torch.Tensor, lang_x: torch.Tensor, attention_mask: torch.Tensor = None, **kwargs):
        """HF ``generate`` conditioned on the images: the Perceiver runs once, its output stays cached on the layers for
        every decode step, and is dropped again when generation ends (also on error)."""
        num_beams = kwargs.pop("num_beams", 1)
        if num_beams > 1:
            vision_x = vision_x.repeat_interleave(num_beams, dim=0)
        eos_token_id = kwargs.pop("eos_token_id", self.eoc_token_id)
        lm = self.lang_encoder
        lm._use_cached_vision_x = True
        self._encode_vision_x(vision_x=vision_x)
        try:
            return lm.generate(input_ids=lang_x, attention_mask=attention_mask, eos_token_id=eos_token_id,
                               num_beams=num_beams, **kwargs)
        finally:
            lm.clear_conditioned_layers()
            lm._use_cached_vision_x = False

    def wrap_fsdp(self, wrapper_kwargs, device_id):
        raise NotImplementedError("FSDP wrapping (reference flamingo.py:202-301) is outside this build's scope; "
                                  "use open_flamingo_amd.train.reducer.GradReducer for data parallel training.")
