"""Training checkpoints in the reference's on-disk format.

File layout of ``open_flamingo/train/train_utils.py:336-375`` -- ``{run_name}/checkpoint_{epoch}.pt`` holding
``epoch``, ``model_state_dict``, ``optimizer_state_dict``, ``lr_scheduler_state_dict`` -- with the model part reduced
the way ``filter_state_dict_to_trainable`` (:299-333) reduces it, so files written here resume in the reference's
``train.py:282-308,417-422,452-454`` and the reference's files (and the released ``checkpoint.pt`` state dicts,
README.md:118-127) load here.  The optimizer part is a ``torch.optim.AdamW`` state dict whichever optimizer produced it
(``FlatAdamW.state_dict`` emits that layout).
"""
import glob
import os
import re

import torch

_ALIAS_PREFIXES = ("lang_encoder.old_decoder_blocks", "lang_encoder.gated_cross_attn_layers")


def trainable_state_dict(model, state_dict=None):
    """What a checkpoint keeps of ``model.state_dict()``:

    * frozen parameters are dropped -- except those whose name contains ``embed`` (the reference keeps embeddings so
      the rows of the added ``<image>`` / ``<|endofchunk|>`` tokens survive even with ``freeze_lm_embeddings``; note
      that MPT calls its table ``transformer.wte``, which that rule does not match, exactly as in the reference);
    * the alias views ``lang_encoder.old_decoder_blocks.*`` / ``lang_encoder.gated_cross_attn_layers.*`` (the same
      tensors are saved under the decoder-layer names) and everything under ``vision_encoder`` are dropped;
    * buffers stay.
    """
    from .sparse_rows import is_trainable          # the table in sparse-rows mode is frozen for autograd only
    sd = dict(model.state_dict() if state_dict is None else state_dict)
    frozen = {name.replace("._checkpoint_wrapped_module", "")
              for name, p in model.named_parameters()
              if not is_trainable(p) and "embed" not in name and "fsdp" not in name}
    return {k: v for k, v in sd.items()
            if k not in frozen and "vision_encoder" not in k and not any(a in k for a in _ALIAS_PREFIXES)}


def checkpoint_path(run_name, epoch):
    return os.path.join(run_name, f"checkpoint_{epoch}.pt")


def latest_checkpoint(run_name):
    """train.py:283-295: the highest-numbered ``checkpoint_<n>.pt`` of a run directory, or None."""
    found = []
    for path in glob.glob(os.path.join(run_name, "checkpoint_*.pt")):
        m = re.search(r"checkpoint_(\d+)\.pt$", path)
        if m:
            found.append((int(m.group(1)), path))
    return max(found)[1] if found else None


def save_checkpoint(model, optimizer, lr_scheduler, epoch, run_name, rank=0, delete_previous_checkpoint=False):
    """Rank 0 writes ``{run_name}/checkpoint_{epoch}.pt``; returns the path (None on other ranks)."""
    if rank != 0:
        return None
    os.makedirs(run_name, exist_ok=True)
    payload = {
        "epoch": epoch,
        "model_state_dict": trainable_state_dict(model),
        "optimizer_state_dict": optimizer.state_dict(),
        "lr_scheduler_state_dict": lr_scheduler.state_dict() if lr_scheduler is not None else {},
    }
    path = checkpoint_path(run_name, epoch)
    torch.save(payload, path)
    if delete_previous_checkpoint and epoch > 0:
        previous = checkpoint_path(run_name, epoch - 1)
        if os.path.exists(previous):
            os.remove(previous)
    return path


def load_model_state(model, state_dict):
    """``model.load_state_dict(..., strict=False)`` after stripping DDP's ``module.`` prefix (train.py:302-308).  Also
    accepts a released ``checkpoint.pt`` (a bare state dict).  Returns torch's (missing, unexpected) report; every key
    of the file must have found a home."""
    sd = state_dict.get("model_state_dict", state_dict)
    sd = {k.replace("module.", ""): v for k, v in sd.items()}
    report = model.load_state_dict(sd, strict=False)
    if report.unexpected_keys:
        raise KeyError(f"checkpoint keys with no counterpart in the model: {report.unexpected_keys[:8]}")
    for mod in model.modules():                      # parameters changed: drop bf16 operand copies made from old values
        invalidate = getattr(mod, "invalidate_weight_cache", None)
        if invalidate is not None:
            invalidate()
    return report


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, map_location="cpu"):
    """Resume: returns the epoch to continue from (saved epoch + 1)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    load_model_state(model, ckpt)
    if optimizer is not None and ckpt.get("optimizer_state_dict"):
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if lr_scheduler is not None and ckpt.get("lr_scheduler_state_dict"):
        lr_scheduler.load_state_dict(ckpt["lr_scheduler_state_dict"])
    return ckpt.get("epoch", -1) + 1
