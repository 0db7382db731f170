"""One optimizer step with the reference's semantics (open_flamingo/train/train_utils.py:94-216, train.py:392-408):
(optional LAION pass) + MMC4-style pass under amp_bf16, labels masked as in :127-150, embedding-gradient rows masked
to <image>/<|endofchunk|> (:174-196), global-norm clip 1.0 (:199-208), AdamW with weight decay only on the
gated cross-attention parameters (train.py:392-408).  Differences, all result-preserving:
one gradient exchange per step through GradReducer instead of one per backward; label masking vectorised on device.
"""
import contextlib

import torch

from . import synthetic


def build_optimizer(model, lr=1e-4, weight_decay=0.1, reducer=None):
    """train.py:384-408: wd only on params whose name contains 'gated_cross_attn'.  With a GradReducer on a GPU the
    fused device-side step epilogue (train/optim.py: clip + AdamW over the reducer's flat buckets) is returned; it
    implements the same update."""
    dense_table = reducer is not None and reducer.embedding is not None and reducer.embedding_rows is None
    if reducer is not None and reducer.buckets and reducer.buckets[0]["flat"].is_cuda and not dense_table:
        from .optim import FlatAdamW
        return FlatAdamW(reducer, lr=lr, weight_decay=weight_decay)
    if getattr(model, "_of_sparse_rows", None) is not None:
        raise RuntimeError("train/sparse_rows.py froze the embedding table for autograd: its two trained rows are only "
                           "updated by the fused step epilogue (build_optimizer(model, reducer=GradReducer) on a GPU)")
    with_wd, without_wd = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad or getattr(p, "exclude_from_optimizer", False):
            continue
        (with_wd if "gated_cross_attn" in n else without_wd).append(p)
    groups = [{"params": with_wd, "weight_decay": weight_decay}, {"params": without_wd, "weight_decay": 0.0}]
    fused = all(p.is_cuda for p in with_wd + without_wd)
    return torch.optim.AdamW(groups, lr=lr, fused=fused)


def _autocast(device_type, enabled=True):
    return torch.autocast(device_type=device_type, dtype=torch.bfloat16, enabled=enabled)


def forward_loss(model, batch, info, amp=True, kind="mmc4"):
    """kind: "mmc4" (interleaved label rule, train_utils.py:127-150) or "laion" (train_utils.py:102-105)."""
    ids = batch["lang_x"]
    if kind == "laion":
        labels = synthetic.make_labels_laion(ids, info["media_token_id"], info["pad_token_id"])
    else:
        assert kind == "mmc4", kind
        labels = synthetic.make_labels(ids, info["media_token_id"], info["eoc_token_id"], info["pad_token_id"])
    with _autocast(ids.device.type, amp):
        out = model(vision_x=batch["vision_x"], lang_x=ids, attention_mask=batch["attention_mask"], labels=labels)
    return out[0]


def mask_embedding_gradient(model, info):
    """train_utils.py:174-196 as written there: only the <image> / <|endofchunk|> rows of the input-embedding gradient
    survive.  (With a GradReducer the same mask is applied by the reducer, which exchanges just those rows.)"""
    w = model.lang_encoder.get_input_embeddings().weight
    if w.grad is None:
        return
    zero_mask = torch.zeros_like(w.grad)
    zero_mask[info["media_token_id"]] = 1
    zero_mask[info["eoc_token_id"]] = 1
    w.grad = w.grad * zero_mask


# The next step's frozen-tower forward (train_step(next_vision_x=...)) is enqueued on its side stream from INSIDE the backward -- at the start of
# the backward of one of the last gated blocks (Flamingo.schedule_vision_prefetch: so that the tower ends with the step epilogue) -- instead of
# behind the whole backward: 105.75 -> 103.1 ms per step same box (profiles/r06zzd_*, r06zze_*; tools/ab_prefetch_point.py flips this).
PREFETCH_IN_BACKWARD = True


def train_step(model, reducer, optimizer, batch_mmc4, info, batch_laion=None, loss_multiplier_laion=1.0,
               loss_multiplier_mmc4=1.0, clip_norm=1.0, amp=True, nan_check=True, lr_scheduler=None,
               mask_embedding_rows=True, next_vision_x=None):
    """Returns the (detached) MMC4 loss tensor, or None if the step was skipped because the loss was NaN.
    reducer=None is the single-process form of the reference loop (embedding-gradient mask applied in place).
    nan_check: True = the reference's host-side ``torch.isnan(loss)`` (train_utils.py:161-169: one host sync per step, and under
    data parallelism a NaN on one rank alone leaves the others waiting in their collectives); "device" = no host sync -- the
    backward runs, the NaN reaches every rank through the all-reduce, and the fused step epilogue (of_adamw_clip) skips the
    update on a non-finite global norm, on all ranks alike (needs the FlatAdamW optimizer; the returned loss is then NaN
    instead of None).  Adam's step count lives on the device and only counts applied updates, so a skipped step leaves the
    bias correction where the reference's `continue` leaves it; a host-side ``lr_scheduler`` cannot see the skip without a
    sync and does advance by one step per NaN batch -- the one remaining difference from the reference, bounded by the number
    of NaN batches (use nan_check=True where that matters more than the sync); False = no check.
    next_vision_x: the ``vision_x`` tensor the NEXT step's first forward will be called with (a data loader is one batch ahead
    anyway).  Its frozen vision-tower forward -- which depends on no trainable parameter (flamingo.py:194-195) -- is enqueued on
    a side stream from inside this step's backward, a few gated blocks before its end (Flamingo.schedule_vision_prefetch), so that it
    runs next to the end of the backward, the HBM-bound clip + AdamW passes and the wait for the last all-reduce, and is done when
    the next step begins.  Same arithmetic, same bits; every step still runs exactly one tower forward per forward pass."""
    fused = hasattr(optimizer, "reducer")         # FlatAdamW: clip + AdamW + zero_grad in two device passes
    from ..hip import path as _path
    _path.scope_of(getattr(model, "module", model)).twins.clear()         # a gradient twin nobody took in the previous backward (the embedding's) is not kept alive
    params = None if fused else [p for g in optimizer.param_groups for p in g["params"]]
    if batch_laion is not None:
        with reducer.no_sync() if reducer is not None else contextlib.nullcontext():
            loss_l = forward_loss(model, batch_laion, info, amp, kind="laion")
            (loss_l * loss_multiplier_laion).backward()
    inner = getattr(model, "module", model)
    in_backward = next_vision_x is not None and PREFETCH_IN_BACKWARD and hasattr(inner, "schedule_vision_prefetch")
    if hasattr(inner, "cancel_vision_prefetch"):
        inner.cancel_vision_prefetch()         # (a request an earlier, aborted step left behind)
    if in_backward:            # fires from the backward below (Flamingo.schedule_vision_prefetch)
        inner.schedule_vision_prefetch(next_vision_x, amp_dtype=torch.bfloat16 if amp else None)
    loss = forward_loss(model, batch_mmc4, info, amp)
    if nan_check == "device":
        assert fused, "nan_check='device' relies on the fused step epilogue (FlatAdamW)"
    elif nan_check and torch.isnan(loss):          # train_utils.py:161-169 (host sync, as in the reference)
        if reducer is not None:
            reducer.zero_grad()
        else:
            optimizer.zero_grad(set_to_none=True)
        if in_backward:
            inner.cancel_vision_prefetch()
        return None
    (loss * loss_multiplier_mmc4).backward()
    if in_backward:
        inner.fire_vision_prefetch()           # (no-op when the backward's hook ran it)
    elif next_vision_x is not None:
        inner.prefetch_vision(next_vision_x, amp_dtype=torch.bfloat16 if amp else None)
    if reducer is None and mask_embedding_rows:
        mask_embedding_gradient(model, info)
    if reducer is not None:
        # waits for the overlapped RCCL all-reduces, restores the two embedding rows; the fused epilogue averages itself
        reducer.finish(average=not hasattr(optimizer, "reducer"))
    if fused:
        optimizer.max_norm = clip_norm
        optimizer.step()
    else:
        torch.nn.utils.clip_grad_norm_(params, clip_norm)
        optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    if reducer is not None:
        reducer.zero_grad(flat_already_zero=fused)
    else:
        optimizer.zero_grad(set_to_none=True)
    return loss.detach()
