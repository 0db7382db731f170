"""The product modules under the wrappers the reference's train.py puts around them ("must survive
DDP(...), checkpoint_wrapper"):

  * activation checkpointing exactly as open_flamingo/train/train.py:369-381 applies it -- non-reentrant
    `checkpoint_wrapper` on every module carrying `_use_gradient_checkpointing` (the Perceiver, every gated block, every
    decoder layer; flamingo.py:57-58, flamingo_lm.py:19-23) -- with and without a GradReducer (in-place bucket gradients,
    beta = 0 "fresh" overwrites, the grouped media projections whose state is freed in the group's backward);
  * `DistributedDataParallel(model)` as train.py:364-366 (a one-rank process group: DDP's reducer hooks, bucket views and
    gradient-ready accounting all run).

Gradients must equal the unwrapped run's.  CPU: the product host code on the host SIMT emulator (tests/emu); GPU: the same
on libofhip."""
import functools
import os

import pytest
import torch
from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import (ActivationWrapper, CheckpointImpl, CheckpointWrapper,
                                                                          apply_activation_checkpointing, checkpoint_wrapper)

from open_flamingo_amd.train import step, synthetic, towers
from open_flamingo_amd.train.reducer import GradReducer


def _build(device, seed=0, emu=False):
    kw = dict(vision_kw=dict(width=64, layers=2, heads=2, patch=14, image=56), perceiver_depth=2, fused_lm_attention=False) if emu else {}
    model, info = towers.build_flamingo("OF-tiny", device=device, seed=seed, gates=0.5, **kw)
    model.train()
    return model, info


def _flag_for_checkpointing(model):
    """what Flamingo(gradient_checkpointing=True) does (flamingo.py:57-58, flamingo_lm.py:19-23)"""
    model._use_gradient_checkpointing = True
    model.perceiver._use_gradient_checkpointing = True
    for layer in model.lang_encoder._get_decoder_layers():
        if layer.gated_cross_attn_layer is not None:
            layer.gated_cross_attn_layer._use_gradient_checkpointing = True
        layer.decoder_layer._use_gradient_checkpointing = True


def _wrap_like_reference_train_py(model, offload):
    """open_flamingo/train/train.py:369-381"""
    # torch 2.0.1 (the reference's pin) spelled the CPU offload `checkpoint_wrapper(offload_to_cpu=True)`; torch >= 2.1 moved it
    # into `offload_wrapper` (save_on_cpu hooks around the wrapped module)
    if offload:
        from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import offload_wrapper

        def wrapper(m):
            return offload_wrapper(checkpoint_wrapper(m, checkpoint_impl=CheckpointImpl.NO_REENTRANT))
    else:
        wrapper = functools.partial(checkpoint_wrapper, checkpoint_impl=CheckpointImpl.NO_REENTRANT)
    apply_activation_checkpointing(
        model, checkpoint_wrapper_fn=wrapper,
        check_fn=lambda m: getattr(m, "_use_gradient_checkpointing", False) and not isinstance(m, ActivationWrapper))
    return sum(isinstance(m, CheckpointWrapper) for m in model.modules())


def _grads(model):
    out = {}
    for k, p in model.named_parameters():
        if p.requires_grad and p.grad is not None and "wte" not in k:
            out[k.replace("_checkpoint_wrapped_module.", "")] = p.grad.detach().float().cpu().clone()
    return out


def _run_plain(device, emu):
    model, info = _build(device, emu=emu)
    batch = synthetic.make_batch(2, 2, 24, info, device, seed=5, image_size=56 if emu else 224)
    loss = step.forward_loss(model, batch, info)
    loss.backward()
    return float(loss), _grads(model)


def _check_same(got, want, tol):
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:6]
    for k in want:
        scale = want[k].abs().max().item() + 1e-12
        assert (got[k] - want[k]).abs().max().item() <= tol * scale + 1e-7, k


def _checkpoint_case(device, emu, with_reducer, offload):
    model, info = _build(device, emu=emu)
    _flag_for_checkpointing(model)
    n = _wrap_like_reference_train_py(model, offload)
    depth = len(model.lang_encoder.old_decoder_blocks)
    assert n >= 1 + depth, n            # the Perceiver, the decoder layers (+ the gated blocks)
    red = GradReducer(model, embedding_rows=None) if with_reducer else None
    batch = synthetic.make_batch(2, 2, 24, info, device, seed=5, image_size=56 if emu else 224)
    loss = step.forward_loss(model, batch, info)
    loss.backward()
    return float(loss), _grads(model)


@pytest.fixture
def on_emulator(monkeypatch):
    from open_flamingo_amd.hip.ops import Ops
    from open_flamingo_amd.src import helpers
    from tests.emu import harness as H
    monkeypatch.setattr(helpers, "_require_hip", lambda t, what: None)
    monkeypatch.setattr(Ops, "default", staticmethod(H.emu_ops))
    helpers._shared.items.clear()
    yield
    helpers._shared.items.clear()


@pytest.mark.parametrize("with_reducer", [False, True])
def test_checkpoint_wrapper_as_reference_train_py_on_emulator(on_emulator, with_reducer):
    l0, g0 = _run_plain("cpu", True)
    l1, g1 = _checkpoint_case("cpu", True, with_reducer, offload=False)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    _check_same(g1, g0, 1e-5)


def _ddp_case(device, emu, backend, port):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    created = not dist.is_initialized()
    if created:
        kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
        dist.init_process_group(backend, init_method="env://", world_size=1, rank=0, **kw)
    try:
        model, info = _build(device, emu=emu)
        ddp = DDP(model, device_ids=[0] if device != "cpu" else None)      # train.py:364-366
        assert ddp.module is model
        batch = synthetic.make_batch(2, 2, 24, info, device, seed=5, image_size=56 if emu else 224)
        labels = synthetic.make_labels(batch["lang_x"], info["media_token_id"], info["eoc_token_id"], info["pad_token_id"])
        with step._autocast("cpu" if device == "cpu" else "cuda"):
            out = ddp(vision_x=batch["vision_x"], lang_x=batch["lang_x"], attention_mask=batch["attention_mask"], labels=labels)
        out[0].backward()
        return float(out[0]), _grads(model)
    finally:
        if created:
            dist.destroy_process_group()


def test_ddp_wrap_as_reference_train_py_on_emulator(on_emulator):
    l0, g0 = _run_plain("cpu", True)
    l1, g1 = _ddp_case("cpu", True, "gloo", 29541)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    _check_same(g1, g0, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("with_reducer,offload", [(False, True), (True, False)])
def test_checkpoint_wrapper_as_reference_train_py_on_gpu(with_reducer, offload):
    l0, g0 = _run_plain("cuda", False)
    l1, g1 = _checkpoint_case("cuda", False, with_reducer, offload)
    assert abs(l0 - l1) <= 1e-4 * abs(l0)
    _check_same(g1, g0, 2e-3)          # split-K / LayerNorm column sums use fp32 atomics in some launches: run-to-run noise


@pytest.mark.gpu
def test_ddp_wrap_as_reference_train_py_on_gpu():
    l0, g0 = _run_plain("cuda", False)
    l1, g1 = _ddp_case("cuda", False, "nccl", 29542)
    assert abs(l0 - l1) <= 1e-4 * abs(l0)
    _check_same(g1, g0, 2e-3)
