"""checkpoint compatibility with the reference.

* the keys a checkpoint keeps == what the reference's ``filter_state_dict_to_trainable`` keeps of the reference model
  (golden tests/golden/checkpoint_keys.json, made by tests/golden/make_golden.py::checkpoint_keys from the real
  train_utils.py), for trainable and frozen input embeddings;
* the fused step epilogue's optimizer state is a ``torch.optim.AdamW`` state dict in the reference's parameter order
  (train.py:384-408): state crosses between FlatAdamW and torch AdamW in both directions and training continues
  identically;
* save -> load round trip through ``{run_name}/checkpoint_{epoch}.pt`` incl. DDP's ``module.`` prefix and the
  latest-checkpoint rule of train.py:283-295.
(CPU: the tiny Flamingo with the oracle's hot-path modules; FlatAdamW's kernels run on the host emulator.)"""
import json
import os

import pytest
import torch

from open_flamingo_amd.train import checkpoint, step, synthetic
from open_flamingo_amd.train.optim import FlatAdamW
from open_flamingo_amd.train.reducer import GradReducer
from tests.cpu_model import tiny_cpu_flamingo

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_keys.json")))


@pytest.mark.parametrize("tag", ["embeddings_trainable", "embeddings_frozen"])
def test_checkpoint_keys_and_optimizer_order_match_reference(tag):
    model, info = tiny_cpu_flamingo(seed=0)
    if tag == "embeddings_frozen":
        model.lang_encoder.get_input_embeddings().requires_grad_(False)
    want = GOLD[tag]
    assert sorted(checkpoint.trainable_state_dict(model).keys()) == want["checkpoint_keys"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    assert [n for n, _ in named if "gated_cross_attn" in n] == want["adamw_with_wd"]
    assert [n for n, _ in named if "gated_cross_attn" not in n] == want["adamw_without_wd"]
    opt = step.build_optimizer(model)                    # CPU -> torch AdamW with the reference's two groups
    by_id = {id(p): n for n, p in named}
    assert [by_id[id(p)] for p in opt.param_groups[0]["params"]] == want["adamw_with_wd"]
    assert [by_id[id(p)] for p in opt.param_groups[1]["params"]] == want["adamw_without_wd"]
    assert opt.param_groups[0]["weight_decay"] == 0.1 and opt.param_groups[1]["weight_decay"] == 0.0


def _trained(fused, steps, lr=1e-3):
    from tests.emu import harness as H
    model, info = tiny_cpu_flamingo(seed=0)
    red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt = FlatAdamW(red, lr=lr, ops=H.emu_ops()) if fused else step.build_optimizer(model, lr=lr)
    batch = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5)
    for _ in range(steps):
        step.train_step(model, red, opt, batch, info, amp=False)
    return model, red, opt, batch, info


def _params(model):
    return torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad]).double()


def test_fused_optimizer_state_is_a_torch_adamw_state_dict():
    m_f, r_f, o_f, batch, info = _trained(True, 2)
    m_t, r_t, o_t, _, _ = _trained(False, 2)
    sd_f, sd_t = o_f.state_dict(), o_t.state_dict()
    assert [g["params"] for g in sd_f["param_groups"]] == [g["params"] for g in sd_t["param_groups"]]
    assert sorted(sd_f["state"]) == sorted(sd_t["state"])
    for i in sd_t["state"]:
        assert float(sd_f["state"][i]["step"]) == float(sd_t["state"][i]["step"]) == 2.0
        for k in ("exp_avg", "exp_avg_sq"):
            a, b = sd_f["state"][i][k], sd_t["state"][i][k]
            assert a.shape == b.shape
            assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-12, (i, k)   # fp32 op order of the clip
    # swap the optimizer states and keep training: both runs must keep agreeing with the never-swapped trajectory
    o_f.load_state_dict(sd_t)
    o_t.load_state_dict(sd_f)
    assert o_f.step_count == 2
    for m, r, o in ((m_f, r_f, o_f), (m_t, r_t, o_t)):
        step.train_step(m, r, o, batch, info, amp=False)
    m_ref, _, _, _, _ = _trained(False, 3)
    want = _params(m_ref)
    for m in (m_f, m_t):
        assert (_params(m) - want).abs().max().item() < 2e-5


def test_checkpoint_round_trip(tmp_path):
    run = str(tmp_path / "run")
    model, red, opt, batch, info = _trained(True, 2)
    sched = torch.optim.lr_scheduler.LambdaLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0),
                                              lambda s: 1.0 / (1 + s))
    sched.step()
    assert checkpoint.latest_checkpoint(run) is None
    checkpoint.save_checkpoint(model, opt, sched, 0, run)
    path = checkpoint.save_checkpoint(model, opt, sched, 1, run, delete_previous_checkpoint=True)
    assert checkpoint.save_checkpoint(model, opt, sched, 1, run, rank=1) is None
    assert os.listdir(run) == ["checkpoint_1.pt"] and checkpoint.latest_checkpoint(run) == path
    blob = torch.load(path, weights_only=False)
    assert sorted(blob) == ["epoch", "lr_scheduler_state_dict", "model_state_dict", "optimizer_state_dict"]
    assert sorted(blob["model_state_dict"]) == GOLD["embeddings_trainable"]["checkpoint_keys"]
    # a DDP-written file carries "module." prefixes (train.py:303 strips them)
    blob["model_state_dict"] = {"module." + k: v for k, v in blob["model_state_dict"].items()}
    torch.save(blob, path)
    # resume into a fresh model with the OTHER optimizer implementation
    fresh, _ = tiny_cpu_flamingo(seed=1)                          # different init: everything trainable must come from the file
    fresh.lang_encoder.load_state_dict({k: v for k, v in model.lang_encoder.state_dict().items()
                                        if "gated_cross_attn" not in k}, strict=False)   # frozen LM = "pretrained" weights
    fresh.vision_encoder.load_state_dict(model.vision_encoder.state_dict())
    red2 = GradReducer(fresh, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt2 = step.build_optimizer(fresh, lr=1e-3)
    sched2 = torch.optim.lr_scheduler.LambdaLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0),
                                               lambda s: 1.0 / (1 + s))
    assert checkpoint.load_checkpoint(path, fresh, opt2, sched2) == 2
    assert sched2.last_epoch == sched.last_epoch
    assert torch.equal(_params(fresh), _params(model))
    step.train_step(model, red, opt, batch, info, amp=False)
    step.train_step(fresh, red2, opt2, batch, info, amp=False)
    assert (_params(fresh) - _params(model)).abs().max().item() < 2e-5
    with pytest.raises(KeyError):
        checkpoint.load_model_state(fresh, {"perceiver.no_such_parameter": torch.zeros(1)})


def test_lr_scheduler_drives_the_fused_step_epilogue():
    """The reference's train.py wraps its optimizer in a torch LR scheduler (get_constant_schedule_with_warmup etc. are
    LambdaLR).  FlatAdamW must be accepted by LambdaLR, and the lr the scheduler sets must be the lr the fused AdamW kernel
    is called with: warm-up factor 0 at step 0 -> parameters do not move; later steps move them."""
    import torch
    from open_flamingo_amd.train import step, synthetic
    from open_flamingo_amd.train.optim import FlatAdamW
    from open_flamingo_amd.train.reducer import GradReducer
    from tests.cpu_model import tiny_cpu_flamingo
    from tests.emu import harness as H
    model, info = tiny_cpu_flamingo(seed=0)
    red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt = FlatAdamW(red, lr=1e-3, ops=H.emu_ops())
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 2
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(1.0, s / 2))      # 0, 0.5, 1.0, ...
    seen = []
    ops = opt.ops
    orig = ops.adamw_clip
    ops.adamw_clip = lambda *a, **kw: (seen.append(kw["lr"]), orig(*a, **kw))[1]
    batch = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5)
    p = red.buckets[0]["params"][-1]
    before = p.detach().clone()
    step.train_step(model, red, opt, batch, info, amp=False, lr_scheduler=sched)
    assert set(seen) == {0.0} and torch.equal(p.detach(), before), "warm-up factor 0: no parameter may move"
    seen.clear()
    step.train_step(model, red, opt, batch, info, amp=False, lr_scheduler=sched)
    assert set(seen) == {5e-4} and not torch.equal(p.detach(), before)
    assert [g["lr"] for g in opt.param_groups] == [1e-3, 1e-3]
    ops.adamw_clip = orig
