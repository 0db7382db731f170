"""train/sparse_rows.py (opt-in): the two trained embedding rows' gradient from two autograd taps must
equal what the reference's dense-then-mask procedure (train_utils.py:174-196) keeps -- lookup part + tied-head part --
while every other gradient, the forward values, generate(), checkpoints and the optimizer-state numbering stay as they
are.  CPU: tiny Flamingo with the oracle's hot-path modules; the fused step epilogue's kernels run on the emulator."""
import pytest
import torch

from open_flamingo_amd.train import checkpoint, sparse_rows, step, synthetic
from open_flamingo_amd.train.optim import FlatAdamW
from open_flamingo_amd.train.reducer import GradReducer
from tests.cpu_model import tiny_cpu_flamingo
from tests.emu import harness as H


def _rows(info):
    return [info["media_token_id"], info["eoc_token_id"]]


def _grads(sparse, amp):
    model, info = tiny_cpu_flamingo(seed=0)
    state = sparse_rows.enable(model, _rows(info)) if sparse else None
    losses = []
    for b in (synthetic.make_batch(2, 1, 16, info, "cpu", seed=6), synthetic.make_batch(2, 2, 24, info, "cpu", seed=5)):
        loss = step.forward_loss(model, b, info, amp=amp)
        loss.backward()
        losses.append(float(loss))
    table = model.lang_encoder.get_input_embeddings().weight
    rows = torch.tensor(_rows(info))
    kept = state.grad_rows() if sparse else table.grad.index_select(0, rows)
    others = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad and p is not table}
    return losses, kept.detach().clone(), others, (state.dense_grad() if sparse else None), table


@pytest.mark.parametrize("amp", [False, True])
def test_tapped_rows_equal_the_masked_dense_gradient(amp):
    l0, kept0, others0, _, table = _grads(False, amp)
    l1, kept1, others1, dense1, _ = _grads(True, amp)
    assert l0 == l1                                           # forward values untouched
    tol = 2e-2 if amp else 1e-5                               # amp: the dense path rounds dW to bf16 before the fp32 sum
    assert (kept1 - kept0).abs().max().item() <= tol * kept0.abs().max().item()
    assert kept0.abs().max().item() > 0
    assert set(others0) == set(others1)
    for k in others0:
        assert torch.equal(others0[k], others1[k]), k         # every other gradient: bit-identical
    assert dense1.shape == table.shape and int((dense1.abs().sum(-1) > 0).sum()) == 2


def test_training_generate_and_checkpoints_are_unchanged():
    finals = []
    for sparse in (False, True):
        model, info = tiny_cpu_flamingo(seed=0)
        if sparse:
            sparse_rows.enable(model, _rows(info))
            with pytest.raises(RuntimeError):
                step.build_optimizer(model)                   # a torch optimizer would silently stop training the rows
        red = GradReducer(model, embedding_rows=_rows(info))
        opt = FlatAdamW(red, lr=1e-3, ops=H.emu_ops())
        b1 = synthetic.make_batch(2, 1, 16, info, "cpu", seed=6)
        b2 = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5)
        for _ in range(2):
            step.train_step(model, red, opt, b2, info, batch_laion=b1, amp=False)
        table = model.lang_encoder.get_input_embeddings().weight
        model.eval()
        with torch.no_grad():
            gen = model.generate(b2["vision_x"][:1], b2["lang_x"][:1, :8], attention_mask=b2["attention_mask"][:1, :8],
                                 max_new_tokens=5, do_sample=False)
        sd = opt.state_dict()
        finals.append(dict(params={k: p.detach().clone() for k, p in model.named_parameters()},
                           keys=sorted(checkpoint.trainable_state_dict(model)), gen=gen, opt=sd,
                           moved=(table.detach()[_rows(info)]).clone()))
    d, s = finals
    assert d["keys"] == s["keys"]
    assert torch.equal(d["gen"], s["gen"])
    for k in d["params"]:
        assert (d["params"][k] - s["params"][k]).abs().max().item() <= 1e-6, k
    assert [g["params"] for g in d["opt"]["param_groups"]] == [g["params"] for g in s["opt"]["param_groups"]]
    for i in d["opt"]["state"]:
        for name in ("exp_avg", "exp_avg_sq"):
            a, b = d["opt"]["state"][i][name], s["opt"]["state"][i][name]
            assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item() + 1e-12


def test_disable_restores_the_dense_path():
    model, info = tiny_cpu_flamingo(seed=0)
    table = model.lang_encoder.get_input_embeddings().weight
    sparse_rows.enable(model, _rows(info))
    assert not table.requires_grad and sparse_rows.is_trainable(table)
    sparse_rows.disable(model)
    assert table.requires_grad and not hasattr(model, "_of_sparse_rows") and not hasattr(table, "_of_trained_rows")
    step.forward_loss(model, synthetic.make_batch(2, 2, 24, info, "cpu", seed=5), info, amp=False).backward()
    assert table.grad is not None and table.grad.shape == table.shape


def test_enable_refuses_heads_it_cannot_tap():
    """Remote-code MPT variants (reference factory.py loads them with trust_remote_code): get_output_embeddings() is None
    (logits = F.linear(x, wte.weight) inline) or the embedding module itself -- the tied-head part of the kept rows'
    gradient cannot be tapped there, so enable() must raise and leave the model untouched."""
    import pytest
    from open_flamingo_amd.train import sparse_rows
    from tests.cpu_model import tiny_cpu_flamingo
    model, info = tiny_cpu_flamingo(seed=0)
    lm = model.lang_encoder
    rows = [info["media_token_id"], info["eoc_token_id"]]
    table = lm.get_input_embeddings().weight
    orig = lm.get_output_embeddings
    for fake in (lambda: None, lambda: lm.get_input_embeddings()):
        lm.get_output_embeddings = fake
        with pytest.raises(NotImplementedError):
            sparse_rows.enable(model, rows)
        assert table.requires_grad and not hasattr(model, "_of_sparse_rows")
    lm.get_output_embeddings = orig
    sparse_rows.enable(model, rows)
    assert not table.requires_grad and len(model._of_sparse_rows.handles) == 2
