"""End-to-end boundary parity against the REAL reference Flamingo (golden: tests/golden/tiny_flamingo.npz, made by
tests/golden/make_golden.py::tiny_flamingo with the reference's flamingo.py / flamingo_lm.py / helpers.py).

CPU (not gpu): our Flamingo / FlamingoLMMixin / FlamingoLayer host logic with the oracle's hot-path modules plugged
in must reproduce the reference loss, gradients, greedy generate() tokens and cached-media logits, and expose the
exact same state_dict key set (the golden was produced by a strict=True load of OUR state_dict into the reference).
GPU: the same model with the libofhip modules (the product) must match within bf16 tolerance."""
import os

import numpy as np
import pytest
import torch

from open_flamingo_amd.train import synthetic
from tests.cpu_model import tiny_cpu_flamingo

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_flamingo.npz")


def _run(model, info, device, amp=False):
    z = np.load(GOLD)
    batch = synthetic.make_batch(2, 2, 24, info, device, seed=5)
    labels = synthetic.make_labels(batch["lang_x"], info["media_token_id"], info["eoc_token_id"], info["pad_token_id"])
    model.train()
    out = model(vision_x=batch["vision_x"], lang_x=batch["lang_x"], attention_mask=batch["attention_mask"], labels=labels)
    out[0].backward()
    sd = model.state_dict(keep_vars=True)
    model.eval()
    with torch.no_grad():
        gen = model.generate(batch["vision_x"][:1], batch["lang_x"][:1, :8], attention_mask=batch["attention_mask"][:1, :8],
                             max_new_tokens=6, do_sample=False)
        model.cache_media(input_ids=batch["lang_x"][:, :12], vision_x=batch["vision_x"])
        cached = model(vision_x=None, lang_x=batch["lang_x"][:, 12:16], attention_mask=batch["attention_mask"][:, 12:16],
                       clear_conditioned_layers=False).logits
        model.uncache_media()
    return z, out, sd, gen, cached


def test_cpu_boundary_matches_reference_flamingo():
    model, info = tiny_cpu_flamingo(seed=0)
    z, out, sd, gen, cached = _run(model, info, "cpu")
    assert sorted(sd.keys()) == list(z["state_dict_keys"])
    np.testing.assert_allclose(float(out[0]), float(z["loss"]), rtol=1e-5)
    np.testing.assert_allclose(out.logits[:, :4, :32].detach().numpy(), z["logits_head"], rtol=1e-3, atol=1e-4)
    assert np.array_equal(gen.numpy(), z["generated"])
    np.testing.assert_allclose(cached[:, :, :32].numpy(), z["cached_logits_head"], rtol=1e-3, atol=1e-4)
    for k in z.files:
        if k.startswith("grad."):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), z[k], rtol=2e-3, atol=1e-6, err_msg=k)
        elif k.startswith("gradnorm."):
            np.testing.assert_allclose(float(sd[k[9:]].grad.norm()), float(z[k]), rtol=2e-3, err_msg=k)


@pytest.mark.gpu
def test_gpu_boundary_matches_reference_flamingo():
    """The product: libofhip-backed modules inside our Flamingo, fp32 residual stream, bf16 MFMA operands.
    Tolerances: loss 1e-2 relative; gradient tensors 5e-2 of their max-abs; the (1,)-shaped tanh-gate gradients
    PER GATE: 5e-2 of the gate's own value + a floor of 2e-2 of the largest gate gradient (they are whole-tensor reductions
    sum(dy * branch) that can cancel almost completely -- golden: layer-1 ff_gate 1.2e-4 next to layer-3 ff_gate 2.7e-2 -- so a
    pure relative rule would measure the cancellation of the small ones; round 3 held every gate to 5e-2 of the LARGEST, which
    would not have noticed a wrong small gate gradient; the sums themselves are bit-reproducible since round 3.  The floor is what
    the erf approximation alone can move such a sum by: its ~5e-7 SYSTEMATIC error over 49 k terms shifted layer-1's ff_gate by
    3.6e-4 = 1 % of the largest gate when round 4 changed the epilogues' erf formula, profiles/r04i_gputests_first.log).  The tight rule
    for gate gradients (relative L2 <= 2e-2 on a conditioned loss) lives in tests/path_checks.py::judge_8c;
    greedy tokens may legitimately differ once logits are within bf16 noise, so only the first generated token and
    >= 50 % agreement are required here (token equality up to ties: test_gpu_greedy_tokens_equal_reference_up_to_ties)."""
    from open_flamingo_amd.train import towers
    model, info = towers.build_flamingo("OF-tiny", device="cpu", seed=0, gates=0.5)   # CPU RNG = the golden's weights
    model.cuda()
    z, out, sd, gen, cached = _run(model, info, "cuda")
    assert sorted(sd.keys()) == list(z["state_dict_keys"])
    assert abs(float(out[0]) - float(z["loss"])) <= 1e-2 * abs(float(z["loss"]))
    got = out.logits[:, :4, :32].detach().float().cpu().numpy()
    assert np.abs(got - z["logits_head"]).max() <= 5e-2 * np.abs(z["logits_head"]).max()
    gate_scale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad.") and k.endswith("_gate"))
    for k in z.files:
        if k.startswith("grad."):
            g = sd[k[5:]].grad.float().cpu().numpy()
            if k.endswith("_gate"):
                err = float(np.abs(g - z[k]).max())
                print(k, f"reference {float(z[k].reshape(-1)[0]):+.3e}  error {err:.1e}")
                assert err <= 5e-2 * float(np.abs(z[k]).max()) + 2e-2 * gate_scale, (k, g, z[k])
            else:
                assert np.abs(g - z[k]).max() <= 5e-2 * np.abs(z[k]).max() + 1e-7, k
        elif k.startswith("gradnorm.") and k.endswith("_gate"):
            assert abs(float(sd[k[9:]].grad.norm()) - float(z[k])) <= 5e-2 * float(z[k]) + 2e-2 * gate_scale, k
        elif k.startswith("gradnorm."):
            assert abs(float(sd[k[9:]].grad.norm()) - float(z[k])) <= 5e-2 * float(z[k]) + 1e-7, k
    gen = gen.cpu().numpy()
    assert gen.shape == z["generated"].shape and gen[0, 8] == z["generated"][0, 8]
    assert (gen == z["generated"]).mean() >= 0.5
    c = cached[:, :, :32].float().cpu().numpy()
    assert np.abs(c - z["cached_logits_head"]).max() <= 5e-2 * np.abs(z["cached_logits_head"]).max()


GEN_GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_flamingo_generate.npz")


def _greedy_against_reference(model, info, device, tie):
    """Greedy decoding step by step (each step a full forward, like the golden generator) for the two prompts of
    tests/golden/tiny_flamingo_generate.npz.  Every token must EQUAL the real reference's until the first step at which
    the reference's own top-1 / top-2 logit margin is below ``tie`` (there a bf16 implementation may legitimately pick
    the runner-up, and the continuations diverge).  Returns (#tokens compared, #tokens in the golden)."""
    z = np.load(GEN_GOLD)
    batch = synthetic.make_batch(2, 2, 24, info, device, seed=5)
    steps, pos, compared, total = int(z["steps"]), 0, 0, 0
    model.eval()
    with torch.no_grad():
        for prompt_len in z["prompt_lens"].tolist():
            ids = batch["lang_x"][:1, :prompt_len]
            alive = True
            for i in range(steps):
                want, margin = int(z["tokens"][pos + i]), float(z["margins"][pos + i])
                total += 1
                if not alive:
                    continue
                out = model(vision_x=batch["vision_x"][:1], lang_x=ids, attention_mask=torch.ones_like(ids))
                got = int(out.logits[0, -1].float().argmax())
                if got != want:
                    assert margin < tie, (f"prompt {prompt_len} step {i}: token {got} != reference {want} although the "
                                          f"reference's margin {margin:.4f} is not a tie (< {tie})")
                    alive = False
                    continue
                compared += 1
                ids = torch.cat([ids, torch.tensor([[want]], device=ids.device)], dim=1)
            pos += steps + 1
    return compared, total


def test_cpu_greedy_tokens_equal_reference():
    model, info = tiny_cpu_flamingo(seed=0)
    compared, total = _greedy_against_reference(model, info, "cpu", tie=1e-4)     # fp32 oracle modules: no ties at all
    assert compared == total == 20


@pytest.mark.gpu
def test_gpu_greedy_tokens_equal_reference_up_to_ties():
    """SURVEY 8c KAT-4 "generate token-equality for greedy decoding" for the product (bf16 MFMA operands): equality is
    required wherever the reference's margin exceeds bf16 noise -- 0.03 = a tenth of the logits' standard deviation
    (0.32) and ~5x the observed bf16 logit error of this model."""
    from open_flamingo_amd.train import towers
    model, info = towers.build_flamingo("OF-tiny", device="cpu", seed=0, gates=0.5)
    model.cuda()
    compared, total = _greedy_against_reference(model, info, "cuda", tie=3e-2)
    assert compared >= 8, (compared, total)          # both prompts start with margins of 0.15 / 0.13
