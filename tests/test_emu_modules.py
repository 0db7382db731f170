"""The PRODUCT host code of the hot path (open_flamingo_amd/src/helpers.py: nn.Modules, autograd Functions, in-place
gradient sinks, inference path; train/reducer.py + train/optim.py around them) executed on CPU with every libofhip
kernel running on the host SIMT emulator (tests/emu) -- TEST INFRASTRUCTURE ONLY: the two guards that make the product
refuse anything but an AMD GPU (`_require_hip`, `Ops.default`) are monkeypatched for the duration of a test.

What this pins without a GPU: the module-level wiring (parameter order, saved tensors, returned gradients), gradient
accumulation straight into GradReducer buckets, the no-grad decode path with its projected-media cache, and a whole
train_step (libofhip modules + fused step epilogue) against the oracle modules + torch AdamW from the same weights."""
import pytest
import torch

from oracle import flamingo_oracle as O
from open_flamingo_amd.hip.ops import Ops
from open_flamingo_amd.src import helpers
from open_flamingo_amd.train import step, synthetic, towers
from open_flamingo_amd.train.optim import FlatAdamW
from open_flamingo_amd.train.reducer import GradReducer
from tests.cpu_model import swap_in_oracle
from tests.emu import harness as H


@pytest.fixture
def on_emulator(monkeypatch):
    monkeypatch.setattr(helpers, "_require_hip", lambda t, what: None)
    monkeypatch.setattr(Ops, "default", staticmethod(H.emu_ops))
    helpers._shared.items.clear()
    yield
    helpers._shared.items.clear()


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-12)


def _tiny(seed=0):
    # 56-pixel images = 16 patch tokens per image and a 2-layer Perceiver: the kernels run on the host emulator, the
    # reference's 256 patches / depth 6 would only repeat the same code paths (minutes instead of seconds)
    model, info = towers.build_flamingo("OF-tiny", device="cpu", seed=seed, gates=0.5, fused_lm_attention=False,
                                        vision_kw=dict(width=64, layers=2, heads=2, patch=14, image=56), perceiver_depth=2)
    model.train()
    return model, info


def test_block_module_matches_oracle_module(on_emulator):
    torch.manual_seed(0)
    blk = helpers.GatedCrossAttentionBlock(dim=128, dim_visual=64, heads=2)
    ref = O.OracleGatedCrossAttentionBlock(dim=128, dim_visual=64, heads=2)
    with torch.no_grad():
        blk.attn_gate.fill_(0.6)
        blk.ff_gate.fill_(-0.4)
    ref.load_state_dict(blk.state_dict(), strict=True)
    x = torch.randn(2, 20, 128)
    media = torch.randn(2, 2, 64, 64)
    locs = torch.zeros(2, 20, dtype=torch.bool)
    locs[:, 1] = locs[0, 9] = True
    w = torch.randn(2, 20, 128)
    outs = []
    for m in (blk, ref):
        xi, mi = x.clone().requires_grad_(True), media.clone().requires_grad_(True)
        kw = dict(quant=O.bf16_round) if m is ref else {}
        y = m(xi, mi, media_locations=locs, **kw)
        (y * w).sum().backward()
        outs.append((y.detach(), xi.grad, mi.grad, {k: p.grad for k, p in m.named_parameters()}))
    (y, dx, dm, g), (y0, dx0, dm0, g0) = outs
    assert _rel(y, y0) < 1e-2 and _rel(dx, dx0) < 3e-2 and _rel(dm, dm0) < 3e-2
    for k in g0:
        assert _rel(g[k], g0[k]) < 3e-2, k
    # no-grad path (eval mode: weight copies are cached): same bits as the training forward, projected media kept for as
    # long as the media tensor is the same
    blk.eval()
    with torch.no_grad():
        y_inf = blk(x, media, media_locations=locs)
        kv = blk._kv_cache[3]
        tok = blk(x[:, :1], media, media_locations=locs, use_cached_media=True)
        assert blk._kv_cache[3] is kv
    assert torch.equal(y_inf, y) and tok.shape == (2, 1, 128)
    blk.train()
    assert "_kv_cache" not in blk.__dict__


def test_gradients_accumulate_in_place_into_reducer_buckets(on_emulator):
    """Two backward passes with a GradReducer (libofhip backward adds into the bucket views, returns None to autograd and
    fires the reducer's callback itself) == plain autograd accumulation without a reducer."""
    got = []
    for use_reducer in (False, True):
        model, info = _tiny()
        red = GradReducer(model, embedding_rows=None) if use_reducer else None
        fired = []
        if red is not None:
            for b in red.buckets:
                for p in b["params"]:
                    p._of_on_grad = (lambda q, f=p._of_on_grad: (fired.append(q), f(q))[1])
        for b in (synthetic.make_batch(2, 1, 16, info, "cpu", seed=6, image_size=56), synthetic.make_batch(2, 2, 24, info, "cpu", seed=5, image_size=56)):
            step.forward_loss(model, b, info, amp=False).backward()
        if red is not None:
            n_params = sum(len(b["params"]) for b in red.buckets)
            assert len(fired) == 2 * n_params
            for b in red.buckets:
                for p, off in zip(b["params"], b["offsets"]):
                    assert p.grad.data_ptr() == b["flat"].data_ptr() + 4 * off
        got.append({k: p.grad.detach().clone() for k, p in model.named_parameters()
                    if p.requires_grad and ("gated_cross_attn" in k or "perceiver" in k)})
    for k in got[0]:
        assert _rel(got[1][k], got[0][k]) < 1e-5, k      # same kernels, same order: only fp32 add order of the two passes


def test_train_step_with_product_modules_tracks_oracle_modules(on_emulator):
    """libofhip modules + GradReducer + fused step epilogue (all kernels emulated) vs the oracle's autograd modules +
    clip_grad_norm_ + torch AdamW, same initial weights and batch, two optimizer steps.  The product rounds GEMM operands
    to bf16 and the oracle model here runs plain fp32, and Adam turns small gradient differences into O(lr) parameter
    differences: trajectories are compared in L2 against the distance travelled (25 %), losses to 0.5 %."""
    batch = None
    runs = []
    for product in (True, False):
        model, info = _tiny()
        if not product:
            swap_in_oracle(model)
        red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
        opt = FlatAdamW(red, lr=1e-3, ops=H.emu_ops()) if product else step.build_optimizer(model, lr=1e-3)
        batch = batch or synthetic.make_batch(2, 2, 24, info, "cpu", seed=5, image_size=56)
        init = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
        losses = [float(step.train_step(model, red, opt, batch, info, amp=False)) for _ in range(2)]
        runs.append((losses, init, {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}))
    (l1, i1, p1), (l0, i0, p0) = runs
    assert all(abs(a - b) <= 5e-3 * abs(b) for a, b in zip(l1, l0)), (l1, l0)
    assert l0[-1] < l0[0]
    # product parameter names differ from the oracle model's only by module class, not by key
    for k in p0:
        travelled = (p0[k] - i0[k]).norm().item()
        assert (p1[k] - p0[k]).norm().item() <= 0.25 * travelled + 1e-7, k


def test_submodules_run_stand_alone_like_the_reference(on_emulator):
    """Reference code may call ``block.ff(x)``, ``block.attn(x, media, ...)`` or a Perceiver layer's attention directly
    (helpers.py:15-22, 39-65, 160-233): the product submodules run the same libofhip kernels without the fused
    residual / gate, forward and backward, against the oracle functions with the same parameters."""
    torch.manual_seed(0)
    w_ = lambda *s: torch.randn(*s)
    # ---- FeedForward
    ff = helpers.FeedForward(64, mult=4)
    x = w_(2, 10, 64)
    xo = x.clone().requires_grad_(True)
    P = {k: v for k, v in ff.named_parameters()}
    yo = O.feed_forward(xo, P, quant=O.bf16_round)
    wt = w_(2, 10, 64)
    (yo * wt).sum().backward()
    ref = (yo.detach(), xo.grad, {k: p.grad.clone() for k, p in P.items()})
    ff.zero_grad()
    xi = x.clone().requires_grad_(True)
    y = ff(xi)
    (y * wt).sum().backward()
    assert _rel(y.detach(), ref[0]) < 1e-2 and _rel(xi.grad, ref[1]) < 3e-2
    for k, p in ff.named_parameters():
        assert _rel(p.grad, ref[2][k]) < 3e-2, k
    # ---- MaskedCrossAttention (dim_head 64 and 128: the kernels' sizes; 32 and 80: zero-padded heads, the true softmax scale)
    for dh in (64, 128, 32, 80):
        att = helpers.MaskedCrossAttention(dim=64, dim_visual=32, dim_head=dh, heads=2)
        media = w_(2, 2, 64, 32)
        locs = torch.zeros(2, 10, dtype=torch.bool)
        locs[:, 1] = locs[0, 6] = True
        P = {k: v for k, v in att.named_parameters()}
        xo, mo = x.clone().requires_grad_(True), media.clone().requires_grad_(True)
        yo = O.masked_cross_attention(xo, mo, locs, P, heads=2, quant=O.bf16_round)
        (yo * wt).sum().backward()
        ref = (yo.detach(), xo.grad, mo.grad, {k: p.grad.clone() for k, p in P.items()})
        att.zero_grad()
        xi, mi = x.clone().requires_grad_(True), media.clone().requires_grad_(True)
        y = att(xi, mi, media_locations=locs)
        (y * wt).sum().backward()
        assert _rel(y.detach(), ref[0]) < 1e-2 and _rel(xi.grad, ref[1]) < 3e-2 and _rel(mi.grad, ref[2]) < 3e-2, dh
        for k, p in att.named_parameters():
            assert _rel(p.grad, ref[3][k]) < 3e-2, (dh, k)
    # ---- PerceiverAttention (dim_head 64; 48: zero-padded heads)
    for dh in (64, 48):
        pa = helpers.PerceiverAttention(dim=64, dim_head=dh, heads=2)
        feats, lat = w_(1, 2, 24, 64), w_(1, 2, 16, 64)
        P = {k: v for k, v in pa.named_parameters()}
        fo, lo = feats.clone().requires_grad_(True), lat.clone().requires_grad_(True)
        yo = O.perceiver_attention(fo, lo, P, heads=2, quant=O.bf16_round)
        wl = w_(1, 2, 16, 64)
        (yo * wl).sum().backward()
        ref = (yo.detach(), fo.grad, lo.grad, {k: p.grad.clone() for k, p in P.items()})
        pa.zero_grad()
        fi, li = feats.clone().requires_grad_(True), lat.clone().requires_grad_(True)
        y = pa(fi, li)
        (y * wl).sum().backward()
        assert _rel(y.detach(), ref[0]) < 1e-2 and _rel(fi.grad, ref[1]) < 3e-2 and _rel(li.grad, ref[2]) < 3e-2, dh
        for k, p in pa.named_parameters():
            assert _rel(p.grad, ref[3][k]) < 3e-2, (dh, k)
    # a whole block with dim_head = 128 (the reference accepts any dim_head; the kernels exist for 64 and 128)
    blk = helpers.GatedCrossAttentionBlock(dim=64, dim_visual=32, dim_head=128, heads=2)
    refb = O.OracleGatedCrossAttentionBlock(dim=64, dim_visual=32, dim_head=128, heads=2)
    with torch.no_grad():
        blk.attn_gate.fill_(0.5)
        blk.ff_gate.fill_(0.5)
    refb.load_state_dict(blk.state_dict(), strict=True)
    yb = blk(x, media, media_locations=locs)
    assert _rel(yb.detach(), refb(x, media, media_locations=locs, quant=O.bf16_round).detach()) < 1e-2
    # ... and with dim_head = 80 (OF-4B's language model uses that head size; the reference's helpers accept it): forward, every
    # gradient, the no-grad path, and a PerceiverResampler with dim_head = 32
    blk = helpers.GatedCrossAttentionBlock(dim=64, dim_visual=32, dim_head=80, heads=2)
    refb = O.OracleGatedCrossAttentionBlock(dim=64, dim_visual=32, dim_head=80, heads=2)
    with torch.no_grad():
        blk.attn_gate.fill_(0.5)
        blk.ff_gate.fill_(0.5)
    refb.load_state_dict(blk.state_dict(), strict=True)
    xo, mo = x.clone().requires_grad_(True), media.clone().requires_grad_(True)
    yo = refb(xo, mo, media_locations=locs, quant=O.bf16_round)
    (yo * wt).sum().backward()
    xi, mi = x.clone().requires_grad_(True), media.clone().requires_grad_(True)
    yb = blk(xi, mi, media_locations=locs)
    (yb * wt).sum().backward()
    assert _rel(yb.detach(), yo.detach()) < 1e-2 and _rel(xi.grad, xo.grad) < 3e-2 and _rel(mi.grad, mo.grad) < 3e-2
    for (k, p), (_, q) in zip(blk.named_parameters(), refb.named_parameters()):
        assert p.grad.shape == q.grad.shape and _rel(p.grad, q.grad) < 3e-2 + 1e-3, k
    with torch.no_grad():
        assert _rel(blk(x, media, media_locations=locs), yo.detach()) < 1e-2
    pr = helpers.PerceiverResampler(dim=64, depth=2, dim_head=32, heads=2, num_latents=16)
    pro = O.OraclePerceiverResampler(dim=64, depth=2, dim_head=32, heads=2, num_latents=16)
    pro.load_state_dict(pr.state_dict(), strict=True)
    feats = w_(1, 2, 1, 24, 64)
    wl = w_(1, 2, 16, 64)
    yo = pro(feats, quant=O.bf16_round)
    (yo * wl).sum().backward()
    yp = pr(feats)
    (yp * wl).sum().backward()
    assert _rel(yp.detach(), yo.detach()) < 1e-2
    for (k, p), (_, q) in zip(pr.named_parameters(), pro.named_parameters()):
        assert _rel(p.grad, q.grad) < 3e-2 + 1e-3, k


def test_step_epilogue_leaves_weight_gradients_for_the_backward_to_overwrite(on_emulator):
    """CPU twin of the GPU test of the same name: after a fused step the nn.Linear weight gradients are stale and marked
    fresh, everything else is cleared; two accumulated backward passes on top of the stale content equal the same passes
    on cleared buffers; a step with no backward in between clears the stale matrices instead of re-applying them."""
    model, info = _tiny()
    red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt = FlatAdamW(red, lr=1e-3, ops=H.emu_ops())
    b1 = synthetic.make_batch(2, 1, 16, info, "cpu", seed=6, image_size=56)
    b2 = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5, image_size=56)
    step.train_step(model, red, opt, b2, info, amp=False)
    mats = [p for b in red.buckets for p in b["overwritable"]]
    small = [p for b in red.buckets for p in b["params"] if all(p is not q for q in b["overwritable"])]
    assert mats and small and all(p._of_grad_fresh for p in mats)
    assert any(float(p.grad.abs().sum()) > 0 for p in mats) and all(float(p.grad.abs().sum()) == 0 for p in small)
    for b in (b1, b2):
        step.forward_loss(model, b, info, amp=False).backward()
    assert not any(p._of_grad_fresh for p in mats)
    got = [p.grad.detach().clone() for p in mats + small]
    red.zero_grad()
    for b in (b1, b2):
        step.forward_loss(model, b, info, amp=False).backward()
    for g, p in zip(got, mats + small):
        assert _rel(g, p.grad) < 1e-5
    red.zero_grad()
    step.train_step(model, red, opt, b2, info, amp=False)
    m_before = [opt._moments_of(p)[0].clone() for p in mats]
    opt.step()
    for p, m0 in zip(mats, m_before):
        assert float(p.grad.abs().sum()) == 0
        assert torch.allclose(opt._moments_of(p)[0], 0.9 * m0, rtol=1e-5, atol=1e-12)


def test_early_norm_partials_change_no_bit_and_are_voided_by_later_gradient_writes(on_emulator):
    """A bucket's share of the global gradient norm is computed when its gradient becomes final (the reducer's callback), not in
    step(): same kernel, same values, same slots -- the slots hold exactly what the late form computes from the finished buckets.
    A gradient written into a bucket AFTER its partial sums were taken (a second backward before the step, with or without
    no_sync) voids them: step() recomputes.  (Parameters of two separate runs are not compared bit for bit here: the emulator's
    workgroups run on parallel host threads and the LayerNorm column sums add in arrival order.)"""
    ops = H.emu_ops()
    P = ops.SUMSQ_PARTS
    model, info = _tiny()
    red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt = FlatAdamW(red, lr=1e-3, ops=ops)
    assert not opt.early_norm                   # opt-in (measured +0.5 ms per step on one GPU)
    opt.early_norm = True
    b1 = synthetic.make_batch(2, 1, 16, info, "cpu", seed=6, image_size=56)
    b2 = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5, image_size=56)
    step.train_step(model, red, opt, b2, info, amp=False)
    assert opt.early_partials_used == len(red.buckets)
    # the slots against the late form, on the finished buckets of another backward
    step.forward_loss(model, b1, info, amp=False).backward()
    red.finish(average=False)
    assert all(b["early_gen"] == red.generation for b in red.buckets)
    for i, b in enumerate(red.buckets):
        late = torch.empty(P)
        ops.sumsq_partial(b["flat"], late)
        assert torch.equal(late, opt._parts[i * P:(i + 1) * P]), i
    # one more backward under no_sync voids them: step() computes every share itself
    with red.no_sync():
        step.forward_loss(model, b2, info, amp=False).backward()
    assert all(b.get("early_gen") is None for b in red.buckets)
    red.finish(average=False)
    want = torch.zeros(1)
    ops.sumsq([b["flat"] for b in red.buckets] + [model.lang_encoder.get_input_embeddings().weight.grad.index_select(
        0, torch.as_tensor([info["media_token_id"], info["eoc_token_id"]])).contiguous()], want)
    opt.step()
    assert opt.early_partials_used == 0
    assert torch.equal(opt._sumsq, want)
    red.zero_grad(flat_already_zero=True)
    opt.early_norm = False                      # off again: the reducer's callback is gone
    assert red.on_bucket_final is None


def test_nan_loss_skips_the_step_on_the_device(on_emulator, monkeypatch):
    """train_step(nan_check="device"): no host-side isnan; a NaN loss makes every gradient NaN, the fused step epilogue sees a
    non-finite global norm and leaves every trainable parameter (and AdamW moment) as it was -- the reference's skip
    (train_utils.py:161-169) without its host sync; the following finite step trains normally."""
    model, info = _tiny()
    red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt = FlatAdamW(red, lr=1e-3, ops=H.emu_ops())
    batch = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5, image_size=56)
    step.train_step(model, red, opt, batch, info, amp=False, nan_check="device")
    snap = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
    moments = [b["m"].clone() for b in red.buckets]
    real = step.forward_loss
    monkeypatch.setattr(step, "forward_loss", lambda *a, **kw: real(*a, **kw) * float("nan"))
    loss = step.train_step(model, red, opt, batch, info, amp=False, nan_check="device")
    monkeypatch.setattr(step, "forward_loss", real)
    assert loss is not None and torch.isnan(loss)
    for k, p in model.named_parameters():
        if p.requires_grad:
            assert torch.equal(p.detach(), snap[k]), k
    assert all(torch.equal(b["m"], m0) for b, m0 in zip(red.buckets, moments))
    # Adam's step is counted on the device and only for applied updates: the skipped step does not advance the bias
    # correction (the reference `continue`s before optimizer.step()), and a checkpoint records the applied count
    assert opt.step_count == 2 and opt.applied_steps() == 1
    loss = step.train_step(model, red, opt, batch, info, amp=False, nan_check="device")
    assert torch.isfinite(loss)
    assert opt.step_count == 3 and opt.applied_steps() == 2
    assert {float(st["step"]) for st in opt.state_dict()["state"].values()} == {2.0}
    moved = [k for k, p in model.named_parameters() if p.requires_grad and not torch.equal(p.detach(), snap[k])]
    assert len(moved) > 10 and all(torch.isfinite(p).all() for p in model.parameters())


def test_grouped_media_projections_match_per_block_projections(on_emulator, monkeypatch):
    """with Flamingo.group_media_projections the to_kv of every gated block runs as ONE grouped GEMM
    right after the Perceiver and the media gradient of all blocks as ONE K-grouped GEMM; loss and every gradient must
    equal the per-block form (same kernels per element, only the launch structure differs)."""
    monkeypatch.setattr(helpers, "can_group_media", lambda media: True)
    from open_flamingo_amd.train.towers import FAMILY
    monkeypatch.setitem(FAMILY, "OF-tiny", dict(FAMILY["OF-tiny"], every=1))     # 4 gated blocks
    results = []
    for grouped in (False, True):
        model, info = towers.build_flamingo("OF-tiny", device="cpu", seed=0, gates=0.5, fused_lm_attention=False,
                                            vision_kw=dict(width=256, layers=1, heads=2, patch=14, image=56), perceiver_depth=2)
        model.train()
        model.group_media_projections = grouped
        batch = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5, image_size=56)
        calls = []
        orig = Ops.gemm_grouped
        monkeypatch.setattr(Ops, "gemm_grouped", lambda self, *a, **kw: (calls.append(kw["kind"]), orig(self, *a, **kw))[1])
        loss = step.forward_loss(model, batch, info, amp=False)
        loss.backward()
        monkeypatch.setattr(Ops, "gemm_grouped", orig)
        assert calls == ([1, 2] if grouped else []), calls      # one grouped projection forward, one grouped media gradient
        results.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()
                                      if p.requires_grad and p.grad is not None}))
        assert not helpers._media_groups        # dropped with the conditioning at the end of Flamingo.forward
    (l0, g0), (l1, g1) = results
    assert abs(l0 - l1) <= 2e-5 * abs(l0), (l0, l1)      # other GEMM kernel (fp32 summation order) before the bf16 rounding of k|v
    assert g0.keys() == g1.keys()
    for k in g0:
        # Perceiver grads: dmedia summed in fp32 once vs 4 bf16->fp32 partial sums.  The (1,)-shaped gate gradients of this
        # random-upstream loss are nearly cancelling sums (|value| ~ 3e-3 of their term mass): 5e-2 for those
        assert _rel(g1[k], g0[k]) < (5e-2 if k.endswith("_gate") else 2e-2), (k, _rel(g1[k], g0[k]))


def test_bf16_twins_travel_between_backwards_and_change_nothing(on_emulator, monkeypatch):
    """Every backward on the fp32 stream ends in a LayerNorm backward that also emits the bf16 copy of its dx and offers it
    (hip/path.py: offer_bf16_twin); the next backward down the stream takes it instead of running a cast pass.  Gated blocks and
    fused frozen MPT blocks alternate in the LM: with the hand-off on, the stream-sized casts of a backward disappear (all but the
    first consumer's) and every gradient is bit-identical to the run with the hand-off off."""
    from open_flamingo_amd.hip import path as P
    from open_flamingo_amd.train import frozen_blocks

    def run(handoff):
        model, info = _tiny()
        lm = model.lang_encoder
        for mod in lm.modules():      # what towers.hold_frozen_linears_in_bf16 does, for the LM blocks only (no autocast on this path)
            if isinstance(mod, torch.nn.Linear) and mod is not lm.get_output_embeddings() and not mod.weight.requires_grad:
                mod.weight.data = mod.weight.data.to(torch.bfloat16)
        assert frozen_blocks.use_fused_frozen_mpt_blocks(lm, allow_cpu=True) > 0
        if not handoff:
            monkeypatch.setattr(P, "offer_bf16_twin", lambda t, twin, scope=None: None)
        casts, taken = [], []
        ops = H.emu_ops()
        orig_cast, orig_take = type(ops).to_bf16, P.take_bf16_twin
        monkeypatch.setattr(type(ops), "to_bf16", lambda self, x, out=None: (casts.append(tuple(x.shape)), orig_cast(self, x, out))[1])
        monkeypatch.setattr(P, "take_bf16_twin", lambda t, scope=None: (lambda r: (taken.append(r is not None), r)[1])(orig_take(t, scope)))
        red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
        batch = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5, image_size=56)
        opt = FlatAdamW(red, lr=1e-3, ops=ops)
        loss = float(step.train_step(model, red, opt, batch, info, amp=False))
        monkeypatch.undo()
        monkeypatch.setattr(helpers, "_require_hip", lambda t, what: None)
        monkeypatch.setattr(Ops, "default", staticmethod(H.emu_ops))
        return loss, {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}, casts, sum(taken)

    l1, p1, casts1, taken1 = run(True)
    l0, p0, casts0, taken0 = run(False)
    assert taken0 == 0 and taken1 >= 2, (taken0, taken1)
    assert len(casts1) <= len(casts0) - taken1, (len(casts1), len(casts0), taken1)
    assert l1 == l0
    # the GEMM-made gradients of the gated blocks (deterministic on the emulator; LayerNorm dw / db are summed by atomics in thread
    # order there) -> bit-identical parameters after the step
    keys = [k for k in p0 if "gated_cross_attn_layer." in k and k.endswith(".weight") and "norm" not in k and ".ff.0." not in k]
    assert len(keys) >= 8
    for k in keys:
        assert torch.equal(p1[k], p0[k]), k


@pytest.mark.parametrize("case", __import__("tests.path_checks", fromlist=["DH64_CASES"]).DH64_CASES)
def test_block_module_against_reference_goldens_at_dim_head_64(on_emulator, case):
    """the cached-media branch (helpers.py:175-178,199-205; T_txt != mask length, media_locations = None)
    of the PRODUCT module against answers produced by the REAL reference at dim_head 64, by the one 8c rule.  The same check runs
    on the GPU in tests/test_gpu_path.py."""
    import os
    from tests import path_checks as PC
    rep = PC.check_block_module_against_dh64_golden(case, "cpu", os.path.join(os.path.dirname(__file__), "golden"))
    assert rep["y"]["hip_rel_l2"] > 0.0


def test_vision_prefetch_changes_no_bit_and_runs_the_tower_once_per_forward(on_emulator, monkeypatch):
    """train_step(next_vision_x=...): the next step's frozen vision-tower forward is enqueued from inside this
    step's backward (Flamingo.schedule_vision_prefetch -> prefetch_vision; on the GPU: a side stream).  Same arithmetic -> the losses of
    two steps on alternating batches (the second consumes prefetched tokens) are bit-identical with and without it; the tower runs exactly once per forward either way;
    a forward with ANOTHER tensor (or a changed one) ignores the prefetched tokens."""
    def run(prefetch):
        model, info = _tiny()
        red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
        opt = FlatAdamW(red, lr=1e-3, ops=H.emu_ops())
        batches = [synthetic.make_batch(2, 2, 24, info, "cpu", seed=5 + i, image_size=56) for i in range(2)]
        calls = []
        vis = model.vision_encoder
        orig = vis.forward
        monkeypatch.setattr(vis, "forward", lambda x: (calls.append(tuple(x.shape)), orig(x))[1])
        losses = []
        for i in range(2):
            nxt = batches[(i + 1) % 2]["vision_x"] if prefetch else None
            losses.append(float(step.train_step(model, red, opt, batches[i % 2], info, amp=False, next_vision_x=nxt)))
        return losses, len(calls), model, info, batches

    # round 6: the tower forward is enqueued from INSIDE the backward (Flamingo.schedule_vision_prefetch: at the start of the backward of
    # gated block n_blocks // 12, or where the gradient of the Perceiver's output is complete), not behind it; behind it with the switch off
    # -- the same losses either way (the enqueue point moves, the arithmetic does not)
    from open_flamingo_amd.src.flamingo import Flamingo
    where = []
    orig_prefetch = Flamingo.prefetch_vision
    monkeypatch.setattr(Flamingo, "prefetch_vision",
                        lambda self, v, amp_dtype=None: (where.append(torch._C._current_graph_task_id() != -1), orig_prefetch(self, v, amp_dtype=amp_dtype))[1])
    l1, n1, model, info, batches = run(True)
    assert where == [True, True]
    l0, n0, *_ = run(False)
    assert l1 == l0, (l1, l0)
    monkeypatch.setattr(step, "PREFETCH_IN_BACKWARD", False)
    where.clear()
    l3, *_ = run(True)
    assert where == [False, False] and l3 == l0
    monkeypatch.setattr(step, "PREFETCH_IN_BACKWARD", True)
    monkeypatch.setattr(Flamingo, "prefetch_vision", orig_prefetch)
    assert n0 == 2 and n1 == 3            # two forwards; with prefetch one more tower run is waiting for a third step
    assert "_of_vision_prefetch" in model.__dict__
    # the waiting tokens belong to batches[1]'s tensor: a forward on another tensor must not take them ...
    other = batches[0]["vision_x"].clone()
    assert model._take_prefetched_vision(other) is None and "_of_vision_prefetch" not in model.__dict__
    # ... nor one on the right tensor after an in-place change
    model.prefetch_vision(batches[1]["vision_x"])
    batches[1]["vision_x"].add_(1.0)
    assert model._take_prefetched_vision(batches[1]["vision_x"]) is None


def test_padded_heads_do_not_add_to_gradients_the_step_epilogue_left_stale(on_emulator):
    """dim_head outside {64, 128}: the attention projections reach the kernels as zero-padded differentiable views, so their gradients
    arrive through autograd's AccumulateGrad (which ADDS) instead of the overwrite-with-beta-0 protocol of train/optim.py -- a
    gradient marked `_of_grad_fresh` (stale content, to be overwritten) is cleared in the forward."""
    torch.manual_seed(1)
    blk = helpers.GatedCrossAttentionBlock(dim=64, dim_visual=32, dim_head=80, heads=2)
    with torch.no_grad():
        blk.attn_gate.fill_(0.5)
        blk.ff_gate.fill_(0.5)
    x, media = torch.randn(2, 10, 64), torch.randn(2, 2, 64, 32)
    locs = torch.zeros(2, 10, dtype=torch.bool)
    locs[:, 1] = True
    blk(x, media, media_locations=locs).square().sum().backward()
    want = {k: p.grad.clone() for k, p in blk.named_parameters()}
    for k, p in blk.named_parameters():
        if k.endswith(("to_q.weight", "to_kv.weight", "to_out.weight")):
            p.grad.fill_(7.0)                   # what an optimizer step leaves behind
            p._of_grad_fresh = True
        else:
            p.grad.zero_()
    blk(x, media, media_locations=locs).square().sum().backward()
    for k, p in blk.named_parameters():
        assert torch.allclose(p.grad, want[k], rtol=1e-5, atol=1e-6), k
        assert not getattr(p, "_of_grad_fresh", False)


def test_fused_attention_branch_inside_a_train_step_and_its_packed_weights_follow_the_optimizer(on_emulator, monkeypatch):
    """OF-tiny's gated blocks (d 256, 8 heads) take the fused attention branch (csrc/xattn_fused.hip) once the text length is a multiple
    of 32.  Two optimizer steps with it and with the separate launches: the same losses to fp32 summation order -- which also proves
    that the fragment-major weight copies the step epilogue re-packs (FlatAdamW.packed_view) are the CURRENT weights (stale copies
    would replay step 0's projections) -- and, bit for bit, packed_view = of_pack_frag16(bf16_view) after the last step."""
    from open_flamingo_amd.hip import path as P
    ops = H.emu_ops()
    calls = []
    orig = Ops.xattn_fused_fwd
    monkeypatch.setattr(Ops, "xattn_fused_fwd", lambda self, *a, **kw: (calls.append(kw.get("probe_only", False)), orig(self, *a, **kw))[1])
    losses = []
    for fused in (True, False):
        monkeypatch.setattr(P, "FUSED_XATTN", fused)
        model, info = _tiny()
        red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
        opt = FlatAdamW(red, lr=3e-3, ops=ops)
        batch = synthetic.make_batch(2, 2, 32, info, "cpu", seed=5, image_size=56)
        n0 = len(calls)
        losses.append([float(step.train_step(model, red, opt, batch, info, amp=False)) for _ in range(2)])
        if fused:
            assert sum(1 for c in calls[n0:] if not c) == 2 * 2, calls        # two gated blocks, two forwards, one launch each
            for blk in model.lang_encoder.gated_cross_attn_layers:
                if blk is None:
                    continue
                for lin in (blk.attn.to_q, blk.attn.to_out):
                    pk, bf = opt.packed_view(lin.weight), opt.bf16_view(lin.weight)
                    assert pk is not None and torch.equal(bf, lin.weight.detach().to(torch.bfloat16))
                    assert torch.equal(pk, ops.pack_frag16(bf))
        else:
            assert len(calls) == n0
    assert losses[0][0] > losses[0][1]
    assert all(abs(a - b) <= 2e-4 * abs(b) for a, b in zip(*losses)), losses
