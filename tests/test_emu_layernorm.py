"""LayerNorm fwd/bwd + element-wise helpers on the host SIMT emulator vs torch fp64."""
import numpy as np
import pytest
import torch

from tests.emu import harness as H


def _ln_ref(x, w, b):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


@pytest.mark.parametrize("x_f32", [1, 0])
# (70, 2048) / (67, 2560) / (66, 4096): the workgroup-per-row backward (dim >= 1536 with dw/db), both instantiations, ragged last
# workgroup; 64: 8-consecutive-columns lane map, 1024 / 2048 / 2560 / 4096: the split map of the wave kernels, 2048 / 4096: of the workgroup kernel
@pytest.mark.parametrize("rows,dim", [(37, 64), (9, 1024), (70, 2048), (67, 2560), (66, 4096)])
def test_layernorm_fwd_bwd(x_f32, rows, dim):
    g = torch.Generator().manual_seed(rows * dim)
    x = torch.randn(rows, dim, generator=g) * 2 + 0.5
    if not x_f32:
        x = x.to(torch.bfloat16)
    w = 1 + 0.1 * torch.randn(dim, generator=g)
    b = 0.1 * torch.randn(dim, generator=g)
    y = torch.zeros(rows, dim + 8, dtype=torch.bfloat16)  # strided destination
    stats = torch.zeros(rows, 2)
    L = H.lib()
    rc = L.of_layernorm_fwd(H.ptr(x), x_f32, x.stride(0), H.ptr(w), H.ptr(b), H.ptr(y), y.stride(0), H.ptr(stats),
                            rows, dim, None)
    assert rc == 0
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = _ln_ref(xd, wd, bd)
    np.testing.assert_allclose(y[:, :dim].double().numpy(), ref.detach().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(stats[:, 0].double().numpy(), x.double().mean(-1).numpy(), rtol=1e-5, atol=1e-6)
    # fp32 output variant
    y32 = torch.zeros(rows, dim)
    rc = L.of_layernorm_fwd_out(H.ptr(x), x_f32, x.stride(0), H.ptr(w), H.ptr(b), H.ptr(y32), 1, dim, H.ptr(stats),
                                rows, dim, None)
    assert rc == 0
    np.testing.assert_allclose(y32.double().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    # backward with residual add, bf16 copy and dw/db accumulation on top of existing values
    dy = torch.randn(rows, dim, generator=g).to(torch.bfloat16)
    resid = torch.randn(rows, dim, generator=g)
    if not x_f32:
        resid = resid.to(torch.bfloat16)
    ref.backward(dy.double())
    dx = torch.zeros_like(resid)
    dxb = torch.zeros(rows, dim, dtype=torch.bfloat16)
    dw0, db0 = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    dw, db = dw0.clone(), db0.clone()
    H.emu_ops().ln_bwd(dy, x, stats, w, resid=resid, dx=dx, dx_bf16=dxb, dw=dw, db=db)
    want_dx = xd.grad + resid.double()
    tol = dict(rtol=1e-4, atol=1e-4) if x_f32 else dict(rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(dx.double().numpy(), want_dx.numpy(), **tol)
    np.testing.assert_allclose(dxb.double().numpy(), want_dx.numpy(), rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose((dw - dw0).double().numpy(), wd.grad.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose((db - db0).double().numpy(), bd.grad.numpy(), rtol=1e-3, atol=1e-3)


def test_layernorm_fwd_many_rows_walks_rows_with_the_next_row_in_flight():
    """More rows than the chip holds waves (> 8192): a wave walks 2 / 4 consecutive rows with the next row's loads issued before the
    current row is reduced and stored (of_ln_fwd_kernel<CPL, true>); ragged last wave; fused residual add writing the sum IN PLACE."""
    ops = H.emu_ops()
    for rows, dim in ((8192 + 5, 16), (2 * 8192 + 7, 8)):
        g = torch.Generator().manual_seed(rows)
        x = torch.randn(rows, dim, generator=g) * 2 + 0.5
        add = torch.randn(rows, dim, generator=g).to(torch.bfloat16)
        w, b = 1 + 0.1 * torch.randn(dim, generator=g), 0.1 * torch.randn(dim, generator=g)
        y, st = torch.zeros(rows, dim, dtype=torch.bfloat16), torch.zeros(rows, 2)
        ops.ln_fwd(x, w, b, y, st)
        ref = _ln_ref(x.double(), w.double(), b.double())
        np.testing.assert_allclose(y.double().numpy(), ref.numpy(), rtol=1e-2, atol=2e-2)
        xs = x.clone()
        y2, st2 = torch.zeros(rows, dim, dtype=torch.bfloat16), torch.zeros(rows, 2)
        ops.ln_fwd_add(xs, add, xs, w, b, y2, st2)
        want = x + add.float()
        assert torch.equal(xs, want)
        np.testing.assert_allclose(y2.double().numpy(), _ln_ref(want.double(), w.double(), b.double()).numpy(), rtol=1e-2, atol=2e-2)
        np.testing.assert_allclose(st2[:, 0].double().numpy(), want.double().mean(-1).numpy(), rtol=1e-5, atol=1e-6)


def test_elementwise_helpers():
    L = H.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1003, generator=g)
    xb = torch.zeros(1003, dtype=torch.bfloat16)
    assert L.of_cast_f32_to_bf16(H.ptr(x), H.ptr(xb), 1003, None) == 0
    assert torch.equal(xb, x.to(torch.bfloat16))
    xf = torch.zeros(1003)
    assert L.of_cast_bf16_to_f32(H.ptr(xb), H.ptr(xf), 1003, None) == 0
    assert torch.equal(xf, xb.float())
    a, b = torch.randn(517, generator=g), torch.randn(517, generator=g)
    o = torch.zeros(517)
    assert L.of_add(H.ptr(a), H.ptr(b), H.ptr(o), 1, 517, None) == 0
    assert torch.equal(o, a + b)
    lat = torch.randn(4, 16, generator=g)
    out = torch.zeros(12, 24)
    assert L.of_broadcast_rows(H.ptr(lat), 4, H.ptr(out), 1, 24, 12, 16, None) == 0
    assert torch.equal(out[:, :16], lat.repeat(3, 1))
    dst0 = torch.randn(4, 16, generator=g)
    dst = dst0.clone()
    src = torch.randn(12, 16, generator=g)
    assert L.of_reduce_rows(H.ptr(src), 1, 12, 16, H.ptr(dst), 4, None) == 0
    np.testing.assert_allclose(dst.numpy(), (dst0 + src.reshape(3, 4, 16).sum(0)).numpy(), rtol=1e-6, atol=1e-6)


def test_layernorm_grouped_rows_and_second_gradient():
    """Perceiver addressing: LN output rows land inside a [N][v+n][D] buffer; backward reads dy from the same
    grouped rows and adds a second contiguous gradient (kv_input = cat(LN_m(x), LN_l(latents)))."""
    ops = H.emu_ops()
    g = torch.Generator().manual_seed(9)
    N, v, n, D = 3, 5, 4, 32
    lat = torch.randn(N * n, D, generator=g)
    w, b = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    buf = torch.zeros(N * (v + n), D, dtype=torch.bfloat16)
    y2 = torch.zeros(N * n, D, dtype=torch.bfloat16)
    st = torch.zeros(N * n, 2)
    ops.ln_fwd_grouped(lat, w, b, buf[v:], D, n, (v + n) * D, y2, st)
    ref = _ln_ref(lat.double(), w.double(), b.double()).reshape(N, n, D)
    got = buf.reshape(N, v + n, D)
    np.testing.assert_allclose(got[:, v:].double().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)
    assert (got[:, :v] == 0).all()
    assert torch.equal(y2.reshape(N, n, D), got[:, v:])
    dybuf = torch.randn(N * (v + n), D, generator=g).to(torch.bfloat16)
    dy2 = torch.randn(N * n, D, generator=g).to(torch.bfloat16)
    dx = torch.zeros(N * n, D)
    dw, db = torch.zeros(D), torch.zeros(D)
    ops.ln_bwd(dybuf[v:], lat, st, w, lddy=D, dy_grp_rows=n, dy_grp_stride=(v + n) * D, dy2=dy2, dx=dx, dw=dw, db=db)
    xd, wd, bd = lat.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    dyt = dybuf.reshape(N, v + n, D)[:, v:].reshape(N * n, D).double() + dy2.double()
    _ln_ref(xd, wd, bd).backward(dyt)
    np.testing.assert_allclose(dx.double().numpy(), xd.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dw.double().numpy(), wd.grad.numpy(), rtol=1e-3, atol=1e-3)


def test_step_epilogue_matches_torch_adamw_with_clipping():
    """of_sumsq_partial/finish + of_adamw_clip vs clip_grad_norm_ + torch.optim.AdamW over several steps, two parameter
    groups (weight decay 0.1 / 0.0) sharing one global norm, odd sizes (vector body + scalar tail)."""
    ops = H.emu_ops()
    g = torch.Generator().manual_seed(3)
    sizes, wds = [1027, 4096], [0.1, 0.0]
    ps = [torch.randn(n, generator=g) for n in sizes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.AdamW([{"params": [ref[0]], "weight_decay": wds[0]}, {"params": [ref[1]], "weight_decay": wds[1]}], lr=1e-2)
    ms = [torch.zeros(n) for n in sizes]
    vs = [torch.zeros(n) for n in sizes]
    for step in range(1, 4):
        grads = [torch.randn(n, generator=g) * (3.0 if step == 2 else 0.01) for n in sizes]   # step 2 clips, others do not
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt.step()
        acc = torch.zeros(1)
        gbuf = [gr.clone() * 4.0 for gr in grads]     # as if 4 ranks' gradients had been SUMMED: grad_scale = 1/4 undoes it
        ops.sumsq(gbuf, acc)
        again = torch.zeros(1)
        ops.sumsq(gbuf, again)
        assert torch.equal(acc, again)      # fixed summation order: bit-identical on every call (and so on every rank)
        np.testing.assert_allclose(float(acc), 16.0 * sum(float((gr.double() ** 2).sum()) for gr in grads), rtol=1e-5)
        bf = [torch.zeros(n, dtype=torch.bfloat16) for n in sizes]
        for p, gb, m, v, wd, b16 in zip(ps, gbuf, ms, vs, wds, bf):
            ops.adamw_clip(p, gb, m, v, acc, step=step, lr=1e-2, weight_decay=wd, max_norm=1.0, p_bf16=b16, zero_grad=True,
                           grad_scale=0.25)
            assert float(gb.abs().max()) == 0.0
        for p, r, b16 in zip(ps, ref, bf):
            np.testing.assert_allclose(p.numpy(), r.detach().numpy(), rtol=2e-5, atol=2e-6)
            assert torch.equal(b16, p.to(torch.bfloat16))


def test_step_epilogue_skips_the_update_on_a_non_finite_gradient_norm():
    """The device-side form of the reference's skip-on-NaN (train_utils.py:161-169): a NaN or Inf anywhere in the gradients makes
    the global norm non-finite and of_adamw_clip leaves parameters, moments and bf16 copies of EVERY buffer untouched (they all
    read the same norm), while still clearing the gradients it was asked to clear; the next finite step updates normally."""
    ops = H.emu_ops()
    g = torch.Generator().manual_seed(4)
    sizes = [1027, 4096]
    ps = [torch.randn(n, generator=g) for n in sizes]
    ms = [torch.rand(n, generator=g) * 0.1 for n in sizes]
    vs = [torch.rand(n, generator=g) * 0.1 for n in sizes]
    bf = [p.to(torch.bfloat16) for p in ps]
    for bad in (float("nan"), float("inf")):
        before = [(p.clone(), m.clone(), v.clone(), b.clone()) for p, m, v, b in zip(ps, ms, vs, bf)]
        grads = [torch.randn(n, generator=g) for n in sizes]
        grads[1][77] = bad                                  # one bad element in ONE buffer
        acc = torch.zeros(1)
        ops.sumsq(grads, acc)
        assert not torch.isfinite(acc).all()
        for p, gb, m, v, b16, zero in zip(ps, grads, ms, vs, bf, (True, False)):
            ops.adamw_clip(p, gb, m, v, acc, step=3, lr=1e-2, weight_decay=0.1, max_norm=1.0, p_bf16=b16, zero_grad=zero)
        for (p0, m0, v0, b0), p, m, v, b16 in zip(before, ps, ms, vs, bf):
            assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0) and torch.equal(b16, b0)
        assert float(grads[0].abs().max()) == 0.0           # cleared as asked
        assert not torch.isfinite(grads[1]).all()           # left for the next backward to overwrite
    grads = [torch.randn(n, generator=g) * 0.01 for n in sizes]
    acc = torch.zeros(1)
    ops.sumsq(grads, acc)
    p0 = ps[0].clone()
    ops.adamw_clip(ps[0], grads[0], ms[0], vs[0], acc, step=4, lr=1e-2, weight_decay=0.1, max_norm=1.0, p_bf16=bf[0], zero_grad=True)
    assert not torch.equal(ps[0], p0) and torch.isfinite(ps[0]).all()


def test_gelu_and_mixed_add_helpers_of_the_frozen_mlp():
    """of_gelu_fwd / of_gelu_bwd / of_add_bf16 (frozen MPT MLP) vs torch on the same bf16 inputs; odd length
    (vector body + scalar tail); of_gelu_bwd in place."""
    ops = H.emu_ops()
    g = torch.Generator().manual_seed(8)
    n = 8 * 300 + 5
    x = (torch.randn(n, generator=g) * 2).to(torch.bfloat16)
    dy = torch.randn(n, generator=g).to(torch.bfloat16)
    y = ops.gelu_fwd(x)
    np.testing.assert_allclose(y.float().numpy(), torch.nn.functional.gelu(x.float()).numpy(), rtol=1e-2, atol=1e-2)
    want = torch.ops.aten.gelu_backward(dy.float(), x.float(), approximate="none")
    buf = dy.clone()
    out = ops.gelu_bwd(buf, x, out=buf)
    assert out.data_ptr() == buf.data_ptr()
    np.testing.assert_allclose(out.float().numpy(), want.numpy(), rtol=1e-2, atol=1e-2)
    a = torch.randn(n, generator=g)
    s = ops.add_bf16(a, dy)
    assert torch.equal(s, a + dy.float())


def test_quick_gelu():
    L = H.lib()
    x = (torch.randn(1003, generator=torch.Generator().manual_seed(2)) * 3).to(torch.bfloat16)
    y = torch.zeros_like(x)
    assert L.of_quick_gelu(H.ptr(x), H.ptr(y), x.numel(), None) == 0
    want = (x.float() * torch.sigmoid(1.702 * x.float())).to(torch.bfloat16)
    np.testing.assert_allclose(y.float().numpy(), want.float().numpy(), rtol=1e-2, atol=1e-3)


def test_fused_cross_entropy_matches_torch():
    """of_ce_fwd / of_ce_bwd (csrc/loss.hip) vs F.cross_entropy on the fp32 upcast of the same bf16 logits: odd vocab (rows
    start at 2-byte alignment), ignored rows, both dtypes."""
    import torch.nn.functional as F
    from tests.emu import harness as H
    ops = H.emu_ops()
    torch.manual_seed(0)
    for dt in (torch.bfloat16, torch.float32):
        rows, V = 37, 2051 + 300
        logits = (torch.randn(rows, V) * 3).to(dt)
        labels = torch.randint(0, V, (rows,))
        labels[[3, 11, 36]] = -100
        lse, loss_rows = torch.empty(rows), torch.empty(rows)
        ops.ce_fwd(logits, labels, lse, loss_rows)
        xf = logits.float().requires_grad_(True)
        want = F.cross_entropy(xf, labels, ignore_index=-100)
        n = (labels != -100).sum()
        assert abs(float(loss_rows.sum() / n) - float(want)) < 1e-5 * abs(float(want))
        assert torch.allclose(lse, torch.logsumexp(logits.float(), -1), rtol=1e-6, atol=1e-5)
        (want * 0.7).backward()
        d = torch.full_like(logits, float("nan"))
        ops.ce_bwd(logits, labels, lse, torch.tensor([0.7 / float(n)]), d)
        tol = 1e-6 if dt == torch.float32 else 4e-3
        assert (d.float() - xf.grad).abs().max() <= tol * xf.grad.abs().max() + 1e-9
        assert (d[[3, 11, 36]] == 0).all()


def test_narrow_step_epilogue_launches_give_the_same_bits():
    """of_sumsq_partial_w / of_adamw_clip_w (ABI v8): a few fat workgroups instead of a grid that covers the chip (so that the step
    epilogue leaves whole CUs to the prefetched vision tower on its side stream).  Every partial slot is summed by the same 256
    virtual threads in the same order, AdamW is element-wise: partial slots, parameters, moments and bf16 copies are bit-identical
    to the plain launches -- sizes with a vector body and a scalar tail, fewer elements than slots, and several slots per workgroup."""
    ops = H.emu_ops()
    g = torch.Generator().manual_seed(5)
    for n in (1027, 70001, 300):
        grad = torch.randn(n, generator=g)
        wide, narrow = torch.full((ops.SUMSQ_PARTS,), -1.0), torch.full((ops.SUMSQ_PARTS,), -2.0)
        ops.sumsq_partial(grad, wide)
        for cus in (3, 96):
            ops.sumsq_partial(grad, narrow, max_workgroups=cus)
            assert torch.equal(wide, narrow), (n, cus)
        acc = torch.zeros(1)
        ops.sumsq_finish(wide, acc)
        p0, m0, v0 = torch.randn(n, generator=g), torch.rand(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.1
        outs = []
        for cus in (0, 2, 96):
            p, gb, m, v, b16 = p0.clone(), grad.clone(), m0.clone(), v0.clone(), torch.zeros(n, dtype=torch.bfloat16)
            ops.adamw_clip(p, gb, m, v, acc, step=2, lr=1e-2, weight_decay=0.1, max_norm=1.0, p_bf16=b16, zero_grad=True, max_workgroups=cus)
            assert float(gb.abs().max()) == 0.0
            outs.append((p, m, v, b16))
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
