"""GPU parity of the individual libofhip kernels (through the C ABI, product loader) on a real MI355X.

* hardware-semantics probes: the transposed-fragment path (ds_read_b64_tr_b16) must be bit-identical to the
  scalar-LDS path (safe=1) -- this is what validates the lane maps the CPU emulator assumes;
* GEMM all layouts / epilogues vs torch fp32 matmul of the same bf16 operands (tolerance = fp32 accumulation
  order + one bf16 output rounding: 1e-2 relative to max-abs);
* attention fwd/bwd vs the fp64 dense restatement (tests/attn_reference.py), tolerance 2e-2 of max-abs
  (P and dS are rounded to bf16 for the MFMA, like the reference's autocast does);
* LayerNorm fwd/bwd vs torch fp64.
"""
import numpy as np
import pytest
import torch

from open_flamingo_amd.hip import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tests.gpu_ops import routed_ops
    return routed_ops()          # the product library; launches that FORCE a kernel (safe >= 2) go to the tools build of the same sources


def _r(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def _rel(got, want):
    return (got.double() - want.double()).abs().max().item() / (want.double().abs().max().item() + 1e-12)


@pytest.mark.parametrize("ta,tb", [(False, True), (True, True)])
def test_probe_transpose_read_matches_scalar_path(ops, ta, tb):
    M, N, K = 256, 384, 200
    A = _r((K, M) if ta else (M, K), 1)
    B = _r((K, N), 2)
    outs = []
    for safe in (0, 1):
        o = torch.zeros(M, N, device="cuda")
        ops.gemm(A, B, o, ta=ta, tb=tb, epi=abi.EPI_ACC_F32, safe=safe)
        outs.append(o)
    ref = (A.float().t() if ta else A.float()) @ B.float()
    assert _rel(outs[1], ref) < 1e-4, "scalar-LDS path wrong: MFMA lane map assumption broken"
    assert torch.equal(outs[0], outs[1]), f"tr-read path differs from scalar path (rel {_rel(outs[0], ref):.3e})"


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (8192, 512, 2048), (2048, 8192, 4096)])
def test_gemm_layouts(ops, ta, tb, M, N, K):
    if ta:
        M = (M + 7) // 8 * 8
    if not ta and K % 8:
        pytest.skip("K-contiguous operand needs K % 8 == 0")
    A = _r((K, M) if ta else (M, K), 3)
    B = _r((K, N) if tb else (N, K), 4)
    ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
    o32 = torch.zeros(M, N, device="cuda")
    ops.gemm(A, B, o32, ta=ta, tb=tb, epi=abi.EPI_ACC_F32)
    assert _rel(o32, ref) < 2e-4
    ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, ob, ta=ta, tb=tb, epi=abi.EPI_STORE_BF16)
    assert _rel(ob, ref) < 1e-2


def test_gemm_epilogues(ops):
    M, N, K = 520, 1024, 256
    A, B = _r((M, K), 5), _r((N, K), 6, 0.1)
    acc = A.float() @ B.float().t()
    gate = torch.tensor([0.37], device="cuda")
    g = float(torch.tanh(gate))
    b_out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    a_out = torch.zeros_like(b_out)
    ops.gemm(A, B, b_out, epi=abi.EPI_GELU, out2=a_out)
    assert _rel(a_out, acc) < 1e-2 and _rel(b_out, torch.nn.functional.gelu(acc)) < 1e-2
    res = torch.randn(M, N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    ops.gemm(A, B, out, epi=abi.EPI_GATE_RESID, aux=res, gate=gate)
    assert _rel(out, res + g * acc) < 2e-4
    resb = res.to(torch.bfloat16)
    outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, outb, epi=abi.EPI_GATE_RESID, aux=resb, gate=gate)
    assert _rel(outb, resb.float() + g * acc) < 1e-2
    c = torch.randn(M, N, device="cuda")
    c0 = c.clone()
    ops.gemm(A, B, c, epi=abi.EPI_ACC_F32, alpha=0.5, beta=1.0, gate=gate)
    assert _rel(c, c0 + 0.5 * g * acc) < 2e-4
    # dX layouts with the gate-gradient dot epilogues
    W = _r((K, N), 7, 0.2)
    acc2 = A.float() @ W.float()
    aux = _r((M, N), 8)
    for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        o = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        dot = torch.zeros(1, device="cuda")
        ops.gemm(A, W, o, tb=True, epi=epi, aux=aux, gate=gate, dot=dot)
        x = aux.double()
        if epi == abi.EPI_DGELU_DOT:
            xx = x.clone().requires_grad_(True)
            torch.nn.functional.gelu(xx).sum().backward()
            want, wdot = g * acc2.double() * xx.grad, (1 - g * g) * (torch.nn.functional.gelu(x) * acc2.double()).sum()
        else:
            want, wdot = g * acc2.double(), (1 - g * g) * (x * acc2.double()).sum()
        assert _rel(o, want) < 1e-2
        assert abs(float(dot) - float(wdot)) <= 2e-3 * abs(float(wdot)) + 5e-2


def _attn_case(ops, q, k, v, heads, tt=None, n=0, T=0, only_imm=True, safe=0):
    from tests.attn_reference import dense_attention
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    dout = _r(tuple(q.shape), 99)
    o = torch.full_like(q, float("nan"))
    lse = torch.full((B, heads, Lq), float("nan"), device="cuda")
    tt32 = tt.to(torch.int32).cuda().contiguous() if tt is not None else None
    q2, k2, v2, o2, do2 = (t.reshape(-1, t.shape[-1]) for t in (q, k, v, o, dout))
    kw = dict(batch=B, Lq=Lq, Lk=Lk, heads=heads, text_time=tt32, n_per_media=n, T_img=T, only_immediate=only_imm,
              safe=safe)
    ops.attn_fwd(q2, k2, v2, o2, lse, **kw)
    dq = torch.full_like(q2, float("nan"))
    dk = torch.full((B * Lk, heads * 64), float("nan"), device="cuda", dtype=torch.bfloat16)
    dv = torch.full_like(dk, float("nan"))
    delta = torch.zeros(B, heads, Lq, device="cuda")
    ops.attn_bwd(q2, k2, v2, o2, lse, do2, dq, dk, dv, delta, **kw)
    qd, kd, vd = (t.double().cpu().requires_grad_(True) for t in (q, k, v))
    ref = dense_attention(qd, kd, vd, heads, tt, n, T, only_imm)
    ref.backward(dout.double().cpu())
    res = {}
    for name, got, want in (("o", o2.reshape(q.shape), ref.detach()), ("dq", dq.reshape(q.shape), qd.grad),
                            ("dk", dk.reshape(k.shape[0], Lk, -1), kd.grad), ("dv", dv.reshape(k.shape[0], Lk, -1), vd.grad)):
        assert torch.isfinite(got).all(), name
        res[name] = _rel(got.cpu(), want)
        assert res[name] < 2e-2, f"{name}: {res}"
    return o2.reshape(q.shape), lse, res


@pytest.mark.parametrize("dh,heads,B,L,causal", [(128, 16, 8, 256, True), (64, 16, 8, 257, False), (64, 4, 3, 300, True)])
def test_resident_forward_matches_tiled_forward_at_tower_shapes(ops, dh, heads, B, L, causal):
    """of_attn_fwd's resident-K/V form (of_attn_fwd_res_kernel: LDS-DMA, progressive loading for causal self-attention, a
    32-row tail block) at the shapes it is selected for -- frozen MPT blocks (256 x 256, head dim 128, causal + ALiBi) and the
    CLIP tower (257 x 257, head dim 64) -- against the tiled kernel (bit for bit) and a dense fp32 softmax."""
    d = heads * dh
    qkv = _r((B * L, 3 * d), 71)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / heads) for i in range(heads)], device="cuda") if causal else None
    outs = []
    for safe in (0, 3, 2):
        o = torch.full((B * L, d), float("nan"), device="cuda", dtype=torch.bfloat16)
        lse = torch.full((B, heads, L), float("nan"), device="cuda")
        ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, batch=B, Lq=L, Lk=L, heads=heads, scale=dh ** -0.5,
                     head_dim=dh, causal=causal, alibi_slopes=slopes, safe=safe)
        outs.append((o, lse))
    for o, lse in outs[:2]:
        assert torch.equal(o, outs[2][0]) and torch.equal(lse, outs[2][1])
    q, k, v = (qkv[:, i * d:(i + 1) * d].float().view(B, L, heads, dh).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) * dh ** -0.5
    if causal:
        pos = torch.arange(L, device="cuda")
        s = s + slopes.view(1, heads, 1, 1) * (pos.view(1, 1, 1, L) - pos.view(1, 1, L, 1))
        s = s.masked_fill(pos.view(1, 1, 1, L) > pos.view(1, 1, L, 1), float("-inf"))
    want = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * L, d)
    assert _rel(outs[0][0], want) < 1e-2
    assert torch.allclose(outs[0][1], s.logsumexp(-1), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("safe", [0, 1])
def test_attention_perceiver_shape(ops, safe):
    kv = _r((4, 320, 2 * 512), 11)
    q = _r((4, 64, 512), 12)
    _attn_case(ops, q, kv[..., :512], kv[..., 512:], 8, safe=safe)


@pytest.mark.parametrize("only_imm", [True, False])
def test_attention_masked_cases(ops, only_imm):
    B, L, T, n, heads = 3, 200, 3, 64, 8
    ml = torch.zeros(B, L, dtype=torch.bool)
    ml[0, [0, 60, 130]] = True
    ml[1, [5, 6, 7, 8, 150]] = True      # consecutive <image>, more tokens than images -> uniform rows
    ml[2, [L - 1]] = True                # everything before the only image -> zero rows
    tt = ml.cumsum(-1)
    q, k, v = _r((B, L, heads * 64), 13), _r((B, T * n, heads * 64), 14), _r((B, T * n, heads * 64), 15)
    o, lse, _ = _attn_case(ops, q, k, v, heads, tt, n, T, only_imm)
    if only_imm:
        zero = (tt == 0).cuda()
        assert (o[zero] == 0).all()


@pytest.mark.parametrize("L", [256, 2048])
def test_attention_five_image_windows_at_cfg5_shapes(ops, L):
    """BASELINE config 5's cross-attention core (B = 8, T = 5, n = 64: Lk = 320; L = 256 and the long-context 2048) with the
    mask quirks at five images -- text_time > T (every key masked: uniform rows), text_time == 0 (zero rows), unused images,
    consecutive <image> tokens -- element-wise against the fp64 dense restatement of helpers.py:192-231
    (tests/attn_reference.py), forward and all three gradients."""
    from tests.test_gpu_path import _t5_media_locations
    B, T, n, heads = 8, 5, 64, 8
    ml = _t5_media_locations(B, L)
    tt = ml.cumsum(-1)
    assert int(tt.max()) > T and bool((tt == 0).any()) and int(tt[3].max()) == 2
    q, k, v = _r((B, L, heads * 64), 61), _r((B, T * n, heads * 64), 62), _r((B, T * n, heads * 64), 63)
    o, lse, res = _attn_case(ops, q, k, v, heads, tt, n, T, True)
    print(res)
    assert (o[(tt == 0).cuda()] == 0).all()
    # uniform rows: the plain mean of ALL T*n values (helpers.py:218-221 with every key at -finfo.max)
    uni = (tt > T).cuda()
    want = v.float().view(B, T * n, heads * 64).mean(1, keepdim=True).expand(B, L, heads * 64)
    assert torch.allclose(o.float()[uni], want[uni], atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("x_f32", [1, 0])
def test_layernorm(ops, x_f32):
    rows, dim = 1000, 2048
    x = _r((rows, dim), 21, 2.0, torch.float32 if x_f32 else torch.bfloat16) + 0.5
    w, b = _r((dim,), 22, 0.1, torch.float32) + 1, _r((dim,), 23, 0.1, torch.float32)
    y = torch.zeros(rows, dim, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(rows, 2, device="cuda")
    ops.ln_fwd(x, w, b, y, stats)
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (dim,), wd, bd, 1e-5)
    assert _rel(y, ref.detach()) < 1e-2
    dy = _r((rows, dim), 24)
    resid = _r((rows, dim), 25, 1.0, x.dtype)
    ref.backward(dy.double())
    dx = torch.zeros_like(resid)
    dxb = torch.zeros(rows, dim, device="cuda", dtype=torch.bfloat16)
    dw, db = torch.zeros(dim, device="cuda"), torch.zeros(dim, device="cuda")
    ops.ln_bwd(dy, x, stats, w, resid=resid, dx=dx, dx_bf16=dxb, dw=dw, db=db)
    want = xd.grad + resid.double()
    assert _rel(dx, want) < (1e-4 if x_f32 else 2e-2)
    assert _rel(dxb, want) < 2e-2
    assert _rel(dw, wd.grad) < 1e-3 and _rel(db, bd.grad) < 1e-3


@pytest.mark.parametrize("big", [4, 6, 7, 16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_pingpong_race_screen(ops, ta, tb, big):
    """The 256x256 kernels (safe=4: 8-wave ping-pong, LDS-DMA slots ordered only by counted vmcnt + segment barriers;
    safe=6: 4-wave register-staged, one barrier per stage): the CPU emulator executes a workgroup's waves cooperatively
    and cannot see a race.  Screen on hardware: many shapes (1..64 stages, single and
    multi wave-of-blocks grids), repeated launches under load, results must be bit-identical to the general kernel
    (same products, same k order) every time."""
    torch.manual_seed(0)
    for (M, N, K) in [(256, 256, 64), (256, 256, 128), (256, 512, 192), (512, 256, 128), (2048, 2048, 2048),
                      (8192, 2048, 512), (4096, 8192, 1024)]:
        A = _r((K, M) if ta else (M, K), M + K)
        B = _r((K, N) if tb else (N, K), N + K)
        want = torch.zeros(M, N, device="cuda")
        ops.gemm(A, B, want, ta=ta, tb=tb, epi=abi.EPI_ACC_F32, safe=2)
        ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
        assert _rel(want, ref) < 2e-4
        for it in range(6):
            got = torch.full((M, N), float("nan"), device="cuda")
            ops.gemm(A, B, got, ta=ta, tb=tb, epi=abi.EPI_ACC_F32, safe=big)
            assert torch.equal(got, want), f"{(M, N, K)} iteration {it}: max diff {(got - want).abs().max().item()}"


def test_big_tile_epilogues_with_the_third_image_of_b_race_screen(ops):
    """gemm_w4m.hip since round 5: every epilogue but the *_DOT ones runs its K loop with a third LDS image of B (B of stage d + 2
    requested in phase 0 of stage d, vmcnt(8) in front of the barrier) and GATE_RESID requests its first residual tile in the epilogue,
    into the idle ring.  The emulator executes a DMA where it is issued and cannot see a wait that is one piece short: on hardware, 1 ..
    16 stages (every prologue / tail form of the three images), one tile and a multi-round grid, GELU with both outputs and gate + residual
    on the fp32 and on the bf16 stream -- against fp32 torch, and six launches one bit pattern."""
    torch.manual_seed(0)
    gate = torch.tensor([0.43], device="cuda")
    g = float(torch.tanh(gate))
    for (M, N) in [(256, 256), (2048, 4096)]:
        for K in (64, 128, 192, 256, 320, 1024):
            A, B = _r((M, K), M + K), _r((N, K), N + K + 1, K ** -0.5)
            acc = A.float() @ B.float().t()
            res = torch.randn(M, N, device="cuda")
            resb = res.to(torch.bfloat16)
            first = None
            for it in range(6):
                y = torch.full((M, N), float("nan"), device="cuda")
                ops.gemm(A, B, y, epi=abi.EPI_GATE_RESID, aux=res, gate=gate, safe=16)
                yb = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                ops.gemm(A, B, yb, epi=abi.EPI_GATE_RESID, aux=resb, gate=gate, safe=16)
                b_out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                a_out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                ops.gemm(A, B, b_out, epi=abi.EPI_GELU, out2=a_out, safe=16)
                got = (y, yb, b_out, a_out)
                if first is None:
                    first = got
                    assert _rel(y, res + g * acc) < 1e-4, (M, N, K)
                    assert _rel(yb, resb.float() + g * acc) < 1e-2, (M, N, K)
                    assert _rel(a_out, acc) < 1e-2 and _rel(b_out, torch.nn.functional.gelu(acc)) < 1e-2, (M, N, K)
                else:
                    assert all(torch.equal(a, b) for a, b in zip(got, first)), f"{(M, N, K)} launch {it}"


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_k_rotation_gives_the_same_product(ops, ta, tb):
    """Launches of_gemm selects itself (safe = 0) run the 256x256 kernel with its K loop rotated per XCD (gemm_w4m.hip: workgroups of
    XCD x start at stage x * stages / 8 and wrap around): the same products in another order of the fp32 additions.  Against the same
    kernel in plain stage order (safe = 16): equal up to the summation order, not bit for bit (so the rotation is really on); launch
    after launch the same bits; 8 stages (the first K with a non-zero rotation), a K that 8 does not divide, a multi-round grid."""
    torch.manual_seed(0)
    differs = False
    for (M, N, K) in [(4096, 8192, 512), (2048, 4096, 576), (8192, 2048, 8192), (8192, 8192, 2048)]:
        A = _r((K, M) if ta else (M, K), M + K)
        B = _r((K, N) if tb else (N, K), N + K)
        assert ops.kernel_label(M, N, K, ta, tb, abi.EPI_ACC_F32) == "w4m256"
        want = torch.zeros(M, N, device="cuda")
        ops.gemm(A, B, want, ta=ta, tb=tb, epi=abi.EPI_ACC_F32, safe=16)
        first = None
        for it in range(4):
            got = torch.full((M, N), float("nan"), device="cuda")
            ops.gemm(A, B, got, ta=ta, tb=tb, epi=abi.EPI_ACC_F32)
            if first is None:
                first = got
                assert _rel(got, want) < 1e-5, (M, N, K)
                differs = differs or not torch.equal(got, want)
            else:
                assert torch.equal(got, first), f"{(M, N, K)} launch {it}: max diff {(got - first).abs().max().item()}"
    assert differs


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_mid_kernel_race_screen(ops, ta, tb):
    """The 8-wave LDS-DMA 128x128 kernel (safe = 5; four-slot ring ordered by counted vmcnt + one barrier per stage): the CPU
    emulator cannot see a race.  Screen on hardware: 1..128 stages, single-tile to multi-wave grids, repeated launches, results
    bit-identical to the general kernel's up to the k order inside a stage (compared in fp32 with a summation-order tolerance)
    and identical across repeats."""
    torch.manual_seed(0)
    for (M, N, K) in [(128, 128, 64), (128, 256, 128), (256, 128, 192), (384, 256, 256), (8192, 512, 2048), (2048, 512, 8192),
                      (4096, 1024, 4096), (1024, 1024, 4096), (20480, 1024, 1024)]:
        A = _r((K, M) if ta else (M, K), M + K)
        B = _r((K, N) if tb else (N, K), N + K)
        want = torch.zeros(M, N, device="cuda")
        ops.gemm(A, B, want, ta=ta, tb=tb, epi=abi.EPI_ACC_F32, safe=2)
        first = None
        for it in range(5):
            got = torch.full((M, N), float("nan"), device="cuda")
            ops.gemm(A, B, got, ta=ta, tb=tb, epi=abi.EPI_ACC_F32, safe=5)
            assert _rel(got, want) < 1e-5, f"{(M, N, K)} iteration {it}: rel {_rel(got, want)}"
            first = got if first is None else first
            assert torch.equal(got, first), f"{(M, N, K)} iteration {it}: not repeatable"


def test_gemm_mid_kernel_epilogues_at_projection_shapes(ops):
    """Every (layout, epilogue) the hot path sends to the 128x128 LDS-DMA kernel, at the gated blocks' projection shapes and
    through of_gemm's OWN selection (safe = 0): to_q, to_out + gate + residual (fp32 stream), dX of to_out with the gate-gradient
    dot, dX of to_q, the four weight gradients (split along K, slabs), the Perceiver FFN's GELU / DGELU launches at N = 4096 rows
    -- element-wise against fp32 products of the same bf16 operands."""
    rows, d, inner = 8192, 2048, 512
    gate = torch.tensor([0.41], device="cuda")
    g = float(torch.tanh(gate))
    x, Wq = _r((rows, d), 71), _r((inner, d), 72, 0.03)
    q = torch.zeros(rows, inner, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, Wq, q)
    assert _rel(q, x.float() @ Wq.float().t()) < 1e-2
    o, Wo, res = _r((rows, inner), 73), _r((d, inner), 74, 0.05), torch.randn(rows, d, device="cuda")
    y = torch.zeros(rows, d, device="cuda")
    ops.gemm(o, Wo, y, epi=abi.EPI_GATE_RESID, aux=res, gate=gate)
    want = res + g * (o.float() @ Wo.float().t())
    assert _rel(y, want) < 2e-4 and float((y - want).abs().max()) < 2e-3
    dy = _r((rows, d), 75)
    dO, dot = torch.zeros(rows, inner, device="cuda", dtype=torch.bfloat16), torch.zeros(1, device="cuda")
    ops.gemm(dy, Wo, dO, tb=True, epi=abi.EPI_SCALE_DOT, aux=o, gate=gate, dot=dot)
    acc = dy.float() @ Wo.float()
    assert _rel(dO, g * acc) < 1e-2
    wdot = (1 - g * g) * (o.double() * acc.double()).sum()
    assert abs(float(dot) - float(wdot)) <= 1e-3 * abs(float(wdot)) + 1e-1
    dq = _r((rows, inner), 76)
    dxn = torch.zeros(rows, d, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dq, Wq, dxn, tb=True)
    assert _rel(dxn, dq.float() @ Wq.float()) < 1e-2
    for A_, B_, shape in ((dy, o, (d, inner)), (dq, x, (inner, d))):            # dW = dY^T X, K = 8192 tokens
        c0 = torch.randn(*shape, device="cuda")
        got = c0.clone()
        ops.gemm(A_, B_, got, ta=True, tb=True, epi=abi.EPI_ACC_F32, beta=1.0)
        assert _rel(got, c0 + A_.float().t() @ B_.float()) < 2e-4
        got2 = torch.full(shape, float("nan"), device="cuda")
        ops.gemm(A_, B_, got2, ta=True, tb=True, epi=abi.EPI_ACC_F32, beta=0.0)
        assert _rel(got2, A_.float().t() @ B_.float()) < 2e-4
    # Perceiver FFN (4096 latent rows): up + GELU with both outputs, dgelu_dot without a gate
    u, W1 = _r((4096, 1024), 77), _r((4096, 1024), 78, 0.05)
    a_out = torch.zeros(4096, 4096, device="cuda", dtype=torch.bfloat16)
    b_out = torch.zeros_like(a_out)
    ops.gemm(u, W1, b_out, epi=abi.EPI_GELU, out2=a_out)
    acc = u.float() @ W1.float().t()
    assert _rel(a_out, acc) < 1e-2 and _rel(b_out, torch.nn.functional.gelu(acc)) < 1e-2


def test_gemm_n_split_of_a_partially_filled_last_round(ops):
    """M = 8192, N = 2560 (OF-4B: 320 big tiles = 1.25 rounds) is launched as 8192 x 2048 on the 256x256 kernel + the 8192 x 512
    strip on the 128x128 kernel: every epilogue of_gemm's own selection (safe = 0) can meet at that shape against the general
    kernel (safe = 2) -- the seam at column 2048 included, aux / second output / gate-gradient partials of both parts."""
    M, N, K = 8192, 2560, 1024
    gate = torch.tensor([0.41], device="cuda")
    A = _r((M, K), 91)
    Wnt, Wnn = _r((N, K), 92, 0.05), _r((K, N), 93, 0.05)
    res = torch.randn(M, N, device="cuda")
    outs = {}
    for safe in (0, 2):
        y = torch.zeros(M, N, device="cuda")
        ops.gemm(A, Wnt, y, epi=abi.EPI_GATE_RESID, aux=res, gate=gate, safe=safe)
        yb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, Wnt, yb, epi=abi.EPI_GATE_RESID, aux=res.to(torch.bfloat16), gate=gate, safe=safe)
        b_out, a_out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, Wnt, b_out, epi=abi.EPI_GELU, out2=a_out, safe=safe)
        dx = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, Wnn, dx, tb=True, safe=safe)
        sd, dot = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), torch.zeros(1, device="cuda")
        ops.gemm(A, Wnn, sd, tb=True, epi=abi.EPI_SCALE_DOT, aux=b_out, gate=gate, dot=dot, safe=safe)
        At = _r((K, M), 94)                                   # TN: dW (M x N) = At^T Wnn, accumulated
        acc = torch.ones(M, N, device="cuda")
        ops.gemm(At, Wnn, acc, ta=True, tb=True, epi=abi.EPI_ACC_F32, beta=1.0, safe=safe)
        outs[safe] = (y, yb, b_out, a_out, dx, sd, dot, acc)
    names = ("gate_resid f32", "gate_resid bf16", "gelu", "pre-activation", "NN store", "scale_dot", "dot", "TN acc")
    for name, got, want in zip(names, outs[0], outs[2]):
        tol = 1e-2 if got.dtype == torch.bfloat16 else 2e-4
        assert _rel(got, want) < tol, (name, _rel(got, want))
        if got.dim() == 2:                                   # the seam: last column of the big part, first of the strip
            assert _rel(got[:, 2040:2056], want[:, 2040:2056]) < tol, name


@pytest.mark.parametrize("M,N,K,safe", [(8192, 8192, 2048, 0), (8192, 512, 2048, 0), (4096, 4096, 1024, 0), (8192, 8192, 2048, 7)])
def test_gate_gradient_dot_is_bit_reproducible(ops, M, N, K, safe):
    """The gate gradient of the *_DOT epilogues (per-workgroup partials + ordered finish, no fp32 atomics): ten launches, one
    bit pattern -- at the benchmark's DGELU_DOT shape (1024 workgroups) and the projection shapes."""
    A, W, aux = _r((M, K), 81), _r((K, N), 82, 0.05), _r((M, N), 83)
    gate = torch.tensor([0.5], device="cuda")
    vals = []
    for _ in range(10):
        o, dot = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.zeros(1, device="cuda")
        ops.gemm(A, W, o, tb=True, epi=abi.EPI_DGELU_DOT, aux=aux, gate=gate, dot=dot, safe=safe)
        vals.append(dot.clone())
    assert all(torch.equal(v, vals[0]) for v in vals), [float(v) for v in vals]
    acc = (A.float() @ W.float()).double()
    g = float(torch.tanh(gate))
    wdot = (1 - g * g) * (torch.nn.functional.gelu(aux.double()) * acc).sum()
    assert abs(float(vals[0]) - float(wdot)) <= 2e-3 * abs(float(wdot)) + 1.0


@pytest.mark.parametrize("beta", [0.0, 1.0])
def test_gemm_split_k_weight_gradient(ops, beta):
    """dW = dY^T X with a small output and K = tokens is split along K (fp32 atomics into C): same result as the
    unsplit general kernel up to fp32 summation order."""
    M, N, K = 512, 2048, 8192
    A, B = _r((K, M), 31), _r((K, N), 32)
    c0 = torch.randn(M, N, device="cuda")
    want, got = c0.clone(), c0.clone()
    gate = torch.tensor([0.3], device="cuda")
    ops.gemm(A, B, want, ta=True, tb=True, epi=abi.EPI_ACC_F32, beta=beta, gate=gate, safe=2)
    ops.gemm(A, B, got, ta=True, tb=True, epi=abi.EPI_ACC_F32, beta=beta, gate=gate, safe=0)
    assert _rel(got, want) < 1e-5
    ref = float(torch.tanh(gate)) * (A.float().t() @ B.float()) + beta * c0
    assert _rel(got, ref) < 2e-4


@pytest.mark.parametrize("M", [1, 3, 8, 16])
@pytest.mark.parametrize("N,K", [(512, 2048), (2048, 512), (8192, 2048), (2048, 8192), (10240, 2560)])
def test_skinny_gemm_decode_shapes(ops, M, N, K):
    """Decode-step projections (OF-3B / OF-4B widths, 1..16 sequences): the weight-streaming kernel (safe = 0 at
    M <= 16) against fp32 matmul and against the tile kernel (safe = 2), three epilogues."""
    A, B = _r((M, K), 31 + M), _r((N, K), 32, 0.05)
    acc = A.float() @ B.float().t()
    gate = torch.tensor([0.37], device="cuda")
    g = float(torch.tanh(gate))
    res = _r((M, N), 33, dtype=torch.float32)
    for safe in (0, 2):
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, B, o, safe=safe)
        assert _rel(o, acc) < 1e-2
        b = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, B, b, epi=abi.EPI_GELU, safe=safe)
        assert _rel(b, torch.nn.functional.gelu(acc)) < 1e-2
        y = torch.empty(M, N, dtype=torch.float32, device="cuda")
        ops.gemm(A, B, y, epi=abi.EPI_GATE_RESID, aux=res, gate=gate, safe=safe)
        assert _rel(y, res + g * acc) < 1e-4 + 1e-3 * (K > 4096)


def _close(got, want, name, rtol=1e-2, atol_rms=2e-3, l2=4e-3):
    """Element-wise |got - want| <= atol + rtol |want| (atol = atol_rms x rms(want): fp32 summation order near zero;
    rtol covers ONE bf16 rounding of the stored value, 2^-8) AND relative L2 -- max-abs-over-max alone would hide
    systematically wrong small elements."""
    got, want = got.double(), want.double()
    rms = want.pow(2).mean().sqrt().item()
    viol = (got - want).abs() - (atol_rms * rms + rtol * want.abs())
    assert viol.max().item() <= 0, f"{name}: {int((viol > 0).sum())} elements out of tolerance (worst excess {viol.max().item():.3e}, rms {rms:.3e})"
    e = (got - want).norm().item() / (want.norm().item() + 1e-30)
    assert e <= l2, f"{name}: rel L2 {e:.3e}"


@pytest.mark.parametrize("safe", [0, 4, 6, 7, 16, 17, 18])
def test_big_tile_gemm_fused_epilogues_at_benchmark_shapes(ops, safe):
    """The launches bench.py times at BASELINE config 2 (per gated block: rows = B*L = 8192, d = 2048, hidden 8192) run the
    256x256 kernel with FUSED epilogues; small-batch tests select the 128x128 kernel.  Every (layout, epilogue) pair the
    step uses, at those shapes, against fp32 torch on the same bf16 operands; safe=0 is of_gemm's own selection (asserted
    to be the big-tile kernel), safe=4 forces it."""
    from open_flamingo_amd.hip.ops import Ops
    gate = torch.tensor([0.37], device="cuda")
    g = float(torch.tanh(gate))
    rows, d, hid, inner = 8192, 2048, 8192, 512
    assert Ops.kernel_label(rows, hid, d, False, False) == "w4m256" and Ops.kernel_label(rows, d, hid, False, True) == "w4m256"
    # (safe = 18: the two-workgroups-per-CU 256x128 kernel, of_gemm's own selection for the *_DOT launches over >= 1024 big tiles;
    #  it does not take the TN weight gradient below, which then runs the general kernel)
    assert Ops.kernel_label(rows, hid, d, False, True, abi.EPI_DGELU_DOT) == "w4h256x128"
    # ---- up-projection + erf-GELU, two outputs (pre-activation kept for the backward): NT 8192 x 8192 x 2048
    u, W1 = _r((rows, d), 41), _r((hid, d), 42, d ** -0.5)
    acc = u.float() @ W1.float().t()
    b = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
    a = torch.empty_like(b)
    ops.gemm(u, W1, b, epi=abi.EPI_GELU, out2=a, safe=safe)
    _close(a, acc, "pre-GELU")
    _close(b, torch.nn.functional.gelu(acc), "GELU")
    # ---- down-projection * tanh(gate) + residual: NT 8192 x 2048 x 8192, fp32 and bf16 residual streams
    W2 = _r((d, hid), 43, hid ** -0.5)
    acc2 = b.float() @ W2.float().t()
    res = _r((rows, d), 44, dtype=torch.float32)
    y = torch.empty(rows, d, device="cuda")
    ops.gemm(b, W2, y, epi=abi.EPI_GATE_RESID, aux=res, gate=gate, safe=safe)
    _close(y, res + g * acc2, "GATE_RESID fp32", rtol=1e-5, atol_rms=1e-4, l2=1e-5)
    resb = res.to(torch.bfloat16)
    yb = torch.empty(rows, d, device="cuda", dtype=torch.bfloat16)
    ops.gemm(b, W2, yb, epi=abi.EPI_GATE_RESID, aux=resb, gate=gate, safe=safe)
    _close(yb, resb.float() + g * acc2, "GATE_RESID bf16")
    # ---- dA = (dY W2) * gate * gelu'(a), gate-gradient dot = (1 - g^2) sum(gelu(a) * dY W2): NN 8192 x 8192 x 2048
    dy = _r((rows, d), 45)
    accd = dy.float() @ W2.float()
    da = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
    dot = torch.zeros(1, device="cuda")
    ops.gemm(dy, W2, da, tb=True, epi=abi.EPI_DGELU_DOT, aux=a, gate=gate, dot=dot, safe=safe)
    ad = a.double().requires_grad_(True)
    ge = torch.nn.functional.gelu(ad)
    ge.sum().backward()
    _close(da, g * accd.double() * ad.grad, "DGELU")
    wdot = (1 - g * g) * (ge.detach() * accd.double()).sum().item()
    ref_scale = (1 - g * g) * (ge.detach() * accd.double()).abs().sum().item()
    assert abs(float(dot) - wdot) <= 1e-5 * ref_scale, ("DGELU dot", float(dot), wdot, ref_scale)
    # ---- dO = (dY1 Wout) * gate, dot = (1 - g^2) sum(o * dY1 Wout): NN 8192 x 512 x 2048 (too few tiles for the big
    #      kernel by itself: only safe=4 runs it there) and a big-tile-eligible 8192 x 2048 x 512
    for (M, N, K) in ((rows, inner, d), (rows, d, inner)):
        dy1, Wo, o = _r((M, K), 46), _r((K, N), 47, K ** -0.5), _r((M, N), 48)
        acco = dy1.float() @ Wo.float()
        dO = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        dot = torch.zeros(1, device="cuda")
        ops.gemm(dy1, Wo, dO, tb=True, epi=abi.EPI_SCALE_DOT, aux=o, gate=gate, dot=dot, safe=safe)
        _close(dO, g * acco, f"SCALE {M}x{N}x{K}")
        wdot = (1 - g * g) * (o.double() * acco.double()).sum().item()
        ref_scale = (1 - g * g) * (o.double() * acco.double()).abs().sum().item()
        assert abs(float(dot) - wdot) <= 1e-5 * ref_scale, ("SCALE dot", float(dot), wdot)
    # ---- dW2 += gate * dY^T b (accumulating into an existing fp32 gradient): TN 2048 x 8192 x 8192
    c = torch.randn(d, hid, device="cuda")
    want = c + g * (dy.float().t() @ b.float())
    ops.gemm(dy, b, c, ta=True, tb=True, epi=abi.EPI_ACC_F32, gate=gate, beta=1.0, safe=safe)
    _close(c, want, "dW2 beta=1", rtol=1e-5, atol_rms=1e-4, l2=1e-5)
    # ---- plain dX: NN 8192 x 2048 x 8192
    du = torch.empty(rows, d, device="cuda", dtype=torch.bfloat16)
    ops.gemm(da, W1, du, tb=True, safe=safe)
    _close(du, da.float() @ W1.float(), "dU")


# ---- the two-workgroups-per-CU 256 x 128 kernel (csrc/gemm_w4h.hip, safe = 18) -------------------------------------------------------
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 704, 1088])
def test_half_tile_kernel_ring_tails_and_wraps_on_hardware(ops, tb, K):
    """1 ... 17 K stages through the ring of five 16-KiB units (prologue only, every tail form, the unit -> slot map wrapped three
    times) with every CU holding two workgroups: bit-equal to the 256x256 kernel on the same MFMAs (safe = 16), five launches one
    bit pattern (the race screen of the slot reuse: a piece landing in a slot that is still being read shows up here)."""
    M, N = 4096, 4096          # 512 half tiles: one full round of two workgroups per CU
    A = _r((M, K), 91)
    B = _r((K, N) if tb else (N, K), 92, K ** -0.5)
    want = torch.empty(M, N, device="cuda")
    ops.gemm(A, B, want, tb=tb, epi=abi.EPI_ACC_F32, safe=16)
    ref = A.float() @ (B.float() if tb else B.float().t())
    _close(want, ref, "256x256", rtol=1e-5, atol_rms=1e-4, l2=1e-5)
    for _ in range(5):
        got = torch.zeros(M, N, device="cuda")
        ops.gemm(A, B, got, tb=tb, epi=abi.EPI_ACC_F32, safe=18)
        assert torch.equal(got, want)


def test_half_tile_kernel_is_bit_equal_to_the_big_tile_kernel_at_benchmark_shapes(ops):
    """The launches of_gemm sends to the half-tile kernel (NN *_DOT over >= 1024 big tiles) and the ones it could take (GELU with two
    outputs, gate + residual): outputs bit-equal to the 256x256 kernel's, the gate-gradient dot equal up to the order of the
    per-tile partials (twice as many) and bit-reproducible over five launches."""
    rows, d, hid = 8192, 2048, 8192
    gate = torch.tensor([0.37], device="cuda")
    u, W1 = _r((rows, d), 41), _r((hid, d), 42, d ** -0.5)
    outs = {}
    for safe in (16, 18):
        b, a = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16), torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
        ops.gemm(u, W1, b, epi=abi.EPI_GELU, out2=a, safe=safe)
        outs[safe] = (b, a)
    assert torch.equal(outs[16][0], outs[18][0]) and torch.equal(outs[16][1], outs[18][1])
    a = outs[16][1]
    W2, dy = _r((d, hid), 43, hid ** -0.5), _r((rows, d), 45)
    for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        res = {}
        for safe in (16, 18, 0):
            das, dots = [], []
            for _ in range(5 if safe == 18 else 1):
                da, dot = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16), torch.zeros(1, device="cuda")
                ops.gemm(dy, W2, da, tb=True, epi=epi, aux=a, gate=gate, dot=dot, safe=safe)
                das.append(da)
                dots.append(dot)
            assert all(torch.equal(das[0], x) for x in das) and all(torch.equal(dots[0], x) for x in dots)
            res[safe] = (das[0], dots[0])
        assert torch.equal(res[16][0], res[18][0]) and torch.equal(res[0][0], res[18][0])
        assert torch.equal(res[0][1], res[18][1])          # safe = 0 IS the half-tile kernel here
        assert abs(float(res[16][1]) - float(res[18][1])) <= 2e-6 * abs(float(res[16][1])) + 1e-6
    res32 = _r((rows, d), 44, dtype=torch.float32)
    bb = outs[16][0]
    ys = []
    for safe in (16, 18):
        y = torch.empty(rows, d, device="cuda")
        ops.gemm(bb, W2, y, epi=abi.EPI_GATE_RESID, aux=res32, gate=gate, safe=safe)
        ys.append(y)
    assert torch.equal(ys[0], ys[1])


# ---- the persistent wave-specialised 256 x 128 kernel (csrc/gemm_w4s.hip, safe = 19; built and measured in round 5, not selected) ---------
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("K,cus", [(1152, 0), (1216, 8), (1280, 64), (2048, 0)])
def test_specialised_kernel_plain_store_is_bit_equal_on_hardware(ops, tb, K, cus):
    """Four MFMA waves + four producer waves per workgroup, one workgroup per CU walking its tiles (cu_limit = workgroups: 512 half
    tiles over 512 / 8 / 64 / 256 workgroups -- one, 64, 8 and two tiles each): the bf16 store is bit-equal to the 256x256 kernel's and
    five launches give one bit pattern (the race screen of the six-unit ring, of the image hand-over between a consumer and its
    producer, and of the accumulator bank in fixed registers -- round 5 found hipcc permuting C++ accumulators with v_accvgpr_mov
    among the MFMAs on hardware only)."""
    M, N = 4096, 4096
    A = _r((M, K), 93)
    B = _r((K, N) if tb else (N, K), 94, K ** -0.5)
    want = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, want, tb=tb, alpha=0.5, safe=16)
    old = ops.cu_limit
    try:
        ops.cu_limit = cus
        for _ in range(5):
            got = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            ops.gemm(A, B, got, tb=tb, alpha=0.5, safe=19)
            assert torch.equal(got, want)
    finally:
        ops.cu_limit = old


def test_specialised_kernel_fused_epilogues_at_benchmark_shapes(ops):
    """GELU with two outputs and both *_DOT epilogues drained by the producer waves under the next tile's K loop, at the shapes of a
    gated block's FFN: the pre-activation is the rounded product bit for bit, GELU / dGELU are taken of the ROUNDED product (the
    reference's own order under autocast) -- compared with fp32 torch by the tolerances of the 256x256 kernel's test."""
    rows, d, hid = 8192, 2048, 8192
    gate = torch.tensor([0.37], device="cuda")
    g = float(torch.tanh(gate))
    u, W1 = _r((rows, d), 41), _r((hid, d), 42, d ** -0.5)
    acc = u.float() @ W1.float().t()
    b, a = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16), torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
    ops.gemm(u, W1, b, epi=abi.EPI_GELU, out2=a, safe=19)
    a16 = torch.empty_like(a)
    ops.gemm(u, W1, torch.empty_like(b), epi=abi.EPI_GELU, out2=a16, safe=16)
    assert torch.equal(a, a16)
    _close(a, acc, "pre-GELU")
    _close(b, torch.nn.functional.gelu(a.float()), "GELU of the rounded product")
    _close(b, torch.nn.functional.gelu(acc), "GELU", rtol=2e-2)
    W2, dy = _r((d, hid), 43, hid ** -0.5), _r((rows, d), 45)
    accd = dy.float() @ W2.float()
    for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        das, dots = [], []
        for _ in range(3):
            da, dot = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16), torch.zeros(1, device="cuda")
            ops.gemm(dy, W2, da, tb=True, epi=epi, aux=a, gate=gate, dot=dot, safe=19)
            das.append(da)
            dots.append(dot)
        assert all(torch.equal(das[0], x) for x in das) and all(torch.equal(dots[0], x) for x in dots)
        ad = a.double().requires_grad_(True)
        ge = torch.nn.functional.gelu(ad)
        ge.sum().backward()
        if epi == abi.EPI_DGELU_DOT:
            want, terms = g * accd.double() * ad.grad, ge.detach() * accd.double()
        else:
            want, terms = g * accd.double(), a.double() * accd.double()
        _close(das[0], want, "dX", rtol=2e-2)
        wdot, ref_scale = (1 - g * g) * terms.sum().item(), (1 - g * g) * terms.abs().sum().item()
        assert abs(float(dots[0]) - wdot) <= 1e-5 * ref_scale, (float(dots[0]), wdot, ref_scale)


# ---- stream-K schedule of the 16x16x32 big-tile kernel (csrc/gemm_w4m.hip): tile counts the workgroup count does not divide --------
_SK_SHAPES = [  # (M, N, K, ta, tb, epi, cu_limit)   tiles / workgroups
    (2048, 4096, 16384, False, True, abi.EPI_STORE_BF16, 0),  # OF-9B L = 256 dX: 128 tiles / 256 -> half a tile per workgroup
    (4096, 2048, 16384, True, True, abi.EPI_ACC_F32, 0),      # the same tile count as a weight gradient (fp32 out)
    (10240, 2560, 8192, True, True, abi.EPI_ACC_F32, 248),    # OF-4B ffn dW, 400 tiles / 248 workgroups: 1 round + 152 shared tiles
    (8192, 2048, 8192, False, False, abi.EPI_STORE_BF16, 192),   # 256 tiles next to a collective holding 64 CUs: 1 1/3 tiles each
    (8192, 8192, 4096, False, True, abi.EPI_STORE_BF16, 224),    # 1024 tiles / 224 workgroups: 4 rounds + 128 shared tiles
]


@pytest.mark.parametrize("M,N,K,ta,tb,epi,cu", _SK_SHAPES)
def test_stream_k_matches_classic_launch_and_is_bit_reproducible(ops, M, N, K, ta, tb, epi, cu):
    """of_gemm's own selection (safe = 0: stream-K for fewer tiles than workgroups, or for a tile count an explicit cu_limit does not
    divide; K >= 4096) with the stream-K workspace vs the classic one-tile-per-workgroup launch (safe = 16) of the
    same kernel and vs fp32 torch: same products, fp32 accumulation, a shared tile's sum split at workgroup boundaries.  Five
    launches of the stream-K form give ONE bit pattern (fixed-order fix-up through the workspace; this is also the race screen of
    the flag / partial-tile hand-off between workgroups on different XCDs)."""
    A = _r((K, M) if ta else (M, K), 81)
    B = _r((K, N) if tb else (N, K), 82, K ** -0.5)
    ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
    dt = torch.float32 if epi == abi.EPI_ACC_F32 else torch.bfloat16
    old = ops.cu_limit
    try:
        ops.cu_limit = cu
        outs = []
        for _ in range(5):
            o = torch.empty(M, N, device="cuda", dtype=dt)
            ops.gemm(A, B, o, ta=ta, tb=tb, epi=epi)
            outs.append(o)
    finally:
        ops.cu_limit = old
    cl = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm(A, B, cl, ta=ta, tb=tb, epi=epi, safe=16)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    if dt == torch.float32:
        _close(outs[0], ref, "stream-K vs fp32", rtol=1e-5, atol_rms=1e-4, l2=1e-5)
        assert (outs[0] - cl).abs().max().item() <= 1e-4 * ref.abs().max().item()
    else:
        _close(outs[0], ref, "stream-K vs fp32")
        assert (outs[0].float() - cl.float()).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()      # one bf16 ulp at the top
    assert not torch.equal(outs[0], cl) or dt == torch.bfloat16     # fp32: the split sums differ in the last bits somewhere


def test_stream_k_fused_epilogues_at_of4b_shapes(ops):
    """The fused epilogues behind shared tiles at OF-4B's widths (d = 2560) under cu_limit = 192 (a collective holding 64 CUs):
    1280 tiles = 6 rounds + 128 shared, 320 = 1 round + 128, 400 = 2 rounds + 16 -- up-projection + GELU (two outputs), gate +
    residual on the fp32 stream, dGELU + gate dot, accumulating weight gradient, against fp32 torch."""
    old_limit, ops.cu_limit = ops.cu_limit, 192
    try:
        _sk_epilogues_of4b(ops)
    finally:
        ops.cu_limit = old_limit


def _sk_epilogues_of4b(ops):
    gate = torch.tensor([0.37], device="cuda")
    g = float(torch.tanh(gate))
    rows, d, hid = 8192, 2560, 10240
    u, W1 = _r((rows, d), 91), _r((hid, d), 92, d ** -0.5)
    acc = u.float() @ W1.float().t()
    b, a = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16), torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16)
    ops.gemm(u, W1, b, epi=abi.EPI_GELU, out2=a)                                   # K = 2560 < 4096: one tile per workgroup
    _close(a, acc, "pre-GELU")
    W2 = _r((d, hid), 93, hid ** -0.5)
    acc2 = b.float() @ W2.float().t()
    res = _r((rows, d), 94, dtype=torch.float32)
    y = torch.empty(rows, d, device="cuda")
    ops.gemm(b, W2, y, epi=abi.EPI_GATE_RESID, aux=res, gate=gate)                   # 320 tiles / 192: 1 round + 128 shared
    _close(y, res + g * acc2, "GATE_RESID fp32 (320 tiles)", rtol=1e-5, atol_rms=1e-4, l2=1e-5)
    dy = _r((rows, d), 95)
    accd = dy.float() @ W2.float()
    da, dot = torch.empty(rows, hid, device="cuda", dtype=torch.bfloat16), torch.zeros(1, device="cuda")
    ops.gemm(dy, W2, da, tb=True, epi=abi.EPI_DGELU_DOT, aux=a, gate=gate, dot=dot)
    ad = a.double().requires_grad_(True)
    ge = torch.nn.functional.gelu(ad)
    ge.sum().backward()
    _close(da, g * accd.double() * ad.grad, "DGELU")
    wdot = (1 - g * g) * (ge.detach() * accd.double()).sum().item()
    ref_scale = (1 - g * g) * (ge.detach() * accd.double()).abs().sum().item()
    assert abs(float(dot) - wdot) <= 1e-5 * ref_scale
    c = torch.randn(d, hid, device="cuda")
    want = c + g * (dy.float().t() @ b.float())
    ops.gemm(dy, b, c, ta=True, tb=True, epi=abi.EPI_ACC_F32, gate=gate, beta=1.0)   # 400 tiles / 192: 2 rounds + 16 shared
    _close(c, want, "dW2 beta=1 (400 tiles)", rtol=1e-5, atol_rms=1e-4, l2=1e-5)


def test_stream_k_partial_tiles_are_never_stale(ops):
    """The partial tiles cross XCDs with system-scope stores / loads instead of device-scope fences (of_platform.h).  A load that
    took a line left in this XCD's L2 by an EARLIER launch would go unnoticed by a test that repeats one problem (the stale bytes
    would be the right ones): alternate two different problems through the same workspace, eight launches, each checked against
    the classic launch of its own operands."""
    M, N, K = 2048, 4096, 8192              # 128 tiles: every tile shared by two workgroups
    probs = []
    for seed in (101, 202):
        A, B = _r((M, K), seed), _r((K, N), seed + 1, K ** -0.5)
        cl = torch.empty(M, N, device="cuda")
        ops.gemm(A, B, cl, tb=True, epi=abi.EPI_ACC_F32, safe=16)
        probs.append((A, B, cl))
    first = {}
    for i in range(8):
        A, B, cl = probs[i % 2]
        o = torch.empty(M, N, device="cuda")
        ops.gemm(A, B, o, tb=True, epi=abi.EPI_ACC_F32)
        assert (o - cl).abs().max().item() <= 1e-4 * cl.abs().max().item(), i
        if i % 2 in first:
            assert torch.equal(o, first[i % 2]), i
        first.setdefault(i % 2, o)


def test_gemm_batch_of_weight_gradients_at_benchmark_shapes(ops):
    """of_gemm_batch: to_out dW (2048 x 512), to_q dW (512 x 2048) over 8192 token rows and to_kv dW (1024 x 1024) over 4096 media rows
    (a strided column block of the grouped d(k|v) buffer) in one launch -- bit-identical to the three separate split-K launches,
    beta 0 and 1, gate on the first."""
    rows, d, inner, mrows = 8192, 2048, 512, 4096
    gate = torch.tensor([0.37], device="cuda")
    dy, o, dq, xn = _r((rows, d), 111), _r((rows, inner), 112), _r((rows, inner), 113), _r((rows, d), 114)
    dkv_all, media = _r((mrows, 24 * 1024), 115), _r((mrows, 1024), 116)
    dkv = dkv_all[:, 5 * 1024:6 * 1024]
    probs = [(dy, o, 1.0, gate), (dq, xn, 0.0, None), (dkv, media, 0.0, None)]
    want = []
    for A, B, beta, g in probs:
        c = torch.ones(A.shape[1], B.shape[1], device="cuda")
        ops.gemm(A, B, c, ta=True, tb=True, epi=abi.EPI_ACC_F32, beta=beta, gate=g)
        want.append(c)
    got = [torch.ones_like(w) for w in want]
    ops.gemm_batch_dw([(A, B, c, beta, g) for (A, B, beta, g), c in zip(probs, got)])
    for g_, w in zip(got, want):
        assert torch.equal(g_, w)
    ref = float(torch.tanh(gate)) * (dy.float().t() @ o.float()) + 1.0
    _close(got[0], ref, "to_out dW", rtol=1e-5, atol_rms=1e-4, l2=1e-5)


def test_product_library_takes_no_kernel_forcing_selector():
    """OfGemmArgs.safe: 0 (of_gemm selects) and 1 (the checked scalar-LDS path) are all the product library accepts -- the selectors this
    suite forces kernels with live in the tools build of the same sources (tests/gpu_ops.py)."""
    from open_flamingo_amd.hip.ops import Ops
    ops = Ops.default()
    A, B = _r((256, 64), 1), _r((256, 64), 2)
    out = torch.empty(256, 256, dtype=torch.bfloat16, device="cuda")
    for safe in (2, 4, 5, 7, 16, 17, 18, 19, 20, -1):
        with pytest.raises(RuntimeError, match="OF_E_ARG"):
            ops.gemm(A, B, out, safe=safe)
    ops.gemm(A, B, out, safe=0)
    want = out.clone()
    ops.gemm(A, B, out, safe=1)
    assert torch.equal(out, want)


@pytest.mark.parametrize("dh,heads,B,L,causal", [(128, 16, 32, 256, True), (64, 16, 16, 256, True), (128, 16, 16, 200, True), (128, 8, 32, 256, False)])
def test_single_pass_backward_matches_the_two_passes_at_tower_shapes(ops, dh, heads, B, L, causal):
    """of_attn_bwd's single-pass form (csrc/attn_bwd_res.hip: dQ, dK, dV of a (batch, head) from one recomputation of P) at the shape
    it is selected for -- the frozen MPT blocks of OF-3B: 32 x 16 heads, 256 x 256, head 128, causal + ALiBi, q | k | v views of the
    fused projection -- and at head 64 / a ragged length / no mask: of_attn_bwd's own choice (safe = 0) and the forced form (3) against the
    two-pass kernels (2) and against autograd through a dense fp32 softmax."""
    d = heads * dh
    qkv = _r((B * L, 3 * d), 73)
    do = _r((B * L, d), 74)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / heads) for i in range(heads)], device="cuda") if causal else None
    kw = dict(batch=B, Lq=L, Lk=L, heads=heads, scale=dh ** -0.5, head_dim=dh, causal=causal, alibi_slopes=slopes)
    o = torch.empty(B * L, d, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, heads, L, device="cuda")
    ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, **kw)
    outs = {}
    for safe in (0, 3, 2):
        dqkv = torch.full_like(qkv, float("nan"))
        delta = torch.zeros(B, heads, L, device="cuda")
        ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], delta, safe=safe, **kw)
        assert torch.isfinite(dqkv.float()).all(), safe
        outs[safe] = dqkv
    if 10 * B * heads >= 7 * 256 * ((B * heads + 255) // 256):
        assert torch.equal(outs[0], outs[3]), "of_attn_bwd did not choose the single pass at a shape its rule covers"
    assert _rel(outs[3], outs[2].double()) < 4e-3
    # run to run: bit-reproducible (no atomics, fixed order)
    again = torch.full_like(qkv, float("nan"))
    ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, again[:, :d], again[:, d:2 * d], again[:, 2 * d:],
                 torch.zeros(B, heads, L, device="cuda"), safe=3, **kw)
    assert torch.equal(again, outs[3])
    nb = min(B, 4)                                   # autograd reference on the first sequences
    x = qkv[:nb * L].float().requires_grad_(True)
    q, k, v = (x[:, i * d:(i + 1) * d].view(nb, L, heads, dh).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) * dh ** -0.5
    if causal:
        pos = torch.arange(L, device="cuda")
        s = s + slopes.view(1, heads, 1, 1) * (pos.view(1, 1, 1, L) - pos.view(1, 1, L, 1))
        s = s.masked_fill(pos.view(1, 1, 1, L) > pos.view(1, 1, L, 1), float("-inf"))
    (s.softmax(-1) @ v).transpose(1, 2).reshape(nb * L, d).backward(do[:nb * L].float())
    for i, name in enumerate(("dq", "dk", "dv")):
        assert _rel(outs[3][:nb * L, i * d:(i + 1) * d], x.grad[:, i * d:(i + 1) * d]) < 2e-2, name


@pytest.mark.parametrize("hv,dh,heads,B,L", [(80, 128, 32, 16, 256), (80, 128, 5, 3, 200), (40, 64, 8, 8, 256), (96, 128, 4, 2, 320)])
def test_compact_heads_equal_the_zero_padded_launch(ops, hv, dh, heads, B, L):
    """OfAttnArgs.head_valid (ABI v11): heads of hv < head_dim columns side by side in memory -- GPT-NeoX head size 80 at the 128-wide
    kernels (OF-4B's frozen RedPajama-3B blocks: 16 x 32 heads, L = 256, causal, right padding) -- against the SAME kernels on zero-padded
    copies, bit for bit (the padded columns add exact zeros to every sum): forward (tiled, resident) and backward (two passes, single
    pass), of_attn's own choice included; and against autograd through a dense fp32 softmax."""
    d, dp = heads * hv, heads * dh
    q, k, v, do = (_r((B * L, d), 90 + i) for i in range(4))
    kv_len = torch.tensor([L - 7 * (i % 5) for i in range(B)], dtype=torch.int32, device="cuda")

    def pad(t):
        p = torch.zeros(B * L, heads, dh, dtype=t.dtype, device="cuda")
        p[..., :hv] = t.view(B * L, heads, hv)
        return p.view(B * L, dp)

    def unpad(t):
        return t.view(B * L, heads, dh)[..., :hv].reshape(B * L, d)

    def run(q, k, v, do, safe, head_valid):
        kw = dict(batch=B, Lq=L, Lk=L, heads=heads, scale=hv ** -0.5, head_dim=dh, head_valid=head_valid, causal=True, kv_len=kv_len, safe=safe)
        o = torch.full_like(q, float("nan"))
        lse = torch.full((B, heads, L), float("nan"), device="cuda")
        ops.attn_fwd(q, k, v, o, lse, **kw)
        dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
        ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, torch.zeros(B, heads, L, device="cuda"), **kw)
        return o, lse, dq, dk, dv

    res = {}
    single = dh == 128 and hv in (80, 96)            # the widths the single-pass backward is instantiated for; others take the two passes
    for safe in (0, 2) + ((3,) if L <= 256 and single else ()):
        got = run(q, k, v, do, safe, hv)
        want = run(pad(q), pad(k), pad(v), pad(do), safe, 0)
        assert torch.equal(got[1], want[1]), ("lse", safe)
        for name, g, w in zip(("o", "dq", "dk", "dv"), (got[0],) + got[2:], (want[0],) + want[2:]):
            assert torch.isfinite(g.float()).all(), (name, safe)
            assert torch.equal(g, unpad(w)), (name, safe)
        res[safe] = got
    nb = min(B, 3)
    x = [t[:nb * L].float().requires_grad_(True) for t in (q, k, v)]
    qh, kh, vh = (t.view(nb, L, heads, hv).transpose(1, 2) for t in x)
    s = qh @ kh.transpose(-1, -2) * hv ** -0.5
    pos = torch.arange(L, device="cuda")
    dead = (pos.view(1, 1, 1, L) > pos.view(1, 1, L, 1)) | (pos.view(1, 1, 1, L) >= kv_len[:nb].view(nb, 1, 1, 1))
    out = (s.masked_fill(dead, float("-inf")).softmax(-1) @ vh).transpose(1, 2).reshape(nb * L, d)
    out.backward(do[:nb * L].float())
    o, _, dq, dk, dv = res[0]
    assert _rel(o[:nb * L], out.detach()) < 1e-2
    for name, g, t in (("dq", dq, x[0]), ("dk", dk, x[1]), ("dv", dv, x[2])):
        assert _rel(g[:nb * L], t.grad) < 2e-2, name


def test_attention_backward_splits_a_ragged_head_count_between_the_two_forms(ops):
    """of_attn_bwd at OF-9B's frozen MPT-7B blocks (BASELINE config 5: 10 sequences x 32 heads = 320 (batch, head) pairs, 256 x 256, head
    128, causal + ALiBi): 1.25 rounds of the 256 CUs, so the single pass alone is not chosen; the first 8 sequences (one whole round)
    take it, the last 2 the two passes -- each sequence bit for bit what a launch of that form alone gives."""
    dh, heads, B, L = 128, 32, 10, 256
    d = heads * dh
    qkv, do = _r((B * L, 3 * d), 75), _r((B * L, d), 76)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / heads) for i in range(heads)], device="cuda")
    kw = dict(batch=B, Lq=L, Lk=L, heads=heads, scale=dh ** -0.5, head_dim=dh, causal=True, alibi_slopes=slopes)
    o = torch.empty(B * L, d, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, heads, L, device="cuda")
    ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, **kw)
    outs = {}
    for safe in (0, 3, 2):
        dqkv = torch.full_like(qkv, float("nan"))
        ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                     torch.zeros(B, heads, L, device="cuda"), safe=safe, **kw)
        assert torch.isfinite(dqkv.float()).all(), safe
        outs[safe] = dqkv
    cut = 8 * L
    assert torch.equal(outs[0][:cut], outs[3][:cut]) and torch.equal(outs[0][cut:], outs[2][cut:])
    assert _rel(outs[0], outs[2].double()) < 4e-3
