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
    from open_flamingo_amd.hip.ops import Ops
    assert torch.cuda.is_available()
    return Ops.default()


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


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_pingpong_race_screen(ops, ta, tb):
    """The 256x256 ping-pong kernel's LDS-DMA slots are only ordered by counted vmcnt + the segment barriers; the CPU emulator
    executes the DMA synchronously and cannot see a race.  Screen on hardware: many shapes (1..64 stages, single and
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
            ops.gemm(A, B, got, ta=ta, tb=tb, epi=abi.EPI_ACC_F32, safe=4)
            assert torch.equal(got, want), f"{(M, N, K)} iteration {it}: max diff {(got - want).abs().max().item()}"


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
