"""Kernel index-logic checks of the GEMM on the host SIMT emulator (tests/emu) against plain fp64 matmul.
Bit-level expectations: operands are exactly-representable bf16, accumulation is fp32 fma in k order, so the
comparison tolerance only covers summation order (1e-5 relative)."""
import numpy as np
import pytest
import torch

from open_flamingo_amd.hip import abi
from tests.emu import harness as H


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(torch.bfloat16)


def _ref(A, B, at, bt):
    a = A.double().t() if at else A.double()
    b = B.double() if bt else B.double().t()
    return a @ b


@pytest.mark.parametrize("at,bt,safe", [(0, 0, 0), (0, 1, 0), (0, 1, 1), (1, 1, 0), (1, 1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (64, 264, 200)])
def test_gemm_layouts_store(at, bt, safe, M, N, K):
    if at and (M % 8):
        M = (M + 7) // 8 * 8
    A = _rand((K, M) if at else (M, K), 1)
    B = _rand((K, N) if bt else (N, K), 2)
    if (not at) and K % 8:
        pytest.skip("K-contiguous operand needs K % 8 == 0")
    out = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32 if at else abi.EPI_STORE_BF16,
           C_out=(torch.zeros(M, N) if at else out), safe=safe)
    ref = _ref(A, B, at, bt)
    if at:
        o32 = torch.zeros(M, N)
        H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o32, safe=safe)
        np.testing.assert_allclose(o32.double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    else:
        np.testing.assert_allclose(out.double().numpy(), ref.to(torch.bfloat16).double().numpy(), rtol=1e-2, atol=1e-2)
        err = (out.double() - ref).abs().max() / ref.abs().max()
        assert err < 5e-3


def test_gemm_epilogues():
    M, N, K = 136, 192, 128
    A, B = _rand((M, K), 3), _rand((N, K), 4) * 0.1
    acc = _ref(A, B, 0, 0)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    # GELU (+ pre-activation)
    b_out, a_out = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out)
    np.testing.assert_allclose(a_out.double().numpy(), acc.numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=1e-2)
    # gate + residual, fp32 and bf16 stream
    res = torch.randn(M, N)
    out = torch.zeros(M, N)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=out, aux=res, gate=gate, io_f32=1)
    np.testing.assert_allclose(out.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    resb = res.to(torch.bfloat16)
    outb = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=outb, aux=resb, gate=gate, io_f32=0)
    np.testing.assert_allclose(outb.double().numpy(), (resb.double() + g * acc).numpy(), rtol=1e-2, atol=2e-2)
    # accumulate fp32 with beta
    c = torch.randn(M, N)
    c0 = c.clone()
    H.gemm(A, B, epi=abi.EPI_ACC_F32, C_out=c, alpha=0.5, beta=1.0, gate=gate)
    np.testing.assert_allclose(c.double().numpy(), (c0.double() + 0.5 * g * acc).numpy(), rtol=1e-5, atol=1e-4)


def test_gemm_dot_epilogues():
    M, N, K = 128, 136, 64
    A = _rand((M, K), 5)
    W = _rand((K, N), 6) * 0.2          # dX = dY W : b_trans = 1
    acc = A.double() @ W.double()
    aux = _rand((M, N), 7)
    gate = torch.tensor([-0.8])
    g = float(torch.tanh(gate))
    for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        out = torch.zeros(M, N, dtype=torch.bfloat16)
        dot = torch.zeros(1)
        H.gemm(A, W, b_trans=1, epi=epi, C_out=out, aux=aux, gate=gate, dot_out=dot)
        x = aux.double()
        if epi == abi.EPI_DGELU_DOT:
            xx = x.clone().requires_grad_(True)
            torch.nn.functional.gelu(xx).sum().backward()
            want = g * acc * xx.grad
            wdot = (1 - g * g) * (torch.nn.functional.gelu(x) * acc).sum()
        else:
            want = g * acc
            wdot = (1 - g * g) * (x * acc).sum()
        np.testing.assert_allclose(out.double().numpy(), want.numpy(), rtol=1e-2, atol=2e-2)
        assert abs(float(dot) - float(wdot)) <= 1e-3 * abs(float(wdot)) + 1e-2


@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(136, 192, 128), (256, 72, 64), (128, 320, 200)])
def test_gemm_narrow_tiles_match_wide_tiles(at, bt, M, N, K):
    """128 x 64 tiles (safe = 3; of_gemm's own choice for grids of <= 256 wide tiles) vs 128 x 128 tiles (safe = 2): the same
    products and k order per output element -> identical fp32 results; ragged M / N / K included."""
    A = _rand((K, M) if at else (M, K), 51)
    B = _rand((K, N) if bt else (N, K), 52)
    o_n, o_w = torch.zeros(M, N), torch.zeros(M, N)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_n, safe=3)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_w, safe=2)
    assert torch.equal(o_n, o_w)
    np.testing.assert_allclose(o_n.double().numpy(), _ref(A, B, at, bt).numpy(), rtol=1e-5, atol=1e-4)


BIG_TILE = [4, 6, 7, 16]   # OfGemmArgs.safe: 4 = 8-wave ping-pong LDS-DMA kernel, 6 / 7 = 4-wave 128x128-per-wave kernel (register staged / LDS-DMA),
                           # 16 = the 4-wave LDS-DMA kernel on 16x16x32 MFMAs


@pytest.mark.parametrize("big", BIG_TILE)
@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (256, 512, 192), (512, 256, 256), (256, 256, 320)])
def test_gemm_pingpong_matches_general_kernel(at, bt, M, N, K, big):
    """The 256x256 kernels (safe=4 / 6 force them) must agree with the general kernel (safe=2) in fp32
    (same products, fp32 accumulation; only the k order inside a 64-deep stage differs) and with fp64 matmul.
    K = 64 / 192 / 256: one stage (no loop), three (both tail iterations, no steady-state loop), four."""
    A = _rand((K, M) if at else (M, K), 11)
    B = _rand((K, N) if bt else (N, K), 12)
    ref = _ref(A, B, at, bt)
    o_fast, o_gen = torch.zeros(M, N), torch.zeros(M, N)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_fast, safe=big)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_gen, safe=2)
    np.testing.assert_allclose(o_fast.double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(o_fast.numpy(), o_gen.numpy(), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 320), (384, 128, 576)])
def test_gemm_mid_kernel_matches_general_kernel(at, bt, M, N, K):
    """The 8-wave LDS-DMA 128x128 kernel (safe = 5; of_gemm's own choice for tile-aligned shapes that do not fill the chip with
    256x256 tiles) vs the general kernel (safe = 2) and fp64.  K = 64 / 320 / 576: one stage (no steady state), five and nine
    stages (the four-slot ring wraps once / twice; every vmcnt tail case)."""
    A = _rand((K, M) if at else (M, K), 31)
    B = _rand((K, N) if bt else (N, K), 32)
    ref = _ref(A, B, at, bt)
    o_mid, o_gen = torch.zeros(M, N), torch.zeros(M, N)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_mid, safe=5)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_gen, safe=2)
    np.testing.assert_allclose(o_mid.double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(o_mid.numpy(), o_gen.numpy(), rtol=1e-6, atol=5e-5)      # k order inside a 64-deep stage differs
    ob = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_STORE_BF16, C_out=ob, safe=5, alpha=0.5)
    np.testing.assert_allclose(ob.double().numpy(), 0.5 * ref.numpy(), rtol=1e-2, atol=2e-2)


def test_gemm_mid_kernel_epilogues_and_auto_selection():
    M, N, K = 256, 128, 192
    A, B = _rand((M, K), 33), _rand((N, K), 34) * 0.1
    acc = _ref(A, B, 0, 0)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    for safe in (5, 0):           # forced, and of_gemm's own selection (2 tiles: not big-tile eligible -> the same kernel)
        b_out, a_out = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
        H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out, safe=safe)
        np.testing.assert_allclose(a_out.double().numpy(), acc.numpy(), rtol=1e-2, atol=1e-2)
        np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=1e-2)
        res = torch.randn(M, N)
        out = torch.zeros(M, N)
        H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=out, aux=res, gate=gate, io_f32=1, safe=safe)
        np.testing.assert_allclose(out.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
        resb, outb = res.to(torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
        H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=outb, aux=resb, gate=gate, io_f32=0, safe=safe)
        np.testing.assert_allclose(outb.double().numpy(), (resb.double() + g * acc).numpy(), rtol=1e-2, atol=2e-2)
        c = torch.randn(M, N)
        c0 = c.clone()
        H.gemm(A, B, epi=abi.EPI_ACC_F32, C_out=c, alpha=0.5, beta=1.0, gate=gate, safe=safe)
        np.testing.assert_allclose(c.double().numpy(), (c0.double() + 0.5 * g * acc).numpy(), rtol=1e-5, atol=1e-4)
        W = _rand((K, N), 35) * 0.2
        acc2 = A.double() @ W.double()
        aux = _rand((M, N), 36)
        for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
            o = torch.zeros(M, N, dtype=torch.bfloat16)
            dot = torch.full((1,), 3.0)                       # accumulates into what the scalar held
            H.gemm(A, W, b_trans=1, epi=epi, C_out=o, aux=aux, gate=gate, dot_out=dot, safe=safe)
            x = aux.double()
            if epi == abi.EPI_DGELU_DOT:
                xx = x.clone().requires_grad_(True)
                torch.nn.functional.gelu(xx).sum().backward()
                want, wdot = g * acc2 * xx.grad, (1 - g * g) * (torch.nn.functional.gelu(x) * acc2).sum()
            else:
                want, wdot = g * acc2, (1 - g * g) * (x * acc2).sum()
            np.testing.assert_allclose(o.double().numpy(), want.numpy(), rtol=1e-2, atol=2e-2)
            assert abs(float(dot) - 3.0 - float(wdot)) <= 1e-3 * abs(float(wdot)) + 1e-2


@pytest.mark.parametrize("safe", [2, 4, 5, 6, 7, 16])
def test_gate_gradient_dot_is_deterministic_and_needs_its_workspace(safe):
    """The *_DOT epilogues reduce the gate gradient from per-workgroup partials in a fixed order (no floating-point atomics):
    repeated launches give the same BITS whatever order the workgroups ran in (the emulator runs them on parallel host threads);
    without the partials workspace the call is refused instead of falling back to atomics."""
    M, N, K = 512, 512, 128
    A, W, aux = _rand((M, K), 41), _rand((K, N), 42) * 0.2, _rand((M, N), 43)
    gate = torch.tensor([0.3])
    vals = []
    for _ in range(3):
        o, dot = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(1)
        H.gemm(A, W, b_trans=1, epi=abi.EPI_DGELU_DOT, C_out=o, aux=aux, gate=gate, dot_out=dot, safe=safe)
        vals.append(dot.clone())
    assert torch.equal(vals[0], vals[1]) and torch.equal(vals[0], vals[2]) and float(vals[0]) != 0.0
    a = abi.OfGemmArgs()
    a.A, a.B, a.M, a.N, a.K, a.lda, a.ldb, a.b_trans, a.epi = A.data_ptr(), W.data_ptr(), M, N, K, K, N, 1, abi.EPI_DGELU_DOT
    o, dot = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(1)
    a.C, a.ldc, a.aux, a.ldaux, a.gate, a.dot_out, a.safe = o.data_ptr(), N, aux.data_ptr(), N, gate.data_ptr(), dot.data_ptr(), safe
    import ctypes
    assert H.lib().of_gemm(ctypes.byref(a), None) == -4     # OF_E_WORKSPACE
    assert H.lib().of_gemm_workspace_bytes(ctypes.byref(a)) >= 4 * (M // 128) * (N // 64)


def test_gemm_split_k():
    """Weight-gradient layout with a small output and deep K: of_gemm splits K and accumulates with fp32 atomics."""
    M, N, K = 128, 256, 2048
    A, B = _rand((K, M), 21), _rand((K, N), 22)
    ref = _ref(A, B, 1, 1)
    for beta in (0.0, 1.0):
        for ws in (None, torch.empty(16 * M * N)):       # fp32 atomics / deterministic slab reduction
            c0 = torch.randn(M, N)
            got = c0.clone()
            H.gemm(A, B, a_trans=1, b_trans=1, epi=abi.EPI_ACC_F32, C_out=got, beta=beta, alpha=0.5, workspace=ws)
            np.testing.assert_allclose(got.double().numpy(), (0.5 * ref + beta * c0.double()).numpy(), rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("big", BIG_TILE)
def test_gemm_pingpong_epilogues(big):
    M, N, K = 256, 256, 64
    A, B = _rand((M, K), 13), _rand((N, K), 14) * 0.1
    acc = _ref(A, B, 0, 0)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    b_out, a_out = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out, safe=big)
    np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=1e-2)
    res = torch.randn(M, N)
    out = torch.zeros(M, N)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=out, aux=res, gate=gate, io_f32=1, safe=big)
    np.testing.assert_allclose(out.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    # bf16 stream, and in place on a strided fp32 stream (C aliases aux, leading dimension > N): every group of every wave reads its
    # own residual tile (the 16x16x32 kernel fetches them by LDS-DMA, three groups deep, with its own LDS image)
    resb, outb = res.to(torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=outb, aux=resb, gate=gate, io_f32=0, safe=big)
    np.testing.assert_allclose(outb.double().numpy(), (resb.double() + g * acc).numpy(), rtol=1e-2, atol=2e-2)
    wide = torch.randn(M, N + 64)
    y = wide[:, 32:32 + N]
    y0 = y.clone()
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=y, aux=y, gate=gate, io_f32=1, safe=big)
    np.testing.assert_allclose(y.double().numpy(), (y0.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    W = _rand((K, N), 15) * 0.2
    acc2 = A.double() @ W.double()
    aux = _rand((M, N), 16)
    out = torch.zeros(M, N, dtype=torch.bfloat16)
    dot = torch.zeros(1)
    H.gemm(A, W, b_trans=1, epi=abi.EPI_DGELU_DOT, C_out=out, aux=aux, gate=gate, dot_out=dot, safe=big)
    xx = aux.double().clone().requires_grad_(True)
    torch.nn.functional.gelu(xx).sum().backward()
    np.testing.assert_allclose(out.double().numpy(), (g * acc2 * xx.grad).numpy(), rtol=1e-2, atol=2e-2)
    wdot = (1 - g * g) * (torch.nn.functional.gelu(aux.double()) * acc2).sum()
    assert abs(float(dot) - float(wdot)) <= 1e-3 * abs(float(wdot)) + 1e-2


@pytest.mark.parametrize("K", [128, 320])
def test_big_tile_residual_epilogue_behind_a_multi_stage_k_loop(K):
    """gemm_w4m.hip, round 5: GATE_RESID runs its K loop with the third image of B in the 32 KiB that used to hold the first residual
    tile of every wave, and requests that tile in the epilogue (gemm_w4_epi.h: PRE = false) -- two and five stages (the prologue with
    stage 1 in flight, the steady state and both tails of the three-image rotation), fp32 and bf16 stream, in place on a strided stream."""
    M, N = 256, 512
    A, B = _rand((M, K), 23), _rand((N, K), 24) * 0.1
    acc = _ref(A, B, 0, 0)
    gate = torch.tensor([0.41])
    g = float(torch.tanh(gate))
    res = torch.randn(M, N)
    out = torch.zeros(M, N)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=out, aux=res, gate=gate, io_f32=1, safe=16)
    np.testing.assert_allclose(out.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    resb, outb = res.to(torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=outb, aux=resb, gate=gate, io_f32=0, safe=16)
    np.testing.assert_allclose(outb.double().numpy(), (resb.double() + g * acc).numpy(), rtol=1e-2, atol=2e-2)
    wide = torch.randn(M, N + 64)
    y = wide[:, 32:32 + N]
    y0 = y.clone()
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=y, aux=y, gate=gate, io_f32=1, safe=16)
    np.testing.assert_allclose(y.double().numpy(), (y0.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    b_out, a_out = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out, safe=16)
    np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(a_out.double().numpy(), acc.numpy(), rtol=1e-2, atol=1e-2)


def test_fast_erf_gelu_accuracy():
    """The epilogues' erf-GELU: |gelu - exact| and |gelu' - exact| stay below bf16 resolution -- the scalar form of the general kernel
    (K = 32: Abramowitz-Stegun 7.1.26) and the packed-math forms of the tiled kernels (K = 64 -> the 128x128 LDS-DMA kernel: 7.1.28
    forward, 7.1.26 with a folded polynomial backward; of_platform.h)."""
    x = torch.linspace(-8, 8, 4001).to(torch.bfloat16)
    M = 4096
    want = torch.nn.functional.gelu(x.double())
    for K in (32, 64):
        A = torch.zeros(M, K, dtype=torch.bfloat16)
        A[:4001, 0] = x
        B = torch.zeros(256, K, dtype=torch.bfloat16)
        B[:, 0] = 1
        out = torch.zeros(M, 256, dtype=torch.bfloat16)
        H.gemm(A, B, epi=abi.EPI_GELU, C_out=out)
        got = out[:4001, 0].double()
        # one bf16 rounding of the result (2^-8 relative) + the <= 1e-6 absolute error of the erf approximations (x |a|)
        assert ((got - want).abs() <= 2.0 ** -8 * want.abs() + 2e-6).all(), K
    # the derivative through the DGELU_DOT epilogue: acc = 1 everywhere, aux = x  ->  out = gelu'(x)
    A1 = torch.zeros(M, 64, dtype=torch.bfloat16)
    A1[:, 0] = 1
    W = torch.zeros(64, 256, dtype=torch.bfloat16)
    W[0, :] = 1
    aux = torch.zeros(M, 256, dtype=torch.bfloat16)
    aux[:4001, :] = x[:, None]
    out, dot = torch.zeros(M, 256, dtype=torch.bfloat16), torch.zeros(1)
    H.gemm(A1, W, b_trans=1, epi=abi.EPI_DGELU_DOT, C_out=out, aux=aux, dot_out=dot)
    xd = x.double().requires_grad_(True)
    torch.nn.functional.gelu(xd).sum().backward()
    got = out[:4001, 7].double()
    assert ((got - xd.grad).abs() <= 2.0 ** -8 * xd.grad.abs() + 2e-6).all()
    # (the gate-gradient dot of these values is checked by test_gemm_dot_epilogues & co.: without a gate its factor 1 - tanh^2 is 0)


_SKINNY = [(M, 4, 8) for M in (1, 16)] + [(M, 36, 520) for M in (1, 2, 3, 5, 8, 13, 16)] + \
          [(M, 260, 2048 + 64) for M in (2, 8)] + [(M, 2052, 512) for M in (1, 4, 16)]


@pytest.mark.parametrize("M,N,K", _SKINNY)
def test_skinny_gemm_decode_shapes(M, N, K):
    """M <= 16 untransposed problems take the weight-streaming kernel (gemm_skinny.hip): every padded-row template,
    both rows-per-workgroup variants (N <= 2048 / > 2048), ragged N and K tails, all three epilogues -- against fp64
    and against the tile kernel (safe = 2) on the same operands."""
    A, B = _rand((M, K), 11 + M), _rand((N, K), 12 + N) * 0.1
    acc = _ref(A, B, 0, 0)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    for safe in (0, 2):
        out = torch.zeros(M, N, dtype=torch.bfloat16)
        H.gemm(A, B, C_out=out, alpha=0.5, safe=safe)
        np.testing.assert_allclose(out.double().numpy(), 0.5 * acc.numpy(), rtol=1e-2, atol=1e-2)
        b_out, a_out = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
        H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out, safe=safe)
        np.testing.assert_allclose(a_out.double().numpy(), acc.numpy(), rtol=1e-2, atol=1e-2)
        np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=1e-2)
        b_only = torch.zeros(M, N, dtype=torch.bfloat16)
        H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_only, safe=safe)                   # inference: no pre-activation output
        assert torch.equal(b_only, b_out)
        res = torch.randn(M, N)
        o32 = torch.zeros(M, N)
        H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=o32, aux=res, gate=gate, io_f32=1, safe=safe)
        np.testing.assert_allclose(o32.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
        resb, o16 = res.to(torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
        H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=o16, aux=resb, io_f32=0, safe=safe)   # no gate: plain residual
        np.testing.assert_allclose(o16.double().numpy(), (resb.double() + acc).numpy(), rtol=1e-2, atol=2e-2)


def test_skinny_gemm_strided_operands():
    """Leading dimensions larger than the logical widths (k | v halves, padded activations)."""
    M, N, K = 4, 40, 264
    Abig, Bbig = _rand((M, K + 24), 21), _rand((N, K + 40), 22) * 0.1
    A, B = Abig[:, 8:8 + K], Bbig[:, 16:16 + K]
    Cbig = torch.zeros(M, N + 16, dtype=torch.bfloat16)
    H.gemm(A, B, C_out=Cbig[:, 8:8 + N])
    np.testing.assert_allclose(Cbig[:, 8:8 + N].double().numpy(), _ref(A, B, 0, 0).numpy(), rtol=1e-2, atol=1e-2)
    assert Cbig[:, :8].abs().sum() == 0 and Cbig[:, 8 + N:].abs().sum() == 0


def test_tile_quantisation_n_split_applies_every_column_once():
    """of_gemm's N-split (gemm.hip: a grid whose last round of 256x256 tiles would be under half full -> whole rounds on the big
    tile + the remainder strip on the 128x128 kernel) with the ACCUMULATING epilogues (had the strip's launch been
    refused after the left part had run, the whole-problem launch behind it would have applied the left columns twice).  288 tiles:
    M = 8192, N = 2304 -> 8192 x 2048 (256 tiles) + 8192 x 256; K = 64.  Both halves are now checked before either is launched."""
    M, N, K = 8192, 2304, 64
    A = _rand((M, K), 61)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    # ACC_F32 with beta = 1, weight-gradient layout (TN): C += alpha g A^T B
    At, Bt = _rand((K, M), 62), _rand((K, N), 63)
    c0 = torch.randn(M, N)
    c = c0.clone()
    H.gemm(At, Bt, a_trans=1, b_trans=1, epi=abi.EPI_ACC_F32, C_out=c, alpha=0.5, beta=1.0, gate=gate)
    want = c0.double() + 0.5 * g * _ref(At, Bt, 1, 1)
    np.testing.assert_allclose(c.double().numpy(), want.numpy(), rtol=1e-5, atol=1e-4)
    # *_DOT: output once, the gate gradient = the sum over ALL columns exactly once (on top of what the scalar held)
    W, aux = _rand((K, N), 64) * 0.2, _rand((M, N), 65)
    acc = A.double() @ W.double()
    for epi in (abi.EPI_SCALE_DOT, abi.EPI_DGELU_DOT):
        o, dot = torch.zeros(M, N, dtype=torch.bfloat16), torch.full((1,), 3.0)
        H.gemm(A, W, b_trans=1, epi=epi, C_out=o, aux=aux, gate=gate, dot_out=dot)
        x = aux.double()
        if epi == abi.EPI_DGELU_DOT:
            xx = x.clone().requires_grad_(True)
            torch.nn.functional.gelu(xx).sum().backward()
            wo, wdot = g * acc * xx.grad, (1 - g * g) * (torch.nn.functional.gelu(x) * acc).sum()
        else:
            wo, wdot = g * acc, (1 - g * g) * (x * acc).sum()
        np.testing.assert_allclose(o.double().numpy(), wo.numpy(), rtol=1e-2, atol=2e-2)
        assert abs(float(dot) - 3.0 - float(wdot)) <= 1e-3 * abs(float(wdot)) + 1e-2
    # in-place gate + residual (C aliases aux)
    B = _rand((N, K), 66) * 0.1
    y = torch.randn(M, N)
    y0 = y.clone()
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=y, aux=y, gate=gate, io_f32=1)
    np.testing.assert_allclose(y.double().numpy(), (y0.double() + g * _ref(A, B, 0, 0)).numpy(), rtol=1e-5, atol=1e-4)


# ---- stream-K schedule of the 16x16x32 big-tile kernel (gemm_w4m.hip; OfGemmArgs.cu_limit, safe = 17 forces it) ------------------
# (tiles, K stages) over G = cu_limit workgroups:  4 tiles / 8: every tile shared by two workgroups;  3 tiles x 8 stages / 8: three
# (2 tiles x 2 stages / 8: half of the workgroups hold no unit at all;)  units per workgroup -> tiles shared by three, workgroups with a tail AND a head segment;  12 / 8: one whole round + 4 shared
# tiles;  16 / 16 and 8 / 8: whole rounds only (the persistent loop, no fix-up);  5 tiles x 3 stages / 8: ranges of 1 and 2 units.
_SK = [(512, 256, 128, 8), (512, 512, 256, 8), (768, 256, 512, 8), (768, 1024, 192, 8), (1024, 1024, 128, 16), (512, 1024, 64, 8), (1280, 256, 192, 8)]


@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K,G", _SK)
def test_stream_k_matches_one_tile_per_workgroup(at, bt, M, N, K, G):
    """Same products, fp32 accumulation; a shared tile's sum is split at the workgroup boundaries (own stages + the others' partial
    tiles, ascending K) -> equal to the classic launch (safe = 16) up to fp32 summation order, exactly equal where no tile is
    shared, and bit-reproducible run to run (fixed-order fix-up, no atomics)."""
    A = _rand((K, M) if at else (M, K), 71)
    B = _rand((K, N) if bt else (N, K), 72)
    ref = _ref(A, B, at, bt)
    o_sk, o_sk2, o_cl = torch.zeros(M, N), torch.zeros(M, N), torch.zeros(M, N)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_sk, safe=17, cu_limit=G)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_sk2, safe=17, cu_limit=G)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_cl, safe=16)
    assert torch.equal(o_sk, o_sk2)
    np.testing.assert_allclose(o_sk.double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(o_sk.numpy(), o_cl.numpy(), rtol=1e-6, atol=5e-5)
    if ((M // 256) * (N // 256)) % G == 0:
        assert torch.equal(o_sk, o_cl)


@pytest.mark.parametrize("M,N,K,G", [(768, 256, 512, 8), (768, 1024, 192, 8)])
def test_stream_k_epilogues(M, N, K, G):
    """Every fused epilogue behind a shared tile: the owner adds the partial tiles on the way through its LDS patch, then the
    epilogue runs as ever (residual tiles by LDS-DMA, *_DOT partial per TILE, beta = 1 accumulation, in-place gate + residual)."""
    A, B = _rand((M, K), 73), _rand((N, K), 74) * 0.1
    acc = _ref(A, B, 0, 0)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    kw = dict(safe=17, cu_limit=G)
    b_out, a_out = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out, **kw)
    np.testing.assert_allclose(a_out.double().numpy(), acc.numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=1e-2)
    ob = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_STORE_BF16, C_out=ob, alpha=0.5, **kw)
    np.testing.assert_allclose(ob.double().numpy(), 0.5 * acc.numpy(), rtol=1e-2, atol=2e-2)
    res = torch.randn(M, N)
    out = torch.zeros(M, N)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=out, aux=res, gate=gate, io_f32=1, **kw)
    np.testing.assert_allclose(out.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    y = res.clone()
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=y, aux=y, gate=gate, io_f32=1, **kw)          # in place
    np.testing.assert_allclose(y.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    resb, outb = res.to(torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=outb, aux=resb, gate=gate, io_f32=0, **kw)
    np.testing.assert_allclose(outb.double().numpy(), (resb.double() + g * acc).numpy(), rtol=1e-2, atol=2e-2)
    c = torch.randn(M, N)
    c0 = c.clone()
    H.gemm(A, B, epi=abi.EPI_ACC_F32, C_out=c, alpha=0.5, beta=1.0, gate=gate, **kw)
    np.testing.assert_allclose(c.double().numpy(), (c0.double() + 0.5 * g * acc).numpy(), rtol=1e-5, atol=1e-4)
    W = _rand((K, N), 75) * 0.2
    acc2 = A.double() @ W.double()
    aux = _rand((M, N), 76)
    for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        vals = []
        for _ in range(2):
            o, dot = torch.zeros(M, N, dtype=torch.bfloat16), torch.full((1,), 3.0)
            H.gemm(A, W, b_trans=1, epi=epi, C_out=o, aux=aux, gate=gate, dot_out=dot, **kw)
            vals.append(dot.clone())
        x = aux.double()
        if epi == abi.EPI_DGELU_DOT:
            xx = x.clone().requires_grad_(True)
            torch.nn.functional.gelu(xx).sum().backward()
            want, wdot = g * acc2 * xx.grad, (1 - g * g) * (torch.nn.functional.gelu(x) * acc2).sum()
        else:
            want, wdot = g * acc2, (1 - g * g) * (x * acc2).sum()
        np.testing.assert_allclose(o.double().numpy(), want.numpy(), rtol=1e-2, atol=2e-2)
        assert abs(float(dot) - 3.0 - float(wdot)) <= 1e-3 * abs(float(wdot)) + 1e-2
        assert torch.equal(vals[0], vals[1])


def test_stream_k_selection_rules():
    """safe = 0.  Sharing a tile costs a partial tile written and read back per workgroup: of_gemm goes stream-K only for K >= 4096
    (64 stages per tile) -- at K = 128, 144 tiles under cu_limit = 128 stay one tile per workgroup (bits of the classic launch),
    with or without a workspace; the workspace query names the stream-K region exactly when the launch would use it.  (of_gemm's
    own stream-K choice at real K runs on the GPU: tests/test_gpu_kernels.py::test_stream_k_matches_classic_launch_...)"""
    import ctypes
    M, N, K = 2304, 4096, 128            # 9 x 16 = 144 tiles
    A, B = _rand((M, K), 77), _rand((N, K), 78)
    o_cl, o_lim, o_nw = torch.zeros(M, N), torch.zeros(M, N), torch.zeros(M, N)
    H.gemm(A, B, epi=abi.EPI_ACC_F32, C_out=o_cl, safe=16)
    H.gemm(A, B, epi=abi.EPI_ACC_F32, C_out=o_lim, cu_limit=128)
    assert torch.equal(o_lim, o_cl)
    a = abi.OfGemmArgs()
    a.A, a.B, a.M, a.N, a.K, a.lda, a.ldb, a.epi = A.data_ptr(), B.data_ptr(), M, N, K, K, K, abi.EPI_ACC_F32
    a.C, a.ldc, a.alpha, a.cu_limit = o_nw.data_ptr(), N, 1.0, 128
    assert H.lib().of_gemm(ctypes.byref(a), None) == 0 and torch.equal(o_nw, o_cl)
    assert H.lib().of_gemm_workspace_bytes(ctypes.byref(a)) == 0
    a.K = a.lda = a.ldb = 8192           # (sizes only: nothing is launched)
    assert H.lib().of_gemm_workspace_bytes(ctypes.byref(a)) >= 128 * 256 * 256 * 4
    a.cu_limit = 144                     # divides the tile count: whole rounds, nothing shared
    assert H.lib().of_gemm_workspace_bytes(ctypes.byref(a)) == 0
    a.cu_limit = 0                       # 144 tiles / 256 workgroups
    assert H.lib().of_gemm_workspace_bytes(ctypes.byref(a)) >= 256 * 256 * 256 * 4


def test_gemm_batch_of_weight_gradients_equals_the_separate_launches():
    """of_gemm_batch (ABI v8): the three 512-wide weight gradients of a gated block -- to_q (512 x d), to_out (d x 512), to_kv (1024 x
    1024 over the media rows) -- as ONE GEMM grid + ONE reduce grid on the 128x128 kernel.  Every problem keeps the K split and the
    slab order of its own of_gemm launch: bit-identical results, beta = 0 and beta = 1, with a gate on one problem; a batch the form
    does not cover (a problem without a workspace) runs as separate launches with the same bits."""
    from open_flamingo_amd.hip.ops import BF16      # noqa: F401
    ops = H.emu_ops()
    rows, d, inner, mrows = 2048, 256, 128, 1024
    dq, xn = _rand((rows, inner), 81), _rand((rows, d), 82)
    dy, o = _rand((rows, d), 83), _rand((rows, inner), 84)
    dkv, media = _rand((mrows, 256), 85), _rand((mrows, 128), 86)
    gate = torch.tensor([0.4])
    probs = [(dq, xn, 0.0, None), (dy, o, 1.0, gate), (dkv, media, 0.0, None)]
    want = []
    for A, B, beta, g in probs:
        c = torch.ones(A.shape[1], B.shape[1])
        ws = torch.empty(16 * c.numel())
        H.gemm(A, B, a_trans=1, b_trans=1, epi=abi.EPI_ACC_F32, C_out=c, beta=beta, gate=g, workspace=ws)
        want.append(c)
    got = [torch.ones_like(w) for w in want]
    ops.gemm_batch_dw([(A, B, c, beta, g) for (A, B, beta, g), c in zip(probs, got)])
    for g_, w in zip(got, want):
        assert torch.equal(g_, w)
    np.testing.assert_allclose(got[0].double().numpy(), _ref(dq, xn, 1, 1).numpy(), rtol=1e-5, atol=2e-4)
    # not coverable as a batch (one problem is big-tile material): separate launches, same interface
    Abig, Bbig = _rand((64, 4096), 87), _rand((64, 8192), 88)
    cb, cs = torch.zeros(4096, 8192), torch.zeros(inner, d)
    ops.gemm_batch_dw([(Abig, Bbig, cb, 0.0, None), (dq, xn, cs, 0.0, None)])
    assert torch.equal(cs, want[0] * 0 + got[0]) and torch.allclose(cb.double(), _ref(Abig, Bbig, 1, 1), rtol=1e-5, atol=2e-4)


# ---- the two-workgroups-per-CU 256 x 128 kernel (gemm_w4h.hip; OfGemmArgs.safe = 18 forces it) ----------------------------------
@pytest.mark.parametrize("bt", [0, 1])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (256, 128, 128), (256, 256, 192), (512, 128, 256), (256, 384, 320), (256, 128, 448),
                                   (256, 128, 704)])
def test_half_tile_kernel_matches_the_big_tile_and_general_kernels(bt, M, N, K):
    """Ring of five 16-KiB units, a stage = three of them (B, A rows 0-63, A rows 64-127 of both wave rows): K = 64 ... 704 = 1 ... 11
    stages -- prologue only, each tail form, the steady state, and the unit -> slot map wrapped twice (period: 5 stages).  fp32
    accumulation in the big-tile kernel's order: equal to the general kernel up to summation order inside a stage, and BIT-EQUAL to
    the 256 x 256 kernel on 16x16x32 MFMAs (safe = 16) where that one takes the shape."""
    A = _rand((M, K), 81)
    B = _rand((K, N) if bt else (N, K), 82)
    ref = _ref(A, B, 0, bt)
    o_h, o_gen = torch.zeros(M, N), torch.zeros(M, N)
    H.gemm(A, B, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_h, safe=18)
    H.gemm(A, B, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_gen, safe=2)
    np.testing.assert_allclose(o_h.double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(o_h.numpy(), o_gen.numpy(), rtol=1e-6, atol=1e-5)
    if N % 256 == 0:
        o_big = torch.zeros(M, N)
        H.gemm(A, B, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o_big, safe=16)
        assert torch.equal(o_h, o_big)
    ob = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, b_trans=bt, epi=abi.EPI_STORE_BF16, C_out=ob, alpha=0.5, safe=18)
    np.testing.assert_allclose(ob.double().numpy(), 0.5 * ref.numpy(), rtol=1e-2, atol=2e-2)


def test_half_tile_kernel_strided_operands_and_tile_order():
    """Leading dimensions larger than the extents (views into wider buffers) and a grid of 4 x 3 tiles through the XCD-aware order."""
    M, N, K = 1024, 384, 128
    Aw, Bw, Cw = _rand((M, K + 64), 83), _rand((N, K + 128), 84), torch.zeros(M, N + 32)
    A, B, C = Aw[:, 64:], Bw[:, 128:], Cw[:, 32:]
    H.gemm(A, B, epi=abi.EPI_ACC_F32, C_out=C, safe=18)
    np.testing.assert_allclose(C.double().numpy(), _ref(A, B, 0, 0).numpy(), rtol=1e-5, atol=1e-4)
    assert float(Cw[:, :32].abs().max()) == 0.0
    Wt = _rand((K, N + 256), 85)
    W = Wt[:, 128:128 + N]
    C2 = torch.zeros(M, N)
    H.gemm(A, W, b_trans=1, epi=abi.EPI_ACC_F32, C_out=C2, safe=18)
    np.testing.assert_allclose(C2.double().numpy(), (A.double() @ W.double()).numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (512, 256, 384)])
def test_half_tile_kernel_epilogues(M, N, K):
    """Every fused epilogue of the half-tile kernel: four 32 x 64 groups per wave through the LDS patch, aux tiles of the *_DOT
    forms by LDS-DMA two groups deep, one gate-gradient partial per tile (deterministic), bit-equal to the 256 x 256 kernel's."""
    A, B = _rand((M, K), 86), _rand((N, K), 87) * 0.1
    acc = _ref(A, B, 0, 0)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    kw = dict(safe=18)
    big = N % 256 == 0
    b_out, a_out = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out, **kw)
    np.testing.assert_allclose(a_out.double().numpy(), acc.numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=1e-2)
    if big:
        b16, a16 = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
        H.gemm(A, B, epi=abi.EPI_GELU, C_out=b16, C2=a16, safe=16)
        assert torch.equal(b16, b_out) and torch.equal(a16, a_out)
    b_only = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_only, **kw)          # no pre-activation output
    assert torch.equal(b_only, b_out)
    res = torch.randn(M, N)
    out = torch.zeros(M, N)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=out, aux=res, gate=gate, io_f32=1, **kw)
    np.testing.assert_allclose(out.double().numpy(), (res.double() + g * acc).numpy(), rtol=1e-5, atol=1e-4)
    y = res.clone()
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=y, aux=y, gate=gate, io_f32=1, **kw)          # in place
    assert torch.equal(y, out)
    resb, outb = res.to(torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GATE_RESID, C_out=outb, aux=resb, gate=gate, io_f32=0, **kw)
    np.testing.assert_allclose(outb.double().numpy(), (resb.double() + g * acc).numpy(), rtol=1e-2, atol=2e-2)
    c = torch.randn(M, N)
    c0 = c.clone()
    H.gemm(A, B, epi=abi.EPI_ACC_F32, C_out=c, alpha=0.5, beta=1.0, gate=gate, **kw)
    np.testing.assert_allclose(c.double().numpy(), (c0.double() + 0.5 * g * acc).numpy(), rtol=1e-5, atol=1e-4)
    W = _rand((K, N), 88) * 0.2
    acc2 = A.double() @ W.double()
    aux = _rand((M, N), 89)
    for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        vals = []
        for _ in range(2):
            o, dot = torch.zeros(M, N, dtype=torch.bfloat16), torch.full((1,), 3.0)
            H.gemm(A, W, b_trans=1, epi=epi, C_out=o, aux=aux, gate=gate, dot_out=dot, **kw)
            vals.append(dot.clone())
        x = aux.double()
        if epi == abi.EPI_DGELU_DOT:
            xx = x.clone().requires_grad_(True)
            torch.nn.functional.gelu(xx).sum().backward()
            want, wdot = g * acc2 * xx.grad, (1 - g * g) * (torch.nn.functional.gelu(x) * acc2).sum()
        else:
            want, wdot = g * acc2, (1 - g * g) * (x * acc2).sum()
        np.testing.assert_allclose(o.double().numpy(), want.numpy(), rtol=1e-2, atol=2e-2)
        assert abs(float(dot) - 3.0 - float(wdot)) <= 1e-3 * abs(float(wdot)) + 1e-2
        assert torch.equal(vals[0], vals[1])
        if big:
            o16, d16 = torch.zeros(M, N, dtype=torch.bfloat16), torch.full((1,), 3.0)
            H.gemm(A, W, b_trans=1, epi=epi, C_out=o16, aux=aux, gate=gate, dot_out=d16, safe=16)
            assert torch.equal(o16, o)


# ---- the persistent wave-specialised 256 x 128 kernel (gemm_w4s.hip; OfGemmArgs.safe = 19 forces it) -----------------------------
@pytest.mark.parametrize("bt", [0, 1])
@pytest.mark.parametrize("M,N,K,cus", [(256, 128, 1152, 0), (256, 256, 1216, 1), (512, 256, 1152, 3), (256, 384, 1280, 2)])
def test_specialised_kernel_plain_store_is_bit_equal_to_the_big_tile_kernel(bt, M, N, K, cus):
    """Four MFMA waves + four producer waves per workgroup; one workgroup walks several tiles (cu_limit = workgroups): a tile's
    accumulators are rounded to bf16 into the consumer's LDS image at the end of its K loop and stored by its producer, 8 rows per
    stage, during the first 16 stages of the NEXT tile's K loop (the last tile's by a drain loop).  K = 1152 / 1216 / 1280: 18 (the
    minimum: 16 chunk stages + both tail forms), 19 and 20 stages through the two-stage ring of six units."""
    A = _rand((M, K), 91)
    B = _rand((K, N) if bt else (N, K), 92)
    ref = _ref(A, B, 0, bt)
    o_p = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, b_trans=bt, epi=abi.EPI_STORE_BF16, C_out=o_p, alpha=0.5, safe=19, cu_limit=cus)
    np.testing.assert_allclose(o_p.double().numpy(), 0.5 * ref.numpy(), rtol=1e-2, atol=2e-2)
    o_gen = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, b_trans=bt, epi=abi.EPI_STORE_BF16, C_out=o_gen, alpha=0.5, safe=18)
    assert torch.equal(o_p, o_gen)


def test_specialised_kernel_gelu_and_strided_outputs():
    M, N, K = 512, 256, 1152
    A, B = _rand((M, K), 93), _rand((N, K), 94) * 0.05
    wide_b, wide_a = torch.zeros(M, N + 64, dtype=torch.bfloat16), torch.zeros(M, N + 64, dtype=torch.bfloat16)
    b_out, a_out = wide_b[:, 32:32 + N], wide_a[:, 32:32 + N]
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_out, C2=a_out, safe=19, cu_limit=3)
    a16, b16 = torch.zeros(M, N, dtype=torch.bfloat16), torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b16, C2=a16, safe=16)
    assert torch.equal(a_out, a16)                      # the pre-activation: the rounded product, bit for bit
    # GELU of the ROUNDED product (the reference's order under autocast) vs the 256x256 kernel's GELU of the fp32 product: one bf16 ulp
    np.testing.assert_allclose(b_out.double().numpy(), torch.nn.functional.gelu(a_out.double()).numpy(), rtol=1e-2, atol=1e-3)
    np.testing.assert_allclose(b_out.double().numpy(), b16.double().numpy(), rtol=2e-2, atol=2e-3)
    assert float(wide_b[:, :32].abs().max()) == 0.0 and float(wide_b[:, 32 + N:].abs().max()) == 0.0
    b_only = torch.zeros(M, N, dtype=torch.bfloat16)
    H.gemm(A, B, epi=abi.EPI_GELU, C_out=b_only, safe=19, cu_limit=1)          # no pre-activation output, one workgroup, four tiles
    assert torch.equal(b_only, b_out.contiguous())


def test_specialised_kernel_dot_epilogues():
    """*_DOT: the producer reads the saved activation chunk by chunk, the gate-gradient partial is per (tile, wave) -- deterministic,
    equal to the 256x256 kernel's up to the order of the partials; SCALE_DOT's output is bit-equal (one rounding less matters only
    where the scale is not a power of two: gate = atanh(0.5), alpha = 1)."""
    M, N, K = 512, 256, 1152
    A, W = _rand((M, K), 95), _rand((K, N), 96) * 0.1
    acc2 = A.double() @ W.double()
    aux = _rand((M, N), 97)
    gate = torch.tensor([0.37])
    g = float(torch.tanh(gate))
    for epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        vals, outs = [], []
        for cus in (3, 1):
            o, dot = torch.zeros(M, N, dtype=torch.bfloat16), torch.full((1,), 3.0)
            H.gemm(A, W, b_trans=1, epi=epi, C_out=o, aux=aux, gate=gate, dot_out=dot, safe=19, cu_limit=cus)
            vals.append(dot.clone())
            outs.append(o)
        assert torch.equal(vals[0], vals[1]) and torch.equal(outs[0], outs[1])
        x = aux.double()
        if epi == abi.EPI_DGELU_DOT:
            xx = x.clone().requires_grad_(True)
            torch.nn.functional.gelu(xx).sum().backward()
            want, wdot = g * acc2 * xx.grad, (1 - g * g) * (torch.nn.functional.gelu(x) * acc2).sum()
        else:
            want, wdot = g * acc2, (1 - g * g) * (x * acc2).sum()
        np.testing.assert_allclose(outs[0].double().numpy(), want.numpy(), rtol=2e-2, atol=2e-2)
        # the dot is taken of the bf16-ROUNDED product (what the reference's autocast matmul hands its autograd): 2^-9 relative noise
        # per term, random sign -> ~2^-9 |term| sqrt(n) in the sum; the bound is on the terms' mass, as in tests/test_gpu_kernels.py
        terms = ((torch.nn.functional.gelu(x) if epi == abi.EPI_DGELU_DOT else x) * acc2).abs().sum()
        assert abs(float(vals[0]) - 3.0 - float(wdot)) <= 3e-5 * (1 - g * g) * float(terms) + 1e-2
        o16, d16 = torch.zeros(M, N, dtype=torch.bfloat16), torch.full((1,), 3.0)
        H.gemm(A, W, b_trans=1, epi=epi, C_out=o16, aux=aux, gate=gate, dot_out=d16, safe=16)
        np.testing.assert_allclose(outs[0].double().numpy(), o16.double().numpy(), rtol=2e-2, atol=2e-3)


# ---- OfGemmArgs.sumsq_out (ABI v9): a weight gradient's share of the global gradient norm leaves with the GEMM that produces it -------
@pytest.mark.parametrize("at,bt", [(0, 0), (1, 1)])      # K-contiguous and K-strided stage steps (the third layout mixes the two: hardware test)
def test_k_rotation_of_self_selected_big_tile_launches(at, bt):
    """A launch of_gemm selects itself (safe = 0, >= 128 big tiles) rotates its K loop per XCD (gemm_w4m.hip: w4m_rotation -- stage
    x * (stages / 8) first on the workgroups with block id & 7 == x, wrapping behind the last stage); the forced kernel (safe = 16) walks
    0, 1, 2, ...  Nine stages (8 does not divide them): both are the product to fp32 summation order, and they are NOT the same bits
    (the rotation is on; every stage is visited exactly once -- a skipped or doubled stage is off by a whole stage's products)."""
    M, N, K = 2048, 4096, 576
    A = _rand((K, M) if at else (M, K), 71)
    B = _rand((K, N) if bt else (N, K), 72)
    ref = _ref(A, B, at, bt)
    o0, o16 = torch.zeros(M, N), torch.zeros(M, N)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o0, safe=0)
    H.gemm(A, B, a_trans=at, b_trans=bt, epi=abi.EPI_ACC_F32, C_out=o16, safe=16)
    scale = float(ref.abs().max())
    assert float((o0.double() - ref).abs().max()) < 2e-6 * scale
    assert float((o16.double() - ref).abs().max()) < 2e-6 * scale
    assert not torch.equal(o0, o16)


def test_weight_gradient_gemm_emits_its_sum_of_squares_per_tile():
    """TN, OF_EPI_ACC_F32, 128 big tiles (the smallest launch of_gemm gives to the 256x256 kernel): one fp32 partial per tile = the sum
    of squares of the FINAL values (alpha, tanh(gate) and beta * old applied), written not added; launches that are not a single
    big-tile launch leave the slots untouched and of_gemm_sumsq_slots() says so beforehand."""
    import ctypes as C
    M, N, K = 2048, 4096, 64
    A, B = _rand((K, M), 101), _rand((K, N), 102)
    gate = torch.tensor([0.37])
    c = torch.randn(M, N)
    want = c.double() + 0.5 * float(torch.tanh(gate)) * _ref(A, B, 1, 1)
    slots = torch.full((200,), -1.0)
    H.gemm(A, B, a_trans=1, b_trans=1, epi=abi.EPI_ACC_F32, C_out=c, alpha=0.5, beta=1.0, gate=gate, sumsq=slots)
    np.testing.assert_allclose(c.double().numpy(), want.numpy(), rtol=1e-5, atol=1e-4)
    per_tile = c.double().view(M // 256, 256, N // 256, 256).pow(2).sum((1, 3)).reshape(-1)        # m-major tile order
    np.testing.assert_allclose(slots[:128].double().numpy(), per_tile.numpy(), rtol=1e-5)
    assert bool((slots[128:] == -1.0).all())
    # not honoured: split along K (small output), a bf16-store launch, a forced kernel -- the query says 0 and nothing is written
    a = abi.OfGemmArgs()
    A2, B2, c2 = _rand((512, 256), 103), _rand((512, 256), 104), torch.zeros(256, 256)
    a.A, a.B, a.C, a.M, a.N, a.K, a.lda, a.ldb, a.ldc = A2.data_ptr(), B2.data_ptr(), c2.data_ptr(), 256, 256, 512, 256, 256, 256
    a.a_trans, a.b_trans, a.epi = 1, 1, abi.EPI_ACC_F32
    assert H.lib().of_gemm_sumsq_slots(C.byref(a)) == 0
    s2 = torch.full((4,), -1.0)
    H.gemm(A2, B2, a_trans=1, b_trans=1, epi=abi.EPI_ACC_F32, C_out=c2, sumsq=s2)
    assert bool((s2 == -1.0).all())
    np.testing.assert_allclose(c2.double().numpy(), _ref(A2, B2, 1, 1).numpy(), rtol=1e-5, atol=1e-4)
