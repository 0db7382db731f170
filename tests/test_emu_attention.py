"""Attention-core kernels on the host SIMT emulator vs the fp64 dense reference (tests/attn_reference.py).
Covers: Perceiver shape (no mask, Lk not a multiple of 64), every mask case of SURVEY 8c KAT-3 (zero rows,
image at last position, consecutive images, more <image> tokens than images -> uniform rows, 'ge' variant),
ragged Lq, fused-kv strided views, tr-read path vs scalar path."""
import numpy as np
import pytest
import torch

from tests.attn_reference import dense_attention
from tests.emu import harness as H


def _r(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def _run(q, k, v, heads, tt=None, n=0, T=0, only_imm=1, safe=0, dout=None):
    B, Lq, _ = q.shape
    o = torch.full_like(q, float("nan"))
    lse = torch.full((B, heads, Lq), float("nan"))
    tt32 = tt.to(torch.int32).contiguous() if tt is not None else None
    a = H.attn_args(q, k, v, o, lse, tt32, n, T, only_imm, heads=heads, safe=safe)
    H.attn_fwd(a)
    res = {"o": o, "lse": lse}
    if dout is not None:
        dq = torch.full_like(q, float("nan"))
        dk = torch.full((B, k.shape[1], heads * 64), float("nan"), dtype=torch.bfloat16)
        dv = torch.full_like(dk, float("nan"))
        delta = torch.zeros(B, heads, Lq)
        a = H.attn_args(q, k, v, o, lse, tt32, n, T, only_imm, heads=heads, safe=safe, dout=dout, dq=dq, dk=dk, dv=dv,
                        delta=delta)
        H.attn_bwd(a)
        res.update(dq=dq, dk=dk, dv=dv)
    return res


def _check(q, k, v, heads, tt=None, n=0, T=0, only_imm=1, safe=0, tol=2e-2):
    dout = _r(q.shape, 99)
    res = _run(q, k, v, heads, tt, n, T, only_imm, safe, dout)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = dense_attention(qd, kd, vd, heads, tt, n, T, bool(only_imm))
    ref.backward(dout.double())
    for name, got, want in (("o", res["o"], ref.detach()), ("dq", res["dq"], qd.grad), ("dk", res["dk"], kd.grad),
                            ("dv", res["dv"], vd.grad)):
        got = got.double()
        assert torch.isfinite(got).all(), name
        scale = want.abs().max().item() + 1e-6
        err = (got - want).abs().max().item()
        assert err <= tol * scale, f"{name}: err {err:.3e} scale {scale:.3e}"
    return res


@pytest.mark.parametrize("safe", [0, 1, 2, 3])  # 0: of_attn's own choice; 1: tiled kernels, scalar-LDS path; 2: tiled kernels; 3: resident-K/V forward
def test_perceiver_shape(safe):
    # 2 media, 2 heads, 64 latent queries, 96 keys (tail block half empty)
    q, k, v = _r((2, 64, 128), 1), _r((2, 96, 128), 2), _r((2, 96, 128), 3)
    _check(q, k, v, 2, safe=safe)


def test_perceiver_fused_kv_view_and_ragged_queries():
    kv = _r((1, 80, 256), 4)          # [k | v] fused, 2 heads
    q = _r((1, 40, 128), 5)           # Lq not a multiple of 16
    _check(q, kv[..., :128], kv[..., 128:], 2)


@pytest.mark.parametrize("dh", [64, 128])
def test_self_attention_vit_like_ragged(dh):
    """Non-causal self-attention with L = 4 blocks + 1 token (CLIP's 257 = 256 patches + class token, scaled down): the
    resident forward keeps a 32-row tail block and skips its empty sub-tiles; must equal the tiled kernel bit for bit."""
    heads, B, L = 2, 2, 81
    q, k, v = _r((B, L, heads * dh), 41), _r((B, L, heads * dh), 42), _r((B, L, heads * dh), 43)
    outs = []
    for safe in (3, 2):
        o = torch.full_like(q, float("nan"))
        lse = torch.full((B, heads, L), float("nan"))
        H.attn_fwd(H.attn_args(q, k, v, o, lse, heads=heads, safe=safe, head_dim=dh))
        outs.append((o, lse))
    ref = dense_attention(q.double(), k.double(), v.double(), heads, head_dim=dh)
    assert (outs[0][0].double() - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_resident_forward_progressive_causal_loading():
    """Causal self-attention with Lq == Lk = 4 key blocks + a ragged tail at head dim 128 (8 waves: two key blocks become
    visible per step) and head dim 64 (4 waves: one per step): the resident forward's progressive LDS-DMA schedule must
    reproduce the tiled kernel bit for bit, with and without right padding."""
    for dh, L in ((128, 300), (64, 272)):
        heads, B = 2, 2
        q, k, v = _r((B, L, heads * dh), 61), _r((B, L, heads * dh), 62), _r((B, L, heads * dh), 63)
        slopes = torch.tensor([0.25, 0.03125])
        outs = []
        for safe in (3, 2):
            o = torch.full_like(q, float("nan"))
            lse = torch.full((B, heads, L), float("nan"))
            H.attn_fwd(H.attn_args(q, k, v, o, lse, heads=heads, safe=safe, head_dim=dh, causal=1, alibi_slopes=slopes))
            outs.append((o, lse))
        assert torch.isfinite(outs[0][0].float()).all()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        ref = dense_attention(q.double(), k.double(), v.double(), heads, head_dim=dh, causal=True, alibi_slopes=slopes)
        assert (outs[0][0].double() - ref).abs().max() <= 2e-2 * ref.abs().max()


CASES = {
    "basic": ([[1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0] * 6, [1] + [0] * 40 + [1] + [0] * 30], 3, 1),
    "before_first_image": ([[0, 0, 0, 1] + [0] * 68, [0] * 72], 2, 1),
    "image_last_and_consecutive": ([[1, 1] + [0] * 69 + [1], [0, 1, 1, 1] + [0] * 68], 3, 1),
    "more_image_tokens_than_images": ([[1, 0, 1, 0, 1, 0, 1, 0] * 9, [1, 1, 1, 1] + [0] * 68], 3, 1),
    "single_image": ([[1] + [0] * 71, [1] + [0] * 71], 1, 1),
    "attend_all_previous": ([[1, 0, 0, 0, 1, 0, 0, 1] * 9, [0, 0, 1] + [0] * 69], 3, 0),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_masked_cases(case):
    locs, T, only_imm = CASES[case]
    n = 16
    ml = torch.tensor(locs, dtype=torch.bool)
    tt = ml.cumsum(-1)
    B, L = ml.shape
    q, k, v = _r((B, L, 128), 6), _r((B, T * n, 128), 7), _r((B, T * n, 128), 8)
    res = _check(q, k, v, 2, tt, n, T, only_imm)
    if only_imm:
        zero_rows = (tt == 0)
        assert (res["o"][zero_rows] == 0).all()
        assert torch.isinf(res["lse"].transpose(1, 2)[zero_rows]).all()


def test_text_time_kernel():
    import ctypes as C
    g = torch.Generator().manual_seed(0)
    ml = (torch.rand(3, 150, generator=g) < 0.1)
    tt = torch.zeros(3, 150, dtype=torch.int32)
    m8 = ml.to(torch.uint8).contiguous()
    rc = H.lib().of_text_time(H.ptr(m8), H.ptr(tt), 3, 150, 150, 0, None)
    assert rc == 0
    assert torch.equal(tt.long(), ml.cumsum(-1))
    tt2 = torch.zeros(3, 7, dtype=torch.int32)
    rc = H.lib().of_text_time(H.ptr(m8), H.ptr(tt2), 3, 150, 7, 1, None)
    assert rc == 0
    assert torch.equal(tt2.long(), ml.sum(-1, keepdim=True).expand(-1, 7))


@pytest.mark.parametrize("dh,safe", [(128, 0), (128, 1), (128, 3), (64, 0), (64, 3)])
@pytest.mark.parametrize("Lq,Lk", [(96, 96), (40, 104)])
def test_causal_alibi_self_attention(dh, safe, Lq, Lk):
    """Causal self-attention with ALiBi (the frozen MPT blocks): head dim 128 and 64, ragged lengths, Lq < Lk
    (queries aligned to the END of the keys, as with a KV prefix)."""
    heads, B = 2, 2
    q, k, v = _r((B, Lq, heads * dh), 31), _r((B, Lk, heads * dh), 32), _r((B, Lk, heads * dh), 33)
    slopes = torch.tensor([0.5, 0.0625])
    dout = _r(q.shape, 34)
    o = torch.full_like(q, float("nan"))
    lse = torch.full((B, heads, Lq), float("nan"))
    kw = dict(heads=heads, safe=safe, head_dim=dh, causal=1, alibi_slopes=slopes)
    H.attn_fwd(H.attn_args(q, k, v, o, lse, **kw))
    dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    delta = torch.zeros(B, heads, Lq)
    H.attn_bwd(H.attn_args(q, k, v, o, lse, dout=dout, dq=dq, dk=dk, dv=dv, delta=delta, **kw))
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = dense_attention(qd, kd, vd, heads, head_dim=dh, causal=True, alibi_slopes=slopes)
    ref.backward(dout.double())
    for name, got, want in (("o", o, ref.detach()), ("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        got = got.double()
        assert torch.isfinite(got).all(), name
        err = (got - want).abs().max().item()
        assert err <= 2e-2 * (want.abs().max().item() + 1e-6), f"{name}: {err:.3e}"


def _bwd_both_forms(q, k, v, heads, dh, **kw):
    """forward once (tiled), then the backward as two passes (safe = 2) and as the single pass (safe = 3)"""
    B, Lq, _ = q.shape
    o = torch.full_like(q, float("nan"))
    lse = torch.full((B, heads, Lq), float("nan"))
    H.attn_fwd(H.attn_args(q, k, v, o, lse, heads=heads, safe=2, head_dim=dh, **kw))
    dout = _r(q.shape, 77)
    outs = []
    for safe in (2, 3):
        dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
        delta = torch.zeros(B, heads, Lq)
        H.attn_bwd(H.attn_args(q, k, v, o, lse, dout=dout, dq=dq, dk=dk, dv=dv, delta=delta, heads=heads, safe=safe, head_dim=dh, **kw))
        outs.append((dq, dk, dv))
    return dout, outs


@pytest.mark.parametrize("dh,L", [(128, 256), (64, 256), (128, 200), (64, 136)])
def test_single_pass_backward_equals_the_two_passes(dh, L):
    """attn_bwd_res.hip (one workgroup per (batch, head): dQ, dK, dV from one recomputation of P) against the two-pass kernels at the
    frozen MPT blocks' shape (causal, ALiBi, L = 256, head 128) and at ragged lengths: dV accumulates over the query tiles in the same
    order with the same bf16 P (bit-equal); delta = rowsum(dO o O) is summed in another fp32 order, so a few dS round the other way
    (dK, dQ: 1 bf16 ulp), and dQ sums dS K over the keys in another order."""
    heads, B = 2, 1
    q, k, v = _r((B, L, heads * dh), 81), _r((B, L, heads * dh), 82), _r((B, L, heads * dh), 83)
    slopes = torch.tensor([0.25, 0.015625])
    dout, (two, one) = _bwd_both_forms(q, k, v, heads, dh, causal=1, alibi_slopes=slopes)
    for name, a, b in zip(("dq", "dk", "dv"), one, two):
        assert torch.isfinite(a.float()).all(), name
        err = (a.float() - b.float()).abs().max().item()
        assert err <= 8e-3 * b.float().abs().max().item(), (name, err)
    assert torch.equal(one[2], two[2])
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = dense_attention(qd, kd, vd, heads, head_dim=dh, causal=True, alibi_slopes=slopes)
    ref.backward(dout.double())
    for name, got, want in (("dq", one[0], qd.grad), ("dk", one[1], kd.grad), ("dv", one[2], vd.grad)):
        err = (got.double() - want).abs().max().item()
        assert err <= 2e-2 * (want.abs().max().item() + 1e-6), f"{name}: {err:.3e}"


def test_single_pass_backward_key_lengths_prefix_and_no_mask():
    """right-padded sequences (kv_len), queries aligned to the end of a longer key sequence, and no mask at all"""
    heads, B, dh = 2, 3, 128
    q, k, v = _r((B, 96, heads * dh), 91), _r((B, 96, heads * dh), 92), _r((B, 96, heads * dh), 93)
    kv_len = torch.tensor([96, 50, 17], dtype=torch.int32)
    _, (two, one) = _bwd_both_forms(q, k, v, heads, dh, causal=1, alibi_slopes=torch.tensor([0.5, 0.0625]), kv_len=kv_len)
    for a, b in zip(one, two):
        assert torch.isfinite(a.float()).all()
        assert (a.float() - b.float()).abs().max() <= 8e-3 * b.float().abs().max()
    assert (one[1][1, 50:] == 0).all() and (one[2][2, 17:] == 0).all()       # padding keys get no gradient
    q2 = _r((B, 72, heads * dh), 94)
    for kw in (dict(causal=1), dict()):                                      # Lq < Lk causal (KV prefix); no mask (ViT-like)
        _, (two, one) = _bwd_both_forms(q2, k, v, heads, dh, **kw)
        for a, b in zip(one, two):
            assert torch.isfinite(a.float()).all()
            assert (a.float() - b.float()).abs().max() <= 8e-3 * b.float().abs().max()


def test_single_pass_backward_is_not_taken_with_text_time_or_long_sequences():
    """of_attn_bwd's selection: media windows (text_time) and L > 256 stay on the two-pass kernels even under safe = 3"""
    heads, dh, L = 1, 64, 320
    q, k, v = _r((1, L, dh), 95), _r((1, L, dh), 96), _r((1, L, dh), 97)
    _, (two, one) = _bwd_both_forms(q, k, v, heads, dh, causal=1)
    for a, b in zip(one, two):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ compact heads (ABI v11: OfAttnArgs.head_valid)
def _compact_case(hv, dh, B, Lq, Lk, heads, seed, **kw):
    """heads of hv < dh columns side by side (GPT-NeoX head size 80 at the 128-wide kernels): every kernel form, forward + backward,
    against the fp64 dense reference at head_dim = hv and against the SAME kernels on zero-padded copies (bit for bit: the padded
    columns contribute exact zeros to every sum)."""
    q, k, v = _r((B, Lq, heads * hv), seed), _r((B, Lk, heads * hv), seed + 1), _r((B, Lk, heads * hv), seed + 2)
    dout = _r(q.shape, seed + 3)
    scale = hv ** -0.5

    def pad(t):
        p = torch.zeros(t.shape[0], t.shape[1], heads, dh, dtype=t.dtype)
        p[..., :hv] = t.view(t.shape[0], t.shape[1], heads, hv)
        return p.view(t.shape[0], t.shape[1], heads * dh)

    def unpad(t):
        return t.view(t.shape[0], t.shape[1], heads, dh)[..., :hv].reshape(t.shape[0], t.shape[1], heads * hv)

    def run(q, k, v, dout, safe_f, safe_b, head_valid):
        o = torch.full_like(q, float("nan"))
        lse = torch.full((B, heads, Lq), float("nan"))
        common = dict(heads=heads, head_dim=dh, head_valid=head_valid, scale=scale, **kw)
        H.attn_fwd(H.attn_args(q, k, v, o, lse, safe=safe_f, **common))
        dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
        delta = torch.zeros(B, heads, Lq)
        H.attn_bwd(H.attn_args(q, k, v, o, lse, dout=dout, dq=dq, dk=dk, dv=dv, delta=delta, safe=safe_b, **common))
        return o, lse, dq, dk, dv

    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = dense_attention(qd, kd, vd, heads, head_dim=hv, causal=bool(kw.get("causal")), alibi_slopes=kw.get("alibi_slopes"))
    ref.backward(dout.double())
    want = dict(o=ref.detach(), dq=qd.grad, dk=kd.grad, dv=vd.grad)
    # tiled + two passes; of_attn's choice; resident forward + single-pass backward (compact: instantiated for 80 and 96 of 128 columns --
    # any other width takes the two passes, which agree with the single pass to a bf16 ulp, not bit for bit)
    single = dh == 128 and hv in (80, 96)
    forms = [(2, 2), (0, 0)] + ([(3, 3 if single else 2)] if Lq <= 256 and Lk <= 256 else [])
    for safe_f, safe_b in forms:
        o, lse, dq, dk, dv = run(q, k, v, dout, safe_f, safe_b, hv)
        for name, got in (("o", o), ("dq", dq), ("dk", dk), ("dv", dv)):
            assert torch.isfinite(got.float()).all(), (name, safe_f)
            err = (got.double() - want[name]).abs().max().item()
            assert err <= 2e-2 * (want[name].abs().max().item() + 1e-6), f"{name} safe {safe_f}: {err:.3e}"
        po, plse, pdq, pdk, pdv = run(pad(q), pad(k), pad(v), pad(dout), safe_f, safe_b, 0)
        assert torch.equal(lse, plse)
        for name, got, padded in (("o", o, po), ("dq", dq, pdq), ("dk", dk, pdk), ("dv", dv, pdv)):
            assert torch.equal(got, unpad(padded)), (name, safe_f)


@pytest.mark.parametrize("hv,dh", [(80, 128), (40, 64)])
def test_compact_heads_causal_self_attention(hv, dh):
    """GPT-NeoX-like: causal, no bias, ragged length (tail block of the resident images half empty), 3 heads (an odd head starts at a
    column that is only 16-byte aligned)"""
    _compact_case(hv, dh, B=2, Lq=104, Lk=104, heads=3, seed=300, causal=1)


def test_compact_heads_alibi_and_no_mask():
    _compact_case(96, 128, B=1, Lq=72, Lk=136, heads=2, seed=310, causal=1, alibi_slopes=torch.tensor([0.5, 0.0625]))
    _compact_case(80, 128, B=1, Lq=64, Lk=96, heads=2, seed=320)


def test_compact_heads_are_refused_where_they_do_not_apply():
    q, k, v = _r((1, 64, 160), 330), _r((1, 64, 160), 331), _r((1, 64, 160), 332)
    o, lse = torch.zeros_like(q), torch.zeros(1, 2, 64)
    for bad in dict(head_valid=84), dict(head_valid=4), dict(head_valid=136), dict(head_valid=80, safe=1):
        a = H.attn_args(q, k, v, o, lse, heads=2, head_dim=128, **bad)
        assert H.lib().of_attn_fwd(H.C.byref(a), None) != 0, bad
