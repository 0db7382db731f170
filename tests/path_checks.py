"""Shared parity checks of the composed hot path (open_flamingo_amd.hip.path) against the oracle.
Used with the emulator Ops on CPU (tests/test_emu_path.py) and with the real library on the GPU
(tests/test_gpu_path.py).  Two families of checks:

* ``check_xattn`` / ``check_perceiver``: max-abs error relative to the max-abs of the reference tensor, against the
  ROUNDING-POINT oracle (bf16 operand rounding emulated): forward 1e-2, gradients 3e-2 (the backward's bf16
  intermediates are not emulated by the oracle).  Tight, catches wiring / index bugs at small sizes.
* ``check_xattn_8c`` / ``check_perceiver_8c``: the criterion SURVEY.md 8c states, against the plain **fp32** oracle:
  forward error (relative L2 and max-abs) <= 2 x the error the reference itself makes under ``autocast(bfloat16)``
  on the same inputs (the oracle IS the reference's arithmetic, pinned by tests/golden; run under torch.autocast
  on the GPU it reproduces the reference's own cast points), every gradient -- the scalar gate gradients included -- by
  relative L2 <= 2e-2, on a loss that keeps the gate gradients well conditioned (``conditioned_upstream``)."""
import torch

from oracle import flamingo_oracle as O
from open_flamingo_amd.hip import path


def rel_err(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)


def _prefilled_sinks(P, dev, seed):
    """In-place accumulation targets holding a known non-zero value (as if an earlier backward had run)."""
    g = torch.Generator().manual_seed(900 + seed)
    base = {k: torch.randn(tuple(v.shape), generator=g) for k, v in P.items()}
    return base, {k: v.clone().to(dev).contiguous() for k, v in base.items()}


def make_bf16_weights(ops, P):
    return {k: ops.to_bf16(v.contiguous()) for k, v in P.items() if v.dim() == 2 and not k.endswith("latents") and "embs" not in k}


def check_xattn(ops, dev, *, B=2, L=40, T=2, n=16, heads=2, d=64, Dv=48, stream_dtype=torch.float32, media_locs=None,
                only_immediate=True, gates=(0.6, -0.4), seed=0, fwd_tol=1e-2, bwd_tol=3e-2, safe=0, inplace=False,
                fresh=False, zero_pad=False):
    m = O.OracleGatedCrossAttentionBlock(dim=d, dim_visual=Dv, heads=heads, dim_head=64,
                                         only_attend_immediate_media=only_immediate)
    st = O.seeded_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 100 + seed)
    st["attn_gate"] = torch.tensor([gates[0]])
    st["ff_gate"] = torch.tensor([gates[1]])
    m.load_state_dict(st)
    g = torch.Generator().manual_seed(200 + seed)
    x = torch.randn(B, L, d, generator=g)
    media = torch.randn(B, T, n, Dv, generator=g)
    if media_locs is None:
        media_locs = torch.zeros(B, L, dtype=torch.bool)
        media_locs[:, 2] = True
        media_locs[0, L // 2] = True
        media_locs[1, L - 3] = True
    if zero_pad:                  # data.py:205-215: unused image slots are all-zero images
        for bi in range(B):
            media[bi, int(media_locs[bi].sum()):] = 0
    w = torch.randn(B, L, d, generator=g)
    if stream_dtype == torch.bfloat16:
        x = x.to(torch.bfloat16).float()
        w = w.to(torch.bfloat16).float()
    # ---- oracle (rounding-point emulation), fp32 on CPU
    xo, mo = x.clone().requires_grad_(True), media.clone().requires_grad_(True)
    yo = m(xo, mo, media_locations=media_locs, quant=O.bf16_round)
    (yo * w).sum().backward()
    # ---- HIP path
    P = {k: v.detach().to(dev).contiguous() for k, v in m.named_parameters()}
    W = make_bf16_weights(ops, P)
    xd = x.to(dev).to(stream_dtype).reshape(B * L, d).contiguous()
    media_bf = ops.to_bf16(media.to(dev).reshape(B * T * n, Dv).contiguous())
    tt = torch.empty(B, L, dtype=torch.int32, device=dev)
    ops.text_time(media_locs.to(torch.uint8).to(dev).contiguous(), tt, L, False)
    kw = dict(B=B, L=L, T=T, n=n, heads=heads, only_immediate=only_immediate, safe=safe)
    y, S = path.xattn_block_fwd(ops, P, W, xd, media_bf, tt, **kw)
    # inference entry: projected media computed once, nothing kept -> the same bits as the training forward
    kv = path.xattn_project_media(ops, W, media_bf, heads)
    y_inf, none = path.xattn_block_fwd(ops, P, W, xd, media_bf, tt, kv=kv, keep=False, **kw)
    assert none is None and torch.equal(y_inf, y)
    dy = w.to(dev).to(stream_dtype).reshape(B * L, d).contiguous()
    base, sinks = _prefilled_sinks(P, dev, seed) if inplace else (None, None)
    # "fresh" Linear weights: the sink holds stale values the step epilogue did not clear; the backward must OVERWRITE them
    stale = {k for k in P if k.endswith(".weight") and P[k].dim() == 2} if (inplace and fresh) else set()
    dx, dmedia, grads = path.xattn_block_bwd(ops, P, W, S, media_bf, tt, dy, sinks=sinks, fresh=stale, **kw)
    if inplace:     # the kernels must have ADDED the gradient to what the sink held, in the sink itself
        assert all(grads[k] is sinks[k] for k in P)
        grads = {k: (grads[k].cpu() if k in stale else grads[k].cpu() - base[k]) for k in P}
    errs = {"y": rel_err(y.reshape(B, L, d), yo.detach())}
    errs["dx"] = rel_err(dx.reshape(B, L, d), xo.grad)
    errs["dmedia"] = rel_err(dmedia.reshape(B, T, n, Dv), mo.grad)
    if zero_pad:
        for bi in range(B):
            used = int(media_locs[bi].sum())
            assert float(dmedia.reshape(B, T, n, Dv)[bi, used:].abs().max()) == 0.0 == float(mo.grad[bi, used:].abs().max())
    for k, p in m.named_parameters():
        errs["d" + k] = rel_err(grads[k], p.grad)
    bad = {k: v for k, v in errs.items() if v > (fwd_tol if k == "y" else bwd_tol) * (4 if stream_dtype == torch.bfloat16 else 1)}
    assert not bad, f"xattn parity failures: {bad}\nall: {errs}"
    return errs


def check_perceiver(ops, dev, *, b=1, T=2, Fv=24, n=16, heads=2, D=64, depth=2, stream_dtype=torch.float32, seed=0,
                    fwd_tol=1e-2, bwd_tol=3e-2, need_dx=True, safe=0, frames=1, embs=False, inplace=False, fresh=False):
    m = O.OraclePerceiverResampler(dim=D, depth=depth, dim_head=64, heads=heads, num_latents=n,
                                   max_num_media=(T + 1 if embs else None), max_num_frames=(frames + 1 if embs else None))
    st = O.seeded_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 300 + seed)
    m.load_state_dict(st)
    g = torch.Generator().manual_seed(400 + seed)
    assert Fv % frames == 0
    x = torch.randn(b, T, frames, Fv // frames, D, generator=g)
    w = torch.randn(b, T, n, D, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = m(xo, quant=O.bf16_round)
    (yo * w).sum().backward()
    P = {k: v.detach().to(dev).contiguous() for k, v in m.named_parameters()}
    W = make_bf16_weights(ops, P)
    N = b * T
    xd = x.to(dev).to(stream_dtype).reshape(N * Fv, D).contiguous()
    kw = dict(N=N, Fv=Fv, n=n, heads=heads, depth=depth, safe=safe, T=T, frames=frames)
    y, S = path.perceiver_fwd(ops, P, W, xd, **kw)
    y_inf, none = path.perceiver_fwd(ops, P, W, xd, keep=False, **kw)
    assert none is None and torch.equal(y_inf, y)
    dy = w.to(dev).to(stream_dtype).reshape(N * n, D).contiguous()
    base, sinks = _prefilled_sinks(P, dev, seed) if inplace else (None, None)
    stale = {k for k in P if k.endswith(".weight") and P[k].dim() == 2} if (inplace and fresh) else set()
    dx, grads = path.perceiver_bwd(ops, P, W, S, dy, need_dx=need_dx, sinks=sinks, fresh=stale, **kw)
    if inplace:
        assert all(grads[k] is sinks[k] for k in P)
        grads = {k: (grads[k].cpu() if k in stale else grads[k].cpu() - base[k]) for k in P}
    errs = {"y": rel_err(y.reshape(b, T, n, D), yo.detach())}
    if need_dx:
        errs["dx"] = rel_err(dx.reshape(b, T, frames, Fv // frames, D), xo.grad)
    for k, p in m.named_parameters():
        errs["d" + k] = rel_err(grads[k], p.grad)
    bad = {k: v for k, v in errs.items() if v > (fwd_tol if k == "y" else bwd_tol) * (4 if stream_dtype == torch.bfloat16 else 1)}
    assert not bad, f"perceiver parity failures: {bad}\nall: {errs}"
    return errs


# =====================================================================================================================
# SURVEY.md 8c criterion: HIP path (amp_bf16 semantics: fp32 streams, bf16 GEMM operands) vs the fp32 oracle, with the
# reference's own autocast(bf16) error as the yardstick.  The oracle may be EXECUTED on the GPU (``oracle_dev="cuda"``):
# it is the same restatement, only fast enough for BASELINE config 2's full batch there.
# =====================================================================================================================
def rel_l2(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return (got - want).norm().item() / (want.norm().item() + 1e-30)


def max_abs(got, want):
    return (got.detach().double().cpu() - want.detach().double().cpu()).abs().max().item()


def _oracle_run(m, inputs, w, autocast, **fw):
    """forward + backward of an oracle module; returns (y, {name: grad}) for inputs and parameters, all fp32."""
    ins = [t.detach().clone().requires_grad_(True) for t in inputs]
    m.zero_grad(set_to_none=True)
    dev = ins[0].device.type
    with torch.autocast(dev, dtype=torch.bfloat16, enabled=autocast):
        y = m(*ins, **fw)
    (y.float() * w).sum().backward()
    g = {f"in{i}": t.grad.detach().float() for i, t in enumerate(ins)}
    g.update({k: p.grad.detach().float().clone() for k, p in m.named_parameters()})
    return y.detach().float(), g


def judge_8c(hip, ref32, refac, *, grad_tol=2e-2, factor=2.0, floor=1e-6):
    """hip / ref32 / refac: (y, grads) triples.  Returns (report, failures).  refac may be None (no autocast
    reference available, e.g. on CPU): then only the gradient rule and a 1e-2 forward rel-L2 bound are applied.
    ONE rule per quantity: forward error <= factor x the reference's own autocast error (rel-L2 and max-abs); every gradient --
    tensors and the (1,)-shaped gate gradients alike -- by relative L2 <= grad_tol.  The gate gradients are sums of B*L*d
    signed terms; the callers make them well conditioned (a loss whose upstream gradient is correlated with the block's output,
    see check_xattn_8c) instead of this function granting cancelling sums an escape hatch."""
    rep, bad = {}, {}
    y, y32 = hip[0], ref32[0]
    e_l2, e_mx = rel_l2(y, y32), max_abs(y, y32)
    if refac is not None:
        a_l2, a_mx = rel_l2(refac[0], y32), max_abs(refac[0], y32)
        rep["y"] = dict(hip_rel_l2=e_l2, autocast_rel_l2=a_l2, hip_max_abs=e_mx, autocast_max_abs=a_mx)
        if e_l2 > factor * a_l2 + floor or e_mx > factor * a_mx + floor:
            bad["y"] = rep["y"]
    else:
        rep["y"] = dict(hip_rel_l2=e_l2, hip_max_abs=e_mx)
        if e_l2 > 1e-2:
            bad["y"] = rep["y"]
    for k, g32 in ref32[1].items():
        if k not in hip[1] or hip[1][k] is None:
            continue
        e = rel_l2(hip[1][k], g32)
        ent = dict(hip_rel_l2=e)
        if refac is not None:
            ent["autocast_rel_l2"] = rel_l2(refac[1][k], g32)
        rep["d" + k] = ent
        if not e <= grad_tol:
            bad["d" + k] = ent
    return rep, bad


def conditioned_upstream(m, ins, w, **fw):
    """Upstream gradient for the 8c checks of a gated block: w + y32 (y32 = the fp32 oracle's output, detached).  With a
    purely random w the two gate gradients <w, branch_out> are sums of millions of signed terms that cancel to ~1/sqrt(N) of
    their term mass, so their RELATIVE error measures the cancellation, not the kernels; adding the output makes each contain
    tanh(gate) * |branch_out|^2 (no cancellation) while every other gradient still sees a dense random direction."""
    with torch.no_grad():
        y32 = m(*ins, **fw).float()
    return w + y32

def hip_xattn(ops, m, x, media, media_locs, w, *, heads, only_immediate=True, stream_dtype=torch.float32, dev="cuda"):
    """The HIP path of one gated block on the oracle module's parameters.  Returns (y, grads) keyed like _oracle_run."""
    B, L, d = x.shape
    _, T, n, Dv = media.shape
    P = {k: v.detach().to(dev).float().contiguous() for k, v in m.named_parameters()}
    W = make_bf16_weights(ops, P)
    xd = x.to(dev).to(stream_dtype).reshape(B * L, d).contiguous()
    media_bf = ops.to_bf16(media.to(dev).float().reshape(B * T * n, Dv).contiguous())
    tt = torch.empty(B, L, dtype=torch.int32, device=dev)
    ops.text_time(media_locs.to(torch.uint8).to(dev).contiguous(), tt, L, False)
    kw = dict(B=B, L=L, T=T, n=n, heads=heads, only_immediate=only_immediate)
    y, S = path.xattn_block_fwd(ops, P, W, xd, media_bf, tt, **kw)
    dy = w.to(dev).to(stream_dtype).reshape(B * L, d).contiguous()
    dx, dmedia, grads = path.xattn_block_bwd(ops, P, W, S, media_bf, tt, dy, **kw)
    g = {"in0": dx.reshape(B, L, d).float(), "in1": dmedia.reshape(B, T, n, Dv).float()}
    g.update({k: v.float() for k, v in grads.items()})
    return y.reshape(B, L, d).float(), g


def hip_perceiver(ops, m, x, w, *, heads, dev="cuda", stream_dtype=torch.float32, need_dx=True):
    b, T, Fr, v, D = x.shape
    P = {k: v_.detach().to(dev).float().contiguous() for k, v_ in m.named_parameters()}
    W = make_bf16_weights(ops, P)
    n = P["latents"].shape[0]
    depth = len(m.layers)
    N, Fv = b * T, Fr * v
    xd = x.to(dev).to(stream_dtype).reshape(N * Fv, D).contiguous()
    kw = dict(N=N, Fv=Fv, n=n, heads=heads, depth=depth, T=T, frames=Fr)
    y, S = path.perceiver_fwd(ops, P, W, xd, **kw)
    dy = w.to(dev).to(stream_dtype).reshape(N * n, D).contiguous()
    dx, grads = path.perceiver_bwd(ops, P, W, S, dy, need_dx=need_dx, **kw)
    g = {k: v_.float() for k, v_ in grads.items()}
    if need_dx:
        g["in0"] = dx.reshape(x.shape).float()
    return y.reshape(b, T, n, D).float(), g


def check_xattn_8c(ops, dev, *, B=2, L=40, T=2, n=16, heads=2, d=64, Dv=48, media_locs=None, only_immediate=True,
                   gates=(0.6, -0.4), seed=0, oracle_dev="cpu"):
    m = O.OracleGatedCrossAttentionBlock(dim=d, dim_visual=Dv, heads=heads, dim_head=64,
                                         only_attend_immediate_media=only_immediate)
    st = O.seeded_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 100 + seed)
    st["attn_gate"] = torch.tensor([gates[0]])
    st["ff_gate"] = torch.tensor([gates[1]])
    m.load_state_dict(st)
    m.to(oracle_dev)
    g = torch.Generator().manual_seed(200 + seed)
    x = torch.randn(B, L, d, generator=g)
    media = torch.randn(B, T, n, Dv, generator=g)
    # the Perceiver hands the blocks bf16-representable media under amp_bf16 only after the cast the blocks do themselves;
    # both sides get the same fp32 media here
    if media_locs is None:
        media_locs = torch.zeros(B, L, dtype=torch.bool)
        media_locs[:, 2] = True
        media_locs[0::2, L // 2] = True
        media_locs[1::2, L - 3] = True
    w = torch.randn(B, L, d, generator=g)
    ins = (x.to(oracle_dev), media.to(oracle_dev))
    fw = dict(media_locations=media_locs.to(oracle_dev))
    w = conditioned_upstream(m, ins, w.to(oracle_dev), **fw)
    ref32 = _oracle_run(m, ins, w, False, **fw)
    refac = _oracle_run(m, ins, w, True, **fw) if oracle_dev != "cpu" else None
    hip = hip_xattn(ops, m, x, media, media_locs, w.cpu(), heads=heads, only_immediate=only_immediate, dev=dev)
    rep, bad = judge_8c(hip, ref32, refac)
    assert not bad, f"SURVEY 8c tolerance failures: {bad}\nall: {rep}"
    return rep


def check_perceiver_8c(ops, dev, *, b=1, T=2, Fv=24, n=16, heads=2, D=64, depth=2, seed=0, oracle_dev="cpu", frames=1):
    m = O.OraclePerceiverResampler(dim=D, depth=depth, dim_head=64, heads=heads, num_latents=n)
    st = O.seeded_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 300 + seed)
    m.load_state_dict(st)
    m.to(oracle_dev)
    g = torch.Generator().manual_seed(400 + seed)
    x = torch.randn(b, T, frames, Fv // frames, D, generator=g)
    w = torch.randn(b, T, n, D, generator=g)
    ref32 = _oracle_run(m, (x.to(oracle_dev),), w.to(oracle_dev), False)
    refac = _oracle_run(m, (x.to(oracle_dev),), w.to(oracle_dev), True) if oracle_dev != "cpu" else None
    hip = hip_perceiver(ops, m, x, w, heads=heads, dev=dev)
    rep, bad = judge_8c(hip, ref32, refac)
    assert not bad, f"SURVEY 8c tolerance failures: {bad}\nall: {rep}"
    return rep


# =====================================================================================================================
# The product MODULE against reference-produced goldens at dim_head 64 (tests/golden/dh64_xattn_*.npz, made by
# tests/golden/make_golden.py --round4 from the REAL reference): the cached-media branch of helpers.py:175-178,199-205 among them
# (T_txt != mask length; media_locations = None).  Judged by the one 8c rule (judge_8c) with the reference's own autocast run,
# stored in the fixture, as the yardstick.  Runs on the GPU (tests/test_gpu_path.py) and on the emulator (tests/test_emu_modules.py).
# =====================================================================================================================
DH64_CASES = ("cached_media_decode", "cached_media_decode_attend_all", "no_media_locations_cached", "basic")


def check_block_module_against_dh64_golden(case, dev, golden_dir):
    import os
    import numpy as np
    from open_flamingo_amd.src.helpers import GatedCrossAttentionBlock
    z = np.load(os.path.join(golden_dir, f"dh64_xattn_{case}.npz"))
    blk = GatedCrossAttentionBlock(dim=64, dim_visual=32, dim_head=64, heads=int(z["heads"]),
                                   only_attend_immediate_media=bool(z["only_immediate"]))
    st = O.seeded_state({k: tuple(v.shape) for k, v in blk.state_dict().items()}, int(z["seed_params"]))
    blk.load_state_dict(st, strict=True)
    blk.to(dev).train()
    L, T_img, n = int(z["L"]), int(z["T_img"]), int(z["n_latents"])

    def rnd(shape, seed):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(*shape, generator=g, dtype=torch.float64).float()

    x = rnd((2, L, 64), int(z["seed_x"])).to(dev).requires_grad_(True)
    media = rnd((2, T_img, n, 32), int(z["seed_media"])).to(dev).requires_grad_(True)
    ml = torch.from_numpy(z["media_locations"]).to(dev) if int(z["has_media_locations"]) else None
    if ml is not None and not int(z["use_cached"]):
        assert ml.shape[1] == L
    y = blk(x, media, media_locations=ml, use_cached_media=bool(z["use_cached"]))
    (y.float() * torch.from_numpy(z["w"]).to(dev)).sum().backward()
    hip_g = {"grad.x": x.grad, "grad.media": media.grad, **{"grad." + k: p.grad for k, p in blk.named_parameters()}}
    want_g = {k: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad.")}
    assert set(hip_g) == set(want_g), set(hip_g) ^ set(want_g)
    hip = (y.detach().float().cpu(), {k: v.detach().float().cpu() for k, v in hip_g.items()})
    ref32 = (torch.from_numpy(z["y"]), want_g)
    refac = (torch.from_numpy(z["amp.y"]), {k: None for k in want_g})
    rep, bad = {}, {}
    e_l2, e_mx = rel_l2(hip[0], ref32[0]), max_abs(hip[0], ref32[0])
    a_l2, a_mx = rel_l2(refac[0], ref32[0]), max_abs(refac[0], ref32[0])
    rep["y"] = dict(hip_rel_l2=e_l2, autocast_rel_l2=a_l2, hip_max_abs=e_mx, autocast_max_abs=a_mx)
    if e_l2 > 2.0 * a_l2 + 1e-6 or e_mx > 2.0 * a_mx + 1e-6:
        bad["y"] = rep["y"]
    for k, g32 in want_g.items():
        e = rel_l2(hip[1][k], g32)
        rep[k] = dict(hip_rel_l2=e, autocast_rel_l2=float(z["amp.rel_l2." + k]))
        if not e <= 2e-2:
            bad[k] = rep[k]
    assert not bad, f"{case}: SURVEY 8c tolerance failures vs the reference-produced golden: {bad}\nall: {rep}"
    return rep
