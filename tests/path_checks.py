"""Shared parity checks of the composed hot path (open_flamingo_amd.hip.path) against the oracle.
Used with the emulator Ops on CPU (tests/test_emu_path.py) and with the real library on the GPU
(tests/test_gpu_path.py).  Tolerances are relative to the max-abs of the reference tensor:
forward 1e-2 vs the rounding-point oracle (bf16 operand rounding emulated), gradients 3e-2 (the backward's
bf16 intermediates are not emulated by the oracle)."""
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
                only_immediate=True, gates=(0.6, -0.4), seed=0, fwd_tol=1e-2, bwd_tol=3e-2, safe=0, inplace=False):
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
    dx, dmedia, grads = path.xattn_block_bwd(ops, P, W, S, media_bf, tt, dy, sinks=sinks, **kw)
    if inplace:     # the kernels must have ADDED the gradient to what the sink held, in the sink itself
        assert all(grads[k] is sinks[k] for k in P)
        grads = {k: grads[k].cpu() - base[k] for k in P}
    errs = {"y": rel_err(y.reshape(B, L, d), yo.detach())}
    errs["dx"] = rel_err(dx.reshape(B, L, d), xo.grad)
    errs["dmedia"] = rel_err(dmedia.reshape(B, T, n, Dv), mo.grad)
    for k, p in m.named_parameters():
        errs["d" + k] = rel_err(grads[k], p.grad)
    bad = {k: v for k, v in errs.items() if v > (fwd_tol if k == "y" else bwd_tol) * (4 if stream_dtype == torch.bfloat16 else 1)}
    assert not bad, f"xattn parity failures: {bad}\nall: {errs}"
    return errs


def check_perceiver(ops, dev, *, b=1, T=2, Fv=24, n=16, heads=2, D=64, depth=2, stream_dtype=torch.float32, seed=0,
                    fwd_tol=1e-2, bwd_tol=3e-2, need_dx=True, safe=0, frames=1, embs=False, inplace=False):
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
    dx, grads = path.perceiver_bwd(ops, P, W, S, dy, need_dx=need_dx, sinks=sinks, **kw)
    if inplace:
        assert all(grads[k] is sinks[k] for k in P)
        grads = {k: grads[k].cpu() - base[k] for k in P}
    errs = {"y": rel_err(y.reshape(b, T, n, D), yo.detach())}
    if need_dx:
        errs["dx"] = rel_err(dx.reshape(b, T, frames, Fv // frames, D), xo.grad)
    for k, p in m.named_parameters():
        errs["d" + k] = rel_err(grads[k], p.grad)
    bad = {k: v for k, v in errs.items() if v > (fwd_tol if k == "y" else bwd_tol) * (4 if stream_dtype == torch.bfloat16 else 1)}
    assert not bad, f"perceiver parity failures: {bad}\nall: {errs}"
    return errs
