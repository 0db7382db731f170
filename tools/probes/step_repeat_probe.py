"""Is a train step repeatable bit for bit?  The model of tests/test_gpu_path.py::test_norm_taps_* (d = 2048, two layers) built three times from
the same seed: full-pass norm twice, norm taps once; three steps each; per run the losses, and between runs the largest relative difference of
a trained matrix.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_flamingo_amd.train import sparse_rows, step, synthetic, towers
from open_flamingo_amd.train.reducer import GradReducer

if os.environ.get("STEP_REPEAT_FUSED_XATTN") == "0":          # the gated blocks' attention branch as five launches (round 6 comparison)
    from open_flamingo_amd.hip import path as _path
    _path.FUSED_XATTN = False
towers.FAMILY["OF-wide-test"] = dict(lm="mpt", d=2048, layers=2, heads=16, vocab=1000, every=1)
vkw = dict(width=64, layers=2, heads=2, patch=14, image=224)


def run(tap, fused_blocks=True):
    model, info = towers.build_flamingo("OF-wide-test", device="cuda", seed=0, gates=0.5, vision_kw=vkw, frozen_bf16=True,
                                        fused_lm_attention="libofhip", tower_layernorm="libofhip", lm_loss="libofhip",
                                        fused_lm_blocks=fused_blocks, perceiver_depth=1)
    model.train()
    rows = [info["media_token_id"], info["eoc_token_id"]]
    sparse_rows.enable(model, rows)
    red = GradReducer(model, embedding_rows=rows)
    opt = step.build_optimizer(model, lr=1e-3, reducer=red)
    opt.tap_norm = tap
    batch = synthetic.make_batch(2, 2, 64, info, "cuda", seed=5)
    losses, grads = [], None
    for i in range(3):
        if i == 0:          # the first backward's gradients, before any update
            step.forward_loss(model, batch, info).backward()
            red.finish(average=False)
            grads = [b["flat"].clone() for b in red.buckets]
            opt.step()
            red.zero_grad(flat_already_zero=True)
        else:
            losses.append(float(step.train_step(model, red, opt, batch, info, nan_check="device")))
    torch.cuda.synchronize()
    return losses, grads, {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad and p.dim() == 2}


def diff(a, b):
    worst = max(((a[2][k] - b[2][k]).norm().item() / (b[2][k].norm().item() + 1e-30), k) for k in b[2])
    g = max((x - y).abs().max().item() / (y.abs().max().item() + 1e-30) for x, y in zip(a[1], b[1]))
    return dict(worst_weight_rel=worst[0], key=worst[1], first_grad_rel_max=g, losses=[a[0], b[0]])


def run_like_the_test(tap):
    model, info = towers.build_flamingo("OF-wide-test", device="cuda", seed=0, gates=0.5, vision_kw=vkw, frozen_bf16=True,
                                        fused_lm_attention="libofhip", tower_layernorm="libofhip", lm_loss="libofhip",
                                        fused_lm_blocks=True, perceiver_depth=1)
    model.train()
    rows = [info["media_token_id"], info["eoc_token_id"]]
    sparse_rows.enable(model, rows)
    red = GradReducer(model, embedding_rows=rows)
    opt = step.build_optimizer(model, lr=1e-3, reducer=red)
    opt.tap_norm = tap
    batch = synthetic.make_batch(2, 2, 64, info, "cuda", seed=5)
    small = synthetic.make_batch(4, 1, 32, info, "cuda", seed=6)
    losses, snaps = [], []
    snap = lambda: {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad and p.dim() == 2}
    for i in range(3):
        losses.append(float(step.train_step(model, red, opt, batch, info, nan_check="device")))
        snaps.append(snap())
    with red.no_sync():
        (0.2 * step.forward_loss(model, small, info)).backward()
    step.forward_loss(model, batch, info).backward()
    red.finish(average=False)
    grads = [b["flat"].clone() for b in red.buckets]
    opt.step()
    torch.cuda.synchronize()
    red.zero_grad(flat_already_zero=True)
    snaps.append(snap())
    return losses, grads, snaps[-1], snaps, float(opt._sumsq)


base = run_like_the_test(False)
for rep in range(6):
    r = run_like_the_test(rep % 2 == 0)
    per_step = [max((a[k] - b[k]).norm().item() / (b[k].norm().item() + 1e-30) for k in b) for a, b in zip(r[3], base[3])]
    print(json.dumps(dict(pair="like the test: %s vs full pass" % ("taps" if rep % 2 == 0 else "full pass"), per_step_worst_weight_rel=per_step,
                          sumsq=[r[4], base[4]], **diff(r, base))), flush=True)
A, A2, B = run(False), run(False), run(True)
print(json.dumps(dict(pair="full pass vs full pass (repeatability)", **diff(A2, A))))
print(json.dumps(dict(pair="taps vs full pass", **diff(B, A))))
C, C2 = run(False, fused_blocks=False), run(False, fused_blocks=False)
print(json.dumps(dict(pair="full pass twice, HF eager frozen blocks", **diff(C2, C))))
