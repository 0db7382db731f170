"""Greedy decode timing of the inference path: OF-3B sized random-init model, T images per prompt,
`Flamingo.generate` with HF's KV cache for the frozen LM.  Two arms on the same model and inputs:

  cached     the product: each gated cross-attention block projects the media once per prompt (to_kv) and reuses the
             keys/values for every generated token, nothing is saved for a backward;
  reproject  what the reference does (helpers.py:189 runs to_kv(media) in every block for every token): the same
             kernels with the projection recomputed per call.

Also times the 24 blocks alone on a single-token input (the hot path's share of a decode step).  Prints JSON lines.
Usage: python tools/bench_decode.py [--family OF-3B] [--batch 1 8] [--images 2] [--prompt 32] [--new 48]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_flamingo_amd.hip import path as _path          # noqa: E402
from open_flamingo_amd.train import synthetic, towers    # noqa: E402


def _generate(model, batch, new):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(batch["vision_x"], batch["lang_x"], attention_mask=batch["attention_mask"],
                         max_new_tokens=new, min_new_tokens=new, do_sample=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def _blocks_only(model, B, T, d, reps=20):
    """One decode step of the hot path alone: 24 blocks on a (B, 1, d) token with conditioned media."""
    blocks = [b for b in model.lang_encoder.gated_cross_attn_layers if b is not None]
    media = torch.randn(B, T, 64, model.vis_dim, device="cuda")
    locs = torch.zeros(B, 8, dtype=torch.bool, device="cuda")
    locs[:, 0] = True
    x = torch.randn(B, 1, d, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.no_grad():
        for _ in range(3):
            for blk in blocks:
                x = blk(x, media, media_locations=locs, use_cached_media=True)
        ev[0].record()
        for _ in range(reps):
            for blk in blocks:
                x = blk(x, media, media_locations=locs, use_cached_media=True)
        ev[1].record()
    torch.cuda.synchronize()
    for blk in blocks:
        blk.release_media_cache()
    return ev[0].elapsed_time(ev[1]) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="OF-3B")
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--prompt", type=int, default=32)
    ap.add_argument("--new", type=int, default=48)
    a = ap.parse_args()
    model, info = towers.build_flamingo(a.family, device="cuda", seed=0, gates=0.5, frozen_bf16=True)
    model.eval()
    project, block_fwd = _path.xattn_project_media, _path.xattn_block_fwd
    arms = {
        "cached": (project, block_fwd),
        "reproject": (project, lambda *args, kv=None, **kw: block_fwd(*args, kv=None, **kw)),   # ignore the cached kv
    }
    for B in a.batch:
        batch = synthetic.make_batch(B, a.images, a.prompt, info, "cuda", seed=3)
        row = dict(family=a.family, batch=B, images=a.images, prompt_tokens=a.prompt, new_tokens=a.new)
        tokens = {}
        for name, (proj, fwd) in arms.items():
            _path.xattn_project_media, _path.xattn_block_fwd = proj, fwd
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                _generate(model, batch, 4)
                secs, out = min((_generate(model, batch, a.new) for _ in range(2)), key=lambda r: r[0])
                row[name + "_blocks_ms_per_token"] = round(_blocks_only(model, B, a.images, model.lang_dim), 3)
            tokens[name] = out
            row[name + "_s"] = round(secs, 4)
            row[name + "_tokens_per_s"] = round(B * a.new / secs, 1)
        _path.xattn_project_media, _path.xattn_block_fwd = project, block_fwd
        row["same_tokens"] = bool(torch.equal(tokens["cached"], tokens["reproject"]))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
