"""Decode-step (T_txt = 1) cost of the gated cross-attention blocks alone, OF-3B widths: per-kernel launches vs the
per-block HIP graph, per batch size.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel GPU times.
Usage: python tools/prof_decode_blocks.py [--blocks 6] [--batch 1 8] [--reps 30]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_flamingo_amd.src.helpers import GatedCrossAttentionBlock   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dim", type=int, default=2048)
    a = ap.parse_args()
    torch.manual_seed(0)
    blocks = [GatedCrossAttentionBlock(dim=a.dim, dim_visual=1024).cuda().eval() for _ in range(a.blocks)]
    with torch.no_grad():
        for b in blocks:
            b.attn_gate.fill_(0.5)
            b.ff_gate.fill_(0.5)
    for B in a.batch:
        media = torch.randn(B, 2, 64, 1024, device="cuda")
        locs = torch.zeros(B, 8, dtype=torch.bool, device="cuda")
        locs[:, 0] = True
        row = dict(batch=B, blocks=a.blocks, dim=a.dim)
        for mode in ("per_kernel", "graph"):
            GatedCrossAttentionBlock.decode_graphs = mode == "graph"
            x = torch.randn(B, 1, a.dim, device="cuda")
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            with torch.no_grad():
                for _ in range(3):
                    for blk in blocks:
                        x = blk(x, media, media_locations=locs, use_cached_media=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ev[0].record()
                for _ in range(a.reps):
                    for blk in blocks:
                        x = blk(x, media, media_locations=locs, use_cached_media=True)
                ev[1].record()
                host = time.perf_counter() - t0
            torch.cuda.synchronize()
            n = a.reps * a.blocks
            row[mode + "_gpu_us_per_block"] = round(ev[0].elapsed_time(ev[1]) * 1e3 / n, 1)
            row[mode + "_host_issue_us_per_block"] = round(host * 1e6 / n, 1)
            if mode == "graph":
                row["graph_captured"] = all(any(s["graph"] is not None for s in blk.__dict__.get("_decode_graph", {}).values())
                                            for blk in blocks)
        print(json.dumps(row), flush=True)
    GatedCrossAttentionBlock.decode_graphs = True


if __name__ == "__main__":
    main()
