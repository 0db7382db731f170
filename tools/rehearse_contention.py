"""One-GPU rehearsal of the multi-GPU CU-contention risk: while the backward runs, RCCL's
collective kernels occupy some CUs on a side stream; a big-tile GEMM with exactly 256 tiles (one per CU: four of the six FFN
GEMMs of a gated block at cfg-2) then needs a second round of tiles.  Here every point where GradReducer would launch a bucket's
all-reduce instead launches `hold` one-wave workgroups that sit on their CUs for the time the exchange of that bucket would take
(bytes / assumed bus bandwidth) on the reducer's side stream.  Reports ms/step vs CUs held.

    python tools/rehearse_contention.py [--busbw 300] [--steps 6]

PROFILING TOOL (loads tools/libofhip_tools.so by path; the package never does)."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from open_flamingo_amd.csrc import build as _build
from open_flamingo_amd.train import step, synthetic, towers
from open_flamingo_amd.hip.ops import Ops
from open_flamingo_amd.train.reducer import GradReducer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--busbw", type=float, default=300.0, help="assumed all-reduce bus bandwidth, GB/s (8 GPUs over xGMI)")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--holds", default="0,8,16,32,64")
    ap.add_argument("--heavy", action="store_true", help="stand-in with a collective kernel's footprint: 256 threads, 128 registers, 32 KiB of LDS")
    ap.add_argument("--reserve", default="off", choices=["off", "match"],
                    help="match: GradReducer(reserve_cus = CUs held) -- the libofhip GEMMs are laid out stream-K for the CUs that are left")
    a = ap.parse_args()
    path = _build.TOOLS_LIB if os.path.exists(_build.TOOLS_LIB) else _build.build(tools=True)
    tl = ctypes.CDLL(path)
    tl.of_tools_hold_cus.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
    tl.of_tools_hold_cus_heavy.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    model, info = towers.build_flamingo("OF-3B", device="cuda", seed=0, gates=0.5, frozen_bf16=True, fused_lm_attention="libofhip",
                                        tower_layernorm="libofhip", lm_loss="libofhip", fused_lm_blocks=True, fused_vision="libofhip")
    model.train()
    towers.use_tuned_vendor_gemms()
    from open_flamingo_amd.train import sparse_rows
    sparse_rows.enable(model, [info["media_token_id"], info["eoc_token_id"]])
    red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt = step.build_optimizer(model, reducer=red)
    batch = synthetic.make_batch(32, 2, 256, info, "cuda", seed=1)
    hold = {"n": 0}
    side = torch.cuda.Stream()

    def launch(flat):        # what GradReducer._launch does, with the collective replaced by CU-holding workgroups
        if hold["n"] <= 0:
            return
        compute = torch.cuda.current_stream()
        side.wait_stream(compute)
        # ring all-reduce time of this bucket: 2 (n-1)/n * bytes / busbw, n = 8
        us = 2 * 7 / 8 * flat.numel() * 4 / (a.busbw * 1e3)
        if a.heavy:
            tl.of_tools_hold_cus_heavy(hold["n"], int(us * 100), 32768, ctypes.c_void_p(side.cuda_stream))
        else:
            tl.of_tools_hold_cus(hold["n"], int(us * 100), ctypes.c_void_p(side.cuda_stream))     # 100-MHz ticks
        if a.reserve == "match":
            Ops.default().cu_limit = 256 - hold["n"]

    red._launch = launch
    orig_finish = red.finish

    def finish(average=True):
        orig_finish(average)
        Ops.default().cu_limit = 0
        torch.cuda.current_stream().wait_stream(side)

    red.finish = finish
    for n in [int(x) for x in a.holds.split(",")]:
        hold["n"] = n
        for _ in range(2):
            step.train_step(model, red, opt, batch, info, nan_check="device")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step.train_step(model, red, opt, batch, info, nan_check="device")
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(json.dumps({"cus_held_per_collective": n, "stand_in": "256 threads, 128 registers, 32 KiB LDS" if a.heavy else "one wave, no LDS",
                          "gemm_layout": "stream-K for 256 - held workgroups" if a.reserve == "match" else "as without a collective",
                          "assumed_busbw_GBps": a.busbw, "ms_per_step": round(ms, 2),
                          "note": "each bucket's exchange replaced by one-wave workgroups resident on the side stream for 2*(7/8)*bytes/busbw"}),
              flush=True)


if __name__ == "__main__":
    main()
