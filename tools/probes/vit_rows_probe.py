"""The CLIP ViT-L/14 tower's four GEMMs per layer (y = x W^T + b, bf16; BASELINE config 2: 64 images x 257 tokens = 16448 rows) on the
vendor library with TunableOp tuning, at M = 16448 (64.25 row tiles of 256: every tiling leaves a ragged last round), M = 16384 (the
patch tokens alone: 256 / 768 / 1024 / 256 tiles of 256 x 256 -- whole rounds of the 256 CUs) and M = 64 (the class tokens alone).
Rotating operand sets behind a 512-MB copy (cold, as in a step) and hot.  PROFILING TOOL.
    python tools/probes/vit_rows_probe.py [out_table.csv]"""
import json, os, sys
import torch
import torch.cuda.tunable as tun

dev = "cuda"
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(30)
tun.set_max_tuning_iterations(100)
NSETS = 4
filler_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
filler_b = torch.empty_like(filler_a)


def timed(fn, cold):
    ts = []
    for i in range(NSETS * 6):
        if cold:
            filler_b.copy_(filler_a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[NSETS:])
    return round(ts[len(ts) // 2], 1)


for name, N, K in (("qkv", 3072, 1024), ("out_proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    rec = {"probe": "vit_rows", "gemm": name, "N": N, "K": K}
    for M in (16448, 16384, 64):
        xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NSETS)]
        w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        for _ in range(3):
            torch.addmm(b, xs[0], w.t())
        torch.cuda.synchronize()
        f = lambda i: torch.addmm(b, xs[i % NSETS], w.t())
        rec[f"M{M}_us_cold"], rec[f"M{M}_us_hot"] = timed(f, True), timed(f, False)
        rec[f"M{M}_tflops_cold"] = round(2.0 * M * N * K / rec[f"M{M}_us_cold"] / 1e6, 1)
    print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    tun.write_file(sys.argv[1])
