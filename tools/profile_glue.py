"""Call sites of the small eager kernels of a train step (fills, dtype copies): torch.profiler events of two steps, grouped by operator,
input shapes and (where the profiler recorded one) the innermost non-torch Python frames.  Only operators launched INSIDE train_step
are counted -- a rocprofv3 kernel table of a whole bench.py run also holds the model's random initialisation, the reference-eager leg and
the GEMM report's buffers.  PROFILING TOOL."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from open_flamingo_amd.train import sparse_rows, step, synthetic, towers
from open_flamingo_amd.train.reducer import GradReducer

model, info = towers.build_flamingo("OF-3B", device="cuda", seed=0, gates=0.5, frozen_bf16=True, fused_lm_attention="libofhip",
                                    tower_layernorm="libofhip", lm_loss="libofhip", fused_lm_blocks=True, fused_vision="libofhip")
model.train()
towers.use_tuned_vendor_gemms()
rows = [info["media_token_id"], info["eoc_token_id"]]
sparse_rows.enable(model, rows)
red = GradReducer(model, embedding_rows=rows)
opt = step.build_optimizer(model, reducer=red)
batch = synthetic.make_batch(32, 2, 256, info, "cuda", seed=1)
kw = dict(nan_check="device", next_vision_x=batch["vision_x"])
for _ in range(3):
    step.train_step(model, red, opt, batch, info, **kw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        step.train_step(model, red, opt, batch, info, **kw)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.name in ("aten::mm", "aten::addmm", "aten::matmul", "aten::linear"):
        continue
    dev = getattr(e, "self_device_time_total", 0)
    if dev <= 0:
        continue
    frames = [f for f in (e.stack or []) if "site-packages/torch" not in f and "dist-packages/torch" not in f and "<built-in" not in f]
    where = " <- ".join(f.split("/")[-1][:70] for f in frames[:3]) or "-"
    key = (e.name, str(e.input_shapes)[:80], where)
    agg[key][0] += dev / 2e3
    agg[key][1] += 1
tot = 0.0
for (name, shapes, where), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    tot += ms
    print(f"{ms:7.3f} ms/step {n // 2:4d}x {name:24s} {shapes:80s} {where}")
print("listed total", round(tot, 2), "ms/step")
