"""Which host-side calls of a train step synchronise with the GPU?  torch.cuda.set_sync_debug_mode("warn") over one step (after
warm-up), warnings printed with the innermost repository / transformers frames; plus host-side wall time of the step's phases.
PROFILING TOOL."""
import os, sys, time, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
for _ in range(3):
    step.train_step(model, red, opt, batch, info, nan_check="device", next_vision_x=batch["vision_x"])
torch.cuda.synchronize()
# host-side time of the prefetch call alone (enqueue only, if nothing in it synchronises)
t0 = time.perf_counter()
model.prefetch_vision(batch["vision_x"], amp_dtype=torch.bfloat16)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"prefetch_vision: host {1e3 * (t1 - t0):.2f} ms to return, {1e3 * (t2 - t0):.2f} ms until the GPU has finished it")
seen = {}
def hook(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if ("open_flamingo_amd" in f.filename or "transformers" in f.filename or "bench" in f.filename)]
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-4:][::-1])
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step.train_step(model, red, opt, batch, info, nan_check="device", next_vision_x=batch["vision_x"])
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronising calls in one step:")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"  {v:4d}x  {k}")
if not seen:
    print("  none")
