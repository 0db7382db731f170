"""Is the step GPU-bound or host-(launch-)bound?  K steps without the per-step host sync (nan_check off): time until the
host has ENQUEUED everything vs time until the GPU has finished.  enqueue << total: the GPU is the bottleneck."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.train import step, synthetic, towers
from open_flamingo_amd.train.reducer import GradReducer
from open_flamingo_amd.train import sparse_rows

model, info = towers.build_flamingo("OF-3B", device="cuda", seed=0, gates=0.5, frozen_bf16=True, fused_lm_attention="libofhip",
                                    tower_layernorm="libofhip", lm_loss="libofhip", fused_lm_blocks=True, fused_vision="libofhip")
model.train()
towers.use_tuned_vendor_gemms()
sparse_rows.enable(model, [info["media_token_id"], info["eoc_token_id"]])
red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
opt = step.build_optimizer(model, reducer=red)
batch = synthetic.make_batch(32, 2, 256, info, "cuda", seed=1)
for _ in range(3):
    step.train_step(model, red, opt, batch, info, nan_check=False)
torch.cuda.synchronize()
K = 8
t0 = time.perf_counter()
for _ in range(K):
    step.train_step(model, red, opt, batch, info, nan_check=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"host_enqueue_ms_per_step": round((t1 - t0) / K * 1e3, 2), "total_ms_per_step": round((t2 - t0) / K * 1e3, 2),
                  "cpu_count": os.cpu_count()}))
