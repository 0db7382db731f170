"""End-to-end sanity on one MI355X: OF-3B-sized model, fixed synthetic batch, N optimizer steps; the loss must fall
(the trainable Perceiver / gated cross-attention parameters memorise the batch).  Prints the loss every few steps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.train import step, synthetic, towers
from open_flamingo_amd.train.reducer import GradReducer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
model, info = towers.build_flamingo("OF-3B", device="cuda", seed=0, gates=0.5, frozen_bf16=True, fused_lm_attention="libofhip")
model.train()
red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
opt = step.build_optimizer(model, lr=1e-4, reducer=red)
batch = synthetic.make_batch(32, 2, 256, info, "cuda", seed=1)
losses = []
for i in range(steps):
    l = step.train_step(model, red, opt, batch, info)
    if i % 5 == 0 or i == steps - 1:
        losses.append((i, round(float(l), 4)))
print(json.dumps({"losses": losses, "grad_norm_last": round(float(opt.grad_norm()), 4)}))
assert losses[-1][1] < losses[0][1] - 0.3, "loss did not fall"
