"""cProfile of the host side of a few train steps (no per-step sync): where the ~108 ms of enqueue time per step go."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.train import step, synthetic, towers, sparse_rows
from open_flamingo_amd.train.reducer import GradReducer

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
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    step.train_step(model, red, opt, batch, info, nan_check=False)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
