"""Which host-side call sites produce the eager glue kernels (copies, fills, casts) of a train step?  torch.profiler over two
steps, grouped by operator + python stack.  PROFILING TOOL."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from open_flamingo_amd.train import sparse_rows, step, synthetic, towers
from open_flamingo_amd.train.reducer import GradReducer


def main():
    fam = sys.argv[1] if len(sys.argv) > 1 else "OF-3B"
    model, info = towers.build_flamingo(fam, device="cuda", seed=0, gates=0.5, frozen_bf16=True, fused_lm_attention="libofhip",
                                        tower_layernorm="libofhip", lm_loss="libofhip", fused_lm_blocks=True, fused_vision="libofhip")
    model.train()
    towers.use_tuned_vendor_gemms()
    sparse_rows.enable(model, [info["media_token_id"], info["eoc_token_id"]])
    red = GradReducer(model, embedding_rows=[info["media_token_id"], info["eoc_token_id"]])
    opt = step.build_optimizer(model, reducer=red)
    batch = synthetic.make_batch(32, 2, 256, info, "cuda", seed=1)
    for _ in range(2):
        step.train_step(model, red, opt, batch, info, nan_check="device")
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            step.train_step(model, red, opt, batch, info, nan_check="device")
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_stack_n=6):
        dev = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
        if dev > 0 and e.key.startswith("aten::") and not any(k in e.key for k in ("mm", "addmm", "matmul", "linear")):
            stack = [s for s in e.stack if "open_flamingo_amd" in s or "transformers" in s][:3]
            rows.append((dev / 2e3, e.count // 2, e.key, " <- ".join(s.split("/")[-1] for s in stack)))
    rows.sort(reverse=True)
    tot = 0
    for ms, n, key, stack in rows[:40]:
        tot += ms
        print(f"{ms:7.3f} ms/step {n:5d}x {key:28s} {stack[:200]}")
    print("listed total", round(tot, 2), "ms/step")


if __name__ == "__main__":
    main()
