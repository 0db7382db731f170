"""Reference-equivalent EAGER step time on the MI355X itself (BASELINE.md section 3 "Also": this is the number the
>=5x target refers to).  /root/reference does not exist on the GPU box, so the hot-path modules are the oracle's
nn.Modules -- a line-for-line PyTorch restatement of reference helpers.py, pinned to it by tests/test_oracle_golden.py
-- plugged into the same Flamingo + the same frozen towers, run as the reference runs them: eager ATen ops under
torch.autocast(bfloat16), gradients exchanged by nothing (1 GPU), the embedding-gradient row mask of
train_utils.py:174-196 (train_step(reducer=None) applies it as the reference writes it: a dense zero_mask multiply),
clip_grad_norm_ + AdamW.  Not part of the product."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from open_flamingo_amd.train import step, synthetic, towers
from tests.cpu_model import swap_in_oracle


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    # --stock-towers: the frozen towers exactly as stock modules run them (MIOpen conv patch embedding, fp32 frozen weights
    # re-cast by autocast each forward); default: the same two tower-side choices bench.py makes,
    # so that the difference to bench.py is the hot path + step epilogue only
    stock = "--stock-towers" in sys.argv
    family = argv[0] if argv else "OF-3B"
    B, T, L = (int(x) for x in (argv[1:4] if len(argv) > 3 else (32, 2, 256)))
    steps, warm = 4, 2
    model, info = towers.build_flamingo(family, device="cuda", seed=0, gates=0.5, frozen_bf16=not stock, fused_lm_attention=not stock,
                                        vision_kw=dict(patch_embed="conv") if stock else None)
    swap_in_oracle(model)
    model.cuda().train()
    opt = step.build_optimizer(model)
    batch = synthetic.make_batch(B, T, L, info, "cuda", seed=1)

    def one():
        return step.train_step(model, None, opt, batch, info)

    losses = [float(one()) for _ in range(warm)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
        losses.append(loss)
    torch.cuda.synchronize()
    losses = [float(l) for l in losses]
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"what": "reference-equivalent eager step (oracle modules, autocast bf16) on MI355X", "family": family, "towers": "stock (conv patch embed, fp32 frozen weights, HF eager MPT attention)" if stock else "as bench.py (GEMM patch embed, bf16-held frozen weights, fused LM attention)",
                      "B": B, "T": T, "L": L, "ms_per_step": round(ms, 2), "images_per_s": round(B * T / ms * 1e3, 2),
                      "loss": float(loss), "losses_per_step": [round(l, 4) for l in losses], "embedding_row_mask": "train_utils.py:174-196 applied", "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))


if __name__ == "__main__":
    main()
