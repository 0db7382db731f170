"""Same-box timing of the frozen towers' stock forms against the product forms (bench.py: TOWER_ARMS): torch SDPA or HF's eager attention
in the MPT blocks, the HF block / CLIP modules instead of the fused autograd nodes, eager LayerNorm / loss.  These were bench.py flags
(--lm-attention, --lm-blocks, --vision, --tower-layernorm, --lm-loss) until round 6; they are A/B arms, not benchmark options.  PROFILING TOOL.

    python tools/ab_tower_arms.py lm_attention=sdpa [vision=modules ...] [-- bench.py flags]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

CHOICES = dict(lm_attention=("libofhip", "sdpa", "eager"), lm_blocks=("fused", "modules"), vision=("libofhip", "sdpa", "modules"),
               tower_layernorm=("libofhip", "eager"), lm_loss=("libofhip", "hf"))
rest = []
for a in sys.argv[1:]:
    k, _, v = a.partition("=")
    if k in CHOICES and v:
        assert v in CHOICES[k], (k, v, CHOICES[k])
        bench.TOWER_ARMS[k] = v
    elif a != "--":
        rest.append(a)
sys.argv = ["bench.py"] + rest
bench.main()
