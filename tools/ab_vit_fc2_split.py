"""Same-box A/B of the CLIP tower's fc2 row split (train/frozen_blocks.py: _FC2_WHOLE_TILES): bench.py's default step (BASELINE config 2) with
fc2 as whole 256-row tiles + the ragged rest against one launch over the 16448 rows, arms alternating, two rounds.  PROFILING TOOL.

    python tools/ab_vit_fc2_split.py [--steps 12 --warmup 4]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, ROOT)
    from open_flamingo_amd.train import frozen_blocks
    frozen_blocks._FC2_WHOLE_TILES = sys.argv[2] == "1"
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-reference-eager"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for rnd in range(2):
        for split in (1, 0):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(split)] + sys.argv[1:], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("fc2_whole_tiles=%d %s" % (split, line[-1][:330] if line else "FAILED " + out.stderr[-400:]), flush=True)
