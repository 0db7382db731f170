"""Same-box A/B of the frozen GPT-NeoX blocks' head layout (train/frozen_blocks.py: _NEOX_COMPACT_HEADS): bench.py's step at BASELINE config 4
(OF-4B, RedPajama-INCITE-3B: head size 80) with compact heads (OfAttnArgs.head_valid, ABI v11) against zero-padded copies + repack passes,
arms alternating, two rounds.  One bench JSON line per arm, prefixed by the arm.  PROFILING TOOL.

    python tools/ab_neox_compact_heads.py [--steps 8 --warmup 3]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, ROOT)
    from open_flamingo_amd.train import frozen_blocks
    frozen_blocks._NEOX_COMPACT_HEADS = sys.argv[2] == "1"
    sys.argv = ["bench.py", "--config", "4", "--no-cpu-baseline", "--no-reference-eager"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for rnd in range(2):
        for compact in (1, 0):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(compact)] + sys.argv[1:], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("compact_heads=%d %s" % (compact, line[-1][:400] if line else "FAILED " + out.stderr[-400:]), flush=True)
