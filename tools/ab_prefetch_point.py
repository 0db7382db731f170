"""Same-box A/B of WHERE the next step's frozen-tower forward is enqueued (train/step.py: PREFETCH_IN_BACKWARD): from inside the backward
(the product: at the start of gated block n_blocks // 12's backward; 100 + k: block k's), against behind the whole backward; bench.py's default step, arms alternating, two rounds.
PROFILING TOOL.

    python tools/ab_prefetch_point.py [--steps 12 --warmup 4]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, ROOT)
    from open_flamingo_amd.train import step
    arm = int(sys.argv[2])             # 0: behind the backward; 1: the product's rule (gated block n_blocks // 12); 100 + k: at the start of gated block k's backward
    step.PREFETCH_IN_BACKWARD = arm >= 1
    if arm >= 100:
        from open_flamingo_amd.src import flamingo
        flamingo.Flamingo.prefetch_at_block = arm - 100
    elif arm == 1:
        from open_flamingo_amd.src import flamingo
        flamingo.Flamingo.prefetch_at_block = None          # the product's rule
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-reference-eager"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for rnd in range(2):
        for arm in [int(a) for a in os.environ.get("AB_ARMS", "1,0").split(",")]:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(arm)] + sys.argv[1:], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("prefetch_in_backward=%d %s" % (arm, line[-1][:330] if line else "FAILED " + out.stderr[-400:]), flush=True)
