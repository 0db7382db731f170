"""Same-box A/B of the frozen MPT MLP routing (train/frozen_blocks.py: _MLP_FUSED_UP / _MLP_FUSED_DOWN / _MLP_FUSED_DGELU): bench.py's step with each
combination.  One bench JSON line per arm, prefixed by the arm.  PROFILING TOOL.

    python tools/ab_frozen_mlp.py [--steps 6 --warmup 3]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    up, down, dgelu = sys.argv[2] == "1", sys.argv[3] == "1", sys.argv[4] == "1"
    sys.path.insert(0, ROOT)
    from open_flamingo_amd.train import frozen_blocks
    frozen_blocks._MLP_FUSED_UP, frozen_blocks._MLP_FUSED_DOWN, frozen_blocks._MLP_FUSED_DGELU = up, down, dgelu
    frozen_blocks._MPT_GEMMS_NATIVE = os.environ.get("AB_NATIVE") == "1"          # the block's plain GEMMs as of_gemm launches
    sys.argv = ["bench.py"] + sys.argv[5:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for rnd in range(2):
        arms = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1))
        if os.environ.get("AB_ARMS"):          # e.g. AB_ARMS=001,101,011,111
            arms = tuple(tuple(int(c) for c in a) for a in os.environ["AB_ARMS"].split(","))
        for arm in arms:
            up, down, dgelu = arm[:3]
            native = arm[3] if len(arm) > 3 else 0
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(up), str(down), str(dgelu), "--no-cpu-baseline"] + sys.argv[1:],
                                 capture_output=True, text=True, env=dict(os.environ, AB_NATIVE=str(native)))
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("up=%d down=%d dgelu=%d native=%d %s" % (up, down, dgelu, native, line[-1] if line else "FAILED " + out.stderr[-400:]), flush=True)
