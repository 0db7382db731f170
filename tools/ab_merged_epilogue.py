"""Same-box A/B of the step epilogue's merged launches (train/optim.py: FlatAdamW.merged_launches -- of_sumsq_partial_multi /
of_adamw_clip_multi, ABI v12): bench.py's default step (BASELINE config 2) with the per-bucket norm passes (bit 0) and / or AdamW
segments (bit 1) as one launch each per 32 against one launch per bucket / segment, arms alternating, two rounds.  PROFILING TOOL.

    python tools/ab_merged_epilogue.py [--steps 12 --warmup 4]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, ROOT)
    from open_flamingo_amd.train import optim
    merged = int(sys.argv[2])          # bit 0: norm passes, bit 1: AdamW segments
    init = optim.FlatAdamW.__init__

    def patched(self, *a, **kw):
        init(self, *a, **kw)
        self.merged_launches = merged
    optim.FlatAdamW.__init__ = patched
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-reference-eager"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for rnd in range(2):
        for merged in (3, 1, 2, 0):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(merged)] + sys.argv[1:], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("merged_launches=%d %s" % (merged, line[-1][:330] if line else "FAILED " + out.stderr[-400:]), flush=True)
