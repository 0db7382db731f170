"""Same-box A/B of the big-tile MFMA shape at step level: bench.py's step with of_gemm's own selection (16x16x32 kernel) and with
every big-tile launch forced onto the 32x32x16 kernel (safe = 7).  One bench JSON line per arm.  PROFILING TOOL.

    python tools/ab_big_tile_mfma.py [--steps 6 --warmup 3] [--family OF-4B ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    force = sys.argv[2] == "1"
    sys.path.insert(0, ROOT)
    from open_flamingo_amd.hip.ops import Ops
    if force:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from tools_lib import tools_ops      # the 32x32x16 kernel lives in tools/libofhip_tools.so only (round 6)
        forced = tools_ops()
        orig = Ops.gemm
        def gemm(self, A, B, out, *, ta=False, tb=False, safe=0, **kw):
            M, K = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
            N = B.shape[1] if tb else B.shape[0]
            if safe == 0 and Ops.kernel_label(M, N, K, ta, tb) == "w4m256":
                return orig(forced, A, B, out, ta=ta, tb=tb, safe=7, **kw)
            return orig(self, A, B, out, ta=ta, tb=tb, safe=safe, **kw)
        Ops.gemm = gemm
    sys.argv = ["bench.py"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for rnd in range(2):
        for force in (0, 1):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(force), "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("%s %s" % ("32x32x16(safe=7)" if force else "16x16x32(product)", line[-1] if line else "FAILED " + out.stderr[-400:]), flush=True)
