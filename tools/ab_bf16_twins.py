"""Same-box A/B of the bf16-twin hand-off between backwards (hip/path.py: offer_bf16_twin): bench.py's step with it and with the
offers disabled (every backward casts its incoming gradient itself).  One bench JSON line per arm.  PROFILING TOOL."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, ROOT)
    if sys.argv[2] == "0":
        from open_flamingo_amd.hip import path as P
        P.TWINS = False
    sys.argv = ["bench.py"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for rnd in range(2):
        for on in (1, 0):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(on), "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("%s %s" % ("twins on " if on else "twins off", line[-1] if line else "FAILED " + out.stderr[-400:]), flush=True)
