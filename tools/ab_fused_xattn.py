"""Same-box A/B of the fused attention branch of the gated blocks (hip/path.py: FUSED_XATTN -> csrc/xattn_fused.hip): bench.py's step
with the ONE launch per block and with the five separate launches it replaces (LN, to_q, attention core, to_out + gate + residual,
the FFN's LN).  One bench JSON line per arm, arms alternating.  PROFILING TOOL."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, ROOT)
    if sys.argv[2] == "0":
        from open_flamingo_amd.hip import path as P
        P.FUSED_XATTN = False
    sys.argv = ["bench.py"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    import json
    for rnd in range(2):
        for on in (1, 0):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(on), "--no-cpu-baseline", "--no-reference-eager"] + sys.argv[1:],
                                 capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("%s FAILED %s" % ("fused   " if on else "separate", out.stderr[-600:]), flush=True)
                continue
            d = json.loads(line[-1])
            print(json.dumps({"arm": "fused" if on else "separate launches", "ms_per_step": d["ms_per_step"], "images_per_s": d["value"],
                              "loss_last_step": d.get("loss_last_step"), "all_gemm_tflops": d["roofline"].get("all_gemm_tflops")}), flush=True)
