"""Same-box A/B of whole-library builds at STEP level: bench.py's step, once per arm and round, each arm with Ops.default() bound to
another build of libofhip (tools/build_ab_variant.sh).  One short line per run.  PROFILING TOOL.

    python tools/ab_lib_builds.py product=open_flamingo_amd/csrc/libofhip.so nt=tools/ab/libofhip_wt3.so [-- bench.py flags]"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--arm":
    sys.path.insert(0, ROOT)
    import torch
    from open_flamingo_amd.hip import abi
    from open_flamingo_amd.hip.ops import Ops
    lib = ctypes.CDLL(os.path.abspath(sys.argv[2]))
    abi.declare(lib)
    Ops._default = Ops(lib, lambda: torch.cuda.current_stream().cuda_stream)
    sys.argv = ["bench.py"] + sys.argv[3:]
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    args = sys.argv[1:]
    extra = args[args.index("--") + 1:] if "--" in args else ["--steps", "12", "--warmup", "4", "--no-roofline"]
    arms = [a.split("=") for a in (args[:args.index("--")] if "--" in args else args)]
    for rnd in range(2):
        for name, path in arms:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", path, "--no-cpu-baseline", "--no-reference-eager"] + extra,
                                 capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(name, "FAILED", out.stderr[-300:], flush=True)
                continue
            d = json.loads(line[-1])
            r = d.get("roofline") or {}
            # loss_last_step: an arm whose build computes garbage (NaN / zeros draw less power: FASTER steps) must not pass as a win
            print(json.dumps({"arm": name, "round": rnd, "ms_per_step": d["ms_per_step"], "loss_last_step": d.get("loss_last_step"),
                              "all_gemm_tflops": r.get("all_gemm_tflops"),
                              "dominant_tflops": r.get("achieved")}), flush=True)
