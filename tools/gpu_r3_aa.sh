#!/bin/bash
# Round 3, GPU call: s_barrier fenced for the compiler (of_platform.h) -- the diagnosis probe on the unrolled-NT build, whole -m gpu suite
# on the product, A/B product vs the unrolled-NT build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03aa}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
timeout 300 python tools/probes/w4m_unroll_diag.py tools/ab/libofhip_w4m_unroll2.so 2>gpurun_out/${TAG}_err.txt | tee gpurun_out/${TAG}_w4m_unroll_diag.jsonl
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/${TAG}_gputests.log
grep -E "passed|failed|error" gpurun_out/${TAG}_gputests.log | tail -3
timeout 600 python tools/bench_gemm_ab.py tools/ab/libofhip_w4m_unroll2.so --only-big --arms old,new > gpurun_out/${TAG}_gemm_ab_unroll2_OF-3B.jsonl 2>> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
python - "$TAG" <<'PY'
import json, sys
for l in open("gpurun_out/%s_gemm_ab_unroll2_OF-3B.jsonl" % sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-24s %s %-20s unrolled %.4f product %.4f  %+5.1f%%  diff %s" % (d["name"], d["layout"], d["MNK"], d["old_ms"], d["new_ms"], 100 * (d["old_ms"] / d["new_ms"] - 1), d["max_abs_diff_old_new"]))
PY
