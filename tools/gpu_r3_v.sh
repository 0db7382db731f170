#!/bin/bash
# Round 3, GPU call: the 8-wave 128x128 kernel on 16x16x32 MFMAs + the new big-tile kernel's NT steady state unrolled by two, against
# the build before them (tools/ab/libofhip_w4m_rolled.so): kernel tests, whole OF-3B launch table with results compared.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03v}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm or gate_gradient" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log
grep -E "passed|failed|error|assert" gpurun_out/${TAG}_tests.log | tail -8
timeout 600 python tools/bench_gemm_ab.py tools/ab/libofhip_w4m_rolled.so --arms old,new > gpurun_out/${TAG}_gemm_ab_OF-3B.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
python - "$TAG" <<'PY'
import json, sys
for l in open("gpurun_out/%s_gemm_ab_OF-3B.jsonl" % sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-24s %s %-20s before %.4f new %.4f  %+5.1f%%  diff %s" % (d["name"], d["layout"], d["MNK"], d["old_ms"], d["new_ms"], 100 * (d["new_ms"] / d["old_ms"] - 1), d["max_abs_diff_old_new"]))
PY
