#!/bin/bash
# Round 3, GPU call: fp32-output epilogues with the split lane map (each 16-byte store contiguous with its neighbours') against the
# build before it (tools/ab/libofhip_pre_split.so): kernel tests, whole OF-3B launch table with results compared, epilogue probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03ac}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm or gate_gradient" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log
grep -E "passed|failed|error|assert" gpurun_out/${TAG}_tests.log | tail -8
timeout 600 python tools/bench_gemm_ab.py tools/ab/libofhip_pre_split.so --arms old,new > gpurun_out/${TAG}_gemm_ab_OF-3B.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
python - "$TAG" <<'PY'
import json, sys
for l in open("gpurun_out/%s_gemm_ab_OF-3B.jsonl" % sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-24s %s %-20s epi %d before %.4f new %.4f  %+5.1f%%  diff %s" % (d["name"], d["layout"], d["MNK"], d["epi"], d["old_ms"], d["new_ms"], 100 * (d["new_ms"] / d["old_ms"] - 1), d["max_abs_diff_old_new"]))
PY
