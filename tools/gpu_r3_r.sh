#!/bin/bash
# Round 3, GPU call: the 4-wave kernel on 16x16x32 MFMAs (gemm_w4m.hip, safe = 16) -- race screen and fused-epilogue parity on
# hardware, K sweep and same-box A/B against the 32x32x16 kernel (safe = 7).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03r}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "race_screen or fused_epilogues_at_benchmark or bit_reproducible" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log
grep -E "passed|failed|error|assert" gpurun_out/${TAG}_tests.log | tail -8
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU unhealthy after the tests"; exit 4; }
timeout 600 python tools/bench_gemm_ab.py --ksweep > gpurun_out/${TAG}_gemm_ksweep_w4m.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
cut -c1-220 gpurun_out/${TAG}_gemm_ksweep_w4m.jsonl
timeout 600 python tools/bench_gemm_ab.py --only-big --arms new_w4dma256,new_w4m256 > gpurun_out/${TAG}_gemm_ab_w4m_OF-3B.jsonl 2>> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
python - "$TAG" <<'PY'
import json, sys
for l in open("gpurun_out/%s_gemm_ab_w4m_OF-3B.jsonl" % sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-24s %s %-20s 32x32 %.4f  16x16 %.4f  %+5.1f%%" % (d["name"], d["layout"], d["MNK"], d["new_w4dma256_ms"], d["new_w4m256_ms"], 100 * (d["new_w4m256_ms"] / d["new_w4dma256_ms"] - 1)))
PY
