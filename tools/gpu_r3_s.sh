#!/bin/bash
# Round 3, GPU call: 16x16x32 big-tile kernel as of_gemm's selection -- whole -m gpu suite, bench line of cfg-2, PMC traffic of the
# dominant launches on the new kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03s}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/${TAG}_gputests.log
grep -E "passed|failed|error" gpurun_out/${TAG}_gputests.log | tail -3
( timeout 600 python bench.py --steps 6 --warmup 3 --gemm-report gpurun_out/${TAG}_of3b_gemm_report.jsonl 2>gpurun_out/${TAG}_bench_err.txt | grep "^{" ) > gpurun_out/${TAG}_of3b_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_of3b_bench.json')); r=d['roofline']; print('of3b', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['kernel'], r['avg_launch_ms'], r['all_gemm_tflops'], r['all_gemm_ms_per_step'], r['traffic'])"
bash tools/gpu_pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
tail -30 gpurun_out/${TAG}_pmc_traffic.txt
