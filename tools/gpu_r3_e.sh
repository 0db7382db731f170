#!/bin/bash
# Round 3, fifth GPU call: whole -m gpu suite on the cleaned-up build, contention rehearsal, OF-4B with the vectorised rotary
# kernels, PMC traffic of the dominant GEMM launches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03e}
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/${TAG}_gputests.log
tail -5 gpurun_out/${TAG}_gputests.log
( timeout 600 python tools/rehearse_contention.py --steps 6 2>&1 | grep "^{" ) > gpurun_out/${TAG}_contention_rehearsal.jsonl
cat gpurun_out/${TAG}_contention_rehearsal.jsonl | cut -c1-120
( timeout 900 python bench.py --family OF-4B --batch 32 --T 2 --L 256 --steps 5 --warmup 2 --no-cpu-baseline --gemm-report gpurun_out/${TAG}_of4b_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_of4b_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_of4b_bench.json')); print('of4b', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'])"
bash tools/gpu_pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
tail -30 gpurun_out/${TAG}_pmc_traffic.txt
