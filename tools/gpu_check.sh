#!/bin/bash
# One gpurun call: GPU parity tests, kernel microbench, full bench, rocprof kernel stats.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-run}
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl) " | tail -25 ) > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
( timeout 300 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${TAG}_kernels.jsonl
cat gpurun_out/${TAG}_kernels.jsonl
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm-report gpurun_out/${TAG}_gemm_report.jsonl 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_bench.json
