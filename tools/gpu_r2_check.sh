#!/bin/bash
# One gpurun call: GPU parity tests (full log), full bench with per-shape GEMM report.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r02}
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl) " | tail -60 ) > gpurun_out/${TAG}_pytest.log
tail -40 gpurun_out/${TAG}_pytest.log
( timeout 600 python bench.py --steps 8 --warmup 3 --gemm-report gpurun_out/${TAG}_gemm_report.jsonl 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_bench.json
