#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-w4}
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "race_screen or big_tile" 2>&1 | tail -15 ) > gpurun_out/${TAG}_pytest.log
tail -6 gpurun_out/${TAG}_pytest.log
( timeout 600 python tools/bench_gemm_w4.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${TAG}_ab.jsonl
cat gpurun_out/${TAG}_ab.jsonl
