#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-k}
( timeout 400 python tools/bench_kernels.py 2>&1 | grep "^{" ) > gpurun_out/${TAG}_kernels.jsonl
grep -vE '"gemm"' gpurun_out/${TAG}_kernels.jsonl | cut -c1-250
