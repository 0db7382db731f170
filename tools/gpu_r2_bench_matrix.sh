#!/bin/bash
# One gpurun call: bench.py A/B of the step-level options on OF-3B cfg-2, then BASELINE configs 4 and 5 (per-GPU batch),
# each with its per-shape GEMM report.  Outputs under gpurun_out/<tag>_*.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r02m}
run() { name=$1; shift; ( timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/${TAG}_${name}.json; cut -c1-330 gpurun_out/${TAG}_${name}.json; }
run default --gemm-report gpurun_out/${TAG}_default_gemm.jsonl
run hf_loss --lm-loss hf --no-roofline
run dense_rows --dense-embedding-rows --no-roofline
run of4b --family OF-4B --gemm-report gpurun_out/${TAG}_of4b_gemm.jsonl
run of9b_L256 --family OF-9B --batch 8 --T 5 --L 256 --gemm-report gpurun_out/${TAG}_of9b_L256_gemm.jsonl
run of9b_L2048 --family OF-9B --batch 8 --T 5 --L 2048 --gemm-report gpurun_out/${TAG}_of9b_L2048_gemm.jsonl
