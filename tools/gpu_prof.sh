#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py (2 timed steps); summary -> gpurun_out/<tag>_kernel_stats.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-prof}
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o bench --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_prof_bench.log 2>&1
tail -2 gpurun_out/${TAG}_prof_bench.log | grep -v amdgpu
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
head -45 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
