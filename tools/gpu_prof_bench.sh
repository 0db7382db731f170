#!/bin/bash
# rocprofv3 kernel stats of the default bench command (3 warm-up + 3 timed steps) -> gpurun_out/<tag>_bench_kernel_stats.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r02b}; shift
OUT=/tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o run --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline "$@" > gpurun_out/${TAG}_prof_bench.log 2>&1
tail -2 gpurun_out/${TAG}_prof_bench.log | cut -c1-300
cp $(find $OUT -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
head -45 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-150
