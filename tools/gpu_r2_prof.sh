#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py for the three model families (2 timed steps each)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r02p}
prof() { name=$1; shift
  rm -rf gpurun_out/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o bench --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > gpurun_out/${TAG}_${name}_prof.log 2>&1
  f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${TAG}_${name}_kernel_stats.csv
  rm -rf gpurun_out/prof_$name
  grep "^{" gpurun_out/${TAG}_${name}_prof.log | cut -c1-200
  head -12 gpurun_out/${TAG}_${name}_kernel_stats.csv | cut -c1-160
}
prof of3b
prof of4b --family OF-4B
prof of9b --family OF-9B --batch 8 --T 5 --L 256
