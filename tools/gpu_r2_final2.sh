#!/bin/bash
# Round-2 (second half) closing measurements in one gpurun call; everything lands in gpurun_out/<tag>_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r02z}
# 1. the bench line the driver will see (+ per-shape GEMM table)
( python bench.py --steps 10 --warmup 3 --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['all_gemm_ms_per_step'], d['cpu_baseline'])"
# 2. rocprofv3 kernel stats of the same command
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of3b_bench_kernel_stats.csv
grep "^{" gpurun_out/${TAG}_prof_bench.log | cut -c1-200
# 3. the other model families of BASELINE.json
for cfg in "OF-4B 32 2 256 of4b" "OF-9B 8 5 256 of9b_L256" "OF-9B 8 5 2048 of9b_L2048"; do
  set -- $cfg
  ( timeout 900 python bench.py --family $1 --batch $2 --T $3 --L $4 --steps 5 --warmup 2 --no-cpu-baseline --gemm-report gpurun_out/${TAG}_$5_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_$5_bench.json
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_$5_bench.json')); print('$5', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'])"
done
# 4. reference-equivalent eager step on this box
( timeout 600 python tests/perf_reference_eager.py OF-3B 32 2 256 2>&1 | grep "^{" ; timeout 600 python tests/perf_reference_eager.py OF-3B 32 2 256 --stock-towers 2>&1 | grep "^{" ) > gpurun_out/${TAG}_reference_eager.jsonl
cut -c1-330 gpurun_out/${TAG}_reference_eager.jsonl
# 5. kernel microbench
( timeout 400 python tools/bench_kernels.py 2>&1 | grep "^{" ) > gpurun_out/${TAG}_kernel_microbench.jsonl
grep -vE '"gemm"' gpurun_out/${TAG}_kernel_microbench.jsonl | cut -c1-200
