#!/bin/bash
# Round-3 closing measurements in one gpurun call; everything lands in gpurun_out/<tag>_*  (copied to profiles/r03_final_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03z}
# 0. the whole -m gpu suite
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -12 ) > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
python __graft_entry__.py smoke 2>&1 | tail -1
# 1. the bench line the driver will see (+ per-shape GEMM table)
( python bench.py --steps 10 --warmup 3 --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['all_gemm_ms_per_step'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
# 2. rocprofv3 kernel stats of the same command
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of3b_bench_kernel_stats.csv
grep "^{" gpurun_out/${TAG}_prof_bench.log | cut -c1-160
# 3. the other model families of BASELINE.json + the reference's two-pass step
for cfg in "OF-4B 32 2 256 of4b" "OF-9B 8 5 256 of9b_L256" "OF-9B 8 5 2048 of9b_L2048"; do
  set -- $cfg
  ( timeout 900 python bench.py --family $1 --batch $2 --T $3 --L $4 --steps 5 --warmup 2 --no-cpu-baseline --gemm-report gpurun_out/${TAG}_$5_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_$5_bench.json
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_$5_bench.json')); print('$5', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'])"
done
( timeout 600 python bench.py --laion-batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" ) > gpurun_out/${TAG}_two_pass_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_two_pass_bench.json')); print('two-pass', d['ms_per_step'], d['value'])"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof4_$TAG -o run --output-format csv -- python bench.py --family OF-4B --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof_of4b.log 2>&1
cp $(find /tmp/prof4_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of4b_bench_kernel_stats.csv
# 4. reference-equivalent eager step on this box
( timeout 600 python tests/perf_reference_eager.py OF-3B 32 2 256 2>&1 | grep "^{" ; timeout 600 python tests/perf_reference_eager.py OF-3B 32 2 256 --stock-towers 2>&1 | grep "^{" ) > gpurun_out/${TAG}_reference_eager.jsonl
cut -c1-330 gpurun_out/${TAG}_reference_eager.jsonl
# 5. kernel microbench + GEMM A/B vs round 2's closing library
( timeout 400 python tools/bench_kernels.py 2>&1 | grep "^{" ) > gpurun_out/${TAG}_kernel_microbench.jsonl
for fam in OF-3B OF-4B; do ( timeout 500 python tools/bench_gemm_ab.py --family $fam 2>&1 | grep "^{" ) > gpurun_out/${TAG}_gemm_ab_$fam.jsonl; done
python - <<PY
import json
for fam in ("OF-3B", "OF-4B"):
    for l in open("gpurun_out/${TAG}_gemm_ab_%s.jsonl" % fam):
        r = json.loads(l)
        print(f"  {fam} {r['name']:22s} {r['layout']} {str(r['MNK']):22s} old {r['old_ms']*1e3:.1f}/{r['old_tflops']:.0f}  new {r['new_ms']*1e3:.1f}/{r['new_tflops']:.0f}")
PY
