cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r04_final
sha256sum open_flamingo_amd/csrc/libofhip.so > gpurun_out/${TAG}_lib.sha
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -12 ) > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
python __graft_entry__.py smoke 2>&1 | tail -1
( python bench.py --steps 20 --warmup 5 --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['all_gemm_tflops'], d['roofline']['all_gemm_ms_per_step'], 'vs_baseline', d['vs_baseline'], d.get('vs_reference_stock_towers'), 'floor', d['floor']['step_frac_of_floor'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-reference-eager > gpurun_out/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of3b_bench_kernel_stats.csv
grep "^{" gpurun_out/${TAG}_prof_bench.log | cut -c1-160
for c in 4 5 5L; do
  ( timeout 900 python bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline --no-reference-eager-stock --gemm-report gpurun_out/${TAG}_cfg${c}_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_cfg${c}_bench.json
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_cfg${c}_bench.json')); print('cfg$c', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'], 'vs_baseline', d.get('vs_baseline'), 'floor', d['floor']['step_frac_of_floor'])"
done
( timeout 600 python bench.py --laion-batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-reference-eager 2>&1 | grep "^{" ) > gpurun_out/${TAG}_two_pass_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_two_pass_bench.json')); print('two-pass', d['ms_per_step'], d['value'])"
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-eager --no-roofline --no-vision-prefetch 2>&1 | grep "^{" ) > gpurun_out/${TAG}_no_prefetch_bench.json
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-eager --no-roofline 2>&1 | grep "^{" ) > gpurun_out/${TAG}_prefetch_bench.json
python -c "import json; print('no prefetch', json.load(open('gpurun_out/${TAG}_no_prefetch_bench.json'))['ms_per_step'], 'prefetch', json.load(open('gpurun_out/${TAG}_prefetch_bench.json'))['ms_per_step'])"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof5_$TAG -o run --output-format csv -- python bench.py --config 5 --steps 3 --warmup 2 --no-cpu-baseline --no-reference-eager > gpurun_out/${TAG}_prof_cfg5.log 2>&1
cp $(find /tmp/prof5_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_cfg5_bench_kernel_stats.csv
