#!/bin/bash
# Round 3, fourth GPU call: fused GPT-NeoX blocks (OF-4B), staggered mid kernel, all three model families.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03d}
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q --timeout 900 -p no:cacheprovider -k "neox or mid or gemm or dot or cfg4 or layout" 2>&1 | tail -15 ) > gpurun_out/${TAG}_gputests.log
tail -4 gpurun_out/${TAG}_gputests.log
timeout 500 python tools/bench_gemm_ab.py --family OF-3B > gpurun_out/${TAG}_gemm_ab_OF-3B.log 2>&1
grep "^{" gpurun_out/${TAG}_gemm_ab_OF-3B.log > gpurun_out/${TAG}_gemm_ab_OF-3B.jsonl
python - <<PY
import json
for l in open("gpurun_out/${TAG}_gemm_ab_OF-3B.jsonl"):
    r = json.loads(l)
    if (r.get("tiles256") or 0) >= 128: continue
    arms = [k[:-3] for k in r if k.endswith("_ms") and k[:-3] in ("old", "new", "builtin_dma")]
    print(f"  {r['name']:22s} {r['layout']} {str(r['MNK']):22s} " + "  ".join(f"{a} {r[a + '_ms'] * 1e3:.1f}/{r[a + '_tflops']:.0f}" for a in arms))
PY
( timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('OF-3B', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['all_gemm_ms_per_step'], d['roofline']['all_gemm_tflops'])"
for cfg in "OF-4B 32 2 256 of4b" "OF-9B 8 5 256 of9b_L256" "OF-9B 8 5 2048 of9b_L2048"; do
  set -- $cfg
  ( timeout 900 python bench.py --family $1 --batch $2 --T $3 --L $4 --steps 5 --warmup 2 --no-cpu-baseline --gemm-report gpurun_out/${TAG}_$5_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_$5_bench.json
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_$5_bench.json')); print('$5', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'], d['roofline']['traffic'])"
done
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run --output-format csv -- python bench.py --family OF-4B --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof_of4b.log 2>&1
cp $(find /tmp/prof_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of4b_bench_kernel_stats.csv
head -28 gpurun_out/${TAG}_of4b_bench_kernel_stats.csv | cut -c1-150
