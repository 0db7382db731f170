#!/bin/bash
# Round 3, first GPU call: the whole -m gpu suite on the new build, same-box GEMM A/B against round 2's closing library,
# the bench line, rocprofv3 kernel stats of the same command.  Everything lands in gpurun_out/<tag>_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03a}
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())"
nproc
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/${TAG}_gputests.log
tail -25 gpurun_out/${TAG}_gputests.log
( timeout 400 python tools/bench_gemm_ab.py --family OF-3B 2>&1 | grep "^{" ) > gpurun_out/${TAG}_gemm_ab_of3b.jsonl
( timeout 300 python tools/bench_gemm_ab.py --family OF-4B 2>&1 | grep "^{" ) > gpurun_out/${TAG}_gemm_ab_of4b.jsonl
( timeout 300 python tools/bench_gemm_ab.py --family OF-9B 2>&1 | grep "^{" ) > gpurun_out/${TAG}_gemm_ab_of9b.jsonl
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03a_gemm_ab_*.jsonl")):
    print(f)
    for l in open(f):
        r = json.loads(l)
        print(f"  {r['name']:22s} {r['layout']} {str(r['MNK']):22s} old {r['old_ms']:.4f} ({r['old_tflops']:.0f})  new {r['new_ms']:.4f} ({r['new_tflops']:.0f})"
              + (f"  mid128 {r['new_mid128_tflops']:.0f} big256 {r['new_big256_tflops']:.0f}" if 'new_mid128_ms' in r else ""))
PY
( timeout 900 python bench.py --steps 10 --warmup 3 --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['all_gemm_ms_per_step'], d['roofline']['all_gemm_tflops']); print(d['cpu_baseline'])"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of3b_bench_kernel_stats.csv
grep "^{" gpurun_out/${TAG}_prof_bench.log | cut -c1-200
head -30 gpurun_out/${TAG}_of3b_bench_kernel_stats.csv | cut -c1-180
