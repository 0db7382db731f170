#!/bin/bash
# Round 3, third GPU call: all big-tile layouts on the 4-wave DMA kernel (asm DMA for K-strided operands, builtin for NT),
# DMA placement variants on TN/NN, epilogue ladder, bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03c}
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/${TAG}_gputests.log
tail -4 gpurun_out/${TAG}_gputests.log
timeout 500 python tools/bench_gemm_ab.py --family OF-3B --only-big > gpurun_out/${TAG}_gemm_ab_OF-3B.log 2>&1
grep "^{" gpurun_out/${TAG}_gemm_ab_OF-3B.log > gpurun_out/${TAG}_gemm_ab_OF-3B.jsonl; tail -2 gpurun_out/${TAG}_gemm_ab_OF-3B.log | cut -c1-200
timeout 300 python tools/bench_gemm_ab.py --ladder > gpurun_out/${TAG}_ladder.log 2>&1
grep "^{" gpurun_out/${TAG}_ladder.log > gpurun_out/${TAG}_gemm_epilogue_ladder.jsonl; cat gpurun_out/${TAG}_gemm_epilogue_ladder.jsonl; tail -2 gpurun_out/${TAG}_ladder.log | cut -c1-200
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_gemm_ab_*.jsonl")):
    print(f)
    for l in open(f):
        r = json.loads(l)
        arms = [k[:-3] for k in r if k.endswith("_ms")]
        print(f"  {r['name']:22s} {r['layout']} {str(r['MNK']):22s} " + "  ".join(f"{a} {r[a + '_ms'] * 1e3:.1f}/{r[a + '_tflops']:.0f}" for a in arms))
PY
( timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['all_gemm_ms_per_step'], d['roofline']['all_gemm_tflops'])"
