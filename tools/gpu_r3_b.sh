#!/bin/bash
# Round 3, second GPU call: LDS-DMA by inline asm (no compiler-inserted vmcnt(0) in front of the transposed-fragment reads).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03b}
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/${TAG}_gputests.log
tail -8 gpurun_out/${TAG}_gputests.log
for fam in OF-3B OF-4B OF-9B; do
  timeout 500 python tools/bench_gemm_ab.py --family $fam > gpurun_out/${TAG}_gemm_ab_$fam.log 2>&1
  grep "^{" gpurun_out/${TAG}_gemm_ab_$fam.log > gpurun_out/${TAG}_gemm_ab_$fam.jsonl; tail -3 gpurun_out/${TAG}_gemm_ab_$fam.log | cut -c1-300
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_gemm_ab_*.jsonl")):
    print(f)
    for l in open(f):
        r = json.loads(l)
        arms = [k[:-3] for k in r if k.endswith("_ms")]
        print(f"  {r['name']:22s} {r['layout']} {str(r['MNK']):22s} " + "  ".join(f"{a} {r[a + '_ms'] * 1e3:.1f}us/{r[a + '_tflops']:.0f}" for a in arms))
PY
( timeout 900 python bench.py --steps 10 --warmup 3 --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['all_gemm_ms_per_step'], d['roofline']['all_gemm_tflops'])"
( timeout 400 python tools/bench_kernels.py 2>&1 | grep "^{" ) > gpurun_out/${TAG}_kernel_microbench.jsonl
grep -vE '"gemm"' gpurun_out/${TAG}_kernel_microbench.jsonl | cut -c1-220
