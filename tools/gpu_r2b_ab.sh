#!/bin/bash
# One gpurun call: new GPU tests of the fused frozen blocks, kernel microbench, bench A/B of the frozen-tower forms.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r02b}
timeout 900 python -m pytest tests/test_gpu_path.py -x -q -k "frozen or clip_tower or layernorm_fwd_add or fused_step" 2>&1 | tail -15 | tee gpurun_out/${TAG}_tests.log
( timeout 400 python tools/bench_kernels.py 2>&1 | grep "^{" ) > gpurun_out/${TAG}_kernels.jsonl
grep -E "ln_|sumsq|vit|mpt" gpurun_out/${TAG}_kernels.jsonl | cut -c1-250
for cfg in "modules modules" "fused modules" "fused sdpa" "fused libofhip"; do
  set -- $cfg
  echo "== lm-blocks $1 vision $2"
  ( timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --lm-blocks $1 --vision $2 2>&1 | grep "^{" ) > gpurun_out/${TAG}_bench_$1_$2.json
  python - gpurun_out/${TAG}_bench_$1_$2.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(d["ms_per_step"], d["value"], d["loss"], d["roofline"]["achieved"], d["roofline"]["all_gemm_ms_per_step"])
except Exception as e:
    print("bench failed", e)
PY
done
