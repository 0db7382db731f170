#!/bin/bash
# Round 3, GPU call: scalar-LDS-address DMA in the 4-wave and the 8-wave 128x128 kernels -- whole OF-3B launch table against the
# previous build (results compared), the dense no-read-gap placement against the product, then the whole -m gpu suite and a bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03o}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
timeout 400 python tools/bench_gemm_libs.py --libs product,tools/ab/libofhip_place_noread.so --safe 7 > gpurun_out/${TAG}_gemm_place_noread.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
cat gpurun_out/${TAG}_gemm_place_noread.jsonl
timeout 600 python tools/bench_gemm_ab.py tools/ab/libofhip_r03_head.so --arms old,new > gpurun_out/${TAG}_gemm_ab_vs_r03_head_OF-3B.jsonl 2>> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
python - "$TAG" <<'PY'
import json, sys
for l in open("gpurun_out/%s_gemm_ab_vs_r03_head_OF-3B.jsonl" % sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-24s %s %-20s head %.4f new %.4f  %+5.1f%%  diff %s" % (d["name"], d["layout"], d["MNK"], d["old_ms"], d["new_ms"], 100 * (d["new_ms"] / d["old_ms"] - 1), d["max_abs_diff_old_new"]))
PY
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/${TAG}_gputests.log
grep -E "passed|failed|error" gpurun_out/${TAG}_gputests.log | tail -3
( timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" ) > gpurun_out/${TAG}_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print('of3b', d['ms_per_step'], d['value'], d['roofline'])"
