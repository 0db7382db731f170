#!/bin/bash
# (as run at commit dd27936: the ring kernel were removed from the sources afterwards; results in profiles/)
# Round 3, sixth GPU call: the 4-wave kernel on the four-slot half-stage ring (gemm_w4r.hip, safe = 16) -- race screen and
# fused-epilogue parity on hardware, K sweep and same-box A/B against the two-slot 4-wave kernel (safe = 7).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03f}
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "race_screen or fused_epilogues_at_benchmark or bit_reproducible" 2>&1 | tail -15 ) > gpurun_out/${TAG}_ring_tests.log
grep -E "passed|failed|error" gpurun_out/${TAG}_ring_tests.log | tail -3
( timeout 600 python tools/bench_gemm_ab.py --ksweep 2>&1 | grep "^{" ) > gpurun_out/${TAG}_gemm_ksweep_ring.jsonl
cat gpurun_out/${TAG}_gemm_ksweep_ring.jsonl | cut -c1-200
( timeout 600 python tools/bench_gemm_ab.py --only-big --arms new,new_w4dma256,new_w4ring256 2>&1 | grep "^{" ) > gpurun_out/${TAG}_gemm_ab_ring_OF-3B.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r03f_gemm_ab_ring_OF-3B.jsonl"):
    d = json.loads(l)
    print(d["name"], d["layout"], d["MNK"], {k: v for k, v in d.items() if k.endswith("_ms")})
PY
