#!/bin/bash
# (as run at commit dd27936: the ring kernel and the -DOF_KC_LINE layout were removed from the sources afterwards; results in profiles/)
# Round 3, seventh GPU call: full-line K-contiguous DMA pieces (-DOF_KC_LINE build) against the product, timing + counters;
# counters of the two-slot / ring 4-wave kernels and the vendor kernel on the same launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03g}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed: giving the box back"; exit 3; }
timeout 300 python tools/bench_gemm_ab.py tools/ab/libofhip_kcline.so --only-big --arms old,new > gpurun_out/${TAG}_gemm_ab_kcline_OF-3B.jsonl 2> gpurun_out/${TAG}_gemm_ab_kcline.err || { echo "A/B failed"; tail -5 gpurun_out/${TAG}_gemm_ab_kcline.err; }
python - "$TAG" <<'PY'
import json, sys
for l in open("gpurun_out/%s_gemm_ab_kcline_OF-3B.jsonl" % sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["name"], d["layout"], d["MNK"], "kcline", d["old_ms"], "product", d["new_ms"], "diff", d["max_abs_diff_old_new"])
PY
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU unhealthy after the A/B"; exit 4; }
bash tools/gpu_pmc_gemm_variants.sh ${TAG} --layouts NT,TN --safes 7,16 | head -150
