#!/bin/bash
# (as run at commit dd27936: the ring kernel and the -DOF_*_PLACE_NOREAD switches were removed from the sources afterwards; results in profiles/)
# Round 3, tenth GPU call: DMA pieces only in MFMA gaps without fragment reads (two-slot kernel: dense in gaps 8-15 of the two
# request phases; ring kernel: every other gap of 8-15 of every phase) against the product placements
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03j}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
for safe in 7 16; do
timeout 400 python tools/bench_gemm_libs.py --libs product,tools/ab/libofhip_place_noread.so --safe $safe > gpurun_out/${TAG}_gemm_place_noread_safe$safe.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
cat gpurun_out/${TAG}_gemm_place_noread_safe$safe.jsonl
done
