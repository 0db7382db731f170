#!/bin/bash
# (as run at commit dd27936: the -DOF_ABL_* switches were removed from the sources afterwards; results in profiles/)
# Round 3, eighth GPU call: where the 4-wave kernel's per-stage wait goes -- timing ablations (no vmcnt wait / no barrier / neither)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03h}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
timeout 400 python tools/bench_gemm_libs.py --libs product,tools/ab/libofhip_abl_novm.so,tools/ab/libofhip_abl_nobar.so,tools/ab/libofhip_abl_neither.so > gpurun_out/${TAG}_gemm_wait_ablation.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
cat gpurun_out/${TAG}_gemm_wait_ablation.jsonl
