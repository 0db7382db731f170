#!/bin/bash
# Round 3, ninth GPU call: channel-camping probe -- the product's big-tile launches with padded operand row pitches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03i}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
timeout 400 python tools/bench_gemm_libs.py --libs product --pads 0,64,128,192,1024,2048 > gpurun_out/${TAG}_gemm_pitch_probe.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
cat gpurun_out/${TAG}_gemm_pitch_probe.jsonl
