#!/bin/bash
# Round 3, GPU call: attention output stores as 16-byte stores after a row-pair exchange (v_permlane16_swap) against the build
# before (tools/ab/libofhip_pre_attn_store.so): attention kernel tests + A/B on the step's attention launches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03ad}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q --timeout 600 -p no:cacheprovider -k "attn or attention or xattn or perceiver or mpt or neox or clip" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log
grep -E "passed|failed|error|assert" gpurun_out/${TAG}_tests.log | tail -8
timeout 400 python tools/bench_attn_ab.py tools/ab/libofhip_pre_attn_store.so > gpurun_out/${TAG}_attn_ab.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
cat gpurun_out/${TAG}_attn_ab.jsonl
