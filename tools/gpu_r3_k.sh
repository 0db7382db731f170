#!/bin/bash
# (as run at commit dd27936: the -DOF_W4_NO_UNROLL2 switch were removed from the sources afterwards; results in profiles/)
# Round 3, GPU call: the two-slot 4-wave kernel with scalar LDS addresses for its DMA pieces and (K-contiguous operands only) the
# steady state unrolled by two stages, against the rolled loop (-DOF_W4_NO_UNROLL2) and the previous build (generic LDS pointers)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03n}
timeout 150 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); print('gpu ok', float(x.sum()))" || { echo "GPU sanity check failed"; exit 3; }
timeout 400 python tools/bench_gemm_libs.py --libs product,tools/ab/libofhip_no_unroll2.so,tools/ab/libofhip_r03_head.so --safe 7 > gpurun_out/${TAG}_gemm_unroll2.jsonl 2> gpurun_out/${TAG}_err.txt || tail -5 gpurun_out/${TAG}_err.txt
cat gpurun_out/${TAG}_gemm_unroll2.jsonl
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "race_screen or fused_epilogues_at_benchmark or bit_reproducible" 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests.log
grep -E "passed|failed|error" gpurun_out/${TAG}_tests.log | tail -3
