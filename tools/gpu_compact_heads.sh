cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q --timeout 600 -p no:cacheprovider -k "compact or neox or attention or single_pass or attn" 2>&1 | tail -8 ) > gpurun_out/r06zp_gputests_compact.log
cat gpurun_out/r06zp_gputests_compact.log
( timeout 300 python tools/probes/compact_heads_probe.py 2>&1 | grep "^{\|Error\|error" ) > gpurun_out/r06zp_compact_heads_probe.jsonl
cat gpurun_out/r06zp_compact_heads_probe.jsonl
( timeout 900 python tools/ab_neox_compact_heads.py --steps 8 --warmup 3 2>&1 ) > gpurun_out/r06zp_ab_neox_compact_heads_step.jsonl
cut -c1-330 gpurun_out/r06zp_ab_neox_compact_heads_step.jsonl
