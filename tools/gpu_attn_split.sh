cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "ragged_head_count or single_pass or compact" 2>&1 | tail -4 ) > gpurun_out/r06zs_gputests_attn_split.log
cat gpurun_out/r06zs_gputests_attn_split.log
( timeout 1500 python tools/ab_lib_builds.py hybrid=open_flamingo_amd/csrc/libofhip.so two_pass=tools/ab/libofhip_attn_bwd_two_pass.so -- --config 5 --steps 10 --warmup 3 --no-roofline 2>&1 ) > gpurun_out/r06zs_ab_attn_bwd_split_cfg5_step.jsonl
cat gpurun_out/r06zs_ab_attn_bwd_split_cfg5_step.jsonl
