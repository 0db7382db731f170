#!/bin/bash
# One gpurun call that decides about tools/patches/overwrite_fresh_grads.patch: baseline bench, apply the patch in the
# box's scratch copy, the GPU tests that exercise it, bench again.  Nothing is changed in the repository; apply the
# patch locally (git apply) only if this prints two passing test lines and a faster second bench line.
# (No rocprofv3 here: profiler calls were charged 5-10x their run time in round 1.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
flt() { grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl) |amdgpu.ids"; }
( timeout 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | flt | tail -1 ) > gpurun_out/patch_ab_before.json
cut -c1-220 gpurun_out/patch_ab_before.json
git apply tools/patches/overwrite_fresh_grads.patch 2>/dev/null || patch -p1 < tools/patches/overwrite_fresh_grads.patch
( timeout 300 python -m pytest tests/test_gpu_path.py tests/test_boundary_flamingo.py -m gpu -q -x 2>&1 | flt | grep -E "passed|failed|FAILED|Error" ) | tee gpurun_out/patch_ab_pytest.log
( timeout 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | flt | tail -1 ) > gpurun_out/patch_ab_after.json
cut -c1-220 gpurun_out/patch_ab_after.json
