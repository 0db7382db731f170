cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r04p
sha256sum open_flamingo_amd/csrc/libofhip.so | cut -c1-16 > gpurun_out/r04p/lib.sha
( timeout 600 python -m pytest tests/test_gpu_path.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "early_norm or loopback or epilogue or prefetch or checkpoint" 2>&1 | tail -8 ) > gpurun_out/r04p/tests.log
tail -3 gpurun_out/r04p/tests.log
( timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "layernorm or ln_" 2>&1 | tail -4 ) > gpurun_out/r04p/tests_ln.log
tail -2 gpurun_out/r04p/tests_ln.log
for i in 1 2; do
  for mode in "" "--late-norm"; do
    ( timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-reference-eager --no-roofline $mode 2>&1 | grep "^{" ) > gpurun_out/r04p/bench_${i}_${mode:-early}.json
    python -c "import json; print('$i', '${mode:-early}', json.load(open('gpurun_out/r04p/bench_${i}_${mode:-early}.json'))['ms_per_step'])"
  done
done
