#!/bin/bash
# Re-tune the frozen towers' vendor GEMMs with TunableOp's ROTATING operand buffers larger than the Infinity Cache (1 GiB): the selection for
# operands that come from HBM, which is what a launch inside a train step sees, then A/B the two tables on this box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
CFG=${CFG:-2}          # BASELINE configuration (bench.py --config): 2 (OF-3B), 4 (OF-4B), 5 / 5L (OF-9B)
export PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop_gfx950_cold_cfg${CFG}_.csv
rm -f $PWD/gpurun_out/tunableop_gfx950_cold_cfg${CFG}_*.csv
( time PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=1024 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${TUNE_MS:-12} \
  PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3 timeout ${TUNE_TIMEOUT:-900} python bench.py --config $CFG --vendor-gemm-table default --steps 2 --warmup 2 --no-cpu-baseline --no-reference-eager --no-roofline 2>&1 | grep "^{" | cut -c1-160 ) 2>&1 | tail -5
ls -la gpurun_out/tunableop_gfx950_cold_cfg${CFG}_*.csv; wc -l gpurun_out/tunableop_gfx950_cold_cfg${CFG}_*.csv
COLD=$(ls gpurun_out/tunableop_gfx950_cold_cfg${CFG}_*.csv | head -1)
for rnd in $(seq 1 ${ROUNDS:-2}); do
  for arm in hot cold; do
    if [ $arm = cold ]; then T=$COLD; else T=open_flamingo_amd/train/tuned/tunableop_gfx950_of3b_cfg2.csv; fi
    OF_TUNED_TABLE=$T OF_CFG=$CFG python - <<'PY'
import os, sys, json, subprocess
t = os.environ["OF_TUNED_TABLE"]
code = ("import sys, runpy; sys.argv = ['bench.py', '--config', '%s', '--steps', '%s', '--warmup', '4', '--no-cpu-baseline', '--no-reference-eager'];"
        "from open_flamingo_amd.train import towers; f = towers.use_tuned_vendor_gemms; towers.use_tuned_vendor_gemms = lambda table=None: f(%r);"
        "runpy.run_path('bench.py', run_name='__main__')" % (os.environ["OF_CFG"], "6" if os.environ["OF_CFG"] == "5L" else "12", os.path.abspath(t)))
out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
d = json.loads(line[-1]) if line else None
print(os.path.basename(t), d["ms_per_step"] if d else "FAILED " + out.stderr[-300:], d and d["loss_last_step"], d and d["config"].get("vendor_gemm_table"), flush=True)
PY
  done
done
