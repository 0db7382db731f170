#!/bin/bash
# Tune the vendor-library GEMMs of the frozen towers with PyTorch TunableOp on this box, then measure with the tuned table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop_gfx950.csv
rm -f $PYTORCH_TUNABLEOP_FILENAME
echo "== untuned"; python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo "== tuning run"
( time PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 timeout 1200 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep "^{" | cut -c1-200 ) 2>&1 | tail -5
ls -la gpurun_out/tunableop_gfx950*.csv; wc -l gpurun_out/tunableop_gfx950*.csv
echo "== tuned"; for i in 1 2; do PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])"; done
echo "== untuned again"; python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
