#!/bin/bash
# Round-2 closing measurements in one gpurun call:
#  1. reference-equivalent eager step on this box (with the reference's embedding-row mask), stock and bench-like towers
#  2. HBM traffic (FETCH_SIZE, WRITE_SIZE, L2 hit rate; separate --pmc passes) of the dominant GEMM launches
#  3. kernel microbench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r02f}
( timeout 600 python tests/perf_reference_eager.py OF-3B 32 2 256 2>&1 | grep "^{" ; timeout 600 python tests/perf_reference_eager.py OF-3B 32 2 256 --stock-towers 2>&1 | grep "^{" ) > gpurun_out/${TAG}_reference_eager.jsonl
cat gpurun_out/${TAG}_reference_eager.jsonl | cut -c1-400
OUT=gpurun_out/pmc_traffic_$TAG; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/$n -o run --output-format csv -- python tools/prof_gemm_shapes.py > $OUT/$n.log 2>&1
done
python - "$OUT" <<'PY' | tee gpurun_out/${TAG}_pmc_traffic.txt
import csv, glob, collections, sys
res = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/*/run_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "of_gemm" in r["Kernel_Name"]:
            res.setdefault((r["Kernel_Name"][:90], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k, v in res.items():
    print(k[0], "|", k[1], [round(x) for x in v])
PY
( timeout 300 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${TAG}_kernels.jsonl
cat gpurun_out/${TAG}_kernels.jsonl | cut -c1-250
