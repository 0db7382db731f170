#!/bin/bash
# HBM traffic of the dominant GEMM launches (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes as the guide
# prescribes; FETCH_SIZE is doubled on gfx950 for wide coalesced reads).  Output: gpurun_out/pmc_traffic_<tag>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-t}; OUT=gpurun_out/pmc_traffic_$TAG; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/$n -o run --output-format csv -- python tools/prof_gemm_shapes.py > $OUT/$n.log 2>&1
done
python - <<'PY' | tee gpurun_out/pmc_traffic_$TAG.txt
import csv, glob, collections
res = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_traffic_*/*/run_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "xattn_fused" in r["Kernel_Name"]:
            res.setdefault((r["Kernel_Name"][:70], r["Grid_Size"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k, v in res.items():
    print(k, [round(x) for x in v[:4]])
PY
sha256sum open_flamingo_amd/csrc/libofhip.so | cut -c1-16 > gpurun_out/pmc_traffic_$TAG.sha16
