#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_attn_hot; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/$n -o run --output-format csv -- python tools/prof_attn_hotpath.py > $OUT/$n.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_attn_hot/*/run_counter_collection.csv")):
    seq = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            seq.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"][36:80], "g": r["Grid_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d, v in seq.items(): print(d, v)
f = glob.glob("gpurun_out/pmc_attn_hot/FETCH_SIZE/run_kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    if "attn" in r["Kernel_Name"]: print("dur", r["Dispatch_Id"], r["Kernel_Name"][36:80], r["Grid_Size"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
