#!/bin/bash
# rocprofv3 counter passes for the GEMM kernel; writes CSVs under gpurun_out/pmc_<tag>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-pmc}; shift
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o run --output-format csv -- python tools/prof_gemm.py "$@" > $OUT/p$i.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head
python - <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/pmc_*/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg.setdefault(k, []).append(float(r["Counter_Value"]))
    print(f)
    for (kn, cn), v in agg.items():
        if "gemm" in kn: print("  ", kn, cn, [round(x) for x in v][::3])
PY
