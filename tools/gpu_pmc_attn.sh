#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_attn; mkdir -p $OUT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o run --output-format csv -- python tools/prof_attn.py > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_attn/p*/run_counter_collection.csv")):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        agg.setdefault((r["Kernel_Name"][:48], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (kn, cn), v in agg.items():
        if "attn" in kn: print(kn, cn, round(v[-1]))
for f in glob.glob("gpurun_out/pmc_attn/p1/run_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]: print(r["Kernel_Name"][:48], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us", r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
PY
