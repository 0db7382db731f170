#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) over tools/prof_attn_self.py -> gpurun_out/pmc_attn_<tag>/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-a}; SCRIPT=${2:-tools/prof_attn_self.py}
OUT=gpurun_out/pmc_attn_$TAG; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa_trace -o run --output-format csv -- python $SCRIPT > $OUT/trace.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pa_p$i -o run --output-format csv -- python $SCRIPT > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections
for f in glob.glob("/tmp/pa_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print("STATS", r["Calls"], "avg_ns", r["AverageNs"], r["Name"][:90])
agg = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/pa_p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            k = r["Kernel_Name"][:70] + " grid " + r.get("Grid_Size", "?") + " lds " + r.get("LDS_Block_Size", "?") + " vgpr " + r.get("VGPR_Count", "?")
            agg.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    print("   ", {c: round(sorted(v)[len(v) // 2]) for c, v in d.items()})
PY
