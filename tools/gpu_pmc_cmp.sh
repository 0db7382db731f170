#!/bin/bash
# Counter comparison of of_gemm_pp / of_gemm_w4 / hipBLASLt on one shape: kernel-trace stats (names, durations) + PMC
# passes chosen from what `rocprofv3 -L` offers.  Output: gpurun_out/pmc_cmp_<tag>/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-cmp}; shift
OUT=gpurun_out/pmc_cmp_$TAG; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python tools/prof_gemm_cmp.py "$@" > $OUT/trace.log 2>&1
python - "$OUT" "$@" <<'PY'
import re, subprocess, sys, os
out = sys.argv[1]; rest = sys.argv[2:]
avail = set(re.findall(r"\b((?:SQ|GRBM|TCC|TCP|TA|TD)_[A-Za-z0-9_]+|FETCH_SIZE|WRITE_SIZE)\b", open(os.path.join(out, "counters_list.txt")).read()))
wish = [["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"],
        ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
        ["SQ_INSTS_MFMA", "SQ_INSTS_SMEM", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_FLAT"],
        ["TA_BUSY_avr", "TA_TA_BUSY_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TD_TD_BUSY_sum"]]
for i, grp in enumerate(wish):
    grp = [c for c in grp if c in avail]
    if not grp:
        continue
    cmd = ["timeout", "300", "rocprofv3", "--pmc", *grp, "--kernel-trace", "-d", f"{out}/p{i}", "-o", "run", "--output-format", "csv", "--",
           "python", "tools/prof_gemm_cmp.py", *rest]
    with open(f"{out}/p{i}.log", "w") as f:
        subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT)
PY
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
def short(n):
    if "of_gemm_pp" in n: return "pp"
    if "of_gemm_w4" in n: return "w4dma" if n.rstrip().endswith("true>(OfGemmArgs)") else "w4"
    if n.startswith("Cijk") or "Cijk" in n: return "blaslt"
    return None
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if short(r["Name"]): print("STATS", short(r["Name"]), r["Calls"], "avg_ns", r["AverageNs"], r["Name"][:400])
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        s = short(r["Kernel_Name"])
        if s: agg.setdefault(r["Counter_Name"], {}).setdefault(s, []).append(float(r["Counter_Value"]))
        if s: agg.setdefault("_regs", {}).setdefault(s, []).append((r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size")))
for c, d in agg.items():
    if c == "_regs":
        print(c, {k: v[0] for k, v in d.items()})
    else:
        print(c, {k: round(sorted(v)[len(v)//2]) for k, v in d.items()})
PY
