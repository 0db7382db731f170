#!/bin/bash
# SQ / TA / TD / TCP counters of the big-tile GEMM kernels and the vendor kernel on the same launch (rocprofv3 --pmc, a few
# counters per pass).  Output: gpurun_out/<tag>_pmc_gemm_variants.txt.  Args: tag [extra args of tools/pmc_gemm_variants.py]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-t}; shift; OUT=gpurun_out/pmc_variants_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0; fails=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TD_TD_BUSY_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $c --kernel-trace -d $OUT/p$i -o run --output-format csv -- python tools/pmc_gemm_variants.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i ($c) failed: $(grep -m2 -i "fault\|error" $OUT/p$i.log)"; fails=$((fails+1)); [ $fails -ge 2 ] && break; }
done
python - "$OUT" <<'PY' > gpurun_out/${TAG}_pmc_gemm_variants.txt
import csv, glob, collections, sys
res = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/run_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" in k or "Cijk" in k:
            res.setdefault(r["Counter_Name"], collections.OrderedDict()).setdefault(k[:84], []).append(float(r["Counter_Value"]))
for c, d in res.items():
    print(c)
    for k, v in d.items():
        v = v[1:] if len(v) > 1 else v            # first launch of a kernel: cold
        print("   %-86s n=%d mean %.0f" % (k, len(v), sum(v) / len(v)))
PY
cat gpurun_out/${TAG}_pmc_gemm_variants.txt | head -120
rm -rf $OUT
