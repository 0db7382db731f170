#!/bin/bash
# MFMA utilisation of the dominant GEMM launches and of the fused attention branch at the library in the tree (rocprofv3 --pmc with
# --kernel-trace only, counters in separate passes): SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (1024 x GRBM_GUI_ACTIVE / 8: that counter is
# summed over the 8 XCDs) = the share of SIMD cycles with the matrix pipe busy.  Output: gpurun_out/pmc_mfma_util_<tag>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-t}; OUT=gpurun_out/pmc_mfma_util_$TAG; mkdir -p $OUT
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/$n -o run --output-format csv -- python tools/prof_gemm_shapes.py > $OUT/$n.log 2>&1
done
python - "$TAG" <<'PY' | tee gpurun_out/pmc_mfma_util_$TAG.txt
import csv, glob, collections, sys
tag = sys.argv[1]
res = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/pmc_mfma_util_{tag}/*/run_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "xattn_fused" in r["Kernel_Name"]:
            res.setdefault((r["Kernel_Name"][28:92], r["Grid_Size"]), collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/pmc_mfma_util_{tag}/SQ_VALU*/run_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "xattn_fused" in r["Kernel_Name"]:
            dur.setdefault((r["Kernel_Name"][28:92], str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in res.items():
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    util = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * avg["GRBM_GUI_ACTIVE"] / 8) if avg.get("GRBM_GUI_ACTIVE") else None
    d = dur.get(k, [])
    print(k, "launches", len(next(iter(v.values()))), "us_under_pmc", round(sum(d) / len(d), 1) if d else None,
          "mfma_util", round(util, 3) if util is not None else None, {c: round(x) for c, x in avg.items()})
PY
sha256sum open_flamingo_amd/csrc/libofhip.so | cut -c1-16 | tee gpurun_out/pmc_mfma_util_$TAG.sha16
