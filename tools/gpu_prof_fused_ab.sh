#!/bin/bash
# rocprofv3 kernel stats of bench.py with the fused attention branch and with the separate launches (tools/ab_fused_xattn.py arms)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r06j}
for arm in 1 0; do
  rm -rf /tmp/prof_$TAG_$arm
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$arm -o run --output-format csv -- python tools/ab_fused_xattn.py --arm $arm --steps 3 --warmup 3 --no-cpu-baseline --no-reference-eager > gpurun_out/${TAG}_prof_arm$arm.log 2>&1
  cp $(find /tmp/prof_${TAG}_$arm -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_arm${arm}_kernel_stats.csv
  grep "^{" gpurun_out/${TAG}_prof_arm$arm.log | cut -c1-200
done
python - <<PY
import csv
for arm in (1, 0):
    rows = list(csv.DictReader(open("gpurun_out/${TAG}_arm%d_kernel_stats.csv" % arm)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("arm", arm, "total kernel ms per step", round(tot / 6e6, 2))
    for r in rows:
        n = r["Name"]
        if any(k in n for k in ("xattn_fused", "of_ln_fwd", "of_attn_q_kernel", "pack_frag", "of_gemm_mid_kernel<false, false, 0>", "GATE_RESID", "w4m_kernel<false, false, 2")) or float(r["TotalDurationNs"]) / tot > 0.02:
            print("   %-110s calls %5s avg_us %8.1f ms/step %7.2f" % (n[:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 6e6))
PY
