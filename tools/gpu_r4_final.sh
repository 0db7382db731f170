#!/bin/bash
# Round-4 closing measurements in one gpurun call; everything lands in gpurun_out/<tag>_*  (copied to profiles/r04_final_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r04z}
sha256sum open_flamingo_amd/csrc/libofhip.so > gpurun_out/${TAG}_lib.sha
# 0. the whole -m gpu suite + smoke
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -12 ) > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
python __graft_entry__.py smoke 2>&1 | tail -1
# 1. the bench line the driver will see (+ per-shape GEMM table): roofline, cpu_baseline, reference_eager (both), floor
( python bench.py --steps 20 --warmup 5 --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['all_gemm_tflops'], d['roofline']['all_gemm_ms_per_step'], 'vs_baseline', d['vs_baseline'], d.get('vs_reference_stock_towers'), 'floor', d['floor']['step_frac_of_floor'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
# 2. rocprofv3 kernel stats of the same command (3 warm-up + 3 timed steps: divide totals by 6)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-reference-eager > gpurun_out/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of3b_bench_kernel_stats.csv
grep "^{" gpurun_out/${TAG}_prof_bench.log | cut -c1-160
# 3. the other BASELINE configurations through --config, + the reference's two-pass step
for c in 4 5 5L; do
  ( timeout 900 python bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline --no-reference-eager-stock --gemm-report gpurun_out/${TAG}_cfg${c}_gemm_report.jsonl 2>&1 | grep "^{" ) > gpurun_out/${TAG}_cfg${c}_bench.json
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_cfg${c}_bench.json')); print('cfg$c', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'], 'vs_baseline', d.get('vs_baseline'), 'floor', d['floor']['step_frac_of_floor'])"
done
( timeout 600 python bench.py --laion-batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-reference-eager 2>&1 | grep "^{" ) > gpurun_out/${TAG}_two_pass_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_two_pass_bench.json')); print('two-pass', d['ms_per_step'], d['value'])"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof5_$TAG -o run --output-format csv -- python bench.py --config 5 --steps 3 --warmup 2 --no-cpu-baseline --no-reference-eager > gpurun_out/${TAG}_prof_cfg5.log 2>&1
cp $(find /tmp/prof5_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_cfg5_bench_kernel_stats.csv
# 4. same-box A/Bs: vision prefetch off; GEMM launches vs round 3's closing library (three families)
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-eager --no-roofline --no-vision-prefetch 2>&1 | grep "^{" ) > gpurun_out/${TAG}_no_prefetch_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_no_prefetch_bench.json')); print('no prefetch', d['ms_per_step'])"
for fam in OF-3B OF-4B OF-9B; do ( timeout 400 python tools/bench_gemm_ab.py tools/ab/libofhip_r03.so --family $fam --arms old,new 2>&1 | grep "^{" ) > gpurun_out/${TAG}_gemm_ab_$fam.jsonl; done
python - <<PY
import json
for fam in ("OF-3B", "OF-4B", "OF-9B"):
    for l in open("gpurun_out/${TAG}_gemm_ab_%s.jsonl" % fam):
        r = json.loads(l)
        print(f"  {fam} {r['name']:22s} {r['layout']} {str(r['MNK']):22s} old {r['old_ms']*1e3:.1f}/{r['old_tflops']:.0f}  new {r['new_ms']*1e3:.1f}/{r['new_tflops']:.0f}")
PY
# 5. tile phase probe, kernel microbench
( timeout 300 python tools/probes/tile_phase_probe.py 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_tile_phase_probe.jsonl
( timeout 400 python tools/bench_kernels.py 2>&1 | grep "^{" ) > gpurun_out/${TAG}_kernel_microbench.jsonl
# 6. HBM traffic of the dominant GEMM launches (separate --pmc passes, --kernel-trace only)
bash tools/gpu_pmc_traffic.sh $TAG > gpurun_out/${TAG}_gemm_hbm_traffic_pmc.txt 2>&1
tail -12 gpurun_out/${TAG}_gemm_hbm_traffic_pmc.txt
# 7. N1 evidence: this library's plain GEMM vs the vendor library like for like, and the frozen MLP routed through the fused
#    epilogues, step-level same-box A/B (frozen_blocks._MLP_FUSED_UP / _DOWN)
( timeout 300 python tools/probes/vendor_plain_vs_ours.py 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_vendor_plain_vs_ours.jsonl
cat gpurun_out/${TAG}_vendor_plain_vs_ours.jsonl
( timeout 900 python tools/ab_frozen_mlp.py --steps 8 --warmup 3 --no-reference-eager --no-roofline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    arm, _, rest = l.partition(' {')
    try:
        d = json.loads('{' + rest)
        print(arm, d['ms_per_step'])
    except Exception:
        print(l[:200])
" ) > gpurun_out/${TAG}_ab_frozen_mlp.txt
cat gpurun_out/${TAG}_ab_frozen_mlp.txt
