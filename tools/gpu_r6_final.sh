#!/bin/bash
# Round-6 closing measurements in one gpurun call; everything lands in gpurun_out/<tag>_*  (copied to profiles/r06_final_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r06_final}
sha256sum open_flamingo_amd/csrc/libofhip.so > gpurun_out/${TAG}_lib.sha
# 0. the whole -m gpu suite + smoke
( timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -12 ) > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.log
# 1. the bench line the driver will see (+ per-shape GEMM table): roofline, cpu_baseline, reference_eager (both), floor
( python bench.py --steps 20 --warmup 5 --gemm-report gpurun_out/${TAG}_default_gemm_report.jsonl 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_default_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_default_bench.json')); r=d['roofline']; print('default', d['ms_per_step'], d['value'], r['frac'], r['all_gemm_tflops'], r['all_gemm_frac'], r['all_gemm_ms_per_step'], 'traffic', r['traffic'], 'vs_baseline', d['vs_baseline'], d.get('vs_reference_stock_towers'), 'floor', d['floor']['step_frac_of_floor'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
# 2. rocprofv3 kernel stats of the same command (3 warm-up + 3 timed steps: divide totals by 6)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-reference-eager > gpurun_out/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "run_kernel_stats.csv" | head -1) gpurun_out/${TAG}_of3b_bench_kernel_stats.csv
grep "^{" gpurun_out/${TAG}_prof_bench.log | cut -c1-160
# 3. the other BASELINE configurations through --config, + the reference's two-pass step, + the 2-rank rehearsal (with the comm sweep) on this one GPU
for c in 4 5 5L; do
  ( timeout 900 python bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline --no-reference-eager-stock --gemm-report gpurun_out/${TAG}_cfg${c}_gemm_report.jsonl 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_cfg${c}_bench.json
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_cfg${c}_bench.json')); print('cfg$c', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'], 'vs_baseline', d.get('vs_baseline'), 'floor', d['floor']['step_frac_of_floor'])"
done
( timeout 600 python bench.py --laion-batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-reference-eager 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_two_pass_bench.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_two_pass_bench.json')); print('two-pass', d['ms_per_step'], d['value'])"
( timeout 900 python bench.py --gpus 2 --one-gpu-loopback --steps 4 --warmup 2 --no-cpu-baseline --no-reference-eager --sweep-comm --sweep-steps 2 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_bench_gpus2_loopback_comm_sweep.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_gpus2_loopback_comm_sweep.json')); print('2 ranks, one GPU, loopback (a rehearsal, not a throughput):', d['n_gpus'], d['overlap']['collectives_per_step'], len(d['comm_sweep']['settings']), 'sweep settings')"
# 4. the fused attention branch: phases, against the five launches; the step-level same-box A/B
( timeout 400 python tools/probes/xattn_fused_probe.py 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_xattn_fused_probe.jsonl
cut -c1-420 gpurun_out/${TAG}_xattn_fused_probe.jsonl
( timeout 700 python tools/ab_fused_xattn.py --steps 12 --warmup 4 2>/dev/null ) > gpurun_out/${TAG}_ab_fused_xattn_step.jsonl
cat gpurun_out/${TAG}_ab_fused_xattn_step.jsonl
( timeout 200 ./tools/probes/frag_stream_probe ) > gpurun_out/${TAG}_frag_stream_probe.jsonl 2>/dev/null
# 5. a launch behind a streaming pass (cold operands), kernel microbench (warm AND cold), vendor plain vs ours
( timeout 400 python tools/probes/interleaved_gemm_probe.py 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_interleaved_gemm_probe.jsonl
cut -c1-330 gpurun_out/${TAG}_interleaved_gemm_probe.jsonl | head -9
( timeout 500 python tools/bench_kernels.py 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_kernel_microbench.jsonl
( timeout 300 python tools/probes/vendor_plain_vs_ours.py 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_vendor_plain_vs_ours.jsonl
cat gpurun_out/${TAG}_vendor_plain_vs_ours.jsonl | cut -c1-260
# 6. HBM traffic of the dominant GEMM launches and of the fused attention branch (separate --pmc passes, --kernel-trace only)
bash tools/gpu_pmc_traffic.sh $TAG > gpurun_out/${TAG}_gemm_hbm_traffic_pmc.log 2>&1
cat gpurun_out/pmc_traffic_$TAG.txt gpurun_out/pmc_traffic_$TAG.sha16
# 7. compact heads (OF-4B's head size 80 on the 128-wide attention kernels): per-layer probe + the step-level same-box A/B against the padded copies;
#    the CLIP tower's fc2 row split and the attention backward's split of a ragged head count (config 5), same box
( timeout 300 python tools/probes/compact_heads_probe.py 2>/dev/null | grep "^{" ) > gpurun_out/${TAG}_compact_heads_probe.jsonl
cut -c1-420 gpurun_out/${TAG}_compact_heads_probe.jsonl
( timeout 900 python tools/ab_neox_compact_heads.py --steps 8 --warmup 3 2>/dev/null | cut -c1-330 ) > gpurun_out/${TAG}_ab_neox_compact_heads_step.jsonl
cut -c1-260 gpurun_out/${TAG}_ab_neox_compact_heads_step.jsonl
( timeout 900 python tools/ab_vit_fc2_split.py --steps 12 --warmup 4 2>/dev/null | cut -c1-330 ) > gpurun_out/${TAG}_ab_vit_fc2_split_step.jsonl
cut -c1-260 gpurun_out/${TAG}_ab_vit_fc2_split_step.jsonl
# 8. the step schedule: where the next step's tower forward is enqueued (product rule against behind-the-backward), narrow against wide step epilogue
( AB_ARMS=1,0 timeout 600 python tools/ab_prefetch_point.py --steps 12 --warmup 4 2>/dev/null | cut -c1-330 ) > gpurun_out/${TAG}_ab_prefetch_point_step.jsonl
cut -c1-200 gpurun_out/${TAG}_ab_prefetch_point_step.jsonl
for c in 192 0 192 0; do echo -n "optimizer-cus $c "; python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-reference-eager --optimizer-cus $c 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss_last_step'])"; done | tee gpurun_out/${TAG}_ab_narrow_step_epilogue.txt
