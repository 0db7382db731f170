"""bench.py -- OpenFlamingo training-step throughput on MI355X (BASELINE.json metric: train images/sec + step ms,
OF-3B = ViT-L/14 + MPT-1B, xattn every layer, amp_bf16, synthetic MMC4-style batch B=32 T=2 F=1 L=256 per GPU).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full optimizer step on one synthetic batch: frozen ViT forward (no_grad), PerceiverResampler
fwd+bwd (libofhip), 24 x [GatedCrossAttentionBlock (libofhip) + frozen MPT block] fwd + bwd, LM head + loss, masked
embedding gradient, gradient exchange (RCCL, overlapped), global-norm clip, AdamW.  Inputs are resident in HBM before
the timed region.  Weights are random-init of the named architecture (no network), data synthetic.

Prints ONE JSON line (rank 0) with the driver's contract plus:
  overlap       -- gradient exchange of the step: bytes all-reduced, wire dtype, collectives per step and the time the
                   compute stream actually waited for RCCL in GradReducer.finish() (HIP events; 0 at one GPU).
  roofline      -- the dominant libofhip kernel family (the bf16 MFMA GEMM) measured live with HIP events on the
                   compute stream inside the timed region: algorithmic FLOPs / summed launch time vs 2.5 PFLOP/s.
  cpu_baseline  -- the oracle (CPU port of the reference arithmetic) timed on the host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

EPI_NAMES = {0: "store_bf16", 1: "gelu", 2: "gate_resid", 3: "dgelu_dot", 4: "scale_dot", 5: "acc_f32"}
LAYOUT_NAMES = {(0, 0): "NT(y=xW^T)", (0, 1): "NN(dX=dY W)", (1, 1): "TN(dW=dY^T X)"}
MFMA_PEAK_TFLOPS = 2500.0   # bf16 dense, /opt/skills/guides/MI355X_MICROARCH.md


def _oracle_hot_path(d, nblk, seed=0):
    from oracle import flamingo_oracle as O
    torch.manual_seed(seed)
    per = O.OraclePerceiverResampler(dim=1024)
    blocks = [O.OracleGatedCrossAttentionBlock(dim=d, dim_visual=1024) for _ in range(nblk)]
    for b in blocks:
        with torch.no_grad():
            b.attn_gate.fill_(0.5)
            b.ff_gate.fill_(0.5)
    return per, blocks


def _oracle_time(per, blocks, B, T, L, d, reps, warm, backward):
    """median seconds of `reps` runs after `warm` warm-ups: (forward, forward + backward) of Perceiver + the given blocks"""
    import statistics
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(B, T, 1, 256, 1024, generator=g)
    x0 = torch.randn(B, L, d, generator=g)
    ml = torch.zeros(B, L, dtype=torch.bool)
    for t in range(T):
        ml[:, t * (L // T)] = True
    fwd, both = [], []
    for i in range(warm + reps):
        t0 = time.perf_counter()
        with torch.set_grad_enabled(backward):
            y = x0.clone().requires_grad_(backward)
            vis = per(feats)
            for b in blocks:
                y = b(y, vis, media_locations=ml)
        t1 = time.perf_counter()
        if backward:
            y.square().mean().backward()
            for m in [per] + blocks:
                m.zero_grad(set_to_none=True)
        t2 = time.perf_counter()
        if i >= warm:
            fwd.append(t1 - t0)
            both.append(t2 - t0)
    return statistics.median(fwd), statistics.median(both)


def cpu_baseline(family_info, threads, T=2, L=256):
    """The reference's CPU path for the hot path, timed on this box's host cores.  What runs is the oracle
    (oracle/flamingo_oracle.py: the reference's helpers.py arithmetic restated line by line in plain PyTorch fp32 and PINNED
    to the real reference by tests/golden -- /root/reference does not exist on the GPU box), kind = "port".
      * primary figure = BASELINE.json configs[0]: B = 1, T = 2 images, L = 32 text tokens, fp32, the WHOLE hot path
        (PerceiverResampler + every gated block of the family, no extrapolation): forward, and forward + backward; median of 5
        after 2 warm-ups;
      * second figure = the benchmark's own per-sequence shapes (T images, L tokens) for B = 2, forward + backward, median of 3
        after 1 warm-up.
    The frozen towers are not part of the path and not part of the sample."""
    torch.set_num_threads(threads)
    d, nblk = family_info["d"], family_info["layers"] // family_info["every"]
    per, blocks = _oracle_hot_path(d, nblk)
    f1, fb1 = _oracle_time(per, blocks, 1, 2, 32, d, reps=5, warm=2, backward=True)
    f1_ng, _ = _oracle_time(per, blocks, 1, 2, 32, d, reps=5, warm=2, backward=False)
    _, fb2 = _oracle_time(per, blocks, 2, T, L, d, reps=3, warm=1, backward=True)
    return {"value": round(2 / fb1, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "pinning": "oracle = line-by-line restatement of reference helpers.py, pinned to the real reference by tests/golden "
                       "(tests/golden/make_golden.py ran /root/reference; tests/test_oracle_golden.py replays it)",
            "sample": f"BASELINE config 1: B=1 T=2 L=32 fp32, whole hot path (PerceiverResampler + all {nblk} gated xattn blocks at "
                      f"d={d}), forward + backward, median of 5 after 2 warm-ups: {fb1 * 1e3:.0f} ms per 2 images "
                      f"(forward {f1 * 1e3:.0f} ms; forward under no_grad {f1_ng * 1e3:.0f} ms)",
            "config1_forward_ms": round(f1_ng * 1e3, 1), "config1_forward_backward_ms": round(fb1 * 1e3, 1),
            "bench_shape_sample": {"value": round(2 * T / fb2, 3), "unit": "images/s",
                                   "sample": f"B=2 T={T} L={L} (the benchmark's per-sequence shapes), whole hot path forward + "
                                             f"backward, median of 3 after 1 warm-up: {fb2 * 1e3:.0f} ms per {2 * T} images"}}


def step_flops(info, B, T, L, vocab_extra=3):
    """Algorithmic FLOPs (2 x MAC) of one train step per GPU, SURVEY.md 8d closed forms: hot path forward x 3 (every hot-path
    parameter is trained), frozen LM forward x 2 (dX-only backward), ViT forward x 1 (no_grad).  The cross-attention core counts
    the 64-key window a text token can see (restricted form), never the masked-out keys."""
    d, layers, every, vocab = info["d"], info["layers"], info["every"], info.get("vocab", 50432) + vocab_extra
    n_img = B * T
    perceiver = n_img * 6 * (2 * 64 * 1024 * 512 + 2 * 320 * 1024 * 1024 + 2 * 2 * 8 * 64 * 320 * 64 + 2 * 64 * 512 * 1024 + 4 * 64 * 1024 * 4096)
    xattn_seq = L * (2 * d * 512 + 2 * 512 * d + 16 * d * d) + 2 * (T * 64) * 1024 * 1024 + 2 * 2 * 8 * L * 64 * 64
    xattn = B * (layers // every) * xattn_seq
    lm = B * L * (layers * (24 * d * d + 4 * L * d) + 2 * d * vocab)
    vit = n_img * 2 * 81.0e9                      # ~81 GMAC per 224-px image for ViT-L/14 (third-party figure)
    return {"hot_path": 3.0 * (perceiver + xattn), "frozen_lm": 2.0 * lm, "vit": vit,
            "total": 3.0 * (perceiver + xattn) + 2.0 * lm + vit}


def reference_eager_step(family, B, T, L, steps=3, warm=2, stock=False):
    """The reference-equivalent EAGER train step on this GPU (what the north_star's ">= 5x reference single-GPU step time" refers
    to): the same Flamingo + frozen towers with the hot-path modules replaced by the oracle's nn.Modules -- the reference's
    helpers.py arithmetic restated line by line, pinned to it by tests/golden (/root/reference does not exist on the GPU box) --
    run as the reference runs them: eager ATen ops under torch.autocast(bfloat16), the dense embedding-gradient row mask of
    train_utils.py:174-196, clip_grad_norm_ + torch AdamW.  A reported BASELINE leg like cpu_baseline, never the product.
    stock = True: the frozen towers exactly as stock modules run them (MIOpen conv patch embedding, fp32 frozen weights re-cast by
    autocast every forward, HF's eager MPT attention) -- what a user of the reference gets on this box; stock = False: the same
    tower-side choices bench.py makes (GEMM patch embedding, bf16-held frozen weights, fused LM attention), so that the
    difference to the product line is the hot path + step epilogue + block fusion only."""
    from open_flamingo_amd.train import step, synthetic, towers
    from tests.cpu_model import swap_in_oracle
    model, info = towers.build_flamingo(family, device="cuda", seed=0, gates=0.5, frozen_bf16=not stock, fused_lm_attention=not stock,
                                        vision_kw=dict(patch_embed="conv") if stock else None)
    swap_in_oracle(model)
    model.cuda().train()
    opt = step.build_optimizer(model)
    batch = synthetic.make_batch(B, T, L, info, "cuda", seed=1)
    losses = []
    for _ in range(warm):
        losses.append(step.train_step(model, None, opt, batch, info))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(step.train_step(model, None, opt, batch, info))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": round(ms, 2), "images_per_s": round(B * T / ms * 1e3, 2), "steps": steps, "warmup": warm,
            "kind": "port", "towers": "stock" if stock else "as bench.py", "losses": [round(float(l), 4) for l in losses],
            "what": "oracle hot-path modules (reference helpers.py restated, pinned by tests/golden) inside the same Flamingo + the "
                    "same frozen towers, eager ATen ops under autocast(bf16), reference's dense embedding-row mask, clip_grad_norm_ + "
                    "torch.optim.AdamW; same box, same process, after the timed region"}


CONFIGS = {   # BASELINE.json configs that fit one GPU: family, per-GPU batch, images per sequence, text length
    "2": ("OF-3B", 32, 2, 256, "BASELINE config 2 (= config 3 per GPU)"),
    "4": ("OF-4B", 32, 2, 256, "BASELINE config 4, per-GPU share"),
    "5": ("OF-9B", 8, 5, 256, "BASELINE config 5, per-GPU share, L = 256 (the reference's max text length)"),
    "5L": ("OF-9B", 8, 5, 2048, "BASELINE config 5, per-GPU share, L = 2048 (MPT-7B context: the long-context reading)"),
}


def _lib_sha16():
    """which build of the HIP library this line was measured on (the .so is built in-tree and shipped, not committed)"""
    import hashlib
    path = os.path.join(ROOT, "open_flamingo_amd", "csrc", "libofhip.so")
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(key, shape):
    """HBM-side bytes per launch of the dominant kernel.  PMC counters cannot be collected inside a timed run (rocprofv3
    --pmc serialises kernels and needs separate passes per counter group), so the live line quotes the committed PMC
    passes of the SAME kernel, epilogue and launch shape (profiles/pmc_traffic.json: keys "ta tb epi kernel M N K") and says
    so; null when there is no pass on record for exactly that launch."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    name = " ".join(str(k) for k in (int(key[0]), int(key[1]), int(key[2]), key[3], *shape))
    try:
        with open(path) as f:
            table = json.load(f)
        ent = table["kernels"][name]
    except (OSError, KeyError, ValueError):
        return None, f"no PMC pass on record for this launch ({name}; profiles/pmc_traffic.json)"
    if table.get("libofhip_sha16") != _lib_sha16():       # the record is of another build of the library: not this line's bytes
        return None, (f"the PMC passes on record ran libofhip {table.get('libofhip_sha16')}, this line {_lib_sha16()} "
                      f"(profiles/pmc_traffic.json; regenerate: tools/gpu_pmc_traffic.sh + tools/make_pmc_traffic_json.py)")
    return ent["fetch_bytes"] + ent["write_bytes"], (
        f"bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE; fetches include "
        f"Infinity-Cache hits), not collected in this run: {ent['launch']}; algorithmic bytes {ent['algorithmic_bytes']}; "
        f"L2 hit rate {ent['l2_hit_rate']}; profiles/pmc_traffic.json")


def _spawn_ranks(n):
    """``python bench.py --gpus N`` with no launcher environment: re-run this command line under
    ``torch.distributed.run`` (one process per GPU, RCCL rendezvous on 127.0.0.1) and pass its output through -- rank 0
    prints the one JSON line.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# The frozen towers' product forms: this repository's attention / LayerNorm / loss kernels inside whole-block autograd nodes.
# The stock alternatives (torch SDPA, HF modules, eager LayerNorm / loss) are what the reference-equivalent eager leg below is built from;
# timing them against the product forms is a profiling job: tools/ab_tower_arms.py sets this dict and calls main().
TOWER_ARMS = dict(lm_attention="libofhip", lm_blocks="fused", vision="libofhip", tower_layernorm="libofhip", lm_loss="libofhip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS),
                    help="which BASELINE.json configuration to time (per-GPU shapes, SURVEY.md 8d): 2 = OF-3B B=32 T=2 L=256 (the one the "
                         "metric is quoted on; default), 4 = OF-4B, 5 = OF-9B B=8 T=5 L=256, 5L = the same at L=2048; --family / "
                         "--batch / --T / --L override single fields")
    ap.add_argument("--family", default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--L", type=int, default=None)
    ap.add_argument("--no-reference-eager", action="store_true",
                    help="skip the reference-equivalent eager step timed after the run (1 GPU only; fills reference_eager / vs_baseline)")
    ap.add_argument("--no-reference-eager-stock", action="store_true", help="skip the second, stock-tower reference-eager run")
    ap.add_argument("--check-right-padding", action="store_true",
                    help="debug: verify on the host (one sync per LM forward) that every attention mask the fused frozen blocks "
                         "reduce to key counts is a prefix of ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--wire-bf16", action="store_true", help="all-reduce gradients in bf16 on the wire")
    ap.add_argument("--frozen-fp32", action="store_true",
                    help="keep the frozen towers' Linear weights in fp32 (autocast re-casts them every forward, as the "
                         "reference does) instead of holding their bf16 copies")
    ap.add_argument("--torch-optimizer", action="store_true",
                    help="clip_grad_norm_ + torch.optim.AdamW(fused) instead of the libofhip step epilogue")
    ap.add_argument("--dense-embedding-rows", action="store_true",
                    help="form the dense (vocab x d) embedding gradient and mask it down to the two trained rows, as the "
                         "reference does (train_utils.py:174-196); default: train/sparse_rows.py forms just those two rows "
                         "(identical values, no dense lookup scatter / tied-head weight-gradient GEMM)")
    ap.add_argument("--vendor-gemm-table", default="tuned", choices=["tuned", "default"],
                    help="kernel selection of the frozen towers' vendor-library GEMMs: the committed TunableOp table "
                         "(open_flamingo_amd/train/tuned/, tuning off at run time) or the libraries' default heuristics")
    ap.add_argument("--nan-check", default="device", choices=["device", "host"],
                    help="the reference's skip-the-step-on-NaN-loss (train_utils.py:161-169): decided on the device by the fused step "
                         "epilogue (non-finite global gradient norm: no update, on every rank alike, no host sync), or with the "
                         "reference's host-side torch.isnan(loss)")
    ap.add_argument("--laion-batch", type=int, default=0,
                    help="also run the reference's LAION pass in every step (train_utils.py:94-118: B image-caption pairs, T = 1, "
                         "32 text tokens, loss multiplier 0.2, its backward under no_sync) before the MMC4-style pass -- the two-pass "
                         "step of the reference's training script (run_train.sh: batch_size_laion = 2 x batch_size_mmc4); BASELINE.json "
                         "names only the MMC4-style batch, so the default (0) is the primary number")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="multi-GPU: CUs to leave to RCCL's kernels while gradient collectives are in flight -- the libofhip GEMMs of the "
                         "backward are then laid out stream-K for 256 - R workgroups (GradReducer.reserve_cus; default 0 = off)")
    ap.add_argument("--no-batched-dw", action="store_true",
                    help="A/B: the three 512-wide weight gradients of a gated block as separate split-K launches instead of one batched launch")
    ap.add_argument("--optimizer-cus", type=int, default=None,
                    help="run the step epilogue's streaming passes (global norm, AdamW) as narrow launches on this many CUs (fat "
                         "workgroups, one per CU; identical results) so that the prefetched vision-tower forward on the side stream "
                         "finds whole CUs free; 0 = launches that cover the chip; default: FlatAdamW's own (192)")
    ap.add_argument("--early-norm", action="store_true",
                    help="A/B: compute every bucket's share of the global gradient norm on the reducer's side stream as soon as the bucket's "
                         "gradient is final (behind its all-reduce) instead of in the step epilogue (train/optim.py: early_norm; same bits; "
                         "measured +0.5 ms per step on one GPU, hence off)")
    ap.add_argument("--no-norm-taps", action="store_true",
                    help="A/B (one GPU): take the FFN weight gradients' share of the global gradient norm in the step epilogue's pass "
                         "over the gradients instead of in their dW GEMMs' epilogues (train/optim.py: tap_norm; same norm up to the "
                         "order of the partial sums)")
    ap.add_argument("--no-vision-prefetch", action="store_true",
                    help="run the frozen vision tower at the start of each step (as the reference does) instead of enqueuing the NEXT "
                         "step's tower forward on a side stream next to the step epilogue (train/step.py: next_vision_x)")
    ap.add_argument("--sweep-comm", action="store_true",
                    help="multi-GPU diagnostics in the SAME invocation, after the timed region (the headline numbers are untouched): a few "
                         "extra steps for every combination of reserve_cus in {0, 8, 16, 32} and gradient wire dtype in {fp32, bf16}, each "
                         "with its step time (max over ranks), the exposed wait for RCCL, and the latency of every bucket's all-reduce "
                         "(HIP events on the side stream) -> 'comm_sweep' on the line: the tuning table of the first run on a real node")
    ap.add_argument("--sweep-steps", type=int, default=4, help="timed steps per --sweep-comm combination (after one untimed step)")
    ap.add_argument("--gemm-report", default=None, help="write a per-(layout,epilogue,shape) GEMM time table (JSON lines)")
    ap.add_argument("--one-gpu-loopback", action="store_true",
                    help="REHEARSAL of the multi-rank path on a one-GPU box: every rank of --gpus N uses device 0 and the ranks talk "
                         "through RCCL's socket transport over `lo` (each rank claims its own NCCL_HOSTID: RCCL refuses two ranks of one "
                         "host on one device).  Rank spawning, the NCCL-backend process group, the side-stream all-reduces, finish(), "
                         "max-over-ranks timing and the overlap record are the multi-GPU ones; the wire is a socket and N replicas share "
                         "one chip, so the THROUGHPUT IS MEANINGLESS (config.one_gpu_loopback says so on the line)")
    args = ap.parse_args()
    # how the frozen towers run is not a benchmark option: the product forms (TOWER_ARMS above; tools/ab_tower_arms.py times the others)
    for k, v in TOWER_ARMS.items():
        setattr(args, k, v)
    cfg_family, cfg_batch, cfg_T, cfg_L, cfg_name = CONFIGS[args.config]
    overridden = any(v is not None for v in (args.family, args.batch, args.T, args.L))
    args.family = args.family or cfg_family
    args.batch, args.T, args.L = args.batch or cfg_batch, args.T or cfg_T, args.L or cfg_L
    if overridden:
        cfg_name = f"custom shapes (started from config {args.config})"
    if args.gpus > 1 and not any(k in os.environ for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")):
        sys.exit(_spawn_ranks(args.gpus))

    from open_flamingo_amd.hip.ops import Ops
    from open_flamingo_amd.train import distributed, step, synthetic, towers
    from open_flamingo_amd.train.reducer import GradReducer

    if args.one_gpu_loopback and args.gpus > 1:
        r = os.environ.get("RANK", "0")
        os.environ.update(NCCL_HOSTID=f"of-one-gpu-rank{r}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET_GDR_LEVEL="0",
                          NCCL_SHM_DISABLE="1", NCCL_P2P_DISABLE="1", LOCAL_RANK="0")
    # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there.  Until the
    # line is printed, file descriptor 1 points at stderr (C-level writes included).
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    device = distributed.init_distributed_device()
    assert device.type == "cuda", "bench.py needs an AMD GPU"
    local_rank, rank, world = distributed.world_info_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    model, info = towers.build_flamingo(args.family, device=device, seed=0, gates=0.5, frozen_bf16=not args.frozen_fp32,
                                        fused_lm_attention=args.lm_attention if args.lm_attention != "eager" else False,
                                        tower_layernorm=args.tower_layernorm, lm_loss=args.lm_loss,
                                        fused_lm_blocks=args.lm_blocks == "fused" and not args.frozen_fp32,
                                        fused_vision=False if (args.vision == "modules" or args.frozen_fp32) else args.vision)
    model.train()
    if args.check_right_padding:
        from open_flamingo_amd.train import frozen_blocks
        frozen_blocks.CHECK_RIGHT_PADDING = True
    n_tuned = towers.use_tuned_vendor_gemms() if args.vendor_gemm_table == "tuned" else 0
    args.sparse_embedding_rows = not args.dense_embedding_rows and not args.torch_optimizer
    if args.sparse_embedding_rows:
        from open_flamingo_amd.train import sparse_rows
        sparse_rows.enable(model, [info["media_token_id"], info["eoc_token_id"]])
    reducer = GradReducer(model, wire_dtype=torch.bfloat16 if args.wire_bf16 else torch.float32,
                          embedding_rows=[info["media_token_id"], info["eoc_token_id"]], reserve_cus=args.reserve_cus)
    reducer.broadcast_parameters()
    opt = step.build_optimizer(model, reducer=None if args.torch_optimizer else reducer)
    if args.optimizer_cus is not None and not args.torch_optimizer:
        opt.narrow_cus = args.optimizer_cus
    if args.early_norm and not args.torch_optimizer:
        opt.early_norm = True
    if args.no_norm_taps and not args.torch_optimizer:
        opt.tap_norm = False
    batch = synthetic.make_batch(args.batch, args.T, args.L, info, device, seed=1 + rank)
    laion = synthetic.make_batch(args.laion_batch, 1, 32, info, device, seed=101 + rank) if args.laion_batch > 0 else None
    step_kw = dict(batch_laion=laion, loss_multiplier_laion=0.2) if laion is not None else {}
    if not args.no_vision_prefetch:       # the next step's first forward sees this tensor (a loader is one batch ahead anyway)
        step_kw["next_vision_x"] = (laion if laion is not None else batch)["vision_x"]
    ops = Ops.default()
    ops.batch_dw = not args.no_batched_dw
    nan_check = "device" if (args.nan_check == "device" and not args.torch_optimizer) else True

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Roofline instrumentation: HIP events around EVERY of_gemm launch cost ~3 ms/step (1540 event records), so the last
    # warm-up step times all of them (per-shape report, all-GEMM summary, which (layout, epilogue) family dominates) and
    # the timed region only brackets the dominant family's launches.
    survey = None
    for w in range(args.warmup):
        if not args.no_roofline and w == args.warmup - 1:
            ops.gemm_timing = []
        loss = step.train_step(model, reducer, opt, batch, info, nan_check=nan_check, **step_kw)
    sync()
    if not args.no_roofline and args.warmup > 0:
        survey, ops.gemm_timing = ops.gemm_timing, None
    dominant = None
    if survey:
        acc = {}
        for key, flops, shape, e0, e1 in survey:
            acc[key] = acc.get(key, 0.0) + e0.elapsed_time(e1)
        dominant = max(acc, key=acc.get)
    if not args.no_roofline:
        ops.gemm_timing = []
        ops.gemm_timing_only = {dominant} if dominant is not None else None
    reducer.overlap_stats(reset=True)
    reducer.time_waits = world > 1          # two HIP events per step around the compute stream's wait for RCCL
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step.train_step(model, reducer, opt, batch, info, nan_check=nan_check, **step_kw)
    sync()
    elapsed = time.perf_counter() - t0
    overlap = reducer.overlap_stats()
    timing, ops.gemm_timing = ops.gemm_timing, None
    ops.gemm_timing_only = None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    images = (args.batch * args.T + args.laion_batch) * world
    value = images * args.steps / elapsed

    comm_sweep = None
    if args.sweep_comm and world > 1:
        # Everything the first run on a real node has to decide, measured in ONE call: how many CUs to leave to RCCL's kernels, and
        # whether halving the bytes on the wire pays.  The settings are plain attributes of the reducer; the model simply trains on.
        comm_sweep = []
        base_wire, base_res = reducer.wire_dtype, reducer.reserve_cus
        for wire in (torch.float32, torch.bfloat16):
            for res in (0, 8, 16, 32):
                reducer.wire_dtype, reducer.reserve_cus = wire, res
                step.train_step(model, reducer, opt, batch, info, nan_check=nan_check, **step_kw)
                sync()
                reducer.overlap_stats(reset=True)
                reducer.time_waits = reducer.time_collectives = True
                t1 = time.perf_counter()
                for _ in range(args.sweep_steps):
                    step.train_step(model, reducer, opt, batch, info, nan_check=nan_check, **step_kw)
                sync()
                el = time.perf_counter() - t1
                st = reducer.overlap_stats()
                reducer.time_collectives = False
                tt_ = torch.tensor([el], device=device, dtype=torch.float64)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                per = st.get("collective_ms", [])
                n_coll = max(1, len(per) // max(1, args.sweep_steps))
                last = per[-n_coll:]                         # the last step's collectives, in launch order
                ms = sorted(m for _, m in per)
                comm_sweep.append({"reserve_cus": res, "wire_dtype": str(wire).replace("torch.", ""),
                                   "ms_per_step": round(float(tt_.item()) / args.sweep_steps * 1e3, 2),
                                   "exposed_wait_ms_per_step": None if st["exposed_wait_ms_per_step"] is None else round(st["exposed_wait_ms_per_step"], 3),
                                   "allreduce_ms": ({"n": len(ms), "min": round(ms[0], 3), "median": round(ms[len(ms) // 2], 3), "max": round(ms[-1], 3),
                                                     "sum_per_step": round(sum(ms) / args.sweep_steps, 3)} if ms else None),
                                   "last_step_buckets": [{"bytes": int(nb), "ms": round(m, 3), "GBps": round(nb / m / 1e6, 1) if m > 0 else None}
                                                         for nb, m in last]})
        reducer.wire_dtype, reducer.reserve_cus = base_wire, base_res

    roofline = None
    if survey and args.gemm_report and rank == 0:
        per = {}
        for key, flops, shape, e0, e1 in survey:
            r = per.setdefault((key, shape), [0.0, 0.0, 0])
            r[0] += flops
            r[1] += e0.elapsed_time(e1)
            r[2] += 1
        with open(args.gemm_report, "w") as f:
            for (key, shape), (fl, ms, n) in sorted(per.items(), key=lambda kv: -kv[1][1]):
                f.write(json.dumps({"layout": LAYOUT_NAMES[key[:2]], "epi": EPI_NAMES[key[2]], "kernel": key[3],
                                    "MNK": list(shape),
                                    "launches_per_step": n, "ms_per_step": round(ms, 3),
                                    "avg_ms": round(ms / n, 4), "tflops": round(fl / ms / 1e9, 1)}) + "\n")
    if timing:
        groups = {}
        for key, flops, shape, e0, e1 in timing:
            gsum = groups.setdefault(key, [0.0, 0.0, 0])
            gsum[0] += flops
            gsum[1] += e0.elapsed_time(e1)
            gsum[2] += 1
        key = max(groups, key=lambda k: groups[k][1])
        fl, ms, n = groups[key]
        sym = {"w4m256": "of_gemm_w4m_kernel", "w4dma256": "of_gemm_w4_kernel", "pingpong256": "of_gemm_pp_kernel", "mid128": "of_gemm_mid_kernel",
               "general128": "of_gemm_kernel", "skinny": "of_gemm_skinny_kernel", "mid128batch": "of_gemm_mid_batch_kernel",
               "w4h256x128": "of_gemm_w4h_kernel"}[key[3]]
        if key[3] == "w4h256x128":          # <BT, EPI, VAR = 0> (A is never transposed on this kernel)
            sym += f"<{str(bool(key[1])).lower()}, {key[2]}, 0>"
        else:                               # w4m256: <AT, BT, EPI, SK = false> (one tile per workgroup)
            sym += f"<{str(bool(key[0])).lower()}, {str(bool(key[1])).lower()}, {key[2]}" + (
                ", false>" if key[3] == "w4m256" else ">" if key[3] in ("mid128", "mid128batch") else ", ...>")
        shapes = {}
        for k2, _, shape, _, _ in timing:
            if k2 == key:
                shapes[shape] = shapes.get(shape, 0) + 1
        top_shape = max(shapes, key=shapes.get)
        if survey:      # every of_gemm launch of the last warm-up step
            all_fl = sum(f for _, f, _, _, _ in survey) * args.steps
            all_ms = sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in survey) * args.steps
        else:
            all_fl = sum(v[0] for v in groups.values())
            all_ms = sum(v[1] for v in groups.values())
        fam = {}
        for k2, f2, _, e0, e1 in (survey or []):
            v = fam.setdefault(k2, [0.0, 0.0, 0])
            v[0] += f2
            v[1] += e0.elapsed_time(e1)
            v[2] += 1
        families = [{"kernel": k[3], "layout": LAYOUT_NAMES[k[:2]], "epi": EPI_NAMES[k[2]], "launches_per_step": v[2],
                     "ms_per_step": round(v[1], 2), "frac": round(v[0] / v[1] / 1e9 / MFMA_PEAK_TFLOPS, 4)}
                    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:4]]
        ach = fl / ms / 1e9
        # the same family's MEDIAN launch: since round 6 the last launches of a backward run next to the prefetched tower's kernels on
        # the side stream (Flamingo.schedule_vision_prefetch) and take longer for it -- the mean (`achieved`, `frac`) includes them
        durs = sorted(e0.elapsed_time(e1) for k2, _, sh, e0, e1 in timing if k2 == key and sh == top_shape)
        med_ms = durs[len(durs) // 2]
        med_fl = next(f for k2, f, sh, _, _ in timing if k2 == key and sh == top_shape)
        traffic, traffic_note = pmc_traffic(key, top_shape)
        if traffic is None:          # a family can hold two launch shapes of equal count (dW1 / dW2): quote the one on record
            for sh in sorted(shapes, key=shapes.get, reverse=True):
                t2, n2 = pmc_traffic(key, sh)
                if t2 is not None:
                    traffic, traffic_note = t2, n2
                    break
                if "ran libofhip" in n2:          # a record exists, of another build of the library: say THAT
                    traffic_note = n2
        roofline = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": f"{sym}  = of_gemm {LAYOUT_NAMES[key[:2]]}, epilogue {EPI_NAMES[key[2]]}",
                    "shapes_MNK": {"x".join(map(str, sh)): c // args.steps for sh, c in shapes.items()},
                    "launches_per_step": n // args.steps, "avg_launch_ms": round(ms / n, 4),
                    "gflop_per_launch": round(fl / n / 1e9, 2),
                    "median_launch_ms": round(med_ms, 4), "frac_at_median": round(med_fl / med_ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                    # the four (layout, epilogue, kernel) families that take the most time in the surveyed warm-up step
                    "families": families,
                    "all_gemm_tflops": round(all_fl / all_ms / 1e9, 1),
                    # the time-weighted figure over EVERY of_gemm launch of a step: what describes the path (`frac` is the ONE family that takes the most time)
                    "all_gemm_frac": round(all_fl / all_ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                    "all_gemm_ms_per_step": round(all_ms / args.steps, 2),
                    "note": "achieved/avg_launch_ms: HIP events around every launch of this family inside the timed region (the mean includes the launches "
                            "that share the chip with the next step's tower forward on the side stream; median_launch_ms / frac_at_median: the family's typical launch); "
                            "all_gemm_*: every of_gemm launch of the last warm-up step"}

    if rank == 0:
        out = {"metric": "train images/sec (+ step ms) OF-3B ViT-L/14+MPT-1B, 1/2/4/8 MI355X", "value": round(value, 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               # the training loss of the last timed step (read after the timed region): a run whose arithmetic broke -- NaN or garbage
               # activations draw less power and time FASTER on this chip -- shows here (a same-box A/B of round 4 was void for that reason)
               "loss_last_step": round(float(loss), 4) if loss is not None else None,
               "config": {"baseline_config": cfg_name,
                          "workload": f"{args.family} (ViT-L/14 + {'MPT-1B' if args.family == 'OF-3B' else args.family}, "
                                      f"xattn_every={info['every']}) full train step, amp_bf16, per-GPU B={args.batch} "
                                      f"T={args.T} F=1 L={args.L} synthetic MMC4-style batch, random-init weights",
                          **({"one_gpu_loopback": f"REHEARSAL: {world} ranks on ONE device over RCCL's socket transport -- the value "
                                                      "on this line is not a throughput"} if args.one_gpu_loopback and world > 1 else {}),
                          "global_batch": args.batch * world, "images_per_step": images, "seq_len": args.L,
                          "parallelism": f"dp{world}",
                          "frozen_tower_weights": "fp32 (re-cast by autocast)" if args.frozen_fp32 else "bf16 copies held",
                          "frozen_vision_tower": args.vision if not args.frozen_fp32 else "modules", "frozen_lm_blocks": args.lm_blocks if not args.frozen_fp32 else "modules",
                          "frozen_lm_attention": args.lm_attention, "frozen_tower_layernorm": args.tower_layernorm,
                          "lm_loss": args.lm_loss, "nan_check": "device (step epilogue)" if nan_check == "device" else "host (torch.isnan(loss))",
                          "vendor_gemm_table": f"TunableOp table, {n_tuned} shapes, tuning off" if n_tuned else "library defaults",
                          "grad_wire_dtype": "bf16" if args.wire_bf16 else "fp32", "reserve_cus": args.reserve_cus,
                          "embedding_row_gradient": "sparse taps" if args.sparse_embedding_rows else "dense, masked",
                          "global_norm": ("torch clip_grad_norm_" if args.torch_optimizer else
                                          f"FFN weight gradients' share from their dW GEMM epilogues ({getattr(opt, 'tapped_buckets', 0)} buckets), "
                                          "the rest in one pass" if getattr(opt, "tapped_buckets", 0) else "one pass over the gradients"),
                          "step_epilogue_launch": (f"narrow: {int(getattr(opt, 'narrow_cus', 0))} fat workgroups" if int(getattr(opt, "narrow_cus", 0)) else "covers the chip"),
                          "vision_tower_schedule": ("at the start of the step" if args.no_vision_prefetch else
                                                    "next step's tower forward on a side stream next to the step epilogue"),
                          "laion_pass": (f"B={args.laion_batch} T=1 L=32, loss x0.2, same optimizer step" if args.laion_batch else "off")},
               "loss": None if loss is None else round(float(loss), 4),
               "libofhip_sha16": _lib_sha16()}
        fl = step_flops(info, args.batch, args.T, args.L)
        if args.laion_batch:
            fl = {k: v + step_flops(info, args.laion_batch, 1, 32)[k] for k, v in fl.items()}
        floor_ms = fl["total"] / (MFMA_PEAK_TFLOPS * 1e12) * 1e3
        out["floor"] = {"algorithmic_tflop_per_step": round(fl["total"] / 1e12, 2), "hot_path_tflop": round(fl["hot_path"] / 1e12, 2),
                        "frozen_lm_tflop": round(fl["frozen_lm"] / 1e12, 2), "vit_tflop": round(fl["vit"] / 1e12, 2),
                        "floor_ms": round(floor_ms, 2), "step_frac_of_floor": round(floor_ms / ms_per_step, 4),
                        "note": "SURVEY.md 8d closed forms (hot path x3, frozen LM x2, ViT x1; restricted cross-attention window) / 2.5 "
                                "PFLOP/s dense bf16: no step on this chip can be faster than floor_ms, so the north_star's '>= 5x the "
                                "reference's single-GPU step' is reachable only while 5 x floor_ms < reference_eager.ms_per_step"}
        out["overlap"] = {"rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1),
                          "backend": (dist.get_backend() if dist.is_initialized() else None),
                          "allreduce_bytes_per_step_per_gpu": int(overlap["allreduce_bytes_per_step"]),
                          "collectives_per_step": overlap["collectives_per_step"], "wire_dtype": overlap["wire_dtype"],
                          "buckets": overlap["buckets"],
                          "exposed_wait_ms_per_step": None if overlap["exposed_wait_ms_per_step"] is None
                          else round(overlap["exposed_wait_ms_per_step"], 3),
                          "note": "one exchange per optimizer step on a side HIP stream (RCCL all-reduce per gated block + "
                                  "per Perceiver layer, launched as the backward produces them); exposed_wait = time the "
                                  "compute stream waited in GradReducer.finish(), HIP events, rank 0"}
        if comm_sweep is not None:
            out["comm_sweep"] = {"steps_per_setting": args.sweep_steps, "settings": comm_sweep,
                                 "note": "measured after the timed region of this same call (value / ms_per_step above are the default setting's); "
                                         "allreduce_ms: HIP events on the side stream around each bucket's collective -- with them on, the side "
                                         "stream waits for every collective before the next is enqueued; GBps = wire bytes of this rank / latency"}
        if roofline is not None:
            out["roofline"] = roofline
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(info, min(os.cpu_count() or 1, 64), T=args.T, L=args.L)
            except Exception as exc:  # the baseline is a reported number, never the thing measured
                out["cpu_baseline"] = {"error": repr(exc)}
        if world == 1 and not args.no_reference_eager:
            try:
                del model, opt, reducer, batch, laion, step_kw          # the eager model gets the HBM back
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                ref = reference_eager_step(args.family, args.batch, args.T, args.L, stock=False)
                out["reference_eager"] = ref
                out["reference_eager_ms_per_step"] = ref["ms_per_step"]
                # BASELINE.md holds no published number for this metric (BASELINE.json "published": {}); the baseline the
                # north_star names is the reference's own single-GPU step, measured here on the same box.  The ratio quoted is
                # the CONSERVATIVE one (the eager model with bench.py's tower-side choices); the stock-tower reference is slower.
                out["vs_baseline"] = round(ref["ms_per_step"] / ms_per_step, 3)
                out["vs_baseline_note"] = ("reference_eager.ms_per_step / ms_per_step, same box, same process (no published number "
                                           "exists for this metric: BASELINE.json published = {}); reference_eager_stock_towers is the "
                                           "reference as a user would run it")
                if not args.no_reference_eager_stock:
                    del ref
                    gc.collect()
                    torch.cuda.empty_cache()
                    stock = reference_eager_step(args.family, args.batch, args.T, args.L, stock=True)
                    out["reference_eager_stock_towers"] = stock
                    out["vs_reference_stock_towers"] = round(stock["ms_per_step"] / ms_per_step, 3)
            except Exception as exc:
                out["reference_eager"] = {"error": repr(exc)}
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
