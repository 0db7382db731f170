"""world_size=2 (gloo, CPU) test of the data-parallel path: GradReducer buckets + one exchange per optimizer step +
2-row embedding exchange + train_step semantics.  The model is the tiny Flamingo with the oracle's hot-path modules
(the reducer only touches .grad tensors, so it is independent of which kernels produced them)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from open_flamingo_amd.train import distributed, step, synthetic
    from open_flamingo_amd.train.reducer import GradReducer
    from tests.cpu_model import tiny_cpu_flamingo
    dev = distributed.init_distributed_device(backend="gloo")
    assert dist.get_world_size() == world and dev.type == "cpu"
    model, info = tiny_cpu_flamingo(seed=0)
    rows = [info["media_token_id"], info["eoc_token_id"]]
    # local reference gradients (no reducer), two passes like LAION + MMC4
    b_laion = synthetic.make_batch(2, 1, 16, info, "cpu", seed=10 + rank)
    b_mmc4 = synthetic.make_batch(2, 2, 24, info, "cpu", seed=20 + rank)
    for b in (b_laion, b_mmc4):
        step.forward_loss(model, b, info, amp=False).backward()
    trainable = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    local = {n: p.grad.detach().clone() for n, p in trainable}
    want = {}
    for n, g in local.items():
        t = g.clone()
        dist.all_reduce(t)
        want[n] = t / world
    emb_name = [n for n, _ in trainable if "wte" in n][0]
    mask = torch.zeros_like(want[emb_name])
    mask[rows] = 1
    want[emb_name] = want[emb_name] * mask
    model.zero_grad(set_to_none=True)
    # the product path: reducer + train_step pieces
    red = GradReducer(model, embedding_rows=rows)
    red.broadcast_parameters()
    with red.no_sync():
        step.forward_loss(model, b_laion, info, amp=False).backward()
    step.forward_loss(model, b_mmc4, info, amp=False).backward()
    red.finish()
    errs = {}
    for n, p in trainable:
        errs[n] = (p.grad - want[n]).abs().max().item() / (want[n].abs().max().item() + 1e-12)
    nz_rows = int((model.lang_encoder.get_input_embeddings().weight.grad.abs().sum(-1) > 0).sum())
    red.zero_grad()
    # a full train_step keeps replicas identical
    opt = step.build_optimizer(model)
    loss = step.train_step(model, red, opt, b_mmc4, info, batch_laion=b_laion, amp=False)
    chk = torch.cat([p.detach().flatten()[:64] for _, p in trainable]).double()
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    # the fused step epilogue (train/optim.py; kernels on the host emulator here) must make the same update as
    # clip_grad_norm_ + torch AdamW from the same state: the reducer then hands it the all-reduced SUM and it folds the
    # 1/world into its AdamW pass
    from open_flamingo_amd.train.optim import FlatAdamW
    from tests.emu import harness as H
    # ... and so must the opt-in sparse embedding-row path (train/sparse_rows.py): its (2, d) gradient leaf is exchanged
    # instead of rows cut out of the dense gradient
    from open_flamingo_amd.train import sparse_rows
    finals = []
    for fused, sparse in ((False, False), (True, False), (True, True)):
        m2, _ = tiny_cpu_flamingo(seed=0)
        if sparse:
            sparse_rows.enable(m2, rows)
        r2 = GradReducer(m2, embedding_rows=rows)
        o2 = FlatAdamW(r2, lr=1e-3, ops=H.emu_ops()) if fused else step.build_optimizer(m2, lr=1e-3)
        for _ in range(2):
            step.train_step(m2, r2, o2, b_mmc4, info, batch_laion=b_laion, amp=False)
        finals.append(torch.cat([p.detach().flatten() for _, p in m2.named_parameters()
                                 if sparse_rows.is_trainable(p)]).double())
    fused_err = max((finals[0] - finals[1]).abs().max().item(), (finals[1] - finals[2]).abs().max().item())
    q.put((rank, max(errs.values()), nz_rows, same, float(loss), len(red.buckets), fused_err))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_reducer_and_train_step_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, nz_rows, same, loss, nb, fused_err in res:
        assert fused_err < 2e-5, (f"rank {rank}: fused step epilogue (dense or sparse embedding rows) diverges from "
                                  f"clip + torch AdamW ({fused_err})")
        assert err < 1e-5, f"rank {rank}: reduced grads differ from mean of local grads ({err})"
        assert nz_rows <= 2, "embedding gradient must be masked to the <image>/<|endofchunk|> rows"
        assert same, "replicas diverged after train_step"
        assert nb == 2 + 6, "tiny model: 2 xattn block buckets + one bucket per Perceiver layer (depth 6)"


def _worker_product_modules(rank, world, port, q):
    """The PRODUCT modules (libofhip kernels on the host emulator) under a world-2 gloo group: gradients are accumulated in
    place into the reducer's buckets, the Perceiver reports its layers as they finish (per-layer buckets), the step
    epilogue leaves weight gradients for the next backward to overwrite, the grouped media projection is on."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from open_flamingo_amd.hip.ops import Ops
    from open_flamingo_amd.src import helpers
    from open_flamingo_amd.train import distributed, step, synthetic, towers
    from open_flamingo_amd.train.optim import FlatAdamW
    from open_flamingo_amd.train.reducer import GradReducer
    from tests.emu import harness as H
    helpers._require_hip = lambda t, what: None                  # test-only: kernels run on the host emulator
    helpers.can_group_media = lambda media: True
    Ops.default = staticmethod(H.emu_ops)
    distributed.init_distributed_device(backend="gloo")
    model, info = towers.build_flamingo("OF-tiny", device="cpu", seed=rank, gates=0.5, fused_lm_attention=False,
                                        vision_kw=dict(width=256, layers=1, heads=2, patch=14, image=56), perceiver_depth=2)
    model.train()
    rows = [info["media_token_id"], info["eoc_token_id"]]
    red = GradReducer(model, embedding_rows=rows)
    red.broadcast_parameters()                                    # ranks started from different seeds on purpose
    opt = FlatAdamW(red, lr=1e-3, ops=H.emu_ops())
    launches = []
    orig = red._launch
    red._launch = lambda flat: (launches.append(flat.numel()), orig(flat))[1]
    b_laion = synthetic.make_batch(2, 1, 16, info, "cpu", seed=10 + rank, image_size=56)
    b_mmc4 = synthetic.make_batch(2, 2, 24, info, "cpu", seed=20 + rank, image_size=56)
    # exchanged buckets == mean over ranks of the local buckets (a bucket exchanged before all of its gradients exist, or
    # twice, would not be)
    with red.no_sync():
        step.forward_loss(model, b_mmc4, info, amp=False).backward()
    want = []
    for b in red.buckets:
        t = b["flat"].clone()
        dist.all_reduce(t)
        want.append(t / world)
    red.zero_grad()
    step.forward_loss(model, b_mmc4, info, amp=False).backward()
    red.finish()
    bucket_err = max(((b["flat"] - w).abs().max() / (w.abs().max() + 1e-12)).item() for b, w in zip(red.buckets, want))
    red.zero_grad()
    launches.clear()
    losses = [float(step.train_step(model, red, opt, b_mmc4, info, batch_laion=b_laion, amp=False))]     # LAION + MMC4 pass, one exchange
    per_step = len(launches)
    chk = torch.cat([p.detach().flatten()[:128] for p in model.parameters() if p.requires_grad]).double()
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    fresh = all(p._of_grad_fresh for b in red.buckets for p in b["overwritable"])
    # a NaN loss on rank 0 ALONE, decided on the device (nan_check="device"): no rank leaves the collectives, the NaN reaches
    # both through the all-reduce, both step epilogues skip -- parameters unchanged and still identical on both ranks
    before = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad]).clone()
    real = step.forward_loss
    if rank == 0:
        step.forward_loss = lambda *a, **kw: real(*a, **kw) * float("nan")
    nan_loss = step.train_step(model, red, opt, b_mmc4, info, amp=False, nan_check="device")
    step.forward_loss = real
    after = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad])
    skipped = torch.equal(before, after) and bool(torch.isnan(nan_loss)) == (rank == 0)
    chk2 = after[::97].double().contiguous()
    gathered2 = [torch.zeros_like(chk2) for _ in range(world)]
    dist.all_gather(gathered2, chk2)
    skipped = skipped and all(torch.equal(gathered2[0], g) for g in gathered2)
    q.put((rank, losses, same, per_step, len(red.buckets), fresh, bucket_err, skipped))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_product_modules_world2_inplace_buckets_and_per_layer_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_product_modules, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, losses, same, per_step, nb, fresh, bucket_err, skipped in res:
        assert skipped, f"rank {rank}: a NaN loss on rank 0 must skip the update on every rank (device-side check)"
        assert bucket_err < 1e-5, f"rank {rank}: exchanged buckets differ from the mean of the local buckets ({bucket_err})"
        assert same, f"rank {rank}: replicas diverged"
        assert all(l == l for l in losses)
        assert nb == 2 + 2 and per_step == nb + 1, (nb, per_step)     # 2 gated blocks + 2 Perceiver layers: one all-reduce per bucket + the two embedding rows, once per step
        assert fresh, "the step epilogue must leave the nn.Linear gradients marked for overwrite"
