"""Data-parallel gradient exchange for the trainable part of Flamingo (Perceiver + gated cross-attention blocks +
the two new embedding rows), RCCL over xGMI.

Replaces ``DistributedDataParallel(model)`` of the reference (open_flamingo/train/train.py:364-366), whose default
reducer all-reduces every trainable gradient inside EACH ``backward()`` -- twice per optimizer step
(train_utils.py:118,172) -- including the dense (vocab x d) embedding gradient that is then masked down to two rows
(train_utils.py:174-196).  Gradients are linear, so:

  * one exchange per optimizer step: hooks only fire the collectives during the LAST backward of the step
    (``with reducer.no_sync():`` around the earlier ones, same contract as DDP.no_sync);
  * buckets follow backward order: one bucket per gated cross-attention block (last block first), then one per
    Perceiver layer (final norm + last layer first ... first layer + latents last; the hand-written Perceiver backward
    reports each layer's gradients as soon as its kernels are enqueued), so each all-reduce overlaps the backward work
    that follows it and the exposed tail is one Perceiver layer (~42 MB fp32), not the whole Perceiver (252 MB).  The
    collectives run on a dedicated side stream; the compute stream only waits in ``finish()`` (before clipping / the
    optimizer), and that wait is timed with HIP events (``overlap_stats``);
  * parameter ``.grad``s are views into the flat fp32 bucket buffers: no flatten/unflatten copies;
  * only the ``<image>`` / ``<|endofchunk|>`` rows of the input-embedding gradient travel (2 x d floats instead of
    vocab x d), and the rest of that gradient is zeroed here -- exactly the reference's post-all-reduce mask.

xGMI is point to point (7 links x ~153 GB/s per GPU): with ~147 MB fp32 per block bucket the collective is
bandwidth-, not latency-bound, and RCCL is free to use all links; ``wire_dtype=torch.bfloat16`` halves the bytes.
"""
import contextlib

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, process_group=None, wire_dtype=torch.float32, embedding_rows=None,
                 force_collectives=False, perceiver_buckets="layer", reserve_cus=0):
        """model: a Flamingo (or any module exposing .perceiver and .lang_encoder.gated_cross_attn_layers);
        embedding_rows: token ids whose input-embedding gradient rows are kept (media + endofchunk);
        perceiver_buckets: "layer" (one bucket per Perceiver layer, backward order) or "one";
        reserve_cus: CUs to leave to RCCL's kernels while collectives are in flight (0 = off).  A 256-tile GEMM launch needs every
        CU: the workgroups a collective displaces run as a SECOND ROUND (measured with a stand-in: +3..5 % per step, DESIGN.md
        section 5).  With reserve_cus = R the libofhip GEMMs of the backward are laid out stream-K for 256 - R workgroups from the
        first bucket's launch until finish() (hip/ops.py: Ops.cu_limit -> OfGemmArgs.cu_limit): every workgroup gets the same share
        of the launch, nothing waits for a second round.  What R should be on a real node (RCCL's channel count x workgroup
        footprint) is a measurement this one-GPU pool cannot make; the default leaves the launches as they are."""
        self.module = model                      # DDP-style handle (train_utils.py:181 reaches through .module)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.wire_dtype = wire_dtype
        # run the side-stream collectives even when world_size == 1 (tests: the single-GPU box can then exercise the
        # exact RCCL / stream code path the multi-GPU runs take)
        self.force_collectives = bool(force_collectives) and dist.is_initialized()
        self.embedding_rows = list(embedding_rows) if embedding_rows is not None else None
        self.reserve_cus = int(reserve_cus)
        self._sync = True
        self._pending = []
        self._stream = None
        # on_bucket_final(bucket_index): set by the fused step epilogue (train/optim.py).  Called once per bucket and optimizer step,
        # on the stream on which the bucket's FINAL values (all-reduced, or complete on one GPU) are ordered -- the side stream -- so
        # that the bucket's share of the global gradient norm is computed under the rest of the backward instead of at the serial end
        # of the step.  The callback's result is valid while bucket["early_gen"] == self.generation: ``generation`` counts zero_grad()
        # calls, and any later gradient write into the bucket (a parameter hook) resets early_gen.
        self.on_bucket_final = None
        self.generation = 0
        self.stats = dict(collectives=0, bytes=0, steps=0)
        self.time_waits, self._wait_events = False, []
        # diagnostics (bench.py --sweep-comm): HIP events on the side stream around every collective -> per-bucket all-reduce latency.
        # Costs one stream wait per collective (the side stream waits for each before the next is enqueued), so it is off by default.
        self.time_collectives, self._coll_events = False, []
        lm = model.lang_encoder
        groups = []
        for blk in reversed([b for b in lm.gated_cross_attn_layers if b is not None]):
            groups.append(("xattn", [p for p in blk.parameters() if p.requires_grad]))
        groups.extend(("perceiver", g) for g in self._perceiver_groups(model.perceiver, perceiver_buckets))
        self.embedding = None
        emb = lm.get_input_embeddings().weight
        # opt-in train/sparse_rows.py: the table is frozen for autograd and the kept rows' gradient arrives in a small
        # leaf (autograd sums the tied-head tap's and the lookup tap's contributions before it accumulates: one
        # post-accumulate callback per backward); it is exchanged like a bucket
        self.sparse = getattr(model, "_of_sparse_rows", None)
        if self.sparse is not None:
            assert self.embedding_rows is not None and list(self.sparse.rows.tolist()) == self.embedding_rows, \
                "sparse_rows.enable(model, rows) and GradReducer(embedding_rows=rows) must name the same rows in the same order"
            self.embedding = emb
            self.sparse.leaf.register_post_accumulate_grad_hook(self._sparse_hook)
        elif emb.requires_grad:
            self.embedding = emb
        self.buckets = []
        # Weight matrices whose gradient the libofhip backward produces with ONE GEMM can be overwritten (beta = 0) instead
        # of zeroed by the step epilogue and then read-modify-written (beta = 1): they go to the back of their bucket so
        # that the epilogue only has to clear the (tiny) front part.  Only modules that implement the protocol
        # (src/helpers.py: `overwrites_fresh_grads`) take part; see FlatAdamW.step.
        owner = set()
        for mod in (model.perceiver, *[b for b in lm.gated_cross_attn_layers if b is not None]):
            if getattr(mod, "overwrites_fresh_grads", False):
                for sub in mod.modules():
                    if isinstance(sub, torch.nn.Linear):
                        owner.add(id(sub.weight))
        for kind, params in groups:
            if not params:
                continue
            front = [p for p in params if id(p) not in owner]
            back = [p for p in params if id(p) in owner]
            params = front + back
            # every parameter starts on a 256-byte boundary (64 fp32 = 128 bytes of bf16: whole L2 lines for the
            # operand DMA): the fused step epilogue keeps fp32 master / bf16 operand copies at the same offsets and
            # the GEMM kernels need 16-byte aligned operands
            offsets, n = [], 0
            for p in params:
                offsets.append(n)
                n += (p.numel() + 63) // 64 * 64
            flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
            for p, off in zip(params, offsets):
                p.grad = flat[off:off + p.numel()].view_as(p)      # gradient-as-bucket-view
            self.buckets.append(dict(flat=flat, params=params, offsets=offsets, ready=0, kind=kind,
                                     overwritable_from=(offsets[len(front)] if back else n), overwritable=back))
        self._param_bucket = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._param_bucket[p] = bi
                p.register_post_accumulate_grad_hook(self._hook)
                # the libofhip backward (src/helpers.py) accumulates straight into the bucket view and then calls
                # _of_on_grad itself, instead of returning a fresh gradient for autograd to add (one add kernel per
                # parameter per backward).  Modules that do not know the protocol ignore the attributes.
                p._of_inplace_grad, p._of_on_grad = True, self._hook_inplace
        if self.embedding is not None and self.sparse is None:
            self.embedding.register_post_accumulate_grad_hook(self._emb_hook)

    @staticmethod
    def _perceiver_groups(perceiver, mode):
        """Trainable Perceiver parameters grouped in the order their gradients become final in the backward
        (reference helpers.py:129-132 run in reverse): [norm + layers.(depth-1)], layers.(depth-2), ..., [layers.0 +
        latents + position tables]."""
        named = [(n, p) for n, p in perceiver.named_parameters() if p.requires_grad]
        if mode != "layer" or not any(n.startswith("layers.") for n, _ in named):
            return [[p for _, p in named]] if named else []
        depth = 1 + max(int(n.split(".")[1]) for n, _ in named if n.startswith("layers."))
        per_layer = [[] for _ in range(depth)]
        for n, p in named:
            if n.startswith("layers."):
                per_layer[int(n.split(".")[1])].append(p)
            elif n.startswith("norm."):
                per_layer[depth - 1].append(p)
            else:                                   # latents, frame_embs, media_time_embs: final after layer 0
                per_layer[0].append(p)
        return [g for g in reversed(per_layer) if g]

    # ------------------------------------------------------------------
    def _side_stream(self, device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _launch(self, flat):
        """all-reduce(avg) of one flat buffer on the side stream (or inline on CPU/gloo).  Returns the collective's work handle when
        the reduced values land in ``flat`` itself (None: no collective, or a wire copy that finish() writes back)."""
        if self.world == 1 and not self.force_collectives:
            return None
        side = self._side_stream(flat.device)
        if self.reserve_cus > 0 and side is not None:          # GEMMs enqueued from here on count on the CUs RCCL leaves
            from ..hip.ops import Ops
            Ops.default().cu_limit = 256 - self.reserve_cus      # process-wide: released in finish() AND in zero_grad() / no_sync()
        compute = None
        if side is not None:
            compute = torch.cuda.current_stream(flat.device)
            side.wait_stream(compute)
            ctx = torch.cuda.stream(side)
        else:
            ctx = contextlib.nullcontext()
        self.stats["collectives"] += 1
        self.stats["bytes"] += flat.numel() * torch.empty((), dtype=self.wire_dtype).element_size()
        with ctx:
            if self.wire_dtype != flat.dtype:
                wire = flat.to(self.wire_dtype)
                if side is not None:
                    wire.record_stream(compute)      # allocated on the side stream, read back on the compute stream in finish()
                e0 = None
                if self.time_collectives and side is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if e0 is not None:
                    work.wait()
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(side)
                    self._coll_events.append((wire.numel() * wire.element_size(), e0, e1))
                self._pending.append((work, flat, wire))
                return None
            e0 = None
            if self.time_collectives and side is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(side)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if e0 is not None:
                work.wait()                  # (a stream wait: the side stream follows the collective's end)
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(side)
                self._coll_events.append((flat.numel() * flat.element_size(), e0, e1))
            self._pending.append((work, flat, None))
            return work

    def _bucket_final(self, bi, work):
        """The bucket's final values exist (after ``work``, if there was a collective): let the step epilogue take its share of the
        global norm now, on the side stream, under the rest of the backward (on_bucket_final, see __init__)."""
        early = self.on_bucket_final
        collective = self.world > 1 or self.force_collectives
        if early is None or (collective and work is None):       # wire copy: the reduced values reach the bucket in finish()
            return
        flat = self.buckets[bi]["flat"]
        side = self._side_stream(flat.device)
        if side is None:                                         # CPU / gloo: inline
            if work is not None:
                work.wait()
            early(bi)
            return
        if work is None:
            side.wait_stream(torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(side):
            if work is not None:
                work.wait()          # orders the side stream behind the collective (a stream wait with RCCL, no host block)
            early(bi)

    def _hook_inplace(self, p):
        """Called by the libofhip backward for a parameter whose gradient it accumulated in place.  torch's engine still
        runs the parameter's post-accumulate hook afterwards although the Function returned None for it (observed on
        torch 2.10: 91 + 91 calls for 91 parameters) -- counted twice, a bucket would be exchanged when only half of its
        gradients exist, and once more at the end.  The mark makes the engine's call for the same backward a no-op."""
        self.buckets[self._param_bucket[p]]["early_gen"] = None      # the bucket changed: an earlier partial norm of it is void
        if not self._sync:
            return
        p._of_notified = True
        self._count(p)

    def _hook(self, p):
        if self._sync and getattr(p, "_of_notified", False):       # already counted by _hook_inplace in this backward
            p._of_notified = False
            return
        self.buckets[self._param_bucket[p]]["early_gen"] = None
        if not self._sync:
            return
        self._count(p)

    def _count(self, p):
        b = self.buckets[self._param_bucket[p]]
        b["ready"] += 1
        if b["ready"] == len(b["params"]):
            b["ready"] = 0
            self._bucket_final(self._param_bucket[p], self._launch(b["flat"]))

    def _sparse_hook(self, leaf):
        if self._sync:
            self._launch(leaf.grad)

    def _emb_hook(self, p):
        if not self._sync:
            return
        if self.embedding_rows is None:      # a fully trainable table: exchange the dense gradient like DDP would
            self._launch(p.grad)
            return
        rows = torch.as_tensor(self.embedding_rows, device=p.grad.device)
        kept = p.grad.index_select(0, rows).contiguous()
        p.grad.zero_()                                   # train_utils.py:184-196: every other row is masked to 0
        self._emb_rows = (rows, kept)
        self._launch(kept)

    def _release_cus(self):
        """Give the CUs reserved for collectives (reserve_cus -> the process-wide Ops.cu_limit, set by the first bucket's launch) back
        to the GEMMs.  finish() does it; so do zero_grad() and no_sync(), so that a step that raised between its first bucket and
        finish() cannot leave every later GEMM of the process -- forwards, other models -- laid out for fewer CUs."""
        if self.reserve_cus > 0 and self._stream is not None:
            from ..hip.ops import Ops
            Ops.default().cu_limit = 0

    @contextlib.contextmanager
    def no_sync(self):
        """Skip the exchange for backward passes that are not the last of the optimizer step."""
        old, self._sync = self._sync, False
        self._release_cus()
        try:
            yield
        finally:
            self._sync = old

    def finish(self, average=True):
        """Make the compute stream wait for all collectives of this step and write back.  average=True applies 1/world
        here (DDP semantics: .grad holds the mean); average=False leaves the SUM and sets ``holds_sum`` for a consumer
        that folds the 1/world into its own pass over the gradients (train/optim.py)."""
        # Every pending collective is waited for (ProcessGroupNCCL: work.wait() = one stream-wait on the work's end event, a few
        # microseconds of host time each, ~30 per step).  Round 3 waited for the LAST one only, trusting that a process group's
        # collectives run in issue order on one RCCL stream -- true for today's defaults, not a contract: a group configured with
        # several streams, or a bucket launched through another group, would have let the step epilogue read a bucket still in
        # flight.  On CPU / gloo the waits complete on the host.
        try:
            for work, _, _ in self._pending:
                work.wait()
        finally:
            self._release_cus()
        for _, flat, wire in self._pending:
            if wire is not None:
                flat.copy_(wire)
            if self.world > 1 and average:
                flat.div_(self.world)
        if self.world > 1 and average:
            for b in self.buckets:
                b["early_gen"] = None            # the buckets were rescaled after their partial norms were taken
        self._pending.clear()
        for b in self.buckets:                       # a torch build whose engine skips the hook of a None gradient
            for p in b["params"]:                    # would leave the marks set: never carry them into the next step
                p._of_notified = False
        self.holds_sum = self.world > 1 and not average
        if self._stream is not None:
            cur = torch.cuda.current_stream()
            if self.time_waits:              # how long the compute stream sits waiting for RCCL: the EXPOSED exchange time
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._stream)
                e1.record(cur)
                self._wait_events.append((e0, e1))
            else:
                cur.wait_stream(self._stream)
        self.stats["steps"] += 1
        if getattr(self, "_emb_rows", None) is not None:
            rows, kept = self._emb_rows     # `kept` is one of the all-reduced buffers above: same sum/mean convention
            self.embedding.grad.index_copy_(0, rows, kept)
            self._emb_rows = None

    def overlap_stats(self, reset=True):
        """Per-step averages since the last reset: collectives launched, bytes put on the wire by this rank, and -- when
        ``time_waits`` was set -- the milliseconds the compute stream waited for the side stream in finish()."""
        n = max(1, self.stats["steps"])
        waited = None
        if self._wait_events:
            torch.cuda.synchronize()
            waited = sum(a.elapsed_time(b) for a, b in self._wait_events) / len(self._wait_events)
        out = dict(collectives_per_step=self.stats["collectives"] / n, allreduce_bytes_per_step=self.stats["bytes"] / n,
                   wire_dtype=str(self.wire_dtype).replace("torch.", ""), buckets=len(self.buckets),
                   exposed_wait_ms_per_step=waited)
        if self._coll_events:          # [(bytes on the wire, milliseconds)] in launch order over the steps since the last reset
            torch.cuda.synchronize()
            out["collective_ms"] = [(nb, a.elapsed_time(b)) for nb, a, b in self._coll_events]
        if reset:
            self.stats = dict(collectives=0, bytes=0, steps=0)
            self._wait_events = []
            self._coll_events = []
        return out

    def zero_grad(self, flat_already_zero=False):
        """Zero in place (the .grad views must stay attached to the buckets).  ``flat_already_zero``: the fused step
        epilogue (train/optim.py) cleared the buckets in its AdamW pass."""
        self.generation += 1
        self._release_cus()
        for b in self.buckets:
            if not flat_already_zero:
                b["flat"].zero_()
                for p in b["overwritable"]:
                    p._of_grad_fresh = False
            b["ready"] = 0
        if self.embedding is not None and self.embedding.grad is not None:
            self.embedding.grad = None
        if self.sparse is not None:
            self.sparse.clear()

    def broadcast_parameters(self, src=0):
        """DDP-constructor equivalent (train.py:366): make every rank start from rank ``src``'s trainable weights."""
        if self.world == 1:
            return
        for b in self.buckets:
            for p in b["params"]:
                dist.broadcast(p.data, src=src, group=self.group)
        if self.embedding is not None:
            dist.broadcast(self.embedding.data, src=src, group=self.group)
