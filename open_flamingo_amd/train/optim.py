"""Step epilogue on the device: global-norm clip + AdamW over the flat buckets of ``GradReducer``.

Replaces, with the same arithmetic, the reference's
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); optimizer.step(); optimizer.zero_grad()
(open_flamingo/train/train_utils.py:199-216) for ``torch.optim.AdamW`` with weight decay only on the gated
cross-attention parameters (open_flamingo/train/train.py:392-408).  Differences, all result-preserving:

* parameters, gradients and both AdamW moments of a bucket are contiguous fp32 buffers (the ``nn.Parameter``s become
  views), so one ``of_sumsq_partial`` + one ``of_adamw_clip`` launch per bucket (plus one ``of_sumsq_finish``) replace ~160 multi-tensor launches; the clip
  coefficient is computed on the device, gradients are zeroed and the bf16 GEMM-operand copies of the new weights are
  written in the same pass (the modules pick them up instead of re-casting every step);
* the input embedding: only the ``<image>`` / ``<|endofchunk|>`` rows ever receive gradient (train_utils.py:174-196) and
  that group has no weight decay, so AdamW leaves every other row untouched (zero moments): updating just those rows is
  the dense update.
"""
import torch

from ..hip.ops import BF16, F32, Ops


class FlatAdamW(torch.optim.Optimizer):
    """A ``torch.optim.Optimizer`` (so ``torch.optim.lr_scheduler.*`` and the HF ``get_*_schedule_with_warmup`` helpers the
    reference's train.py uses accept it): ``param_groups`` hold the real parameters -- group 0 = gated cross-attention
    (weight decay), group 1 = the rest -- and every ``step()`` reads each group's current ``lr``."""

    def __init__(self, reducer, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1, max_norm=1.0, ops=None):
        if reducer.embedding is not None and reducer.embedding_rows is None:
            raise NotImplementedError("FlatAdamW updates only the kept embedding rows; a fully trainable table needs "
                                      "torch.optim.AdamW (train/step.py::build_optimizer picks it)")
        self.reducer, self.ops = reducer, ops
        self.betas, self.eps, self.max_norm = betas, eps, max_norm
        self.step_count = 0
        groups = [{"lr": lr, "params": [], "weight_decay": weight_decay},
                  {"lr": lr, "params": [], "weight_decay": 0.0}]                  # LR schedulers mutate ["lr"]
        self._views = {}            # param.data_ptr() -> (bf16 view, parameter version it mirrors, numel)
        for b in reducer.buckets:
            flat_g = b["flat"]
            flat_p = torch.zeros_like(flat_g)
            flat_b = torch.empty(flat_g.shape, dtype=BF16, device=flat_g.device)
            for p, off in zip(b["params"], b["offsets"]):
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + n].view(p.shape)
            b.update(flat_p=flat_p, flat_bf16=flat_b, m=torch.zeros_like(flat_g), v=torch.zeros_like(flat_g),
                     wd=weight_decay if b.get("kind") == "xattn" else 0.0)
            groups[0 if b["wd"] else 1]["params"].extend(b["params"])
        self.embedding = reducer.embedding
        if self.embedding is not None:
            rows = torch.as_tensor(reducer.embedding_rows, device=self.embedding.device)
            d = self.embedding.shape[1]
            self._emb = dict(rows=rows, m=torch.zeros(len(rows), d, device=rows.device),
                             v=torch.zeros(len(rows), d, device=rows.device))
            groups[1]["params"].append(self.embedding)
        super().__init__([g for g in groups if g["params"]] or groups,
                         dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 2:      # keep the two-group shape (reference train.py:392-408) even if one is empty
            have = {g["weight_decay"] != 0.0: g for g in self.param_groups}
            self.param_groups = [have.get(True, dict(groups[0], betas=tuple(betas), eps=eps)),
                                 have.get(False, dict(groups[1], betas=tuple(betas), eps=eps))]
        self._sumsq = None
        self._applied = None          # device int32: updates applied so far (NaN-skipped steps do not count)
        # Early norm partials: a bucket's per-workgroup sums of squares are computed as soon as its gradient is final (GradReducer
        # calls back on its side stream, right behind the bucket's all-reduce), under the remaining backward, instead of in step():
        # the same kernel over the same values into the same slots -- the same bits -- and 25 HBM passes less at the serial end.
        # OFF by default: on one GPU the concurrent passes cost the GEMMs they run next to more than the 1.4 ms they save at the end
        # (same box, alternating: 109.5 / 109.6 ms per step with it, 109.1 / 109.0 without; profiles/r04p_*) -- like every other
        # overlap of HBM-bound with MFMA-bound work tried on this chip.  Whether the balance differs when the
        # step's end also waits for the last all-reduce is a measurement a multi-GPU node has to make: bench.py --early-norm.
        self._parts = None
        self.early_norm = False
        # The two streaming passes of step() as NARROW launches: this many fat workgroups (one per CU) instead of grids that cover the
        # chip -- the same bits (of_sumsq_partial_w / of_adamw_clip_w), HBM stays saturated, and the next step's frozen-tower forward on
        # the side stream (train/step.py: next_vision_x) or a late all-reduce finds 64 whole CUs free instead of time-slicing with 4096
        # workgroups.  Same box, alternating, round 6 (profiles/r06zzc_*): 106.27 / 106.24 ms per step wide, 105.90 / 105.81 at 192
        # (160 and 224 the same, 128 = wide, 96 +0.4); without the prefetch a tie; config 4 -0.3 ms, config 5 -0.6 ms.  0 = wide launches.
        self.narrow_cus = 192
        # Fragment-major copies of the gated blocks' to_q / to_out weights (the fused attention branch streams them from L2 straight into
        # MFMA registers, csrc/xattn_fused.hip): re-packed from the fresh bf16 copies by ONE launch per step() instead of two per block
        # and forward.
        self._pack_pairs, self._packed = [], {}
        for blk in [b for b in reducer.module.lang_encoder.gated_cross_attn_layers if b is not None]:
            for lin in (blk.attn.to_q, blk.attn.to_out):
                p = lin.weight
                if reducer._param_bucket.get(p) is not None and p.dim() == 2 and p.shape[0] % 16 == 0 and p.shape[1] % 32 == 0:
                    self._packed[id(p)] = torch.empty(p.numel(), dtype=BF16, device=p.device)
        self.refresh_bf16()
        # Norm taps (round 5; one GPU): the FFN weight gradients of a gated block are 91 % of its bucket and each leaves ONE big-tile
        # GEMM whose epilogue can emit the sum of squares of what it writes (OfGemmArgs.sumsq_out, one partial per 256x256 tile) into
        # slots of the array of_sumsq_finish adds anyway -- the global-norm pass then reads only the rest of the bucket: 3.4 of
        # 3.78 GB less per step at OF-3B.  Valid only when nothing changes the gradient behind the GEMM: no all-reduce (world = 1), the
        # matrices at the END of their bucket, every tapped matrix written by a GEMM that honoured the slots this step
        # (`_of_sumsq_valid`, set by the modules' backward); anything else falls back to the full pass for that bucket.
        self.tap_norm = True
        P = self._ops().SUMSQ_PARTS
        total = 0
        for b in reducer.buckets:
            taps = [p for p in b.get("overwritable", ()) if p.dim() == 2 and p.shape[0] % 256 == 0 and p.shape[1] % 256 == 0
                    and (p.shape[0] // 256) * (p.shape[1] // 256) >= 128]
            n = len(taps)
            if n and all(a is q for a, q in zip(b["params"][-n:], taps)):
                b["taps"], b["tap_from"], b["tap_lo"] = taps, b["offsets"][len(b["params"]) - n], total
                total += sum((p.shape[0] // 256) * (p.shape[1] // 256) for p in taps)
                b["tap_hi"] = total
            else:
                b["taps"] = []
        self._tap_groups = (total + P - 1) // P          # whole groups of P slots in FRONT of the buckets' partials
        self._parts_for(len(reducer.buckets) + 1)     # allocated here, on the construction stream, not inside a side-stream callback
        self._parts[:self._tap_groups * P].zero_()
        self._register_taps()
        model = reducer.module
        for mod in [model.perceiver] + [b for b in model.lang_encoder.gated_cross_attn_layers if b is not None]:
            mod.__dict__["_w_bf16_provider"] = self

    def _register_taps(self):
        """Hand the tapped matrices their slots -- only while step() can use them (one rank, taps on, no early partials): the dW GEMMs
        otherwise run their plain epilogue instead of computing sums nobody reads.  Contract of a tap: between the dW GEMM that
        wrote `_of_sumsq_valid = True` and step() nothing else may write that gradient (a hook that rescales .grad, a second backward
        that adds to it through AccumulateGrad): step() checks the gradient's version counter against the one the backward recorded
        and falls back to the full pass for the bucket when they differ."""
        on = bool(self.__dict__.get("_tap_norm", True) and self._tap_groups and self.reducer.world == 1 and self.reducer.on_bucket_final is None)
        for b in self.reducer.buckets:
            lo = b.get("tap_lo", 0)
            for p in b.get("taps", ()):
                n = (p.shape[0] // 256) * (p.shape[1] // 256)
                if on:
                    p._of_sumsq_slots, p._of_sumsq_valid = self._parts[lo:lo + n], False
                else:
                    p.__dict__.pop("_of_sumsq_slots", None)
                    p._of_sumsq_valid = False
                lo += n

    @property
    def tap_norm(self):
        return self.__dict__.get("_tap_norm", True)

    @tap_norm.setter
    def tap_norm(self, on):
        self.__dict__["_tap_norm"] = bool(on)
        if "_tap_groups" in self.__dict__:
            self._register_taps()

    # ------------------------------------------------------------------ bf16 operand copies for the modules
    def _ops(self):
        if self.ops is None:
            self.ops = Ops.default()
        return self.ops

    def refresh_bf16(self):
        """(Re)build every bf16 copy from the fp32 masters -- after construction, ``load_state_dict`` or any other
        out-of-band parameter write."""
        ops = self._ops()
        for b in self.reducer.buckets:
            ops.to_bf16(b["flat_p"], out=b["flat_bf16"])
            for p, off in zip(b["params"], b["offsets"]):
                n = p.numel()
                self._views[p.data_ptr()] = (b["flat_bf16"][off:off + n].view(p.shape), p._version, p.numel())
        self._pack_pairs = [(self._views[p.data_ptr()][0], self._packed[id(p)]) for b in self.reducer.buckets for p in b["params"]
                            if id(p) in self._packed]
        self._repack()

    def _repack(self):
        if self._pack_pairs:
            self._ops().pack_frag16_batch(self._pack_pairs)

    def packed_view(self, p):
        """fragment-major copy (of_pack_frag16) of parameter ``p``'s current bf16 copy, or None (not a packed matrix / written behind our back)"""
        pk = self._packed.get(id(p))
        return pk if pk is not None and self.bf16_view(p) is not None else None

    def bf16_view(self, p):
        """bf16 copy of parameter ``p`` kept current by step(), or None if ``p`` was written behind our back."""
        ent = self._views.get(p.data_ptr())
        if ent is None or ent[1] != p._version or ent[2] != p.numel():
            return None
        return ent[0]

    def _parts_for(self, nbufs):
        """[tap groups | one group of P partials per bucket | the embedding rows' group]: of_sumsq_finish adds a prefix of it"""
        ops = self._ops()
        need = (getattr(self, "_tap_groups", 0) + nbufs) * ops.SUMSQ_PARTS
        if self._parts is None or self._parts.numel() < need:
            old = self._parts
            self._parts = torch.empty(need, dtype=F32, device=self.reducer.buckets[0]["flat"].device)
            if old is not None:
                self._parts[:old.numel()].copy_(old)
        return self._parts

    @property
    def early_norm(self):
        return self.reducer.on_bucket_final is not None

    @early_norm.setter
    def early_norm(self, on):          # off: the reducer does not touch its side stream for this at all
        self.reducer.on_bucket_final = self._early_partial if on else None
        if "_tap_groups" in self.__dict__:
            self._register_taps()

    @torch.no_grad()
    def _early_partial(self, bi):
        ops, P = self._ops(), self._ops().SUMSQ_PARTS
        parts = self._parts_for(len(self.reducer.buckets) + 1)
        T = self._tap_groups
        ops.sumsq_partial(self.reducer.buckets[bi]["flat"], parts[(T + bi) * P:(T + bi + 1) * P], int(getattr(self, "narrow_cus", 0)))
        self.reducer.buckets[bi]["early_gen"] = self.reducer.generation

    # ------------------------------------------------------------------ optimizer API subset used by train_step
    @torch.no_grad()
    def step(self, closure=None):
        """NOTE for code that reads gradients: after step() the ``.grad`` of the nn.Linear weights is UNDEFINED (it still
        holds the all-reduced SUM of the step just applied, left for the next backward's dW GEMM to overwrite); read
        gradients between backward and step(), or call ``reducer.zero_grad()`` first."""
        assert closure is None, "FlatAdamW does not re-evaluate the model"
        ops = self._ops()
        self.step_count += 1
        for b in self.reducer.buckets:               # a matrix no backward wrote since the last step still holds the
            for p in b.get("overwritable", ()):      # previous step's gradient (it was not cleared): clear it now
                if getattr(p, "_of_grad_fresh", False):
                    p.grad.zero_()
                    p._of_grad_fresh = False
        # GradReducer.finish(average=False) leaves the all-reduced SUM in the buckets; the 1/world is folded into the
        # AdamW pass instead of 25 extra read-modify-write passes over the gradients
        gs = 1.0 / self.reducer.world if getattr(self.reducer, "holds_sum", False) else 1.0
        dev = self.reducer.buckets[0]["flat"].device
        if self._sumsq is None:
            self._sumsq = torch.zeros(1, dtype=F32, device=dev)
        g_rows = None
        sparse = getattr(self.reducer, "sparse", None)
        if sparse is not None:                       # train/sparse_rows.py: the kept rows arrive directly
            if sparse.grad_rows() is not None:
                g_rows = sparse.grad_rows().contiguous()
        elif self.embedding is not None and self.embedding.grad is not None:
            g_rows = self.embedding.grad.index_select(0, self._emb["rows"]).contiguous()
        # global norm without floating-point atomics (of_sumsq_partial / of_sumsq_finish): every rank computes the same
        # bits from the same all-reduced buffers, so the clip coefficient cannot differ between replicas
        # narrow_cus > 0: both streaming passes as narrow launches (that many fat workgroups, one per CU: the same bits) so that
        # work on another stream -- the next step's vision-tower forward, train/step.py -- finds whole CUs free
        nw = int(getattr(self, "narrow_cus", 0))
        bufs = [b["flat"] for b in self.reducer.buckets] + ([g_rows] if g_rows is not None else [])
        P = ops.SUMSQ_PARTS
        parts = self._parts_for(len(self.reducer.buckets) + 1)
        side = getattr(self.reducer, "_stream", None)
        gen = self.reducer.generation
        early = [i for i, b in enumerate(self.reducer.buckets) if b.get("early_gen") == gen and self.early_norm]
        # finish() waits for the COLLECTIVES, not for the of_sumsq kernels queued behind them on the side stream: wait for that stream
        # whenever early partials may have been launched this step -- also when every one of them was voided since (a stale side-stream
        # pass would otherwise write the slots after the main-stream pass below: a wrong, or rank-divergent, clip norm)
        if self.early_norm and side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        T = self._tap_groups
        use_taps = bool(self.tap_norm and T and self.reducer.world == 1 and not getattr(self.reducer, "force_collectives", False)
                        and not self.early_norm and gs == 1.0)
        tapped = 0
        for i, g in enumerate(bufs):
            if i in early:
                continue                             # this step's partial sums of the bucket are in their slots already
            b = self.reducer.buckets[i] if i < len(self.reducer.buckets) else None
            if use_taps and b is not None and b["taps"]:
                if all(getattr(p, "_of_sumsq_valid", False) and getattr(p, "_of_sumsq_version", None) == p.grad._version for p in b["taps"]):
                    g = g[:b["tap_from"]]            # the tapped matrices' sums of squares came with their GEMMs
                    tapped += 1
                else:
                    parts[b["tap_lo"]:b["tap_hi"]].zero_()      # (a matrix no GEMM wrote this step, or not through a tapped launch)
            if g.numel():
                ops.sumsq_partial(g, parts[(T + i) * P:(T + i + 1) * P], nw)
            else:                                    # a bucket made of tapped matrices only: nothing left for the pass
                parts[(T + i) * P:(T + i + 1) * P].zero_()
        for b in self.reducer.buckets:
            b["early_gen"] = None
        self.early_partials_used = len(early)        # (tests, tools)
        self.tapped_buckets = tapped                 # (tests, tools)
        ops.sumsq_finish(parts[(0 if use_taps else T * P):(T + len(bufs)) * P], self._sumsq)
        if T:
            for b in self.reducer.buckets:
                for p in b["taps"]:
                    p._of_sumsq_valid = False
        # Adam's step = the number of updates actually APPLIED, counted on the device: a step skipped for a non-finite norm
        # (the reference `continue`s before optimizer.step(), train_utils.py:161-169) does not advance the bias correction.
        # (step_count, the host's count of step() calls, only seeds the counter and names checkpoints' "step" after a sync.)
        if self._applied is None:
            self._applied = torch.full((1,), self.step_count - 1, dtype=torch.int32, device=dev)
        ops.step_advance(self._sumsq, self._applied)
        for b, lr in ((b, self.param_groups[0 if b["wd"] else 1]["lr"]) for b in self.reducer.buckets):
            # front part: small vectors whose kernels ADD into the gradient -> cleared here; back part: weight matrices
            # the next backward overwrites (their "fresh" mark makes its dW GEMM run with beta = 0): no zero pass, and no
            # read of the old value in the GEMM epilogue -- 2 x 3.5 GB of HBM traffic per step at OF-3B
            k = b.get("overwritable_from", b["flat"].numel())
            for lo, hi, zero in ((0, k, True), (k, b["flat"].numel(), False)):
                if hi > lo:
                    ops.adamw_clip(b["flat_p"][lo:hi], b["flat"][lo:hi], b["m"][lo:hi], b["v"][lo:hi], self._sumsq,
                                   step=self.step_count, lr=lr, betas=self.betas, eps=self.eps, weight_decay=b["wd"],
                                   max_norm=self.max_norm, p_bf16=b["flat_bf16"][lo:hi], zero_grad=zero, grad_scale=gs,
                                   applied=self._applied, max_workgroups=nw)
            for p in b.get("overwritable", ()):
                p._of_grad_fresh = True
        self._repack()
        if g_rows is not None:
            e = self._emb
            p_rows = self.embedding.data.index_select(0, e["rows"]).contiguous()
            ops.adamw_clip(p_rows, g_rows, e["m"], e["v"], self._sumsq, step=self.step_count,
                           lr=self.param_groups[1]["lr"], betas=self.betas, eps=self.eps, weight_decay=0.0,
                           max_norm=self.max_norm, zero_grad=False, grad_scale=gs, applied=self._applied)
            self.embedding.data.index_copy_(0, e["rows"], p_rows)
            self.embedding.grad = None
            if sparse is not None:
                sparse.clear()

    def zero_grad(self, set_to_none=True):
        self.reducer.zero_grad()

    # ------------------------------------------------------------------ checkpoint interchange with torch.optim.AdamW
    def _reference_order(self):
        """The parameter numbering ``torch.optim.AdamW`` has in the reference (train.py:384-408): trainable parameters
        in ``named_parameters()`` order, the ``gated_cross_attn`` ones (weight decay) first, then the rest."""
        from .sparse_rows import is_trainable
        named = [(n, p) for n, p in self.reducer.module.named_parameters()
                 if is_trainable(p) and not getattr(p, "exclude_from_optimizer", False)]
        with_wd = [p for n, p in named if "gated_cross_attn" in n]
        without = [p for n, p in named if "gated_cross_attn" not in n]
        return with_wd, without

    def _moments_of(self, p):
        """(exp_avg, exp_avg_sq) views for a bucketed parameter, or None for the embedding."""
        bi = self.reducer._param_bucket.get(p)
        if bi is None:
            return None
        b = self.reducer.buckets[bi]
        off = b["offsets"][[id(q) for q in b["params"]].index(id(p))]
        n = p.numel()
        return b["m"][off:off + n].view(p.shape), b["v"][off:off + n].view(p.shape)

    def state_dict(self):
        """A ``torch.optim.AdamW`` state dict (what train_utils.py:354 saves), so a run can move between the fused
        step epilogue, ``torch.optim.AdamW`` and the reference's own training script at any checkpoint."""
        with_wd, without = self._reference_order()
        applied = self.applied_steps()
        state, idx = {}, 0
        for p in with_wd + without:
            mom = self._moments_of(p)
            if mom is None:                                          # input embedding: only two rows have moments
                m, v = torch.zeros_like(p.data), torch.zeros_like(p.data)
                m.index_copy_(0, self._emb["rows"], self._emb["m"])
                v.index_copy_(0, self._emb["rows"], self._emb["v"])
            else:
                m, v = mom[0].clone(), mom[1].clone()
            if applied > 0:
                state[idx] = {"step": torch.tensor(float(applied)), "exp_avg": m, "exp_avg_sq": v}
            idx += 1
        groups = []
        start = 0
        for g, members in zip(self.param_groups, (with_wd, without)):
            groups.append({"lr": g["lr"], "betas": tuple(self.betas), "eps": self.eps,
                           "weight_decay": g["weight_decay"], "amsgrad": False, "maximize": False, "foreach": None,
                           "capturable": False, "differentiable": False, "fused": None,
                           "decoupled_weight_decay": True, **{k: v for k, v in g.items()
                                                              if k not in ("lr", "params", "weight_decay")},
                           "params": list(range(start, start + len(members)))})
            start += len(members)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        with_wd, without = self._reference_order()
        params = with_wd + without
        if sum(len(g["params"]) for g in sd["param_groups"]) != len(params):
            raise ValueError("optimizer state dict does not match this model's trainable parameters")
        steps = set()
        for idx, p in enumerate(params):
            st = sd["state"].get(idx)
            mom = self._moments_of(p)
            if st is None:
                continue
            steps.add(int(float(st["step"])))
            m, v = st["exp_avg"].to(p.device, F32), st["exp_avg_sq"].to(p.device, F32)
            if mom is None:
                self._emb["m"].copy_(m.index_select(0, self._emb["rows"]))
                self._emb["v"].copy_(v.index_select(0, self._emb["rows"]))
            else:
                mom[0].copy_(m)
                mom[1].copy_(v)
        if len(steps) > 1:
            raise ValueError(f"parameters at different step counts {sorted(steps)}: not an AdamW run of this model")
        self.step_count = steps.pop() if steps else 0
        self._applied = None          # re-seeded from step_count by the next step()
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            for k, val in saved.items():
                if k not in ("params", "weight_decay", "betas", "eps") and k in g:
                    g[k] = val
            if "initial_lr" in saved:
                g["initial_lr"] = saved["initial_lr"]
        self.refresh_bf16()

    def applied_steps(self):
        """Updates applied so far (host int; synchronises): step() calls minus the ones skipped for a non-finite norm."""
        return self.step_count if self._applied is None else int(self._applied.item())

    def grad_norm(self):
        """Global gradient norm of the last step() (device scalar tensor, pre-clip)."""
        gs = 1.0 / self.reducer.world if getattr(self.reducer, "holds_sum", False) else 1.0
        return self._sumsq.sqrt() * gs
