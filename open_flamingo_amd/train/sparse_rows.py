"""Gradient of the two trainable embedding rows without the dense (vocab x d) detour  (opt-in).

The reference trains ``lang_encoder.get_input_embeddings()`` but masks its gradient down to the ``<image>`` and
``<|endofchunk|>`` rows before the optimizer sees it (open_flamingo/train/train_utils.py:174-196).  Autograd still
produces the dense gradient first: the embedding lookup's scatter into a zeroed (vocab, d) fp32 matrix, and -- the MPT /
GPT-NeoX heads are tied to the table -- the LM head's weight gradient, a (vocab x d x tokens) GEMM; then the two are
added and all but two rows are thrown away.  At OF-3B cfg-2 that is a 1.7 TFLOP GEMM plus ~2.5 GB of fp32 traffic per
step (rocprofv3, profiles/r01_v4_bench_kernel_stats.md) for 2 x 2048 numbers.

By linearity the kept rows are
    dL/dW[r] = sum over positions with input id == r of dL/d(inputs_embeds)   (lookup part)
             + (dL/dlogits[..., r])^T  hidden                                    (tied-head part)
so the table is frozen for autograd and two identity "taps" (forward hooks on the embedding module and on the LM head)
deliver exactly those sums into a small (rows, d) leaf's ``.grad``.  Forward values are untouched; ``W[rows]`` itself is
updated in place by the optimizer (train/optim.py) from that leaf's gradient.

Opt-in (``enable(model, rows)`` / ``bench.py --sparse-embedding-rows``) until measured on a GPU; requires the fused step
epilogue (a torch optimizer built from ``requires_grad`` parameters would no longer see the table).
"""
import torch


class _TapLookup(torch.autograd.Function):
    """Identity on the looked-up embeddings; backward also accumulates their gradient rows by token id."""

    @staticmethod
    def forward(ctx, emb, ids, rows, leaf):
        ctx.save_for_backward(ids, rows)
        return emb.view_as(emb)

    @staticmethod
    def backward(ctx, g):
        ids, rows = ctx.saved_tensors
        hit = (ids.unsqueeze(-1) == rows).to(torch.float32)                       # (..., R)
        g_rows = hit.reshape(-1, hit.shape[-1]).t() @ g.reshape(-1, g.shape[-1]).float()
        return (g if ctx.needs_input_grad[0] else None), None, None, g_rows


class _TapHead(torch.autograd.Function):
    """Identity on the logits of a head tied to the table; backward also forms the kept rows of the head's weight
    gradient from the matching logit columns (the full gradient passes through untouched, no copy)."""

    @staticmethod
    def forward(ctx, logits, hidden, rows, leaf):
        ctx.save_for_backward(hidden, rows)
        return logits.view_as(logits)

    @staticmethod
    def backward(ctx, g):
        hidden, rows = ctx.saved_tensors
        cols = g.index_select(-1, rows).float()                                   # (..., R): two strided columns
        g_rows = cols.reshape(-1, cols.shape[-1]).t() @ hidden.reshape(-1, hidden.shape[-1]).float()
        return g, None, None, g_rows


class SparseRows:
    """State of the opt-in mode, attached to the model as ``model._of_sparse_rows``."""

    def __init__(self, table, rows):
        self.table = table                                                        # the (vocab, d) nn.Parameter, frozen for autograd
        self.rows = torch.as_tensor(list(rows), device=table.device, dtype=torch.long)
        self.leaf = torch.zeros(len(self.rows), table.shape[1], dtype=torch.float32, device=table.device,
                                requires_grad=True)                               # only its .grad is ever used
        self.handles = []

    def grad_rows(self):
        return self.leaf.grad

    def clear(self):
        self.leaf.grad = None

    def dense_grad(self):
        """The gradient the reference's masking leaves (zeros except the kept rows) -- for tests and tools."""
        g = torch.zeros_like(self.table, dtype=torch.float32)
        if self.leaf.grad is not None:
            g.index_copy_(0, self.rows, self.leaf.grad)
        return g


def enable(model, rows):
    """Freeze the input-embedding table for autograd and install the two taps.  Returns the SparseRows state."""
    lm = model.lang_encoder
    emb_mod = lm.get_input_embeddings()
    table = emb_mod.weight
    kind = _head_kind(lm, emb_mod, table)      # raises before anything is changed
    state = SparseRows(table, rows)
    table.requires_grad_(False)
    table._of_trained_rows = state          # still "trainable" for checkpoints / optimizer-state numbering
    state.handles.append(emb_mod.register_forward_hook(
        lambda mod, inputs, out: _TapLookup.apply(out, inputs[0], state.rows, state.leaf)))
    if kind == "tied_module":                                                     # HF MptForCausalLM / GPT-NeoX: lm_head.weight is wte.weight
        head = lm.get_output_embeddings()
        state.handles.append(head.register_forward_hook(
            lambda mod, inputs, out: _TapHead.apply(out, inputs[0], state.rows, state.leaf)))
    model._of_sparse_rows = state
    return state


def _head_kind(lm, emb_mod, table):
    """Where the logits come from, decided BEFORE anything is frozen:
      "tied_module" -- a distinct nn.Module whose weight IS the table (gets the head tap);
      "untied"      -- a distinct module with its own weight and a config that says the head is not tied (no head part);
    anything else raises: a head that is the embedding module itself (remote-code MPT-7B returns ``wte`` from
    get_output_embeddings(): a hook there would see token ids, not hidden states) or no head module at all
    (mpt-1b-redpajama-200b computes ``F.linear(x, wte.weight)`` inline) would silently lose the tied-head part of the
    kept rows' gradient."""
    head = lm.get_output_embeddings() if hasattr(lm, "get_output_embeddings") else None
    if head is None:
        raise NotImplementedError("sparse_rows: this LM has no output-embedding module (logits are computed inline from the "
                                  "embedding table); the tied-head part of the kept rows' gradient cannot be tapped -- "
                                  "use the dense masked gradient (the default)")
    if head is emb_mod:
        raise NotImplementedError("sparse_rows: get_output_embeddings() returns the input-embedding module itself; its "
                                  "forward hook would see token ids, not hidden states -- use the dense masked gradient")
    w = getattr(head, "weight", None)
    if w is table:
        return "tied_module"
    tied_cfg = bool(getattr(getattr(lm, "config", None), "tie_word_embeddings", True))
    if w is not None and w.data_ptr() != table.data_ptr() and not tied_cfg:
        return "untied"
    raise NotImplementedError("sparse_rows: cannot prove how the LM head relates to the embedding table (weight is not the "
                              "table object and config.tie_word_embeddings is not False) -- use the dense masked gradient")


def disable(model):
    state = getattr(model, "_of_sparse_rows", None)
    if state is None:
        return
    for h in state.handles:
        h.remove()
    state.table.requires_grad_(True)
    del state.table._of_trained_rows
    del model._of_sparse_rows


def is_trainable(p):
    """requires_grad, or the embedding table in sparse-rows mode (frozen for autograd, two rows trained)."""
    return p.requires_grad or getattr(p, "_of_trained_rows", None) is not None
