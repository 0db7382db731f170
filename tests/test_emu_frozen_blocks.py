"""Frozen-tower blocks as single autograd nodes (open_flamingo_amd/train/frozen_blocks.py) executed on CPU
with the libofhip kernels on the host SIMT emulator (tests/emu -- TEST INFRASTRUCTURE ONLY) against the HF modules' own
eager forward / autograd under bf16 autocast: same logits, same gradient back to the input embeddings."""
import pytest
import torch

from open_flamingo_amd.hip.ops import Ops
from open_flamingo_amd.train import frozen_blocks, towers
from tests.emu import harness as H


@pytest.fixture
def on_emulator(monkeypatch):
    monkeypatch.setattr(Ops, "default", staticmethod(H.emu_ops))
    yield


def _rel(a, b):
    return (a.double() - b.double()).norm().item() / (b.double().norm().item() + 1e-30)


def test_layernorm_fwd_add_matches_torch(on_emulator):
    torch.manual_seed(0)
    ops = Ops.default()
    for dim, xdt, ydt in ((256, torch.float32, torch.bfloat16), (640, torch.float32, torch.float32), (128, torch.bfloat16, torch.bfloat16)):
        rows = 37
        x = torch.randn(rows, dim).to(xdt)
        add = torch.randn(rows, dim).to(torch.bfloat16)
        w, b = torch.randn(dim), torch.randn(dim)
        xsum = torch.empty_like(x)
        y = torch.empty(rows, dim, dtype=ydt)
        st = torch.empty(rows, 2)
        ops.ln_fwd_add(x, add, xsum, w, b, y, st)
        want_sum = (x.float() + add.float()).to(xdt)
        assert torch.equal(xsum, want_sum)
        # statistics are taken over the fp32 sum (before it is rounded to a bf16 stream, like a fused add would)
        s = x.float() + add.float()
        want = torch.nn.functional.layer_norm(s, (dim,), w, b, 1e-5)
        assert torch.allclose(y.float(), want, atol=2e-2 if ydt == torch.bfloat16 else 2e-5, rtol=1e-2 if ydt == torch.bfloat16 else 1e-5)
        assert torch.allclose(st[:, 0], s.mean(-1), atol=1e-5)
        # in place: xsum may alias x
        x2 = x.clone()
        ops.ln_fwd_add(x2, add, x2, w, b, y, st)
        assert torch.equal(x2, want_sum)


def test_layernorm_bwd_without_parameter_gradients_matches_with(on_emulator):
    """The dw/db-free instantiation (frozen towers) returns the same dx / bf16 copy as the reducing one."""
    torch.manual_seed(0)
    ops = Ops.default()
    for dim in (256, 1024, 2048):
        rows = 19
        x = torch.randn(rows, dim)
        w, b = torch.randn(dim), torch.zeros(dim)
        y = torch.empty(rows, dim, dtype=torch.bfloat16)
        st = torch.empty(rows, 2)
        ops.ln_fwd(x, w, b, y, st)
        dy = torch.randn(rows, dim).to(torch.bfloat16)
        resid = torch.randn(rows, dim)
        outs = []
        for red in (False, True):
            dx, dxb = torch.empty(rows, dim), torch.empty(rows, dim, dtype=torch.bfloat16)
            kw = dict(dw=torch.zeros(dim), db=torch.zeros(dim)) if red else {}
            ops.ln_bwd(dy, x, st, w, resid=resid, dx=dx, dx_bf16=dxb, **kw)
            outs.append((dx, dxb))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.layer_norm(xr, (dim,), w, b, 1e-5).backward(dy.float())
        assert torch.allclose(outs[0][0], xr.grad + resid, atol=1e-4, rtol=1e-4)


# (d, heads, routes): head dim 64 and 128; routes = (up + GELU fused, down + residual fused, dGELU fused, plain GEMMs as of_gemm launches)
@pytest.mark.parametrize("d,heads,routes", [(128, 2, (0, 0, 1, 0)), (256, 2, (0, 0, 1, 0)), (256, 2, (0, 0, 0, 0)), (256, 2, (1, 1, 1, 0)),
                                            (256, 2, (0, 0, 1, 1)), (256, 2, (1, 1, 1, 1))])
def test_fused_frozen_mpt_block_matches_hf_eager(on_emulator, d, heads, routes, monkeypatch):
    """routes: which GEMMs of the block run as of_gemm launches (frozen_blocks._MLP_FUSED_* / _MPT_GEMMS_NATIVE).  The product's setting is
    (0, 0, 1, 0), set from same-box A/Bs; the other routes must stay correct for the next A/B."""
    from transformers import MptConfig, MptForCausalLM
    monkeypatch.setattr(frozen_blocks, "_MLP_FUSED_UP", bool(routes[0]))
    monkeypatch.setattr(frozen_blocks, "_MLP_FUSED_DOWN", bool(routes[1]))
    monkeypatch.setattr(frozen_blocks, "_MLP_FUSED_DGELU", bool(routes[2]))
    monkeypatch.setattr(frozen_blocks, "_MPT_GEMMS_NATIVE", bool(routes[3]))
    torch.manual_seed(0)
    lm = MptForCausalLM(MptConfig(d_model=d, n_heads=heads, n_layers=2, vocab_size=128, max_seq_len=64))
    lm.requires_grad_(False)
    for mod in lm.modules():       # what towers.hold_frozen_linears_in_bf16 does to the frozen Linear layers
        if isinstance(mod, torch.nn.Linear) and mod is not lm.get_output_embeddings():
            mod.weight.data = mod.weight.data.to(torch.bfloat16)
    ids = torch.randint(0, 128, (3, 40))
    am = torch.ones(3, 40, dtype=torch.long)
    am[1, 30:] = 0
    am[2, 17:] = 0

    def run():
        emb = lm.get_input_embeddings()(ids).detach().requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = lm(inputs_embeds=emb, attention_mask=am, use_cache=False).logits
        valid = am.bool()[..., None]
        (out.float() * valid).square().mean().backward()
        return out.float().detach() * valid, emb.grad.detach() * valid

    ref_o, ref_g = run()
    assert frozen_blocks.use_fused_frozen_mpt_blocks(lm, allow_cpu=True) == 2
    calls = []
    orig = frozen_blocks._FrozenMptBlockFn.apply
    frozen_blocks._FrozenMptBlockFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        got_o, got_g = run()
    finally:
        frozen_blocks._FrozenMptBlockFn.apply = orig
    assert len(calls) == 2, "the fused path must have been taken by both blocks"
    assert _rel(got_o, ref_o) < 2e-2, _rel(got_o, ref_o)
    assert _rel(got_g, ref_g) < 3e-2, _rel(got_g, ref_g)
    # a call the fused form does not cover inside a forward that got the light-weight mask (output_attentions): the blocks ask
    # the stand-in for HF's full boolean mask and run their own forward -- same logits as the unpatched model
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        out = lm(inputs_embeds=lm.get_input_embeddings()(ids), attention_mask=am, use_cache=False, output_attentions=True)
    assert out.attentions is not None and len(out.attentions) == 2
    assert _rel(out.logits.float() * am.bool()[..., None], ref_o) < 2e-2
    # a call the fused form does not cover (KV cache) goes through the module's own forward
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        out = lm(input_ids=ids[:1, :5], use_cache=True)
    assert out.past_key_values is not None


@pytest.mark.parametrize("parallel,rotary_pct,hs,compact", [(True, 0.25, 80, True), (False, 1.0, 80, True), (True, 1.0, 64, True),
                                                            (False, 1.0, 80, False)])
def test_fused_frozen_neox_block_matches_hf_eager(on_emulator, monkeypatch, parallel, rotary_pct, hs, compact):
    """SURVEY 8f N1 for OF-4B: whole frozen GPT-NeoX layers (RedPajama-INCITE-3B's head size 80: compact heads at the 128-wide
    attention kernels, OfAttnArgs.head_valid -- or, compact = False, the zero-padded copies of rounds 2-5, the other arm of
    tools/ab_neox_compact_heads.py; rotary embedding in one libofhip pass; parallel and sequential residual layouts; biases in the
    GEMMs) as one autograd node each vs the HF modules' eager forward / autograd under autocast(bf16), right padding."""
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    monkeypatch.setattr(frozen_blocks, "_NEOX_COMPACT_HEADS", compact)
    torch.manual_seed(0)
    cfg = GPTNeoXConfig(hidden_size=2 * hs, num_hidden_layers=2, num_attention_heads=2, intermediate_size=8 * hs, vocab_size=128,
                        max_position_embeddings=64, rotary_pct=rotary_pct, use_parallel_residual=parallel,
                        attn_implementation="eager")
    lm = GPTNeoXForCausalLM(cfg)
    lm.requires_grad_(False)
    for mod in lm.modules():
        if isinstance(mod, torch.nn.Linear) and mod is not lm.get_output_embeddings():
            mod.weight.data = mod.weight.data.to(torch.bfloat16)
            mod.bias.data = (torch.randn_like(mod.bias) * 0.1).to(torch.bfloat16)        # HF inits biases to 0: make them count
    ids = torch.randint(0, 128, (3, 40))
    am = torch.ones(3, 40, dtype=torch.long)
    am[1, 30:] = 0
    am[2, 17:] = 0

    def run():
        emb = lm.get_input_embeddings()(ids).detach().requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = lm(inputs_embeds=emb, attention_mask=am, use_cache=False).logits
        valid = am.bool()[..., None]
        (out.float() * valid).square().mean().backward()
        return out.float().detach() * valid, emb.grad.detach() * valid

    ref_o, ref_g = run()
    assert frozen_blocks.use_fused_frozen_neox_blocks(lm, allow_cpu=True) == 2
    calls = []
    orig = frozen_blocks._FrozenNeoXBlockFn.apply
    frozen_blocks._FrozenNeoXBlockFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        got_o, got_g = run()
        assert len(calls) == 2, "the fused path must have been taken by both layers"
        assert _rel(got_o, ref_o) < 2e-2, _rel(got_o, ref_o)
        assert _rel(got_g, ref_g) < 3e-2, _rel(got_g, ref_g)
        calls.clear()
        lm.eval()                      # eval + mask: HF's own layer forward on HF's own mask (left padding must stay right)
        left = am.flip(1)
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            got = lm(input_ids=ids, attention_mask=left, use_cache=False).logits.float() * left.bool()[..., None]
        assert calls == []
    finally:
        frozen_blocks._FrozenNeoXBlockFn.apply = orig
    for mod in lm.modules():
        if hasattr(mod, "_of_eager_forward"):
            mod.forward = mod._of_eager_forward
    lm.config._of_lite_mask = False
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        want = lm(input_ids=ids, attention_mask=left, use_cache=False).logits.float() * left.bool()[..., None]
    assert _rel(got, want) < 1e-6
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):     # KV cache: the module's own forward
        out = lm(input_ids=ids[:1, :5], use_cache=True)
    assert out.past_key_values is not None


def test_left_padded_eval_forward_keeps_hf_masking(on_emulator):
    """the fused blocks collapse the attention mask to a per-sequence key COUNT, which is only right for
    right-padded batches.  The reference's eval wrapper LEFT-pads (eval/models/open_flamingo.py:57): an eval-mode masked
    forward must therefore NOT take the fused path (same logits as the unpatched model on the real positions), unless the
    caller vouches for right padding; training mode keeps the fused path (right padding is the data pipeline's contract)."""
    from transformers import MptConfig, MptForCausalLM
    torch.manual_seed(0)
    lm = MptForCausalLM(MptConfig(d_model=128, n_heads=2, n_layers=2, vocab_size=128, max_seq_len=64))
    lm.requires_grad_(False)
    for mod in lm.modules():
        if isinstance(mod, torch.nn.Linear) and mod is not lm.get_output_embeddings():
            mod.weight.data = mod.weight.data.to(torch.bfloat16)
    lm.eval()
    ids = torch.randint(0, 128, (3, 24))
    left = torch.ones(3, 24, dtype=torch.long)
    left[1, :9] = 0
    left[2, :17] = 0
    right = left.flip(1)
    assert not frozen_blocks.right_padded(left) and frozen_blocks.right_padded(right)

    def logits(am):
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            return lm(input_ids=ids, attention_mask=am, use_cache=False).logits.float() * am.bool()[..., None]

    ref_left, ref_right = logits(left), logits(right)
    calls = []
    orig = frozen_blocks._FrozenMptBlockFn.apply
    frozen_blocks._FrozenMptBlockFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        frozen_blocks.use_fused_frozen_mpt_blocks(lm, allow_cpu=True)
        got = logits(left)
        assert calls == [] and _rel(got, ref_left) < 1e-6, "eval + mask: HF's own block forward on the real mask"
        assert _rel(logits(right), ref_right) < 1e-6 and calls == []
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):     # no mask at all: nothing to get wrong
            lm(input_ids=ids, use_cache=False)
        assert len(calls) == 2
        calls.clear()
        frozen_blocks.use_fused_frozen_mpt_blocks(lm, allow_cpu=True, assume_right_padding=True)
        assert _rel(logits(right), ref_right) < 2e-2 and len(calls) == 2
        calls.clear()
        lm.train()                                                              # training: right padding by contract
        assert _rel(logits(right), ref_right) < 2e-2 and len(calls) == 2
        # ... a contract the debug switch verifies at run time: a left-padded TRAINING batch is refused
        frozen_blocks.CHECK_RIGHT_PADDING = True
        try:
            logits(right)
            with pytest.raises(ValueError, match="not right-padded"):
                logits(left)
        finally:
            frozen_blocks.CHECK_RIGHT_PADDING = False
    finally:
        frozen_blocks._FrozenMptBlockFn.apply = orig


@pytest.mark.parametrize("attention", ["libofhip", "sdpa"])
def test_fused_clip_tower_matches_hf_modules(on_emulator, attention):
    """The frozen CLIP tower's fused forward (one q|k|v GEMM, residual adds inside the LayerNorm passes, last add folded into
    post_layernorm) vs the HF modules under bf16 autocast."""
    torch.manual_seed(0)
    vis = towers.ClipVisualStandIn(width=128, layers=3, heads=2, patch=14, image=56)     # head dim 64, 17 tokens (ragged)
    vis.requires_grad_(False)
    for mod in vis.modules():
        if isinstance(mod, torch.nn.Linear):
            mod.weight.data = mod.weight.data.to(torch.bfloat16)
            mod.bias.data = mod.bias.data.to(torch.bfloat16)
    x = torch.randn(3, 3, 56, 56)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ref_pooled, ref_tok = vis(x)
        vm = getattr(vis.model, "vision_model", vis.model)
        vm._of_allow_cpu = True
        vis.fused = attention
        got_pooled, got_tok = vis(x)
        assert got_tok.dtype == torch.float32 and got_tok.shape == ref_tok.shape
        assert _rel(got_tok, ref_tok) < 2e-2, _rel(got_tok, ref_tok)
        assert _rel(got_pooled, ref_pooled) < 2e-2
        # a tower the fused form does not cover (fp32 weights) runs its modules
        vis.model.float()
        from open_flamingo_amd.train.frozen_blocks import clip_tower_tokens_fused
        assert clip_tower_tokens_fused(vm, x) is None


def test_per_row_position_ids_keep_hf_rotary(on_emulator):
    """the fused GPT-NeoX layer takes one [L][rot] rotary table for the whole batch.  A caller-supplied per-row
    ``position_ids`` (HF then hands down a (B, L, rot) table with differing rows) must not take that path -- it silently applied
    row 0's positions to every row -- while the default arange (a (1, L, rot) table) still does."""
    from transformers import GPTNeoXConfig, GPTNeoXForCausalLM
    torch.manual_seed(0)
    hs = 64
    cfg = GPTNeoXConfig(hidden_size=2 * hs, num_hidden_layers=1, num_attention_heads=2, intermediate_size=4 * hs, vocab_size=64,
                        max_position_embeddings=64, rotary_pct=1.0, use_parallel_residual=False, attn_implementation="eager")
    lm = GPTNeoXForCausalLM(cfg)
    lm.requires_grad_(False)
    for mod in lm.modules():
        if isinstance(mod, torch.nn.Linear) and mod is not lm.get_output_embeddings():
            mod.weight.data = mod.weight.data.to(torch.bfloat16)
            mod.bias.data = mod.bias.data.to(torch.bfloat16)
    ids = torch.randint(0, 64, (2, 24))
    pos = torch.arange(24)[None].repeat(2, 1)
    pos[1] += 5

    def run(**kw):
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            return lm(input_ids=ids, use_cache=False, **kw).logits.float()

    want = run(position_ids=pos)
    lm.train()
    assert frozen_blocks.use_fused_frozen_neox_blocks(lm, allow_cpu=True) == 1
    calls = []
    orig = frozen_blocks._FrozenNeoXBlockFn.apply
    frozen_blocks._FrozenNeoXBlockFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        got = run(position_ids=pos)
        assert calls == [], "per-row positions must stay on HF's rotary embedding"
        assert _rel(got, want) < 1e-6, _rel(got, want)
        run()
        assert len(calls) == 1, "the default positions still take the fused layer"
    finally:
        frozen_blocks._FrozenNeoXBlockFn.apply = orig
