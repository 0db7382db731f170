"""Generate the golden fixtures in this directory from the REAL reference implementation.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports ``/root/reference/open_flamingo/src/helpers.py`` by file path (with a 1-line
``einops_exts.rearrange_many`` shim -- that package is not installable offline),
feeds it seeded weights/inputs, and stores inputs + outputs + autograd gradients as ``.npz``.
The fixtures are what pins ``oracle/flamingo_oracle.py`` to the reference; nothing on the GPU box
reads /root/reference.

Files written:
  small_perceiver.npz        tiny PerceiverResampler (weights stored), fwd + all grads
  small_perceiver_embs.npz   same with frame_embs / media_time_embs (F=2)
  small_xattn_<case>.npz     tiny GatedCrossAttentionBlock, one file per mask case (KAT-3)
  full_perceiver.npz         OF-3B-sized Perceiver (dim 1024, 6 layers): weights rebuilt from
                             oracle.seeded_state(seed), only output/grad summaries stored
  full_xattn.npz             OF-3B-sized block (d=2048): same idea
  tiny_flamingo.npz          whole reference Flamingo (tiny towers): loss, gradients, generate(), cached-media logits
  checkpoint_keys.json       what the reference's checkpoint filter keeps + its AdamW parameter order
  dh64_xattn_<case>.npz      (--round4) GatedCrossAttentionBlock at dim_head 64, cached-media branch: fp64 answers + the
                             reference's own autocast(bf16) run
"""

import importlib.util
import os
import sys
import types

import numpy as np
import torch
from einops import rearrange

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.flamingo_oracle import seeded_state  # noqa: E402

shim = types.ModuleType("einops_exts")
shim.rearrange_many = lambda tensors, pattern, **kw: tuple(rearrange(t, pattern, **kw) for t in tensors)
sys.modules["einops_exts"] = shim
spec = importlib.util.spec_from_file_location("ref_helpers", "/root/reference/open_flamingo/src/helpers.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def shapes_of(mod):
    return {k: tuple(v.shape) for k, v in mod.state_dict().items()}


def load_seeded(mod, seed, dtype):
    st = seeded_state(shapes_of(mod), seed, dtype)
    mod.to(dtype)
    mod.load_state_dict(st, strict=True)
    return st


def grads_of(mod):
    return {"grad." + k: v.grad.detach().numpy() for k, v in mod.named_parameters()}


def rnd(shape, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64).to(dtype)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def small_perceiver(with_embs):
    dt = torch.float64
    kw = dict(dim=32, depth=2, dim_head=8, heads=2, num_latents=4)
    Fr = 1
    if with_embs:
        kw.update(max_num_media=4, max_num_frames=3)
        Fr = 2
    m = ref.PerceiverResampler(**kw)
    st = load_seeded(m, 11, dt)
    x = rnd((2, 3, Fr, 6, 32), 12, dt).requires_grad_(True)
    w = rnd((2, 3, 4, 32), 13, dt)
    y = m(x)
    (y * w).sum().backward()
    save("small_perceiver_embs.npz" if with_embs else "small_perceiver.npz",
         **{"param." + k: v for k, v in st.items()}, x=x, w=w, y=y, **{"grad.x": x.grad}, **grads_of(m),
         heads=2)


XATTN_CASES = {
    # name: (media_locations rows as 0/1 lists, T_img, only_immediate, use_cached, T_txt override)
    "basic": ([[1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0]], 3, True, False, None),
    "before_first_image": ([[0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], 3, True, False, None),
    "image_last_and_consecutive": ([[1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1], [0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0]], 3, True, False, None),
    "more_image_tokens_than_images": ([[1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 0], [1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0]], 3, True, False, None),
    "single_image_laion": ([[1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], 1, True, False, None),
    "attend_all_previous": ([[1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0]], 3, False, False, None),
    "cached_media_decode": ([[1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], 3, True, True, 2),
    "no_media_locations_cached": (None, 3, True, True, 5),
    # train/data.py:205-215 pads a sample's image list with all-zero images up to MAX_NUM_IMAGES: fewer <image> tokens than
    # media slots, the unused slots hold zeros (the flag in position 5 zeroes media[b, t] for t >= number of <image> tokens)
    "zero_padded_images": ([[1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], 3, True, False, None, True),
}


def small_xattn(case, gates=None):
    dt = torch.float64
    locs, T_img, only_imm, cached, t_txt = XATTN_CASES[case][:5]
    zero_pad = len(XATTN_CASES[case]) > 5 and XATTN_CASES[case][5]
    m = ref.GatedCrossAttentionBlock(dim=32, dim_visual=24, dim_head=8, heads=2,
                                     only_attend_immediate_media=only_imm)
    st = load_seeded(m, 21, dt)
    tag = case
    if gates is not None:
        with torch.no_grad():
            m.attn_gate.fill_(gates)
            m.ff_gate.fill_(gates)
        st["attn_gate"] = m.attn_gate.detach().clone()
        st["ff_gate"] = m.ff_gate.detach().clone()
        tag = f"{case}_gate{gates:g}"
    L = 12 if t_txt is None else t_txt
    x = rnd((2, L, 32), 22, dt).requires_grad_(True)
    media = rnd((2, T_img, 4, 24), 23, dt)
    ml = None if locs is None else torch.tensor(locs, dtype=torch.bool)
    if zero_pad:
        for b in range(2):
            media[b, int(ml[b].sum()):] = 0
    media.requires_grad_(True)
    w = rnd((2, L, 32), 24, dt)
    y = m(x, media, media_locations=ml, use_cached_media=cached)
    (y * w).sum().backward()
    save(f"small_xattn_{tag}.npz", **{"param." + k: v for k, v in st.items()}, x=x, media=media, w=w, y=y,
         media_locations=(np.zeros((0,), dtype=bool) if ml is None else ml.numpy()),
         has_media_locations=int(ml is not None), only_immediate=int(only_imm), use_cached=int(cached),
         heads=2, **{"grad.x": x.grad, "grad.media": media.grad}, **grads_of(m))


DH64_CASES = {
    # the two cached-media cases of XATTN_CASES + one cumsum case, at dim_head = 64 (the head size the HIP attention kernels exist
    # for: the dim_head-8 files above can only be replayed through the oracle).  VERDICT r3 weak #1.
    "cached_media_decode": ([[1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], 3, True, True, 2),
    "cached_media_decode_attend_all": ([[1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], 3, False, True, 3),
    "no_media_locations_cached": (None, 3, True, True, 5),
    "basic": ([[1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0]], 3, True, False, None),
}


def dh64_xattn(case):
    """GatedCrossAttentionBlock(dim=64, dim_visual=32, dim_head=64, heads=2), n = 16 latents per image: the REAL reference in fp64
    (= the known answer) and under torch.autocast(bfloat16) in fp32 (= the reference's own amp error, the yardstick of SURVEY 8c).
    Weights are oracle.seeded_state(shapes, 41) -- rebuilt by the test, not stored.  The upstream gradient is w + y (detached):
    keeps the two scalar gate gradients free of cancellation (tests/path_checks.py: conditioned_upstream)."""
    locs, T_img, only_imm, cached, t_txt = DH64_CASES[case]
    kw = dict(dim=64, dim_visual=32, dim_head=64, heads=2, only_attend_immediate_media=only_imm)
    L = 12 if t_txt is None else t_txt
    ml = None if locs is None else torch.tensor(locs, dtype=torch.bool)
    out = {}
    w_eff = None
    for tag, dt, amp in (("", torch.float64, False), ("amp.", torch.float32, True)):
        m = ref.GatedCrossAttentionBlock(**kw)
        load_seeded(m, 41, dt)
        x = rnd((2, L, 64), 42, dt).requires_grad_(True)
        media = rnd((2, T_img, 16, 32), 43, dt).requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            y = m(x, media, media_locations=ml, use_cached_media=cached)
        if w_eff is None:
            w_eff = (rnd((2, L, 64), 44, dt) + y.detach()).double()
        (y.to(dt) * w_eff.to(dt)).sum().backward()
        g = {"grad.x": x.grad, "grad.media": media.grad, **{k: torch.from_numpy(v) for k, v in grads_of(m).items()}}
        out[tag + "y"] = y.detach().float()
        if not amp:
            out.update({k: v.float() for k, v in g.items()})
            g64 = g
        else:           # of the autocast run only the forward is a yardstick (judge_8c); its gradient errors are kept as numbers
            for k, v in g.items():
                out["amp.rel_l2." + k] = float((v.double() - g64[k]).norm() / (g64[k].norm() + 1e-30))
    save(f"dh64_xattn_{case}.npz", w=w_eff.float(), media_locations=(np.zeros((0,), dtype=bool) if ml is None else ml.numpy()),
         has_media_locations=int(ml is not None), only_immediate=int(only_imm), use_cached=int(cached), heads=2, T_img=T_img,
         n_latents=16, L=L, seed_params=41, seed_x=42, seed_media=43, **out)


def summarize(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 64).long()
    return np.concatenate([t[idx].numpy(), [t.sum().item(), t.abs().sum().item(), (t * t).sum().item()]])


def full_perceiver():
    dt = torch.float32
    torch.manual_seed(0)
    m = ref.PerceiverResampler(dim=1024)
    load_seeded(m, 31, dt)
    x = rnd((1, 2, 1, 256, 1024), 32, dt)
    w = rnd((1, 2, 64, 1024), 33, dt)
    y = m(x)
    (y * w).sum().backward()
    out = {"y.summary": summarize(y), "y.head": y[0, :, :4, :16].detach().numpy()}
    for k, v in m.named_parameters():
        out["gradsum." + k] = summarize(v.grad)
    save("full_perceiver.npz", seed_params=31, seed_x=32, seed_w=33, **out)


def full_perceiver_b2t3():
    """KAT-1 second shape of SURVEY 8c: x (2, 3, 1, 256, 1024)."""
    dt = torch.float32
    m = ref.PerceiverResampler(dim=1024)
    load_seeded(m, 31, dt)
    x = rnd((2, 3, 1, 256, 1024), 34, dt)
    w = rnd((2, 3, 64, 1024), 35, dt)
    y = m(x)
    (y * w).sum().backward()
    out = {"y.summary": summarize(y), "y.head": y[1, :, :4, :16].detach().numpy()}
    for k, v in m.named_parameters():
        out["gradsum." + k] = summarize(v.grad)
    save("full_perceiver_b2t3.npz", seed_params=31, seed_x=34, seed_w=35, **out)


def full_xattn():
    dt = torch.float32
    m = ref.GatedCrossAttentionBlock(dim=2048, dim_visual=1024)
    load_seeded(m, 41, dt)
    L = 32
    x = rnd((1, L, 2048), 42, dt).requires_grad_(True)
    media = rnd((1, 2, 64, 1024), 43, dt).requires_grad_(True)
    w = rnd((1, L, 2048), 44, dt)
    ml = torch.zeros(1, L, dtype=torch.bool)
    ml[0, 3] = True
    ml[0, 17] = True
    y = m(x, media, media_locations=ml)
    (y * w).sum().backward()
    out = {"y.summary": summarize(y), "y.head": y[0, :8, :16].detach().numpy(),
           "gradsum.x": summarize(x.grad), "gradsum.media": summarize(media.grad)}
    for k, v in m.named_parameters():
        out["gradsum." + k] = summarize(v.grad)
    save("full_xattn.npz", seed_params=41, seed_x=42, seed_media=43, seed_w=44,
         media_positions=np.array([3, 17]), L=L, **out)


def tiny_flamingo():
    """End-to-end pin of the boundary: the REAL reference Flamingo/FlamingoLMMixin/
    FlamingoLayer (open_clip stubbed at import time, HF tiny MPT + CLIP stand-in towers) loaded with the state_dict
    of our model (strict=True, so every key name matches), then loss, gradients and greedy generate()."""
    sys.modules.setdefault("open_clip", types.ModuleType("open_clip"))
    sys.path.insert(0, "/root/reference")
    from open_flamingo.src.flamingo import Flamingo as RefFlamingo
    from open_flamingo.src.flamingo_lm import FlamingoLMMixin as RefMixin
    from open_flamingo.src.utils import extend_instance as ref_extend
    from open_flamingo_amd.train import step, synthetic, towers
    from tests.cpu_model import tiny_cpu_flamingo

    mine, info = tiny_cpu_flamingo(seed=0)
    torch.manual_seed(123)
    vision = towers.VisionStandIn(width=64, layers=2, heads=2, patch=14, image=224)
    lm, attr = towers.build_lang_encoder("OF-tiny")
    ref_extend(lm, RefMixin)
    lm.set_decoder_layers_attr_name(attr)
    ref = RefFlamingo(vision, lm, info["eoc_token_id"], info["media_token_id"], vis_dim=64,
                      cross_attn_every_n_layers=info["every"])
    missing = ref.load_state_dict(mine.state_dict(), strict=True)
    print("strict load ok:", missing)
    ref.requires_grad_(False)
    ref.perceiver.requires_grad_(True)
    ref.lang_encoder.gated_cross_attn_layers.requires_grad_(True)
    ref.lang_encoder.get_input_embeddings().requires_grad_(True)
    batch = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5)
    labels = synthetic.make_labels(batch["lang_x"], info["media_token_id"], info["eoc_token_id"], info["pad_token_id"])
    ref.train()
    out = ref(vision_x=batch["vision_x"], lang_x=batch["lang_x"], attention_mask=batch["attention_mask"], labels=labels)
    out[0].backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.state_dict(keep_vars=True).items()
             if getattr(p, "grad", None) is not None and "old_decoder_blocks" not in k and ".transformer.blocks." not in k}
    keep = ["perceiver.latents", "perceiver.layers.0.0.to_kv.weight", "perceiver.norm.weight",
            "lang_encoder.gated_cross_attn_layers.1.attn_gate", "lang_encoder.gated_cross_attn_layers.1.ff_gate",
            "lang_encoder.gated_cross_attn_layers.3.attn.to_q.weight", "lang_encoder.gated_cross_attn_layers.3.ff.3.weight",
            "lang_encoder.gated_cross_attn_layers.1.attn.norm.bias"]
    ref.eval()
    with torch.no_grad():
        gen = ref.generate(batch["vision_x"][:1], batch["lang_x"][:1, :8], attention_mask=batch["attention_mask"][:1, :8],
                           max_new_tokens=6, do_sample=False)
        # cached-media scoring flow (eval/models/open_flamingo.py:155-313)
        ref.cache_media(input_ids=batch["lang_x"][:, :12], vision_x=batch["vision_x"])
        cached_logits = ref(vision_x=None, lang_x=batch["lang_x"][:, 12:16], attention_mask=batch["attention_mask"][:, 12:16],
                            clear_conditioned_layers=False).logits
        ref.uncache_media()
    save("tiny_flamingo.npz", loss=out[0].detach(), logits_head=out.logits[:, :4, :32].detach(), generated=gen,
         cached_logits_head=cached_logits[:, :, :32],
         state_dict_keys=np.array(sorted(ref.state_dict().keys())),
         **{"grad." + k: grads[k] for k in keep},
         **{"gradnorm." + k: v.norm() for k, v in grads.items() if "wte" not in k})


def tiny_flamingo_generate_margins():
    """Greedy generate() of the REAL reference on the tiny model with the per-step logit margins (top-1 minus top-2): a
    bf16 implementation must reproduce every token until the first step whose margin is within bf16 noise of a tie."""
    sys.modules.setdefault("open_clip", types.ModuleType("open_clip"))
    sys.path.insert(0, "/root/reference")
    from open_flamingo.src.flamingo import Flamingo as RefFlamingo
    from open_flamingo.src.flamingo_lm import FlamingoLMMixin as RefMixin
    from open_flamingo.src.utils import extend_instance as ref_extend
    from open_flamingo_amd.train import synthetic, towers
    from tests.cpu_model import tiny_cpu_flamingo

    mine, info = tiny_cpu_flamingo(seed=0)
    torch.manual_seed(123)
    vision = towers.VisionStandIn(width=64, layers=2, heads=2, patch=14, image=224)
    lm, attr = towers.build_lang_encoder("OF-tiny")
    ref_extend(lm, RefMixin)
    lm.set_decoder_layers_attr_name(attr)
    refm = RefFlamingo(vision, lm, info["eoc_token_id"], info["media_token_id"], vis_dim=64,
                       cross_attn_every_n_layers=info["every"])
    refm.load_state_dict(mine.state_dict(), strict=True)
    refm.eval()
    batch = synthetic.make_batch(2, 2, 24, info, "cpu", seed=5)
    toks, margins, scales = [], [], []
    with torch.no_grad():
        for prompt_len in (8, 13):
            ids = batch["lang_x"][:1, :prompt_len]
            for _ in range(10):         # greedy decoding step by step through the reference's forward
                out = refm(vision_x=batch["vision_x"][:1], lang_x=ids, attention_mask=torch.ones_like(ids))
                last = out.logits[0, -1].double()
                top = last.topk(2)
                toks.append(int(top.indices[0]))
                margins.append(float(top.values[0] - top.values[1]))
                scales.append(float(last.std()))
                ids = torch.cat([ids, top.indices[:1].view(1, 1)], dim=1)
            toks.append(-1)
            margins.append(0.0)
            scales.append(0.0)
    save("tiny_flamingo_generate.npz", tokens=np.array(toks), margins=np.array(margins), logit_std=np.array(scales),
         prompt_lens=np.array([8, 13]), steps=10)


def checkpoint_keys():
    """the key set the reference's ``filter_state_dict_to_trainable`` (train_utils.py:299-333) leaves of
    the tiny reference Flamingo, with the LM input embeddings trainable (the default) and frozen
    (``freeze_lm_embeddings``), and the parameter order of its AdamW groups (train.py:384-408)."""
    import json
    sys.modules.setdefault("open_clip", types.ModuleType("open_clip"))
    sys.path.insert(0, "/root/reference")
    from open_flamingo.src.flamingo import Flamingo as RefFlamingo
    from open_flamingo.src.flamingo_lm import FlamingoLMMixin as RefMixin
    from open_flamingo.src.utils import extend_instance as ref_extend
    from open_flamingo_amd.train import towers
    from tests.cpu_model import tiny_cpu_flamingo

    _, info = tiny_cpu_flamingo(seed=0)        # (imports transformers/accelerate, which probe for a real wandb)
    sys.modules["wandb"] = types.ModuleType("wandb")          # train_utils imports it at module level; not installed
    from open_flamingo.train.train_utils import filter_state_dict_to_trainable
    del sys.modules["wandb"]
    out = {}
    for tag, train_emb in (("embeddings_trainable", True), ("embeddings_frozen", False)):
        torch.manual_seed(123)
        vision = towers.VisionStandIn(width=64, layers=2, heads=2, patch=14, image=224)
        lm, attr = towers.build_lang_encoder("OF-tiny")
        ref_extend(lm, RefMixin)
        lm.set_decoder_layers_attr_name(attr)
        ref = RefFlamingo(vision, lm, info["eoc_token_id"], info["media_token_id"], vis_dim=64,
                          cross_attn_every_n_layers=info["every"])
        ref.requires_grad_(False)                      # factory.py:104-114
        ref.perceiver.requires_grad_(True)
        ref.lang_encoder.gated_cross_attn_layers.requires_grad_(True)
        if train_emb:
            ref.lang_encoder.get_input_embeddings().requires_grad_(True)
        kept = filter_state_dict_to_trainable(ref, ref.state_dict())
        named = [(n, p) for n, p in ref.named_parameters() if p.requires_grad]
        out[tag] = dict(checkpoint_keys=sorted(kept.keys()),
                        adamw_with_wd=[n for n, _ in named if "gated_cross_attn" in n],
                        adamw_without_wd=[n for n, _ in named if "gated_cross_attn" not in n])
        print(tag, len(kept), "keys;", len(named), "optimizer params")
    with open(os.path.join(HERE, "checkpoint_keys.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    if "--round2" in sys.argv:          # fixtures added in round 2 (the earlier files stay byte-identical)
        torch.set_num_threads(8)
        small_xattn("zero_padded_images")
        full_perceiver_b2t3()
        tiny_flamingo_generate_margins()
        sys.exit(0)
    if "--round4" in sys.argv:          # dim_head-64 cached-media fixtures (the earlier files stay byte-identical)
        torch.set_num_threads(8)
        for c in DH64_CASES:
            dh64_xattn(c)
        sys.exit(0)
    if "--only-checkpoint-keys" in sys.argv:
        checkpoint_keys()
        sys.exit(0)
    tiny_flamingo()
    checkpoint_keys()
    sys.exit(0) if "--only-flamingo" in sys.argv else None
    torch.set_num_threads(8)
    small_perceiver(False)
    small_perceiver(True)
    for c in XATTN_CASES:
        small_xattn(c)
    small_xattn("basic", gates=0.0)
    full_perceiver()
    full_xattn()
