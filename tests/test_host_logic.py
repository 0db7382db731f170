"""Host-side logic of the boundary (no GPU, no kernels): label masking, synthetic batches, state-dict names,
interleave rule, conditioning control flow of Flamingo/FlamingoLMMixin, factory freezing, loud failure on CPU
tensors, C-ABI symbol table."""
import ctypes
import os
import re

import pytest
import torch

from open_flamingo_amd.hip import abi
from open_flamingo_amd.train import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _labels_loop(input_ids, media, eoc, pad):
    """Literal restatement of the reference loops, train_utils.py:127-150."""
    labels = input_ids.clone()
    labels[labels == pad] = -100
    for i in range(labels.shape[0]):
        j = 0
        while j < labels.shape[1] and labels[i][j] != media:
            labels[i][j] = -100
            j += 1
        for e in torch.where(labels[i] == eoc)[0]:
            t = e + 1
            while t < labels.shape[1] and labels[i][t] != media:
                labels[i][t] = -100
                t += 1
    labels[labels == media] = -100
    return labels


def test_label_masking_matches_reference_loops():
    g = torch.Generator().manual_seed(0)
    for trial in range(20):
        ids = torch.randint(0, 12, (4, 40), generator=g)     # small vocab -> many specials; 9=media 10=eoc 11=pad
        got = synthetic.make_labels(ids, 9, 10, 11)
        assert torch.equal(got, _labels_loop(ids, 9, 10, 11)), trial


def test_synthetic_batch_contract():
    info = dict(vocab=1000, media_token_id=1001, eoc_token_id=1000, pad_token_id=1002)
    b = synthetic.make_batch(3, 2, 32, info, "cpu")
    assert b["vision_x"].shape == (3, 2, 1, 3, 224, 224) and b["lang_x"].shape == (3, 32)
    assert (b["lang_x"][:, 0] == 1001).all() and (b["lang_x"][:, 16] == 1001).all() and (b["lang_x"][:, 15] == 1000).all()
    assert ((b["lang_x"] == 1001).sum(-1) == 2).all()


def test_header_and_ctypes_prototypes_agree():
    """Every function declared in include/of_hip.h has a ctypes prototype and vice versa."""
    src = open(os.path.join(ROOT, "include", "of_hip.h")).read()
    declared = set(re.findall(r"\b(?:int|size_t)\s+(of_[a-z0-9_]+)\s*\(", src))
    assert declared == set(abi.PROTOTYPES), declared ^ set(abi.PROTOTYPES)


def test_library_exports_every_symbol():
    """The gfx950 library loads (no GPU needed for dlopen) and exports the whole ABI; built by build()."""
    from open_flamingo_amd.csrc import build
    path = build.build(emu=False)
    lib = ctypes.CDLL(path)
    assert abi.declare(lib, require_all=True) == []
    assert lib.of_abi_version() == abi.OF_ABI_VERSION and lib.of_build_kind() == 1


def test_struct_layouts_match_header():
    """sizeof the two argument structs as the C compiler sees them == ctypes layout."""
    import subprocess
    import tempfile
    code = '#include "of_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu\\n", sizeof(OfGemmArgs), sizeof(OfAttnArgs));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).split()
    assert int(out[0]) == ctypes.sizeof(abi.OfGemmArgs) and int(out[1]) == ctypes.sizeof(abi.OfAttnArgs)


def test_modules_have_reference_state_dict_and_fail_loudly_on_cpu():
    from open_flamingo_amd.src.helpers import GatedCrossAttentionBlock, PerceiverResampler
    from oracle.flamingo_oracle import OracleGatedCrossAttentionBlock, OraclePerceiverResampler
    p, po = PerceiverResampler(dim=64, depth=2, heads=2), OraclePerceiverResampler(dim=64, depth=2, heads=2)
    assert {k: tuple(v.shape) for k, v in p.state_dict().items()} == {k: tuple(v.shape) for k, v in po.state_dict().items()}
    po.load_state_dict(p.state_dict(), strict=True)
    b, bo = GatedCrossAttentionBlock(dim=128, dim_visual=64, heads=2), OracleGatedCrossAttentionBlock(dim=128, dim_visual=64, heads=2)
    b.load_state_dict(bo.state_dict(), strict=True)
    assert list(b.state_dict()) == list(bo.state_dict())
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        b(torch.zeros(1, 4, 128), torch.zeros(1, 1, 64, 64), torch.zeros(1, 4, dtype=torch.bool))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        p(torch.zeros(1, 1, 1, 8, 64))
    GatedCrossAttentionBlock(dim=128, dim_visual=64, dim_head=32)           # any head size up to 128 (zero-padded to a kernel size)
    with pytest.raises(NotImplementedError):
        GatedCrossAttentionBlock(dim=128, dim_visual=64, dim_head=160)


def test_golden_state_dict_names():
    """Key names of the reference modules (captured in the golden fixtures) == ours (checkpoint compatibility)."""
    import numpy as np
    from open_flamingo_amd.src.helpers import GatedCrossAttentionBlock, PerceiverResampler
    z = np.load(os.path.join(ROOT, "tests", "golden", "small_xattn_basic.npz"))
    ref_keys = sorted(k[len("param."):] for k in z.files if k.startswith("param."))
    assert ref_keys == sorted(GatedCrossAttentionBlock(dim=64, dim_visual=64).state_dict())
    z = np.load(os.path.join(ROOT, "tests", "golden", "small_perceiver_embs.npz"))
    ref_keys = sorted(k[len("param."):] for k in z.files if k.startswith("param."))
    ours = PerceiverResampler(dim=64, depth=2, heads=2, num_latents=4, max_num_media=4, max_num_frames=3)
    assert ref_keys == sorted(ours.state_dict())


def test_laion_label_rule_and_single_process_embedding_mask():
    """LAION pass labels (train_utils.py:102-105: only pad and <image> ignored, the trailing <|endofchunk|>/EOS trained on)
    and the single-process form of the embedding-gradient mask (train_utils.py:174-196) in train_step(reducer=None)."""
    from open_flamingo_amd.train import step
    from tests.cpu_model import tiny_cpu_flamingo
    ids = torch.tensor([[9, 3, 4, 10, 5, 11, 11], [9, 1, 2, 3, 10, 6, 11]])   # 9=media 10=eoc 11=pad
    got = synthetic.make_labels_laion(ids, 9, 11)
    want = ids.clone()
    want[want == 11] = -100
    want[want == 9] = -100
    assert torch.equal(got, want) and (got == 10).sum() == 2 and (got[0] == 5).any()
    assert (synthetic.make_labels(ids, 9, 10, 11)[0] == 5).sum() == 0       # the interleaved rule drops what follows <eoc>
    model, info = tiny_cpu_flamingo(seed=0)
    opt = step.build_optimizer(model, lr=1e-3)
    b_laion = synthetic.make_batch(2, 1, 16, info, "cpu", seed=3)
    b_mmc4 = synthetic.make_batch(2, 2, 24, info, "cpu", seed=4)
    emb = model.lang_encoder.get_input_embeddings().weight
    before = emb.detach().clone()
    step.train_step(model, None, opt, b_mmc4, info, batch_laion=b_laion, amp=False)
    moved = ((emb.detach() - before).abs().sum(-1) > 0).nonzero().flatten().tolist()
    assert set(moved) <= {info["media_token_id"], info["eoc_token_id"]} and moved, moved


def test_tuned_vendor_gemm_table_is_inert_without_a_gpu():
    """train/towers.py::use_tuned_vendor_gemms: the committed TunableOp table names gfx950 + the library versions it was
    measured with; on a machine without an AMD GPU the call does nothing (no TunableOp state is touched)."""
    import csv
    from open_flamingo_amd.train import towers
    assert towers.use_tuned_vendor_gemms() == 0
    path = os.path.join(ROOT, "open_flamingo_amd", "train", "tuned", "tunableop_gfx950_of3b_cfg2.csv")
    rows = list(csv.reader(open(path)))
    validators = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert validators["GCN_ARCH_NAME"].startswith("gfx950") and "HIPBLASLT_VERSION" in validators
    assert sum(r[0].startswith("Gemm") for r in rows) >= 10


def test_bf16_twin_registry_semantics():
    """hip/path.py: offer_bf16_twin / take_bf16_twin -- matched by storage address + element count + version counter of the fp32 tensor
    (views included), consumed by the first taker, never served after an in-place change of the fp32 tensor, at most four entries."""
    import torch
    from open_flamingo_amd.hip import path as P
    P.DEFAULT_SCOPE.twins.clear()
    x = torch.randn(6, 8)
    tw = x.to(torch.bfloat16)
    P.offer_bf16_twin(x, tw)
    got = P.take_bf16_twin(x.view(2, 3, 8).reshape(6, 8))             # a view of what was offered
    assert got is not None and got.data_ptr() == tw.data_ptr() and got.shape == (6, 8)
    assert P.take_bf16_twin(x) is None                                 # consumed by the first taker
    P.offer_bf16_twin(x, tw)
    x.add_(1.0)                                                        # the fp32 tensor changed after the offer: the twin is stale
    assert P.take_bf16_twin(x) is None
    P.DEFAULT_SCOPE.twins.clear()
    y = torch.randn(6, 8)
    P.offer_bf16_twin(y, y.to(torch.bfloat16))
    assert P.take_bf16_twin(torch.randn(6, 8)) is None                 # another tensor of the same shape
    assert P.take_bf16_twin(y[:3]) is None                             # a part of it
    assert P.take_bf16_twin(y.to(torch.bfloat16)) is None              # not fp32
    keep = [torch.randn(4, 4) for _ in range(6)]
    for t in keep:
        P.offer_bf16_twin(t, t.to(torch.bfloat16))
    assert len(P.DEFAULT_SCOPE.twins) == 4 and P.take_bf16_twin(keep[0]) is None and P.take_bf16_twin(keep[5]) is not None
    P.DEFAULT_SCOPE.twins.clear()


def test_host_state_is_scoped_to_the_model_instance():
    """the bf16-twin registry and the shared per-forward artefacts belong to a hip/path.py Scope that
    Flamingo.__init__ gives to its whole module tree -- two models in one process never see each other's entries, a module used on
    its own works in the default scope."""
    import torch
    from open_flamingo_amd.hip import path as P
    a, b = torch.nn.Sequential(torch.nn.Linear(2, 2)), torch.nn.Sequential(torch.nn.Linear(2, 2))
    sa, sb = P.adopt(a), P.adopt(b)
    assert sa is not sb and P.scope_of(a[0]) is sa and P.scope_of(b[0]) is sb and P.scope_of(torch.nn.Linear(1, 1)) is P.DEFAULT_SCOPE
    x = torch.randn(4, 8)
    tw = x.to(torch.bfloat16)
    P.offer_bf16_twin(x, tw, sa)
    assert P.take_bf16_twin(x, sb) is None and P.take_bf16_twin(x) is None          # another model / no model: not visible
    assert P.take_bf16_twin(x, sa) is not None and P.take_bf16_twin(x, sa) is None   # its own model: once
    made = []
    key = torch.zeros(3)
    assert sa.shared.get(key, "t", lambda: made.append(1) or "A") == "A"
    assert sb.shared.get(key, "t", lambda: made.append(1) or "B") == "B" and sa.shared.get(key, "t", lambda: "never") == "A"
    assert len(made) == 2
    from tests.cpu_model import tiny_cpu_flamingo
    m1, _ = tiny_cpu_flamingo(seed=0, oracle=False)
    m2, _ = tiny_cpu_flamingo(seed=1, oracle=False)
    s1, s2 = P.scope_of(m1), P.scope_of(m2)
    assert s1 is not s2 and s1 is not P.DEFAULT_SCOPE
    assert all(P.scope_of(m) is s1 for m in m1.modules()) and all(P.scope_of(m) is s2 for m in m2.modules())
