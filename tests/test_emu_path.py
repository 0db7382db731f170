"""Composed forward/backward of the hot path (open_flamingo_amd.hip.path) on the host SIMT emulator vs the
oracle: checks the kernel schedule, buffer wiring and every hand-derived gradient on CPU."""
import pytest
import torch

from tests import path_checks as PC
from tests.emu import harness as H


@pytest.mark.parametrize("stream_dtype", [torch.float32, torch.bfloat16])
def test_xattn_block_path(stream_dtype):
    errs = PC.check_xattn(H.emu_ops(), "cpu", stream_dtype=stream_dtype)
    print(errs)


def test_xattn_block_path_quirk_rows_and_ge():
    L = 40
    ml = torch.zeros(2, L, dtype=torch.bool)
    ml[0, [1, 5, 9, 30]] = True      # 4 <image> tokens but only T=2 images -> uniform rows
    ml[1, [7]] = True                 # rows before the first image -> zero rows
    PC.check_xattn(H.emu_ops(), "cpu", media_locs=ml, seed=1)
    PC.check_xattn(H.emu_ops(), "cpu", media_locs=ml, only_immediate=False, seed=2)


def test_xattn_zero_gates_identity():
    errs = PC.check_xattn(H.emu_ops(), "cpu", gates=(0.0, 0.0), seed=3, fwd_tol=1e-6)
    assert errs["y"] == 0.0


@pytest.mark.parametrize("stream_dtype", [torch.float32, torch.bfloat16])
def test_perceiver_path(stream_dtype):
    errs = PC.check_perceiver(H.emu_ops(), "cpu", stream_dtype=stream_dtype)
    print(errs)


def test_perceiver_path_with_frame_and_media_time_embs():
    """helpers.py:117-119,123-124: max_num_frames / max_num_media position tables (F=2 frames)."""
    errs = PC.check_perceiver(H.emu_ops(), "cpu", T=3, Fv=32, frames=2, embs=True, seed=4)
    assert "dframe_embs" in errs and "dmedia_time_embs" in errs


def test_paths_accumulate_into_existing_grads():
    """Gradient sinks (path._GradOut): every parameter gradient is added to the pre-existing buffer, none overwritten."""
    PC.check_xattn(H.emu_ops(), "cpu", inplace=True, seed=5)
    PC.check_perceiver(H.emu_ops(), "cpu", T=3, Fv=32, frames=2, embs=True, inplace=True, seed=6)
    # Linear weights marked fresh (left uncleared by the step epilogue) are overwritten, everything else still adds
    PC.check_xattn(H.emu_ops(), "cpu", inplace=True, fresh=True, seed=7)
    PC.check_perceiver(H.emu_ops(), "cpu", T=3, Fv=32, frames=2, embs=True, inplace=True, fresh=True, seed=8)


def test_perceiver_backward_reports_layers_as_they_finish():
    """perceiver_bwd(on_ready=...) names every parameter exactly once, final norm first, then layer by layer from the
    last to the first, latents last -- the order train/reducer.py's per-layer Perceiver buckets are launched in."""
    import torch
    from oracle import flamingo_oracle as O
    from open_flamingo_amd.hip import path
    ops = H.emu_ops()
    m = O.OraclePerceiverResampler(dim=64, depth=3, heads=2, num_latents=16)
    P = {k: v.detach().contiguous() for k, v in m.named_parameters()}
    W = PC.make_bf16_weights(ops, P)
    x = torch.randn(2 * 24, 64)
    kw = dict(N=2, Fv=24, n=16, heads=2, depth=3)
    y, S = path.perceiver_fwd(ops, P, W, x, **kw)
    calls = []
    path.perceiver_bwd(ops, P, W, S, torch.randn_like(y), on_ready=lambda names: calls.append(list(names)), **kw)
    flat = [k for c in calls for k in c]
    assert sorted(flat) == sorted(P) and len(set(flat)) == len(flat)
    assert calls[0] == ["norm.weight", "norm.bias"] and calls[-1] == ["latents"]
    assert [c[0].split(".")[1] for c in calls[1:-1]] == ["2", "1", "0"]
    from open_flamingo_amd.train.reducer import GradReducer
    groups = GradReducer._perceiver_groups(m, "layer")
    names = {id(p): k for k, p in m.named_parameters()}
    order = [[names[id(p)] for p in g] for g in groups]
    assert len(order) == 3 and "norm.weight" in order[0] and order[0][0].startswith("layers.2.") and "latents" in order[-1]


def test_xattn_zero_padded_images():
    """KAT-3 zero-padded images (train/data.py:205-215) on the emulator: see tests/test_gpu_path.py."""
    ml = torch.zeros(2, 40, dtype=torch.bool)
    ml[0, [0, 12]] = True
    ml[1, 0] = True
    PC.check_xattn(H.emu_ops(), "cpu", T=3, media_locs=ml, seed=13, zero_pad=True)
