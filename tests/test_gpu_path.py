"""GPU parity of the composed hot path (kernel schedules of open_flamingo_amd.hip.path) vs the oracle, through the
C ABI on a real MI355X.  Small shapes exercise ragged tiles; the OF-3B shapes (d=2048, Dv=1024, 8 heads, n=64,
L=256, v=256) are BASELINE.json's model dimensions at a reduced batch so the CPU oracle finishes in seconds."""
import pytest
import torch

from tests import path_checks as PC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from open_flamingo_amd.hip.ops import Ops
    return Ops.default()


@pytest.mark.parametrize("stream_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("safe", [0, 1])
def test_xattn_small(ops, stream_dtype, safe):
    PC.check_xattn(ops, "cuda", stream_dtype=stream_dtype, safe=safe)


def test_xattn_quirks(ops):
    L = 40
    ml = torch.zeros(2, L, dtype=torch.bool)
    ml[0, [1, 5, 9, 30]] = True
    ml[1, [7]] = True
    PC.check_xattn(ops, "cuda", media_locs=ml, seed=1)
    PC.check_xattn(ops, "cuda", media_locs=ml, only_immediate=False, seed=2)
    errs = PC.check_xattn(ops, "cuda", gates=(0.0, 0.0), seed=3, fwd_tol=1e-6)
    assert errs["y"] == 0.0


def test_xattn_of3b_dims(ops):
    errs = PC.check_xattn(ops, "cuda", B=2, L=256, T=2, n=64, heads=8, d=2048, Dv=1024, seed=5)
    print({k: f"{v:.1e}" for k, v in errs.items()})


@pytest.mark.parametrize("stream_dtype", [torch.float32, torch.bfloat16])
def test_perceiver_small(ops, stream_dtype):
    PC.check_perceiver(ops, "cuda", stream_dtype=stream_dtype)


def test_perceiver_of3b_dims(ops):
    errs = PC.check_perceiver(ops, "cuda", b=1, T=2, Fv=256, n=64, heads=8, D=1024, depth=6, seed=6)
    print({k: f"{v:.1e}" for k, v in errs.items()})


def test_perceiver_frame_and_media_time_embs(ops):
    errs = PC.check_perceiver(ops, "cuda", T=3, Fv=32, frames=2, embs=True, seed=4)
    assert "dframe_embs" in errs and "dmedia_time_embs" in errs


@pytest.mark.parametrize("d,heads_lm", [(2560, 8), (4096, 8)])
def test_xattn_of4b_of9b_dims(ops, d, heads_lm):
    """BASELINE configs 4 / 5: RedPajama-3B (d=2560) and MPT-7B (d=4096) hidden sizes (LayerNorm CPL=5 / CPL=8 kernels,
    GEMM shapes of those families) at a reduced batch."""
    errs = PC.check_xattn(ops, "cuda", B=2, L=128, T=2, n=64, heads=8, d=d, Dv=1024, seed=9)
    print({k: f"{v:.1e}" for k, v in errs.items()})
