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
