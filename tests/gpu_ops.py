"""TEST INFRASTRUCTURE: the Ops the GPU tests run on.  Everything a product caller can ask for goes to the product library
(open_flamingo_amd/csrc/libofhip.so through Ops.default()); the kernel-FORCING selectors the suite compares kernels with
(OfGemmArgs.safe >= 2: one named kernel instead of of_gemm's own choice) exist in tools/libofhip_tools.so only -- the same sources
compiled with -DOF_TOOLS_BUILD, built on demand -- and a launch that carries one is routed there."""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen, see open_flamingo_amd/hip/lib.py)

from open_flamingo_amd.csrc import build as _build
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops

_forced = None


def forced_kernel_ops():
    global _forced
    if _forced is None:
        path = _build.build(tools=True)          # (objects are cached by mtime: a no-op when the snapshot carries the library)
        lib = ctypes.CDLL(path)
        abi.declare(lib, require_all=True)
        _forced = Ops(lib, lambda: torch.cuda.current_stream().cuda_stream)
    return _forced


class RoutedOps(Ops):
    def __init__(self):
        base = Ops.default()
        super().__init__(base.lib, base._stream_fn)

    def gemm(self, *a, **kw):
        if kw.get("safe", 0) not in (0, 1):
            return forced_kernel_ops().gemm(*a, **kw)
        return super().gemm(*a, **kw)


def routed_ops():
    assert torch.cuda.is_available()
    return RoutedOps()
