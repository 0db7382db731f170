"""Ops bound to tools/libofhip_tools.so: the same sources as the product library compiled with -DOF_TOOLS_BUILD, which only
adds of_tools_hold_cus (tools/rehearse_contention.py).  The timing ablations of rounds 1-2 (loop parts switched off, DMA
placements, register staging of the ping-pong kernel) were removed from the kernel sources in round 3 -- their results are in
profiles/r01_* / r02_*.  PROFILING TOOLS ONLY -- the package (open_flamingo_amd.hip.lib) never loads this file."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (must precede the dlopen, see open_flamingo_amd/hip/lib.py)

from open_flamingo_amd.csrc import build as _build
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops


def tools_ops():
    path = _build.TOOLS_LIB
    if not os.path.exists(path):
        path = _build.build(tools=True)
    lib = ctypes.CDLL(path)
    abi.declare(lib, require_all=True)
    return Ops(lib, lambda: torch.cuda.current_stream().cuda_stream)


class RoutedOps(Ops):
    """The product library for everything a product caller can ask for (safe = 0 / 1) and tools/libofhip_tools.so for the kernel-forcing
    selectors (OfGemmArgs.safe >= 2: the product library answers OF_E_ARG to those since round 6)."""

    def __init__(self):
        base = Ops.default()
        super().__init__(base.lib, base._stream_fn)
        self._forced = tools_ops()

    def gemm(self, *a, **kw):
        if kw.get("safe", 0) not in (0, 1):
            return self._forced.gemm(*a, **kw)
        return super().gemm(*a, **kw)


def routed_ops():
    return RoutedOps()
