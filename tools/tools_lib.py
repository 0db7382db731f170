"""Ops bound to tools/libofhip_tools.so: the same sources as the product library compiled with -DOF_TOOLS_BUILD, which
adds the timing ablations / A-B variants of the GEMM kernels (OfGemmArgs.safe = 5, >= 16; several give wrong results by
design).  PROFILING TOOLS ONLY -- the package (open_flamingo_amd.hip.lib) never loads this file."""
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
