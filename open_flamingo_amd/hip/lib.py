"""Loader for the gfx950 C-ABI library ``open_flamingo_amd/csrc/libofhip.so``.

There is no fallback: if the library is missing or was not built for gfx950 the import of the hot path fails
loudly.  Build it with ``python -m open_flamingo_amd.csrc.build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: libofhip.so needs libamdhip64.so.7 and has to bind to
# the HIP runtime instance torch has already loaded (torch bundles its own copy under the same SONAME); loading
# libofhip first would pull a second runtime from /opt/rocm and every launch would fail with hipErrorNoDevice.

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# the ONE library the package ever loads: no environment override (A/B tooling under tools/ opens other builds itself)
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libofhip.so")
_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Return the ctypes handle of libofhip.so (cached).  Raises HipLibraryMissing if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: the MI355X hot path has no CPU/PyTorch fallback. "
            "Build it with `python -m open_flamingo_amd.csrc.build` (hipcc --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    abi.declare(lib, require_all=True)
    if lib.of_abi_version() != abi.OF_ABI_VERSION:
        raise HipLibraryMissing(f"{LIB_PATH}: ABI version {lib.of_abi_version()} != {abi.OF_ABI_VERSION}; rebuild")
    if lib.of_build_kind() != 1:
        raise HipLibraryMissing(f"{LIB_PATH} is not a gfx950 device build")
    _lib = lib
    return lib
