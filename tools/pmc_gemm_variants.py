"""Launch one plain GEMM shape through several kernels (for rocprofv3 --pmc passes: tools/gpu_pmc_gemm_variants.sh).

    python tools/pmc_gemm_variants.py [--lib path.so] [--layouts NT,NN,TN]

8192 x 2048 x 8192 (one 256x256 tile per CU, 128 stages), bf16 store / fp32 accumulate epilogue, random operands: the
4-wave kernel on 32x32x16 (safe = 7) and on 16x16x32 MFMAs (safe = 16) and the vendor library (torch.mm).  (profiles/r03g_*: run at
commit dd27936, which still had the half-stage-ring variant of the 4-wave kernel as safe = 16.)  PROFILING TOOL."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tools_lib import routed_ops      # product library; kernel-forcing selectors (safe >= 2) -> tools/libofhip_tools.so

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--layouts", default="NT,TN")
ap.add_argument("--safes", default="7,16")
a = ap.parse_args()
if a.lib:
    lib = ctypes.CDLL(a.lib)
    abi.declare(lib, require_all=False)
    ops = Ops(lib, lambda: torch.cuda.current_stream().cuda_stream)
else:
    ops = routed_ops()
M, N, K = 8192, 2048, 8192
g = torch.Generator(device="cuda").manual_seed(1)
def r(*s): return torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
for lay in a.layouts.split(","):
    ta, tb = {"NT": (False, False), "NN": (False, True), "TN": (True, True)}[lay]
    A = r(K, M) if ta else r(M, K)
    B = r(K, N) if tb else r(N, K)
    f32 = lay == "TN"
    C = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    for safe in [int(x) for x in a.safes.split(",")]:
        for _ in range(3):
            ops.gemm(A, B, C, ta=ta, tb=tb, epi=abi.EPI_ACC_F32 if f32 else abi.EPI_STORE_BF16, safe=safe)
    if not a.lib:
        At = A.t() if ta else A
        Bt = B if tb else B.t()
        for _ in range(3):
            torch.mm(At, Bt)
torch.cuda.synchronize()
