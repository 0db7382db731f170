"""Which K stages of a big-tile GEMM build go wrong?  A = ones, B = indicator of one 64-deep stage (and the mirror image): every
output must be 64 for every stage.  Found the VALU-write -> asm-MFMA hazard of gemm_w4m.hip in round 3 (an experimental build with
the K loop unrolled by two lost 16 of stage 0's 64 k: profiles/r03y_*, r03ak_*; of_platform.h: of_mfma_acc_guard).
DIAGNOSIS TOOL.      python tools/probes/w4m_unroll_diag.py path/to/libofhip.so"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
lib = ctypes.CDLL(sys.argv[1]); abi.declare(lib, require_all=False)
ops = Ops(lib, lambda: torch.cuda.current_stream().cuda_stream)
M = N = 256
for nd in (4, 5, 6, 7, 8, 9, 12, 16):
    K = 64 * nd
    A = torch.ones(M, K, device="cuda", dtype=torch.bfloat16)
    bad = {}
    for s in range(nd):
        B = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
        B[:, 64 * s:64 * s + 64] = 1
        C = torch.zeros(M, N, device="cuda")
        ops.gemm(A, B, C, epi=abi.EPI_ACC_F32, safe=16)
        torch.cuda.synchronize()
        vals = torch.unique(C).tolist()
        if vals != [64.0]:
            bad[s] = vals[:6]
    # and the mirrored probe: B = ones, A = indicator (which operand is stale)
    bad_a = {}
    Bo = torch.ones(N, K, device="cuda", dtype=torch.bfloat16)
    for s in range(nd):
        A2 = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
        A2[:, 64 * s:64 * s + 64] = 1
        C = torch.zeros(M, N, device="cuda")
        ops.gemm(A2, Bo, C, epi=abi.EPI_ACC_F32, safe=16)
        torch.cuda.synchronize()
        vals = torch.unique(C).tolist()
        if vals != [64.0]:
            bad_a[s] = vals[:6]
    print(json.dumps(dict(stages=nd, wrong_stage_of_B=bad, wrong_stage_of_A=bad_a)), flush=True)
