"""of_gemm's own selection (safe = 0 -> the half-tile kernel of csrc/gemm_w4h.hip for the *_DOT launches over >= 1024 big tiles) against the
256x256 kernel (safe = 16) on the *_DOT launch shapes of every BASELINE model family: same box, interleaved.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools_lib import routed_ops      # product library; kernel-forcing selectors (safe >= 2) -> tools/libofhip_tools.so
from bench_gemm_ab import make, timed

ops = routed_ops()
E = abi
CASES = [("OF-3B ffn_dh", 8192, 8192, 2048), ("OF-4B ffn_dh", 8192, 10240, 2560), ("OF-9B L256 ffn_dh", 2048, 16384, 4096),
         ("OF-9B L2048 ffn_dh", 16384, 16384, 4096), ("OF-3B LAION+MMC4 rows 10240", 10240, 8192, 2048)]
for name, M, N, K in CASES:
    for epi in (E.EPI_DGELU_DOT, E.EPI_SCALE_DOT):
        A, B, C, kw = make(M, N, K, 0, 1, epi)
        arms = {"w4m256": 16, "w4h256x128": 18, "auto": 0}
        best = {k: 1e9 for k in arms}
        for k, sf in arms.items():
            for _ in range(3):
                ops.gemm(A, B, C, tb=True, epi=epi, safe=sf, **kw)
        torch.cuda.synchronize()
        for _ in range(4):
            for k, sf in arms.items():
                best[k] = min(best[k], timed(lambda: ops.gemm(A, B, C, tb=True, epi=epi, safe=sf, **kw), 10))
        rec = dict(case=name, epi="dgelu_dot" if epi == E.EPI_DGELU_DOT else "scale_dot", MNK=[M, N, K], tiles256=(M // 256) * (N // 256),
                   auto_label=Ops.kernel_label(M, N, K, False, True, epi))
        for k, ms in best.items():
            rec[k + "_us"] = round(ms * 1e3, 1)
        rec["w4h_vs_w4m"] = round(best["w4h256x128"] / best["w4m256"], 4)
        print(json.dumps(rec), flush=True)
        del A, B, C, kw
