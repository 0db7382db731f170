"""Time the same big-tile GEMM launches through several builds of libofhip (HIP events, interleaved rounds, random operands).

    python tools/bench_gemm_libs.py --libs product,tools/ab/libofhip_abl_novm.so,... [--safe 7]

One JSON line per (layout, shape): us per build.  Used for timing ablations (builds that skip a wait or a barrier give WRONG
results by design: only their time is read).  PROFILING TOOL: loads libraries by path with ctypes; the package never does."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import make, timed, load

ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True)
ap.add_argument("--safe", type=int, default=7)
ap.add_argument("--pads", default="0", help="comma-separated leading-dimension paddings (elements) of both operands: channel-camping probe")
a = ap.parse_args()
libs = {}
for p in a.libs.split(","):
    libs[os.path.basename(p).replace("libofhip_", "").replace(".so", "")] = Ops.default() if p == "product" else load(p)
E = abi
for lay, ta, tb, epi in (("NT", 0, 0, E.EPI_STORE_BF16), ("NN", 0, 1, E.EPI_STORE_BF16), ("TN", 1, 1, E.EPI_ACC_F32)):
    for (M, N, K) in ((8192, 2048, 8192), (8192, 8192, 2048)):
        A, B, C, kw = make(M, N, K, ta, tb, epi)
        fns = {}
        for pad in [int(x) for x in a.pads.split(",")]:
            Ap, Bp = A, B
            if pad:      # same values behind a padded row pitch
                Ap = torch.empty(A.shape[0], A.shape[1] + pad, device="cuda", dtype=A.dtype)[:, :A.shape[1]]
                Bp = torch.empty(B.shape[0], B.shape[1] + pad, device="cuda", dtype=B.dtype)[:, :B.shape[1]]
                Ap.copy_(A)
                Bp.copy_(B)
            for k, o in libs.items():
                fns[k + (("_pad%d" % pad) if pad else "")] = (lambda o=o, Ap=Ap, Bp=Bp: o.gemm(Ap, Bp, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=a.safe, **kw))
        best = {k: 1e9 for k in fns}
        for fn in fns.values():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for _ in range(4):
            for k, fn in fns.items():
                best[k] = min(best[k], timed(fn, 10))
        print(json.dumps(dict(layout=lay, MNK=[M, N, K], **{k + "_us": round(v * 1e3, 1) for k, v in best.items()})), flush=True)
