"""A GEMM launched back to back on the SAME operands (what every hot-loop probe does) against the same launch cycling through R copies of
(A, B, C) that together exceed the Infinity Cache several times (PyTorch TunableOp's "rotating buffer"; no filler kernel in between):
this library's selection and the vendor library (default heuristics, and the committed TunableOp table if OF_TABLE=1).  HIP events around
blocks of launches, median of the blocks.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import make

if os.environ.get("OF_TABLE") == "1":
    from open_flamingo_amd.train import towers
    print(json.dumps(dict(tunableop_entries=towers.use_tuned_vendor_gemms())), flush=True)
ops = Ops.default()
E = abi
CASES = [("NT up_proj", 8192, 8192, 2048, 0, 0), ("NT down_proj", 8192, 2048, 8192, 0, 0), ("NT Wqkv", 8192, 6144, 2048, 0, 0),
         ("NN dX", 8192, 2048, 8192, 0, 1), ("TN dW", 2048, 8192, 8192, 1, 1)]
R = int(os.environ.get("ROT_COPIES", "8"))


def blocks(fn_of_i, n_sets, per_block=16, nblocks=8):
    for i in range(per_block):
        fn_of_i(i % n_sets)
    torch.cuda.synchronize()
    ts = []
    k = 0
    for b in range(nblocks):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(per_block):
            fn_of_i(k % n_sets)
            k += 1
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / per_block * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


for name, M, N, K, ta, tb in CASES:
    sets = []
    for r in range(R):
        A, B, C, _ = make(M, N, K, ta, tb, E.EPI_STORE_BF16)
        sets.append((A, B, C))
    rec = dict(case=name, MNK=[M, N, K], copies=R, MB_per_copy=round((sets[0][0].numel() + sets[0][1].numel() + sets[0][2].numel()) * 2 / 1e6))

    def ours(i):
        A, B, C = sets[i]
        ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb))

    def vendor(i):
        A, B, C = sets[i]
        torch.mm(A.t() if ta else A, B if tb else B.t(), out=C)
    for lab, fn in (("ours", ours), ("vendor", vendor)):
        rec[lab + " same operands us"] = blocks(fn, 1)
        rec[lab + " rotating us"] = blocks(fn, R)
        rec[lab + " same operands again us"] = blocks(fn, 1)
    print(json.dumps(rec), flush=True)
    del sets
