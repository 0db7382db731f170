"""A/B of the big-tile GEMM kernels on one MI355X, random operands, interleaved rounds in one process:
   pp = 8-wave ping-pong LDS-DMA (safe=4), w4 = 4-wave 128x128-per-wave register staged (safe=6), blaslt = torch.matmul.
Prints one JSON line per (shape, layout, epilogue): median ms and TFLOP/s per arm."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops

ops = Ops.default()
dev = "cuda"
ROUNDS, ITERS = 5, 10


def r(shape, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / ITERS


CASES = [  # (name, M, N, K, ta, tb, epi)
    ("up+gelu        NT", 8192, 8192, 2048, False, False, abi.EPI_GELU),
    ("down+gate_res  NT", 8192, 2048, 8192, False, False, abi.EPI_GATE_RESID),
    ("dA dgelu_dot   NN", 8192, 8192, 2048, False, True, abi.EPI_DGELU_DOT),
    ("dW2 acc_f32    TN", 2048, 8192, 8192, True, True, abi.EPI_ACC_F32),
    ("dU store       NN", 8192, 2048, 8192, False, True, abi.EPI_STORE_BF16),
    ("dW1 acc_f32    TN", 8192, 2048, 8192, True, True, abi.EPI_ACC_F32),
    ("square store   NT", 8192, 8192, 8192, False, False, abi.EPI_STORE_BF16),
    ("square4k store NT", 4096, 4096, 4096, False, False, abi.EPI_STORE_BF16),
    ("kv-grouped     NT", 4096, 24576, 1024, False, False, abi.EPI_STORE_BF16),
]
only = sys.argv[1:]
for name, M, N, K, ta, tb, epi in CASES:
    if only and not any(o in name for o in only):
        continue
    A = r((K, M) if ta else (M, K))
    B = r((K, N) if tb else (N, K), K ** -0.5)
    gate = torch.tensor([0.4], device=dev)
    kw = dict(ta=ta, tb=tb, epi=epi)
    if epi == abi.EPI_GELU:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw["out2"] = torch.empty_like(out)
    elif epi == abi.EPI_GATE_RESID:
        out = torch.empty(M, N, device=dev)
        kw.update(aux=torch.randn(M, N, device=dev), gate=gate)
    elif epi == abi.EPI_DGELU_DOT:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw.update(aux=r((M, N)), gate=gate, dot=torch.zeros(1, device=dev))
    elif epi == abi.EPI_ACC_F32:
        out = torch.zeros(M, N, device=dev)
        kw.update(beta=1.0)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Am = A.t() if ta else A
    Bm = B if tb else B.t()
    arms = {"pp": lambda: ops.gemm(A, B, out, safe=4, **kw), "w4": lambda: ops.gemm(A, B, out, safe=6, **kw),
            "w4dma": lambda: ops.gemm(A, B, out, safe=7, **kw),
            "blaslt": lambda: torch.matmul(Am, Bm)}
    # same bits: both kernels accumulate a stage's four 16-deep k-steps in the same order
    if epi in (abi.EPI_STORE_BF16,):
        o1, o2 = torch.empty_like(out), torch.empty_like(out)
        ops.gemm(A, B, o1, safe=4, **kw)
        ops.gemm(A, B, o2, safe=6, **kw)
        same = bool(torch.equal(o1, o2))
        ops.gemm(A, B, o2, safe=7, **kw)
        same = same and bool(torch.equal(o1, o2))
    else:
        same = None
    for f in arms.values():
        f()
    torch.cuda.synchronize()
    ts = {k: [] for k in arms}
    for _ in range(ROUNDS):
        for k, f in arms.items():
            ts[k].append(timed(f))
    res = {"case": name.strip(), "MNK": [M, N, K], "bit_identical_pp_w4": same}
    for k, v in ts.items():
        v.sort()
        med = v[len(v) // 2]
        res[k + "_ms"] = round(med, 4)
        res[k + "_tflops"] = round(2.0 * M * N * K / med / 1e9, 1)
    print(json.dumps(res), flush=True)
