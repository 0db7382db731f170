"""What do the dirty lines a GEMM leaves in the eight XCD L2s cost the NEXT kernel?  For each build of the library (product = plain
write-back stores in the tiled epilogues; round 4's variants stored the epilogue outputs system-scope / agent-scope write-through /
non-temporal: an experiment, not kept -- profiles/README.md r04q_*) times, with HIP events, a GEMM launch, the kernel that follows it (a LayerNorm forward over the GEMM's output or an ATen
copy of cold buffers) and the pair.      python tools/probes/gemm_store_policy_probe.py wt1=tools/ab/libofhip_wt1.so ...
PROFILING TOOL."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import load, make

libs = {"product": Ops.default()}
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    libs[name] = load(path)
E = abi
dev = "cuda"
NB = 6
src = [torch.randn(8192, 8192, device=dev).to(torch.bfloat16) for _ in range(NB)]
dst = [torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
w, b = torch.randn(2048, device=dev), torch.randn(2048, device=dev)
w8, b8 = torch.randn(8192, device=dev), torch.randn(8192, device=dev)

CASES = [("NN store_bf16 8192x2048x8192 -> LayerNorm(out)", 8192, 2048, 8192, 0, 1, E.EPI_STORE_BF16, "ln"),
         ("NT gelu (two outputs) 8192x8192x2048 -> ATen copy (cold)", 8192, 8192, 2048, 0, 0, E.EPI_GELU, "copy"),
         ("NT gate+fp32 residual 8192x2048x8192 -> LayerNorm(out)", 8192, 2048, 8192, 0, 0, E.EPI_GATE_RESID, "ln"),
         ("TN dW fp32 2048x8192x8192 -> ATen copy (cold)", 2048, 8192, 8192, 1, 1, E.EPI_ACC_F32, "copy"),
         ("NN dgelu_dot 8192x8192x2048 -> ATen copy (cold)", 8192, 8192, 2048, 0, 1, E.EPI_DGELU_DOT, "copy")]
for name, M, N, K, ta, tb, epi, follower in CASES:
    A, B, C, kw = make(M, N, K, ta, tb, epi)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    st = torch.empty(M, 2, device=dev)
    rec = {"case": name}
    samples = {lab: [] for lab in libs}
    for rnd in range(5):                      # interleaved rounds: no build is always first (clock / thermal drift, first touches)
        for lab, ops in libs.items():
            g = lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, **kw)
            if follower == "ln":
                f = lambda k: Ops.default().ln_fwd(C, w if N == 2048 else w8, b if N == 2048 else b8, y, st)
            else:
                f = lambda k: dst[k].copy_(src[k])
            ev = []
            for i in range(10):
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(); g(); e1.record(); f(i % NB); e2.record()
                ev.append((e0, e1, e2))
            torch.cuda.synchronize()
            if rnd:
                samples[lab] += ev[2:]
    med = lambda xs: round(sorted(xs)[len(xs) // 2] * 1e3, 1)
    for lab, ev in samples.items():
        rec[lab] = {"gemm_us": med([a.elapsed_time(b_) for a, b_, c in ev]), "follower_us": med([b_.elapsed_time(c) for a, b_, c in ev]),
                    "pair_us": med([a.elapsed_time(c) for a, b_, c in ev])}
    print(json.dumps(rec), flush=True)
