"""Same-box A/B of two libofhip builds on the GEMM launches of a train step (HIP events, interleaved rounds, random operands).

    python tools/bench_gemm_ab.py [old.so] [--family OF-3B|OF-4B|OF-9B|OF-9B-L2048] > out.jsonl

`old.so` defaults to tools/ab/libofhip_r02.so (round 2's closing build, built from git by hand: see profiles/README.md); the
new arm is the product library.  One JSON line per (shape, layout, epilogue): ms and TFLOP/s of both arms, which kernel the new
arm's selection lands in, and -- for shapes both the big-tile and the 128x128 LDS-DMA kernel accept -- the forced alternatives.
PROFILING TOOL: loads libraries by path with ctypes; the package never does that."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tools_lib import routed_ops      # product library; kernel-forcing selectors (safe >= 2) -> tools/libofhip_tools.so

HERE = os.path.dirname(os.path.abspath(__file__))


def load(path):
    lib = ctypes.CDLL(path)
    abi.declare(lib, require_all=False)
    return Ops(lib, lambda: torch.cuda.current_stream().cuda_stream)


def family_shapes(fam):
    d, rows, Nm = {"OF-3B": (2048, 8192, 4096), "OF-4B": (2560, 8192, 4096), "OF-9B": (4096, 2048, 2560),
                   "OF-9B-L2048": (4096, 16384, 2560)}[fam]
    E = abi
    s = [("ffn_up+gelu", rows, 4 * d, d, 0, 0, E.EPI_GELU), ("ffn_down+gate+res", rows, d, 4 * d, 0, 0, E.EPI_GATE_RESID),
         ("ffn_dh dgelu_dot", rows, 4 * d, d, 0, 1, E.EPI_DGELU_DOT), ("ffn_du", rows, d, 4 * d, 0, 1, E.EPI_STORE_BF16),
         ("ffn_dW2", d, 4 * d, rows, 1, 1, E.EPI_ACC_F32), ("ffn_dW1", 4 * d, d, rows, 1, 1, E.EPI_ACC_F32),
         ("to_q", rows, 512, d, 0, 0, E.EPI_STORE_BF16), ("to_out+gate+res", rows, d, 512, 0, 0, E.EPI_GATE_RESID),
         ("to_out dX scale_dot", rows, 512, d, 0, 1, E.EPI_SCALE_DOT), ("to_q dX", rows, d, 512, 0, 1, E.EPI_STORE_BF16),
         ("to_out dW", d, 512, rows, 1, 1, E.EPI_ACC_F32), ("to_q dW", 512, d, rows, 1, 1, E.EPI_ACC_F32),
         ("media to_kv dW", 1024, 1024, Nm, 1, 1, E.EPI_ACC_F32)]
    if fam == "OF-3B":      # the Perceiver's launches (N = 64 media items: 4096 latent rows, 20480 key rows)
        s += [("perc ffn_up+gelu", 4096, 4096, 1024, 0, 0, E.EPI_GELU), ("perc ffn_down+res", 4096, 1024, 4096, 0, 0, E.EPI_GATE_RESID),
              ("perc ffn_dh dgelu", 4096, 4096, 1024, 0, 1, E.EPI_DGELU_DOT), ("perc ffn_du", 4096, 1024, 4096, 0, 1, E.EPI_STORE_BF16),
              ("perc ffn_dW", 4096, 1024, 4096, 1, 1, E.EPI_ACC_F32), ("perc to_kv", 20480, 1024, 1024, 0, 0, E.EPI_STORE_BF16),
              ("perc to_kv dX", 20480, 1024, 1024, 0, 1, E.EPI_STORE_BF16), ("perc to_kv dW", 1024, 1024, 20480, 1, 1, E.EPI_ACC_F32),
              ("perc to_q", 4096, 512, 1024, 0, 0, E.EPI_STORE_BF16), ("perc to_out+res", 4096, 1024, 512, 0, 0, E.EPI_GATE_RESID),
              ("perc to_q dW", 512, 1024, 4096, 1, 1, E.EPI_ACC_F32)]
    return s


def make(M, N, K, ta, tb, epi, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g).to(torch.bfloat16)
    B = (torch.randn((K, N) if tb else (N, K), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    kw = {}
    if epi == abi.EPI_GATE_RESID:
        C = torch.empty(M, N, device=dev)
        kw = dict(aux=torch.randn(M, N, device=dev, generator=g), gate=torch.tensor([0.5], device=dev))
    elif epi == abi.EPI_ACC_F32:
        C = torch.empty(M, N, device=dev)
    elif epi == abi.EPI_GELU:
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    elif epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT):
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(aux=torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16), gate=torch.tensor([0.5], device=dev),
                  dot=torch.zeros(1, device=dev))
    else:
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    return A, B, C, kw


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def ladder(ops):
    """Where a K = 2048 FFN launch's time goes beyond its main loop: the same GEMM with increasingly expensive epilogues."""
    E = abi
    M = N = 8192
    K = 2048
    for ta, tb, steps in ((0, 0, (("store_bf16", E.EPI_STORE_BF16, {}), ("gelu (one output)", E.EPI_GELU, {"no_out2": True}),
                                  ("gelu + pre-activation", E.EPI_GELU, {}), ("acc_f32 beta0", E.EPI_ACC_F32, {}))),
                          (0, 1, (("store_bf16", E.EPI_STORE_BF16, {}), ("scale_dot (aux + dot)", E.EPI_SCALE_DOT, {}),
                                  ("dgelu_dot", E.EPI_DGELU_DOT, {}), ("dgelu_dot no dot", E.EPI_DGELU_DOT, {"no_dot": True})))):
        for name, epi, opt in steps:
            A, B, C, kw = make(M, N, K, ta, tb, epi)
            if opt.get("no_out2"):
                kw.pop("out2")
            if opt.get("no_dot"):
                kw.pop("dot")
            rec = dict(ladder="NT NN".split()[tb], epilogue=name, MNK=[M, N, K])
            for label, safe in (("w4dma256", 7), ("pp256", 4)):
                fn = lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=safe, **kw)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                ms = min(timed(fn, 10) for _ in range(4))
                rec[label + "_ms"] = round(ms, 4)
                rec[label + "_tflops"] = round(2.0 * M * N * K / ms / 1e9, 1)
            print(json.dumps(rec), flush=True)
    # K sweep at M = N = 8192, plain store: per-tile fixed cost = intercept / 4 tiles per CU
    for K in (512, 1024, 2048, 4096, 8192):
        A, B, C, kw = make(M, N, K, 0, 0, E.EPI_STORE_BF16)
        fn = lambda: ops.gemm(A, B, C, epi=E.EPI_STORE_BF16, safe=7)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        print(json.dumps(dict(ksweep="NT store_bf16 w4dma256", MNK=[M, N, K], ms=round(min(timed(fn, 10) for _ in range(4)), 4))), flush=True)


def ksweep(ops):
    """us per 1024 of K (slope) and per-launch intercept of the two 256x256 kernels: 8192 x 8192, one tile round = 4 per CU"""
    E = abi
    M = N = 8192
    for ta, tb, epi in ((0, 0, E.EPI_STORE_BF16), (0, 1, E.EPI_STORE_BF16), (1, 1, E.EPI_ACC_F32)):
        for K in (1024, 2048, 4096, 8192):
            A, B, C, kw = make(M, N, K, ta, tb, epi)
            rec = dict(ksweep="NT NN ?? TN".split()[ta * 2 + tb], MNK=[M, N, K])
            fns = {"w4dma256": lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=7, **kw),
                   "w4m256": lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=16, **kw)}
            best = {k: 1e9 for k in fns}
            for fn in fns.values():
                for _ in range(3):
                    fn()
            torch.cuda.synchronize()
            for _ in range(4):
                for k, fn in fns.items():
                    best[k] = min(best[k], timed(fn, 10))
            for k, ms in best.items():
                rec[k + "_ms"] = round(ms, 4)
                rec[k + "_tflops"] = round(2.0 * M * N * K / ms / 1e9, 1)
            print(json.dumps(rec), flush=True)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("old", nargs="?", default=os.path.join(HERE, "ab", "libofhip_r02.so"))
    ap.add_argument("--family", default="OF-3B")
    ap.add_argument("--ladder", action="store_true", help="epilogue ladder on the K = 2048 FFN shapes instead of the family table")
    ap.add_argument("--only-big", action="store_true", help="only shapes with >= 128 big tiles")
    ap.add_argument("--builtin", default=os.path.join(HERE, "ab", "libofhip_builtin_dma.so"),
                    help="same sources as the product library built with -DOF_DMA_VIA_BUILTIN (tools/build_ab_variant.sh)")
    ap.add_argument("--arms", default="", help="comma-separated subset of the arms to time (default: all)")
    ap.add_argument("--ksweep", action="store_true", help="K sweep at M = N = 8192 of the 4-wave kernel on 32x32x16 and on 16x16x32 MFMAs, three layouts")
    a = ap.parse_args()
    fam = a.family
    if a.ksweep:
        return ksweep(routed_ops())
    old = load(a.old)
    builtin = load(a.builtin) if os.path.exists(a.builtin) else None
    new = routed_ops()
    if a.ladder:
        return ladder(new)
    for name, M, N, K, ta, tb, epi in family_shapes(fam):
        if a.only_big and not (M % 256 == 0 and N % 256 == 0 and (M // 256) * (N // 256) >= 128):
            continue
        A, B, C, kw = make(M, N, K, ta, tb, epi)
        arms = {"old": lambda: old.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, **kw),
                "new": lambda: new.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, **kw)}
        if builtin is not None:
            arms["builtin_dma"] = lambda: builtin.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, **kw)
        big_ok = M % 256 == 0 and N % 256 == 0 and K % 64 == 0
        mid_ok = M % 128 == 0 and N % 128 == 0 and K % 64 == 0
        if big_ok and mid_ok:     # both tilings possible: force each on the new build
            arms["new_mid128"] = lambda: new.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=5, **kw)
            arms["new_pp256"] = lambda: new.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=4, **kw)
            arms["new_w4dma256"] = lambda: new.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=7, **kw)
            arms["new_w4m256"] = lambda: new.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=16, **kw)      # 16x16x32 MFMAs
            if builtin is not None:
                arms["builtin_w4dma256"] = lambda: builtin.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=7, **kw)
        if a.arms:
            arms = {k: v for k, v in arms.items() if k in a.arms.split(",")}
        best = {k: 1e9 for k in arms}
        for k, fn in arms.items():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for _ in range(4):                      # interleaved rounds: box drift hits every arm alike
            for k, fn in arms.items():
                best[k] = min(best[k], timed(fn, 10))
        fl = 2.0 * M * N * K
        if "old" in arms and "new" in arms:      # same operands through both libraries: the results must agree
            arms["old"]()
            c_old = C.float().clone()
            arms["new"]()
            diff_old_new = float((C.float() - c_old).abs().max())
        else:
            diff_old_new = None
        rec = dict(family=fam, name=name, MNK=[M, N, K], layout="NT NN ?? TN".split()[ta * 2 + tb], epi=epi,
                   tiles256=(M // 256) * (N // 256) if big_ok else None, tiles128=(M // 128) * (N // 128) if mid_ok else None)
        for k, ms in best.items():
            rec[k + "_ms"] = round(ms, 4)
            rec[k + "_tflops"] = round(fl / ms / 1e9, 1)
        rec["max_abs_diff_old_new"] = diff_old_new
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
