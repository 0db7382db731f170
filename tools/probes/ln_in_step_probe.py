"""Why does the LayerNorm forward (8192 x 2048 fp32 -> bf16, 100.7 MB) run at 3.4 TB/s inside a train step and at 4.8 TB/s in the
microbench ("why is open")?  Times ONLY the probed launches (one HIP-event pair around each) in these settings:

  alone_same     the microbench: the same input / output buffers every launch (warm in the 256-MB infinity cache)
  alone_rotate   24 distinct buffer sets in turn (2.3 GB: every launch reads from and writes to HBM)
  mm_rotate      a bf16 8192 x 8192 x 2048 matrix product (the MFMA-bound, power-limited neighbour of every LayerNorm of the step)
                 before each launch, rotating buffers
  write_rotate   the input rewritten (an fp32 copy: what the producing GEMM's epilogue does) right before each launch, rotating
  mm_write_rotate   product, then the rewrite, then the LayerNorm

kernels: ln (of_layernorm_fwd), ln_add (of_layernorm_fwd_add: bf16 branch added to the fp32 stream first), cast (ATen's fp32 -> bf16
copy: the same bytes as ln with no row statistics -- what a streaming kernel gets), bwd / bwd_dw (of_layernorm_bwd without / with dw, db).
With a second library (python tools/probes/ln_in_step_probe.py old.so) every libofhip kernel is timed for both builds.

PROFILING TOOL."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from open_flamingo_amd.hip.ops import Ops

libs = {"new": Ops.default()}
if len(sys.argv) > 1:
    from bench_gemm_ab import load
    libs["old"] = load(sys.argv[1])
dev = "cuda"
NB = 24
g = torch.Generator(device=dev).manual_seed(3)
A = torch.randn(8192, 2048, device=dev, generator=g).to(torch.bfloat16)
B = torch.randn(2048, 8192, device=dev, generator=g).to(torch.bfloat16)
C = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
BYTES = {"ln": 6, "ln_add": 2 + 4 + 4 + 2, "cast": 6, "bwd": 2 + 4 + 4 + 4 + 2, "bwd_dw": 2 + 4 + 4 + 4 + 2}      # per element

for rows, dim in ((8192, 2048), (16448, 1024)):
    xs = [torch.randn(rows, dim, device=dev, generator=g) for _ in range(NB)]
    ys = [torch.empty(rows, dim, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    st = [torch.empty(rows, 2, device=dev) for _ in range(NB)]
    dxs = [torch.empty(rows, dim, device=dev) for _ in range(NB)]
    dxb = [torch.empty(rows, dim, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    src = torch.randn(rows, dim, device=dev, generator=g)
    add = torch.randn(rows, dim, device=dev, generator=g).to(torch.bfloat16)
    w, b = torch.randn(dim, device=dev, generator=g), torch.randn(dim, device=dev, generator=g)
    dw, db = torch.zeros(dim, device=dev), torch.zeros(dim, device=dev)

    def run(ops, kernel, mm, write, rotate, n=48):
        ev = []
        for i in range(n + 8):
            k = i % NB if rotate else 0
            if mm:
                torch.mm(A, B, out=C)
            if write:
                xs[k].copy_(src)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            if kernel == "ln":
                ops.ln_fwd(xs[k], w, b, ys[k], st[k])
            elif kernel == "ln_add":
                ops.ln_fwd_add(xs[k], add, dxs[k], w, b, ys[k], st[k])
            elif kernel == "cast":
                ys[k].copy_(xs[k])
            elif kernel == "bwd":           # the frozen towers' form
                ops.ln_bwd(ys[k], xs[k], st[k], w, resid=src, dx=dxs[k], dx_bf16=dxb[k])
            else:                           # the hot path's form (workgroup per row at dim >= 1536)
                ops.ln_bwd(ys[k], xs[k], st[k], w, resid=src, dx=dxs[k], dx_bf16=dxb[k], dw=dw, db=db)
            e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
        t = sorted(s.elapsed_time(e) * 1e3 for s, e in ev[8:])
        return t[len(t) // 2], t[0]

    for ops in libs.values():               # row statistics for the backward launches
        for k in range(NB):
            ops.ln_fwd(xs[k], w, b, ys[k], st[k])
    check = {}
    for lab, ops in libs.items():
        ops.ln_fwd(xs[0], w, b, ys[0], st[0])
        ops.ln_fwd_add(xs[1], add, dxs[1], w, b, ys[1], st[1])
        check[lab] = [t.clone() for t in (ys[0], st[0], ys[1], st[1], dxs[1])]
    if "old" in check:
        print(json.dumps({"rows": rows, "dim": dim, "old_vs_new_bit_identical": all(torch.equal(p, q) for p, q in zip(check["old"], check["new"]))}))
    for name, mm, write, rotate in (("alone_same", 0, 0, 0), ("alone_rotate", 0, 0, 1), ("mm_rotate", 1, 0, 1), ("write_rotate", 0, 1, 1),
                                    ("mm_write_rotate", 1, 1, 1)):
        for kernel in ("ln", "ln_add", "cast", "bwd", "bwd_dw"):
            if kernel in ("bwd", "bwd_dw") and (name.startswith("write") or name.startswith("mm_write") or len(libs) > 1):
                continue
            rec = {"rows": rows, "dim": dim, "setting": name, "kernel": kernel}
            for lab, ops in libs.items():
                if kernel == "cast" and lab == "old":
                    continue
                run(ops, kernel, mm, write, rotate, 8)
                med, best = run(ops, kernel, mm, write, rotate)
                rec[lab + "_median_us"], rec[lab + "_min_us"] = round(med, 1), round(best, 1)
                rec[lab + "_TBps"] = round(rows * dim * BYTES[kernel] / 1e6 / med, 2)
            print(json.dumps(rec), flush=True)
