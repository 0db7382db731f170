"""dq Wq -> LayerNorm backward at BASELINE config 2's shape (8192 rows, d 2048): the ONE launch (csrc/xattn_fused.hip: of_xattn_dq_ln_bwd)
against of_gemm(dq, Wq, NN) + of_layernorm_bwd, HIP events over rotating buffer sets.  PROFILING TOOL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_flamingo_amd.hip.ops import Ops, BF16, F32

ops = Ops.default()
dev = "cuda"
rows, d, NSETS, REPS = 8192, 2048, 6, 6
g = torch.Generator(device=dev).manual_seed(1)
Wq = (torch.randn(512, d, device=dev, generator=g) * d ** -0.5).to(BF16)
WqT_pk = ops.pack_frag16_t(Wq)
gamma = torch.rand(d, device=dev, generator=g) + 0.5
xs = [torch.randn(rows, d, device=dev, generator=g) for _ in range(NSETS)]
dys = [torch.randn(rows, d, device=dev, generator=g) for _ in range(NSETS)]
dqs = [torch.randn(rows, 512, device=dev, generator=g).to(BF16) for _ in range(NSETS)]
st = torch.empty(rows, 2, device=dev)
xn = torch.empty(rows, d, dtype=BF16, device=dev)
ops.ln_fwd(xs[0], gamma, torch.zeros(d, device=dev), xn, st)
dx, dxb = torch.empty(rows, d, device=dev), torch.empty(rows, d, dtype=BF16, device=dev)
dxn = torch.empty(rows, d, dtype=BF16, device=dev)
dw, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)


def fused(i):
    assert ops.xattn_dq_ln_bwd(dqs[i % NSETS], WqT_pk, xs[i % NSETS], st, gamma, dys[i % NSETS], dx, dxb, dw, db)


def separate(i):
    ops.gemm(dqs[i % NSETS], Wq, dxn, tb=True)
    ops.ln_bwd(dxn, xs[i % NSETS], st, gamma, resid=dys[i % NSETS], dx=dx, dx_bf16=dxb, dw=dw, db=db)


def timed(fn):
    ts = []
    for i in range(NSETS * REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[NSETS:])
    return round(ts[len(ts) // 2], 1)


fused(0)
a = dx.clone()
separate(0)
print(json.dumps({"probe": "xattn_dq_ln_bwd", "rows": rows, "d": d, "max_rel_diff_dx": float((a - dx).abs().max() / dx.abs().max()),
                  "fused_us": timed(fused), "gemm_plus_ln_bwd_us": timed(separate), "fused_us_again": timed(fused)}), flush=True)
