"""Same-box A/B of two libofhip builds on the step's LayerNorm launches (HIP events, interleaved rounds): forward fp32 -> bf16,
forward with fused residual add, backward without / with dw, db (wave-per-row and workgroup-per-row forms), at d = 2048 and 1024.
Outputs of both builds are compared.  PROFILING TOOL.      python tools/bench_ln_ab.py old.so"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from open_flamingo_amd.hip.ops import Ops
from bench_gemm_ab import load, timed

old, new = load(sys.argv[1]), Ops.default()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
for rows, dim in ((8192, 2048), (16448, 1024), (8192, 2560), (2048, 4096)):
    x = torch.randn(rows, dim, device=dev, generator=g)
    add = torch.randn(rows, dim, device=dev, generator=g).to(torch.bfloat16)
    w, b = torch.randn(dim, device=dev, generator=g), torch.randn(dim, device=dev, generator=g)
    dy = torch.randn(rows, dim, device=dev, generator=g).to(torch.bfloat16)
    resid = torch.randn(rows, dim, device=dev, generator=g)
    res, fns = {}, {}
    for lab, ops in (("old", old), ("new", new)):
        y, st = torch.zeros(rows, dim, device=dev, dtype=torch.bfloat16), torch.zeros(rows, 2, device=dev)
        xs, y2 = torch.zeros_like(x), torch.zeros(rows, dim, device=dev, dtype=torch.bfloat16)
        dx, dxb = torch.zeros_like(x), torch.zeros(rows, dim, device=dev, dtype=torch.bfloat16)
        dx2, dxb2 = torch.zeros_like(x), torch.zeros(rows, dim, device=dev, dtype=torch.bfloat16)
        dw, db = torch.zeros(dim, device=dev), torch.zeros(dim, device=dev)
        f = {"fwd": lambda ops=ops, y=y, st=st: ops.ln_fwd(x, w, b, y, st),
             "fwd_add": lambda ops=ops, xs=xs, y2=y2, st=st: ops.ln_fwd_add(x, add, xs, w, b, y2, st),
             "bwd": lambda ops=ops, st=st, dx=dx, dxb=dxb: ops.ln_bwd(dy, x, st, w, resid=resid, dx=dx, dx_bf16=dxb),
             "bwd_dw": lambda ops=ops, st=st, dx2=dx2, dxb2=dxb2, dw=dw, db=db: ops.ln_bwd(dy, x, st, w, resid=resid, dx=dx2, dx_bf16=dxb2, dw=dw, db=db)}
        for fn in f.values():
            fn()
        torch.cuda.synchronize()
        res[lab] = [t.clone() for t in (y, xs, y2, dx, dxb, dx2, dxb2)]
        for k, fn in f.items():
            fns[lab + "_" + k] = fn
    diff = max(float((a.float() - c.float()).abs().max()) for a, c in zip(res["old"], res["new"]))
    best = {k: 1e9 for k in fns}
    for _ in range(4):
        for k, fn in fns.items():
            best[k] = min(best[k], timed(fn, 10))
    print(json.dumps(dict(rows=rows, dim=dim, max_abs_diff_old_new=diff, **{k + "_us": round(v * 1e3, 1) for k, v in best.items()})), flush=True)
