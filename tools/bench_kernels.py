"""Micro-benchmark of the libofhip kernels at OF-3B cfg-2 shapes on one MI355X (HIP events).  Prints one JSON
line per kernel; torch.matmul (hipBLASLt/rocBLAS) is timed beside each GEMM for orientation only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tools_lib import routed_ops      # product library; kernel-forcing selectors (safe >= 2) -> tools/libofhip_tools.so


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def timeit_cold(fn_of_set, nsets, iters=24, warm=None):
    """the same over `nsets` rotating buffer sets (together > the 256-MB infinity cache: what a launch sees inside a train step --
    round 4 found the same-buffer loop 20-40 % flattering for the streaming kernels)"""
    for i in range(warm if warm is not None else nsets):
        fn_of_set(i % nsets)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn_of_set(i % nsets)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ops = routed_ops()
    dev = "cuda"
    out = []
    shapes = [  # (name, M, N, K, ta, tb, epi)
        ("ffn_up   y=xW1^T +gelu", 8192, 8192, 2048, False, False, abi.EPI_GELU),
        ("ffn_down y=bW2^T +gate+res", 8192, 2048, 8192, False, False, abi.EPI_GATE_RESID),
        ("ffn_dh   da=dyW2 dgelu", 8192, 8192, 2048, False, True, abi.EPI_DGELU_DOT),
        ("ffn_du   du=daW1", 8192, 2048, 8192, False, True, abi.EPI_STORE_BF16),
        ("ffn_dW2  dy^T b", 2048, 8192, 8192, True, True, abi.EPI_ACC_F32),
        ("ffn_dW1  da^T u", 8192, 2048, 8192, True, True, abi.EPI_ACC_F32),
        ("to_q", 8192, 512, 2048, False, False, abi.EPI_STORE_BF16),
        ("to_out +gate+res", 8192, 2048, 512, False, False, abi.EPI_GATE_RESID),
        ("to_kv media", 4096, 1024, 1024, False, False, abi.EPI_STORE_BF16),
        ("perceiver to_kv", 20480, 1024, 1024, False, False, abi.EPI_STORE_BF16),
    ]
    for name, M, N, K, ta, tb, epi in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
        B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
        kw = {}
        if epi == abi.EPI_GATE_RESID:
            C = torch.empty(M, N, device=dev)
            kw = dict(aux=torch.randn(M, N, device=dev), gate=torch.tensor([0.5], device=dev))
        elif epi == abi.EPI_ACC_F32:
            C = torch.empty(M, N, device=dev)
        elif epi == abi.EPI_GELU:
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            kw = dict(out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        elif epi == abi.EPI_DGELU_DOT:
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            kw = dict(aux=torch.randn(M, N, device=dev).to(torch.bfloat16), gate=torch.tensor([0.5], device=dev),
                      dot=torch.zeros(1, device=dev))
        else:
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm(A, B, C, ta=ta, tb=tb, epi=epi, **kw))
        ms_v1 = timeit(lambda: ops.gemm(A, B, C, ta=ta, tb=tb, epi=epi, safe=2, **kw))
        At = A.t() if ta else A
        Bt = B if tb else B.t()
        ms_t = timeit(lambda: torch.matmul(At, Bt))
        fl = 2.0 * M * N * K
        out.append(dict(kernel="gemm", name=name, M=M, N=N, K=K, ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1),
                        general128_tflops=round(fl / ms_v1 / 1e9, 1), torch_ms=round(ms_t, 4), torch_tflops=round(fl / ms_t / 1e9, 1)))
        print(json.dumps(out[-1]), flush=True)
    # LayerNorm (HBM bound)
    NS = 6          # rotating sets for the cold figures: 6 x (100 ... 270 MB)
    xs_ = [torch.randn(8192, 2048, device=dev) for _ in range(NS)]
    x = xs_[0]
    w, b = torch.ones(2048, device=dev), torch.zeros(2048, device=dev)
    ys_ = [torch.empty(8192, 2048, device=dev, dtype=torch.bfloat16) for _ in range(NS)]
    y = ys_[0]
    st = torch.empty(8192, 2, device=dev)
    ms = timeit(lambda: ops.ln_fwd(x, w, b, y, st))
    cold = timeit_cold(lambda i: ops.ln_fwd(xs_[i], w, b, ys_[i], st), NS)
    by = 8192 * 2048 * (4 + 2)
    print(json.dumps(dict(kernel="ln_fwd", rows=8192, dim=2048, ms=round(ms, 4), GBps=round(by / ms / 1e6, 1), cold_ms=round(cold, 4),
                          cold_GBps=round(by / cold / 1e6, 1))), flush=True)
    dys_ = [torch.randn(8192, 2048, device=dev).to(torch.bfloat16) for _ in range(NS)]
    dy = dys_[0]
    dxs_ = [torch.empty_like(x) for _ in range(NS)]
    dx = dxs_[0]
    dxbs_ = [torch.empty_like(y) for _ in range(NS)]
    dxb = dxbs_[0]
    dw, db = torch.zeros(2048, device=dev), torch.zeros(2048, device=dev)
    by = 8192 * 2048 * (2 + 4 + 4 + 4 + 2)
    for name, kwf, nbytes in (("ln_bwd", lambda i: dict(resid=xs_[i], dx=dxs_[i], dx_bf16=dxbs_[i], dw=dw, db=db), by),
                              ("ln_bwd_no_dw (frozen towers)", lambda i: dict(resid=xs_[i], dx=dxs_[i], dx_bf16=dxbs_[i]), by),
                              ("ln_bwd_no_dw_no_bf16_copy", lambda i: dict(resid=xs_[i], dx=dxs_[i]), by - 8192 * 2048 * 2)):
        ms = timeit(lambda: ops.ln_bwd(dy, x, st, w, **kwf(0)))
        cold = timeit_cold(lambda i: ops.ln_bwd(dys_[i], xs_[i], st, w, **kwf(i)), NS)
        print(json.dumps(dict(kernel=name, rows=8192, dim=2048, ms=round(ms, 4), GBps=round(nbytes / ms / 1e6, 1), cold_ms=round(cold, 4),
                              cold_GBps=round(nbytes / cold / 1e6, 1))), flush=True)
    sums_ = [torch.empty_like(x) for _ in range(NS)]
    ms = timeit(lambda: ops.ln_fwd_add(x, dy, sums_[0], w, b, y, st))
    cold = timeit_cold(lambda i: ops.ln_fwd_add(xs_[i], dys_[i], sums_[i], w, b, ys_[i], st), NS)
    by = 8192 * 2048 * (4 + 2 + 4 + 2)
    print(json.dumps(dict(kernel="ln_fwd_add", rows=8192, dim=2048, ms=round(ms, 4), GBps=round(by / ms / 1e6, 1), cold_ms=round(cold, 4),
                          cold_GBps=round(by / cold / 1e6, 1))), flush=True)
    # the frozen towers' element-wise passes (8192 x 8192 bf16), warm and cold
    ga_ = [torch.randn(8192, 8192, device=dev).to(torch.bfloat16) for _ in range(4)]
    gb_ = [torch.empty_like(t) for t in ga_]
    for name, fn, nbytes in (("gelu_fwd", lambda i: ops.gelu_fwd(ga_[i], out=gb_[i]), 8192 * 8192 * 4),
                             ("gelu_bwd", lambda i: ops.gelu_bwd(ga_[i], ga_[(i + 1) % 4], out=gb_[i]), 8192 * 8192 * 6)):
        ms = timeit(lambda: fn(0))
        cold = timeit_cold(fn, 4)
        print(json.dumps(dict(kernel=name, elements=8192 * 8192, ms=round(ms, 4), GBps=round(nbytes / ms / 1e6, 1), cold_ms=round(cold, 4),
                              cold_GBps=round(nbytes / cold / 1e6, 1))), flush=True)
    del xs_, ys_, dys_, dxs_, dxbs_, sums_, ga_, gb_
    g = torch.randn(36_700_000, device=dev)
    parts = torch.empty(ops.SUMSQ_PARTS, device=dev)
    pp_, mm_, vv_ = torch.randn_like(g), torch.zeros_like(g), torch.zeros_like(g)
    pb_ = torch.empty(g.numel(), dtype=torch.bfloat16, device=dev)
    acc_ = torch.ones(1, device=dev)
    # (one bucket's four fp32 arrays + the bf16 copy are 660 MB: the same-buffer loop is already cold in the infinity cache;
    #  sumsq over one 147-MB array is not -- rotate it)
    ms = timeit(lambda: ops.adamw_clip(pp_, g, mm_, vv_, acc_, step=3, lr=1e-4, weight_decay=0.1, p_bf16=pb_, zero_grad=False))
    print(json.dumps(dict(kernel="adamw_clip (one gated-block bucket, bf16 copy, no zero)", n=g.numel(), ms=round(ms, 4),
                          GBps=round(g.numel() * 30 / ms / 1e6, 1), cold_ms=round(ms, 4), note="660 MB per launch: cold as it is")), flush=True)
    ms = timeit(lambda: ops.sumsq_partial(g, parts))
    gs_ = [g, pp_, mm_, vv_]
    cold = timeit_cold(lambda i: ops.sumsq_partial(gs_[i], parts), 4)
    print(json.dumps(dict(kernel="sumsq_partial (one gated-block bucket)", n=g.numel(), ms=round(ms, 4), GBps=round(g.numel() * 4 / ms / 1e6, 1),
                          cold_ms=round(cold, 4), cold_GBps=round(g.numel() * 4 / cold / 1e6, 1))), flush=True)
    del g, gs_
    # causal-LM loss at cfg-2: 8192 rows x 50435 logits (odd vocabulary: rows start at 2-byte alignment)
    lg = torch.randn(8192, 50435, device=dev).to(torch.bfloat16)
    lab = torch.randint(0, 50435, (8192,), device=dev)
    lse_, lr_ = torch.empty(8192, device=dev), torch.empty(8192, device=dev)
    ms = timeit(lambda: ops.ce_fwd(lg, lab, lse_, lr_), iters=10)
    print(json.dumps(dict(kernel="ce_fwd", ms=round(ms, 4), GBps=round(lg.numel() * 2 / ms / 1e6, 1))), flush=True)
    dl = torch.empty_like(lg)
    gs = torch.full((1,), 1.0 / 8192, device=dev)
    ms = timeit(lambda: ops.ce_bwd(lg, lab, lse_, gs, dl), iters=10)
    print(json.dumps(dict(kernel="ce_bwd", ms=round(ms, 4), GBps=round(lg.numel() * 4 / ms / 1e6, 1))), flush=True)
    del lg, dl
    # CLIP ViT-L/14 self-attention at cfg-2: 64 images x 16 heads x 257 tokens, head dim 64, fused q|k|v buffer
    Nv, Hv, Sv = 64, 16, 257
    qkv = torch.randn(Nv * Sv, 3 * 1024, device=dev).to(torch.bfloat16)
    ov = torch.empty(Nv * Sv, 1024, device=dev, dtype=torch.bfloat16)
    lsev = torch.empty(Nv, Hv, Sv, device=dev)
    ms = timeit(lambda: ops.attn_fwd(qkv[:, :1024], qkv[:, 1024:2048], qkv[:, 2048:], ov, lsev, batch=Nv, Lq=Sv, Lk=Sv, heads=Hv))
    qv, kv_, vv = (qkv[:, i * 1024:(i + 1) * 1024].view(Nv, Sv, Hv, 64).transpose(1, 2) for i in range(3))
    ms_t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qv, kv_, vv))
    ms_tiled = timeit(lambda: ops.attn_fwd(qkv[:, :1024], qkv[:, 1024:2048], qkv[:, 2048:], ov, lsev, batch=Nv, Lq=Sv, Lk=Sv, heads=Hv, safe=2))
    print(json.dumps(dict(kernel="vit_attn_fwd", ms=round(ms, 4), tiled_ms=round(ms_tiled, 4), torch_sdpa_ms=round(ms_t, 4))), flush=True)
    # attention cores
    B_, L, T, n, H = 32, 256, 2, 64, 8
    q = torch.randn(B_ * L, H * 64, device=dev).to(torch.bfloat16)
    kv = torch.randn(B_ * T * n, 2 * H * 64, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    lse = torch.empty(B_, H, L, device=dev)
    ml = torch.zeros(B_, L, dtype=torch.uint8, device=dev)
    ml[:, 0] = 1
    ml[:, L // 2] = 1
    tt = torch.empty(B_, L, dtype=torch.int32, device=dev)
    ops.text_time(ml, tt, L, False)
    kw = dict(batch=B_, Lq=L, Lk=T * n, heads=H, text_time=tt, n_per_media=n, T_img=T)
    ms = timeit(lambda: ops.attn_fwd(q, kv[:, :512], kv[:, 512:], o, lse, **kw))
    ms_tiled = timeit(lambda: ops.attn_fwd(q, kv[:, :512], kv[:, 512:], o, lse, safe=2, **kw))
    print(json.dumps(dict(kernel="xattn_core_fwd", ms=round(ms, 4), tiled_ms=round(ms_tiled, 4), gflop_restricted=round(4 * B_ * H * L * 64 * 64 / 1e9, 2))), flush=True)
    do = torch.randn_like(q)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    delta = torch.empty(B_, H, L, device=dev)
    ms = timeit(lambda: ops.attn_bwd(q, kv[:, :512], kv[:, 512:], o, lse, do, dq, dkv[:, :512], dkv[:, 512:], delta, **kw))
    print(json.dumps(dict(kernel="xattn_core_bwd", ms=round(ms, 4))), flush=True)
    N = 64
    q = torch.randn(N * 64, 512, device=dev).to(torch.bfloat16)
    kv = torch.randn(N * 320, 1024, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    lse = torch.empty(N, H, 64, device=dev)
    kw = dict(batch=N, Lq=64, Lk=320, heads=H)
    ms = timeit(lambda: ops.attn_fwd(q, kv[:, :512], kv[:, 512:], o, lse, **kw))
    ms_tiled = timeit(lambda: ops.attn_fwd(q, kv[:, :512], kv[:, 512:], o, lse, safe=2, **kw))
    print(json.dumps(dict(kernel="perceiver_core_fwd", ms=round(ms, 4), tiled_ms=round(ms_tiled, 4), gflop=round(4 * N * H * 64 * 320 * 64 / 1e9, 2))), flush=True)
    do = torch.randn_like(q)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    delta = torch.empty(N, H, 64, device=dev)
    ms = timeit(lambda: ops.attn_bwd(q, kv[:, :512], kv[:, 512:], o, lse, do, dq, dkv[:, :512], dkv[:, 512:], delta, **kw))
    print(json.dumps(dict(kernel="perceiver_core_bwd", ms=round(ms, 4))), flush=True)


    # frozen-MPT self-attention shape: B=32, 16 heads, head dim 128, L=256, causal + ALiBi, fused qkv layout
    Bm, Hm, Lm, dh = 32, 16, 256, 128
    d = Hm * dh
    qkv = torch.randn(Bm * Lm, 3 * d, device=dev).to(torch.bfloat16)
    o = torch.empty(Bm * Lm, d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(Bm, Hm, Lm, device=dev)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / Hm) for i in range(Hm)], device=dev)
    kw = dict(batch=Bm, Lq=Lm, Lk=Lm, heads=Hm, scale=dh ** -0.5, head_dim=dh, causal=True, alibi_slopes=slopes)
    ms = timeit(lambda: ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, **kw))
    gf = 4.0 * Bm * Hm * Lm * Lm * dh / 2 / 1e9
    ms_tiled = timeit(lambda: ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, safe=2, **kw))
    print(json.dumps(dict(kernel="mpt_causal_alibi_attn_fwd", ms=round(ms, 4), tiled_ms=round(ms_tiled, 4), tflops=round(gf / ms, 1),
                          GBps=round(4 * Bm * Lm * d * 2 / ms / 1e6, 1))), flush=True)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(Bm, Hm, Lm, device=dev)
    ms = timeit(lambda: ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, do, dqkv[:, :d], dqkv[:, d:2 * d],
                                     dqkv[:, 2 * d:], delta, **kw))
    print(json.dumps(dict(kernel="mpt_causal_alibi_attn_bwd", ms=round(ms, 4), tflops=round(2.5 * gf / ms, 1))), flush=True)


if __name__ == "__main__":
    main()
