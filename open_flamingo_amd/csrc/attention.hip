// Windowed flash-style attention core (head dim 64) for the OpenFlamingo hot path, gfx950.
//
// Serves PerceiverAttention (open_flamingo/src/helpers.py:55-64, no mask) and MaskedCrossAttention
// (helpers.py:192-231).  The reference materialises sim (B,h,L,T*n) and a boolean mask; here each query
// row carries a key window [lo,hi) + a "uniform" flag derived from text_time (helpers.py:196-229):
//     only_immediate: tt==0 -> empty window (row output 0); 1<=tt<=T -> keys of media tt-1;
//                     tt>T  -> all keys masked with -finfo.max => softmax uniform over all T*n keys
//     otherwise ('ge'): tt==0 -> uniform over all keys; tt>=1 -> keys [0, min(tt,T)*n)
// Scores, probabilities and the mask never touch HBM.
// The same kernels also serve causal self-attention with ALiBi (head dim 64 or 128; SURVEY.md 8f N1, the frozen
// MPT blocks): causal = 1 gives query i the window [0, i + 1 + Lk - Lq), alibi_slopes[h] adds slope * (j - i) to the
// scaled score (the relative form of MPT's bias: softmax is shift invariant per row).
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16, all matrices via of_mfma(A-rows, B-cols)):
//   forward / dq pass (one wave = 16 query rows, key blocks of 64 staged in LDS):
//     S^T tile  = K(16 keys x 64 d) . Q^T        -> lane holds query l&15, keys 4g+r : softmax row
//                                                    statistics reduce over 16 registers + 2 shuffles
//     O^T      += V^T(16 d x 32 keys) . P^T       -> P fragment is the lane's own registers (no
//                                                    cross-lane movement), V^T fragment comes from the
//                                                    LDS transpose read (ds_read_b64_tr_b16)
//     dP^T tile = V . dO^T ; dQ^T += K^T . dS^T   (same shapes, K^T via transpose read)
//   dk/dv pass (one wave = 16 keys, query tiles of 64 staged in LDS):
//     S tile    = Q(16 q x 64 d) . K^T            -> lane holds key l&15, queries 4g+r
//     dV^T     += dO^T . P ; dK^T += Q^T . dS     (dO^T / Q^T via transpose read)
// Two backward passes instead of atomics: deterministic, and the recomputed S is ~1% of the path FLOPs.
//
// What bounds these kernels is VALU issue, not MFMA, LDS or HBM (rocprofv3 --pmc, profiles/r02_final_*attention*_pmc.txt):
// a 16-query x 64-key step is 16-32 MFMAs but, even after the diet below, ~200 VALU instructions (16 of them quarter-rate
// v_exp_f32).  Hence: fragment-read addresses computed once per lane (FragOff), hardware bf16 packing, scores in the log2
// domain, an unmasked path for key blocks every row of a tile sees completely, row reductions by permlane swaps, scalar
// block loops.  For self-attention with >= 4 query tiles over >= 4 key blocks of_attn_fwd_res_kernel keeps K and V of a
// (batch, head) resident in LDS (LDS-DMA, one workgroup per head); everything else runs one workgroup per 64-query tile.
#include "attn_core.h"

namespace {
using namespace ofa;

// ------------------------------------------------------------------------------------------------
// forward (BWD=false) and dq pass (BWD=true) share the key-block loop
// CMP: compact heads (OfAttnArgs.head_valid = hv < DH): head h owns columns [h hv, (h + 1) hv) of every matrix; the kernel runs at DH with
// the missing columns read as zeros and never stored (GPT-NeoX head size 80 at DH = 128 without padded copies in HBM)
template <int DH, bool BWD, bool SAFE, bool CMP = false>
OF_GLOBAL void OF_BOUNDS(256, 2) of_attn_q_kernel(OfAttnArgs p) {
    constexpr int NKS = DH / 32, NDT = DH / 16, IMG = 64 * DH * 2;
    char* smem = of_smem();
    char* k_n = smem;             // K normal image
    char* v_img = smem + IMG;     // fwd: V transpose image;  dq: V normal image
    char* k_t = smem + 2 * IMG;   // dq only: K transpose image (the forward launch does not allocate it)
    int* s_win = (int*)(smem + (BWD ? 3 : 2) * IMG);  // [64][3] lo,hi,uni ; then [2] block range
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    // grid (heads, query tiles, batch): workgroup ids round-robin over the 8 XCDs, so with the head index fastest every
    // query tile of one (batch, head) runs on XCD h % 8 and re-reads its K / V blocks from that XCD's own L2
    const int q0 = of_bid_y() * 64, h = of_bid_x();
    const long batch = of_bid_z();
    const int hv = CMP ? p.head_valid : DH;
    const int hc = h * hv;
    const float slope = p.alibi_slopes ? p.alibi_slopes[h] : 0.f;

    if (tid < 64) {
        Window w = row_window(p, batch, q0 + tid);
        s_win[tid * 3 + 0] = w.lo;
        s_win[tid * 3 + 1] = w.hi;
        s_win[tid * 3 + 2] = w.uni;
        int lo = w.hi > w.lo ? w.lo : 0x7fffffff, hi = w.hi > w.lo ? w.hi : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            int olo = of_shfl_xor_i(lo, m), ohi = of_shfl_xor_i(hi, m);
            lo = olo < lo ? olo : lo;
            hi = ohi > hi ? ohi : hi;
        }
        if (tid == 0) {
            s_win[192] = lo;
            s_win[193] = hi;
        }
    }
    of_sync();
    const int my_row = q0 + wave * 16 + i16;
    const int lo_i = s_win[(wave * 16 + i16) * 3 + 0], hi_i = s_win[(wave * 16 + i16) * 3 + 1],
              uni_i = s_win[(wave * 16 + i16) * 3 + 2];
    const int rlo = s_win[192], rhi = s_win[193];
    const int kb_lo = of_uniform(rhi > rlo ? rlo / 64 : 0), kb_hi = of_uniform(rhi > rlo ? (rhi + 63) / 64 : 0);

    const bf16_t* qb = p.q + (size_t)batch * p.Lq * p.ldq;
    const bf16_t* kb_ptr = p.k + (size_t)batch * p.Lk * p.ldk;
    const bf16_t* vb_ptr = p.v + (size_t)batch * p.Lk * p.ldv;
    s16x8 qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = gload_frag(qb, p.ldq, my_row, p.Lq, hc + ks * 32 + g * 8, !CMP || ks * 32 + g * 8 < hv);
    const int my_pos = my_row + (p.Lk - p.Lq);   // key index aligned with this query (ALiBi distance origin)
    const FragOff<DH> fo = make_frag_off<DH>(lane);
    const RowCtx rc = make_row_ctx(lo_i, hi_i, uni_i, my_pos, p.scale, slope);
    const TileRange tr = make_tile_range(lo_i, hi_i);
    const bool has_alibi = p.alibi_slopes != nullptr;
    const float dscale = uni_i ? 0.f : 1.f;      // dq pass: a uniform row's probabilities do not depend on q or k
    float m_i = NEG_BIG, l_i = 0.f, lse_i = 0.f, delta_i = 0.f;
    const size_t stat_idx = ((size_t)batch * p.heads + h) * p.Lq + my_row;
    if (BWD) {
        const bf16_t* dob = p.dout + (size_t)batch * p.Lq * p.lddo;
        const bf16_t* ob = p.o + (size_t)batch * p.Lq * p.ldo;
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool cv = !CMP || ks * 32 + g * 8 < hv;
            dof[ks] = gload_frag(dob, p.lddo, my_row, p.Lq, hc + ks * 32 + g * 8, cv);
            const s16x8 o8 = gload_frag(ob, p.ldo, my_row, p.Lq, hc + ks * 32 + g * 8, cv);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += of_bf16_to_f32((bf16_t)dof[ks][e]) * of_bf16_to_f32((bf16_t)o8[e]);
        }
        d = of_rows_sum(d);
        delta_i = d;
        lse_i = my_row < p.Lq ? p.lse[stat_idx] * LOG2E : __builtin_inff();      // log2 domain, like the scores
        if (g == 0 && my_row < p.Lq) p.delta[stat_idx] = d;
    }
    f32x4 acc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K / V tiles are software-pipelined through registers: the global loads of key block kb+1 are issued before key
    // block kb is multiplied (the kernel is otherwise parked on their latency: SQ_WAIT_ANY was 62 % of the wave cycles)
    u32x4 rk[DH / 32], rv[DH / 32];
    if (kb_lo < kb_hi) {
        tile_g2r<DH>(kb_ptr, p.ldk, (long)kb_lo * 64, p.Lk, hc, tid, rk, hv);
        tile_g2r<DH>(vb_ptr, p.ldv, (long)kb_lo * 64, p.Lk, hc, tid, rv, hv);
    }
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        const long key0 = (long)kb * 64;
        if (!BWD) {
            tile_r2s<DH>(rk, tid, k_n, nullptr);
            tile_r2s<DH>(rv, tid, nullptr, v_img);
        } else {
            tile_r2s<DH>(rk, tid, k_n, k_t);
            tile_r2s<DH>(rv, tid, v_img, nullptr);
        }
        of_sync();
        if (kb + 1 < kb_hi) {
            tile_g2r<DH>(kb_ptr, p.ldk, key0 + 64, p.Lk, hc, tid, rk, hv);
            tile_g2r<DH>(vb_ptr, p.ldv, key0 + 64, p.Lk, hc, tid, rv, hv);
        }
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) s[t] = of_mfma(frag_n2<DH>(k_n, fo.n[ks], t * 16), qf[ks], s[t]);
        }
        const float mb = score_block_any(s, rc, tr, (int)key0, 64, g, has_alibi);
        if (!BWD) {
            softmax_pv<DH, SAFE, 4>(s, mb, v_img, fo, lane, acc, m_i, l_i);
        } else {
            f32x4 dp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) dp[t] = of_mfma(frag_n2<DH>(v_img, fo.n[ks], t * 16), dof[ks], dp[t]);
            }
            // masked scores are NEG_BIG: exp2(NEG_BIG - lse) = 0 (lse = +inf marks rows without any visible key)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dp[t][r] = of_exp2(s[t][r] - lse_i) * (dp[t][r] - delta_i) * dscale;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const s16x8 dsf = pack8(dp[2 * s2], dp[2 * s2 + 1]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    acc[dt] = of_mfma(frag_t2<DH, SAFE>(k_t, fo.t[dt], s2 * 32, dt * 16, lane), dsf, acc[dt]);
            }
        }
        of_sync();
    }
    {
        const bool live = my_row < p.Lq;
        const float osc = BWD ? p.scale : (l_i > 0.f ? 1.0f / l_i : 0.f);
        u32x2 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            o[dt] = u32x2{of_pack_bf16(acc[dt][0] * osc, acc[dt][1] * osc), of_pack_bf16(acc[dt][2] * osc, acc[dt][3] * osc)};
        const long row = live ? my_row : 0;
        bf16_t* ob = BWD ? p.dq + ((size_t)batch * p.Lq + row) * p.lddq + hc : p.o + ((size_t)batch * p.Lq + row) * p.ldo + hc;
        store_row_blocks(ob, o, g, live, CMP ? hv : 0x40000000);
        if (!BWD && live && g == 0) p.lse[stat_idx] = l_i > 0.f ? (m_i + of_log2(l_i)) * LN2 : __builtin_inff();
    }
}

// ------------------------------------------------------------------------------------------------
// Forward with K and V of one (batch, head) RESIDENT in LDS ("LDS-staged K/V tiles"): one workgroup per (batch, head)
// instead of one per 64-query tile.  Every key block is fetched once per (batch, head) by LDS-DMA (global_load_lds, no
// VGPR staging; the image swizzles of frag_n / frag_t are applied to the SOURCE address because the DMA destination is
// lane-linear), all query tiles of the head then run against the resident images without a barrier: wave w owns the
// 16-row tiles w, w+4, w+8, ...  Causal self-attention (Lq == Lk: the frozen MPT blocks) loads progressively -- the t-th
// tile of every wave needs key blocks <= t, block t+1 is in flight while step t computes; everything else loads all blocks
// up front (32-80 KB: two workgroups per CU overlap each other).  Same arithmetic, masks and statistics as
// of_attn_q_kernel<DH, false>; eligible when the two images fit the CU's 160 KB.
// HV: DH = every head owns DH columns; 0 = compact heads of a run-time width (OfAttnArgs.head_valid); 80 / 96 (at DH 128) = compact heads
// of that width with the k-steps of S and the d tiles of O that hold no column < HV skipped (3 of 4 k-steps, 5 of 8 d tiles at 80)
template <int DH, int NW, int HV = DH>      // NW waves per workgroup: 8 at head dim 128 (128 KB of images -> one workgroup per CU), else 4
OF_GLOBAL void OF_BOUNDS(NW * 64, 2) of_attn_fwd_res_kernel(OfAttnArgs p) {
    constexpr bool CMP = HV != DH;
    constexpr int NKS = HV > 0 ? (HV + 31) / 32 : DH / 32, NDT = DH / 16, IMG = 64 * DH * 2;
    constexpr int NDTV = HV > 0 ? (HV + 15) / 16 : NDT, NDTE = (NDTV + 1) & ~1;      // d tiles with data; ... in the pairs store_row_blocks takes
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int h = of_bid_x();
    const long batch = of_bid_y();
    const int hv = HV > 0 ? HV : p.head_valid;
    const int hc = h * hv;
    const float slope = p.alibi_slopes ? p.alibi_slopes[h] : 0.f;
    const bool has_alibi = p.alibi_slopes != nullptr;
    const int lk32 = (p.Lk + 31) & ~31;
    char* k_n = smem;
    char* v_t = smem + (size_t)lk32 * DH * 2;
    const int nkb = (p.Lk + 63) / 64, ntiles = (p.Lq + 15) / 16, nsteps = (ntiles + NW - 1) / NW;
    constexpr int BPS = NW / 4;                       // key blocks a step of NW 16-row tiles can newly need (causal)
    const bool progressive = p.causal && p.Lq == p.Lk;
    const bf16_t* qb = p.q + (size_t)batch * p.Lq * p.ldq;
    const bf16_t* kb_ptr = p.k + (size_t)batch * p.Lk * p.ldk;
    const bf16_t* vb_ptr = p.v + (size_t)batch * p.Lk * p.ldv;
    const FragOff<DH> fo = make_frag_off<DH>(lane);

    auto issue = [&](int kb) OF_INLINE_LAMBDA {
        const int rows_blk = lk32 - kb * 64 < 64 ? lk32 - kb * 64 : 64;
        dma_block<DH, false, NW>(kb_ptr, p.ldk, (long)kb * 64, p.Lk, rows_blk, hc, wave, lane, k_n + (size_t)kb * IMG, hv);
        dma_block<DH, true, NW>(vb_ptr, p.ldv, (long)kb * 64, p.Lk, rows_blk, hc, wave, lane, v_t + (size_t)kb * IMG, hv);
    };
    const int first = progressive ? (BPS < nkb ? BPS : nkb) : nkb;
    for (int kb = 0; kb < first; ++kb) issue(kb);
    s16x8 qf[NKS], qn[NKS];
    {
        const int row0 = wave * 16 + i16;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qn[ks] = gload_frag(qb, p.ldq, row0, p.Lq, hc + ks * 32 + g * 8, !CMP || ks * 32 + g * 8 < hv);
    }
    of_wait_vm<0>();
    of_sync();

    u32x2 po[NDTE];                // previous tile's packed output, stored while the next tile computes
    long po_row = -1;
    bool po_live = false;          // the wave holds a finished tile (rows beyond Lq have po_row = -1 and store nothing)
    for (int t = 0; t < nsteps; ++t) {
        const bool more = progressive && (t + 1) * BPS < nkb;
        if (more) {
            for (int kb = (t + 1) * BPS; kb < (t + 2) * BPS && kb < nkb; ++kb) issue(kb);
        }
        if (po_live) {         // wave-uniform: set by the whole wave's tile
            store_row_blocks(p.o + ((size_t)batch * p.Lq + (po_row >= 0 ? po_row : 0)) * p.ldo + hc, po, g, po_row >= 0, CMP ? hv : 0x40000000);
            po_row = -1;
            po_live = false;
        }
        const int ti = t * NW + wave;
        if (ti < ntiles) {     // wave-uniform
            const int my_row = ti * 16 + i16;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[ks] = qn[ks];
            if (ti + NW < ntiles) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    qn[ks] = gload_frag(qb, p.ldq, my_row + NW * 16, p.Lq, hc + ks * 32 + g * 8, !CMP || ks * 32 + g * 8 < hv);
            }
            const Window w = row_window(p, batch, my_row);
            int rlo = w.hi > w.lo ? w.lo : 0x7fffffff, rhi = w.hi > w.lo ? w.hi : 0;
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) {
                const int olo = of_shfl_xor_i(rlo, m), ohi = of_shfl_xor_i(rhi, m);
                rlo = olo < rlo ? olo : rlo;
                rhi = ohi > rhi ? ohi : rhi;
            }
            const int kb_lo = of_uniform(rhi > rlo ? rlo / 64 : 0), kb_hi = of_uniform(rhi > rlo ? (rhi + 63) / 64 : 0);
            const RowCtx rc = make_row_ctx(w.lo, w.hi, w.uni, my_row + (p.Lk - p.Lq), p.scale, slope);
            const TileRange tr = make_tile_range(w.lo, w.hi);
            float m_i = NEG_BIG, l_i = 0.f;
            f32x4 acc[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kb = kb_lo; kb < kb_hi; ++kb) {
                const int key0 = kb * 64;
                const char* kimg = k_n + (size_t)kb * IMG;
                const char* vimg = v_t + (size_t)kb * IMG;
                f32x4 s[4];
                if (lk32 - key0 >= 64) {         // workgroup-uniform: a full block of four 16-key sub-tiles
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks) s[tt] = of_mfma(frag_n2<DH>(kimg, fo.n[ks], tt * 16), qf[ks], s[tt]);
                    }
                    const float mb = score_block_any(s, rc, tr, key0, 64, g, has_alibi);
                    softmax_pv<DH, false, 4, NDTV>(s, mb, vimg, fo, lane, acc, m_i, l_i);
                } else {                          // 32-row tail block of the images
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks) s[tt] = of_mfma(frag_n2<DH>(kimg, fo.n[ks], tt * 16), qf[ks], s[tt]);
                    const float mb = score_block_any(s, rc, tr, key0, 32, g, has_alibi);
                    softmax_pv<DH, false, 2, NDTV>(s, mb, vimg, fo, lane, acc, m_i, l_i);
                }
            }
            {
                const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
#pragma unroll
                for (int dt = 0; dt < NDTE; ++dt)
                    po[dt] = u32x2{of_pack_bf16(acc[dt][0] * inv, acc[dt][1] * inv), of_pack_bf16(acc[dt][2] * inv, acc[dt][3] * inv)};
                po_live = true;
                po_row = my_row < p.Lq ? my_row : -1;
                if (my_row < p.Lq && g == 0)
                    p.lse[((size_t)batch * p.heads + h) * p.Lq + my_row] = l_i > 0.f ? (m_i + of_log2(l_i)) * LN2 : __builtin_inff();
            }
        }
        if (more) {            // workgroup-uniform
            of_wait_vm<0>();
            of_sync();
        }
    }
    if (po_live)
        store_row_blocks(p.o + ((size_t)batch * p.Lq + (po_row >= 0 ? po_row : 0)) * p.ldo + hc, po, g, po_row >= 0, CMP ? hv : 0x40000000);
}

// ------------------------------------------------------------------------------------------------
// dk/dv pass: one workgroup per (key block of 64, head, batch); a wave owns 16 keys.
template <int DH, bool SAFE, bool CMP = false>
OF_GLOBAL void OF_BOUNDS(256, 2) of_attn_dkv_kernel(OfAttnArgs p) {
    constexpr int NKS = DH / 32, NDT = DH / 16, IMG = 64 * DH * 2;
    char* smem = of_smem();
    char* q_n = smem;
    char* do_n = smem + IMG;
    char* q_t = smem + 2 * IMG;
    char* do_t = smem + 3 * IMG;
    int* s_win = (int*)(smem + 4 * IMG);        // [64][3]
    float* s_lse = (float*)(s_win + 64 * 3);    // [64] row log-sum-exp in the log2 domain, [64] delta
    float* s_delta = s_lse + 64;
    int* s_flag = (int*)(s_lse + 128);          // [0] some row sees some key of this block  [1] every row sees every key
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int kblk = of_bid_y(), h = of_bid_x();     // head fastest: see of_attn_q_kernel
    const long batch = of_bid_z();
    const int hv = CMP ? p.head_valid : DH;
    const int hc = h * hv;
    const float slope = p.alibi_slopes ? p.alibi_slopes[h] : 0.f;
    const int key_lo = kblk * 64, key_hi = key_lo + 64;
    const int my_key = key_lo + wave * 16 + i16;
    const FragOff<DH> fo = make_frag_off<DH>(lane);
    const float scale2 = p.scale * LOG2E, slope2 = slope * LOG2E;

    const bf16_t* kb_ptr = p.k + (size_t)batch * p.Lk * p.ldk;
    const bf16_t* vb_ptr = p.v + (size_t)batch * p.Lk * p.ldv;
    const bf16_t* qb = p.q + (size_t)batch * p.Lq * p.ldq;
    const bf16_t* dob = p.dout + (size_t)batch * p.Lq * p.lddo;
    s16x8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const bool cv = !CMP || ks * 32 + g * 8 < hv;
        kf[ks] = gload_frag(kb_ptr, p.ldk, my_key, p.Lk, hc + ks * 32 + g * 8, cv);
        vf[ks] = gload_frag(vb_ptr, p.ldv, my_key, p.Lk, hc + ks * 32 + g * 8, cv);
    }
    f32x4 acck[NDT], accv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        acck[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        accv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nqt = (p.Lq + 63) / 64;
    // causal: query i sees keys < i + 1 + Lk - Lq, so query tiles that end at or before key_lo - (Lk - Lq) never see this block
    int qt_lo = 0;
    if (p.causal) {
        qt_lo = (key_lo - (p.Lk - p.Lq)) / 64;
        qt_lo = qt_lo < 0 ? 0 : (qt_lo > nqt ? nqt : qt_lo);
    }
    // head dim 64: Q / dO tiles are software-pipelined through registers like K / V in the q kernel (at head dim 128
    // the 32 extra VGPRs would spill: that instantiation loads synchronously)
    constexpr bool PREFETCH = DH == 64 || !SAFE;
    u32x4 rq[DH / 32], rdo[DH / 32];
    if (PREFETCH && qt_lo < nqt) {
        tile_g2r<DH>(qb, p.ldq, (long)qt_lo * 64, p.Lq, hc, tid, rq, hv);
        tile_g2r<DH>(dob, p.lddo, (long)qt_lo * 64, p.Lq, hc, tid, rdo, hv);
    }
    for (int qt = qt_lo; qt < nqt; ++qt) {
        const int q0 = qt * 64;
        if (tid < 64) {
            Window w = row_window(p, batch, q0 + tid);
            s_win[tid * 3 + 0] = w.lo;
            s_win[tid * 3 + 1] = w.hi;
            s_win[tid * 3 + 2] = w.uni;
            const int row = q0 + tid;
            const size_t si = ((size_t)batch * p.heads + h) * p.Lq + row;
            s_lse[tid] = row < p.Lq ? p.lse[si] * LOG2E : __builtin_inff();
            s_delta[tid] = row < p.Lq ? p.delta[si] : 0.f;
            int hit = (w.hi > w.lo && w.lo < key_hi && w.hi > key_lo) ? 1 : 0;
            int all = (w.hi > w.lo && !w.uni && w.lo <= key_lo && w.hi >= key_hi) ? 1 : 0;   // this row sees the whole key block
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                hit |= of_shfl_xor_i(hit, m);
                all &= of_shfl_xor_i(all, m);
            }
            if (tid == 0) {
                s_flag[0] = hit;
                s_flag[1] = all;
            }
        }
        of_sync();
        const int hit = of_uniform(s_flag[0]), all_visible = of_uniform(s_flag[1]);
        if (PREFETCH) {
            if (hit) {
                tile_r2s<DH>(rq, tid, q_n, q_t);
                tile_r2s<DH>(rdo, tid, do_n, do_t);
                of_sync();
            }
            if (qt + 1 < nqt) {
                tile_g2r<DH>(qb, p.ldq, q0 + 64, p.Lq, hc, tid, rq, hv);
                tile_g2r<DH>(dob, p.lddo, q0 + 64, p.Lq, hc, tid, rdo, hv);
            }
        }
        if (hit) {
            if (!PREFETCH) {
                load_tile64<DH>(qb, p.ldq, q0, p.Lq, hc, tid, q_n, q_t, hv);
                load_tile64<DH>(dob, p.lddo, q0, p.Lq, hc, tid, do_n, do_t, hv);
                of_sync();
            }
            // two 16-query tiles at a time (= one 32-deep k-step of the dV / dK MFMAs): keeps only 4 score fragments live
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f32x4 pm[2], ds[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * s2 + tt;
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        s = of_mfma(frag_n2<DH>(q_n, fo.n[ks], t * 16), kf[ks], s);
                        dp = of_mfma(frag_n2<DH>(do_n, fo.n[ks], t * 16), vf[ks], dp);
                    }
                    // log2-domain score of (query q0 + 16 t + 4 g + r, key my_key); lse / delta of the four rows in one read each
                    const f32x4 l4 = *(const f32x4*)(s_lse + t * 16 + g * 4), d4 = *(const f32x4*)(s_delta + t * 16 + g * 4);
                    const float bg = slope2 * (float)(my_key - (q0 + t * 16 + g * 4 + p.Lk - p.Lq));
                    if (all_visible) {       // workgroup-uniform: no window test, no uniform rows
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = of_exp2(s[r] * scale2 + (bg - slope2 * (float)r) - l4[r]);
                            pm[tt][r] = pv;
                            ds[tt][r] = pv * (dp[r] - d4[r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int qr = t * 16 + g * 4 + r;
                            const int lo = s_win[qr * 3 + 0], hi = s_win[qr * 3 + 1], uni = s_win[qr * 3 + 2];
                            const bool valid = my_key >= lo && my_key < hi;
                            const float sv = uni ? 0.f : s[r] * scale2 + (bg - slope2 * (float)r);
                            const float pv = valid ? of_exp2(sv - l4[r]) : 0.f;
                            pm[tt][r] = pv;
                            ds[tt][r] = uni ? 0.f : pv * (dp[r] - d4[r]);
                        }
                    }
                }
                const s16x8 pf = pack8(pm[0], pm[1]);
                const s16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    accv[dt] = of_mfma(frag_t2<DH, SAFE>(do_t, fo.t[dt], s2 * 32, dt * 16, lane), pf, accv[dt]);
                    acck[dt] = of_mfma(frag_t2<DH, SAFE>(q_t, fo.t[dt], s2 * 32, dt * 16, lane), dsf, acck[dt]);
                }
            }
        }
        of_sync();
    }
    {
        const bool live = my_key < p.Lk;
        const long row = live ? my_key : 0;
        u32x2 ok[NDT], ov[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            ok[dt] = u32x2{of_pack_bf16(acck[dt][0] * p.scale, acck[dt][1] * p.scale), of_pack_bf16(acck[dt][2] * p.scale, acck[dt][3] * p.scale)};
            ov[dt] = u32x2{of_pack_bf16(accv[dt][0], accv[dt][1]), of_pack_bf16(accv[dt][2], accv[dt][3])};
        }
        store_row_blocks(p.dk + ((size_t)batch * p.Lk + row) * p.lddk + hc, ok, g, live, CMP ? hv : 0x40000000);
        store_row_blocks(p.dv + ((size_t)batch * p.Lk + row) * p.lddv + hc, ov, g, live, CMP ? hv : 0x40000000);
    }
}

// text_time: one wave per sequence, 64-wide chunks with an in-wave inclusive scan.
struct TextTimeArgs {
    const uint8_t* ml;
    int32_t* tt;
    int Lm, Lq, use_cached;
};
OF_GLOBAL void of_text_time_k(TextTimeArgs a) {
    const int b = of_bid_x(), lane = of_tid();
    const uint8_t* row = a.ml + (size_t)b * a.Lm;
    int carry = 0;
    if (a.use_cached) {
        int c = 0;
        for (int i = lane; i < a.Lm; i += 64) c += row[i] ? 1 : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) c += of_shfl_xor_i(c, m);
        for (int i = lane; i < a.Lq; i += 64) a.tt[(size_t)b * a.Lq + i] = c;
        return;
    }
    for (int base = 0; base < a.Lm; base += 64) {
        const int i = base + lane;
        int v = (i < a.Lm && row[i]) ? 1 : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int o = __builtin_bit_cast(int, of_shfl(__builtin_bit_cast(float, v), lane >= off ? lane - off : 0));
            if (lane >= off) v += o;
        }
        if (i < a.Lm) a.tt[(size_t)b * a.Lq + i] = carry + v;
        carry += __builtin_bit_cast(int, of_shfl(__builtin_bit_cast(float, v), 63));
    }
}

int check(const OfAttnArgs& a, bool bwd) {
    if (!a.q || !a.k || !a.v || !a.o || !a.lse) return OF_E_ARG;
    if (a.batch <= 0 || a.heads <= 0 || a.Lq <= 0 || a.Lk <= 0) return OF_E_ARG;
    if ((a.ldq & 7) || (a.ldk & 7) || (a.ldv & 7) || (a.ldo & 7)) return OF_E_ALIGN;
    if (((uintptr_t)a.q & 15) || ((uintptr_t)a.k & 15) || ((uintptr_t)a.v & 15) || ((uintptr_t)a.o & 15)) return OF_E_ALIGN;
    if (a.text_time && (a.n_per_media <= 0 || a.T_img <= 0)) return OF_E_ARG;
    if (a.head_dim != 0 && a.head_dim != 64 && a.head_dim != 128) return OF_E_SHAPE;
    {   // compact heads: a multiple of 8 columns (every access is 16 bytes wide)
        const int dh = a.head_dim == 128 ? 128 : 64;
        if (a.head_valid != 0 && a.head_valid != dh && ((a.head_valid & 7) || a.head_valid < 8 || a.head_valid > dh)) return OF_E_SHAPE;
        if (a.head_valid != 0 && a.head_valid != dh && a.safe == 1) return OF_E_ARG;     // no scalar-LDS self-check instantiation of that form
    }
    if (a.causal && a.text_time) return OF_E_ARG;
    if (bwd) {
        if (!a.dout || !a.dq || !a.dk || !a.dv || !a.delta) return OF_E_ARG;
        if ((a.lddo & 7) || (a.lddq & 7) || (a.lddk & 7) || (a.lddv & 7)) return OF_E_ALIGN;
        // dq / dk / dv leave through 16-byte stores (store_row_blocks), dout is read 16 bytes wide
        if (((uintptr_t)a.dout & 15) || ((uintptr_t)a.dq & 15) || ((uintptr_t)a.dk & 15) || ((uintptr_t)a.dv & 15)) return OF_E_ALIGN;
    }
    return 0;
}

}  // namespace

namespace {
// LDS bytes of the resident-K/V forward, or 0 when this launch stays on the tiled kernel
template <int DH>
size_t resident_smem(const OfAttnArgs& a) {
    if (a.safe != 0 && a.safe != 3) return 0;                    // safe = 1: scalar-LDS path, 2: tiled kernel (both of_attn_q_kernel)
    const size_t need = (size_t)((a.Lk + 31) & ~31) * DH * 2 * 2;
    if (need > 160 * 1024) return 0;
    if (a.safe == 3) return need;                                // self-check: the resident form whenever the images fit
    // Measured (profiles/r02_final_*): the resident form wins where a (batch, head) has >= 4 query tiles re-reading >= 4
    // key blocks (CLIP ViT 257 x 257: 65 vs 71 us) and ties on the frozen MPT blocks (256 x 256, head dim 128); the gated
    // cross-attention (128 keys) and Perceiver (64 queries) cores stay on the tiled kernel (11 vs 14 us, 13 vs 16 us).
    if (a.Lq < 256 || a.Lk < 256) return 0;
    if ((long)a.batch * a.heads < 64) return 0;                   // few heads: the tiled grid has more workgroups
    // One workgroup per (batch, head), 160 KB / need of them per CU: a last round of those slots that is mostly empty costs a whole round
    // (OF-9B's frozen MPT-7B blocks, 10 x 32 = 320 heads at one workgroup per CU: 48.6 us against 38.8 us for the tiled kernel; 384 heads:
    // 49.9 against 44.0; 512 and 192 heads: a tie -- profiles/r06zw_attn_fwd_forms_probe.jsonl): resident where the rounds are >= 80 % full
#ifndef OF_AB_ATTN_FWD_RESIDENT_ANY_ROUNDS                         // tools/ab builds only (step-level A/B of this rule)
    {
        const long slots = (long)OF_NUM_CUS * (long)(160 * 1024 / need), n = (long)a.batch * a.heads, rounds = (n + slots - 1) / slots;
        if (10 * n < 8 * rounds * slots) return 0;
    }
#endif
    return need;
}
template <int DH>
bool compact(const OfAttnArgs& a) {
    return a.head_valid != 0 && a.head_valid != DH;
}
template <int DH>
int launch_fwd(const OfAttnArgs& a, of_stream_t s) {
    const bool cmp = compact<DH>(a);
    if (const size_t res = resident_smem<DH>(a)) {
        const of_dim3 grid{(unsigned)a.heads, (unsigned)a.batch, 1};
        if (cmp && DH == 128 && a.head_valid == 80) return of_launch(of_attn_fwd_res_kernel<128, 8, 80>, grid, 512, res, s, a);
        if (cmp && DH == 128 && a.head_valid == 96) return of_launch(of_attn_fwd_res_kernel<128, 8, 96>, grid, 512, res, s, a);
        if (cmp) return of_launch(of_attn_fwd_res_kernel<DH, (DH == 128 ? 8 : 4), 0>, grid, DH == 128 ? 512 : 256, res, s, a);
        return of_launch(of_attn_fwd_res_kernel<DH, (DH == 128 ? 8 : 4)>, grid, DH == 128 ? 512 : 256, res, s, a);
    }
    of_dim3 grid{(unsigned)a.heads, (unsigned)((a.Lq + 63) / 64), (unsigned)a.batch};
    const size_t smem = 2 * (64 * DH * 2) + 196 * sizeof(int);
    if (cmp) return of_launch(of_attn_q_kernel<DH, false, false, true>, grid, 256, smem, s, a);
    if (a.safe == 1) return of_launch(of_attn_q_kernel<DH, false, true>, grid, 256, smem, s, a);
    return of_launch(of_attn_q_kernel<DH, false, false>, grid, 256, smem, s, a);
}
// the single-pass backward (attn_bwd_res.hip) or the two passes below
bool bwd_single_pass(const OfAttnArgs& a) {
    if (a.safe == 1 || a.safe == 2) return false;                // 1: scalar-LDS self-check path, 2: the two-pass kernels
    if (!attn_bwd_res_fits(a)) return false;
    if (a.safe == 3) return true;                                // self-check: the single pass whenever it fits
#ifdef OF_AB_ATTN_BWD_TWO_PASS                                    // tools/ab builds only (step-level A/B of the two forms)
    return false;
#endif
    // One workgroup of 4 waves per (batch, head), one per CU, ~49 us each at 256 x 256 x 128; the two passes cost ~0.27 us per head
    // of a full chip (profiles/r06z*_attn_bwd_single_pass_probe*.jsonl: 98 vs 139 us at the frozen MPT-1B blocks' 512 heads).  The
    // single pass wins where its rounds of OF_NUM_CUS workgroups are at least ~70 % full: 512 or 256 heads yes, MPT-7B's 320
    // (B 10 x 32 heads: two rounds for 1.25 rounds of work) no.
    if (a.Lq < 128 || a.Lk < 128) return false;
    const long n = (long)a.batch * a.heads, rounds = (n + OF_NUM_CUS - 1) / OF_NUM_CUS;
    return 10 * n >= 7 * rounds * OF_NUM_CUS;
}
// `a` restricted to the sequences [b0, b0 + nb) of its batch
OfAttnArgs batch_slice(const OfAttnArgs& a, int b0, int nb) {
    OfAttnArgs r = a;
    r.batch = nb;
    r.q += (size_t)b0 * a.Lq * a.ldq;
    r.k += (size_t)b0 * a.Lk * a.ldk;
    r.v += (size_t)b0 * a.Lk * a.ldv;
    r.o += (size_t)b0 * a.Lq * a.ldo;
    r.dout += (size_t)b0 * a.Lq * a.lddo;
    r.dq += (size_t)b0 * a.Lq * a.lddq;
    r.dk += (size_t)b0 * a.Lk * a.lddk;
    r.dv += (size_t)b0 * a.Lk * a.lddv;
    r.lse += (size_t)b0 * a.heads * a.Lq;
    r.delta += (size_t)b0 * a.heads * a.Lq;
    if (a.kv_len) r.kv_len += b0;
    if (a.text_time) r.text_time += (size_t)b0 * a.Lq;
    return r;
}
template <int DH>
int launch_bwd(const OfAttnArgs& a, of_stream_t s) {
    constexpr int IMG = 64 * DH * 2;
    if (bwd_single_pass(a)) return attn_bwd_res_launch(a, s);
    // A head count the single pass is not chosen for because its last round of OF_NUM_CUS workgroups would be mostly empty (OF-9B's frozen
    // MPT-7B blocks: 10 x 32 = 320 heads): the whole rounds' worth of SEQUENCES takes the single pass (49 us per round against 69 us
    // for the same heads on the two passes), the remaining sequences the two passes.  Every (batch, head) is computed by one form or the
    // other exactly as a launch of that form alone would: run to run the same bits.
#ifndef OF_AB_ATTN_BWD_TWO_PASS
    if (a.safe == 0 && attn_bwd_res_fits(a) && a.Lq >= 128 && a.Lk >= 128) {
        const long n = (long)a.batch * a.heads;
        const long whole = n / OF_NUM_CUS * OF_NUM_CUS;
        if (whole > 0 && whole % a.heads == 0 && whole < n) {
            const int nb = (int)(whole / a.heads);
            const int rc = attn_bwd_res_launch(batch_slice(a, 0, nb), s);
            if (rc) return rc;
            OfAttnArgs rest = batch_slice(a, nb, a.batch - nb);
            rest.safe = 2;
            return launch_bwd<DH>(rest, s);
        }
    }
#endif
    of_dim3 gq{(unsigned)a.heads, (unsigned)((a.Lq + 63) / 64), (unsigned)a.batch};
    const size_t smem_q = 3 * IMG + 196 * sizeof(int);
    const bool cmp = compact<DH>(a);
    int rc = cmp           ? of_launch(of_attn_q_kernel<DH, true, false, true>, gq, 256, smem_q, s, a)
             : a.safe == 1 ? of_launch(of_attn_q_kernel<DH, true, true>, gq, 256, smem_q, s, a)
                           : of_launch(of_attn_q_kernel<DH, true, false>, gq, 256, smem_q, s, a);
    if (rc) return rc;
    of_dim3 gk{(unsigned)a.heads, (unsigned)((a.Lk + 63) / 64), (unsigned)a.batch};
    const size_t smem_k = 4 * IMG + 64 * 3 * sizeof(int) + 128 * sizeof(float) + 16;
    if (cmp) return of_launch(of_attn_dkv_kernel<DH, false, true>, gk, 256, smem_k, s, a);
    return a.safe == 1 ? of_launch(of_attn_dkv_kernel<DH, true>, gk, 256, smem_k, s, a)
                  : of_launch(of_attn_dkv_kernel<DH, false>, gk, 256, smem_k, s, a);
}
}  // namespace

extern "C" int of_attn_fwd(const OfAttnArgs* args, void* stream) {
    if (!args) return OF_E_ARG;
    int rc = check(*args, false);
    if (rc) return rc;
    return args->head_dim == 128 ? launch_fwd<128>(*args, (of_stream_t)stream) : launch_fwd<64>(*args, (of_stream_t)stream);
}

extern "C" int of_attn_bwd(const OfAttnArgs* args, void* stream) {
    if (!args) return OF_E_ARG;
    int rc = check(*args, true);
    if (rc) return rc;
    return args->head_dim == 128 ? launch_bwd<128>(*args, (of_stream_t)stream) : launch_bwd<64>(*args, (of_stream_t)stream);
}

extern "C" int of_text_time(const uint8_t* media_locations, int32_t* text_time, int B, int Lm, int Lq, int use_cached,
                            void* stream) {
    if (!media_locations || !text_time || B <= 0 || Lm < 0 || Lq <= 0) return OF_E_ARG;
    if (!use_cached && Lm != Lq) return OF_E_SHAPE;
    TextTimeArgs a{media_locations, text_time, Lm, Lq, use_cached};
    return of_launch(of_text_time_k, of_dim3{(unsigned)B, 1, 1}, 64, 0, (of_stream_t)stream, a);
}
