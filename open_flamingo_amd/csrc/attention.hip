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
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
// one [64 rows][DH] bf16 LDS image; two swizzles of the same data:
//   "normal"    (ds_read_b128 fragments, k = column): 16-B slot s of row r at slot s ^ f(r)
//   "transpose" (ds_read_b64_tr_b16 fragments, k = row): 32-B chunk c of row r at chunk c ^ f(r)
// DH = 64: 128-B rows, two rows per 256-B bank row -> f = (r>>1)&7 / (r>>1)&3; DH = 128: 256-B rows -> f = r&15 / r&7.
template <int DH>
OF_DEV int img_n_off(int row, int slot) {
    return DH == 64 ? row * 128 + ((slot ^ ((row >> 1) & 7)) << 4) : row * 256 + ((slot ^ (row & 15)) << 4);
}
template <int DH>
OF_DEV int img_t_off(int row, int col) {
    return DH == 64 ? row * 128 + ((((col >> 4)) ^ ((row >> 1) & 3)) << 5) + ((col & 15) << 1)
                    : row * 256 + ((((col >> 4)) ^ (row & 7)) << 5) + ((col & 15) << 1);
}

// cooperative load of a 64 x DH bf16 tile (rows row0.., columns col0..col0+DH-1 of a row-major matrix) into
// the "normal" image (ds_read_b128 fragments, k = column) and/or the "transpose" image (tr-read
// fragments, k = row).  Rows >= nrows are zero-filled.
template <int DH>
OF_DEV void load_tile64(const bf16_t* __restrict__ src, long ld, long row0, long nrows, int col0, int tid,
                        char* img_n, char* img_t) {
    constexpr int SPR = DH / 8;   // 16-byte slots per row
#pragma unroll
    for (int c = 0; c < SPR / 4; ++c) {
        int id = c * 256 + tid;
        int row = id / SPR, cs = id % SPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + row < nrows) v = *(const u32x4*)(src + (size_t)(row0 + row) * ld + col0 + cs * 8);
        if (img_n) *(u32x4*)(img_n + img_n_off<DH>(row, cs)) = v;
        if (img_t) *(u32x4*)(img_t + img_t_off<DH>(row, cs * 8)) = v;
    }
}
// the same tile load in two halves so that the global loads of tile i+1 can be in flight while tile i is multiplied
template <int DH>
OF_DEV void tile_g2r(const bf16_t* __restrict__ src, long ld, long row0, long nrows, int col0, int tid, u32x4 (&r)[DH / 32]) {
    constexpr int SPR = DH / 8;
#pragma unroll
    for (int c = 0; c < SPR / 4; ++c) {
        int id = c * 256 + tid;
        int row = id / SPR, cs = id % SPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + row < nrows) v = *(const u32x4*)(src + (size_t)(row0 + row) * ld + col0 + cs * 8);
        r[c] = v;
    }
}
template <int DH>
OF_DEV void tile_r2s(const u32x4 (&r)[DH / 32], int tid, char* img_n, char* img_t) {
    constexpr int SPR = DH / 8;
#pragma unroll
    for (int c = 0; c < SPR / 4; ++c) {
        int id = c * 256 + tid;
        int row = id / SPR, cs = id % SPR;
        if (img_n) *(u32x4*)(img_n + img_n_off<DH>(row, cs)) = r[c];
        if (img_t) *(u32x4*)(img_t + img_t_off<DH>(row, cs * 8)) = r[c];
    }
}
template <int DH>
OF_DEV s16x8 frag_n(const char* img, int row_base, int kk, int lane) {
    return *(const s16x8*)(img + img_n_off<DH>(row_base + (lane & 15), kk * 4 + (lane >> 4)));
}
// k-slot e = 4h+j of lane group g  <->  image row kbase + 16h + 4g + j ; matrix column = col_base + (lane&15)
template <int DH, bool SAFE>
OF_DEV s16x8 frag_t(const char* img, int kbase, int col_base, int lane) {
    const int g = lane >> 4, i = lane & 15;
    s16x8 f;
    if (!SAFE) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            s16x4 t = of_lds_tr(img + img_t_off<DH>(kbase + h * 16 + g * 4 + (i >> 2), col_base + (i & 3) * 4));
            f[h * 4 + 0] = t[0];
            f[h * 4 + 1] = t[1];
            f[h * 4 + 2] = t[2];
            f[h * 4 + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            f[e] = *(const short*)(img + img_t_off<DH>(kbase + (e >> 2) * 16 + g * 4 + (e & 3), col_base + i));
    }
    return f;
}
OF_DEV s16x8 gload_frag(const bf16_t* __restrict__ base, long ld, long row, long nrows, int col) {
    s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nrows) z = *(const s16x8*)(base + (size_t)row * ld + col);
    return z;
}
OF_DEV s16x8 pack8(const f32x4& a, const f32x4& b) {
    s16x8 f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f[r] = (short)of_f32_to_bf16(a[r]);
        f[4 + r] = (short)of_f32_to_bf16(b[r]);
    }
    return f;
}

struct Window {
    int lo, hi, uni;
};
// key window of a query row (see file header); row_valid = row < Lq
OF_DEV Window row_window(const OfAttnArgs& p, long batch, int row) {
    Window w{0, 0, 0};
    if (row >= p.Lq) return w;
    if (p.causal) {
        w.hi = row + 1 + (p.Lk - p.Lq);
        const int len = p.kv_len ? p.kv_len[batch] : p.Lk;   // right-padded sequences: keys >= len are padding
        if (w.hi > len) w.hi = len;
        if (w.hi > p.Lk) w.hi = p.Lk;
        if (w.hi < 0) w.hi = 0;
        return w;
    }
    if (!p.text_time) {
        w.hi = p.Lk;
        return w;
    }
    const int tt = p.text_time[batch * p.Lq + row];
    const int n = p.n_per_media, T = p.T_img;
    if (p.only_immediate) {
        if (tt == 0) return w;
        if (tt <= T) {
            w.lo = (tt - 1) * n;
            w.hi = tt * n;
        } else {
            w.hi = T * n;
            w.uni = 1;
        }
    } else {
        if (tt == 0) {
            w.hi = T * n;
            w.uni = 1;
        } else {
            w.hi = (tt < T ? tt : T) * n;
        }
    }
    if (w.hi > p.Lk) w.hi = p.Lk;
    return w;
}

// ------------------------------------------------------------------------------------------------
// forward (BWD=false) and dq pass (BWD=true) share the key-block loop
template <int DH, bool BWD, bool SAFE>
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
    const int hc = h * DH;
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
    const int kb_lo = rhi > rlo ? rlo / 64 : 0, kb_hi = rhi > rlo ? (rhi + 63) / 64 : 0;

    const bf16_t* qb = p.q + (size_t)batch * p.Lq * p.ldq;
    const bf16_t* kb_ptr = p.k + (size_t)batch * p.Lk * p.ldk;
    const bf16_t* vb_ptr = p.v + (size_t)batch * p.Lk * p.ldv;
    s16x8 qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = gload_frag(qb, p.ldq, my_row, p.Lq, hc + ks * 32 + g * 8);
    const int my_pos = my_row + (p.Lk - p.Lq);   // key index aligned with this query (ALiBi distance origin)
    float m_i = NEG_BIG, l_i = 0.f, lse_i = 0.f, delta_i = 0.f;
    const size_t stat_idx = ((size_t)batch * p.heads + h) * p.Lq + my_row;
    if (BWD) {
        const bf16_t* dob = p.dout + (size_t)batch * p.Lq * p.lddo;
        const bf16_t* ob = p.o + (size_t)batch * p.Lq * p.ldo;
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            dof[ks] = gload_frag(dob, p.lddo, my_row, p.Lq, hc + ks * 32 + g * 8);
            const s16x8 o8 = gload_frag(ob, p.ldo, my_row, p.Lq, hc + ks * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += of_bf16_to_f32((bf16_t)dof[ks][e]) * of_bf16_to_f32((bf16_t)o8[e]);
        }
        d += of_shfl_xor(d, 16);
        d += of_shfl_xor(d, 32);
        delta_i = d;
        lse_i = my_row < p.Lq ? p.lse[stat_idx] : __builtin_inff();
        if (g == 0 && my_row < p.Lq) p.delta[stat_idx] = d;
    }
    f32x4 acc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K / V tiles are software-pipelined through registers: the global loads of key block kb+1 are issued before key
    // block kb is multiplied (the kernel is otherwise parked on their latency: SQ_WAIT_ANY was 62 % of the wave cycles)
    u32x4 rk[DH / 32], rv[DH / 32];
    if (kb_lo < kb_hi) {
        tile_g2r<DH>(kb_ptr, p.ldk, (long)kb_lo * 64, p.Lk, hc, tid, rk);
        tile_g2r<DH>(vb_ptr, p.ldv, (long)kb_lo * 64, p.Lk, hc, tid, rv);
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
            tile_g2r<DH>(kb_ptr, p.ldk, key0 + 64, p.Lk, hc, tid, rk);
            tile_g2r<DH>(vb_ptr, p.ldv, key0 + 64, p.Lk, hc, tid, rv);
        }
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) s[t] = of_mfma(frag_n<DH>(k_n, t * 16, ks, lane), qf[ks], s[t]);
        }
        // masked scores
        float mb = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = (int)key0 + t * 16 + g * 4 + r;
                const bool valid = j >= lo_i && j < hi_i;
                float sv = uni_i ? 0.f : s[t][r] * p.scale + slope * (float)(j - my_pos);
                sv = valid ? sv : NEG_BIG;
                s[t][r] = sv;
                mb = sv > mb ? sv : mb;
            }
        if (!BWD) {
            float o16 = of_shfl_xor(mb, 16);
            mb = o16 > mb ? o16 : mb;
            float o32 = of_shfl_xor(mb, 32);
            mb = o32 > mb ? o32 : mb;
            const float m_new = mb > m_i ? mb : m_i;
            const float alpha = of_exp(m_i - m_new);
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = s[t][r] > 0.5f * NEG_BIG ? of_exp(s[t][r] - m_new) : 0.f;
                    s[t][r] = pv;
                    rs += pv;
                }
            rs += of_shfl_xor(rs, 16);
            rs += of_shfl_xor(rs, 32);
            l_i = l_i * alpha + rs;
            m_i = m_new;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                acc[dt][0] *= alpha;
                acc[dt][1] *= alpha;
                acc[dt][2] *= alpha;
                acc[dt][3] *= alpha;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const s16x8 pf = pack8(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    acc[dt] = of_mfma(frag_t<DH, SAFE>(v_img, s2 * 32, dt * 16, lane), pf, acc[dt]);
            }
        } else {
            f32x4 dp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) dp[t] = of_mfma(frag_n<DH>(v_img, t * 16, ks, lane), dof[ks], dp[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = s[t][r] > 0.5f * NEG_BIG ? of_exp(s[t][r] - lse_i) : 0.f;
                    dp[t][r] = uni_i ? 0.f : pv * (dp[t][r] - delta_i);
                }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const s16x8 dsf = pack8(dp[2 * s2], dp[2 * s2 + 1]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    acc[dt] = of_mfma(frag_t<DH, SAFE>(k_t, s2 * 32, dt * 16, lane), dsf, acc[dt]);
            }
        }
        of_sync();
    }
    if (my_row < p.Lq) {
        if (!BWD) {
            const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
            bf16_t* ob = p.o + ((size_t)batch * p.Lq + my_row) * p.ldo + hc;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                u32x2 o = {of_pack_bf16(acc[dt][0] * inv, acc[dt][1] * inv),
                           of_pack_bf16(acc[dt][2] * inv, acc[dt][3] * inv)};
                *(u32x2*)(ob + dt * 16 + g * 4) = o;
            }
            if (g == 0) p.lse[stat_idx] = l_i > 0.f ? m_i + of_log(l_i) : __builtin_inff();
        } else {
            bf16_t* dqb = p.dq + ((size_t)batch * p.Lq + my_row) * p.lddq + hc;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                u32x2 o = {of_pack_bf16(acc[dt][0] * p.scale, acc[dt][1] * p.scale),
                           of_pack_bf16(acc[dt][2] * p.scale, acc[dt][3] * p.scale)};
                *(u32x2*)(dqb + dt * 16 + g * 4) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dk/dv pass: one workgroup per (key block of 64, head, batch); a wave owns 16 keys.
template <int DH, bool SAFE>
OF_GLOBAL void OF_BOUNDS(256, 2) of_attn_dkv_kernel(OfAttnArgs p) {
    constexpr int NKS = DH / 32, NDT = DH / 16, IMG = 64 * DH * 2;
    char* smem = of_smem();
    char* q_n = smem;
    char* do_n = smem + IMG;
    char* q_t = smem + 2 * IMG;
    char* do_t = smem + 3 * IMG;
    int* s_win = (int*)(smem + 4 * IMG);        // [64][3]
    float* s_stat = (float*)(s_win + 64 * 3);   // [64][2] lse, delta
    int* s_flag = (int*)(s_stat + 128);
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int kblk = of_bid_y(), h = of_bid_x();     // head fastest: see of_attn_q_kernel
    const long batch = of_bid_z();
    const int hc = h * DH;
    const float slope = p.alibi_slopes ? p.alibi_slopes[h] : 0.f;
    const int key_lo = kblk * 64, key_hi = key_lo + 64;
    const int my_key = key_lo + wave * 16 + i16;

    const bf16_t* kb_ptr = p.k + (size_t)batch * p.Lk * p.ldk;
    const bf16_t* vb_ptr = p.v + (size_t)batch * p.Lk * p.ldv;
    const bf16_t* qb = p.q + (size_t)batch * p.Lq * p.ldq;
    const bf16_t* dob = p.dout + (size_t)batch * p.Lq * p.lddo;
    s16x8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        kf[ks] = gload_frag(kb_ptr, p.ldk, my_key, p.Lk, hc + ks * 32 + g * 8);
        vf[ks] = gload_frag(vb_ptr, p.ldv, my_key, p.Lk, hc + ks * 32 + g * 8);
    }
    f32x4 acck[NDT], accv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        acck[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        accv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nqt = (p.Lq + 63) / 64;
    // head dim 64: Q / dO tiles are software-pipelined through registers like K / V in the q kernel (at head dim 128
    // the 32 extra VGPRs would spill: that instantiation loads synchronously)
    constexpr bool PREFETCH = DH == 64;
    u32x4 rq[DH / 32], rdo[DH / 32];
    if (PREFETCH) {
        tile_g2r<DH>(qb, p.ldq, 0, p.Lq, hc, tid, rq);
        tile_g2r<DH>(dob, p.lddo, 0, p.Lq, hc, tid, rdo);
    }
    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = qt * 64;
        if (tid < 64) {
            Window w = row_window(p, batch, q0 + tid);
            s_win[tid * 3 + 0] = w.lo;
            s_win[tid * 3 + 1] = w.hi;
            s_win[tid * 3 + 2] = w.uni;
            const int row = q0 + tid;
            const size_t si = ((size_t)batch * p.heads + h) * p.Lq + row;
            s_stat[tid * 2 + 0] = row < p.Lq ? p.lse[si] : __builtin_inff();
            s_stat[tid * 2 + 1] = row < p.Lq ? p.delta[si] : 0.f;
            int hit = (w.hi > w.lo && w.lo < key_hi && w.hi > key_lo) ? 1 : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) hit |= of_shfl_xor_i(hit, m);
            if (tid == 0) s_flag[0] = hit;
        }
        of_sync();
        const int hit = s_flag[0];
        if (PREFETCH) {
            if (hit) {
                tile_r2s<DH>(rq, tid, q_n, q_t);
                tile_r2s<DH>(rdo, tid, do_n, do_t);
                of_sync();
            }
            if (qt + 1 < nqt) {
                tile_g2r<DH>(qb, p.ldq, q0 + 64, p.Lq, hc, tid, rq);
                tile_g2r<DH>(dob, p.lddo, q0 + 64, p.Lq, hc, tid, rdo);
            }
        }
        if (hit) {
            if (!PREFETCH) {
                load_tile64<DH>(qb, p.ldq, q0, p.Lq, hc, tid, q_n, q_t);
                load_tile64<DH>(dob, p.lddo, q0, p.Lq, hc, tid, do_n, do_t);
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
                        s = of_mfma(frag_n<DH>(q_n, t * 16, ks, lane), kf[ks], s);
                        dp = of_mfma(frag_n<DH>(do_n, t * 16, ks, lane), vf[ks], dp);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qr = t * 16 + g * 4 + r;
                        const int lo = s_win[qr * 3 + 0], hi = s_win[qr * 3 + 1], uni = s_win[qr * 3 + 2];
                        const float lse = s_stat[qr * 2 + 0], delta = s_stat[qr * 2 + 1];
                        const bool valid = my_key >= lo && my_key < hi;
                        const float sv = uni ? 0.f : s[r] * p.scale + slope * (float)(my_key - (q0 + qr + p.Lk - p.Lq));
                        const float pv = valid ? of_exp(sv - lse) : 0.f;
                        pm[tt][r] = pv;
                        ds[tt][r] = uni ? 0.f : pv * (dp[r] - delta);
                    }
                }
                const s16x8 pf = pack8(pm[0], pm[1]);
                const s16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    accv[dt] = of_mfma(frag_t<DH, SAFE>(do_t, s2 * 32, dt * 16, lane), pf, accv[dt]);
                    acck[dt] = of_mfma(frag_t<DH, SAFE>(q_t, s2 * 32, dt * 16, lane), dsf, acck[dt]);
                }
            }
        }
        of_sync();
    }
    if (my_key < p.Lk) {
        bf16_t* dkb = p.dk + ((size_t)batch * p.Lk + my_key) * p.lddk + hc;
        bf16_t* dvb = p.dv + ((size_t)batch * p.Lk + my_key) * p.lddv + hc;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            u32x2 ok = {of_pack_bf16(acck[dt][0] * p.scale, acck[dt][1] * p.scale),
                        of_pack_bf16(acck[dt][2] * p.scale, acck[dt][3] * p.scale)};
            u32x2 ov = {of_pack_bf16(accv[dt][0], accv[dt][1]), of_pack_bf16(accv[dt][2], accv[dt][3])};
            *(u32x2*)(dkb + dt * 16 + g * 4) = ok;
            *(u32x2*)(dvb + dt * 16 + g * 4) = ov;
        }
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
    if (a.causal && a.text_time) return OF_E_ARG;
    if (bwd) {
        if (!a.dout || !a.dq || !a.dk || !a.dv || !a.delta) return OF_E_ARG;
        if ((a.lddo & 7) || (a.lddq & 7) || (a.lddk & 7) || (a.lddv & 7)) return OF_E_ALIGN;
    }
    return 0;
}

}  // namespace

namespace {
template <int DH>
int launch_fwd(const OfAttnArgs& a, of_stream_t s) {
    of_dim3 grid{(unsigned)a.heads, (unsigned)((a.Lq + 63) / 64), (unsigned)a.batch};
    const size_t smem = 2 * (64 * DH * 2) + 196 * sizeof(int);
    if (a.safe) return of_launch(of_attn_q_kernel<DH, false, true>, grid, 256, smem, s, a);
    return of_launch(of_attn_q_kernel<DH, false, false>, grid, 256, smem, s, a);
}
template <int DH>
int launch_bwd(const OfAttnArgs& a, of_stream_t s) {
    constexpr int IMG = 64 * DH * 2;
    of_dim3 gq{(unsigned)a.heads, (unsigned)((a.Lq + 63) / 64), (unsigned)a.batch};
    const size_t smem_q = 3 * IMG + 196 * sizeof(int);
    int rc = a.safe ? of_launch(of_attn_q_kernel<DH, true, true>, gq, 256, smem_q, s, a)
                    : of_launch(of_attn_q_kernel<DH, true, false>, gq, 256, smem_q, s, a);
    if (rc) return rc;
    of_dim3 gk{(unsigned)a.heads, (unsigned)((a.Lk + 63) / 64), (unsigned)a.batch};
    const size_t smem_k = 4 * IMG + 64 * 3 * sizeof(int) + 128 * sizeof(float) + 16;
    return a.safe ? of_launch(of_attn_dkv_kernel<DH, true>, gk, 256, smem_k, s, a)
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
