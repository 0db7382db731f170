// Shared pieces of the windowed flash-attention kernels (attention.hip) and of the fused gated-cross-attention branch
// (xattn_fused.hip): LDS tile images + fragment reads, the key window of a query row (reference helpers.py:196-229), log2-domain
// scores and the online-softmax / PV step.  Moved here verbatim from attention.hip (round 6).
#pragma once
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace ofa {
// A lane (row i16, group g = lane >> 4) holds 4 packed-bf16 columns (8 bytes) of every 16-column block of its output row.
// Stored as they are, one store instruction writes 32-byte pieces of 16 rows and every 128-byte line takes four instructions.
// Blocks 2m / 2m+1 are exchanged between the lane groups g and g ^ 1 first (v_permlane16_swap), so that a lane owns 8 consecutive
// columns: 16-byte stores, half as many, 64 contiguous bytes of a row per instruction.  All 64 lanes must call (cross-lane);
// `live` masks the stores of rows beyond the end, `hv` the 8-column pieces at or beyond column hv (compact heads, OfAttnArgs.head_valid:
// the kernels' columns hv .. head_dim - 1 do not exist in memory).
template <int NB>
OF_DEV void store_row_blocks(bf16_t* rowp, const u32x2 (&v)[NB], int g, bool live, int hv = 0x40000000) {
#pragma unroll
    for (int m = 0; m < NB / 2; ++m) {
        unsigned a0 = v[2 * m][0], a1 = v[2 * m][1], b0 = v[2 * m + 1][0], b1 = v[2 * m + 1][1];
        of_pair_rows16(a0, b0);
        of_pair_rows16(a1, b1);
        if (live && (2 * m + (g & 1)) * 16 + 4 * (g & ~1) < hv) *(u32x4*)(rowp + (2 * m + (g & 1)) * 16 + 4 * (g & ~1)) = u32x4{a0, a1, b0, b1};
    }
}


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
// fragments, k = row).  Rows >= nrows are zero-filled, and so are the columns >= hv (compact heads: OfAttnArgs.head_valid).
template <int DH>
OF_DEV void load_tile64(const bf16_t* __restrict__ src, long ld, long row0, long nrows, int col0, int tid,
                        char* img_n, char* img_t, int hv = DH) {
    constexpr int SPR = DH / 8;   // 16-byte slots per row
#pragma unroll
    for (int c = 0; c < SPR / 4; ++c) {
        int id = c * 256 + tid;
        int row = id / SPR, cs = id % SPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + row < nrows && cs * 8 < hv) v = *(const u32x4*)(src + (size_t)(row0 + row) * ld + col0 + cs * 8);
        if (img_n) *(u32x4*)(img_n + img_n_off<DH>(row, cs)) = v;
        if (img_t) *(u32x4*)(img_t + img_t_off<DH>(row, cs * 8)) = v;
    }
}
// the same tile load in two halves so that the global loads of tile i+1 can be in flight while tile i is multiplied
template <int DH>
OF_DEV void tile_g2r(const bf16_t* __restrict__ src, long ld, long row0, long nrows, int col0, int tid, u32x4 (&r)[DH / 32],
                     int hv = DH) {
    constexpr int SPR = DH / 8;
#pragma unroll
    for (int c = 0; c < SPR / 4; ++c) {
        int id = c * 256 + tid;
        int row = id / SPR, cs = id % SPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + row < nrows && cs * 8 < hv) v = *(const u32x4*)(src + (size_t)(row0 + row) * ld + col0 + cs * 8);
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
OF_DEV s16x8 gload_frag(const bf16_t* __restrict__ base, long ld, long row, long nrows, int col, bool col_valid = true) {
    s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nrows && col_valid) z = *(const s16x8*)(base + (size_t)row * ld + col);
    return z;
}
// two score fragments -> one bf16 MFMA operand (v_cvt_pk_bf16_f32: round-to-nearest-even, two values per instruction)
OF_DEV s16x8 pack8(const f32x4& a, const f32x4& b) {
    const u32x4 r = {of_pack_bf16(a[0], a[1]), of_pack_bf16(a[2], a[3]), of_pack_bf16(b[0], b[1]), of_pack_bf16(b[2], b[3])};
    return __builtin_bit_cast(s16x8, r);
}

// Per-lane byte offsets of the fragment reads, computed ONCE per kernel: inside the key-block loops every LDS address is
// (block base + lane offset) + immediate.  (Recomputing the swizzles per read was ~340 of the ~680 VALU instructions a
// wave spent per 16-query x 64-key step -- the attention kernels were VALU-issue bound, not MFMA- or LDS-bound.)
//   n[ks]: row (lane&15) of a 16-row fragment, k-slot ks*4 + (lane>>4)            -> + t*16 rows as an immediate
//   t[dt]: transpose-read address of row 4*(lane>>4) + ((lane&15)>>2), columns dt*16 + 4*(lane&3)
//                                                                                   -> + (kbase + 16h) rows as an immediate
// (adding a multiple of 16 rows never changes the swizzle term of either image: see img_n_off / img_t_off)
template <int DH>
struct FragOff {
    int n[DH / 32];
    int t[DH / 16];
};
template <int DH>
OF_DEV FragOff<DH> make_frag_off(int lane) {
    FragOff<DH> f;
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int ks = 0; ks < DH / 32; ++ks) f.n[ks] = img_n_off<DH>(i, ks * 4 + g);
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) f.t[dt] = img_t_off<DH>(g * 4 + (i >> 2), dt * 16 + (i & 3) * 4);
    return f;
}
template <int DH>
OF_DEV s16x8 frag_n2(const char* img, int off_ks, int row_base) {
    return *(const s16x8*)(img + off_ks + row_base * (DH * 2));
}
template <int DH, bool SAFE>
OF_DEV s16x8 frag_t2(const char* img, int off_dt, int kbase, int col_base, int lane) {
    if (SAFE) return frag_t<DH, true>(img, kbase, col_base, lane);
    s16x8 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const s16x4 t = of_lds_tr(img + off_dt + (kbase + h * 16) * (DH * 2));
        f[h * 4 + 0] = t[0];
        f[h * 4 + 1] = t[1];
        f[h * 4 + 2] = t[2];
        f[h * 4 + 3] = t[3];
    }
    return f;
}

// LDS-DMA (global_load_lds, no VGPR staging) of rows of a row-major bf16 matrix into a block image: a 1-KiB piece = RPK whole rows.
// The image swizzles of frag_n / frag_t are applied to the SOURCE address because the DMA destination is lane-linear.  Rows past the
// end are clamped to the last row: finite data, their scores are masked / their probabilities 0.  Compact heads (hv < DH: columns
// hv .. DH - 1 of a head do not exist): those units of the image get the row's unit 0 again -- finite data that only ever meets the zero
// columns of a register-loaded operand (q, dO) or lands in output columns that are not stored.
template <int DH, bool TR>
OF_DEV void dma_piece(const bf16_t* __restrict__ src, long ld, long row0, long nrows, int col0, int pc, int lane, char* img, int hv = DH) {
    constexpr int RPK = DH == 128 ? 4 : 8;          // rows per 1-KiB DMA piece
    constexpr int LPR = 64 / RPK;                   // lanes (16-byte units) per row
    const int r = lane / LPR, qpos = lane % LPR;
    const int row = pc * RPK + r;                   // row inside the block image
    long arow = row0 + row;
    if (arow >= nrows) arow = nrows - 1;
    int unit;                                        // 16-byte source unit of the row that belongs at LDS position qpos
    if (!TR) {
        unit = qpos ^ (DH == 128 ? (row & 15) : ((row >> 1) & 7));
    } else {
        const int c = (qpos >> 1) ^ (DH == 128 ? (row & 7) : ((row >> 1) & 3));
        unit = (c << 1) | (qpos & 1);
    }
    if (hv < DH && unit * 8 >= hv) unit = 0;
    of_glds16(src + (size_t)arow * ld + col0 + unit * 8, img + pc * 1024);
}
template <int DH, bool TR, int NW>
OF_DEV void dma_block(const bf16_t* __restrict__ src, long ld, long row0, long nrows, int rows_blk, int col0, int wave, int lane,
                      char* img, int hv = DH) {
    constexpr int RPK = DH == 128 ? 4 : 8;
    for (int pc = wave; pc * RPK < rows_blk; pc += NW) dma_piece<DH, TR>(src, ld, row0, nrows, col0, pc, lane, img, hv);
}

// 16-byte slot `slot` of row `row` of a bf16 image with `row_bytes` per row (a multiple of 256): the 16 rows of a fragment read (same
// slot) hit 16 different slots of a 256-byte bank row
OF_DEV int ximg_off(int row, int slot, int row_bytes) { return row * row_bytes + ((slot & ~15) << 4) + (((slot & 15) ^ (row & 15)) << 4); }

struct Window {
    int lo, hi, uni;
};
// key window of a text position from its text_time (reference helpers.py:196-229; see attention.hip's header): n keys per media,
// T media, Lk keys in all
OF_DEV Window media_window(int tt, int n, int T, int only_immediate, int Lk) {
    Window w{0, 0, 0};
    if (only_immediate) {
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
    if (w.hi > Lk) w.hi = Lk;
    return w;
}
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
    return media_window(p.text_time[batch * p.Lq + row], p.n_per_media, p.T_img, p.only_immediate, p.Lk);
}

// What a lane needs to know about ITS query row inside the key-block loops.  Scores live in the LOG2 domain (scale and
// slope carry a factor log2 e) so that a probability is one v_sub + one bare v_exp_f32:
// key j is visible iff (unsigned)(j - lo) < (unsigned)width; score2 = s * scale + slope * (j - pos), with scale = slope = 0
// for a "uniform" row (every key of the window scores 0: helpers.py:223-229).
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
struct RowCtx {
    int lo, width, pos;
    float scale, slope;
};
OF_DEV RowCtx make_row_ctx(int lo, int hi, int uni, int pos, float scale, float slope) {
    RowCtx rc;
    rc.lo = lo;
    rc.width = hi > lo ? hi - lo : 0;
    rc.pos = pos;
    rc.scale = uni ? 0.f : scale * LOG2E;
    rc.slope = uni ? 0.f : slope * LOG2E;
    return rc;
}
// The visible key range shared by ALL 16 query rows of a wave's tile: [max lo, min hi) (wave-uniform).  A key block inside
// it needs no per-element window test -- the common case (Perceiver / ViT: every block but a ragged tail; causal: every
// block left of the diagonal one; gated cross-attention: the 64 latents of one image).
struct TileRange {
    int max_lo, min_hi;
};
OF_DEV TileRange make_tile_range(int lo, int hi) {
    if (hi <= lo) {          // empty window (zeroed row, or a row past Lq): never "fully visible"
        lo = 0x7fffffff;
        hi = 0;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        const int olo = of_shfl_xor_i(lo, m), ohi = of_shfl_xor_i(hi, m);
        lo = olo > lo ? olo : lo;
        hi = ohi < hi ? ohi : hi;
    }
    return TileRange{of_uniform(lo), of_uniform(hi)};
}
// s[tt][r] = raw dot product of query row (lane&15) with key key0 + 16 tt + 4 (lane>>4) + r  ->  scaled, biased (and, if
// MASKED, windowed: NEG_BIG outside the row's window) log2-domain score; returns the lane's maximum.
template <bool MASKED>
OF_DEV float score_block(f32x4 (&s)[4], const RowCtx& rc, int key0, int g, bool has_alibi) {
    const int jg = key0 + g * 4 - rc.lo;
    float mb = NEG_BIG;
    if (has_alibi) {   // kernel-uniform
        const float bl = rc.slope * (float)(key0 + g * 4 - rc.pos);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sv = s[tt][r] * rc.scale + (rc.slope * (float)(tt * 16 + r) + bl);
                if (MASKED) sv = (unsigned)(jg + tt * 16 + r) < (unsigned)rc.width ? sv : NEG_BIG;
                s[tt][r] = sv;
                mb = of_max(mb, sv);
            }
    } else {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sv = s[tt][r] * rc.scale;
                if (MASKED) sv = (unsigned)(jg + tt * 16 + r) < (unsigned)rc.width ? sv : NEG_BIG;
                s[tt][r] = sv;
                mb = of_max(mb, sv);
            }
    }
    return mb;
}
OF_DEV float score_block_any(f32x4 (&s)[4], const RowCtx& rc, const TileRange& tr, int key0, int nkeys, int g, bool has_alibi) {
    if (key0 >= tr.max_lo && key0 + nkeys <= tr.min_hi) return score_block<false>(s, rc, key0, g, has_alibi);   // wave-uniform
    return score_block<true>(s, rc, key0, g, has_alibi);
}
// One online-softmax step of the forward: log2-domain scores of 16 queries x (16 NSUB) keys -> running max / sum,
// O^T += V^T P^T.  vimg = transpose image of the key block; NSUB = 16-key sub-tiles the block holds (4; 2 for the tail block
// of a resident image).
// NDT = d tiles of O that are computed (compact heads of a compile-time width: the tiles without a column of the head are skipped)
template <int DH, bool SAFE, int NSUB, int NDT = DH / 16>
OF_DEV void softmax_pv(f32x4 (&s)[4], float mb, const char* vimg, const FragOff<DH>& fo, int lane, f32x4 (&acc)[DH / 16],
                       float& m_i, float& l_i) {
    mb = of_rows_max(mb);
    const float m_new = of_max(mb, m_i);
    const float alpha = of_exp2(m_i - m_new);
    const float m_sub = m_new > 0.5f * NEG_BIG ? m_new : 0.f;   // nothing visible yet: exp2(NEG_BIG - 0) = 0 for every masked key
    float rs = 0.f;
#pragma unroll
    for (int tt = 0; tt < NSUB; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = of_exp2(s[tt][r] - m_sub);
            s[tt][r] = pv;
            rs += pv;
        }
    rs = of_rows_sum(rs);
    l_i = l_i * alpha + rs;
    if (of_wave_any(m_new != m_i)) {     // wave-uniform: an unchanged maximum (alpha = 1 in every lane) leaves O alone
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            acc[dt][0] *= alpha;
            acc[dt][1] *= alpha;
            acc[dt][2] *= alpha;
            acc[dt][3] *= alpha;
        }
    }
    m_i = m_new;
#pragma unroll
    for (int s2 = 0; s2 < NSUB / 2; ++s2) {
        const s16x8 pf = pack8(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            acc[dt] = of_mfma(frag_t2<DH, SAFE>(vimg, fo.t[dt], s2 * 32, dt * 16, lane), pf, acc[dt]);
    }
}

// attn_bwd_res.hip: dQ, dK, dV of one (batch, head) in one pass (short self-attention without text_time)
bool attn_bwd_res_fits(const OfAttnArgs& a);
int attn_bwd_res_launch(const OfAttnArgs& a, of_stream_t s);
}  // namespace ofa
