// Single-pass attention backward for short self-attention (Lq, Lk <= 256; causal + ALiBi + key lengths or no mask), gfx950:
// dQ, dK and dV of one (batch, head) from ONE recomputation of S / P, one workgroup per (batch, head).
//
// The two-pass form (attention.hip: of_attn_q_kernel<DH, true> + of_attn_dkv_kernel) recomputes S, P and dP twice and reads q, k, v and
// dO twice (12 tensor passes); the frozen MPT blocks behind every gated block (reference flamingo_lm.py:63-65 -> HF MptAttention:
// 16 heads x 128, L = 256, causal, ALiBi) spend 69 + 69 us per layer there.  Here (8 tensor passes):
//   * 4 waves, one per SIMD, 512 registers each; wave w owns the 16-key MFMA tiles w, w + 4, w + 8, w + 12 (one of every 64-key
//     block: causal work is balanced over the SIMDs): their V fragments stay in registers, K in an LDS image that serves the plain
//     and the transpose fragment reads; dK^T / dV^T accumulate over all query tiles in 256 FIXED accumulation registers
//     (of_accbank64.h; no atomics, fixed order: bit-reproducible), every MFMA is inline asm (see there for why);
//   * queries stream through LDS in tiles of 32 rows (Q and dO, one image each that serves the plain and the transpose fragment reads,
//     loaded through registers, double buffered: tile i + 1 lands while tile i is multiplied); every fragment read serves all key
//     tiles of the wave; K blocks and V fragments arrive one tile before the first query tile that sees them;
//   * phase 1 of a tile, per wave: S = Q K^T and dP = dO V^T of its visible keys x 32 queries, P = exp2(S - lse), dS = P (dP - delta);
//     dV^T += dO^T P, dK^T += Q^T dS; dS (bf16, the operand the dK MFMA takes) also goes to an LDS image [32 queries][256 keys];
//   * phase 2, after one barrier: four dQ^T tiles (64 d x 16 queries) per wave = K^T (transpose reads of the LDS-resident K image) x
//     dS^T over the visible keys; scaled, stored.  delta = rowsum(dO o O) and lse of the NEXT tile are fetched under phase 2.
// Measured (frozen MPT-1B blocks of OF-3B: 32 x 16 heads, behind a 512-MB copy; profiles/r06z*_attn_bwd_single_pass_probe*.jsonl):
// 96-98 us against 139 us for the two passes; per workgroup 46 us = prologue 6 + S / dP + softmax 12 + dV / dK 6 + phase 2 6 + load issue
// 4 + waits 7 + epilogue 4, two rounds of 256.  What it is bound by: instruction issue of ONE wave per SIMD (the 512-register budget the
// 64 keys x 128 x 2 fp32 accumulators force) -- MFMA, softmax VALU and LDS waits of a wave run in series.
// Two barriers per 32 queries.  Same arithmetic as the two-pass kernels (log2-domain scores, bf16 P / dS operands, fp32 sums): dV comes
// out bit-identical; delta is summed in another fp32 order (a few dS round the other way: dK, dQ differ by a bf16 ulp here and there)
// and dQ adds its products over the keys in another order.
#include "attn_core.h"
#include "of_accbank64.h"

namespace {
using namespace ofa;

#if defined(OF_TOOLS_BUILD) && !defined(OF_HOST_EMU)
// tools/libofhip_tools.so only (tools/probes/attn_bwd_single_pass_probe.py): wave 0 of every workgroup records the 100-MHz wall clock at
// entry [0], after the prologue [1], at the start of the epilogue [11] and at exit [10], and sums over the query tiles: issue of the next
// tile's loads [2], S / dP + softmax [3], dS writes + dV / dK MFMAs [4], issue of the statistics loads and of the next K block / V
// fragments [5], wait at the dS barrier [6], phase 2 [7], wait for those loads + LDS writes + statistics + dQ store [8], wait at the tile barrier [9]
__device__ unsigned long long* of_br_stamps = nullptr;
}
extern "C" int of_tools_set_br_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(of_br_stamps), &p, sizeof(p)); }
namespace {
#define BR_STAMP_DECL() unsigned long long br_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, br_last = 0
#define BR_STAMP_AT(i) (br_t[i] = br_last = wall_clock64())
#define BR_STAMP_ADD(i)                              \
    do {                                             \
        const unsigned long long now_ = wall_clock64(); \
        br_t[i] += now_ - br_last;                   \
        br_last = now_;                              \
    } while (0)
#define BR_STAMP_FLUSH()                                                                    \
    do {                                                                                    \
        if (of_br_stamps && of_tid() == 0)                                                  \
            for (int i_ = 0; i_ < 12; ++i_) of_br_stamps[((size_t)of_bid_y() * of_gdim_x() + of_bid_x()) * 12 + i_] = br_t[i_]; \
    } while (0)
#else
#define BR_STAMP_DECL()
#define BR_STAMP_AT(i)
#define BR_STAMP_ADD(i)
#define BR_STAMP_FLUSH()
#endif

constexpr int BR_NW = 4;           // waves; each owns four 16-key tiles of Lk <= 256
constexpr int BR_KT = 4;           // key tiles per wave: tile j of wave w = keys 16 (4 j + w) ..
constexpr int BR_QT = 32;          // query rows per tile
constexpr int BR_DSROW = 512;      // bytes per query row of the dS image (256 keys bf16)

template <int N>
struct BrInt {
    static constexpr int value = N;
};

// HV < DH: compact heads (OfAttnArgs.head_valid = HV, see attention.hip; a COMPILE-TIME width here -- this kernel has no register to spare
// for a run-time one): columns HV .. DH - 1 of a head are zeros in every operand that comes through registers (Q, dO, O, V), a copy of the
// row's first 16 bytes in the LDS-DMA'd K image (it only meets Q's zeros and dQ columns that are not stored), and are not stored.  The
// k-steps of S / dP and the d tiles of dV / dK that hold no column < HV are skipped: head size 80 at DH = 128 runs 3 of 4 k-steps and 5 of 8
// d tiles, and keeps 48 instead of 64 registers of V fragments.  (Phase 2 runs all its d tiles: the waves with d tiles 0 .. 3 set its time
// whatever the others skip, and a conditional skip made hipcc copy accumulators right in front of the asm MFMAs -- tests/test_isa_lint.py.)
template <int DH, int HV = DH>
OF_GLOBAL void OF_BOUNDS(256, 1) of_attn_bwd_res_kernel(OfAttnArgs p) {
    constexpr bool CMP = HV != DH;
    constexpr int NKS = DH / 32, NDT = DH / 16, IMG = 64 * DH * 2, TIMG = BR_QT * DH * 2;
    constexpr int NKSV = (HV + 31) / 32, NDTV = (HV + 15) / 16;   // k-steps / d tiles that hold data
    constexpr int NDTE = (NDTV + 1) & ~1;                          // ... rounded up to the pairs store_row_blocks takes
    constexpr int EPT = DH / 8;                                   // elements of a row per thread in the delta pass (8 threads per row)
    BR_STAMP_DECL();
    BR_STAMP_AT(0);
    char* smem = of_smem();
    char* k_img = smem;                                            // K of the (batch, head): four 64-key images
    char* qd = smem + 4 * IMG;                                     // 2 buffers x {Q tile, dO tile}
    char* ds_img = qd + 4 * TIMG;                                  // dS of the current tile
    float* s_lse = (float*)(ds_img + BR_QT * BR_DSROW);            // [256] log2-domain lse, [256] delta
    float* s_delta = s_lse + 256;
    const int tid = of_tid(), lane = tid & 63, wave = of_uniform(tid >> 6), g = lane >> 4, i16 = lane & 15;
    const int h = of_bid_x();
    const long batch = of_bid_y();
    constexpr int hv = HV;
    const int hc = h * hv;
    const float slope = p.alibi_slopes ? p.alibi_slopes[h] : 0.f;
    const float scale2 = p.scale * LOG2E, slope2 = slope * LOG2E;
    const int key0 = wave * 16 + i16;                              // this lane's key of the wave's tile 0; tile j: + 64 j
    const int koff = p.Lk - p.Lq;                                  // key index aligned with query 0 (causal window, ALiBi origin)
    // keys [0, min(row + cofs, lim)) are visible to query row < Lq (row_window of attn_core.h without text_time)
    int lim = p.Lk;
    if (p.causal && p.kv_len && p.kv_len[batch] < lim) lim = p.kv_len[batch];
    const int cofs = p.causal ? 1 + koff : 0x3fffffff;
    auto row_hi = [&](int row) OF_INLINE_LAMBDA -> int {
        int hi = row + cofs;
        hi = hi > lim ? lim : hi;
        return row < p.Lq ? hi : 0;
    };
    const bf16_t* qb = p.q + (size_t)batch * p.Lq * p.ldq;
    const bf16_t* kb_ptr = p.k + (size_t)batch * p.Lk * p.ldk;
    const bf16_t* vb_ptr = p.v + (size_t)batch * p.Lk * p.ldv;
    const bf16_t* dob = p.dout + (size_t)batch * p.Lq * p.lddo;
    const bf16_t* ob = p.o + (size_t)batch * p.Lq * p.ldo;
    const float* lse_b = p.lse + ((size_t)batch * p.heads + h) * p.Lq;
    // Every image has the "normal" swizzle of attn_core.h (16-byte slot s of row r at slot s ^ f(r)): it serves the 16-byte fragment reads
    // (k = column) AND the transpose reads (k = row; 8-byte pieces, rows 4 g + (lane >> 2) & 3: the 16 rows x 2 slots of one read cover every
    // bank twice) -- one LDS-DMA piece per KiB instead of two.  k-slot ks / d tile dt only flip bits of the swizzled slot index:
    const int fo_n0 = img_n_off<DH>(i16, g);                                                   // row lane & 15, k-slot g: ^ (ks << 6)
    const int fo_t0 = img_n_off<DH>(g * 4 + (i16 >> 2), (i16 & 3) >> 1) + (i16 & 1) * 8;       // transpose read, d tile 0: ^ (dt << 5)
    const int nqt = (p.Lq + BR_QT - 1) / BR_QT;
    const int lk32 = (p.Lk + 31) & ~31;

    // Query tiles travel through registers: an LDS-DMA instruction holds its wave for ~165 ns here (one wave per SIMD: nothing else
    // issues meanwhile; measured 4.6 us per head for 4 pieces per wave and tile), a plain global load does not.  The loads of tile
    // qi + 1 are issued at the top of tile qi and written to the other buffer at its end.  Thread t: 16-byte chunk (c 256 + t) & 15
    // of row (c 256 + t) >> 4 (DH 128).
    constexpr int TCH = TIMG / 16 / 256;                           // 16-byte chunks per thread and image
    struct TileRegs {
        u32x4 q[TCH], d[TCH];
    };
    auto tile_load = [&](int qi, TileRegs& tr, int tid) OF_INLINE_LAMBDA {
#pragma unroll
        for (int c = 0; c < TCH; ++c) {
            const int id = c * 256 + tid, row = id / (DH / 8), cs = id % (DH / 8);
            long arow = (long)qi * BR_QT + row;
            if (arow >= p.Lq) arow = p.Lq - 1;
            // compact heads: a piece beyond the head's columns = zeros.  The load goes to the head's first piece (always there) and
            // tile_store replaces the VALUE ("cond ? *p : 0" here made hipcc select between the global address and a zero parked in
            // scratch; a select on the loaded value here would wait for the load at the top of the tile)
            if (!CMP) {
                tr.q[c] = *(const u32x4*)(qb + (size_t)arow * p.ldq + hc + cs * 8);
                tr.d[c] = *(const u32x4*)(dob + (size_t)arow * p.lddo + hc + cs * 8);
            } else {
                const int co = cs * 8 < hv ? cs * 8 : 0;
                tr.q[c] = *(const u32x4*)(qb + (size_t)arow * p.ldq + hc + co);
                tr.d[c] = *(const u32x4*)(dob + (size_t)arow * p.lddo + hc + co);
            }
        }
    };
    auto tile_store = [&](int qi, const TileRegs& tr, int tid) OF_INLINE_LAMBDA {
        char* base = qd + (qi & 1) * 2 * TIMG;
#pragma unroll
        for (int c = 0; c < TCH; ++c) {
            const int id = c * 256 + tid, row = id / (DH / 8), cs = id % (DH / 8);
            if (!CMP) {
                *(u32x4*)(base + img_n_off<DH>(row, cs)) = tr.q[c];
                *(u32x4*)(base + TIMG + img_n_off<DH>(row, cs)) = tr.d[c];
            } else {
                const bool cv = cs * 8 < hv;
                u32x4 q4, d4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q4[e] = cv ? tr.q[c][e] : 0u;
                    d4[e] = cv ? tr.d[c][e] : 0u;
                }
                *(u32x4*)(base + img_n_off<DH>(row, cs)) = q4;
                *(u32x4*)(base + TIMG + img_n_off<DH>(row, cs)) = d4;
            }
        }
    };
    // delta / lse of tile qi: 8 threads per row, EPT elements each
    struct Stat {
        u32x4 o[EPT / 8], d[EPT / 8];
        float lse;
    };
    auto stat_issue = [&](int qi, Stat& st, int tid) OF_INLINE_LAMBDA {
        const int row = qi * BR_QT + (tid >> 3), c = hc + (tid & 7) * EPT;
        const long r = row < p.Lq ? row : p.Lq - 1;
#pragma unroll
        for (int e = 0; e < EPT / 8; ++e) {
            const bool cv = !CMP || (tid & 7) * EPT + 8 * e < hv;      // see tile_load
            if (!CMP) {
                st.o[e] = *(const u32x4*)(ob + (size_t)r * p.ldo + c + 8 * e);
                st.d[e] = *(const u32x4*)(dob + (size_t)r * p.lddo + c + 8 * e);
            } else {
                const int co = cv ? c + 8 * e : hc;                  // stat_finish drops these products
                st.o[e] = *(const u32x4*)(ob + (size_t)r * p.ldo + co);
                st.d[e] = *(const u32x4*)(dob + (size_t)r * p.lddo + co);
            }
        }
        st.lse = lse_b[r];
    };
    auto stat_finish = [&](int qi, const Stat& st) OF_INLINE_LAMBDA {
        const int row = qi * BR_QT + (tid >> 3);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPT / 8; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned o2 = !CMP || (tid & 7) * EPT + 8 * e < hv ? st.o[e][c] : 0u;      // compact heads: columns that do not exist
                d += of_bf16_to_f32((bf16_t)(o2 & 0xffff)) * of_bf16_to_f32((bf16_t)(st.d[e][c] & 0xffff));
                d += of_bf16_to_f32((bf16_t)(o2 >> 16)) * of_bf16_to_f32((bf16_t)(st.d[e][c] >> 16));
            }
#pragma unroll
        for (int m = 1; m <= 4; m <<= 1) d += of_shfl_xor(d, m);
        if ((tid & 7) == 0) {
            s_delta[row] = row < p.Lq ? d : 0.f;
            s_lse[row] = row < p.Lq ? st.lse * LOG2E : __builtin_inff();
        }
    };

    // K blocks (LDS image, LDS-DMA) and this wave's V fragments (registers) arrive when the first query tile that sees them is one
    // tile away: causal self-attention needs block j from tile 2 j on, so the prologue loads a quarter of K and V and the rest
    // arrives under the tiles before (no mask: everything here).  blocks_for(qi) = 64-key blocks tile qi reads.
    const int nkb = (lk32 + 63) >> 6;
    auto blocks_for = [&](int qi) OF_INLINE_LAMBDA -> int {
        const int last = qi * BR_QT + BR_QT - 1 < p.Lq ? qi * BR_QT + BR_QT - 1 : p.Lq - 1;
        const int nb = (row_hi(last) + 63) >> 6;
        return nb < nkb ? nb : nkb;
    };
    s16x8 vf[BR_KT][NKSV];                                         // K fragments are read from the resident image tile by tile
#pragma unroll
    for (int j = 0; j < BR_KT; ++j)
#pragma unroll
        for (int ks = 0; ks < NKSV; ++ks) vf[j][ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    auto load_blocks = [&](int from, int to, int lane) OF_INLINE_LAMBDA {
#pragma unroll
        for (int j = 0; j < BR_KT; ++j)
            if (j >= from && j < to) {                              // workgroup-uniform
                const int rows_blk = lk32 - j * 64 < 64 ? lk32 - j * 64 : 64;
                dma_block<DH, false, BR_NW>(kb_ptr, p.ldk, (long)j * 64, p.Lk, rows_blk, hc, wave, lane, k_img + (size_t)j * IMG, hv);
#pragma unroll
                for (int ks = 0; ks < NKSV; ++ks)         // row / column from the caller's lane copy: see of_opaque_i
                    vf[j][ks] = gload_frag(vb_ptr, p.ldv, wave * 16 + (lane & 15) + 64 * j, p.Lk, hc + ks * 32 + (lane >> 4) * 8,
                                           !CMP || ks * 32 + (lane >> 4) * 8 < hv);
            }
    };
    // prologue: tile 0, the K blocks / V fragments it needs, its statistics
    TileRegs tr0;
    tile_load(0, tr0, tid);
    int loaded = of_uniform(blocks_for(0));
    load_blocks(0, loaded, lane);
    tile_store(0, tr0, tid);
    {
        Stat st;
        stat_issue(0, st, tid);
        stat_finish(0, st);
    }
    // dV^T / dK^T accumulators: tile (j, dt) of dV at bank slot 2 (j NDT + dt), of dK at the next one -- fixed accumulation registers
    // (of_accbank64.h: as C++ values hipcc shuttles 256 accumulators between register files and spills hundreds)
    of_accbank64_t bank;
    of_accbank64_zero(bank);
    // phase 1: where this lane's dS values go.  Key kappa = 32 s + 16 hh + 4 gk + e sits at 16-byte slot 4 s + gk, element 4 hh + e of
    // its query's row: the order in which the transpose-read K fragment of phase 2 walks the keys of a 32-deep k-step (frag_t, attn_core.h).
    // Tile j of wave w: s = 2 j + (w >> 1), hh = w & 1.
    int ds_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ds_off[r] = ximg_off(4 * g + r, 4 * (wave >> 1) + (i16 >> 2), BR_DSROW) + (4 * (wave & 1) + (i16 & 3)) * 2;
    // phase 2: this wave's dQ^T tiles = (d tiles p2_dt .. p2_dt + P2T - 1) x (query sub-tile p2_qs)
    constexpr int P2T = NDT / 2;                                   // d tiles per wave (two waves per query sub-tile)
    const int p2_dt = P2T * (wave & 1), p2_qs = wave >> 1;
    of_wait_vm<0>();
    of_sync();
    BR_STAMP_AT(1);

    for (int qi = 0; qi < nqt; ++qi) {
        const int q0 = qi * BR_QT;
        const char* q_img = qd + (qi & 1) * 2 * TIMG;
        const char* do_img = q_img + TIMG;
        const bool more = qi + 1 < nqt;
        // per-iteration copies the optimiser cannot hoist address terms out of (of_opaque_i)
        const int lane_i = of_opaque_i(lane), fo_n = of_opaque_i(fo_n0), fo_t = of_opaque_i(fo_t0);
        const int last_row = q0 + BR_QT - 1 < p.Lq ? q0 + BR_QT - 1 : p.Lq - 1;
        const int hi_first = of_uniform(row_hi(q0)), hi_last = of_uniform(row_hi(last_row));   // row_hi grows with the row
        const int ks_end = (hi_last + 31) >> 5;                      // 32-key steps phase 2 reads
        // this wave's tiles j < nvis hold a key some row of the tile sees; only the last of them can be partly masked (the band between
        // hi_first and hi_last is < 32 keys wide: at most three consecutive key tiles, one per wave)
        int nvis = 0;
#pragma unroll
        for (int j = 0; j < BR_KT; ++j) nvis += 16 * (4 * j + wave) < hi_last ? 1 : 0;
        const bool last_full = nvis > 0 && 16 * (4 * (nvis - 1) + wave) + 16 <= hi_first && q0 + BR_QT <= p.Lq;
        u32x2 pk[BR_KT][2], dk2[BR_KT][2];                            // [key tile][query sub-tile]: packed P and dS
        auto phase1a = [&](auto nv_c) OF_INLINE_LAMBDA {              // S, dP, P, dS of the wave's NV visible key tiles
            constexpr int NV = decltype(nv_c)::value;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                f32x4 s[NV], dp[NV];
#pragma unroll
                for (int ks = 0; ks < NKSV; ++ks) {
                    const s16x8 qf = frag_n2<DH>(q_img, fo_n ^ (ks << 6), tt * 16), dof = frag_n2<DH>(do_img, fo_n ^ (ks << 6), tt * 16);
                    // K fragments two key tiles at a time: one LDS latency per pair, not per tile (all four: 8 more registers than there are)
#pragma unroll
                    for (int j0 = 0; j0 < NV; j0 += 2) {
                        s16x8 kf[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            if (j0 + jj < NV) kf[jj] = *(const s16x8*)(k_img + (j0 + jj) * IMG + wave * 16 * DH * 2 + (fo_n ^ (ks << 6)));   // key0 + 64 j
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = j0 + jj;
                            if (j < NV) {
                                if (ks == 0) {
                                    dp[j] = of_mfma_v0(dof, vf[j][ks]);
                                    s[j] = of_mfma_v0(qf, kf[jj]);
                                } else {
                                    of_mfma_v(dof, vf[j][ks], dp[j]);
                                    of_mfma_v(qf, kf[jj], s[j]);
                                }
                            }
                        }
                    }
                    of_accbank64_fence();
                }
#pragma unroll
                for (int j = 0; j < NV; ++j) of_mfma_settle2(s[j], dp[j]);
                // log2-domain score of (query q0 + 16 tt + 4 g + r, key key0 + 64 j)
                const f32x4 l4 = *(const f32x4*)(s_lse + q0 + tt * 16 + g * 4), d4 = *(const f32x4*)(s_delta + q0 + tt * 16 + g * 4);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int my_key = key0 + 64 * j;
                    const float bg = slope2 * (float)(my_key - (q0 + tt * 16 + g * 4 + koff));
                    f32x4 pm, ds;
                    if (j < NV - 1 || last_full) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = of_exp2(s[j][r] * scale2 + (bg - slope2 * (float)r) - l4[r]);
                            pm[r] = pv;
                            ds[r] = pv * (dp[j][r] - d4[r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool valid = my_key < row_hi(q0 + tt * 16 + g * 4 + r);
                            const float pv = valid ? of_exp2(s[j][r] * scale2 + (bg - slope2 * (float)r) - l4[r]) : 0.f;
                            pm[r] = pv;
                            ds[r] = pv * (dp[j][r] - d4[r]);
                        }
                    }
                    pk[j][tt] = u32x2{of_pack_bf16(pm[0], pm[1]), of_pack_bf16(pm[2], pm[3])};
                    dk2[j][tt] = u32x2{of_pack_bf16(ds[0], ds[1]), of_pack_bf16(ds[2], ds[3])};
                    of_accbank64_fence();
                }
            }
        };
        auto phase1b = [&](auto nv_c) OF_INLINE_LAMBDA {              // dS to LDS; dV^T += dO^T P, dK^T += Q^T dS
            constexpr int NV = decltype(nv_c)::value;
            s16x8 pf[NV], dsf[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                pf[j] = __builtin_bit_cast(s16x8, u32x4{pk[j][0][0], pk[j][0][1], pk[j][1][0], pk[j][1][1]});
                dsf[j] = __builtin_bit_cast(s16x8, u32x4{dk2[j][0][0], dk2[j][0][1], dk2[j][1][0], dk2[j][1][1]});
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *(short*)(ds_img + (ds_off[e & 3] ^ ((j & 1) << 7)) + (j >> 1) * 256 + (e >> 2) * 16 * BR_DSROW) = dsf[j][e];
                of_mfma_operands2(pf[j], dsf[j]);
            }
#pragma unroll
            for (int dt = 0; dt < NDTV; ++dt) {
                const s16x8 a_do = frag_t2<DH, false>(do_img, fo_t ^ (dt << 5), 0, dt * 16, lane), a_q = frag_t2<DH, false>(q_img, fo_t ^ (dt << 5), 0, dt * 16, lane);
                of_mfma_guard_nomem();
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    of_accbank64_mfma(bank, 2 * (j * NDT + dt), a_do, pf[j]);
                    of_accbank64_mfma(bank, 2 * (j * NDT + dt) + 1, a_q, dsf[j]);
                }
            }
        };
        if (nvis == 4) phase1a(BrInt<4>{});
        else if (nvis == 3) phase1a(BrInt<3>{});
        else if (nvis == 2) phase1a(BrInt<2>{});
        else if (nvis == 1) phase1a(BrInt<1>{});
        of_accbank64_fence();
        BR_STAMP_ADD(3);
        // the next tile's loads: issued here, in code common to every nvis (a value loaded inside the variants is merged by register
        // copies behind them -- which wait for the load), and behind the register peak of phase 1
        TileRegs trn;
        if (more) tile_load(qi + 1, trn, of_opaque_i(tid));
        BR_STAMP_ADD(2);
        if (nvis == 4) phase1b(BrInt<4>{});
        else if (nvis == 3) phase1b(BrInt<3>{});
        else if (nvis == 2) phase1b(BrInt<2>{});
        else if (nvis == 1) phase1b(BrInt<1>{});
        // a tile no row sees but inside the last k-step phase 2 reads (its other half is visible): zeros
        if (nvis < BR_KT && 16 * (4 * nvis + wave) < ks_end * 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *(short*)(ds_img + (ds_off[e & 3] ^ ((nvis & 1) << 7)) + (nvis >> 1) * 256 + (e >> 2) * 16 * BR_DSROW) = 0;
        }
        of_accbank64_fence();
        BR_STAMP_ADD(4);
        Stat st;
        if (more) {
            stat_issue(qi + 1, st, of_opaque_i(tid));
            // K blocks / V fragments the NEXT tile is the first to see: requested behind phase 1 (which reads vf: requested in front of
            // it, hipcc would make phase 1 wait for them), arrived at the wait that closes this tile
            const int want = of_uniform(blocks_for(qi + 1));
            if (want > loaded) {
                load_blocks(loaded, want, of_opaque_i(lane));
                loaded = want;
            }
        }
        of_accbank64_fence();
        BR_STAMP_ADD(5);
        of_sync();
        BR_STAMP_ADD(6);
        f32x4 acc[P2T];
#pragma unroll
        for (int j = 0; j < P2T; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ks_end > 0) {
            const char* dsr = ds_img + (p2_qs * 16 + i16) * BR_DSROW;
            const int tbase = fo_t ^ (p2_dt << 5);
            for (int ks = 0; ks < ks_end; ++ks) {
                const char* kimg = k_img + (size_t)(ks >> 1) * IMG;
                const int slot = 4 * ks + g;
                const s16x8 b = *(const s16x8*)(dsr + ((slot & ~15) << 4) + (((slot & 15) ^ i16) << 4));
#pragma unroll
                for (int j = 0; j < P2T; ++j) {
                    const s16x8 a = frag_t2<DH, false>(kimg, tbase ^ (j << 5), (ks & 1) * 32, (p2_dt + j) * 16, lane);
                    if (ks == 0) acc[j] = of_mfma_v0(a, b);
                    else of_mfma_v(a, b, acc[j]);
                }
                of_accbank64_fence();
            }
#pragma unroll
            for (int j = 0; j < P2T; j += 2) of_mfma_settle2(acc[j], acc[j + 1]);
        }
        BR_STAMP_ADD(7);
        of_accbank64_fence();
        of_wait_vm0_visible();         // the next tile's images and statistics (and, long since, the previous tile's dq stores)
        if (more) {
            tile_store(qi + 1, trn, of_opaque_i(tid));
            of_accbank64_fence();
            stat_finish(qi + 1, st);
            of_accbank64_fence();
        }
        {
            const int row = q0 + p2_qs * 16 + i16;
            const bool live = row < p.Lq;
            u32x2 o2[P2T];
#pragma unroll
            for (int j = 0; j < P2T; ++j)
                o2[j] = u32x2{of_pack_bf16(acc[j][0] * p.scale, acc[j][1] * p.scale), of_pack_bf16(acc[j][2] * p.scale, acc[j][3] * p.scale)};
            store_row_blocks(p.dq + ((size_t)batch * p.Lq + (live ? row : 0)) * p.lddq + hc + p2_dt * 16, o2, g, live,
                             CMP ? hv - p2_dt * 16 : 0x40000000);
        }
        of_accbank64_fence();
        BR_STAMP_ADD(8);
        of_sync();
        BR_STAMP_ADD(9);
    }
    BR_STAMP_AT(11);
    of_mfma_acc_settle();
#pragma unroll
    for (int j = 0; j < BR_KT; ++j) {
        const int my_key = key0 + 64 * j;
        const bool live = my_key < p.Lk;
        const long row = live ? my_key : 0;
        u32x2 ok[NDTE], ov[NDTE];
#pragma unroll
        for (int dt = 0; dt < NDTE; ++dt) {
            const f32x4 av = of_accbank64_read(bank, 2 * (j * NDT + dt)), ak = of_accbank64_read(bank, 2 * (j * NDT + dt) + 1);
            ok[dt] = u32x2{of_pack_bf16(ak[0] * p.scale, ak[1] * p.scale), of_pack_bf16(ak[2] * p.scale, ak[3] * p.scale)};
            ov[dt] = u32x2{of_pack_bf16(av[0], av[1]), of_pack_bf16(av[2], av[3])};
        }
        store_row_blocks(p.dk + ((size_t)batch * p.Lk + row) * p.lddk + hc, ok, g, live, CMP ? hv : 0x40000000);
        store_row_blocks(p.dv + ((size_t)batch * p.Lk + row) * p.lddv + hc, ov, g, live, CMP ? hv : 0x40000000);
    }
    BR_STAMP_AT(10);
    BR_STAMP_FLUSH();
}

template <int DH>
constexpr size_t br_smem() {
    return (size_t)4 * (64 * DH * 2) + 4 * (BR_QT * DH * 2) + BR_QT * BR_DSROW + 512 * sizeof(float);
}
}  // namespace

namespace ofa {
bool attn_bwd_res_fits(const OfAttnArgs& a) {
    if (a.text_time || a.Lk > 256 || a.Lq > 256) return false;
    if (a.head_dim != 0 && a.head_dim != 64 && a.head_dim != 128) return false;
    // compact heads: the widths this kernel is instantiated for (GPT-NeoX head sizes 80 -- RedPajama-INCITE-3B, Pythia-2.8B -- and 96 --
    // GPT-NeoX-20B -- at head_dim 128); any other width takes the two passes
    const int dh = a.head_dim == 128 ? 128 : 64;
    if (a.head_valid != 0 && a.head_valid != dh && !(dh == 128 && (a.head_valid == 80 || a.head_valid == 96))) return false;
    return true;
}
int attn_bwd_res_launch(const OfAttnArgs& a, of_stream_t s) {
    const of_dim3 grid{(unsigned)a.heads, (unsigned)a.batch, 1};
    if (a.head_dim == 128) {
        if (a.head_valid == 80) return of_launch(of_attn_bwd_res_kernel<128, 80>, grid, BR_NW * 64, br_smem<128>(), s, a);
        if (a.head_valid == 96) return of_launch(of_attn_bwd_res_kernel<128, 96>, grid, BR_NW * 64, br_smem<128>(), s, a);
        return of_launch(of_attn_bwd_res_kernel<128>, grid, BR_NW * 64, br_smem<128>(), s, a);
    }
    return of_launch(of_attn_bwd_res_kernel<64>, grid, BR_NW * 64, br_smem<64>(), s, a);
}
}  // namespace ofa
