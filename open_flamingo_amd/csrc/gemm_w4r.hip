// 256x256 bf16 MFMA GEMM tile, FOUR waves x (128 x 128) per wave, operands by LDS-DMA into a RING of four 32-deep half
// stages (gfx950).  Same math, epilogues (gemm_w4_epi.h) and fragment order as gemm_w4.hip; what differs is how far ahead of
// its use an operand piece is requested and how evenly the requests reach the texture unit.
//
// gemm_w4.hip keeps two 64-KiB slots of 64-deep stages: a slot is refilled in ONE burst of 16 pieces per wave during the
// two phases after the barrier that freed it (4 waves x 1 piece per 64 cycles against an acceptance rate of ~1 per 35-38) and
// the pieces then have 0.5-1 stage (1000-2000 cycles) to land before a vmcnt(0).  Here the same 128 KiB are four slots of
// 32 KiB (A half stage 16 KiB + B half stage 16 KiB, 32 deep):
//   * half stage d lives in slot d & 3; its barrier sits in the MIDDLE of half stage d - 1's MFMAs (after the phase that read
//     the last fragments of d - 1 ... see the loop), so at barrier d the slot of d is read out and takes half stage d + 4:
//     three half stages (96 KiB, 24 pieces per wave) are always in flight or landed, a piece has 2.5-3.5 half stages
//     (2500-3500 cycles) to land, and the wait is a counted vmcnt(16), never 0;
//   * the 8 pieces a wave owes per half stage are spread over its 32 MFMA gaps, one every fourth gap, and the four waves use
//     different gaps: the CU asks for one 1-KiB piece per MFMA gap (32 cycles) -- a steady stream at the average rate
//     instead of bursts the issuing waves queue behind;
//   * two barriers per 64 of K instead of one (a barrier costs the single wave of a SIMD a few cycles; the fragment reads of
//     the next phase are already in registers when it is reached).
// LDS images of a half stage (the swizzles of gemm_tile256.h on 32-deep chunks):
//   K-strided operand:    [128-column half][32 k-rows][256 B], 32-B piece c of k-row r at c ^ ((r & 3) << 1); a 1-KiB DMA piece
//                         = 4 k-rows x 256 B (full 128-B lines);
//   K-contiguous operand: [128-row half][128 rows][64 B], 16-B slot s of row r at s ^ f(r); a 1-KiB DMA piece = 16 rows x 64 B
//                         (half lines: the other half of each line is the next half stage's piece).
#include <type_traits>
#include "gemm_tile256.h"
#include "gemm_w4_epi.h"

namespace {
using namespace oft;

constexpr int HK = 32;                          // K depth of a ring slot
constexpr int H_OPER = 256 * HK * 2;            // 16 KiB per operand per half stage
constexpr int H_HALF = H_OPER / 2;              // 8 KiB: tile rows (or columns) 0-127 / 128-255
constexpr int H_SLOT = 2 * H_OPER;              // 32 KiB
constexpr int H_NS = 4;
constexpr int SMEM_W4R = H_NS * H_SLOT;         // 128 KiB

// per-lane element offset of 1-KiB chunk c (0..7) of half hf of one operand's half stage at k0 = 0
template <bool TR>
OF_DEV unsigned hchunk_off(long ld, int hf, int c, int lane) {
    if (!TR) {
        const int row = hf * 128 + c * 16 + (lane >> 2);
        const int lslot = (lane & 3) ^ fN(row);
        return (unsigned)(row * ld + lslot * 8);
    } else {
        const int krow = c * 4 + (lane >> 4);
        const int pc = (lane & 15) >> 1, half16 = lane & 1;
        const int col = hf * 128 + ((pc ^ fT(krow)) << 4) + half16 * 8;
        return (unsigned)(krow * ld + col);
    }
}

// this lane's 16-byte piece of a 32-row operand fragment: k-step ks (16 deep) of the half stage
template <bool TR>
OF_DEV s16x8 hfrag32(const char* oper, int row_base, int ks, int lane) {
    if (!TR) {
        const int row = row_base + (lane & 31);
        const int slot = ks * 2 + (lane >> 5);
        return *(const s16x8*)(oper + (row >> 7) * H_HALF + (row & 127) * 64 + ((slot ^ fN(row)) << 4));
    } else {
        const int q = lane >> 4, i = lane & 15;
        s16x8 f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int krow = ks * 16 + (q >> 1) * 8 + hh * 4 + (i >> 2);
            const int col = row_base + (q & 1) * 16 + (i & 3) * 4;
            const int cw = col & 127;
            s16x4 t = of_lds_tr(oper + (col >> 7) * H_HALF + krow * 256 + ((((cw >> 4)) ^ fT(krow)) << 5) + ((cw & 15) << 1));
            f[hh * 4 + 0] = t[0];
            f[hh * 4 + 1] = t[1];
            f[hh * 4 + 2] = t[2];
            f[hh * 4 + 3] = t[3];
        }
        return f;
    }
}

template <bool AT, bool BT, int EPI>
OF_GLOBAL void OF_BOUNDS(256, 1) of_gemm_w4r_kernel(OfGemmArgs p) {
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63;
    const int wave = of_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = p.M / TM, tiles_n = p.N / TN;
    int pm, pn;
    ofg::tile_coords(of_bid_x(), of_gdim_x(), tiles_m, tiles_n, pm, pn);
    const int m0 = pm * TM, n0 = pn * TN;

    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // DMA duty of this wave per half stage: chunks q = jj*4 + wave (jj = 0..3; half q >> 3, chunk q & 7) of the A image and of
    // the B image = 8 pieces of 1 KiB.  Source = wave-uniform base + per-lane 32-bit byte offset (loop invariant) + scalar
    // offset of the half stage (advanced on the scalar unit).
    const of_buf_t gA = of_buf_make(chunk_base<AT>(p.A, p.lda, m0));
    const bf16_t* Bmat = p.B;
    int nB = n0;
    if (!BT && p.group_kind == 1) {       // grouped B along N: this tile's columns belong to weight matrix n0 / extent
        const int grp = n0 / p.group_extent;
        Bmat = (const bf16_t*)p.groups[grp];
        nB = n0 - grp * p.group_extent;
    }
    const of_buf_t gB = of_buf_make(chunk_base<BT>(Bmat, p.ldb, nB));
    unsigned offA[4], offB[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int q = jj * 4 + wave;
        offA[jj] = 2u * hchunk_off<AT>(p.lda, q >> 3, q & 7, lane);
        offB[jj] = 2u * hchunk_off<BT>(p.ldb, q >> 3, q & 7, lane);
    }
    const unsigned stepA = 2u * (AT ? (unsigned)HK * (unsigned)p.lda : (unsigned)HK);
    const unsigned stepB = 2u * (BT ? (unsigned)HK * (unsigned)p.ldb : (unsigned)HK);
    unsigned sA = 0, sB = 0;              // scalar byte offsets of the next A / B half stage to request
    const int nh = p.K / HK;              // half stages (K % 64 == 0: even, >= 2)
    auto dma_a = [&](char* slot, int jj) OF_INLINE_LAMBDA {
        of_buf_load16_lds<AT || BT>(gA, offA[jj], sA, slot + (jj * 4 + wave) * 1024);
    };
    auto dma_b = [&](char* slot, int jj) OF_INLINE_LAMBDA {
        of_buf_load16_lds<AT || BT>(gB, offB[jj], sB, slot + H_OPER + (jj * 4 + wave) * 1024);
    };

    constexpr bool AUXL = EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT;
    if (AUXL) ofg::epilogue_group_aux_dma<AT || BT>(p, m0 + wm * 128, n0 + wn * 128, lane, smem + SMEM_W4R + wave * ofg::AUX_LDS_BYTES);

    s16x8 fa[2][4], fb[2][4];     // [register buffer][32-row fragment]
    // one operand fragment of k-step ks of a half stage, in the order the next phase's MFMAs need them
    // (fb0 fa0 fb1 fb2 fb3 fa1 fa2 fa3), one per MFMA gap
    auto read_one = [&](const char* slot, int ks, int buf, int i) OF_INLINE_LAMBDA {
        constexpr int is_a[8] = {0, 1, 0, 0, 0, 1, 1, 1}, idx[8] = {0, 0, 1, 2, 3, 1, 2, 3};
        if (is_a[i]) fa[buf][idx[i]] = hfrag32<AT>(slot, wm * 128 + idx[i] * 32, ks, lane);
        else fb[buf][idx[i]] = hfrag32<BT>(slot + H_OPER, wn * 128 + idx[i] * 32, ks, lane);
    };

    // ---- prologue: half stages 0, 1, 2 and the A image of 3 requested (the loop's first phase asks for B of 3)
#pragma unroll
    for (int h = 0; h < 4; ++h)
        if (h < nh) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) dma_a(smem + h * H_SLOT, jj);
            sA += stepA;
            if (h < 3) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) dma_b(smem + h * H_SLOT, jj);
                sB += stepB;
            }
        }
    // half stage 0 has landed: the groups of four requested after B(0) may be outstanding
    if (nh >= 4) of_wait_vm<20>();
    else if (nh == 3) of_wait_vm<16>();
    else of_wait_vm<8>();
    of_barrier_raw();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_one(smem, 0, 0, i);

    // The K loop, compiled once per wave: wave w uses MFMA gaps 4j + w for its DMA pieces (a compile-time constant inside).
    auto main_loop = [&](auto wavec) OF_INLINE_LAMBDA {
        constexpr int WV = decltype(wavec)::value;
        // Half stage d (slot d & 3) = two phases of 16 MFMAs.
        //   phase 0: MFMAs of k-step 0 | reads k-step 1 of d | requests B(d + 3) into slot (d + 3) & 3 (free since barrier d - 1)
        //   s_waitcnt vmcnt(pieces of d + 2, d + 3); lgkmcnt(0); s_barrier     <- d + 1 landed for everybody, slot d read out
        //   phase 1: MFMAs of k-step 1 | reads k-step 0 of d + 1 | requests A(d + 4) into slot d & 3
        auto half_stage = [&](int d, const bool LDA, const bool LDB, const int later, const bool RD) OF_INLINE_LAMBDA {
            char* cur = smem + (d & 3) * H_SLOT;
            char* nxt = smem + ((d + 1) & 3) * H_SLOT;
            char* prv = smem + ((d + 3) & 3) * H_SLOT;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i >> 2][i & 3] = of_mfma32(fb[0][i & 3], fa[0][i >> 2], acc[i >> 2][i & 3]);
                if (i < 8) read_one(cur, 1, 1, i);
#ifdef OF_RING_PLACE_NOREAD      // A/B variant (tools/ab builds only): pieces only in the gaps without fragment reads
                if (LDB && i >= 8 && ((i - 8) & 1) == (WV & 1)) dma_b(prv, (i - 8) >> 1);
#else
                if (LDB && (i & 3) == WV) dma_b(prv, i >> 2);
#endif
                of_sched_fence();
            }
            if (LDB) sB += stepB;
            if (later >= 2) of_wait_vm<16>();
            else if (later == 1) of_wait_vm<8>();
            else of_wait_vm<0>();
            of_wait_lgkm0();
            of_barrier_raw();
            of_sched_fence();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i >> 2][i & 3] = of_mfma32(fb[1][i & 3], fa[1][i >> 2], acc[i >> 2][i & 3]);
                if (i < 8 && RD) read_one(nxt, 0, 0, i);
#ifdef OF_RING_PLACE_NOREAD
                if (LDA && i >= 8 && ((i - 8) & 1) == (WV & 1)) dma_a(cur, (i - 8) >> 1);
#else
                if (LDA && (i & 3) == WV) dma_a(cur, i >> 2);
#endif
                of_sched_fence();
            }
            if (LDA) sA += stepA;
        };
        int d = 0;
        for (; d + 4 < nh; ++d) half_stage(d, true, true, 2, true);
        if (d + 3 < nh) {
            half_stage(d, false, true, 2, true);
            ++d;
        }
        if (d + 2 < nh) {
            half_stage(d, false, false, 1, true);
            ++d;
        }
        if (d + 1 < nh) {
            half_stage(d, false, false, 0, true);
            ++d;
        }
        half_stage(d, false, false, 0, false);
    };
    if (wave == 0) main_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) main_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) main_loop(std::integral_constant<int, 2>{});
    else main_loop(std::integral_constant<int, 3>{});
    of_barrier_raw();          // the ring is idle from here

    oft::w4_epilogue<EPI, AT || BT>(p, acc, smem, SMEM_W4R, m0, n0, wm, wn, wave, lane);
}

template <bool AT, bool BT, int EPI>
int launch_w4r(const OfGemmArgs& a, of_stream_t s) {
    of_dim3 grid{(unsigned)((a.M / TM) * (a.N / TN)), 1, 1};
    // *_DOT epilogues: + 4 KiB per wave behind the ring for the first group's aux tile
    constexpr int smem_bytes = SMEM_W4R + ((EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) ? 4 * ofg::AUX_LDS_BYTES : 0);
    const int rc = of_launch(of_gemm_w4r_kernel<AT, BT, EPI>, grid, 256, smem_bytes, s, a);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, (int)grid.x, s);
}
}  // namespace

int of_gemm_w4r_try(const OfGemmArgs& a, of_stream_t s) {
    if ((a.M % TM) || (a.N % TN) || (a.K % DK)) return OF_E_SHAPE;
    const int layout = a.a_trans * 2 + a.b_trans;
    if (layout == 0) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4r<false, false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_w4r<false, false, OF_EPI_GELU>(a, s);
            case OF_EPI_GATE_RESID: return launch_w4r<false, false, OF_EPI_GATE_RESID>(a, s);
            case OF_EPI_ACC_F32: return launch_w4r<false, false, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 1) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4r<false, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch_w4r<false, true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_w4r<false, true, OF_EPI_SCALE_DOT>(a, s);
            case OF_EPI_ACC_F32: return launch_w4r<false, true, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 3) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4r<true, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_ACC_F32: return launch_w4r<true, true, OF_EPI_ACC_F32>(a, s);
        }
    }
    return OF_E_SHAPE;
}
