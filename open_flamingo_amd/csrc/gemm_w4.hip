// 256x256 bf16 MFMA GEMM tile, FOUR waves x (128 x 128) per wave, register-staged operands (gfx950).
// Same math, LDS images (gemm_tile256.h), layouts and epilogues as gemm_pp.hip / gemm.hip.
//
// Why a second big-tile kernel: the 8-wave ping-pong kernel (gemm_pp.hip) is limited by energy per FLOP (it clocks down
// on random operands) and by the CU's LDS-DMA acceptance rate.  This one spends fewer instructions per MFMA:
//   * one wave per SIMD, 512 registers: a wave owns 128(M) x 128(N) = 4x4 v_mfma_f32_32x32x16_bf16 fragments (256
//     accumulator registers); per 64-deep K stage a wave reads 16 + 16 operand fragments for 64 MFMAs (the 128 x 64
//     waves of gemm_pp.hip: 16 + 8 for 32) -> 1/3 fewer LDS reads per FLOP;
//   * operands travel global -> VGPR (global_load_dwordx4, asynchronous: the issuing wave is not held) -> LDS
//     (ds_write_b128, lane-linear 1-KiB pieces, conflict-free) one stage ahead; the loads of stage d+2 are issued as the
//     registers of stage d+1 are written, so a load has a whole stage (~2000 cycles) to land;
//   * ONE barrier per K stage.  Stage d lives in slot d&1.  In iteration d a wave
//         phase 0..2: MFMAs of k-steps 0..2 | reads the fragments of the next k-step | writes stage d+1 into the other
//                     slot (free since barrier d-1: its last readers finished before it) and re-issues the loads
//         s_waitcnt lgkmcnt(0); s_barrier          <- every wave's writes of stage d+1 done, reads of slot d&1 done
//         phase 3:    MFMAs of k-step 3 | reads k-step 0 of stage d+1 (covered by these 16 MFMAs)
//     Fragments are double buffered in registers (k-step j+1 is read while k-step j computes); nothing waits on LDS
//     latency except through the compiler's counted lgkmcnt.
//   * the instruction interleave is pinned (a scheduling fence after every MFMA): each of the 64 MFMA gaps of a stage
//     carries at most ONE LDS instruction -- a fragment read (gaps 0-7 of a phase) or a staging write (gaps 8-15 of
//     phases 0 and 1) -- and the 16 global loads ride with the reads of phases 1 and 2.
#include <type_traits>
#include "gemm_tile256.h"
#include "gemm_w4_epi.h"

namespace {
using namespace oft;

constexpr int SMEM_W4 = NSLOT * STAGE_BYTES;    // 128 KiB

// DMA (product: safe = 7): operands travel global -> LDS directly (buffer_load ... lds, no VGPR round trip, no ds_write)
//   instead of through staging registers (DMA = false, safe = 6: kept as the measured A/B partner and as a second
//   implementation for the race screens).  A(d+2) is issued in phase 3 of iteration d (into the slot that barrier d just
//   freed), B(d+1) in phase 0 of iteration d; one s_waitcnt vmcnt(0) in front of the stage's barrier covers both.  Inside
//   those two phases waves of odd / even index use the odd / even MFMA gaps (half as many waves meet at the texture unit per
//   gap: +1..3 % over the unstaggered placements, profiles/r02_gemm_big_tile_ab.jsonl, r03c_gemm_ab_OF-3B.jsonl).
template <bool AT, bool BT, int EPI, bool DMA>
OF_GLOBAL void OF_BOUNDS(256, 1) of_gemm_w4_kernel(OfGemmArgs p) {
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

    // staging duty of this wave: 1-KiB chunks c = jj*4 + wave (jj = 0..3) of both halves of both operands = 16 pieces.
    // Source = wave-uniform base (advanced per stage on the scalar unit) + per-lane 32-bit byte offset (loop invariant).
    const of_buf_t gA = of_buf_make(chunk_base<AT>(p.A, p.lda, m0));
    // grouped B along N (OfGemmArgs.group_kind 1): this tile's columns belong to weight matrix n0 / extent
    const bf16_t* Bmat = p.B;
    int nB = n0;
    if (!BT && p.group_kind == 1) {
        const int grp = n0 / p.group_extent;
        Bmat = (const bf16_t*)p.groups[grp];
        nB = n0 - grp * p.group_extent;
    }
    const of_buf_t gB = of_buf_make(chunk_base<BT>(Bmat, p.ldb, nB));
    unsigned sA = 0, sB = 0;          // scalar byte offsets of the stage being loaded
    unsigned offA[2][4], offB[2][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            offA[hf][jj] = 2u * chunk_off<AT>(p.lda, hf, jj * 4 + wave, lane);
            offB[hf][jj] = 2u * chunk_off<BT>(p.ldb, hf, jj * 4 + wave, lane);
        }
    const unsigned stepA = 2u * (AT ? (unsigned)DK * (unsigned)p.lda : (unsigned)DK);
    const unsigned stepB = 2u * (BT ? (unsigned)DK * (unsigned)p.ldb : (unsigned)DK);
    const int nd = p.K / DK;
    const int wdst = wave * 1024 + lane * 16;     // + op * OPER_BYTES + hf * HALF_BYTES + jj * 4096

    u32x4 stg[16];       // piece j = op * 8 + hf * 4 + jj
    auto load_piece = [&](int j) OF_INLINE_LAMBDA {
        const int op = j >> 3, hf = (j >> 2) & 1, jj = j & 3;
        if (op == 0) stg[j] = of_buf_load16(gA, offA[hf][jj], sA);
        else stg[j] = of_buf_load16(gB, offB[hf][jj], sB);
    };
    auto next_stage_src = [&]() OF_INLINE_LAMBDA {
        sA += stepA;
        sB += stepB;
    };
    auto store_piece = [&](char* slot, int j) OF_INLINE_LAMBDA {
        const int op = j >> 3, hf = (j >> 2) & 1, jj = j & 3;
        *(u32x4*)(slot + op * OPER_BYTES + hf * HALF_BYTES + jj * 4096 + wdst) = stg[j];
    };

    // DMA variant: piece j of the stage at scalar offsets sA / sB straight into `slot`
    // ahead = 0: the stage at (sA, sB); 1: the stage after it
    const unsigned smem_u = of_lds_base(smem) + (unsigned)wave * 1024u;      // LDS byte address of this wave's piece 0 of slot 0
    auto dma_piece = [&](char* slot, int j, int ahead) OF_INLINE_LAMBDA {
        const int op = j >> 3, hf = (j >> 2) & 1, jj = j & 3;
        const unsigned dst = smem_u + (unsigned)(slot - smem) + (unsigned)(op * OPER_BYTES + hf * HALF_BYTES + jj * 4096);
        if (op == 0) of_buf_load16_lds_at<AT || BT>(gA, offA[hf][jj], sA + (ahead ? stepA : 0u), dst);
        else of_buf_load16_lds_at<AT || BT>(gB, offB[hf][jj], sB + (ahead ? stepB : 0u), dst);
    };
    const int par = wave & 1;
    constexpr bool AUXL = EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT;
    if (AUXL) ofg::epilogue_group_aux_dma<(AT || BT) && DMA>(p, m0 + wm * 128, n0 + wn * 128, lane, smem + SMEM_W4 + wave * ofg::AUX_LDS_BYTES);
    s16x8 fa[2][4], fb[2][4];     // [register buffer][32-row fragment]
    // one operand fragment of k-step ks16 (16 deep) of a stage: slot order = the order the next phase's MFMAs need them
    // (fb0 fa0 fb1 fb2 fb3 fa1 fa2 fa3), one per MFMA gap
    auto read_one = [&](const char* stage, int ks16, int buf, int i) OF_INLINE_LAMBDA {
        const int h = ks16 >> 1, ks = ks16 & 1;
        constexpr int is_a[8] = {0, 1, 0, 0, 0, 1, 1, 1}, idx[8] = {0, 0, 1, 2, 3, 1, 2, 3};
        if (is_a[i]) fa[buf][idx[i]] = frag32<AT>(stage, wm * 128 + idx[i] * 32, h, ks, lane);
        else fb[buf][idx[i]] = frag32<BT>(stage + OPER_BYTES, wn * 128 + idx[i] * 32, h, ks, lane);
    };

    // ---- prologue: stage 0 into slot 0; stage 1 in flight (staging registers / DMA variant: its A half into slot 1)
    if (DMA) {
#pragma unroll
        for (int j = 0; j < 16; ++j) dma_piece(smem, j, 0);
        sA += stepA;                       // (sA, sB) = stage 1 from here on: "the next stage"
        sB += stepB;
        of_wait_vm<0>();
        if (nd > 1) {                      // what phase 3 of "iteration -1" would have issued for stage 1
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((i & 1) == par) dma_piece(smem + STAGE_BYTES, i >> 1, 0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) load_piece(j);
        next_stage_src();
#pragma unroll
        for (int j = 0; j < 16; ++j) store_piece(smem, j);
        if (nd > 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) load_piece(j);
            next_stage_src();
        }
    }
    of_wait_lgkm0();
    of_barrier_raw();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_one(smem, 0, 0, i);

    // The K loop, compiled once per wave parity (odd waves issue their DMA pieces in odd gaps, even waves in even gaps: half as
    // many waves meet at the texture unit per gap) -- the parity is a compile-time constant inside.
    auto main_loop = [&](auto parc) OF_INLINE_LAMBDA {
        constexpr int PARC = decltype(parc)::value;
        // One phase = the 16 MFMAs of one k-step (register buffer `buf`); every MFMA gap carries at most one LDS instruction:
        //   gaps 0-7:  one fragment read of the next k-step into the other buffer (+ one global load, gaps of phases 1 / 2)
        //   gaps 8-15: one staging write (phases 0 / 1)
        // of_sched_fence() after every gap pins exactly this interleave.  Piece j is written in phase j/8 and re-loaded for
        // stage d+2 one phase later (a load has 3.5 phases = 7/8 of a stage to land before its write; the last write is a full
        // phase ahead of the barrier).
        // DMA variant: gaps 8-15 carry one DMA piece instead (dma0 = first piece, into dma_slot).
        auto phase = [&](int buf, const char* rd_stage, int rd_ks16, bool rd, char* nxt, int wr0, int ld0, bool WR, bool LD,
                         char* dma_slot, int win, bool dma_on) OF_INLINE_LAMBDA {
    #pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i >> 2][i & 3] = of_mfma32(fb[buf][i & 3], fa[buf][i >> 2], acc[i >> 2][i & 3]);
                if (i < 8) {
                    if (rd) read_one(rd_stage, rd_ks16, buf ^ 1, i);
                    if (!DMA && LD && ld0 >= 0) load_piece(ld0 + i);
                } else if (!DMA && WR && wr0 >= 0) {
                    store_piece(nxt, wr0 + i - 8);
                }
                if (DMA && dma_on && win >= 0 && win < 2 && (i & 1) == PARC) dma_piece(dma_slot, win * 8 + (i >> 1), win == 0);
                of_sched_fence();
            }
        };
        // One K stage.  WR: stage d+1 exists (write it), LD: stage d+2 exists (load it).
        auto stage_body = [&](char* cur, char* nxt, const bool WR, const bool LD) OF_INLINE_LAMBDA {
            phase(0, cur, 1, true, nxt, 0, -1, WR, LD, nxt, 1, WR);           // DMA: window phase 1 of stage d+1 -> nxt
            phase(1, cur, 2, true, nxt, 8, 0, WR, LD, nxt, 2, WR);            // DMA: window phase 2 of stage d+1 -> nxt
            phase(0, cur, 3, true, nxt, -1, 8, WR, LD, nullptr, -1, false);
            if (!DMA && LD) next_stage_src();
            if (DMA) of_wait_vm<0>();      // own DMA pieces of stage d+1 have landed ...
            of_wait_lgkm0();       // own writes of stage d+1 and reads of this slot are done ...
            of_barrier_raw();      // ... and so are everybody else's
            of_sched_fence();
            phase(1, nxt, 0, WR, nxt, -1, -1, false, false, cur, 0, LD);      // DMA: window phase 0 of stage d+2 -> cur (free since the barrier)
            if (DMA) {
                sA += stepA;
                sB += stepB;
            }
        };

        int d = 0;
        // K-contiguous operands only (measured: NT -1..3 %, layouts with transposed-fragment reads +1..2 %): steady state two
        // stages per trip, both slot addresses compile-time constants (fragment reads and M0 values become immediates instead
        // of per-stage address arithmetic bunched into the first MFMA gaps of a phase)
        if (!(AT || BT))
            for (; d + 3 < nd; d += 2) {
                stage_body(smem, smem + STAGE_BYTES, true, true);
                stage_body(smem + STAGE_BYTES, smem, true, true);
            }
        for (; d + 2 < nd; ++d) stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, true, true);
        if (d + 1 < nd) {
            stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, true, false);
            ++d;
        }
        stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, false, false);
    };
    if (DMA && par) main_loop(std::integral_constant<int, 1>{});
    else main_loop(std::integral_constant<int, 0>{});
    of_barrier_raw();          // the last stage's k-step-3 fragments were read before its barrier: LDS is idle from here

    oft::w4_epilogue<EPI, (AT || BT) && DMA>(p, acc, smem, SMEM_W4, m0, n0, wm, wn, wave, lane);
}

template <bool AT, bool BT, int EPI>
int launch_w4(const OfGemmArgs& a, of_stream_t s) {
    of_dim3 grid{(unsigned)((a.M / TM) * (a.N / TN)), 1, 1};
    // *_DOT epilogues: + 4 KiB per wave behind the ring for the first group's aux tile
    constexpr int smem_bytes = SMEM_W4 + ((EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) ? 4 * ofg::AUX_LDS_BYTES : 0);
    if (a.safe == 6) return of_launch(of_gemm_w4_kernel<AT, BT, EPI, false>, grid, 256, smem_bytes, s, a);   // register staged
    return of_launch(of_gemm_w4_kernel<AT, BT, EPI, true>, grid, 256, smem_bytes, s, a);                        // LDS-DMA (product)
}
template <bool AT, bool BT, int EPI>
int launch_w4_dot(const OfGemmArgs& a, of_stream_t s) {
    const int rc = launch_w4<AT, BT, EPI>(a, s);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, (a.M / TM) * (a.N / TN), s);
}
}  // namespace

int of_gemm_w4_try(const OfGemmArgs& a, of_stream_t s) {
    if ((a.M % TM) || (a.N % TN) || (a.K % DK)) return OF_E_SHAPE;
    const int layout = a.a_trans * 2 + a.b_trans;
    if (layout == 0) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4<false, false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_w4<false, false, OF_EPI_GELU>(a, s);
            case OF_EPI_GATE_RESID: return launch_w4<false, false, OF_EPI_GATE_RESID>(a, s);
            case OF_EPI_ACC_F32: return launch_w4<false, false, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 1) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4<false, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch_w4_dot<false, true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_w4_dot<false, true, OF_EPI_SCALE_DOT>(a, s);
            case OF_EPI_ACC_F32: return launch_w4<false, true, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 3) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4<true, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_ACC_F32: return launch_w4<true, true, OF_EPI_ACC_F32>(a, s);
        }
    }
    return OF_E_SHAPE;
}
