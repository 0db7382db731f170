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
//   * the instruction interleave of a phase is pinned with sched_group_barrier: 1 MFMA, then <= 1 LDS read,
//     <= 1 LDS write + 1 global load -- every MFMA gap carries at most ~3 other instructions (the pipe hides ~5).
#include "gemm_tile256.h"

namespace {
using namespace oft;

constexpr int SMEM_W4 = NSLOT * STAGE_BYTES;    // 128 KiB

template <bool AT, bool BT, int EPI>
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
    const of_buf_t gB = of_buf_make(chunk_base<BT>(p.B, p.ldb, n0));
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

    s16x8 fa[2][4], fb[2][4];     // [register buffer][32-row fragment]
    // one operand fragment of k-step ks16 (16 deep) of a stage: slot order = the order the next phase's MFMAs need them
    // (fb0 fa0 fb1 fb2 fb3 fa1 fa2 fa3), one per MFMA gap
    auto read_one = [&](const char* stage, int ks16, int buf, int i) OF_INLINE_LAMBDA {
        const int h = ks16 >> 1, ks = ks16 & 1;
        constexpr int is_a[8] = {0, 1, 0, 0, 0, 1, 1, 1}, idx[8] = {0, 0, 1, 2, 3, 1, 2, 3};
        if (is_a[i]) fa[buf][idx[i]] = frag32<AT>(stage, wm * 128 + idx[i] * 32, h, ks, lane);
        else fb[buf][idx[i]] = frag32<BT>(stage + OPER_BYTES, wn * 128 + idx[i] * 32, h, ks, lane);
    };

    // ---- prologue: stage 0 into slot 0, stage 1 in flight in the staging registers
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
    of_wait_lgkm0();
    of_barrier_raw();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_one(smem, 0, 0, i);

    // One phase = the 16 MFMAs of one k-step (register buffer `buf`), each followed by at most one of: a fragment read
    // of the next k-step into the other buffer (gaps 0-7), or one staging piece -- its LDS write into `nxt` and the
    // re-issue of its global load (gaps 8-13).  of_sched_fence() after every gap pins exactly this interleave.
    auto phase = [&](int buf, const char* rd_stage, int rd_ks16, bool rd, char* nxt, int j0, int nj, bool WR, bool LD) OF_INLINE_LAMBDA {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i >> 2][i & 3] = of_mfma32(fb[buf][i & 3], fa[buf][i >> 2], acc[i >> 2][i & 3]);
            if (rd && i < 8) read_one(rd_stage, rd_ks16, buf ^ 1, i);
            if (i >= 8 && i - 8 < nj) {
                if (WR) store_piece(nxt, j0 + i - 8);
                if (LD) load_piece(j0 + i - 8);
            }
            of_sched_fence();
        }
    };
    // One K stage.  WR: stage d+1 exists (write it), LD: stage d+2 exists (load it).
    auto stage_body = [&](const char* cur, char* nxt, const bool WR, const bool LD) OF_INLINE_LAMBDA {
        phase(0, cur, 1, true, nxt, 0, 6, WR, LD);
        phase(1, cur, 2, true, nxt, 6, 5, WR, LD);
        phase(0, cur, 3, true, nxt, 11, 5, WR, LD);
        if (LD) next_stage_src();
        of_wait_lgkm0();       // own writes of stage d+1 and reads of this slot are done ...
        of_barrier_raw();      // ... and so are everybody else's
        of_sched_fence();
        phase(1, nxt, 0, WR, nxt, 0, 0, false, false);
    };

    int d = 0;
    for (; d + 2 < nd; ++d) stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, true, true);
    if (d + 1 < nd) {
        stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, true, false);
        ++d;
    }
    stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, false, false);
    of_barrier_raw();          // the last stage's k-step-3 fragments were read before its barrier: LDS is idle from here

    // ---------------------------------------------------------------- epilogue, staged through LDS (as gemm_pp.hip)
    // Each wave transposes its accumulators through a private 32-row x 64-column fp32 patch (row pitch 272 B) so that a
    // lane ends up with 8 consecutive n of one row: aux loads and output stores are 16-byte, 8 lanes per row segment.
    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    const float sc = gv * p.alpha;
    float dot = 0.f;
    constexpr int PITCH = 64 * 4 + 16;
    char* patch = smem + wave * (32 * PITCH);
    const int wr_off = (lane & 31) * PITCH + (lane >> 5) * 16;
    const int rd_row = lane >> 3, rd_col = (lane & 7) * 8;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int np = 0; np < 2; ++np) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(f32x4*)(patch + wr_off + (nt * 32 + q * 8) * 4) =
                        f32x4{acc[mt][np * 2 + nt][4 * q], acc[mt][np * 2 + nt][4 * q + 1], acc[mt][np * 2 + nt][4 * q + 2],
                              acc[mt][np * 2 + nt][4 * q + 3]};
            of_wave_sync();
#pragma unroll 1
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + rd_row;
                const f32x4 v0 = *(const f32x4*)(patch + r * PITCH + rd_col * 4), v1 = *(const f32x4*)(patch + r * PITCH + rd_col * 4 + 16);
                const float a8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                ofg::epilogue_row8<EPI>(p, a8, m0 + wm * 128 + mt * 32 + r, n0 + wn * 128 + np * 64 + rd_col, gv, sc, dot);
            }
            of_wave_sync();
        }
    }
    ofg::epilogue_finish<EPI>(p, gv, dot, lane, wave, 4, (float*)(smem + 4 * 32 * PITCH));
}

template <bool AT, bool BT, int EPI>
int launch_w4(const OfGemmArgs& a, of_stream_t s) {
    of_dim3 grid{(unsigned)((a.M / TM) * (a.N / TN)), 1, 1};
    return of_launch(of_gemm_w4_kernel<AT, BT, EPI>, grid, 256, SMEM_W4, s, a);
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
            case OF_EPI_DGELU_DOT: return launch_w4<false, true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_w4<false, true, OF_EPI_SCALE_DOT>(a, s);
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
