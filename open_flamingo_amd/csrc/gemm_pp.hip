// 256x256 bf16 MFMA GEMM tile, ping-pong scheduled (gfx950) -- the fast path of of_gemm for tile-aligned shapes
// (M % 256 == 0, N % 256 == 0, K % 64 == 0: every large GEMM of the OpenFlamingo model family at the benchmark
// batch sizes).  Same math, layouts and epilogues as gemm.hip.
//
// Structure
//   * 8 waves = 2 groups (G0 = waves 0-3, G1 = waves 4-7; waves w and w+4 share a SIMD) x 4 N-columns.  A wave owns a
//     128(M) x 64(N) block of the tile = 4x2 v_mfma_f32_32x32x16_bf16 fragments (128 accumulator VGPRs).
//   * Operands travel global -> LDS by DMA (global_load_lds_dwordx4, no VGPR round trip) in 64-deep K stages into two
//     64-KiB slots (A 32 KiB + B 32 KiB each; 128 KiB, one workgroup per CU).
//   * PING-PONG: a stage is four barrier-delimited segments per wave -- L0 C0 L1 C1: Lh reads the 12 operand fragments
//     of k-half h LDS -> VGPR and issues 4 DMA pieces, Ch runs the 16 MFMAs of k-half h.  G1 runs exactly one segment
//     behind G0 (one extra s_barrier up front), so on every SIMD one wave is in an MFMA segment while its partner is
//     in a load segment.  Wall-clock segments of stage p: T0 = G0.L0|G1.C1(p-1), T1 = G0.C0|G1.L0, T2 = G0.L1|G1.C0,
//     T3 = G0.C1|G1.L1.
//   * The measured limiter of this tile on MI355X is the CU's texture path: it accepts one 1-KiB DMA piece per ~38
//     cycles (~27 B/clk/CU, L2-resident operands) and a wave's global_load_lds stalls at issue until accepted.  So
//     DMA pieces are issued ONLY from load segments (a stalled MFMA wave would idle the matrix pipe), 16 per wall
//     segment, one uninterrupted stream with no drain anywhere:
//         T0(p): G0 -> B rows   0-127 of stage p+1        T1(p): G1 -> B rows 128-255 of stage p+1
//         T2(p): G0 -> A rows 128-255 of stage p+1        T3(p): G1 -> A rows   0-127 of stage p+2
//     A rows 0-127 are read only by G0, whose last read of stage p ends with T2(p): that region of the slot is free
//     one segment before the rest, which is what lets the stream run two stages ahead without a third slot.
//   * Ordering (there is no other): after its 4 pieces a wave waits s_waitcnt vmcnt(4) -- its PREVIOUS 4 pieces
//     (issued two segments earlier) have landed -- then lgkmcnt(0) (its fragment reads are done), then the segment's
//     bare s_barrier publishes both facts.  Checked against every reader/writer pair in the K loop's comment.
//   * LDS images are lane-linear for the DMA (dest = wave base + lane*16) and swizzled on the SOURCE address.  Each
//     operand image is two 16-KiB halves (tile rows 0-127 / 128-255), each 16 1-KiB chunks:
//       K-contiguous operand: chunk = 8 rows x 128 B stored [k-half][8 rows][64 B] (lanes 0-31 fetch the first, lanes
//           32-63 the second 64 bytes of the same 8 full lines); 16-B slot s of row r lives at slot s ^ f(r),
//           f = {0,3,2,1}[(r>>2)&3]      -> conflict-free ds_read_b128 reads of 32-row fragments (SQ_LDS_BANK_CONFLICT 0)
//       K-strided operand:    chunk = 4 k-rows x 256 B (128 columns); 32-B piece c of k-row r lives at c ^ ((r&3)<<1)
//           -> conflict-free ds_read_b64_tr_b16 (transpose) reads: a half-wave covers 4 k-rows x 2 pieces
//   * The MFMA is issued operand-swapped (D^T = B^T A^T): a lane owns output row m = lane&31 and 4 consecutive n per
//     accumulator quad, so epilogue loads/stores are 8-byte (bf16) / 16-byte (fp32) vectors (gemm_common.h).
#include "gemm_tile256.h"

namespace {
using namespace oft;

template <bool AT, bool BT, int EPI>
OF_GLOBAL void OF_BOUNDS(512, 2) of_gemm_pp_kernel(OfGemmArgs p) {
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63;
    const int wave = of_uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;   // wm doubles as the ping-pong group
    const int tiles_m = p.M / TM, tiles_n = p.N / TN;
    int pm, pn;
    ofg::tile_coords(of_bid_x(), of_gdim_x(), tiles_m, tiles_n, pm, pn);
    const int m0 = pm * TM, n0 = pn * TN;

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // this wave's DMA duty: 4 chunks of A half (1 - wm) and 4 chunks of B half wm per stage
    const int hfA = 1 - wm, hfB = wm;
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        srcA[j] = chunk_src<AT>(p.A, p.lda, m0, hfA, j * 4 + wn, lane);
        srcB[j] = chunk_src<BT>((BT && p.group_kind == 2) ? (const bf16_t*)p.groups[0] : p.B, p.ldb, n0, hfB, j * 4 + wn, lane);
    }
    // grouped B along K (OfGemmArgs.group_kind 2, K-strided B only): stage s reads rows [s*64, s*64+64) of the virtual
    // [K][N] matrix, which live in weight matrix s / spg at local row (s % spg) * 64
    const int spg = (BT && p.group_kind == 2) ? p.group_extent / DK : 0;
    int b_stage = 0;              // K stage the next issueB will fetch
    const size_t stepA = AT ? (size_t)DK * p.lda : (size_t)DK;
    const size_t stepB = BT ? (size_t)DK * p.ldb : (size_t)DK;
    const int nd = p.K / DK;
    const int offA = hfA * HALF_BYTES + wn * 1024, offB = OPER_BYTES + hfB * HALF_BYTES + wn * 1024;

    auto issueA = [&](char* slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            of_glds16<false>(srcA[j], slot + offA + j * 4096);
            srcA[j] += stepA;
        }
    };
    auto issueB = [&](char* slot) {
        if (spg && b_stage && b_stage % spg == 0) {     // first stage of the next weight matrix: re-base the sources
#pragma unroll
            for (int j = 0; j < 4; ++j) srcB[j] = chunk_src<BT>((const bf16_t*)p.groups[b_stage / spg], p.ldb, n0, hfB, j * 4 + wn, lane);
        }
        ++b_stage;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            of_glds16<false>(srcB[j], slot + offB + j * 4096);
            srcB[j] += stepB;
        }
    };
    s16x8 fa[4][2], fb[2][2];
    auto load_frags = [&](const char* stage, int h, int gi) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < 2; ++t) fb[t][ks] = frag32<BT>(stage + OPER_BYTES, wn * 64 + t * 32, h, ks, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t][ks] = frag32<AT>(stage, wm * 128 + t * 32, h, ks, lane);
        }
    };
    auto compute = [&]() {
        of_sched_fence();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = of_mfma32(fb[nt][ks], fa[mt][ks], acc[mt][nt]);
    };
    // end of a load segment: the 4 pieces issued two segments ago have landed (or everything, if nothing was issued now)
    auto publish = [&](bool issued) {
        if (issued) of_wait_vm<4>();
        else of_wait_vm<0>();
        of_wait_lgkm0();               // this segment's fragment reads
        of_sched_fence();
        of_barrier_raw();
    };

    // prologue: stage 0 complete for everybody (the 8 waves' duties cover all 64 chunks); G1 starts A rows 0-127 of stage 1
    issueB(smem);
    issueA(smem);
    if (wm == 1 && nd > 1) {
        issueA(smem + STAGE_BYTES);
        of_wait_vm<4>();
    } else {
        of_wait_vm<0>();
    }
    of_barrier_raw();
    if (wm == 1) of_barrier_raw();   // stagger: G1 runs one segment behind G0
    // epilogue operand of this wave's first 32 x 64 group: requested here, lands during the K loop -- in registers (16 / 32),
    // or, for the *_DOT epilogues, by DMA in the 32 KiB of LDS behind the ring (no registers at all)
    constexpr bool AUXL = EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT;
    ofg::AuxPre pre0[4];
    if (AUXL) ofg::epilogue_group_aux_dma<false>(p, m0 + wm * 128, n0 + wn * 64, lane, smem + SMEM_PP + wave * ofg::AUX_LDS_BYTES);
    else ofg::epilogue_group_aux<EPI>(p, m0 + wm * 128, n0 + wn * 64, lane, pre0);

    // Stage p lives in slot p&1.  Reader/writer pairs (T = wall segment, see header):
    //   B rows 0-127 of p+1   written G0 T0(p), landed+published end of T2(p); first read T0(p+1).
    //                         overwrites B of p-1, last read G1.L1(p-1) = T3(p-1), published by its barrier.
    //   B rows 128-255 of p+1 written G1 T1(p), published end of T3(p); first read T0(p+1).   overwrites: same as above.
    //   A rows 128-255 of p+1 written G0 T2(p), published end of T0(p+1); first read G1.L0(p+1) = T1(p+1).
    //                         overwrites A 128-255 of p-1, last read G1.L1(p-1) = T3(p-1).
    //   A rows 0-127 of p+2   written G1 T3(p), published end of T1(p+1); first read G0.L0(p+2) = T0(p+2).
    //                         overwrites A 0-127 of p (same slot), last read G0.L1(p) = T2(p), published by its barrier.
    for (int d = 0; d < nd; ++d) {
        const char* stage = smem + (d & 1) * STAGE_BYTES;
        char* other = smem + ((d + 1) & 1) * STAGE_BYTES;
        // ---- L0
        const bool i0 = d + 1 < nd;
        load_frags(stage, 0, d);
        if (i0) issueB(other);
        publish(i0);
        // ---- C0
        compute();
        of_sched_fence();
        of_barrier_raw();
        // ---- L1
        const bool i1 = wm == 0 ? d + 1 < nd : d + 2 < nd;
        load_frags(stage, 1, d);
        if (i1) issueA(wm == 0 ? other : smem + (d & 1) * STAGE_BYTES);
        publish(i1);
        // ---- C1
        compute();
        of_sched_fence();
        of_barrier_raw();
    }
    if (wm == 0) of_barrier_raw();   // balances G1's stagger barrier

    // ---------------------------------------------------------------- epilogue, staged through LDS
    // The ring is idle now (every wave's last fragment read and DMA wait are behind its last barrier).  Each wave sends its
    // four 32 x 64 accumulator groups through a private LDS patch (ofg::epilogue_group).  The aux row segments of group g + 1
    // are requested before group g is processed; group 0's were requested before the K loop (pre0).
    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    const float sc = gv * p.alpha;
    float dot = 0.f;
    char* patch = smem + wave * ofg::PATCH_BYTES;
    if constexpr (AUXL) {
        // aux tiles alternate between two 4-KiB buffers per wave: E behind the ring (group 0 landed there during the K loop: its
        // last wait was vmcnt(0)) and R inside the idle ring.  vmcnt is counted by hand: a group issues 4 stores.
        char* bufE = smem + SMEM_PP + wave * ofg::AUX_LDS_BYTES;
        char* bufR = smem + 8 * ofg::PATCH_BYTES + 256 + wave * ofg::AUX_LDS_BYTES;
        of_wait_vm<0>();
        ofg::epilogue_group_aux_dma<false>(p, m0 + wm * 128 + 32, n0 + wn * 64, lane, bufR);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            // the next group's tile goes where the previous group's was (its reads are consumed: their stores are issued)
            if (mt >= 1 && mt < 3) ofg::epilogue_group_aux_dma<false>(p, m0 + wm * 128 + (mt + 1) * 32, n0 + wn * 64, lane, (mt & 1) ? bufE : bufR);
            if (mt == 1 || mt == 2) of_wait_vm<8>();        // newer than this group's pieces: 4 stores + the next group's 4 pieces
            if (mt == 3) of_wait_vm<4>();                   // ... 4 stores
            ofg::epilogue_group_auxlds<EPI>(p, acc[mt][0], acc[mt][1], patch, (mt & 1) ? bufR : bufE, m0 + wm * 128 + mt * 32, n0 + wn * 64, lane, gv,
                                            sc, dot);
        }
    } else {
        ofg::AuxPre pre[2][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) pre[0][it] = pre0[it];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt < 3) ofg::epilogue_group_aux<EPI>(p, m0 + wm * 128 + (mt + 1) * 32, n0 + wn * 64, lane, pre[(mt + 1) & 1]);
            ofg::epilogue_group<EPI>(p, acc[mt][0], acc[mt][1], patch, m0 + wm * 128 + mt * 32, n0 + wn * 64, lane, gv, sc, dot, pre[mt & 1]);
        }
    }
    ofg::epilogue_finish<EPI>(p, dot, lane, wave, 8, (float*)(smem + 8 * ofg::PATCH_BYTES), of_bid_x());
}

template <bool AT, bool BT, int EPI>
int launch_pp(const OfGemmArgs& a, of_stream_t s) {
    of_dim3 grid{(unsigned)((a.M / TM) * (a.N / TN)), 1, 1};
    // *_DOT epilogues: + 4 KiB per wave behind the ring for the first group's aux tile (160 KiB in all)
    constexpr int smem_bytes = SMEM_PP + ((EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) ? 8 * ofg::AUX_LDS_BYTES : 0);
    const int rc = of_launch(of_gemm_pp_kernel<AT, BT, EPI>, grid, 512, smem_bytes, s, a);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, (int)grid.x, s);
}
}  // namespace

int of_gemm_pp_try(const OfGemmArgs& a, of_stream_t s) {
    if ((a.M % TM) || (a.N % TN) || (a.K % DK)) return OF_E_SHAPE;
    const int layout = a.a_trans * 2 + a.b_trans;
    if (layout == 0) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_pp<false, false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_pp<false, false, OF_EPI_GELU>(a, s);
            case OF_EPI_GATE_RESID: return launch_pp<false, false, OF_EPI_GATE_RESID>(a, s);
            case OF_EPI_ACC_F32: return launch_pp<false, false, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 1) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_pp<false, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch_pp<false, true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_pp<false, true, OF_EPI_SCALE_DOT>(a, s);
            case OF_EPI_ACC_F32: return launch_pp<false, true, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 3) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_pp<true, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_ACC_F32: return launch_pp<true, true, OF_EPI_ACC_F32>(a, s);
        }
    }
    return OF_E_SHAPE;
}
