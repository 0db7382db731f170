// 128x128 bf16 MFMA GEMM tile, EIGHT waves, operands by LDS-DMA into a four-slot ring (gfx950) -- the kernel of of_gemm for
// tile-aligned shapes that do not fill the chip with 256x256 tiles: the 512-wide projections of the gated blocks (to_q, to_out,
// their dX and dW), the Perceiver's 1024-wide projections, the media to_kv weight gradients.  Same math, LDS images
// (gemm_tile256.h: one 16-KiB half image per operand), layouts and epilogues as the big-tile kernels.
//
// Why: those launches have 64-256 tiles of 128x128.  On the register-staged general kernel (gemm.hip) a CU then holds ONE
// 4-wave workgroup whose every k-tile is a serial global load -> LDS write -> barrier -> LDS read -> MFMA chain: 190-430 TFLOP/s
// (profiles/r02_final_default_gemm_report.jsonl).  A 128x128x64 stage moves 32 KiB for 512 MFMA cycles per SIMD -- twice the
// bytes per FLOP of the 256x256 tile -- so this tile is bound by the CU's LDS-DMA acceptance rate (~27-37 B/clk/CU,
// DESIGN.md 4.1), not by the matrix cores; the structure therefore keeps the DMA queue full and everything else out of its way:
//   * ring of NS = 4 stages x 32 KiB, three stages (96 KiB) in flight per CU: a piece has ~3 stage times to land;
//   * 8 waves = 4 (M) x 2 (N), a wave owns 32 x 64 = two 32x32x16 accumulators (32 registers): two waves per SIMD, so one
//     wave's MFMAs run while its partner is held at the texture unit by a DMA issue;
//   * ONE barrier per stage: iteration kt waits for its own four pieces of stage kt (vmcnt), the barrier publishes everybody's
//     and retires stage kt-1's readers, whose slot then takes stage kt+3 -- issued BEFORE the stage's MFMAs by waves 0-3 and
//     AFTER them by waves 4-7 (the SIMD partners), so that a SIMD's matrix pipe has work while one of its waves queues at
//     the texture unit;
//   * split-K slices (grid.y, weight gradients with a deep K) write fp32 slabs that of_splitk_reduce_kernel combines in a fixed
//     order, exactly like the general kernel.
#include "gemm_tile256.h"

namespace {
using namespace oft;

constexpr int MT = 128, MN = 128;
constexpr int MID_OPER = 128 * DK * 2;          // 16 KiB: one operand image (= a "half" of gemm_tile256.h)
constexpr int MID_STAGE = 2 * MID_OPER;         // 32 KiB
constexpr int MID_NS = 4;                       // ring slots
constexpr int MID_PD = MID_NS - 1;              // stages in flight
constexpr int SMEM_MID = MID_NS * MID_STAGE;    // 128 KiB

// One workgroup's work: tile `tile_id` of `ntiles` (the XCD-aware tile map takes the pair like a block id / grid size), K slice
// `slice` of p.ksplit.  Called by the plain kernel (block ids) and by the batch kernel (several problems in one grid).
template <bool AT, bool BT, int EPI>
OF_DEV void mid_tile(const OfGemmArgs& p, const int tile_id, const int ntiles, const int slice) {
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63;
    const int wave = of_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = p.M / MT, tiles_n = p.N / MN;
    int pm, pn;
    ofg::tile_coords(tile_id, ntiles, tiles_m, tiles_n, pm, pn);
    const int m0 = pm * MT, n0 = pn * MN;

    // K range of this slice (p.ksplit > 1: OF_EPI_ACC_F32 only, slice `slice` of the K stages)
    const int nk_all = p.K / DK;
    const int per = (nk_all + p.ksplit - 1) / p.ksplit;
    const int kt0 = slice * per;
    const int nk = nk_all - kt0 < per ? nk_all - kt0 : per;
    if (nk <= 0) return;

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    // DMA duty of this wave: chunks {wave, wave + 8} of the A image and of the B image = 4 pieces of 1 KiB per stage.
    // Source = wave-uniform base (buffer descriptor) + per-lane 32-bit byte offset (loop invariant) + scalar stage offset.
    const of_buf_t gA = of_buf_make(chunk_base<AT>(p.A, p.lda, m0));
    const of_buf_t gB = of_buf_make(chunk_base<BT>(p.B, p.ldb, n0));
    unsigned offA[2], offB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        offA[j] = 2u * chunk_off<AT>(p.lda, 0, j * 8 + wave, lane);
        offB[j] = 2u * chunk_off<BT>(p.ldb, 0, j * 8 + wave, lane);
    }
    const unsigned stepA = 2u * (AT ? (unsigned)DK * (unsigned)p.lda : (unsigned)DK);
    const unsigned stepB = 2u * (BT ? (unsigned)DK * (unsigned)p.ldb : (unsigned)DK);
    unsigned sA = (unsigned)kt0 * stepA, sB = (unsigned)kt0 * stepB;      // stage the next issue() fetches
    const unsigned smem_u = of_lds_base(smem) + (unsigned)wave * 1024u;      // LDS byte address of this wave's chunk 0 of slot 0
    auto issue = [&](char* slot) OF_INLINE_LAMBDA {
        const unsigned dst = smem_u + (unsigned)(slot - smem);
#pragma unroll
        for (int j = 0; j < 2; ++j) of_buf_load16_lds_at<AT || BT>(gA, offA[j], sA, dst + j * 8 * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) of_buf_load16_lds_at<AT || BT>(gB, offB[j], sB, dst + MID_OPER + j * 8 * 1024);
        sA += stepA;
        sB += stepB;
    };

    const bool late = wave >= 4;     // waves w and w + 4 share a SIMD (see the K loop)
    // epilogue operand of this wave's 32 x 64 group: requested here, lands during the K loop
    const bool sliced = EPI == OF_EPI_ACC_F32 && p.ksplit > 1;
    ofg::AuxPre pre[4];
    ofg::epilogue_group_aux<EPI>(p, m0 + wm * 32, n0 + wn * 64, lane, pre);

    // prologue: stages 0 .. PD-1 in flight
#pragma unroll
    for (int st = 0; st < MID_PD; ++st)
        if (st < nk) issue(smem + st * MID_STAGE);

    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of stage kt have landed: at most the pieces of the (<= 2) later stages stay outstanding
        const int later = nk - 1 - kt;
        if (later >= 2) of_wait_vm<8>();
        else if (later == 1) of_wait_vm<4>();
        else of_wait_vm<0>();
        of_barrier_raw();            // everybody's pieces of stage kt landed; everybody finished reading stage kt - 1
        of_sched_fence();
        // The refill of the slot this barrier freed (stage kt + 3 into the slot of stage kt - 1) may go before or after this
        // stage's MFMAs -- either way it is ~3 stages ahead of its use.  A wave is HELD at the texture unit while its pieces
        // are accepted (all 32 pieces of a stage queue there: ~1200 cycles), so the two waves of a SIMD take opposite orders:
        // waves 0-3 refill first, waves 4-7 compute first -- one of the pair always has MFMAs to issue.
        const bool refill = kt + MID_PD < nk;
        char* refill_slot = smem + ((kt + MID_PD) % MID_NS) * MID_STAGE;
        if (refill && !late) issue(refill_slot);
        const char* stage = smem + (kt % MID_NS) * MID_STAGE;
        s16x8 fa[4], fb[4][2];
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            fa[k16] = frag32<AT>(stage, wm * 32, k16 >> 1, k16 & 1, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) fb[k16][t] = frag32<BT>(stage + MID_OPER, wn * 64 + t * 32, k16 >> 1, k16 & 1, lane);
        }
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = of_mfma32(fb[k16][t], fa[k16], acc[t]);
        of_wait_lgkm0();             // fragment reads done before the next barrier lets the slot be overwritten
        of_sched_fence();
        if (refill && late) issue(refill_slot);
    }
    of_barrier_raw();                // the ring is idle: every wave's last fragment read is behind this barrier

    // ---------------------------------------------------------------- epilogue (one 32 x 64 group per wave)
    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    const float sc = gv * p.alpha;
    float dot = 0.f;
    char* patch = smem + wave * ofg::PATCH_BYTES;
    if (sliced) {                    // this slice's partial tile -> its fp32 slab (combined by of_splitk_reduce_kernel)
        OfGemmArgs q = p;
        q.C = (float*)p.workspace + (size_t)slice * p.M * p.N;
        q.ldc = p.N;
        q.beta = 0.f;
        ofg::epilogue_group<EPI>(q, acc[0], acc[1], patch, m0 + wm * 32, n0 + wn * 64, lane, gv, sc, dot, pre);
        return;
    }
    ofg::epilogue_group<EPI>(p, acc[0], acc[1], patch, m0 + wm * 32, n0 + wn * 64, lane, gv, sc, dot, pre);
    ofg::epilogue_finish<EPI>(p, dot, lane, wave, 8, (float*)(smem + 8 * ofg::PATCH_BYTES), tile_id);
}

template <bool AT, bool BT, int EPI>
OF_GLOBAL void OF_BOUNDS(512, 2) of_gemm_mid_kernel(OfGemmArgs p) {
    mid_tile<AT, BT, EPI>(p, of_bid_x(), of_gdim_x(), of_bid_y());
}

// Several independent problems of one (layout, epilogue) in ONE grid (of_gemm_batch: the three 512-wide weight gradients of a gated
// block -- to_q, to_out, to_kv: off the critical path, 64 tiles each -- were three split-K launches + three reduce launches with the
// chip half empty at every launch boundary).  Workgroup b belongs to the problem whose cumulative range holds b; inside a problem
// the workgroups are (K slice, tile), slices of a tile far apart in the grid.
template <bool AT, bool BT, int EPI>
OF_GLOBAL void OF_BOUNDS(512, 2) of_gemm_mid_batch_kernel(OfGemmBatchArgs m) {
    const int bid = of_bid_x();
    int i = 0, first = 0;
#pragma unroll
    for (int j = 0; j + 1 < OF_GEMM_BATCH_MAX; ++j)
        if (j + 1 < m.n && bid >= m.wg_end[j]) {
            i = j + 1;
            first = m.wg_end[j];
        }
    OfGemmArgs p = m.a[0];            // (selected by uniform branches: a dynamic index into the kernel-argument struct would go through scratch)
    if (i == 1) p = m.a[1];
    if (i == 2) p = m.a[2];
    if (i == 3) p = m.a[3];
    const int tiles = (p.M / MT) * (p.N / MN), local = bid - first;
    mid_tile<AT, BT, EPI>(p, local % tiles, tiles, local / tiles);
}

template <bool AT, bool BT, int EPI>
int launch_mid(const OfGemmArgs& a, of_stream_t s) {
    of_dim3 grid{(unsigned)((a.M / MT) * (a.N / MN)), (unsigned)(a.ksplit > 1 ? a.ksplit : 1), 1};
    const int rc = of_launch(of_gemm_mid_kernel<AT, BT, EPI>, grid, 512, SMEM_MID, s, a);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, (int)grid.x, s);
}
}  // namespace

// the weight-gradient form only (TN, fp32 accumulate): what of_gemm_batch is for
int of_gemm_mid_batch_launch(const OfGemmBatchArgs& m, int total_wg, of_stream_t s) {
    return of_launch(of_gemm_mid_batch_kernel<true, true, OF_EPI_ACC_F32>, of_dim3{(unsigned)total_wg, 1, 1}, 512, SMEM_MID, s, m);
}

// Eligibility: M, N multiples of 128, K a multiple of 64; split-K (a.ksplit > 1) only with OF_EPI_ACC_F32 and fp32 slabs in
// a.workspace (the caller, of_gemm, has checked their size).  Byte offsets are 32-bit: operands up to 4 GiB.
bool of_gemm_mid_eligible(const OfGemmArgs& a) {
    if ((a.M % MT) || (a.N % MN) || (a.K % DK) || a.group_kind || a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
    if (a.ksplit > 1 && (a.epi != OF_EPI_ACC_F32 || !a.workspace)) return false;
    const size_t a_bytes = 2u * (size_t)(a.a_trans ? a.K : a.M) * a.lda, b_bytes = 2u * (size_t)(a.b_trans ? a.K : a.N) * a.ldb;
    if (a_bytes >= (1ull << 32) || b_bytes >= (1ull << 32)) return false;
    const int layout = a.a_trans * 2 + a.b_trans;
    switch (a.epi) {
        case OF_EPI_STORE_BF16:
        case OF_EPI_ACC_F32: return layout == 0 || layout == 1 || layout == 3;
        case OF_EPI_GELU:
        case OF_EPI_GATE_RESID: return layout == 0;
        case OF_EPI_DGELU_DOT:
        case OF_EPI_SCALE_DOT: return layout == 1;
    }
    return false;
}

int of_gemm_mid_try(const OfGemmArgs& a, of_stream_t s) {
    if (!of_gemm_mid_eligible(a)) return OF_E_SHAPE;
    const int layout = a.a_trans * 2 + a.b_trans;
    if (layout == 0) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_mid<false, false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_mid<false, false, OF_EPI_GELU>(a, s);
            case OF_EPI_GATE_RESID: return launch_mid<false, false, OF_EPI_GATE_RESID>(a, s);
            case OF_EPI_ACC_F32: return launch_mid<false, false, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 1) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_mid<false, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch_mid<false, true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_mid<false, true, OF_EPI_SCALE_DOT>(a, s);
            case OF_EPI_ACC_F32: return launch_mid<false, true, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 3) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_mid<true, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_ACC_F32: return launch_mid<true, true, OF_EPI_ACC_F32>(a, s);
        }
    }
    return OF_E_SHAPE;
}
