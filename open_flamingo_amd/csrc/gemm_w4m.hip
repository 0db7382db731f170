// 256x256 bf16 MFMA GEMM tile, FOUR waves x (128 x 128) per wave, LDS-DMA operands -- the 4-wave kernel of gemm_w4.hip on
// v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16 (gfx950).
//
// Why a second instruction shape: with every CU busy the big tile's K loop is bound by the chip's POWER budget, not by cycles
// (1.54 us per stage at ~1.56 GHz on 256 CUs against 0.93 us at ~2.4 GHz on 128).  On random operands a bare stream of
// 16x16x32 MFMAs sustains 2030 TFLOP/s where the same FLOPs as 32x32x16 sustain 1760 (tools/probes/mfma_power_probe.hip,
// profiles/r03q_*: both reach 2450 on zeros, i.e. both issue at full rate -- the difference is energy per FLOP: a 32x32x16
// reads and writes 4x the accumulator registers per operand register it reads).  The vendor library's 256x256x64 kernel uses
// the same shape.
//
// Structure (everything not listed is gemm_w4.hip's: tile mapping, LDS-DMA pieces, the two 64-KiB slots, epilogues):
//   * a wave owns 128 x 128 = 8 x 8 accumulators of 16 x 16 (256 registers); one 64-deep stage = FOUR phases of 32 MFMAs:
//     (k-step of 32, half of the wave's rows).  In registers: the 8 B fragments of both k-steps and two sets of 4 A fragments
//     (96 registers; both operands of a whole k-step double buffered would be 128 and spills);
//   * stage d lives in slot d & 1.  Phases 0-2 read the rest of the slot (A rows 64-127 of k-step 0, k-step 1), ONE barrier,
//     phase 3 reads the first fragments of stage d + 1 from the other slot -- the schedule of gemm_w4.hip: A of stage d + 2 is
//     requested in phase 3 (into the slot the barrier freed), B of stage d + 1 in phase 0, eight pieces over the 32 MFMA gaps
//     of the phase, odd waves two gaps after the even ones; one vmcnt(0) in front of the barrier covers both;
//   * round 5, for operands that come from HBM instead of the Infinity Cache (a launch inside a train step): the
//     epilogues that keep nothing in LDS during the K loop have a THIRD image of B behind the two slots (W4M_B3: B of stage d + 2
//     requested in phase 0 of stage d, vmcnt(8) in front of the barrier), and launches of_gemm selected itself walk K rotated per
//     XCD (w4m_rotation);
//   * the MFMAs are inline asm accumulating in place (of_mfma_acc: through the builtin hipcc shuttled the 64 small accumulators
//     between AGPRs and VGPRs, 390 v_accvgpr_* per 128 MFMAs).  The price: the compiler's hazard recognizer does not know them,
//     so every K stage opens with of_mfma_acc_guard() and tests/test_isa_lint.py checks the cross-compiled ISA for a VALU write
//     of an MFMA operand right in front of an MFMA (found on hardware: DESIGN.md 4.1);
//   * LDS images: the K-contiguous image of gemm_tile256.h serves 16-row fragments conflict-free as it is (a lane's 16-byte
//     slot is (k-step, lane >> 4)); the K-strided image gets one more swizzle bit (piece ^ ((k-row >> 3) & 1)): the four
//     16-lane groups of a transposed read sit 8 k-rows apart in the same 32-byte column piece.
#include <type_traits>
#include "gemm_tile256.h"
#include "gemm_w4_epi.h"

// K rotation of a tile's stage loop (see the kernel): the workgroups of XCD x (block id & 7) start at stage x * (stages / 8).  The 32
// tiles an XCD runs at a time share A and B panels through its L2 and must walk K together; the eight XCDs need not -- measured on
// operands that are NOT in the Infinity Cache (a launch behind a streaming pass, as in a train step: tools/probes/interleaved_gemm_probe.py,
// profiles/r05q_*): NT 8192 x 2048 x 8192 241 -> 218 us, NN 235 -> 226, NT K = 2048 238 -> 230, hot-loop times unchanged; a rotation per
// tile loses the L2 sharing (hot 199 -> 239 us), a window of rotations inside the XCD adds nothing.  K-strided operands (TN) gain
// nothing: there a K offset moves the address by whole rows.
#ifdef OF_W4M_NO_ROTATION          // tools/ab builds only (tools/build_ab_variant.sh): the other arm of the same-box A/B
OF_DEV int w4m_rotation(int, int) { return 0; }
#else
OF_DEV int w4m_rotation(int bid, int nd) { return (bid & 7) * (nd >> 3); }
#endif

#if defined(OF_TOOLS_BUILD) && !defined(OF_HOST_EMU)
// tools/libofhip_tools.so only (tools/probes/tile_phase_probe.py): where a tile's time goes.  Wave 0 of every workgroup stamps
// the 100-MHz wall clock at kernel entry, after the prologue (stage 0 landed), after the K loop and after its last epilogue
// instruction, plus the hardware id (XCC / SE / CU) it ran on: 8 x u64 per workgroup into a buffer the tool registers.
__device__ unsigned long long* of_tools_stamps = nullptr;
extern "C" int of_tools_set_stamp_buffer(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(of_tools_stamps), &p, sizeof(p));
}
__device__ int of_w4m_map_knob = 0;          // tile walk of the launch: ofg::tile_coords_knob
extern "C" int of_tools_set_w4m_map_knob(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(of_w4m_map_knob), &v, sizeof(v)); }
#define OF_W4M_TILE_COORDS(vt, nt, tm, tn, pm, pn) ofg::tile_coords_knob(vt, nt, tm, tn, pm, pn, of_w4m_map_knob)
// knob bits 12..15 = rotation rule, bits 16..23 = its multiplier m (0: 1)
OF_DEV int w4m_rotation_knob(int vt, int bid, int pm, int pn, int tiles_m, int tiles_n, int nd) {
    const int mode = (of_w4m_map_knob >> 12) & 15;
    int m = (of_w4m_map_knob >> 16) & 255;
    if (!m) m = 1;
    int r = 0;
    if (mode == 0) return w4m_rotation(bid, nd);          // the product's rule
    if (mode == 15) return 0;                             // none (rounds 1-4)
    if (mode == 1) r = vt * m;                            // per tile
    else if (mode == 2) r = (bid & 7) * (nd / 8) * m;     // per XCD
    else if (mode == 3) r = pm * m;                       // per tile row
    else if (mode == 4) r = pn * m;                       // per tile column
    else if (mode == 5) r = (pm + pn) * m;
    else if (mode == 6) r = (bid >> 3) * m;               // per position inside the XCD's run
    else if (mode == 7) r = (bid & 7) * (nd / 8) + ((bid >> 3) % m);          // per XCD + a window of m stages inside it (L2 keeps that much history)
    else if (mode == 8) r = (bid & 7) * (nd / 8) + (pm % m);
    else if (mode == 9) r = (bid & 7) * (nd / 8) + (pn % m) ;
    return (int)((unsigned)r % (unsigned)nd);
}
#define OF_W4M_ROTATION(vt, bid, pm, pn, tm, tn, nd) w4m_rotation_knob(vt, bid, pm, pn, tm, tn, nd)
#define OF_STAMP(i) (of_stamp_t[i] = wall_clock64())          /* wave-uniform: stays in SGPRs until the one store block at the end */
#define OF_STAMP_DECL() unsigned long long of_stamp_t[5] = {0, 0, 0, 0, 0}
#define OF_STAMP_FLUSH()                                                                                              \
    do {                                                                                                              \
        if (of_tools_stamps) {                                                                                        \
            of_wait_vm<0>();          /* probe only: every store of the wave acknowledged (the product wave just ends) */ \
            OF_STAMP(4);                                                                                              \
            if (of_tid() == 0) {                                                                                      \
                unsigned hw;          /* workgroups go round-robin over the 8 XCDs: XCC = block id & 7 */             \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                     \
                unsigned long long* o = of_tools_stamps + (size_t)of_bid_x() * 8;                                     \
                for (int i_ = 0; i_ < 5; ++i_) o[i_] = of_stamp_t[i_];                                                \
                o[7] = ((unsigned long long)(of_bid_x() & 7) << 32) | hw;                                             \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#else
#define OF_STAMP(i) ((void)0)
#define OF_STAMP_DECL() ((void)0)
#define OF_STAMP_FLUSH() ((void)0)
#define OF_W4M_TILE_COORDS(vt, nt, tm, tn, pm, pn) ofg::tile_coords(vt, nt, tm, tn, pm, pn)
#define OF_W4M_ROTATION(vt, bid, pm, pn, tm, tn, nd) w4m_rotation(bid, nd)
#endif

namespace {
using namespace oft;

constexpr int SMEM_W4M = NSLOT * STAGE_BYTES;    // 128 KiB
// GATE_RESID: the 32 KiB behind the ring either hold the first residual tile of every wave, requested before the K loop (round 3), or the
// third image of B -- then the epilogue requests that tile itself (gemm_w4_epi.h: PRE = false).
#ifdef OF_W4M_RESID_PRE      // tools/ab builds only
constexpr bool W4M_B3_RESID = false;
#else
constexpr bool W4M_B3_RESID = true;
#endif
template <int EPI>
struct W4M_B3 {
#ifdef OF_W4M_NO_B3          // tools/ab builds only: the other arm of the same-box A/B
    static constexpr bool value = false;
#else
    static constexpr bool value = EPI == OF_EPI_STORE_BF16 || EPI == OF_EPI_GELU || EPI == OF_EPI_ACC_F32 || (EPI == OF_EPI_GATE_RESID && W4M_B3_RESID);
#endif
};

// ---- stream-K layout of the optional workspace region behind the *_DOT partials (of_gemm fills p.sk_grid, gemm.hip):
//   [ flags: one int per workgroup, zeroed by of_gemm in front of the launch ][ slabs: one 256 x 256 fp32 partial tile per workgroup ]
constexpr size_t SK_SLAB_BYTES = (size_t)TM * TN * 4;
OF_HOSTDEV size_t sk_flags_bytes(int grid) { return ((size_t)grid * 4 + 255) & ~(size_t)255; }
OF_HOSTDEV size_t sk_dot_bytes(const OfGemmArgs& a) { return of_gemm_dot_bytes(a); }

// Schedule.  The grid is G workgroups.  Classic launch (p.sk_grid == 0): G = number of tiles, workgroup b owns tile b.  Stream-K
// launch (p.sk_grid = G): every workgroup takes the same share of the launch's (tile, K-stage) units --
//   * rounds = tiles / G whole tiles, b, b + G, b + 2G, ... (no fix-up: exactly the classic launch's work per workgroup);
//   * the remaining r = tiles - rounds * G tiles are r * nd units in (tile, stage) order, workgroup b takes units
//     [r nd b / G, r nd (b + 1) / G): less than one tile's worth, so at most the TAIL of one tile and the HEAD of the next.
// A tile shared between workgroups is finished by the one that holds its LAST K stage (the owner); the others write their fp32
// partial tile to their slab and raise their flag.  A workgroup walks its unit range BACKWARDS -- head of the later tile first
// (never an owner unless it is the whole tile), tail of the earlier tile last -- so partial tiles are published early, owners
// wait late and only for workgroups with LOWER ids: no deadlock even when fewer than G workgroups are resident (the hardware
// dispatches in id order; a workgroup's first segment never waits).  The owner adds the partials in ascending K order, on top
// of its own: the same bits for the same (shape, G), whatever the timing.
// SK = false: the classic launch compiled on its own -- one tile per workgroup, none of the schedule arithmetic, flags or partial
// tile code (as one kernel the bookkeeping cost every tile of every launch 0.7-1 us of prologue: profiles/r04g_tile_phase_probe.jsonl).
template <bool AT, bool BT, int EPI, bool SK>
OF_GLOBAL void OF_BOUNDS(256, 1) of_gemm_w4m_kernel(OfGemmArgs p) {
    constexpr bool ASMD = AT || BT;       // LDS-DMA form (of_platform.h): inline asm wherever a transposed-fragment read follows
    OF_STAMP_DECL();
    OF_STAMP(0);
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63;
    const int wave = of_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = p.M / TM, tiles_n = p.N / TN, ntiles = tiles_m * tiles_n;
    const int nd_all = p.K / DK;
    const int G = of_gdim_x(), bid = of_bid_x();
    // (32-bit arithmetic: r * nd * G < 2^31 for every K < 2 M; the classic launch skips the divisions altogether -- a 64-bit
    // division is ~150 instructions, and this code sits in front of every tile)
    int rounds = 1;
    unsigned rem_units = 0, r0 = 0, ue = 0;      // [r0, ue): the not yet processed part of this workgroup's remainder units
    if constexpr (SK) {
        rounds = (int)((unsigned)ntiles / (unsigned)G);
        rem_units = (unsigned)(ntiles - rounds * G) * (unsigned)nd_all;
        r0 = rem_units * (unsigned)bid / (unsigned)G;
        ue = rem_units * (unsigned)(bid + 1) / (unsigned)G;
    }
    int* sk_flags = (int*)((char*)p.workspace + sk_dot_bytes(p));
    float* sk_slabs = (float*)((char*)sk_flags + sk_flags_bytes(G));

    f32x4 acc[8][8];      // [16-row block of M][16-column block of N]
    unsigned offA[2][4], offB[2][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            offA[hf][jj] = 2u * mchunk_off<AT>(p.lda, hf, jj * 4 + wave, lane);
            offB[hf][jj] = 2u * mchunk_off<BT>(p.ldb, hf, jj * 4 + wave, lane);
        }
    const unsigned stepA = 2u * (AT ? (unsigned)DK * (unsigned)p.lda : (unsigned)DK);
    const unsigned stepB = 2u * (BT ? (unsigned)DK * (unsigned)p.ldb : (unsigned)DK);
    const unsigned smem_u = of_lds_base(smem) + (unsigned)wave * 1024u;
    constexpr bool AUXL = EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT;
    // B3: a THIRD image of B behind the two slots (every epilogue but the *_DOT ones can leave 32 KiB of the CU's 160 free there): B
    // of stage d + 2 is requested in phase 0 of stage d -- six to seven phases before the barrier that needs it instead of two to three.
    // A's images stay two: requested in phase 3, three to four phases ahead.  For operands that come from HBM, not from the Infinity
    // Cache.
    constexpr bool B3 = W4M_B3<EPI>::value;
    constexpr unsigned B_OFF0 = OPER_BYTES, B_OFF1 = STAGE_BYTES + OPER_BYTES, B_OFF2 = B3 ? (unsigned)SMEM_W4M : B_OFF0;

    for (int seg = 0;; ++seg) {
        // ---- this segment: tile `vt` (virtual block id for the XCD-aware tile map), K stages [s0, s0 + nd)
        int vt, s0, nd;
        if constexpr (!SK) {
            if (seg > 0) break;
            vt = bid;
            s0 = 0;
            nd = nd_all;
        } else if (seg < rounds) {
            vt = bid + seg * G;
            s0 = 0;
            nd = nd_all;
        } else {
            if (ue <= r0) break;
            const unsigned t = (ue - 1) / (unsigned)nd_all;
            const unsigned ts = t * (unsigned)nd_all, us = ts > r0 ? ts : r0;
            vt = rounds * G + (int)t;
            s0 = (int)(us - ts);
            nd = (int)(ue - us);
            ue = us;
        }
        const bool last_k = !SK || s0 + nd == nd_all;       // this workgroup finishes the tile (epilogue), else it publishes a partial tile
        int pm, pn;
        OF_W4M_TILE_COORDS(vt, ntiles, tiles_m, tiles_n, pm, pn);
        const int m0 = pm * TM, n0 = pn * TN;
        if (SK && seg > 0) of_barrier_raw();          // the previous segment's epilogue is done with the ring

#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;

        // DMA duty of this wave: 1-KiB chunks c = jj*4 + wave (jj = 0..3) of both halves of both operands = 16 pieces per stage
        const of_buf_t gA = of_buf_make(chunk_base<AT>(p.A, p.lda, m0));
        const bf16_t* Bmat = p.B;
        int nB = n0;
        if (!BT && p.group_kind == 1) {       // grouped B along N: this tile's columns belong to weight matrix n0 / extent
            const int grp = n0 / p.group_extent;
            Bmat = (const bf16_t*)of_uniform_ptr(p.groups[grp]);      // loaded behind the previous segment's stores: a vector load
            nB = n0 - grp * p.group_extent;
        }
        const of_buf_t gB = of_buf_make(chunk_base<BT>(Bmat, p.ldb, nB));
        // K rotation: a workgroup that holds ALL K stages of its tile starts at stage `rot` and wraps around (same products, another
        // order of the fp32 additions).  Without it every workgroup of the launch reads K offset 64 d of its panels at the same time;
        // with a power-of-two row pitch those addresses differ in high bits only and land on the same few HBM channels.
        int rot = 0;
        // (only launches of_gemm selected itself: a kernel forced through OfGemmArgs.safe keeps stage order 0, 1, 2, ... -- the order of
        // the general kernel, which the race screens of tests/test_gpu_kernels.py compare with bit for bit)
        if (p.safe == 0 && (!SK || (s0 == 0 && nd == nd_all))) rot = OF_W4M_ROTATION(vt, bid, pm, pn, tiles_m, tiles_n, nd_all);
        int kidx2 = rot + s0 + 1;                                                    // stage index of (sA2, sB2)
        if (kidx2 >= nd_all) kidx2 -= nd_all;
        unsigned sA = (unsigned)(rot + s0) * stepA, sB = (unsigned)(rot + s0) * stepB;          // scalar byte offsets of the next stage to request
        unsigned sA2 = (unsigned)kidx2 * stepA, sB2 = (unsigned)kidx2 * stepB;       // ... and of the one after it (wraps to stage 0 behind the last)
        auto next_stage = [&]() OF_INLINE_LAMBDA {
            sA = sA2;
            sB = sB2;
            ++kidx2;
            sA2 += stepA;
            sB2 += stepB;
            if (kidx2 == nd_all) {
                kidx2 = 0;
                sA2 = 0u;
                sB2 = 0u;
            }
        };
        // piece j (0..15 = op * 8 + hf * 4 + jj) of the stage at (sA, sB) -- ahead = 1: of the stage after it -- into the slot at
        // byte offset slot_off
        auto dma_piece = [&](unsigned a_off, unsigned b_off, int j, int ahead) OF_INLINE_LAMBDA {
            const int op = j >> 3, hf = (j >> 2) & 1, jj = j & 3;
            const unsigned dst = smem_u + (op == 0 ? a_off : b_off) + (unsigned)(hf * HALF_BYTES + jj * 4096);
            if (op == 0) of_buf_load16_lds_at<ASMD>(gA, offA[hf][jj], ahead ? sA2 : sA, dst);
            else of_buf_load16_lds_at<ASMD>(gB, offB[hf][jj], ahead ? sB2 : sB, dst);
        };

        if (last_k) {      // the epilogue's first aux / residual tile travels during the K loop (gemm_w4_epi.h)
            if (AUXL) ofg::epilogue_group_aux_dma<ASMD>(p, m0 + wm * 128, n0 + wn * 128, lane, smem + SMEM_W4M + wave * ofg::AUX_LDS_BYTES);
            if (EPI == OF_EPI_GATE_RESID && !B3) ofg::epilogue_group_resid_dma<ASMD>(p, m0 + wm * 128, n0 + wn * 128, lane, smem + SMEM_W4M + wave * ofg::RESID_LDS_BYTES);
        }

        s16x8 fa[2][4], fb[2][8];     // [register buffer][16-row fragment]: B of a whole k-step, A of half of the wave's rows
        // Fragment reads, one per call.  A stage is four phases (k-step ks = phase >> 1, row half ah = phase & 1 of the wave's 128
        // rows); fb[ks] holds the 8 B fragments of k-step ks, fa[phase & 1] the 4 A fragments of (ks, ah).
        auto read_a = [&](const char* stage, int ks, int ah, int buf, int r) OF_INLINE_LAMBDA {
            fa[buf][r] = mfrag16<AT>(stage, wm * 128 + ah * 64 + r * 16, ks, lane);
        };
        auto read_b = [&](const char* b_img, int ks, int r) OF_INLINE_LAMBDA {
            fb[ks][r] = mfrag16<BT>(b_img, wn * 128 + r * 16, ks, lane);
        };
        // the 12 fragments a phase that starts a k-step needs, in the order of first use: fb0 fa0 fb1 .. fb7 fa1 fa2 fa3
        auto read_kstep = [&](const char* stage, const char* b_img, int ks, int abuf, int r) OF_INLINE_LAMBDA {
            if (r == 1) read_a(stage, ks, 0, abuf, 0);
            else if (r < 9) read_b(b_img, ks, r == 0 ? 0 : r - 1);
            else read_a(stage, ks, 0, abuf, r - 8);
        };

        // ---- prologue: stage 0 landed in slot 0; A of stage 1 requested into slot 1 (what phase 3 of "stage -1" would have done)
#pragma unroll
        for (int j = 0; j < 16; ++j) dma_piece(0u, B_OFF0, j, 0);
        next_stage();                      // (sA, sB) = stage 1 from here on: "the next stage"
        if (!B3) {
            of_wait_vm<0>();
            if (nd > 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dma_piece(STAGE_BYTES, 0u, j, 0);
            }
        } else if (nd > 1) {               // three B images: all of stage 1 is requested here (its B in what phase 0 of "stage -1" would have done)
#pragma unroll
            for (int j = 0; j < 16; ++j) dma_piece(STAGE_BYTES, B_OFF1, j, 0);
            of_wait_vm<16>();              // stage 0 has landed, stage 1 is in flight
        } else {
            of_wait_vm<0>();
        }
        of_barrier_raw();
        OF_STAMP(1);
#pragma unroll
        for (int r = 0; r < 12; ++r) read_kstep(smem, smem + B_OFF0, 0, 0, r);

        // The K loop, compiled once per wave parity (odd waves request their pieces two MFMA gaps after the even ones).
        auto main_loop = [&](auto parc) OF_INLINE_LAMBDA {
            constexpr int PARC = decltype(parc)::value;
            // One phase = 32 MFMAs: B fragments fb[ks] x A fragments fa[ph & 1] (rows 64 (ph & 1) .. of the wave's 128).  Its MFMA gaps
            // carry the fragment reads of the NEXT phase (4 or 12, from the gap after the previous read on, every other gap) and,
            // with `dma`, eight LDS-DMA pieces (gaps 4j + 2 PARC).
            auto phase = [&](int ph, const char* rd_stage, const char* rd_b, bool rd, unsigned dma_a, unsigned dma_b, int dma0, bool dma, int ahead) OF_INLINE_LAMBDA {
                const int ks = ph >> 1, ah = ph & 1;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    of_mfma_acc(fb[ks][i & 7], fa[ph & 1][i >> 3], acc[ah * 4 + (i >> 3)][i & 7]);
                    if (rd && !(i & 1)) {
                        const int r = i >> 1;
                        if (ph == 0 || ph == 2) {          // next phase: same k-step, the other row half
                            if (r < 4) read_a(rd_stage, ks, 1, (ph + 1) & 1, r);
                        } else if (r < 12) {               // next phase starts a k-step (ph 1: k-step 1 of this stage; ph 3: k-step 0 of the next)
                            read_kstep(rd_stage, rd_b, ph == 1 ? 1 : 0, (ph + 1) & 1, r);
                        }
                    }
                    if (dma && (i & 3) == 2 * PARC) dma_piece(dma_a, dma_b, dma0 + (i >> 2), ahead);
                    of_sched_fence();
                }
            };
            // One K stage: A in slot `cur`, B in the image at b_cur.  WR: stage d + 1 exists, LD: stage d + 2 exists.  (sA, sB) = offsets
            // of stage d + 1, (sA2, sB2) of stage d + 2.  b_nxt: B image of stage d + 1; b_ld (B3): the third image, free since the
            // barrier of stage d - 1, takes B of stage d + 2.
            auto stage_body = [&](char* cur, char* nxt, unsigned b_cur, unsigned b_nxt, unsigned b_ld, const bool WR, const bool LD) OF_INLINE_LAMBDA {
                const unsigned cur_u = (unsigned)(cur - smem);
                of_mfma_acc_guard();       // fragments may have been moved between registers on the way into this stage (of_platform.h)
                if (B3) phase(0, cur, smem + b_cur, true, 0u, b_ld, 8, LD, 1);          // + B of stage d + 2 -> the third image
                else phase(0, cur, smem + b_cur, true, 0u, b_nxt, 8, WR, 0);           // + B of stage d + 1 -> nxt (its A went there in phase 3 of stage d - 1)
                phase(1, cur, smem + b_cur, true, 0u, 0u, 0, false, 0);
                phase(2, cur, smem + b_cur, true, 0u, 0u, 0, false, 0);
                // own pieces of stage d + 1 have landed (B3: the eight pieces of B of stage d + 2 behind them may still fly -- loads retire
                // in order) ...
                if (B3 && LD) of_wait_vm<8>();
                else of_wait_vm<0>();
                of_wait_lgkm0();       // ... own reads of this slot are done ...
                of_barrier_raw();      // ... and so are everybody else's
                of_sched_fence();
                phase(3, nxt, smem + b_nxt, WR, cur_u, 0u, 0, LD, 1);            // + A of stage d + 2 -> cur (free since the barrier)
                next_stage();
            };
            int d = 0;
            unsigned bc = B_OFF0, bn = B_OFF1, bl = B_OFF2;          // B images of stage d, d + 1, d + 2 (two of them alternate unless B3)
            auto rotate_b = [&]() OF_INLINE_LAMBDA {
                const unsigned t = bc;
                bc = bn;
                bn = B3 ? bl : t;
                bl = t;
            };
            // (gemm_w4.hip unrolls its NT steady state by two stages for compile-time slot addresses; here that measured +-1 % and
            // is not done.  It is how the VALU-write -> asm-MFMA hazard of of_mfma_acc_guard() was found: the unrolled build moved stage
            // 0's fragments between registers right in front of the loop and lost half a k-step -- DESIGN.md 4.1.)
            for (; d + 2 < nd; ++d) {
                stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, bc, bn, bl, true, true);
                rotate_b();
            }
            if (d + 1 < nd) {
                stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, bc, bn, bl, true, false);
                rotate_b();
                ++d;
            }
            stage_body(smem + (d & 1) * STAGE_BYTES, smem + ((d + 1) & 1) * STAGE_BYTES, bc, bn, bl, false, false);
        };
        if (wave & 1) main_loop(std::integral_constant<int, 1>{});
        else main_loop(std::integral_constant<int, 0>{});
        of_mfma_acc_settle();
        of_barrier_raw();          // the ring is idle from here
        OF_STAMP(2);

        if (SK && !last_k) {
            // ---- partial tile (stream-K): raw accumulators -> this workgroup's slab, fragment-major so that every store is one
            // contiguous KiB per wave; then the flag.  Every lane waits for its (system-scope) stores in front of the barrier, the
            // flag goes up behind it.
            // (buffer-descriptor addressing: wave-uniform base + scalar fragment offset + one per-lane offset -- with plain pointers
            // every fragment, 4 KiB from the last, needs a 64-bit address of its own in VGPRs, which this kernel does not have)
            const of_buf_t slab = of_buf_make(of_uniform_ptr(sk_slabs + (size_t)bid * (TM * TN)));
#pragma unroll
            for (int a = 0; a < 8; ++a) {
#pragma unroll
                for (int b = 0; b < 8; ++b) of_buf_store16_sys(slab, (unsigned)tid * 16u, (unsigned)(a * 8 + b) * 4096u, __builtin_bit_cast(u32x4, acc[a][b]));
                of_sched_fence();
            }
            of_wait_vm<0>();       // system-scope stores: acknowledged = visible to every XCD
            of_sync();
            if (tid == 0) of_flag_publish(sk_flags + bid, 1);
            continue;
        }
        // (fewer remainder units than workgroups: some workgroups hold none -- they publish nothing and are skipped)
        auto sk_has_units = [&](int w) OF_INLINE_LAMBDA { return rem_units * (unsigned)w / (unsigned)G != rem_units * (unsigned)(w + 1) / (unsigned)G; };
        // ---- owner of a shared tile: the partial tiles of K stages [0, s0) -- workgroups sk_first .. bid - 1, ascending K -- are
        // added group by group on the way through the LDS patch (acc_to_patch below); here: wait until all of them are published
        int sk_first = bid;
        if (SK && s0 > 0) {
            const unsigned ts = (unsigned)(vt - rounds * G) * (unsigned)nd_all;
            sk_first = (int)(((ts + 1) * (unsigned)G + rem_units - 1) / rem_units) - 1;      // the workgroup whose range holds unit ts
            if (tid == 0)
                for (int w = sk_first; w < bid; ++w)
                    if (sk_has_units(w)) of_flag_await(sk_flags + w, 1);
            of_sync();
        }
        // (Prefetching the next group's partial fragments into registers was tried: it pushed hipcc into spilling ACCUMULATORS to
        // scratch right behind the last MFMA -- in front of of_mfma_acc_settle(), which tests/test_isa_lint.py caught.  The eight
        // 16-byte loads of a group therefore pay one global-load latency per group and contributor: ~20 us per shared tile.)
        auto acc_to_patch = [&](int g, char* patch) OF_INLINE_LAMBDA {
            const int mt = g >> 1, np = g & 1;
            const f32x4 t[2][4] = {{acc[2 * mt][4 * np], acc[2 * mt][4 * np + 1], acc[2 * mt][4 * np + 2], acc[2 * mt][4 * np + 3]},
                                   {acc[2 * mt + 1][4 * np], acc[2 * mt + 1][4 * np + 1], acc[2 * mt + 1][4 * np + 2], acc[2 * mt + 1][4 * np + 3]}};
            ofg::patch_write16(patch, t, lane);
            if constexpr (SK) {
                for (int w = sk_first; w < bid; ++w) {          // ascending K = ascending workgroup id (no iteration unless the tile is shared)
                    if (!sk_has_units(w)) continue;
                    const of_buf_t slab = of_buf_make(of_uniform_ptr(sk_slabs + (size_t)w * (TM * TN)));
                    f32x4 q[2][4];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            q[i][j] = __builtin_bit_cast(f32x4, of_buf_load16_sys(slab, (unsigned)tid * 16u, (unsigned)((2 * mt + i) * 8 + 4 * np + j) * 4096u));
                    ofg::patch_add16(patch, q, lane);
                }
            }
        };
        // gate-gradient partial of this TILE: slot = its position in the (m-major) tile grid, whichever workgroup finishes it
        w4_epilogue_with<EPI, ASMD, decltype(acc_to_patch), true, !(EPI == OF_EPI_GATE_RESID && B3)>(p, acc_to_patch, smem, SMEM_W4M, m0, n0, wm, wn, wave, lane, pm * tiles_n + pn);
        OF_STAMP(3);
    }
    OF_STAMP_FLUSH();
}

template <bool AT, bool BT, int EPI>
int launch_w4m(const OfGemmArgs& a, of_stream_t s) {
    const int ntiles = (a.M / TM) * (a.N / TN);
    of_dim3 grid{(unsigned)(a.sk_grid > 0 ? a.sk_grid : ntiles), 1, 1};
    // *_DOT epilogues: + 4 KiB per wave behind the ring for the first group's aux tile
    // GATE_RESID: + 8 KiB per wave for the first group's residual tile (160 KiB in all: the CU's whole LDS)
    // plain-store / GELU / fp32 epilogues: + 32 KiB for the third image of B (also 160 KiB)
    constexpr int smem_bytes = SMEM_W4M + ((EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) ? 4 * ofg::AUX_LDS_BYTES
                                           : EPI == OF_EPI_GATE_RESID ? 4 * ofg::RESID_LDS_BYTES : W4M_B3<EPI>::value ? OPER_BYTES : 0);
    if (a.sk_grid > 0 && ntiles % a.sk_grid) {       // tiles will be shared: every workgroup's flag starts at 0
        const int rc = of_memset_async((char*)a.workspace + sk_dot_bytes(a), 0, sk_flags_bytes(a.sk_grid), s);
        if (rc) return rc;
    }
    const int rc = a.sk_grid > 0 ? of_launch(of_gemm_w4m_kernel<AT, BT, EPI, true>, grid, 256, smem_bytes, s, a)
                                 : of_launch(of_gemm_w4m_kernel<AT, BT, EPI, false>, grid, 256, smem_bytes, s, a);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, ntiles, s);
}
}  // namespace

// Eligibility, separate from the launch so that a caller splitting a problem over two kernels can check both parts before it
// launches either (of_gemm's N-split).  Byte offsets inside the kernel are 32-bit (per-lane offset + scalar stage offset against a
// buffer descriptor based at the tile's first row / column): an operand whose 256-row window (K-contiguous) or whole K extent
// (K-strided) spans >= 4 GiB is refused, the general kernel takes it (gemm_mid.hip has the same guard).
bool of_gemm_w4m_eligible(const OfGemmArgs& a) {
    if ((a.M % TM) || (a.N % TN) || (a.K % DK) || a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
    const unsigned long long a_span = 2ull * (unsigned long long)(a.a_trans ? a.K : TM) * (unsigned long long)a.lda;
    const unsigned long long b_span = 2ull * (unsigned long long)(a.b_trans ? a.K : TN) * (unsigned long long)a.ldb;
    if (a_span >= (1ull << 32) || b_span >= (1ull << 32)) return false;
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

// Workspace a stream-K launch over `grid` workgroups needs (0: grid divides the tile count -- nothing is shared): the *_DOT
// partials of the launch, one flag and one fp32 partial tile per workgroup.
size_t of_gemm_w4m_sk_bytes(const OfGemmArgs& a, int grid) {
    const int ntiles = (a.M / TM) * (a.N / TN);
    if (grid <= 0 || ntiles % grid == 0) return 0;
    return sk_dot_bytes(a) + sk_flags_bytes(grid) + (size_t)grid * SK_SLAB_BYTES;
}

int of_gemm_w4m_try(const OfGemmArgs& a, of_stream_t s) {
    if (!of_gemm_w4m_eligible(a)) return OF_E_SHAPE;
    if (a.sk_grid < 0 || (a.sk_grid & 7)) return OF_E_ARG;         // the XCD-aware tile map wants whole groups of 8 workgroups
    if (a.sk_grid > 0) {
        const size_t need = of_gemm_w4m_sk_bytes(a, a.sk_grid);
        if (need && (!a.workspace || a.workspace_bytes < need || ((uintptr_t)a.workspace & 15))) return OF_E_WORKSPACE;
    }
    const int layout = a.a_trans * 2 + a.b_trans;
    if (layout == 0) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4m<false, false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_w4m<false, false, OF_EPI_GELU>(a, s);
            case OF_EPI_GATE_RESID: return launch_w4m<false, false, OF_EPI_GATE_RESID>(a, s);
            case OF_EPI_ACC_F32: return launch_w4m<false, false, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 1) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4m<false, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch_w4m<false, true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_w4m<false, true, OF_EPI_SCALE_DOT>(a, s);
            case OF_EPI_ACC_F32: return launch_w4m<false, true, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 3) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4m<true, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_ACC_F32: return launch_w4m<true, true, OF_EPI_ACC_F32>(a, s);
        }
    }
    return OF_E_SHAPE;
}
