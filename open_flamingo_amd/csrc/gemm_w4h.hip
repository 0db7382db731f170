// 256 x 128 bf16 MFMA GEMM tile, FOUR waves x (128 x 64) per wave, LDS-DMA operands, TWO WORKGROUPS PER CU (gfx950).
//
// Why (DESIGN.md 4.1, round 4's tile phase probe): the 256 x 256 kernel of gemm_w4m.hip holds ONE wave per SIMD (448 registers, 128-160
// KiB of LDS), so while that wave runs a tile's epilogue -- 12-17 us of VALU / store issue behind a 47-us K loop for the erf-GELU and
// *_DOT forms at K = 2048 -- nothing on the SIMD issues MFMAs.  Here a wave owns 128 x 64 = 8 x 4 accumulators of 16 x 16 (128
// registers), the kernel allocates <= 256 registers and 80 KiB of LDS, and a CU holds two workgroups: one is in its K loop while the
// other one runs its epilogue, launches and drains.  The hardware arbitrates MFMA issue oldest-first, so the two drift half a tile
// apart on their own; OF_W4H_PRIO pins that (a workgroup raises its priority in the second half of its K loop).
// The price: 1.5 x the operand bytes per FLOP through LDS-DMA and LDS (a 256 x 128 tile against 256 x 256).
//
// Operand ring.  K-contiguous operands must travel as whole 128-byte lines (half-line pieces double the L2 requests:
// profiles/README.md, "ring of four half stages"), so a stage stays 64 deep: 48 KiB.  Two such slots do not fit twice into a CU's
// 160 KiB.  The ring is therefore FIVE 16-KiB UNITS and a stage is three of them, in this order: B (128 columns), A0 (rows 0-63
// of both wave rows: tile rows 0-63 and 128-191), A1 (rows 64-127 of both).  The four phases of a stage run ROW-HALF major --
//     p0 (A0, k-step 0)   p1 (A0, k-step 1)   p2 (A1, k-step 0)   p3 (A1, k-step 1)        16 MFMAs each
// -- with the fragments of a phase read during the phase before it, so B(d) and A0(d) are dead after p0 and A1(d) after p2:
//     barrier b1 (p0 | p1): A1(d) has landed;           B(d), A0(d) are free  -> request A1(d+1) (p1), B(d+2) (p2) into them
//     barrier b2 (p2 | p3): B(d+1), A0(d+1) have landed; A1(d) is free        -> request A0(d+2) (p3)
// Unit u lives in slot u % 5; requests are issued in unit order, so the counted waits are vmcnt(8) at both barriers (four pieces per
// wave and unit).  A request has one stage (64 MFMAs per wave) to land.
// Everything else is gemm_w4m.hip's: 16x16x32 MFMAs accumulating in place by inline asm (of_mfma_acc: guarded at the top of a stage,
// settled in front of the epilogue, linted on the ISA by tests/test_isa_lint.py), the LDS images of gemm_tile256.h (a unit is one
// "half" image of 128 rows), the XCD-aware tile order, the epilogue through a private LDS patch per wave (gemm_common.h).
// Per-element summation order is the 256 x 256 kernel's in stage order (k-steps ascending): the two kernels agree bit for bit when that
// kernel is forced (tools build); launches of_gemm sends to the 256 x 256 kernel itself rotate their K loop per XCD (gemm_w4m.hip) and
// agree to summation order only.
#include <type_traits>
#include "gemm_tile256.h"

namespace {
using namespace oft;

constexpr int HT_M = 256, HT_N = 128;
constexpr int UNIT_BYTES = HALF_BYTES;          // 16 KiB: 128 rows (or columns) x 64 of K
constexpr int NUNIT = 5;
constexpr int SMEM_W4H = NUNIT * UNIT_BYTES;    // 80 KiB: two workgroups fill the CU's 160 KiB

#if defined(OF_TOOLS_BUILD) && !defined(OF_HOST_EMU)
// tools/libofhip_tools.so only (tools/probes/w4h_probe.py): scheduling knobs of the study build.  bit 0: priority 1 in the second half
// of the K loop; bit 1: priority 1 for odd rounds of workgroups ((bid / 512) & 1) instead; bit 2: priority 2 during the epilogue;
// bit 3 (host side): launch with 96 KiB of LDS, i.e. ONE workgroup per CU (what a K loop does with the CU to itself).
__device__ int of_w4h_knob = 0;
static int g_w4h_knob_host = 0;
extern "C" int of_tools_set_w4h_knob(int v) {
    g_w4h_knob_host = v;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(of_w4h_knob), &v, sizeof(v));
}
#define OF_W4H_LDS_EXTRA() ((g_w4h_knob_host & 8) ? 16384 : 0)
#define OF_W4H_KNOB() of_uniform(of_w4h_knob)
// ... and the phase stamps of gemm_w4m.hip's study build (wave 0 of every workgroup: entry, prologue done, K loop done, last epilogue
// instruction, stores acknowledged; + the XCC / SE / CU it ran on), 8 x u64 per workgroup
__device__ unsigned long long* of_w4h_stamps = nullptr;
extern "C" int of_tools_set_w4h_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(of_w4h_stamps), &p, sizeof(p)); }
#define OF_STAMP(i) (of_stamp_t[i] = wall_clock64())
#define OF_STAMP_DECL() unsigned long long of_stamp_t[5] = {0, 0, 0, 0, 0}, of_cyc_t[2] = {0, 0}
#define OF_STAMP_FLUSH()                                                                                              \
    do {                                                                                                              \
        if (of_w4h_stamps) {                                                                                          \
            of_wait_vm<0>();                                                                                          \
            OF_STAMP(4);                                                                                              \
            if (of_tid() == 0) {                                                                                      \
                unsigned hw;                                                                                          \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                     \
                unsigned long long* o = of_w4h_stamps + (size_t)of_bid_x() * 8;                                       \
                for (int i_ = 0; i_ < 5; ++i_) o[i_] = of_stamp_t[i_];                                                \
                o[5] = of_cyc_t[0];                                                                                   \
                o[6] = of_cyc_t[1];                                                                                   \
                o[7] = ((unsigned long long)(of_bid_x() & 7) << 32) | hw;                                             \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#define OF_W4H_EPI_PRIO()                                          \
    do {                                                           \
        if (OF_W4H_KNOB() & 4) __builtin_amdgcn_s_setprio(2);      \
        else of_setprio_lo();                                      \
    } while (0)
#define OF_CYC(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(of_cyc_t[i])::"memory")
#else
#define OF_W4H_LDS_EXTRA() 0
#define OF_W4H_EPI_PRIO() of_setprio_lo()
#define OF_CYC(i) ((void)0)
#define OF_W4H_KNOB() 0
#define OF_STAMP(i) ((void)0)
#define OF_STAMP_DECL() ((void)0)
#define OF_STAMP_FLUSH() ((void)0)
#endif

OF_DEV int m5(int x) { return x >= NUNIT ? x - NUNIT : x; }

// Epilogue of a wave's 128 x 64: four 32 x 64 accumulator groups through the wave's private LDS patch.  The operand ring is idle
// by now: patches at its start, the aux buffers of the *_DOT forms (two 4-KiB buffers per wave, groups g and g + 1 in flight) behind
// them.  Latencies left exposed here (the first aux tile is requested now, not during the K loop) are covered by the CU's other
// workgroup.
template <int EPI, bool ASMD, class ToPatch>
OF_DEV void w4h_epilogue(const OfGemmArgs& p, ToPatch to_patch, char* smem, int m0w, int n0w, int wave, int lane, int dot_slot) {
    constexpr bool AUXL = EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT;
    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    const float sc = gv * p.alpha;
    float dot = 0.f;
    char* patch = smem + wave * ofg::PATCH_BYTES;
    if constexpr (AUXL) {
        char* buf0 = smem + 4 * ofg::PATCH_BYTES + 256 + wave * ofg::AUX_LDS_BYTES;
        char* buf1 = buf0 + 4 * ofg::AUX_LDS_BYTES;
        of_wait_vm<0>();
        ofg::epilogue_group_aux_dma<ASMD>(p, m0w, n0w, lane, buf0);
        ofg::epilogue_group_aux_dma<ASMD>(p, m0w + 32, n0w, lane, buf1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // vmcnt by hand: a group = 4 pieces, its row passes = 4 stores; group g + 1 goes into the buffer group g - 1 was read from
            if (g == 1 || g == 2) ofg::epilogue_group_aux_dma<ASMD>(p, m0w + (g + 1) * 32, n0w, lane, (g & 1) ? buf0 : buf1);
            if (g == 0 || g == 3) of_wait_vm<4>();
            else of_wait_vm<8>();
            to_patch(g, patch);
            ofg::epilogue_group_rows_auxlds<EPI>(p, patch, (g & 1) ? buf1 : buf0, m0w + g * 32, n0w, lane, gv, sc, dot);
        }
    } else {
        ofg::AuxPre pre[2][4];
        ofg::epilogue_group_aux<EPI>(p, m0w, n0w, lane, pre[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) ofg::epilogue_group_aux<EPI>(p, m0w + (g + 1) * 32, n0w, lane, pre[(g + 1) & 1]);
            to_patch(g, patch);
            ofg::epilogue_group_rows<EPI>(p, patch, m0w + g * 32, n0w, lane, gv, sc, dot, pre[g & 1]);
        }
    }
    ofg::epilogue_finish<EPI>(p, dot, lane, wave, 4, (float*)(smem + 4 * ofg::PATCH_BYTES), dot_slot);
}

// VAR: 0 = the product schedule.  Study variants (tools build only, host knob bits 4-6): 1 = the four LDS-DMA pieces of a phase in its
// first MFMA gaps instead of spread over it; 2 = no counted vmcnt waits (WRONG RESULTS: timing only); 3 = no barriers (likewise).
template <bool BT, int EPI, int VAR = 0>
OF_GLOBAL void OF_BOUNDS(256, 2) of_gemm_w4h_kernel(OfGemmArgs p) {
    constexpr bool ASMD = BT;             // LDS-DMA form (of_platform.h): inline asm wherever a transposed-fragment read follows
    OF_STAMP_DECL();
    OF_STAMP(0);
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63;
    const int wave = of_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = p.M / HT_M, tiles_n = p.N / HT_N, ntiles = tiles_m * tiles_n;
    const int nd = p.K / DK;
    int pm, pn;
    ofg::tile_coords(of_bid_x(), ntiles, tiles_m, tiles_n, pm, pn);
    const int m0 = pm * HT_M, n0 = pn * HT_N;

    f32x4 acc[8][4];      // [16-row block of M][16-column block of N]
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;

    // DMA duty of this wave: 1-KiB chunks q = jj*4 + wave (jj = 0..3) of every unit.  Chunk q of an A unit = 8 rows of wave row q >> 3
    // (local rows 64 (q >> 3) + 8 (q & 7) ..), the row half `ah` of the unit is a scalar offset of 64 rows.
    unsigned offA[4], offB[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int q = jj * 4 + wave;
        offA[jj] = 2u * chunk_off<false>(p.lda, q >> 3, q & 7, lane);
        offB[jj] = 2u * mchunk_off<BT>(p.ldb, 0, q, lane);
    }
    const unsigned stepA = 2u * (unsigned)DK;
    const unsigned stepB = 2u * (BT ? (unsigned)DK * (unsigned)p.ldb : (unsigned)DK);
    const unsigned halfA = 2u * 64u * (unsigned)p.lda;
    const unsigned smem_u = of_lds_base(smem) + (unsigned)wave * 1024u;
    const of_buf_t gA = of_buf_make(chunk_base<false>(p.A, p.lda, m0));
    const of_buf_t gB = of_buf_make(chunk_base<BT>(p.B, p.ldb, n0));
    // piece jj of unit kind (0: B, 1: A0, 2: A1) of the stage at byte offsets (sa, sb) into ring slot `slot`
    auto dma_piece = [&](int kind, int jj, int slot, unsigned sa, unsigned sb) OF_INLINE_LAMBDA {
        const unsigned dst = smem_u + (unsigned)slot * (unsigned)UNIT_BYTES + (unsigned)jj * 4096u;
        if (kind == 0) of_buf_load16_lds_at<ASMD>(gB, offB[jj], sb, dst);
        else of_buf_load16_lds_at<ASMD>(gA, offA[jj], sa + (kind == 2 ? halfA : 0u), dst);
    };

    s16x8 fa[2][4], fb[2][4];     // fa[phase & 1]: the 4 A fragments of a phase; fb[ks]: the 4 B fragments of k-step ks
    auto read_a = [&](const char* unit, int ks, int buf, int r) OF_INLINE_LAMBDA { fa[buf][r] = mfrag16<false>(unit, wm * 64 + r * 16, ks, lane); };
    auto read_b = [&](const char* unit, int ks, int r) OF_INLINE_LAMBDA { fb[ks][r] = mfrag16<BT>(unit, wn * 64 + r * 16, ks, lane); };
    // the 8 fragments a phase that starts a k-step needs, in the order of first use: b0 a0 b1 b2 b3 a1 a2 a3
    auto read8 = [&](const char* ua, const char* ub, int ks, int abuf, int r) OF_INLINE_LAMBDA {
        if (r == 0) read_b(ub, ks, 0);
        else if (r == 1) read_a(ua, ks, abuf, 0);
        else if (r < 5) read_b(ub, ks, r - 1);
        else read_a(ua, ks, abuf, r - 4);
    };

    // ---- prologue: units 0-2 (stage 0) and 3-4 (B, A0 of stage 1) requested; B(0), A0(0) landed; their k-step 0 fragments read
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dma_piece(k, jj, k, 0u, 0u);
    if (nd > 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) dma_piece(k, jj, 3 + k, stepA, stepB);
        of_wait_vm<12>();
    } else {
        of_wait_vm<4>();
    }
    of_barrier_raw();
    OF_STAMP(1);
    OF_CYC(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) read8(smem + UNIT_BYTES, smem, 0, 0, r);

    // The K loop, compiled once per wave parity (odd waves request their pieces two MFMA gaps after the even ones).
    auto main_loop = [&](auto parc) OF_INLINE_LAMBDA {
        constexpr int PARC = decltype(parc)::value;
        // One phase = 16 MFMAs: B fragments fb[ks] x A fragments fa[ph & 1] -> accumulator rows 4 (ph >> 1) ...  `rd(r)` issues the
        // r-th fragment read of the NEXT phase (nrd of them, in the first MFMA gaps: the last one has >= 8 MFMAs to return behind),
        // `dma(jj)` one of four LDS-DMA pieces.
        auto phase = [&](int ph, int nrd, auto rd, bool has_dma, auto dma) OF_INLINE_LAMBDA {
            const int ks = ph & 1, ah = ph >> 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                of_mfma_acc(fb[ks][i & 3], fa[ph & 1][i >> 2], acc[ah * 4 + (i >> 2)][i & 3]);
                if (i < nrd) rd(i);
                if (VAR == 1) {
                    if (has_dma && i >= 2 * PARC && i < 2 * PARC + 4) dma(i - 2 * PARC);
                } else if (has_dma && (i & 3) == 2 * PARC) dma(i >> 2);
                of_sched_fence();
            }
        };
        // Stage d; `base` = slot of its unit B.  E1 / E2: stage d + 1 / d + 2 exists.  (sa1, sb1) = byte offsets of stage d + 1.
        auto stage_body = [&](int base, unsigned sa1, unsigned sb1, const bool E1, const bool E2) OF_INLINE_LAMBDA {
            const int sB = base, sA0 = m5(base + 1), sA1 = m5(base + 2), sBn = m5(base + 3), sA0n = m5(base + 4);
            const char* uB = smem + sB * UNIT_BYTES;
            const char* uA0 = smem + sA0 * UNIT_BYTES;
            const char* uA1 = smem + sA1 * UNIT_BYTES;
            const char* uBn = smem + sBn * UNIT_BYTES;
            const char* uA0n = smem + sA0n * UNIT_BYTES;
            of_mfma_acc_guard();       // fragments may have been moved between registers on the way into this stage (of_platform.h)
            phase(0, 8, [&](int r) OF_INLINE_LAMBDA { read8(uA0, uB, 1, 1, r); }, false, [&](int) OF_INLINE_LAMBDA {});
            if (VAR == 2) {
            } else if (E1) of_wait_vm<8>();   // own pieces of A1(d) have landed (behind them: B, A0 of stage d + 1)
            else of_wait_vm<0>();
            of_wait_lgkm0();           // own reads of B(d), A0(d) are done ...
            if (VAR != 3) of_barrier_raw();          // ... and so are everybody else's
            of_sched_fence();
            phase(1, 4, [&](int r) OF_INLINE_LAMBDA { read_a(uA1, 0, 0, r); }, E1, [&](int jj) OF_INLINE_LAMBDA { dma_piece(2, jj, sB, sa1, sb1); });
            phase(2, 4, [&](int r) OF_INLINE_LAMBDA { read_a(uA1, 1, 1, r); }, E2, [&](int jj) OF_INLINE_LAMBDA { dma_piece(0, jj, sA0, sa1 + stepA, sb1 + stepB); });
            if (VAR == 2) {
            } else if (E1 && E2) of_wait_vm<8>();       // own pieces of B, A0 of stage d + 1 have landed (behind them: A1(d+1), B(d+2))
            else if (E1) of_wait_vm<4>();
            else of_wait_vm<0>();
            of_wait_lgkm0();           // own reads of A1(d) are done
            if (VAR != 3) of_barrier_raw();
            of_sched_fence();
            phase(3, E1 ? 8 : 0, [&](int r) OF_INLINE_LAMBDA { read8(uA0n, uBn, 0, 0, r); }, E2,
                  [&](int jj) OF_INLINE_LAMBDA { dma_piece(1, jj, sA1, sa1 + stepA, sb1 + stepB); });
        };
        int d = 0, base = 0;
        unsigned sa1 = stepA, sb1 = stepB;
        const int prio = OF_W4H_KNOB();
        for (; d + 2 < nd; ++d) {
            if ((prio & 1) && d == (nd >> 1)) of_setprio_hi();
            stage_body(base, sa1, sb1, true, true);
            base = m5(base + 3);
            sa1 += stepA;
            sb1 += stepB;
        }
        if (d + 1 < nd) {
            stage_body(base, sa1, sb1, true, false);
            base = m5(base + 3);
            sa1 += stepA;
            sb1 += stepB;
            ++d;
        }
        stage_body(base, sa1, sb1, false, false);
    };
    if (OF_W4H_KNOB() & 2) {
        if ((of_bid_x() >> 9) & 1) of_setprio_hi();
    }
    if (OF_W4H_KNOB() & 128) of_setprio_hi();       // the whole K loop above the (priority-0) epilogue of the CU's other workgroup
    if (wave & 1) main_loop(std::integral_constant<int, 1>{});
    else main_loop(std::integral_constant<int, 0>{});
    of_mfma_acc_settle();
    OF_W4H_EPI_PRIO();
    of_barrier_raw();          // the ring is idle from here
    OF_STAMP(2);
    OF_CYC(1);

    auto acc_to_patch = [&](int g, char* patch) OF_INLINE_LAMBDA {
        const f32x4 t[2][4] = {{acc[2 * g][0], acc[2 * g][1], acc[2 * g][2], acc[2 * g][3]},
                               {acc[2 * g + 1][0], acc[2 * g + 1][1], acc[2 * g + 1][2], acc[2 * g + 1][3]}};
        ofg::patch_write16(patch, t, lane);
    };
    w4h_epilogue<EPI, ASMD>(p, acc_to_patch, smem, m0 + wm * 128, n0 + wn * 64, wave, lane, pm * tiles_n + pn);
    OF_STAMP(3);
    OF_STAMP_FLUSH();
}

template <bool BT, int EPI>
int launch_w4h(const OfGemmArgs& a, of_stream_t s) {
    const int ntiles = (a.M / HT_M) * (a.N / HT_N);
    int rc;
#if defined(OF_TOOLS_BUILD) && !defined(OF_HOST_EMU)
    const int var = (g_w4h_knob_host >> 4) & 7;
    const of_dim3 grid{(unsigned)ntiles, 1, 1};
    if (var == 1) rc = of_launch(of_gemm_w4h_kernel<BT, EPI, 1>, grid, 256, SMEM_W4H + OF_W4H_LDS_EXTRA(), s, a);
    else if (var == 2) rc = of_launch(of_gemm_w4h_kernel<BT, EPI, 2>, grid, 256, SMEM_W4H + OF_W4H_LDS_EXTRA(), s, a);
    else if (var == 3) rc = of_launch(of_gemm_w4h_kernel<BT, EPI, 3>, grid, 256, SMEM_W4H + OF_W4H_LDS_EXTRA(), s, a);
    else
#endif
        rc = of_launch(of_gemm_w4h_kernel<BT, EPI, 0>, of_dim3{(unsigned)ntiles, 1, 1}, 256, SMEM_W4H + OF_W4H_LDS_EXTRA(), s, a);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, ntiles, s);
}
}  // namespace

// Eligibility, separate from the launch (of_gemm's selection checks it first).  A is K-contiguous (y = x W^T and dX = dY W: the
// launches with a fat epilogue); 32-bit byte offsets as in gemm_w4m.hip.
bool of_gemm_w4h_eligible(const OfGemmArgs& a) {
    if ((a.M % HT_M) || (a.N % HT_N) || (a.K % DK) || a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
    if (a.a_trans || a.group_kind) return false;
    const unsigned long long a_span = 2ull * (unsigned long long)HT_M * (unsigned long long)a.lda;
    const unsigned long long b_span = 2ull * (unsigned long long)(a.b_trans ? a.K : HT_N) * (unsigned long long)a.ldb;
    if (a_span >= (1ull << 32) || b_span >= (1ull << 32)) return false;
    switch (a.epi) {
        case OF_EPI_STORE_BF16:
        case OF_EPI_ACC_F32: return true;
        case OF_EPI_GELU:
        case OF_EPI_GATE_RESID: return !a.b_trans;
        case OF_EPI_DGELU_DOT:
        case OF_EPI_SCALE_DOT: return a.b_trans != 0;
    }
    return false;
}

int of_gemm_w4h_try(const OfGemmArgs& a, of_stream_t s) {
    if (!of_gemm_w4h_eligible(a)) return OF_E_SHAPE;
    if (!a.b_trans) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4h<false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_w4h<false, OF_EPI_GELU>(a, s);
            case OF_EPI_GATE_RESID: return launch_w4h<false, OF_EPI_GATE_RESID>(a, s);
            case OF_EPI_ACC_F32: return launch_w4h<false, OF_EPI_ACC_F32>(a, s);
        }
    } else {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4h<true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch_w4h<true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_w4h<true, OF_EPI_SCALE_DOT>(a, s);
            case OF_EPI_ACC_F32: return launch_w4h<true, OF_EPI_ACC_F32>(a, s);
        }
    }
    return OF_E_SHAPE;
}
