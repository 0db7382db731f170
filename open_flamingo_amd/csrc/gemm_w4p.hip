// Persistent, software-pipelined 256 x 128 bf16 MFMA GEMM tile (gfx950): ONE workgroup per CU, four waves x (128 x 64), and a tile's
// epilogue runs INSIDE the K loop of the workgroup's next tile.
//
// Why this form (round 5's probes).  The K loops of the big-tile kernels run at the chip's power cap (~1.54 "MFMA-GHz",
// 256 x 256 and 256 x 128 tiles alike); what a K = 2048 launch loses is the time in which no MFMA issues: prologue, epilogue, drain,
// dispatch -- 14 % of a plain-store launch, 25-33 % of the erf-GELU / *_DOT launches.  gemm_w4h.hip put two workgroups on a CU so
// that one's epilogue runs under the other's K loop: it gains 2-5 % on the *_DOT launches and nothing elsewhere, because a K loop
// that has its SIMD to itself issues MFMAs only 68 % of the time (its own LDS-DMA issue stalls it), and because an epilogue wave
// beside a K-loop wave gets a third of the issue slots.  Here the SAME wave does both, on a fixed schedule:
//   * at the end of a tile's K loop the wave rounds its 128 accumulator registers to bf16 into a private 16-KiB LDS image of its
//     128 x 64 sub-tile (< 1 us; the first stage of the next tile is already in flight) -- bf16 is what every epilogue taken here
//     stores, and rounding the product first is the reference's own order (autocast: the Linear's bf16 output feeds GELU);
//   * during the first 16 K stages of the NEXT tile it drains that image, one 8-row chunk per stage: one ds_read_b128, the epilogue
//     math cut into steps of a few VALU instructions that sit in fixed MFMA gaps (the loop is bound by power, not by issue: a stage
//     of 64 MFMAs takes ~2400 cycles at the capped clock and has room for them), the 16-byte stores right behind the stage's barrier;
//   * the ring is gemm_w4m.hip's: two 48-KiB stages (A 256 x 64 + B 128 x 64), one barrier per stage, B of stage d + 1 requested in
//     phase 0, A of stage d + 2 in phase 3.  LDS: 96 KiB ring + 64 KiB images = the CU's 160 KiB.
// Tiles: workgroup b takes tiles b, b + G, b + 2G, ... (G = 256 or the tile count) in the XCD-aware order; the last tile's image is
// drained without MFMAs.  Epilogues: the bf16-output ones; K >= 1152 (18 stages: 16 carry a chunk, the two tail stages none).
// Results: STORE_BF16 is bit-equal to the 256 x 256 kernel; GELU rounds the product to bf16 first (= its own pre-activation output).
#include <type_traits>
#include "gemm_tile256.h"

namespace {
using namespace oft;

constexpr int PT_M = 256, PT_N = 128;
constexpr int P_OPER_A = OPER_BYTES;                 // 32 KiB: A image of a stage (two half images of 128 rows)
constexpr int P_STAGE = OPER_BYTES + HALF_BYTES;     // 48 KiB: + B image (128 columns)
constexpr int P_RING = 2 * P_STAGE;                  // 96 KiB
constexpr int P_IMG = 16384;                         // per wave: 128 rows x 64 bf16, 16-byte slot s of row r at slot s ^ (r & 7)
constexpr int SMEM_W4P = P_RING + 4 * P_IMG;         // 160 KiB
constexpr int P_CHUNKS = 16;                         // 8-row chunks of a wave's image: one per K stage
constexpr int P_MIN_STAGES = P_CHUNKS + 2;

OF_DEV f32x2 p_unpack(unsigned w) { return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; }

// ---- the epilogue of one 8-row chunk of a wave's image, cut into steps the K loop places in MFMA gaps.  A lane owns 8 consecutive
// columns of one row: 16 bytes of the image, one 16-byte store per output.
struct ChunkState {
    u32x4 raw;              // 8 bf16 of the rounded product
    f32x2 a[4], t[4], u[4];
    u32x4 out;
};
// erf-GELU of of_platform.h (of_gelu2: Abramowitz-Stegun 7.1.28) over the chunk's four column pairs at once: step k is a handful of
// independent packed instructions
constexpr int GELU_STEPS = 16;
OF_DEV void p_gelu_step(ChunkState& s, int k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2& x = s.t[q];
        f32x2& p = s.u[q];
        switch (k) {
            case 0: s.a[q] = p_unpack(s.raw[q]); break;
            case 1: x = __builtin_elementwise_abs(s.a[q]) * 0.70710678118654752f; break;
            case 2: p = of_fma2(x, of_splat2(0.0000430638f), of_splat2(0.0002765672f)); break;
            case 3: p = of_fma2(x, p, of_splat2(0.0001520143f)); break;
            case 4: p = of_fma2(x, p, of_splat2(0.0092705272f)); break;
            case 5: p = of_fma2(x, p, of_splat2(0.0422820123f)); break;
            case 6: p = of_fma2(x, p, of_splat2(0.0705230784f)); break;
            case 7: p = of_fma2(x, p, of_splat2(1.0f)); break;
            case 8: p = p * p; break;
            case 9: p = p * p; break;
            case 10: p = p * p; break;
            case 11: p = p * p; break;
            case 12: p = f32x2{of_rcp(p[0]), of_rcp(p[1])}; break;
            case 13: p = of_fma2(p, of_splat2(-0.5f), of_splat2(0.5f)); break;      // 0.5 erf(|a| / sqrt 2)
            case 14: p = s.a[q] * (of_splat2(0.5f) + __builtin_elementwise_copysign(p, s.a[q])); break;
            case 15: s.out[q] = of_pack_bf16(p[0], p[1]); break;
        }
    }
}
template <int EPI>
constexpr int epi_steps() { return EPI == OF_EPI_GELU ? GELU_STEPS : 0; }

template <bool BT, int EPI>
OF_GLOBAL void OF_BOUNDS(256, 1) of_gemm_w4p_kernel(OfGemmArgs p) {
    constexpr bool ASMD = BT;             // LDS-DMA form (of_platform.h): inline asm wherever a transposed-fragment read follows
    constexpr int NSTEP = epi_steps<EPI>();
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63;
    const int wave = of_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = p.M / PT_M, tiles_n = p.N / PT_N, ntiles = tiles_m * tiles_n;
    const int nd = p.K / DK;
    const int G = of_gdim_x(), bid = of_bid_x();
    const int my_tiles = (ntiles - bid + G - 1) / G;

    f32x4 acc[8][4];      // [16-row block of M][16-column block of N]
    unsigned offA[2][4], offB[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) offA[hf][jj] = 2u * chunk_off<false>(p.lda, hf, jj * 4 + wave, lane);
        offB[jj] = 2u * mchunk_off<BT>(p.ldb, 0, jj * 4 + wave, lane);
    }
    const unsigned stepA = 2u * (unsigned)DK;
    const unsigned stepB = 2u * (BT ? (unsigned)DK * (unsigned)p.ldb : (unsigned)DK);
    const unsigned smem_u = of_lds_base(smem) + (unsigned)wave * 1024u;
    // this wave's image of the tile it has just finished
    char* img = smem + P_RING + wave * P_IMG;
    const int img_rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);      // + chunk * 1024
    const int img_wr = (lane & 15) * 128 + ((lane >> 4) & 1) * 8;                  // + a * 2048 + (((2 b + (lane >> 5)) ^ (lane & 7)) << 4)
    const unsigned out_off = 2u * (unsigned)((lane >> 3) * p.ldc + (lane & 7) * 8);   // + chunk * 8 rows (scalar), bytes
    const unsigned out_chunk = 2u * 8u * (unsigned)p.ldc;
    float sc = p.alpha;
    if (p.gate) sc *= of_tanh(*p.gate);

    of_buf_t outC = of_buf_make(p.C), outC2 = of_buf_make(p.C2 ? p.C2 : p.C);      // re-based per tile
    ChunkState cs;
    cs.raw = u32x4{0, 0, 0, 0};
    cs.out = u32x4{0, 0, 0, 0};
    // the steps of chunk `c` (the stage index of the K loop that carries it)
    auto chunk_read = [&](int c) OF_INLINE_LAMBDA { cs.raw = *(const u32x4*)(img + c * 1024 + img_rd); };
    auto chunk_store = [&](int c, int which) OF_INLINE_LAMBDA {
        const unsigned so = (unsigned)c * out_chunk;
        if (EPI == OF_EPI_STORE_BF16) {
            if (which == 0) of_buf_store16(outC, out_off, so, cs.raw);
        } else if (EPI == OF_EPI_GELU) {
            if (which == 0) {
                if (p.C2) of_buf_store16(outC2, out_off, so, cs.raw);
            } else {
                of_buf_store16(outC, out_off, so, cs.out);
            }
        }
    };
    auto chunk_step = [&](int k) OF_INLINE_LAMBDA {
        if (EPI == OF_EPI_GELU) p_gelu_step(cs, k);
    };
    // the accumulators of the tile whose K loop has just ended -> this wave's image (rounded to bf16; STORE_BF16 scales first)
    auto dump = [&]() OF_INLINE_LAMBDA {
        of_mfma_acc_settle();
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 v = acc[a][b];
                const float m = EPI == OF_EPI_STORE_BF16 ? sc : 1.0f;
                const u32x2 w = {of_pack_bf16(m * v[0], m * v[1]), of_pack_bf16(m * v[2], m * v[3])};
                *(u32x2*)(img + a * 2048 + img_wr + (((2 * b + (lane >> 5)) ^ (lane & 7)) << 4)) = w;
            }
        of_wave_sync();
    };

    for (int seg = 0; seg <= my_tiles; ++seg) {
        const bool has_tile = seg < my_tiles;
        int pm = 0, pn = 0;
        if (has_tile) ofg::tile_coords(bid + seg * G, ntiles, tiles_m, tiles_n, pm, pn);
        const int m0 = pm * PT_M, n0 = pn * PT_N;
        const of_buf_t gA = of_buf_make(chunk_base<false>(p.A, p.lda, m0));
        const of_buf_t gB = of_buf_make(chunk_base<BT>(p.B, p.ldb, n0));
        unsigned sA = 0, sB = 0;              // scalar byte offsets of the next stage to request
        // piece j (0..11: 0-7 = A (hf = j >> 2, jj = j & 3), 8-11 = B) of the stage at (sA, sB) -- ahead = 1: of the stage after it --
        // into the slot at byte offset slot_off
        auto dma_piece = [&](unsigned slot_off, int j, int ahead) OF_INLINE_LAMBDA {
            if (j < 8) {
                const int hf = j >> 2, jj = j & 3;
                of_buf_load16_lds_at<ASMD>(gA, offA[hf][jj], sA + (ahead ? stepA : 0u), smem_u + slot_off + (unsigned)(hf * HALF_BYTES + jj * 4096));
            } else {
                const int jj = j - 8;
                of_buf_load16_lds_at<ASMD>(gB, offB[jj], sB + (ahead ? stepB : 0u), smem_u + slot_off + (unsigned)(P_OPER_A + jj * 4096));
            }
        };
        if (has_tile) {       // ---- the ring is idle (barrier below): stage 0 -> slot 0, A of stage 1 -> slot 1
#pragma unroll
            for (int j = 0; j < 12; ++j) dma_piece(0, j, 0);
            sA += stepA;
            sB += stepB;
#pragma unroll
            for (int j = 0; j < 8; ++j) dma_piece(P_STAGE, j, 0);
        }
        if (seg > 0) dump();
        if (!has_tile) {      // ---- the last tile's image: drained without a K loop to hide under
            for (int c = 0; c < P_CHUNKS; ++c) {
                chunk_read(c);
#pragma unroll
                for (int k = 0; k < NSTEP; ++k) chunk_step(k);
                chunk_store(c, 0);
                chunk_store(c, 1);
            }
            break;
        }
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;

        s16x8 fa[2][4], fb[2][4];     // fa[phase & 1]: the 4 A fragments of a phase; fb[ks]: the 4 B fragments of k-step ks
        auto read_a = [&](const char* stage, int ks, int ah, int buf, int r) OF_INLINE_LAMBDA {
            fa[buf][r] = mfrag16<false>(stage, wm * 128 + ah * 64 + r * 16, ks, lane);
        };
        auto read_b = [&](const char* stage, int ks, int r) OF_INLINE_LAMBDA { fb[ks][r] = mfrag16<BT>(stage + P_OPER_A, wn * 64 + r * 16, ks, lane); };
        // the 8 fragments a phase that starts a k-step needs, in the order of first use: b0 a0 b1 b2 b3 a1 a2 a3
        auto read8 = [&](const char* stage, int ks, int abuf, int r) OF_INLINE_LAMBDA {
            if (r == 0) read_b(stage, ks, 0);
            else if (r == 1) read_a(stage, ks, 0, abuf, 0);
            else if (r < 5) read_b(stage, ks, r - 1);
            else read_a(stage, ks, 0, abuf, r - 4);
        };
        of_wait_vm<8>();           // stage 0 has landed (behind it: A of stage 1)
        of_barrier_raw();
#pragma unroll
        for (int r = 0; r < 8; ++r) read8(smem, 0, 0, r);

        const bool drain = seg > 0;       // the previous tile's image rides in the first P_CHUNKS stages
        auto main_loop = [&](auto parc) OF_INLINE_LAMBDA {
            constexpr int PARC = decltype(parc)::value;
            // One phase = 16 MFMAs: B fragments fb[ks] x A fragments fa[ph & 1] -> accumulator rows 4 (ph & 1) ...; `rd(r)` = the r-th
            // fragment read of the NEXT phase (first gaps), `dma(j)` = LDS-DMA pieces, `epi(i)` = what the chunk's epilogue does in gap i
            auto phase = [&](int ph, int nrd, auto rd, int ndma, auto dma, auto epi) OF_INLINE_LAMBDA {
                const int ks = ph >> 1, ah = ph & 1;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    of_mfma_acc(fb[ks][i & 3], fa[ph & 1][i >> 2], acc[ah * 4 + (i >> 2)][i & 3]);
                    if (i < nrd) rd(i);
                    if (ndma == 4 && (i & 3) == 2 * PARC) dma(i >> 2);
                    if (ndma == 8 && (i & 1) == PARC) dma(i >> 1);
                    epi(i);
                    of_sched_fence();
                }
            };
            // One K stage in slot `cur`.  WR: stage d + 1 exists, LD: stage d + 2 exists.  (sA, sB) = offsets of stage d + 1.
            // EON: the stage carries chunk `c` of the previous tile's image: read in phase 0, math in phases 0-2, stores behind the barrier.
            auto stage_body = [&](char* cur, char* nxt, const bool WR, const bool LD, const bool EON, int c) OF_INLINE_LAMBDA {
                const unsigned cur_u = (unsigned)(cur - smem), nxt_u = (unsigned)(nxt - smem);
                of_mfma_acc_guard();       // fragments may have been moved between registers on the way into this stage (of_platform.h)
                // epilogue steps: gap g = 16 ph + i; the image read in gap 1, step k in gap 6 + 2k (k < 16: through gap 36), the stores
                // in gaps 1 / 3 of phase 3
                auto epi_gap = [&](int g) OF_INLINE_LAMBDA {
                    if (!EON) return;
                    if (g == 1) chunk_read(c);
                    if (NSTEP > 0 && g >= 6 && g < 48 && ((g - 6) & 1) == 0 && (g - 6) / 2 < NSTEP) chunk_step((g - 6) / 2);
                    if (g == 48 + 1) chunk_store(c, 0);
                    if (g == 48 + 3) chunk_store(c, 1);
                };
                phase(0, 4, [&](int r) OF_INLINE_LAMBDA { read_a(cur, 0, 1, 1, r); }, WR ? 4 : 0,
                      [&](int j) OF_INLINE_LAMBDA { dma_piece(nxt_u, 8 + j, 0); }, [&](int i) OF_INLINE_LAMBDA { epi_gap(i); });       // + B of stage d + 1 -> nxt
                phase(1, 8, [&](int r) OF_INLINE_LAMBDA { read8(cur, 1, 0, r); }, 0, [&](int) OF_INLINE_LAMBDA {}, [&](int i) OF_INLINE_LAMBDA { epi_gap(16 + i); });
                phase(2, 4, [&](int r) OF_INLINE_LAMBDA { read_a(cur, 1, 1, 1, r); }, 0, [&](int) OF_INLINE_LAMBDA {}, [&](int i) OF_INLINE_LAMBDA { epi_gap(32 + i); });
                of_wait_vm<0>();       // own pieces of stage d + 1 have landed (and last stage's chunk stores are acknowledged) ...
                of_wait_lgkm0();       // ... own reads of this slot are done ...
                of_barrier_raw();      // ... and so are everybody else's
                of_sched_fence();
                phase(3, WR ? 8 : 0, [&](int r) OF_INLINE_LAMBDA { read8(nxt, 0, 0, r); }, LD ? 8 : 0,
                      [&](int j) OF_INLINE_LAMBDA { dma_piece(cur_u, j, 1); },                       // + A of stage d + 2 -> cur
                      [&](int i) OF_INLINE_LAMBDA { epi_gap(48 + i); });
                sA += stepA;
                sB += stepB;
            };
            int d = 0;
            if (drain)
                for (; d < P_CHUNKS; ++d) stage_body(smem + (d & 1) * P_STAGE, smem + ((d + 1) & 1) * P_STAGE, true, true, true, d);
            for (; d + 2 < nd; ++d) stage_body(smem + (d & 1) * P_STAGE, smem + ((d + 1) & 1) * P_STAGE, true, true, false, 0);
            stage_body(smem + (d & 1) * P_STAGE, smem + ((d + 1) & 1) * P_STAGE, true, false, false, 0);
            ++d;
            stage_body(smem + (d & 1) * P_STAGE, smem + ((d + 1) & 1) * P_STAGE, false, false, false, 0);
        };
        // the previous tile's outputs: re-base the store descriptors (its chunks are stored during this K loop)
        if (wave & 1) main_loop(std::integral_constant<int, 1>{});
        else main_loop(std::integral_constant<int, 0>{});
        of_barrier_raw();          // the ring is idle from here
        // this tile's outputs (stored during the next K loop, or by the drain): wave-uniform base of the wave's 128 x 64
        const size_t o = (size_t)(m0 + wm * 128) * p.ldc + n0 + wn * 64;
        outC = of_buf_make((const bf16_t*)p.C + o);
        if (p.C2) outC2 = of_buf_make((const bf16_t*)p.C2 + o);
    }
}

template <bool BT, int EPI>
int launch_w4p(const OfGemmArgs& a, of_stream_t s) {
    const int ntiles = (a.M / PT_M) * (a.N / PT_N);
    // one workgroup per CU the caller lets the launch count on (OfGemmArgs.cu_limit; 0 = all), whole groups of 8 for the XCD-aware tile
    // order where there are that many
    int grid = a.cu_limit > 0 && a.cu_limit < OF_NUM_CUS ? a.cu_limit : OF_NUM_CUS;
    if (grid > ntiles) grid = ntiles;
    if (grid >= 8) grid &= ~7;
    return of_launch(of_gemm_w4p_kernel<BT, EPI>, of_dim3{(unsigned)grid, 1, 1}, 256, SMEM_W4P, s, a);
}
}  // namespace

bool of_gemm_w4p_eligible(const OfGemmArgs& a) {
    if ((a.M % PT_M) || (a.N % PT_N) || (a.K % DK) || a.M <= 0 || a.N <= 0 || a.K < P_MIN_STAGES * DK) return false;
    if (a.a_trans || a.group_kind) return false;
    const unsigned long long a_span = 2ull * (unsigned long long)PT_M * (unsigned long long)a.lda;
    const unsigned long long b_span = 2ull * (unsigned long long)(a.b_trans ? a.K : PT_N) * (unsigned long long)a.ldb;
    const unsigned long long c_span = 2ull * 128ull * (unsigned long long)a.ldc;
    if (a_span >= (1ull << 32) || b_span >= (1ull << 32) || c_span >= (1ull << 32)) return false;
    if ((a.ldc & 7) || ((uintptr_t)a.C & 15) || (a.C2 && ((uintptr_t)a.C2 & 15))) return false;      // 16-byte output stores
    switch (a.epi) {
        case OF_EPI_STORE_BF16: return true;
        case OF_EPI_GELU: return !a.b_trans;
    }
    return false;
}

int of_gemm_w4p_try(const OfGemmArgs& a, of_stream_t s) {
    if (!of_gemm_w4p_eligible(a)) return OF_E_SHAPE;
    if (!a.b_trans) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4p<false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_w4p<false, OF_EPI_GELU>(a, s);
        }
    } else {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4p<true, OF_EPI_STORE_BF16>(a, s);
        }
    }
    return OF_E_SHAPE;
}
