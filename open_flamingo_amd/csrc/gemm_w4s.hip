// Persistent, wave-specialised 256 x 128 bf16 MFMA GEMM tile (gfx950): ONE workgroup of EIGHT waves per CU -- four CONSUMER waves
// (128 x 64 each: 64 MFMAs and 24 fragment reads per K stage, nothing else) and four PRODUCER waves that issue every LDS-DMA piece
// and run the epilogue of the workgroup's PREVIOUS tile while the consumers are in the K loop of the next one.
//
// Why this form (round 5's probes).  The K loops of the big-tile kernels run at the chip's power cap (~1.54
// "MFMA-GHz"); what a K = 2048 launch loses is the time in which no MFMA issues -- prologue, epilogue, drain, dispatch: 14 % of a
// plain-store launch, 25-33 % of the erf-GELU / *_DOT launches.  Two attempts to fill it:
//   * gemm_w4h.hip, two workgroups per CU: a K loop that has its SIMD to itself issues MFMAs 68 % of the time (the wave is held at
//     the texture unit for ~40 cycles per LDS-DMA piece it issues, 12 pieces per 64 MFMAs) and an epilogue wave beside a K-loop wave
//     gets a third of the issue slots: -2..-5 % on the *_DOT launches, nothing elsewhere;
//   * the same wave doing both on a fixed schedule ("gemm_w4p", removed; record: profiles/r05f_w4p_probe_single_wave_pipelined.jsonl): parity-green, 25-40 % SLOWER -- the single wave's
//     own DMA issue again, with the two-slot ring's half stage of lookahead.
// Here the MFMA waves issue no memory operation but their fragment reads; a SIMD holds one consumer and one producer wave, the
// producer's DMA issue stalls and the epilogue's VALU / store issue sit beside the consumer's MFMAs instead of in front of them.
//   * Tile hand-over: at the end of a tile's K loop a consumer rounds its 128 accumulator registers to bf16 into a private 16-KiB
//     LDS image of its 128 x 64 sub-tile (< 1 us, while the producers' requests for the next tile's first stages are in flight);
//     producer wave w + 4 drains consumer w's image during the first 16 K stages of the next tile, one 8-row chunk per stage: one
//     ds_read_b128, the epilogue math, one 16-byte store per output.  bf16 is what every epilogue taken here stores, and rounding
//     the product first is the reference's own order (autocast: the Linear's bf16 output feeds GELU).
//   * Ring: six 16-KiB units = two stages of (B, A rows 0-63, A rows 64-127 of both consumer rows), the phases row-half major as
//     in gemm_w4h.hip -- B(d), A0(d) are dead after phase 0, A1(d) after phase 2 -- so that a unit of stage d + 2 is requested 1.5
//     stages before its first read.  Two barriers per stage (all eight waves): b1 behind phase 0, b2 behind phase 2.
//     LDS: 96 KiB ring + 64 KiB images = the CU's 160 KiB; <= 256 registers per wave.
//   * Tiles: workgroup b takes tiles b, b + G, b + 2G, ... (G = 256, or the tile count, or OfGemmArgs.cu_limit) in the XCD-aware order;
//     the last tile's image is drained without a K loop to hide under.
// Epilogues: the bf16-output ones (STORE_BF16, GELU with one or two outputs, DGELU_DOT, SCALE_DOT); K >= 1152 (18 stages).
// Results: STORE_BF16 and SCALE_DOT outputs are bit-equal to the 256 x 256 kernel's; GELU / DGELU_DOT round the product to bf16 before
// the epilogue math (GELU's pre-activation output is bit-equal); gate-gradient partials are per (tile, wave): deterministic.
#include <type_traits>
#include "gemm_tile256.h"

namespace {
using namespace oft;

constexpr int ST_M = 256, ST_N = 128;
constexpr int S_UNIT = HALF_BYTES;                   // 16 KiB: 128 rows (or columns) x 64 of K
constexpr int S_STAGE = 3 * S_UNIT;                  // B, A0, A1
constexpr int S_RING = 2 * S_STAGE;                  // 96 KiB
constexpr int S_IMG = 16384;                         // per consumer wave: 128 rows x 64 bf16, 16-byte slot s of row r at slot s ^ (r & 7)
constexpr int SMEM_W4S = S_RING + 4 * S_IMG;         // 160 KiB
constexpr int S_CHUNKS = 16;                         // 8-row chunks of an image: one per K stage
constexpr int S_MIN_STAGES = S_CHUNKS + 2;

OF_DEV f32x2 s_unpack(unsigned w) { return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; }

template <bool BT, int EPI>
OF_GLOBAL void OF_BOUNDS(512, 2) of_gemm_w4s_kernel(OfGemmArgs p) {
    // LDS-DMA by inline asm in every instantiation: the producers' stream carries the hand-counted vmcnt waits of the ring next to
    // ordinary stores (and LDS reads of the images); through the builtins hipcc would order those against the pending DMA itself
    constexpr bool ASMD = true;
    constexpr bool DOT = EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT;
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63;
    const int wave8 = of_uniform(tid >> 6);
    const bool producer = wave8 >= 4;
    const int wave = wave8 & 3;           // consumer: its 128 x 64; producer: the DMA chunks it issues and the consumer whose image it drains
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = p.M / ST_M, tiles_n = p.N / ST_N, ntiles = tiles_m * tiles_n;
    const int nd = p.K / DK;
    const int G = of_gdim_x(), bid = of_bid_x();
    const int my_tiles = (ntiles - bid + G - 1) / G;
    char* img = smem + S_RING + wave * S_IMG;

    if (producer) {
        // ================================================================================================ producer wave
        // DMA duty: 1-KiB chunks q = jj*4 + wave (jj = 0..3) of every unit (chunk q of an A unit = 8 rows of consumer row q >> 3; the
        // row half of the unit is a scalar offset of 64 rows)
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
        // epilogue duty: chunk c of the image = rows 8c .. 8c + 7; a lane owns 8 consecutive columns of one row
        const int img_rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);      // + c * 1024
        const unsigned out_off = 2u * (unsigned)((lane >> 3) * p.ldc + (lane & 7) * 8);      // bytes; + c * 8 rows (scalar)
        const unsigned out_chunk = 2u * 8u * (unsigned)p.ldc;
        const unsigned aux_off = 2u * (unsigned)((lane >> 3) * p.ldaux + (lane & 7) * 8);
        const unsigned aux_chunk = 2u * 8u * (unsigned)p.ldaux;
        float gv = 1.0f;
        if (p.gate) gv = of_tanh(*p.gate);
        const float sc = gv * p.alpha;
        of_buf_t outC = of_buf_make(p.C), outC2 = of_buf_make(p.C), auxB = of_buf_make(p.C);      // re-based per tile
        float dot = 0.f;
        int dot_slot = 0;
        // *_DOT: the saved activation (aux) of a chunk.  Chunk 0's comes through registers (requested at the tile switch, in front of
        // the ring's first requests); chunk c + 1's by LDS-DMA into the image slot of chunk c the moment that one has been read -- one
        // piece, a whole K stage ahead of its use, counted by hand like the ring's (an ordinary load here would have hipcc wait for
        // every LDS-DMA piece in flight in front of its first use: the producer would arrive late at every barrier).
        u32x4 xr0 = {0, 0, 0, 0};
        const unsigned img_u = of_lds_base(img);
        auto aux_request = [&](int c) OF_INLINE_LAMBDA {       // aux chunk c -> image slot c - 1 (c >= 1)
            of_wait_lgkm0();                                    // the reads of slot c - 1 are done ...
            of_wave_sync();                                     // ... by every lane (a compiler fence on hardware; the emulator's rendezvous)
            of_buf_load16_lds_at<ASMD>(auxB, aux_off, (unsigned)c * aux_chunk, img_u + (unsigned)(c - 1) * 1024u);
        };
        auto drain_chunk = [&](int c) OF_INLINE_LAMBDA {
            const u32x4 raw = *(const u32x4*)(img + c * 1024 + img_rd);
            const unsigned so = (unsigned)c * out_chunk;
            if (EPI == OF_EPI_STORE_BF16) {
                of_buf_store16(outC, out_off, so, raw);
            } else if (EPI == OF_EPI_GELU) {
                if (p.C2) of_buf_store16(outC2, out_off, so, raw);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 g = of_gelu2(s_unpack(raw[q]));
                    o[q] = of_pack_bf16(g[0], g[1]);
                }
                of_buf_store16(outC, out_off, so, o);
            } else {      // *_DOT: x = the saved activation (aux), a = the rounded product
                u32x4 xr = xr0;
                if (c > 0) xr = *(const u32x4*)(img + (c - 1) * 1024 + lane * 16);
                u32x4 o;
                f32x2 d2 = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 av = s_unpack(raw[q]), xv = s_unpack(xr[q]);
                    f32x2 ov;
                    if (EPI == OF_EPI_DGELU_DOT) {
                        f32x2 ge, dg;
                        of_gelu_both2(xv, ge, dg);
                        d2 = of_fma2(ge, av, d2);
                        ov = av * dg * sc;
                    } else {
                        d2 = of_fma2(xv, av, d2);
                        ov = av * sc;
                    }
                    o[q] = of_pack_bf16(ov[0], ov[1]);
                }
                dot += d2[0] + d2[1];
                of_buf_store16(outC, out_off, so, o);
                if (c + 1 < S_CHUNKS) aux_request(c + 1);
            }
        };
        auto tile_done = [&]() OF_INLINE_LAMBDA {       // the gate-gradient partial of (tile, wave) -> its slot of the workspace
            if (DOT && p.dot_out) {
                const float s = of_wave_sum(dot);
                if (lane == 0) ((float*)p.workspace)[dot_slot] = s;
                dot = 0.f;
            }
        };

        for (int seg = 0; seg <= my_tiles; ++seg) {
            const bool has_tile = seg < my_tiles;
            int pm = 0, pn = 0;
            if (has_tile) ofg::tile_coords(bid + seg * G, ntiles, tiles_m, tiles_n, pm, pn);
            const int m0 = pm * ST_M, n0 = pn * ST_N;
            const of_buf_t gA = of_buf_make(chunk_base<false>(p.A, p.lda, m0));
            const of_buf_t gB = of_buf_make(chunk_base<BT>(p.B, p.ldb, n0));
            // piece j (0..11: unit j >> 2 = 0: B, 1: A0, 2: A1; chunk jj = j & 3 of this wave) of stage e into its slot
            auto dma_piece = [&](int j, int e) OF_INLINE_LAMBDA {
                const int kind = j >> 2, jj = j & 3;
                const unsigned dst = smem_u + (unsigned)((e & 1) * S_STAGE + kind * S_UNIT + jj * 4096);
                if (kind == 0) of_buf_load16_lds_at<ASMD>(gB, offB[jj], (unsigned)e * stepB, dst);
                else of_buf_load16_lds_at<ASMD>(gA, offA[jj], (unsigned)e * stepA + (kind == 2 ? halfA : 0u), dst);
            };
            if (DOT && seg > 0) xr0 = of_buf_load16(auxB, aux_off, 0u);      // aux chunk 0 of the tile about to be drained
            if (!has_tile) {      // ---- the last tile's image: behind the consumers' dump, no K loop to hide under
                of_barrier_raw();                         // T: the image is complete
                for (int c = 0; c < S_CHUNKS; ++c) {
                    if (DOT) of_wait_vm<0>();             // the aux chunk requested by the previous pass has landed
                    drain_chunk(c);
                }
                tile_done();
                break;
            }
            // ---- the ring is idle (the consumers are behind b2 of the previous tile's last stage): stages 0 and 1
#pragma unroll
            for (int j = 0; j < 24; ++j) dma_piece(j % 12, j / 12);
            of_wait_vm<16>();          // B(0), A0(0) have landed (behind them: A1(0) and stage 1)
            of_barrier_raw();          // T: ... everybody's; the consumers' image of the previous tile is complete
            const bool drain = seg > 0;
            int d = 0;
            // stage d: E1 / E2 = stage d + 1 / d + 2 exists
            // (All twelve pieces of a stage are the producers'.  Measured alternatives, profiles/r05i_*, r05j_*: with 8 pieces per producer
            // and stage (timing only) the K loop runs at the 256 x 256 kernel's rate, with 12 it is 8 % behind -- it is bound by the
            // producers' DMA issue; handing the B unit to the consumers (4 pieces per wave and stage in their MFMA gaps) is 4 % WORSE.
            // The two barrier intervals of a stage are the same 32 MFMAs long for the consumers, so the producers issue SIX pieces in
            // each: B and half of A0 of stage d + 2 behind b1 (their slots are free there), the rest of A0 and A1 behind b2.)
            auto stage = [&](const bool E1, const bool E2) OF_INLINE_LAMBDA {
                if (drain && d < S_CHUNKS) {
                    if (DOT && d > 0) of_wait_vm<12>();          // aux chunk d has landed (behind it: last stage's requests)
                    drain_chunk(d);       // (its stores -- and aux request -- are younger than every piece waited for below)
                }
                if (E1) of_wait_vm<12>();    // A1(d) has landed (behind it: stage d + 1)
                else of_wait_vm<0>();
                of_barrier_raw();            // b1: B(d), A0(d) are free
                if (E2) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) dma_piece(j, d + 2);
                }
                if (E1 && E2) of_wait_vm<10>();      // B, A0 of stage d + 1 have landed (behind them: A1(d+1), six pieces of stage d + 2)
                else if (E1) of_wait_vm<4>();
                else of_wait_vm<0>();
                of_barrier_raw();            // b2: A1(d) is free
                if (E2) {
#pragma unroll
                    for (int j = 6; j < 12; ++j) dma_piece(j, d + 2);
                }
                ++d;
            };
            for (; d + 2 < nd;) stage(true, true);
            stage(true, false);
            stage(false, false);
            if (drain) tile_done();
            // this tile's outputs are stored during the next K loop (or by the drain above)
            const size_t o = (size_t)(m0 + wm * 128) * p.ldc + n0 + wn * 64;
            outC = of_buf_make((const bf16_t*)p.C + o);
            if (EPI == OF_EPI_GELU && p.C2) outC2 = of_buf_make((const bf16_t*)p.C2 + o);
            if (DOT) auxB = of_buf_make((const bf16_t*)p.aux + (size_t)(m0 + wm * 128) * p.ldaux + n0 + wn * 64);
            dot_slot = (pm * tiles_n + pn) * 4 + wave;
        }
        return;
    }

    // ==================================================================================================== consumer wave
    of_accbank_t acc;     // accumulator k = 4 (16-row block of M) + (16-column block of N), in fixed registers (of_platform.h)
    const int img_wr = (lane & 15) * 128 + ((lane >> 4) & 1) * 8;                  // + a * 2048 + (((2 b + (lane >> 5)) ^ (lane & 7)) << 4)
    float mul = 1.0f;     // STORE_BF16 scales the product before rounding (as the 256 x 256 kernel does); the others round the bare product
    if (EPI == OF_EPI_STORE_BF16) {
        mul = p.alpha;
        if (p.gate) mul *= of_tanh(*p.gate);
    }
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
    for (int seg = 0; seg < my_tiles; ++seg) {
        of_barrier_raw();              // T: stage 0's first units have landed (and the previous tile's image is the producers')
        of_accbank_zero(acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) read8(smem + S_UNIT, smem, 0, 0, r);
        of_sched_fence();

        // One phase = 16 MFMAs: B fragments fb[ks] x A fragments fa[ph & 1] -> accumulator rows 4 (ph >> 1) ...; `rd(r)` = the r-th
        // fragment read of the NEXT phase, in the first MFMA gaps
        auto phase = [&](int ph, int nrd, auto rd) OF_INLINE_LAMBDA {
            const int ks = ph & 1, ah = ph >> 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                of_accbank_mfma(acc, (ah * 4 + (i >> 2)) * 4 + (i & 3), fb[ks][i & 3], fa[ph & 1][i >> 2]);
                if (i < nrd) rd(i);
                of_sched_fence();
            }
        };
        auto stage_body = [&](const char* cur, const char* nxt, const bool E1) OF_INLINE_LAMBDA {
            const char* uB = cur;
            const char* uA0 = cur + S_UNIT;
            const char* uA1 = cur + 2 * S_UNIT;
            of_mfma_acc_guard();       // fragments may have been moved between registers on the way into this stage (of_platform.h)
            phase(0, 8, [&](int r) OF_INLINE_LAMBDA { read8(uA0, uB, 1, 1, r); });
            of_wait_lgkm0();           // own reads of B(d), A0(d) are done ...
            of_barrier_raw();          // b1: ... everybody's; A1(d) has landed (the producers waited for it)
            of_sched_fence();
            phase(1, 4, [&](int r) OF_INLINE_LAMBDA { read_a(uA1, 0, 0, r); });
            phase(2, 4, [&](int r) OF_INLINE_LAMBDA { read_a(uA1, 1, 1, r); });
            of_wait_lgkm0();           // own reads of A1(d) are done
            of_barrier_raw();          // b2: B, A0 of stage d + 1 have landed
            of_sched_fence();
            phase(3, E1 ? 8 : 0, [&](int r) OF_INLINE_LAMBDA { read8(nxt + S_UNIT, nxt, 0, 0, r); });
        };
        int d = 0;
        for (; d + 1 < nd; ++d) stage_body(smem + (d & 1) * S_STAGE, smem + ((d + 1) & 1) * S_STAGE, true);
        stage_body(smem + (d & 1) * S_STAGE, smem + ((d + 1) & 1) * S_STAGE, false);
        // ---- the accumulators -> this wave's image (drained by its producer during the next tile's K loop, or by the final drain)
        of_mfma_acc_settle();
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 v = of_accbank_read(acc, a * 4 + b);
                const u32x2 w = {of_pack_bf16(mul * v[0], mul * v[1]), of_pack_bf16(mul * v[2], mul * v[3])};
                *(u32x2*)(img + a * 2048 + img_wr + (((2 * b + (lane >> 5)) ^ (lane & 7)) << 4)) = w;
            }
        of_wait_lgkm0();
    }
    of_barrier_raw();                  // T: the last image is complete
}

template <bool BT, int EPI>
int launch_w4s(const OfGemmArgs& a, of_stream_t s) {
    const int ntiles = (a.M / ST_M) * (a.N / ST_N);
    // one workgroup per CU the caller lets the launch count on (OfGemmArgs.cu_limit; 0 = all), whole groups of 8 for the XCD-aware tile
    // order where there are that many
    int grid = a.cu_limit > 0 && a.cu_limit < OF_NUM_CUS ? a.cu_limit : OF_NUM_CUS;
    if (grid > ntiles) grid = ntiles;
    if (grid >= 8) grid &= ~7;
    const int rc = of_launch(of_gemm_w4s_kernel<BT, EPI>, of_dim3{(unsigned)grid, 1, 1}, 512, SMEM_W4S, s, a);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, ntiles * 4, s);
}
}  // namespace

// Eligibility, separate from the launch (of_gemm's selection checks it first).  A is K-contiguous (y = x W^T and dX = dY W); 32-bit
// byte offsets as in gemm_w4m.hip (operands) and for the wave's 128-row output / aux window.
bool of_gemm_w4s_eligible(const OfGemmArgs& a) {
    if ((a.M % ST_M) || (a.N % ST_N) || (a.K % DK) || a.M <= 0 || a.N <= 0 || a.K < S_MIN_STAGES * DK) return false;
    if (a.a_trans || a.group_kind) return false;
    const unsigned long long a_span = 2ull * (unsigned long long)ST_M * (unsigned long long)a.lda;
    const unsigned long long b_span = 2ull * (unsigned long long)(a.b_trans ? a.K : ST_N) * (unsigned long long)a.ldb;
    const unsigned long long c_span = 2ull * 128ull * (unsigned long long)a.ldc, x_span = 2ull * 128ull * (unsigned long long)a.ldaux;
    if (a_span >= (1ull << 32) || b_span >= (1ull << 32) || c_span >= (1ull << 32) || x_span >= (1ull << 32)) return false;
    if ((a.ldc & 7) || ((uintptr_t)a.C & 15) || (a.C2 && ((uintptr_t)a.C2 & 15))) return false;      // 16-byte output stores
    switch (a.epi) {
        case OF_EPI_STORE_BF16: return true;
        case OF_EPI_GELU: return !a.b_trans;
        case OF_EPI_DGELU_DOT:
        case OF_EPI_SCALE_DOT: return a.b_trans && !(a.ldaux & 7) && !((uintptr_t)a.aux & 15);
    }
    return false;
}

int of_gemm_w4s_try(const OfGemmArgs& a, of_stream_t s) {
    if (!of_gemm_w4s_eligible(a)) return OF_E_SHAPE;
    if (!a.b_trans) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4s<false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch_w4s<false, OF_EPI_GELU>(a, s);
        }
    } else {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_w4s<true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch_w4s<true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch_w4s<true, OF_EPI_SCALE_DOT>(a, s);
        }
    }
    return OF_E_SHAPE;
}
