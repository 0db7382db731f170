// 256x256 bf16 MFMA GEMM tile with an LDS-DMA operand pipeline (gfx950) -- the fast path of of_gemm for
// tile-aligned shapes (M % 256 == 0, N % 256 == 0, K % 32 == 0: every GEMM of the OpenFlamingo model family at
// the benchmark batch sizes).  Same math, layouts and epilogues as gemm.hip.
//
// Why a second kernel: the 128x128 register-staged kernel spends more LDS cycles (ds_write_b128 staging + fragment
// reads) than MFMA cycles per K-tile.  Here
//   * the tile is 256x256 (8 waves as 2(M) x 4(N), 128x64 outputs per wave = 8x4 MFMA 16x16x32 fragments): every
//     staged operand byte feeds twice as many MFMAs;
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4, 16 B/lane, no VGPR round trip, no ds_write);
//   * K advances in 32-wide stages through a 4-slot LDS ring (4 x (16 KiB A + 16 KiB B) = 128 KiB, one workgroup
//     per CU): stage g+3 is issued while stage g is multiplied, so two full stages of MFMA work (~2k cycles) cover
//     the HBM/L2 latency; waits are counted (s_waitcnt vmcnt(8/4/0)), never a drain, and there is exactly one bare
//     s_barrier per stage:
//         wait(stage g landed for my loads) ; barrier   -> everybody's part of stage g landed AND everybody has
//                                                           finished reading slot (g-1)%4
//         issue DMA for stage g+3 into slot (g-1)%4 ; 12 fragment reads of slot g%4 ; 32 MFMAs
//   * the LDS images are lane-linear for the DMA (dest = wave base + lane*16) and swizzled on the SOURCE address:
//       K-contiguous operand: [256 rows][32 k] (64-B rows); 16-B slot s of row r lives at slot s ^ f(r),
//           f = {0,3,2,1}[(r>>2)&3]  -> conflict-free ds_read_b128 fragment reads
//       K-strided operand:    [32 k][256 cols] (512-B rows); 32-B chunk c of k-row r lives at chunk
//           c ^ ((r&3) | ((r>>3)&1)<<2) -> conflict-free ds_read_b64_tr_b16 (transpose) fragment reads
#include "gemm_common.h"

namespace {

constexpr int TM = 256, TN = 256, SK = 32;
constexpr int OPER_BYTES = 256 * 32 * 2;       // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OPER_BYTES;
constexpr int NSTAGE = 4;
constexpr int SMEM256 = NSTAGE * STAGE_BYTES;  // 128 KiB

OF_DEV int fN(int row) { return (4 - ((row >> 2) & 3)) & 3; }
OF_DEV int fT(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// issue this thread's two 16-byte DMA pieces of one operand stage
template <bool TR>
OF_DEV void stage_issue(const bf16_t* __restrict__ base, long ld, int row0, int k0, char* oper, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * 8 + wave;  // 1-KiB chunk of the 16-KiB image written by this wave-instruction
        const bf16_t* src;
        if (!TR) {
            const int row = c * 16 + (lane >> 2);
            const int lslot = (lane & 3) ^ fN(row);
            src = base + (size_t)(row0 + row) * ld + k0 + lslot * 8;
        } else {
            const int krow = c * 2 + (lane >> 5);
            const int pc = (lane & 31) >> 1, half = lane & 1;
            const int col = ((pc ^ fT(krow)) << 4) + half * 8;
            src = base + (size_t)(k0 + krow) * ld + row0 + col;
        }
        of_glds16(src, oper + c * 1024);
    }
}
template <bool TR>
OF_DEV s16x8 frag256(const char* oper, int row_base, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!TR) {
        const int row = row_base + i;
        return *(const s16x8*)(oper + row * 64 + ((g ^ fN(row)) << 4));
    } else {
        s16x8 f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int krow = g * 8 + h * 4 + (i >> 2);
            const int col = row_base + (i & 3) * 4;
            s16x4 t = of_lds_tr(oper + krow * 512 + ((((col >> 4)) ^ fT(krow)) << 5) + ((col & 15) << 1));
            f[h * 4 + 0] = t[0];
            f[h * 4 + 1] = t[1];
            f[h * 4 + 2] = t[2];
            f[h * 4 + 3] = t[3];
        }
        return f;
    }
}

template <bool AT, bool BT, int EPI>
OF_GLOBAL void OF_BOUNDS(512, 2) of_gemm256_kernel(OfGemmArgs p) {
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_m = p.M / TM, tiles_n = p.N / TN;
    int pm, pn;
    ofg::tile_coords(of_bid_x(), of_gdim_x(), tiles_m, tiles_n, pm, pn);
    const int m0 = pm * TM, n0 = pn * TN;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = p.K / SK;
    // prologue: stages 0..2 in flight
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        if (s < nsteps) {
            stage_issue<AT>(p.A, p.lda, m0, s * SK, smem + s * STAGE_BYTES, wave, lane);
            stage_issue<BT>(p.B, p.ldb, n0, s * SK, smem + s * STAGE_BYTES + OPER_BYTES, wave, lane);
        }
    }
    for (int gi = 0; gi < nsteps; ++gi) {
        // my own DMA pieces of stage gi have landed once at most the later stages' pieces (4 per stage) are pending
        const int later = nsteps - 1 - gi;
        if (later >= 2) of_wait_vm<8>();
        else if (later == 1) of_wait_vm<4>();
        else of_wait_vm<0>();
        of_barrier_raw();
        if (gi + NSTAGE - 1 < nsteps) {
            char* dst = smem + ((gi + NSTAGE - 1) & (NSTAGE - 1)) * STAGE_BYTES;
            stage_issue<AT>(p.A, p.lda, m0, (gi + NSTAGE - 1) * SK, dst, wave, lane);
            stage_issue<BT>(p.B, p.ldb, n0, (gi + NSTAGE - 1) * SK, dst + OPER_BYTES, wave, lane);
        }
        const char* sa = smem + (gi & (NSTAGE - 1)) * STAGE_BYTES;
        const char* sb = sa + OPER_BYTES;
        s16x8 fa[8], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = frag256<BT>(sb, wn * 64 + t * 16, lane);
#pragma unroll
        for (int t = 0; t < 8; ++t) fa[t] = frag256<AT>(sa, wm * 128 + t * 16, lane);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = of_mfma(fb[nt], fa[mt], acc[mt][nt]);
    }

    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    const float sc = gv * p.alpha;
    float dot = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            ofg::epilogue_frag<EPI>(p, acc[mt][nt], m0 + wm * 128 + mt * 16 + i16, n0 + wn * 64 + nt * 16 + g * 4, gv, sc, dot);
    ofg::epilogue_finish<EPI>(p, gv, dot, lane);
}

template <bool AT, bool BT, int EPI>
int launch256(const OfGemmArgs& a, of_stream_t s) {
    of_dim3 grid{(unsigned)((a.M / TM) * (a.N / TN)), 1, 1};
    return of_launch(of_gemm256_kernel<AT, BT, EPI>, grid, 512, SMEM256, s, a);
}

}  // namespace

int of_gemm256_try(const OfGemmArgs& a, of_stream_t s) {
    if ((a.M % TM) || (a.N % TN) || (a.K % SK)) return OF_E_SHAPE;
    const int layout = a.a_trans * 2 + a.b_trans;
    if (layout == 0) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch256<false, false, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_GELU: return launch256<false, false, OF_EPI_GELU>(a, s);
            case OF_EPI_GATE_RESID: return launch256<false, false, OF_EPI_GATE_RESID>(a, s);
            case OF_EPI_ACC_F32: return launch256<false, false, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 1) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch256<false, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_DGELU_DOT: return launch256<false, true, OF_EPI_DGELU_DOT>(a, s);
            case OF_EPI_SCALE_DOT: return launch256<false, true, OF_EPI_SCALE_DOT>(a, s);
            case OF_EPI_ACC_F32: return launch256<false, true, OF_EPI_ACC_F32>(a, s);
        }
    } else if (layout == 3) {
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch256<true, true, OF_EPI_STORE_BF16>(a, s);
            case OF_EPI_ACC_F32: return launch256<true, true, OF_EPI_ACC_F32>(a, s);
        }
    }
    return OF_E_SHAPE;
}
