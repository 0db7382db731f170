// LDS images of one 256 x 256 x 64 bf16 operand stage (gfx950), shared by the two big-tile GEMM kernels
// (gemm_pp.hip: 8 waves, LDS-DMA, ping-pong; gemm_w4.hip: 4 waves, 128x128 per wave, register staged).
// A stage = A image (32 KiB) + B image (32 KiB).  Images are lane-linear for a wave-wide 1-KiB piece
// (dest = piece base + lane*16) and swizzled on the SOURCE address, so the same image can be filled by LDS-DMA
// (global_load_lds: the destination cannot scatter) or by ds_write_b128 (sequential, conflict-free):
//   K-contiguous operand: each 16-KiB half (tile rows 0-127 / 128-255) is 16 1-KiB chunks of 8 rows x 128 B stored
//       [k-half][8 rows][64 B]; 16-B slot s of row r lives at slot s ^ f(r), f = {0,3,2,1}[(r>>2)&3]
//       -> conflict-free ds_read_b128 reads of 32-row MFMA fragments (SQ_LDS_BANK_CONFLICT 0 measured)
//   K-strided operand:    chunk = 4 k-rows x 256 B (128 columns); 32-B piece c of k-row r lives at c ^ ((r&3)<<1)
//       -> conflict-free ds_read_b64_tr_b16 (transpose) reads
#pragma once
#include "gemm_common.h"

namespace oft {

constexpr int TM = 256, TN = 256;
constexpr int DK = 64;                          // K depth of one DMA stage (full 128-byte lines of a K-contiguous row)
constexpr int OPER_BYTES = 256 * DK * 2;        // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OPER_BYTES;     // 64 KiB
constexpr int NSLOT = 2;
constexpr int SMEM_PP = NSLOT * STAGE_BYTES;    // 128 KiB

OF_DEV int fN(int row) { return (4 - ((row >> 2) & 3)) & 3; }
OF_DEV int fT(int krow) { return (krow & 3) << 1; }

constexpr int HALF_BYTES = OPER_BYTES / 2;        // 16 KiB: tile rows (or columns) 0-127 / 128-255 of one operand

// per-thread global source of 1-KiB chunk c (0..15) of half hf of one operand at k0 = 0 (advanced by DK [* ld] per stage)
template <bool TR>
OF_DEV const bf16_t* chunk_src(const bf16_t* __restrict__ base, long ld, int row0, int hf, int c, int lane) {
    if (!TR) {
        const int row = hf * 128 + c * 8 + ((lane >> 2) & 7);
        const int kh = lane >> 5;
        const int lslot = (lane & 3) ^ fN(row);
        return base + (size_t)(row0 + row) * ld + kh * 32 + lslot * 8;
    } else {
        const int krow = c * 4 + (lane >> 4);
        const int pc = (lane & 15) >> 1, half16 = lane & 1;
        const int col = hf * 128 + ((pc ^ fT(krow)) << 4) + half16 * 8;
        return base + (size_t)krow * ld + row0 + col;
    }
}

// The same source split into a wave-uniform base and a per-lane 32-bit element offset (base + offset = chunk_src at
// k0 = 0): lets the loads use the scalar-base + vector-offset addressing form, with the per-stage advance on the scalar.
template <bool TR>
OF_DEV const bf16_t* chunk_base(const bf16_t* __restrict__ base, long ld, int row0) {
    return TR ? base + row0 : base + (size_t)row0 * ld;
}
template <bool TR>
OF_DEV unsigned chunk_off(long ld, int hf, int c, int lane) {
    if (!TR) {
        const int row = hf * 128 + c * 8 + ((lane >> 2) & 7);
        const int kh = lane >> 5;
        const int lslot = (lane & 3) ^ fN(row);
        return (unsigned)(row * ld + kh * 32 + lslot * 8);
    } else {
        const int krow = c * 4 + (lane >> 4);
        const int pc = (lane & 15) >> 1, half16 = lane & 1;
        const int col = hf * 128 + ((pc ^ fT(krow)) << 4) + half16 * 8;
        return (unsigned)(krow * ld + col);
    }
}

// this lane's 16-byte piece of a 32-row operand fragment: k-half h (32 deep) of the stage, k-step ks (16 deep) of the half
template <bool TR>
OF_DEV s16x8 frag32(const char* oper, int row_base, int h, int ks, int lane) {
    if (!TR) {
        const int row = row_base + (lane & 31);
        const int slot = ks * 2 + (lane >> 5);
        return *(const s16x8*)(oper + (row >> 7) * HALF_BYTES + ((row & 127) >> 3) * 1024 + h * 512 + (row & 7) * 64 +
                               ((slot ^ fN(row)) << 4));
    } else {
        const int q = lane >> 4, i = lane & 15;
        s16x8 f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int krow = h * 32 + ks * 16 + (q >> 1) * 8 + hh * 4 + (i >> 2);
            const int col = row_base + (q & 1) * 16 + (i & 3) * 4;
            const int cw = col & 127;
            s16x4 t = of_lds_tr(oper + (col >> 7) * HALF_BYTES + krow * 256 + ((((cw >> 4)) ^ fT(krow)) << 5) + ((cw & 15) << 1));
            f[hh * 4 + 0] = t[0];
            f[hh * 4 + 1] = t[1];
            f[hh * 4 + 2] = t[2];
            f[hh * 4 + 3] = t[3];
        }
        return f;
    }
}

OF_DEV int fT16(int krow) { return ((krow & 3) << 1) ^ ((krow >> 3) & 1); }

// ---- the same images for kernels on 16x16x32 MFMAs (gemm_w4m.hip, gemm_mid.hip): 16-row fragments.  The K-contiguous image
// serves them conflict-free as it is; the K-strided image gets one more swizzle bit (piece ^ ((k-row >> 3) & 1)): the four
// 16-lane groups of a transposed read sit 8 k-rows apart in the same 32-byte column piece.
// per-lane element offset of 1-KiB chunk c (0..15) of half hf of one operand at k0 = 0
template <bool TR>
OF_DEV unsigned mchunk_off(long ld, int hf, int c, int lane) {
    if (!TR) return chunk_off<false>(ld, hf, c, lane);
    const int krow = c * 4 + (lane >> 4);
    const int pc = (lane & 15) >> 1, half16 = lane & 1;
    const int col = hf * 128 + ((pc ^ fT16(krow)) << 4) + half16 * 8;
    return (unsigned)(krow * ld + col);
}

// this lane's 16-byte piece of a 16-row operand fragment: k-step ks (32 deep) of the stage.  Lane l holds row l & 15,
// k = 32 ks + 8 (l >> 4) + 0..7 -- the same k assignment for both operands, which is all the MFMA needs.
template <bool TR>
OF_DEV s16x8 mfrag16(const char* oper, int row_base, int ks, int lane) {
    if (!TR) {
        const int row = row_base + (lane & 15);
        const int slot = lane >> 4;       // 16-byte slot of the row's 64 bytes of k-half ks
        return *(const s16x8*)(oper + (row >> 7) * HALF_BYTES + ((row & 127) >> 3) * 1024 + ks * 512 + (row & 7) * 64 + ((slot ^ fN(row)) << 4));
    } else {
        const int q = lane >> 4, i = lane & 15;
        s16x8 f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int krow = ks * 32 + q * 8 + hh * 4 + (i >> 2);
            const int col = row_base + (i & 3) * 4;
            const int cw = col & 127;
            s16x4 t = of_lds_tr(oper + (col >> 7) * HALF_BYTES + krow * 256 + ((((cw >> 4)) ^ fT16(krow)) << 5) + ((cw & 15) << 1));
            f[hh * 4 + 0] = t[0];
            f[hh * 4 + 1] = t[1];
            f[hh * 4 + 2] = t[2];
            f[hh * 4 + 3] = t[3];
        }
        return f;
    }
}

}  // namespace oft
