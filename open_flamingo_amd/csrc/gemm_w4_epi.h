// Epilogue of the 4-wave 256x256 kernel (gemm_w4.hip; kept apart from its K loop: a second K loop on a ring of four half
// stages shared it -- built, parity-green, 2-8 % slower, removed after commit dd27936): a wave owns 128 x 128 = 4 x 4 accumulators of 32 x 32 and sends them as eight 32 x 64 groups through a private LDS patch.
// `ring_bytes` = size of the (now idle) operand ring at the start of dynamic LDS; the *_DOT epilogues keep one 4-KiB aux
// buffer per wave behind it (requested before the K loop) and one inside it.  ASMDMA: LDS-DMA form of the kernel (of_platform.h).
#pragma once
#include "gemm_tile256.h"

namespace oft {

// `to_patch(g, patch)` writes the wave's accumulator group g (rows 32 (g >> 1).., columns 64 (g & 1).. of its 128 x 128) into the
// patch: ofg::patch_write32 for 32x32x16 accumulators, ofg::patch_write16 for 16x16x32 ones.
template <int EPI, bool ASMDMA, class ToPatch>
OF_DEV void w4_epilogue_with(const OfGemmArgs& p, ToPatch to_patch, char* smem, int ring_bytes, int m0, int n0, int wm, int wn, int wave, int lane) {
    constexpr bool AUXL = EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT;
    // ---------------------------------------------------------------- epilogue, staged through LDS (as gemm_pp.hip)
    // Each wave sends its eight 32 x 64 accumulator groups through a private LDS patch (ofg::epilogue_group); the aux row
    // segments of group g + 1 are requested before group g is processed (group 0's right here: this kernel has no registers
    // to spare across the K loop).
    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    const float sc = gv * p.alpha;
    float dot = 0.f;
    char* patch = smem + wave * ofg::PATCH_BYTES;
    if constexpr (AUXL) {
        // *_DOT epilogues: aux tiles by DMA, alternating between two 4-KiB buffers per wave -- E behind the ring (group 0 was
        // requested before the K loop) and R inside the idle ring; vmcnt counted by hand (a group issues 4 stores): gemm_pp.hip
        char* bufE = smem + ring_bytes + wave * ofg::AUX_LDS_BYTES;
        char* bufR = smem + 4 * ofg::PATCH_BYTES + 256 + wave * ofg::AUX_LDS_BYTES;
        of_wait_vm<0>();
        ofg::epilogue_group_aux_dma<ASMDMA>(p, m0 + wm * 128, n0 + wn * 128 + 64, lane, bufR);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int mt = g >> 1, np = g & 1;
            if (g >= 1 && g < 7) ofg::epilogue_group_aux_dma<ASMDMA>(p, m0 + wm * 128 + ((g + 1) >> 1) * 32, n0 + wn * 128 + ((g + 1) & 1) * 64, lane, (g & 1) ? bufE : bufR);
            if (g >= 1 && g < 7) of_wait_vm<8>();
            if (g == 7) of_wait_vm<4>();
            to_patch(g, patch);
            ofg::epilogue_group_rows_auxlds<EPI>(p, patch, (g & 1) ? bufR : bufE, m0 + wm * 128 + mt * 32, n0 + wn * 128 + np * 64, lane, gv, sc, dot);
        }
    } else {
        ofg::AuxPre pre[2][4];
        ofg::epilogue_group_aux<EPI>(p, m0 + wm * 128, n0 + wn * 128, lane, pre[0]);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int mt = g >> 1, np = g & 1;
            if (g < 7) ofg::epilogue_group_aux<EPI>(p, m0 + wm * 128 + ((g + 1) >> 1) * 32, n0 + wn * 128 + ((g + 1) & 1) * 64, lane, pre[(g + 1) & 1]);
            to_patch(g, patch);
            ofg::epilogue_group_rows<EPI>(p, patch, m0 + wm * 128 + mt * 32, n0 + wn * 128 + np * 64, lane, gv, sc, dot, pre[g & 1]);
        }
    }
    ofg::epilogue_finish<EPI>(p, dot, lane, wave, 4, (float*)(smem + 4 * ofg::PATCH_BYTES), of_bid_x());
}

template <int EPI, bool ASMDMA>
OF_DEV void w4_epilogue(const OfGemmArgs& p, f32x16 (&acc)[4][4], char* smem, int ring_bytes, int m0, int n0, int wm, int wn, int wave, int lane) {
    w4_epilogue_with<EPI, ASMDMA>(
        p, [&](int g, char* patch) OF_INLINE_LAMBDA { ofg::patch_write32(patch, acc[g >> 1][(g & 1) * 2], acc[g >> 1][(g & 1) * 2 + 1], lane); }, smem,
        ring_bytes, m0, n0, wm, wn, wave, lane);
}

}  // namespace oft
