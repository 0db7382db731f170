// Epilogue of the 4-wave 256x256 kernel (gemm_w4.hip; kept apart from its K loop: a second K loop on a ring of four half
// stages shared it -- built, parity-green, 2-8 % slower, removed after commit dd27936): a wave owns 128 x 128 = 4 x 4 accumulators of 32 x 32 and sends them as eight 32 x 64 groups through a private LDS patch.
// `ring_bytes` = size of the (now idle) operand ring at the start of dynamic LDS; the *_DOT epilogues keep one 4-KiB aux
// buffer per wave behind it (requested before the K loop) and one inside it.  ASMDMA: LDS-DMA form of the kernel (of_platform.h).
#pragma once
#include "gemm_tile256.h"

namespace oft {

// `to_patch(g, patch)` writes the wave's accumulator group g (rows 32 (g >> 1).., columns 64 (g & 1).. of its 128 x 128) into the
// patch: ofg::patch_write32 for 32x32x16 accumulators, ofg::patch_write16 for 16x16x32 ones.
// GATE_RESID with the residual tiles by LDS-DMA (RESID_DMA: the caller has requested group 0's tile into the 8-KiB slot behind
// the ring before its K loop and launched with 4 * ofg::RESID_LDS_BYTES of LDS behind the ring): three groups in flight -- slot 0
// behind the ring, slots 1 / 2 behind the patches inside the idle ring (4 * PATCH_BYTES + 256 + 8 * 8 KiB = 100608 B); group g + 3
// takes group g's slot.  vmcnt by hand as for the *_DOT epilogues: PC pieces and ST stores per group.
// PRE = false: nothing was requested before the K loop (the 32 KiB behind the ring hold the kernel's third image of B instead); all
// three slots lie inside the idle ring and group 0's tile is requested here, in front of groups 1 and 2.
template <bool ASMDMA, bool F32, bool PRE, class ToPatch>
OF_DEV void w4_epilogue_resid_dma(const OfGemmArgs& p, ToPatch to_patch, char* smem, int ring_bytes, int m0, int n0, int wm, int wn, int wave,
                                  int lane, float gv, float sc, char* patch) {
    constexpr int PC = F32 ? 8 : 4, ST = F32 ? 8 : 4;
    float dot = 0.f;
    auto slot = [&](int g) OF_INLINE_LAMBDA -> char* {
        const int sl = g % 3;
        if (!PRE) return smem + 4 * ofg::PATCH_BYTES + 256 + (sl * 4 + wave) * ofg::RESID_LDS_BYTES;
        return sl == 0 ? smem + ring_bytes + wave * ofg::RESID_LDS_BYTES
                       : smem + 4 * ofg::PATCH_BYTES + 256 + ((sl - 1) * 4 + wave) * ofg::RESID_LDS_BYTES;
    };
    auto request = [&](int g) OF_INLINE_LAMBDA {
        ofg::epilogue_group_resid_dma<ASMDMA>(p, m0 + wm * 128 + (g >> 1) * 32, n0 + wn * 128 + (g & 1) * 64, lane, slot(g));
    };
    of_wait_vm<0>();
    if (!PRE) request(0);
    request(1);
    request(2);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g == 0 && !PRE) of_wait_vm<2 * PC>();
        if (g == 1) of_wait_vm<PC + ST + PC>();
        if (g >= 2 && g <= 5) of_wait_vm<2 * (ST + PC)>();
        if (g == 6) of_wait_vm<ST + PC + ST>();
        if (g == 7) of_wait_vm<2 * ST>();
        to_patch(g, patch);
        ofg::epilogue_group_rows_residlds<F32>(p, patch, slot(g), m0 + wm * 128 + (g >> 1) * 32, n0 + wn * 128 + (g & 1) * 64, lane, gv, sc, dot);
        if (g < 5) request(g + 3);
    }
}

// dot_slot: where this tile's gate-gradient partial goes in the workspace (-1: the workgroup's id)
template <int EPI, bool ASMDMA, class ToPatch, bool RESID_DMA = false, bool RESID_PRE = true>
OF_DEV void w4_epilogue_with(const OfGemmArgs& p, ToPatch to_patch, char* smem, int ring_bytes, int m0, int n0, int wm, int wn, int wave, int lane,
                             int dot_slot = -1) {
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
    if constexpr (RESID_DMA && EPI == OF_EPI_GATE_RESID) {
        if (p.io_f32) w4_epilogue_resid_dma<ASMDMA, true, RESID_PRE>(p, to_patch, smem, ring_bytes, m0, n0, wm, wn, wave, lane, gv, sc, patch);
        else w4_epilogue_resid_dma<ASMDMA, false, RESID_PRE>(p, to_patch, smem, ring_bytes, m0, n0, wm, wn, wave, lane, gv, sc, patch);
        return;
    }
    // (Round 4 tried the aux tiles SIX groups deep -- five more 4-KiB slots per wave inside the idle ring -- on the reading that every
    // group waited out a global-load latency: no gain, DGELU_DOT +1..+6 % (profiles/r04b_*, r04h_*).  tools/probes/tile_phase_probe.py
    // then showed what the 18 us of a DGELU_DOT epilogue are: ~9 us of VALU issue (erf-GELU and its derivative: two quarter-rate
    // transcendentals + ~30 full-rate operations per element on ONE wave per SIMD) and ~5 us of dependent LDS round trips in
    // the rolled row loop, on top of the 3.6 us of a plain store -- not memory latency.)
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
    ofg::epilogue_finish<EPI>(p, dot, lane, wave, 4, (float*)(smem + 4 * ofg::PATCH_BYTES), dot_slot < 0 ? of_bid_x() : dot_slot);
}

template <int EPI, bool ASMDMA>
OF_DEV void w4_epilogue(const OfGemmArgs& p, f32x16 (&acc)[4][4], char* smem, int ring_bytes, int m0, int n0, int wm, int wn, int wave, int lane) {
    w4_epilogue_with<EPI, ASMDMA>(
        p, [&](int g, char* patch) OF_INLINE_LAMBDA { ofg::patch_write32(patch, acc[g >> 1][(g & 1) * 2], acc[g >> 1][(g & 1) * 2 + 1], lane); }, smem,
        ring_bytes, m0, n0, wm, wn, wave, lane);
}

}  // namespace oft
