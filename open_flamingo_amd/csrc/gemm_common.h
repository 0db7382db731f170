// Shared pieces of the two GEMM kernels (gemm.hip: general 128x128 register-staged tile; gemm_pp.hip: 256x256
// ping-pong LDS-DMA tile): block->tile map and the fused epilogues of include/of_hip.h.
#pragma once
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace ofg {

// XCD-aware tile order: block b runs on XCD b%8, so give each XCD a contiguous run of tile ids (bijective for any
// nwg), then walk tiles in groups of GM m-tiles so neighbouring ids share A panels and sweep n.
OF_DEV void tile_coords(int bid, int nwg, int tiles_m, int tiles_n, int& pm, int& pn) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int GM = 8;
    int width = GM * tiles_n;
    int group = id / width;
    int first_m = group * GM;
    int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    int in = id - group * width;
    pm = first_m + in % gsz;
    pn = in / gsz;
}

#if defined(OF_TOOLS_BUILD)
// tools/libofhip_tools.so only (tools/probes/interleaved_gemm_probe.py): the same walk with the group height and the XCD assignment
// as knobs -- knob & 0xff = GM (0: 8); (knob >> 8) & 15: 0 = a contiguous run of ids per XCD (the product's), 1 = id = block id (tiles
// round-robin over the XCDs), 2 = contiguous runs, groups of GM n-tiles swept along m
OF_DEV void tile_coords_knob(int bid, int nwg, int tiles_m, int tiles_n, int& pm, int& pn, int knob) {
    int GM = knob & 0xff;
    if (!GM) GM = 8;
    const int mode = (knob >> 8) & 15;
    int id = bid;
    if (mode != 1) {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ta = mode == 2 ? tiles_n : tiles_m, tb = mode == 2 ? tiles_m : tiles_n;
    int width = GM * tb;
    int group = id / width;
    int first = group * GM;
    int gsz = ta - first < GM ? ta - first : GM;
    int in = id - group * width;
    const int a = first + in % gsz, b = in / gsz;
    pm = mode == 2 ? b : a;
    pn = mode == 2 ? a : b;
}
#endif

OF_DEV void unpack4(u32x2 r, float (&x)[4]) {
    x[0] = of_bf16_to_f32((bf16_t)(r[0] & 0xffff));
    x[1] = of_bf16_to_f32((bf16_t)(r[0] >> 16));
    x[2] = of_bf16_to_f32((bf16_t)(r[1] & 0xffff));
    x[3] = of_bf16_to_f32((bf16_t)(r[1] >> 16));
}

// One accumulator fragment = C[m][n..n+3] (the MFMA is issued operand-swapped so a lane owns 4 consecutive n).
template <int EPI>
OF_DEV void epilogue_frag(const OfGemmArgs& p, const f32x4 a, int m, int n, float gv, float sc, float& dot) {
    if (m >= p.M || n >= p.N) return;
    const size_t off = (size_t)m * p.ldc + n;
    if (EPI == OF_EPI_STORE_BF16) {
        u32x2 o = {of_pack_bf16(sc * a[0], sc * a[1]), of_pack_bf16(sc * a[2], sc * a[3])};
        *(u32x2*)((bf16_t*)p.C + off) = o;
    } else if (EPI == OF_EPI_GELU) {
        if (p.C2) {
            u32x2 o = {of_pack_bf16(a[0], a[1]), of_pack_bf16(a[2], a[3])};
            *(u32x2*)((bf16_t*)p.C2 + off) = o;
        }
        u32x2 o = {of_pack_bf16(of_gelu(a[0]), of_gelu(a[1])), of_pack_bf16(of_gelu(a[2]), of_gelu(a[3]))};
        *(u32x2*)((bf16_t*)p.C + off) = o;
    } else if (EPI == OF_EPI_GATE_RESID) {
        const size_t aoff = (size_t)m * p.ldaux + n;
        if (p.io_f32) {
            const f32x4 r = *(const f32x4*)((const float*)p.aux + aoff);
            *(f32x4*)((float*)p.C + off) = f32x4{r[0] + sc * a[0], r[1] + sc * a[1], r[2] + sc * a[2], r[3] + sc * a[3]};
        } else {
            float r[4];
            unpack4(*(const u32x2*)((const bf16_t*)p.aux + aoff), r);
            u32x2 o = {of_pack_bf16(r[0] + sc * a[0], r[1] + sc * a[1]), of_pack_bf16(r[2] + sc * a[2], r[3] + sc * a[3])};
            *(u32x2*)((bf16_t*)p.C + off) = o;
        }
    } else if (EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) {
        float x[4], o[4];
        unpack4(*(const u32x2*)((const bf16_t*)p.aux + (size_t)m * p.ldaux + n), x);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (EPI == OF_EPI_DGELU_DOT) {
                float ge, dg;
                of_gelu_both(x[e], ge, dg);
                dot += ge * a[e];
                o[e] = sc * a[e] * dg;
            } else {
                dot += x[e] * a[e];
                o[e] = sc * a[e];
            }
        }
        u32x2 ov = {of_pack_bf16(o[0], o[1]), of_pack_bf16(o[2], o[3])};
        *(u32x2*)((bf16_t*)p.C + off) = ov;
    } else {  // OF_EPI_ACC_F32
        float* c = (float*)p.C + off;
        f32x4 o = {sc * a[0], sc * a[1], sc * a[2], sc * a[3]};
        if (p.beta != 0.f) {
            const f32x4 old = *(const f32x4*)c;
            o[0] += p.beta * old[0];
            o[1] += p.beta * old[1];
            o[2] += p.beta * old[2];
            o[3] += p.beta * old[3];
        }
        *(f32x4*)c = o;
    }
}

OF_DEV void unpack8(u32x4 r, float (&x)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
        x[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
    }
}
OF_DEV u32x4 pack8(const float (&o)[8]) {
    return u32x4{of_pack_bf16(o[0], o[1]), of_pack_bf16(o[2], o[3]), of_pack_bf16(o[4], o[5]), of_pack_bf16(o[6], o[7])};
}
// Eight consecutive n of one output row (tile-aligned shapes only: no bounds checks): 16-byte bf16 / 2 x 16-byte fp32
// loads and stores, eight lanes cover one full 128-byte (bf16) or 256-byte (fp32) row segment.
// The aux operand of a row segment (residual / saved activation), loaded AHEAD of the epilogue math so that its global
// latency overlaps the LDS transposition of the accumulators instead of being paid once per row group.
struct AuxPre {
    u32x4 lo, hi;      // bf16 aux: lo only (8 values); fp32 aux: lo | hi (2 x 4 values)
};
template <int EPI>
OF_DEV AuxPre epilogue_aux_load(const OfGemmArgs& p, int m, int n, int hi = 4) {
    AuxPre r{};
    if (EPI == OF_EPI_GATE_RESID) {
        const size_t aoff = (size_t)m * p.ldaux + n;
        if (p.io_f32) {
            r.lo = *(const u32x4*)((const float*)p.aux + aoff);
            r.hi = *(const u32x4*)((const float*)p.aux + aoff + hi);
        } else {
            r.lo = *(const u32x4*)((const bf16_t*)p.aux + aoff);
        }
    } else if (EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) {
        r.lo = *(const u32x4*)((const bf16_t*)p.aux + (size_t)m * p.ldaux + n);
    }
    return r;
}
// `hi`: column distance of a[4..7] from a[0..3] in the fp32-output forms (4: eight consecutive n; 32: the split lane map of
// epilogue_group_rows); the bf16-output forms always take eight consecutive n.
template <int EPI>
OF_DEV void epilogue_row8(const OfGemmArgs& p, const float (&a)[8], int m, int n, float gv, float sc, float& dot,
                          const AuxPre* pre = nullptr, int hi = 4) {
    const size_t off = (size_t)m * p.ldc + n;
    float o[8];
    if (EPI == OF_EPI_STORE_BF16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sc * a[e];
        *(u32x4*)((bf16_t*)p.C + off) = pack8(o);
    } else if (EPI == OF_EPI_GELU) {
#ifdef OF_SAVED_NT         // tools/ab builds only (round 6 A/B): tensors only the backward reads, stored with the non-temporal policy
        if (p.C2) __builtin_nontemporal_store(pack8(a), (u32x4*)((bf16_t*)p.C2 + off));
#else
        if (p.C2) *(u32x4*)((bf16_t*)p.C2 + off) = pack8(a);
#endif
#pragma unroll
        for (int e = 0; e < 8; e += 2) {          // packed fp32 math, two elements per instruction (of_platform.h)
            const f32x2 g = of_gelu2(f32x2{a[e], a[e + 1]});
            o[e] = g[0];
            o[e + 1] = g[1];
        }
        *(u32x4*)((bf16_t*)p.C + off) = pack8(o);
    } else if (EPI == OF_EPI_GATE_RESID) {
        const size_t aoff = (size_t)m * p.ldaux + n;
        if (p.io_f32) {
            const f32x4 r0 = pre ? __builtin_bit_cast(f32x4, pre->lo) : *(const f32x4*)((const float*)p.aux + aoff);
            const f32x4 r1 = pre ? __builtin_bit_cast(f32x4, pre->hi) : *(const f32x4*)((const float*)p.aux + aoff + hi);
            *(f32x4*)((float*)p.C + off) = f32x4{r0[0] + sc * a[0], r0[1] + sc * a[1], r0[2] + sc * a[2], r0[3] + sc * a[3]};
            *(f32x4*)((float*)p.C + off + hi) = f32x4{r1[0] + sc * a[4], r1[1] + sc * a[5], r1[2] + sc * a[6], r1[3] + sc * a[7]};
        } else {
            float r[8];
            unpack8(pre ? pre->lo : *(const u32x4*)((const bf16_t*)p.aux + aoff), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = r[e] + sc * a[e];
            *(u32x4*)((bf16_t*)p.C + off) = pack8(o);
        }
    } else if (EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) {
        float x[8];
        unpack8(pre ? pre->lo : *(const u32x4*)((const bf16_t*)p.aux + (size_t)m * p.ldaux + n), x);
        if (EPI == OF_EPI_DGELU_DOT) {
            f32x2 d2 = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {      // packed fp32 math, two elements per instruction (of_platform.h)
                f32x2 ge, dg;
                of_gelu_both2(f32x2{x[e], x[e + 1]}, ge, dg);
                const f32x2 av = {a[e], a[e + 1]};
                d2 = of_fma2(ge, av, d2);
                const f32x2 ov = av * dg * sc;
                o[e] = ov[0];
                o[e + 1] = ov[1];
            }
            dot += d2[0] + d2[1];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dot += x[e] * a[e];
                o[e] = sc * a[e];
            }
        }
        *(u32x4*)((bf16_t*)p.C + off) = pack8(o);
    } else {  // OF_EPI_ACC_F32
        float* c = (float*)p.C + off;
        f32x4 o0 = {sc * a[0], sc * a[1], sc * a[2], sc * a[3]}, o1 = {sc * a[4], sc * a[5], sc * a[6], sc * a[7]};
        if (p.beta != 0.f) {
            const f32x4 c0 = *(const f32x4*)c, c1 = *(const f32x4*)(c + hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] += p.beta * c0[e];
                o1[e] += p.beta * c1[e];
            }
        }
        *(f32x4*)c = o0;
        *(f32x4*)(c + hi) = o1;
        if (p.sumsq_out) {      // the gradient norm's share of these eight values (OfGemmArgs.sumsq_out; `dot` is the lane's running sum)
#pragma unroll
            for (int e = 0; e < 4; ++e) dot = __builtin_fmaf(o0[e], o0[e], __builtin_fmaf(o1[e], o1[e], dot));
        }
    }
}

// One 32(M) x 64(N) accumulator group of a wave -- two 32x32 MFMA fragments side by side -- leaves through the wave's private
// LDS patch (32 rows x 64 fp32, row pitch 272 B) so that a lane ends up with 8 consecutive n of one row: aux loads and
// output stores are 16-byte, 8 lanes cover a full 128-byte (bf16) / 256-byte (fp32) row segment.  Shared by the three
// tiled DMA kernels (gemm_pp.hip, gemm_w4.hip, gemm_mid.hip).
constexpr int PATCH_PITCH = 64 * 4 + 16;
constexpr int PATCH_BYTES = 32 * PATCH_PITCH;
template <int EPI>
constexpr bool epi_has_aux() { return EPI == OF_EPI_GATE_RESID || EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT; }
// Lane map of a row pass over the 64 columns of a group: eight lanes per row.  bf16 outputs: a lane takes 8 consecutive n (one
// 16-byte store; the 8 lanes cover a 128-byte line).  fp32 outputs (weight gradients, the fp32 stream): a lane takes n = 4j..4j+3
// and 32+4j..32+4j+3 (j = lane & 7), so that each of its two 16-byte stores / residual loads is contiguous with its
// neighbours' -- with 8 consecutive n per lane every store instruction wrote every other 16 bytes of a line and each line was
// completed by a second instruction.
template <int EPI>
OF_DEV bool epi_split_cols(const OfGemmArgs& p) { return EPI == OF_EPI_ACC_F32 || (EPI == OF_EPI_GATE_RESID && p.io_f32); }
// the four aux row segments (residual / saved activation) this lane needs for the group at (m_base, n_base)
template <int EPI>
OF_DEV void epilogue_group_aux(const OfGemmArgs& p, int m_base, int n_base, int lane, AuxPre (&pre)[4]) {
    if (!epi_has_aux<EPI>()) return;
    const bool split = epi_split_cols<EPI>(p);
    const int rd_row = lane >> 3, rd_col = (lane & 7) * (split ? 4 : 8);
#pragma unroll
    for (int it = 0; it < 4; ++it) pre[it] = epilogue_aux_load<EPI>(p, m_base + it * 8 + rd_row, n_base + rd_col, split ? 32 : 4);
}
// a0 / a1: the fragments of columns [0, 32) / [32, 64) of the group.  `pre` = epilogue_group_aux of the SAME group, requested
// a whole group earlier by the callers (software pipelining: a group's aux latency hides behind the previous group's
// transposition, math and stores; the first group's is requested before the K loop where registers allow).
// The four row passes are unrolled with a scheduling fence between them: without the fence the compiler interleaves the four
// copies of the erf-GELU math of the *_DOT epilogues on top of the live accumulators and spills.
// the wave's two 32x32 MFMA fragments of a group -> its LDS patch (lane l holds row l & 31, columns 8q + 4 (l >> 5) + 0..3)
OF_DEV void patch_write32(char* patch, const f32x16& a0, const f32x16& a1, int lane) {
    const int wr_off = (lane & 31) * PATCH_PITCH + (lane >> 5) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        *(f32x4*)(patch + wr_off + (q * 8) * 4) = f32x4{a0[4 * q], a0[4 * q + 1], a0[4 * q + 2], a0[4 * q + 3]};
        *(f32x4*)(patch + wr_off + (32 + q * 8) * 4) = f32x4{a1[4 * q], a1[4 * q + 1], a1[4 * q + 2], a1[4 * q + 3]};
    }
}
// the same group held as 2 (M) x 4 (N) fragments of 16x16 MFMAs (lane l holds row l & 15, columns 4 (l >> 4) + 0..3 of a fragment)
OF_DEV void patch_write16(char* patch, const f32x4 (&t)[2][4], int lane) {
    const int wr_off = (lane & 15) * PATCH_PITCH + (lane >> 4) * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) *(f32x4*)(patch + wr_off + a * 16 * PATCH_PITCH + b * 64) = t[a][b];
}
// ... plus another partial sum of the same group, added lane by lane to the entries this lane has just written (stream-K fix-up
// of gemm_w4m.hip: the accumulators themselves stay untouched in their accumulation registers -- arithmetic on them in C made
// hipcc park all 256 in VGPRs)
OF_DEV void patch_add16(char* patch, const f32x4 (&t)[2][4], int lane) {
    const int wr_off = (lane & 15) * PATCH_PITCH + (lane >> 4) * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4* e = (f32x4*)(patch + wr_off + a * 16 * PATCH_PITCH + b * 64);
            const f32x4 c = *e;
            *e = f32x4{c[0] + t[a][b][0], c[1] + t[a][b][1], c[2] + t[a][b][2], c[3] + t[a][b][3]};
        }
}
// the row passes of a group whose accumulators are in the patch (written by this wave, not yet synchronised)
template <int EPI>
OF_DEV void epilogue_group_rows(const OfGemmArgs& p, char* patch, int m_base, int n_base, int lane, float gv, float sc, float& dot,
                                const AuxPre (&pre)[4]) {
    const bool split = epi_split_cols<EPI>(p);
    const int hi = split ? 32 : 4;
    const int rd_row = lane >> 3, rd_col = (lane & 7) * (split ? 4 : 8);
    of_wave_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + rd_row;
        const f32x4 v0 = *(const f32x4*)(patch + r * PATCH_PITCH + rd_col * 4), v1 = *(const f32x4*)(patch + r * PATCH_PITCH + (rd_col + hi) * 4);
        const float a8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        epilogue_row8<EPI>(p, a8, m_base + r, n_base + rd_col, gv, sc, dot, epi_has_aux<EPI>() ? &pre[it] : nullptr, hi);
        if (EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT || EPI == OF_EPI_GELU) of_sched_fence();
    }
    of_wave_sync();
}
template <int EPI>
OF_DEV void epilogue_group(const OfGemmArgs& p, const f32x16& a0, const f32x16& a1, char* patch, int m_base, int n_base, int lane,
                           float gv, float sc, float& dot, const AuxPre (&pre)[4]) {
    patch_write32(patch, a0, a1, lane);
    epilogue_group_rows<EPI>(p, patch, m_base, n_base, lane, gv, sc, dot, pre);
}

// *_DOT epilogues in the big-tile kernels: the saved bf16 activation of a group travels global -> LDS by DMA instead of through
// registers (the erf-GELU math on top of the live accumulators leaves none: register prefetch spilled to scratch).  A group's
// aux tile = 32 rows x 128 B = four 1-KiB pieces; piece `it`, lane l holds row it*8 + (l>>3), bytes (l&7)*16.. -- exactly the
// vector lane l needs in row pass `it`, so the read is base + it*1024 + lane*16 (conflict-free).  Completion is the CALLER's
// job: s_waitcnt vmcnt(n) with n = vector-memory operations issued after these four (vmcnt retires in order on gfx950).
constexpr int AUX_LDS_BYTES = 4096;
// ASM: the LDS-DMA form of the kernel's K loop (of_platform.h: a kernel issues ALL its LDS-DMA either through the builtins --
// the compiler then owns M0 -- or by inline asm, never both)
template <bool ASM>
OF_DEV void epilogue_group_aux_dma(const OfGemmArgs& p, int m_base, int n_base, int lane, char* lds_dst) {
    const bf16_t* src = (const bf16_t*)p.aux + (size_t)(m_base + (lane >> 3)) * p.ldaux + n_base + (lane & 7) * 8;
#pragma unroll
    // non-temporal: the saved activation is read once, by this workgroup (134 MB per 8192 x 8192 launch) -- with the default policy it
    // pushed the B panels every workgroup of the XCD re-reads out of the 4-MB L2 (round 6, same box: the DGELU_DOT family 1090 -> 1102
    // TFLOP/s, step -0.1 ms; profiles/r06m_ab_aux_nt_step.jsonl)
    for (int it = 0; it < 4; ++it) {
#ifdef OF_AUX_NOT_NT      // tools/ab builds only: the other arm of that A/B
        of_glds16<ASM>(src + (size_t)it * 8 * p.ldaux, lds_dst + it * 1024);
#else
        of_glds16_nt<ASM>(src + (size_t)it * 8 * p.ldaux, lds_dst + it * 1024);
#endif
    }
}
// The residual of the GATE_RESID epilogue the same way (big-tile kernel on 16x16x32 MFMAs: three groups in flight instead of one
// through registers).  bf16 stream: the *_DOT layout above (4 pieces).  fp32 stream: a group's residual tile = 32 rows x 256 B =
// eight 1-KiB pieces of 4 rows; piece q, lane l holds row 4q + (l >> 4), 16-byte chunk (l & 15) ^ (8 * ((l >> 4) & 1)) of the row:
// odd rows have their two 128-byte halves exchanged, so that the split lane map of the fp32 row passes (a lane reads chunks j and
// 8 + j of row lane >> 3) hits 64 distinct banks per 16 lanes.
constexpr int RESID_LDS_BYTES = 8192;
template <bool ASM>
OF_DEV void epilogue_group_resid_dma(const OfGemmArgs& p, int m_base, int n_base, int lane, char* lds_dst) {
    if (p.io_f32) {
        const int r4 = lane >> 4, ch = (lane & 15) ^ ((r4 & 1) << 3);
        const float* src = (const float*)p.aux + (size_t)(m_base + r4) * p.ldaux + n_base + ch * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#ifdef OF_RESID_NT        // tools/ab builds only (round 6 A/B: the fp32 residual tile, also read once, with the non-temporal policy)
            of_glds16_nt<ASM>(src + (size_t)q * 4 * p.ldaux, lds_dst + q * 1024);
#else
            of_glds16<ASM>(src + (size_t)q * 4 * p.ldaux, lds_dst + q * 1024);
#endif
        }
    } else {
        epilogue_group_aux_dma<ASM>(p, m_base, n_base, lane, lds_dst);
    }
}
// row passes of a GATE_RESID group whose accumulators are in the patch and whose residual tile is in LDS (landed)
template <bool F32>
OF_DEV void epilogue_group_rows_residlds(const OfGemmArgs& p, char* patch, const char* res_lds, int m_base, int n_base, int lane, float gv,
                                         float sc, float& dot) {
    const int hi = F32 ? 32 : 4;
    const int rd_row = lane >> 3, j = lane & 7, rd_col = j * (F32 ? 4 : 8);
    of_wave_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + rd_row;
        const f32x4 v0 = *(const f32x4*)(patch + r * PATCH_PITCH + rd_col * 4), v1 = *(const f32x4*)(patch + r * PATCH_PITCH + (rd_col + hi) * 4);
        AuxPre pre;
        if (F32) {
            const char* row = res_lds + (r >> 2) * 1024 + (r & 3) * 256;
            const int x = (r & 1) << 3;
            pre.lo = *(const u32x4*)(row + ((j ^ x) << 4));
            pre.hi = *(const u32x4*)(row + (((8 + j) ^ x) << 4));
        } else {
            pre.lo = *(const u32x4*)(res_lds + it * 1024 + lane * 16);
        }
        const float a8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        epilogue_row8<OF_EPI_GATE_RESID>(p, a8, m_base + r, n_base + rd_col, gv, sc, dot, &pre, hi);
    }
    of_wave_sync();
}
// as epilogue_group_rows with the aux tile in LDS (aux_lds, landed); row passes in a ROLLED loop (one copy of the math)
template <int EPI>
OF_DEV void epilogue_group_rows_auxlds(const OfGemmArgs& p, char* patch, const char* aux_lds, int m_base, int n_base, int lane, float gv,
                                       float sc, float& dot) {
    static_assert(EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT, "bf16 aux epilogues only");
    const int rd_row = lane >> 3, rd_col = (lane & 7) * 8;
    of_wave_sync();
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + rd_row;
        const f32x4 v0 = *(const f32x4*)(patch + r * PATCH_PITCH + rd_col * 4), v1 = *(const f32x4*)(patch + r * PATCH_PITCH + rd_col * 4 + 16);
        AuxPre pre;
        pre.lo = *(const u32x4*)(aux_lds + it * 1024 + lane * 16);
        const float a8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        epilogue_row8<EPI>(p, a8, m_base + r, n_base + rd_col, gv, sc, dot, &pre);
    }
    of_wave_sync();
}
template <int EPI>
OF_DEV void epilogue_group_auxlds(const OfGemmArgs& p, const f32x16& a0, const f32x16& a1, char* patch, const char* aux_lds, int m_base,
                                  int n_base, int lane, float gv, float sc, float& dot) {
    patch_write32(patch, a0, a1, lane);
    epilogue_group_rows_auxlds<EPI>(p, patch, aux_lds, m_base, n_base, lane, gv, sc, dot);
}

// Gate-gradient reduction of the *_DOT epilogues, deterministic: every workgroup writes ONE partial sum (wave sums meet in
// LDS first, fixed order) to its own slot of the caller's workspace; of_dot_finish_kernel (one workgroup, launched behind the
// GEMM by dot_finish below) adds the slots in a fixed order and accumulates (1 - tanh(gate)^2) * total into *dot_out.
// No floating-point atomics: the same operands give the same gate gradient on every run and every rank.
// `red` = nwaves floats of LDS nobody else touches any more; every wave must call.
template <int EPI>
OF_DEV void epilogue_finish(const OfGemmArgs& p, float dot, int lane, int wave, int nwaves, float* red, int slot) {
    if (EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) {
        if (p.dot_out) {
            dot = of_wave_sum(dot);
            if (lane == 0) red[wave] = dot;
            of_sync();
            if (wave == 0 && lane == 0) {
                float s = 0.f;
                for (int w = 0; w < nwaves; ++w) s += red[w];
                ((float*)p.workspace)[slot] = s;
            }
        }
    } else if (EPI == OF_EPI_ACC_F32) {      // OfGemmArgs.sumsq_out: the tile's sum of squares, waves added in a fixed order
        if (p.sumsq_out) {
            dot = of_wave_sum(dot);
            if (lane == 0) red[wave] = dot;
            of_sync();
            if (wave == 0 && lane == 0) {
                float s = 0.f;
                for (int w = 0; w < nwaves; ++w) s += red[w];
                p.sumsq_out[slot] = s;
            }
        }
    }
}

}  // namespace ofg

// implemented in gemm.hip: the second launch of a *_DOT GEMM with dot_out (see epilogue_finish); nslots = workgroups of the GEMM
int of_gemm_dot_finish(const OfGemmArgs& a, int nslots, of_stream_t s);
OF_HOSTDEV bool of_gemm_has_dot(const OfGemmArgs& a) {
    return (a.epi == OF_EPI_DGELU_DOT || a.epi == OF_EPI_SCALE_DOT) && a.dot_out;
}
// slots of per-workgroup gate-gradient partials a *_DOT launch with dot_out may use (the finest tiling any kernel picks), and the
// bytes they take at the start of the workspace (a multiple of 256: the stream-K region of the big-tile kernel follows)
OF_HOSTDEV size_t of_gemm_dot_slots(const OfGemmArgs& a) { return (size_t)((a.M + 127) / 128) * ((a.N + 63) / 64); }
OF_HOSTDEV size_t of_gemm_dot_bytes(const OfGemmArgs& a) {
    return of_gemm_has_dot(a) ? ((of_gemm_dot_slots(a) * sizeof(float) + 255) & ~(size_t)255) : 0;
}
// several problems in one grid (gemm_mid.hip: of_gemm_mid_batch_kernel; gemm.hip: of_gemm_batch, of_splitk_reduce_batch_kernel)
constexpr int OF_GEMM_BATCH_MAX = 4;
struct OfGemmBatchArgs {
    int n;
    int wg_end[OF_GEMM_BATCH_MAX];      // cumulative workgroup counts: problem i owns workgroups [wg_end[i-1], wg_end[i])
    OfGemmArgs a[OF_GEMM_BATCH_MAX];
};
int of_gemm_mid_batch_launch(const OfGemmBatchArgs& m, int total_wg, of_stream_t s);
// implemented in gemm_mid.hip (8 waves, 128x128 tile, 4-slot LDS-DMA ring); OF_E_SHAPE when not eligible
int of_gemm_mid_try(const OfGemmArgs& a, of_stream_t s);
bool of_gemm_mid_eligible(const OfGemmArgs& a);     // what of_gemm_mid_try would accept, without launching
// implemented in gemm_w4m.hip: the 4-wave 256x256 kernel on 16x16x32 MFMAs; same eligibility and return convention as of_gemm_w4_try
int of_gemm_w4m_try(const OfGemmArgs& a, of_stream_t s);
bool of_gemm_w4m_eligible(const OfGemmArgs& a);
size_t of_gemm_w4m_sk_bytes(const OfGemmArgs& a, int grid);      // workspace of a stream-K launch over `grid` workgroups (0: none needed)
// implemented in gemm_w4h.hip: 256x128 tile, 4 waves x (128x64), two workgroups per CU (a tile's epilogue under the other's K loop)
int of_gemm_w4h_try(const OfGemmArgs& a, of_stream_t s);
bool of_gemm_w4h_eligible(const OfGemmArgs& a);
// implemented in gemm_w4s.hip: the same tile, ONE persistent workgroup of 8 waves per CU -- 4 MFMA waves + 4 waves that issue the LDS-DMA
// and run the previous tile's epilogue under the K loop
int of_gemm_w4s_try(const OfGemmArgs& a, of_stream_t s);
bool of_gemm_w4s_eligible(const OfGemmArgs& a);
// implemented in gemm_skinny.hip: M <= 16 rows (decode step), HBM-bound weight streaming; OF_E_SHAPE when not eligible
int of_gemm_skinny_try(const OfGemmArgs& a, of_stream_t s);
inline bool of_gemm_is_skinny(const OfGemmArgs& a) {
    return a.safe == 0 && a.M <= 16 && !a.a_trans && !a.b_trans &&
           (a.epi == OF_EPI_STORE_BF16 || a.epi == OF_EPI_GELU || a.epi == OF_EPI_GATE_RESID);
}
// implemented in gemm_pp.hip; returns OF_E_SHAPE when the shape/layout is not eligible (caller falls back)
int of_gemm_pp_try(const OfGemmArgs& a, of_stream_t s);
// implemented in gemm_w4.hip (4 waves x 128x128, register staged); same eligibility and return convention as of_gemm_pp_try
int of_gemm_w4_try(const OfGemmArgs& a, of_stream_t s);
