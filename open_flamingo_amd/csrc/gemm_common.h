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
OF_DEV AuxPre epilogue_aux_load(const OfGemmArgs& p, int m, int n) {
    AuxPre r{};
    if (EPI == OF_EPI_GATE_RESID) {
        const size_t aoff = (size_t)m * p.ldaux + n;
        if (p.io_f32) {
            r.lo = *(const u32x4*)((const float*)p.aux + aoff);
            r.hi = *(const u32x4*)((const float*)p.aux + aoff + 4);
        } else {
            r.lo = *(const u32x4*)((const bf16_t*)p.aux + aoff);
        }
    } else if (EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) {
        r.lo = *(const u32x4*)((const bf16_t*)p.aux + (size_t)m * p.ldaux + n);
    }
    return r;
}
template <int EPI>
OF_DEV void epilogue_row8(const OfGemmArgs& p, const float (&a)[8], int m, int n, float gv, float sc, float& dot,
                          const AuxPre* pre = nullptr) {
    const size_t off = (size_t)m * p.ldc + n;
    float o[8];
    if (EPI == OF_EPI_STORE_BF16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sc * a[e];
        *(u32x4*)((bf16_t*)p.C + off) = pack8(o);
    } else if (EPI == OF_EPI_GELU) {
        if (p.C2) *(u32x4*)((bf16_t*)p.C2 + off) = pack8(a);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = of_gelu(a[e]);
        *(u32x4*)((bf16_t*)p.C + off) = pack8(o);
    } else if (EPI == OF_EPI_GATE_RESID) {
        const size_t aoff = (size_t)m * p.ldaux + n;
        if (p.io_f32) {
            const f32x4 r0 = pre ? __builtin_bit_cast(f32x4, pre->lo) : *(const f32x4*)((const float*)p.aux + aoff);
            const f32x4 r1 = pre ? __builtin_bit_cast(f32x4, pre->hi) : *(const f32x4*)((const float*)p.aux + aoff + 4);
            *(f32x4*)((float*)p.C + off) = f32x4{r0[0] + sc * a[0], r0[1] + sc * a[1], r0[2] + sc * a[2], r0[3] + sc * a[3]};
            *(f32x4*)((float*)p.C + off + 4) = f32x4{r1[0] + sc * a[4], r1[1] + sc * a[5], r1[2] + sc * a[6], r1[3] + sc * a[7]};
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
#pragma unroll
        for (int e = 0; e < 8; ++e) {
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
        *(u32x4*)((bf16_t*)p.C + off) = pack8(o);
    } else {  // OF_EPI_ACC_F32
        float* c = (float*)p.C + off;
        f32x4 o0 = {sc * a[0], sc * a[1], sc * a[2], sc * a[3]}, o1 = {sc * a[4], sc * a[5], sc * a[6], sc * a[7]};
        if (p.beta != 0.f) {
            const f32x4 c0 = *(const f32x4*)c, c1 = *(const f32x4*)(c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] += p.beta * c0[e];
                o1[e] += p.beta * c1[e];
            }
        }
        *(f32x4*)c = o0;
        *(f32x4*)(c + 4) = o1;
    }
}

// Gate-gradient reduction of the *_DOT epilogues: one fp32 atomic per WORKGROUP (wave sums meet in LDS first).  All
// launches of a step add into the same scalar, and same-address device atomics serialise at ~12 ns each: one per wave
// cost 0.1 ms on a 1024-tile GEMM.  `red` = nwaves floats of LDS nobody else touches any more; every wave must call.
template <int EPI>
OF_DEV void epilogue_finish(const OfGemmArgs& p, float gv, float dot, int lane, int wave, int nwaves, float* red) {
    if (EPI == OF_EPI_DGELU_DOT || EPI == OF_EPI_SCALE_DOT) {
        if (p.dot_out) {
            dot = of_wave_sum(dot);
            if (lane == 0) red[wave] = dot;
            of_sync();
            if (wave == 0 && lane == 0) {
                float s = 0.f;
                for (int w = 0; w < nwaves; ++w) s += red[w];
                of_atomic_add(p.dot_out, (1.0f - gv * gv) * s);
            }
        }
    }
}

}  // namespace ofg

// implemented in gemm_skinny.hip: M <= 16 rows (decode step), HBM-bound weight streaming; OF_E_SHAPE when not eligible
int of_gemm_skinny_try(const OfGemmArgs& a, of_stream_t s);
inline bool of_gemm_is_skinny(const OfGemmArgs& a) {
    return a.safe == 0 && a.M <= 16 && !a.a_trans && !a.b_trans &&
           (a.epi == OF_EPI_STORE_BF16 || a.epi == OF_EPI_GELU || a.epi == OF_EPI_GATE_RESID);
}
// implemented in gemm_pp.hip; returns OF_E_SHAPE when the shape/layout is not eligible (caller falls back)
int of_gemm_pp_try(const OfGemmArgs& a, of_stream_t s);
#ifdef OF_TOOLS_BUILD
int of_gemm_pp_ablate(const OfGemmArgs& a, int mask, of_stream_t s);
int of_gemm_w4_ablate(const OfGemmArgs& a, int mask, of_stream_t s);
#endif
// implemented in gemm_w4.hip (4 waves x 128x128, register staged); same eligibility and return convention as of_gemm_pp_try
int of_gemm_w4_try(const OfGemmArgs& a, of_stream_t s);
