// LayerNorm forward/backward (eps 1e-5, affine) for the OpenFlamingo hot path, gfx950.
// Replaces aten::native_layer_norm (+backward) under nn.LayerNorm in open_flamingo/src/helpers.py:18,
// 33-34,105,152.  HBM-bound: one wave per row, 16-byte vector accesses, fp32 statistics.  The forward
// writes the bf16 GEMM operand directly (optionally into a strided destination so the Perceiver's
// cat(LN(x), LN(latents)) of helpers.py:53 is never materialised by a copy; optionally after adding a bf16 branch
// output to the stream: of_layernorm_fwd_add); the backward fuses the residual-stream add, the bf16 operand copy
// of the result, and the dw/db column reductions -- as one wave per row, or for wide rows with dw/db one WORKGROUP
// per row (of_ln_bwd_wg_kernel: 16 instead of 64 accumulator registers per lane, twice the waves per SIMD).
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace {

struct LnArgs {
    const void* x; int x_f32; long ldx;
    const float* w; const float* b;
    void* y; int y_f32; long ldy;
    long y_grp_rows, y_grp_stride;   // grp_rows > 0: row r lands at (r / grp_rows) * grp_stride + (r % grp_rows) * ldy
    bf16_t* y2;                      // optional second, contiguous bf16 copy (row stride dim)
    const bf16_t* add; long ldadd;   // optional bf16 branch output added to x before the statistics (x + add is the new
    void* xsum; long ldsum;          //   residual stream, written to xsum in x's dtype): "x = x + f(x); LN(x)" in one pass
    float* stats;
    long rows; int dim;
    // backward
    const void* dy; int dy_f32; long lddy;
    long dy_grp_rows, dy_grp_stride;
    const bf16_t* dy2;               // optional second upstream gradient (bf16, contiguous), added to dy
    const void* resid;
    void* dx; int dx_f32; long lddx;
    bf16_t* dx_bf16;
    float* dw; float* db;
    int rpw;                         // rows per wave (set by the launcher)
    float* partials;                 // optional [grid][2][dim] scratch: per-workgroup dw/db partial sums (no atomics)
};

OF_DEV void load8(const void* base, int is_f32, size_t off, float (&v)[8]) {
    if (is_f32) {
        const f32x4 a = *(const f32x4*)((const float*)base + off);
        const f32x4 b = *(const f32x4*)((const float*)base + off + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        const u32x4 r = *(const u32x4*)((const bf16_t*)base + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
            v[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
        }
    }
}
OF_DEV void store8(void* base, int is_f32, size_t off, const float (&v)[8]) {
    if (is_f32) {
        *(f32x4*)((float*)base + off) = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)((float*)base + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        u32x4 r = {of_pack_bf16(v[0], v[1]), of_pack_bf16(v[2], v[3]), of_pack_bf16(v[4], v[5]), of_pack_bf16(v[6], v[7])};
        *(u32x4*)((bf16_t*)base + off) = r;
    }
}

// Both kernels keep a whole row in registers: one wave per row, lane l owns the 8-column chunks l, l+64, ... (CPL
// chunks per lane, dim <= CPL*512), so x / dy / resid are read from HBM exactly once, with all of a row's loads in
// flight together.  RPW rows per wave; the backward accumulates its dw/db column partials in registers across those
// rows and reduces them once per workgroup (LDS, one add per wave per column) and once per grid (global atomics).
// Forward.  gamma / beta are staged once per workgroup in LDS (2 * dim floats) AFTER the first row's loads have been issued and are
// read back at the point of use: held in registers (rounds 1-3) they cost 16 * CPL registers -- the kernel ran at 2-4 waves per
// SIMD -- and every wave pulled 2 * dim floats through its CU's vector cache ahead of its 1 * dim of x.  With them in LDS a row's
// working set is 8 * CPL + ~16 registers: eight waves per SIMD, i.e. at 8192 x 2048 every row of the tensor has its loads in flight
// at once; with more rows than that a wave walks rpw rows and (PF) has row r + 1's loads out before it reduces and stores row r.
// Cold (input not in the 256-MB infinity cache: the step's case) 8192 x 2048 fp32 -> bf16: see profiles/r04o_ln_in_step_probe.jsonl.
// A row's addresses are a wave-uniform row pointer (SGPRs) + ONE per-lane element offset + a compile-time chunk stride: 64-bit
// per-chunk addresses of five arrays were 40 of the old kernel's registers.
OF_DEV void ln_load8(const void* rowp, int is_f32, unsigned eo, float (&v)[8]) {
    if (is_f32) {
        const f32x4 a = *(const f32x4*)((const float*)rowp + eo);
        const f32x4 b = *(const f32x4*)((const float*)rowp + eo + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        const u32x4 r = *(const u32x4*)((const bf16_t*)rowp + eo);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
            v[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
        }
    }
}
OF_DEV void ln_store8(void* rowp, int is_f32, unsigned eo, const float (&v)[8]) {
    if (is_f32) {
        *(f32x4*)((float*)rowp + eo) = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)((float*)rowp + eo + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        u32x4 r = {of_pack_bf16(v[0], v[1]), of_pack_bf16(v[2], v[3]), of_pack_bf16(v[4], v[5]), of_pack_bf16(v[6], v[7])};
        *(u32x4*)((bf16_t*)rowp + eo) = r;
    }
}
OF_DEV const void* ln_row(const void* base, int is_f32, size_t elem_off) {
    return of_uniform_ptr((const char*)base + elem_off * (is_f32 ? 4 : 2));
}

template <int CPL, bool PF>
OF_GLOBAL void OF_BOUNDS(256, (PF ? (CPL <= 2 ? 6 : CPL <= 4 ? 4 : 2) : (CPL <= 2 ? 8 : CPL <= 4 ? 5 : CPL <= 5 ? 4 : 3))) of_ln_fwd_kernel(LnArgs a) {
    const int rpw = a.rpw;
    float* sw = (float*)of_smem();
    float* sb = sw + a.dim;
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const unsigned lo = (unsigned)lane * 8, dim = (unsigned)a.dim;
    const float inv_dim = 1.0f / (float)a.dim;
    const long row0 = ((long)of_bid_x() * 4 + wave) * rpw;
    float v[CPL][8];
    auto load_row = [&](long row, auto& d) {
        if (row >= a.rows) return;      // wave-uniform
        const void* xr = ln_row(a.x, a.x_f32, (size_t)row * a.ldx);
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            if (lo + j * 512 < dim) ln_load8(xr, a.x_f32, lo + j * 512, d[j]);
        if (a.add) {
            const void* ar = ln_row(a.add, 0, (size_t)row * a.ldadd);
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                if (lo + j * 512 < dim) {
                    float t[8];
                    ln_load8(ar, 0, lo + j * 512, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[j][e] += t[e];
                }
            }
        }
    };
    load_row(row0, v);
    for (unsigned c = tid * 4; c < dim; c += 1024) {
        *(f32x4*)(sw + c) = *(const f32x4*)(a.w + c);
        *(f32x4*)(sb + c) = *(const f32x4*)(a.b + c);
    }
    of_sync();
    for (int rr = 0; rr < rpw; ++rr) {
        const long row = row0 + rr;
        if (row >= a.rows) return;  // wave-uniform; after the workgroup's only barrier
        float nx[PF ? CPL : 1][8];
        if constexpr (PF) {
            if (rr + 1 < rpw) load_row(row + 1, nx);
        }
        if (a.add) {
            void* sr = (void*)ln_row(a.xsum, a.x_f32, (size_t)row * a.ldsum);
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (lo + j * 512 < dim) ln_store8(sr, a.x_f32, lo + j * 512, v[j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            if (lo + j * 512 < dim) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += v[j][e];
            }
        }
        const float mean = of_wave_sum(sum) * inv_dim;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            if (lo + j * 512 < dim) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sq += (v[j][e] - mean) * (v[j][e] - mean);
            }
        }
        const float rstd = of_rsqrt(of_wave_sum(sq) * inv_dim + 1e-5f);
        if (lane == 0 && a.stats) {
            a.stats[row * 2] = mean;
            a.stats[row * 2 + 1] = rstd;
        }
        const size_t yo = a.y_grp_rows > 0 ? (size_t)(row / a.y_grp_rows) * a.y_grp_stride + (size_t)(row % a.y_grp_rows) * a.ldy
                                           : (size_t)row * a.ldy;
        void* yr = (void*)ln_row(a.y, a.y_f32, yo);
        void* y2r = a.y2 ? (void*)ln_row(a.y2, 0, (size_t)row * a.dim) : nullptr;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const unsigned eo = lo + j * 512;
            if (eo < dim) {
                const f32x4 w0 = *(const f32x4*)(sw + eo), w1 = *(const f32x4*)(sw + eo + 4);
                const f32x4 b0 = *(const f32x4*)(sb + eo), b1 = *(const f32x4*)(sb + eo + 4);
                float o[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (v[j][e] - mean) * rstd * w0[e] + b0[e];
                    o[4 + e] = (v[j][4 + e] - mean) * rstd * w1[e] + b1[e];
                }
                ln_store8(yr, a.y_f32, eo, o);
                if (y2r) ln_store8(y2r, 0, eo, o);
            }
        }
        if constexpr (PF) {
            if (rr + 1 < rpw) {
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[j][e] = nx[j][e];
            }
        }
    }
}

// RED = false: no dw/db (a frozen tower's LayerNorm): without the 16 * CPL column accumulators a row's working set fits
// 128 registers, twice the waves per SIMD hide the row's one memory latency.
template <int CPL, bool RED>
OF_GLOBAL void OF_BOUNDS(256, (RED ? (CPL > 5 ? 1 : 2) : (CPL <= 2 ? 4 : (CPL <= 5 ? 3 : 2)))) of_ln_bwd_kernel(LnArgs a) {
    const int rpw = a.rpw;
    float* sw = (float*)of_smem();
    float* sb = sw + a.dim;
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const int nchunk = a.dim >> 3;
    if constexpr (RED) {
        for (int c = tid; c < a.dim; c += 256) {
            sw[c] = 0.f;
            sb[c] = 0.f;
        }
        of_sync();
    }
    const float inv_dim = 1.0f / (float)a.dim;
    float aw[RED ? CPL : 1][8], ab[RED ? CPL : 1][8];
#pragma unroll
    for (int j = 0; j < (RED ? CPL : 1); ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) aw[j][e] = ab[j][e] = 0.f;
    const bool wr = a.dx || a.dx_bf16;
    for (int rr = 0; rr < rpw; ++rr) {
        const long row = ((long)of_bid_x() * 4 + wave) * rpw + rr;
        if (row >= a.rows) break;  // wave-uniform
        const float mean = a.stats[row * 2], rstd = a.stats[row * 2 + 1];
        const size_t xo = (size_t)row * a.ldx;
        const size_t go = a.dy_grp_rows > 0 ? (size_t)(row / a.dy_grp_rows) * a.dy_grp_stride + (size_t)(row % a.dy_grp_rows) * a.lddy
                                            : (size_t)row * a.lddy;
        const size_t dxo = (size_t)row * a.lddx;
        constexpr bool PRE = CPL <= 4;     // wider rows: the 8 extra registers per chunk would spill
        float xh[CPL][8], gv[CPL][8], rv[PRE ? CPL : 1][8];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + j * 64;
            if (c < nchunk) {
                load8(a.x, a.x_f32, xo + c * 8, xh[j]);
                load8(a.dy, a.dy_f32, go + c * 8, gv[j]);
                // the residual gradient is only needed after the two row reductions: issue its loads with the others so
                // that a row pays ONE memory latency, not two (8 waves per CU: latency, not bandwidth, bounds this kernel)
                if (PRE && wr && a.resid) load8(a.resid, a.dx_f32, dxo + c * 8, rv[PRE ? j : 0]);
                if (a.dy2) {
                    float g2[8];
                    load8(a.dy2, 0, (size_t)row * a.dim + c * 8, g2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[j][e] += g2[e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + j * 64;
            if (c < nchunk) {
                const f32x4 w0 = *(const f32x4*)(a.w + c * 8), w1 = *(const f32x4*)(a.w + c * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float wgt = e < 4 ? w0[e] : w1[e - 4];
                    xh[j][e] = (xh[j][e] - mean) * rstd;
                    if constexpr (RED) {
                        aw[j][e] += gv[j][e] * xh[j][e];
                        ab[j][e] += gv[j][e];
                    }
                    gv[j][e] *= wgt;
                    c1 += gv[j][e];
                    c2 += gv[j][e] * xh[j][e];
                }
            }
        }
        c1 = of_wave_sum(c1) * inv_dim;
        c2 = of_wave_sum(c2) * inv_dim;
        if (wr) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int c = lane + j * 64;
                if (c < nchunk) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = rstd * (gv[j][e] - c1 - xh[j][e] * c2);
                    if (a.resid) {
                        if (!PRE) load8(a.resid, a.dx_f32, dxo + c * 8, rv[0]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += rv[PRE ? j : 0][e];
                    }
                    if (a.dx) store8(a.dx, a.dx_f32, dxo + c * 8, o);
                    if (a.dx_bf16) store8(a.dx_bf16, 0, dxo + c * 8, o);
                }
            }
        }
    }
    if constexpr (RED) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = lane + j * 64;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    of_atomic_add(sw + c * 8 + e, aw[j][e]);
                    of_atomic_add(sb + c * 8 + e, ab[j][e]);
                }
            }
        }
        of_sync();
        if (a.partials) {   // combined by of_ln_colsum_kernel in workgroup order (deterministic, no same-address atomics)
            float* pw = a.partials + (size_t)of_bid_x() * 2 * a.dim;
            for (int c = tid; c < a.dim; c += 256) {
                pw[c] = sw[c];
                pw[a.dim + c] = sb[c];
            }
        } else {
            for (int c = tid; c < a.dim; c += 256) {
                of_atomic_add(a.dw + c, sw[c]);
                of_atomic_add(a.db + c, sb[c]);
            }
        }
    }
}

// Backward WITH dw/db for wide rows (dim >= 1536): one WORKGROUP per row instead of one wave.  A lane then owns CPT (1-2)
// 8-column chunks of the row instead of CPL (4-8): the dw/db column accumulators shrink from 16*CPL to 16*CPT registers
// and R rows can be in flight per workgroup at 3 workgroups per CU (the wave-per-row form needs 242 registers at
// dim 2048: 2 waves per SIMD, 87-104 us for 8192 x 2048 where the traffic is worth ~55).  The two row sums cross the four
// waves through LDS (one barrier per R rows, double-buffered slots).  Every column has exactly one owner lane per
// workgroup, so the per-workgroup partial rows are written straight from registers (no LDS atomics).
template <int CPT, int R>
OF_GLOBAL void OF_BOUNDS(256, 3) of_ln_bwd_wg_kernel(LnArgs a) {
    float* red = (float*)of_smem();     // [2][R][2][4]
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const int nchunk = a.dim >> 3;
    const float inv_dim = 1.0f / (float)a.dim;
    const long r_begin = (long)of_bid_x() * a.rpw;
    const long r_end = r_begin + a.rpw < a.rows ? r_begin + a.rpw : a.rows;
    float wv[CPT][8], aw[CPT][8], ab[CPT][8];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = tid + j * 256;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            aw[j][e] = ab[j][e] = 0.f;
            wv[j][e] = c < nchunk ? a.w[c * 8 + e] : 0.f;
        }
    }
    const bool wr = a.dx || a.dx_bf16;
    int buf = 0;
    for (long r0 = r_begin; r0 < r_end; r0 += R) {
        float xh[R][CPT][8], gv[R][CPT][8], rv[R][CPT][8];
        float mean[R], rstd[R], c1[R], c2[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const long row = r0 + rr;
            c1[rr] = c2[rr] = 0.f;
            mean[rr] = rstd[rr] = 0.f;
            if (row < r_end) {   // workgroup-uniform
                mean[rr] = a.stats[row * 2];
                rstd[rr] = a.stats[row * 2 + 1];
                const size_t xo = (size_t)row * a.ldx;
                const size_t go = a.dy_grp_rows > 0 ? (size_t)(row / a.dy_grp_rows) * a.dy_grp_stride + (size_t)(row % a.dy_grp_rows) * a.lddy
                                                    : (size_t)row * a.lddy;
                const size_t dxo = (size_t)row * a.lddx;
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    const int c = tid + j * 256;
                    if (c < nchunk) {
                        load8(a.x, a.x_f32, xo + c * 8, xh[rr][j]);
                        load8(a.dy, a.dy_f32, go + c * 8, gv[rr][j]);
                        if (wr && a.resid) load8(a.resid, a.dx_f32, dxo + c * 8, rv[rr][j]);
                        if (a.dy2) {
                            float g2[8];
                            load8(a.dy2, 0, (size_t)row * a.dim + c * 8, g2);
#pragma unroll
                            for (int e = 0; e < 8; ++e) gv[rr][j][e] += g2[e];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            if (r0 + rr < r_end) {
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    if (tid + j * 256 < nchunk) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            xh[rr][j][e] = (xh[rr][j][e] - mean[rr]) * rstd[rr];
                            aw[j][e] += gv[rr][j][e] * xh[rr][j][e];
                            ab[j][e] += gv[rr][j][e];
                            gv[rr][j][e] *= wv[j][e];
                            c1[rr] += gv[rr][j][e];
                            c2[rr] += gv[rr][j][e] * xh[rr][j][e];
                        }
                    }
                }
            }
            c1[rr] = of_wave_sum(c1[rr]);
            c2[rr] = of_wave_sum(c2[rr]);
            if (lane == 0) {
                red[((buf * R + rr) * 2 + 0) * 4 + wave] = c1[rr];
                red[((buf * R + rr) * 2 + 1) * 4 + wave] = c2[rr];
            }
        }
        of_sync();
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const long row = r0 + rr;
            if (row < r_end && wr) {
                const float* q1 = red + ((buf * R + rr) * 2 + 0) * 4;
                const float* q2 = red + ((buf * R + rr) * 2 + 1) * 4;
                const float s1 = ((q1[0] + q1[1]) + (q1[2] + q1[3])) * inv_dim;
                const float s2 = ((q2[0] + q2[1]) + (q2[2] + q2[3])) * inv_dim;
                const size_t dxo = (size_t)row * a.lddx;
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    const int c = tid + j * 256;
                    if (c < nchunk) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            o[e] = rstd[rr] * (gv[rr][j][e] - s1 - xh[rr][j][e] * s2);
                            if (a.resid) o[e] += rv[rr][j][e];
                        }
                        if (a.dx) store8(a.dx, a.dx_f32, dxo + c * 8, o);
                        if (a.dx_bf16) store8(a.dx_bf16, 0, dxo + c * 8, o);
                    }
                }
            }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = tid + j * 256;
        if (c < nchunk) {
            if (a.partials) {   // combined by of_ln_colsum_kernel in workgroup order (deterministic)
                float* pw = a.partials + (size_t)of_bid_x() * 2 * a.dim + c * 8;
                store8(pw, 1, 0, aw[j]);
                store8(pw + a.dim, 1, 0, ab[j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    of_atomic_add(a.dw + c * 8 + e, aw[j][e]);
                    of_atomic_add(a.db + c * 8 + e, ab[j][e]);
                }
            }
        }
    }
}

// dw[c] += sum_b partials[b][0][c], db[c] += sum_b partials[b][1][c].  grid (columns / 256, COLSUM_SLICES): slice y sums
// its share of the partial rows (coalesced across the 256 columns of the block) and adds once per column.
constexpr int COLSUM_SLICES = 16;
OF_GLOBAL void of_ln_colsum_kernel(LnArgs a) {
    const int i = of_bid_x() * 256 + of_tid();
    if (i >= 2 * a.dim) return;
    const int nb = a.rpw;   // number of partial rows (reuses the field)
    const int per = (nb + COLSUM_SLICES - 1) / COLSUM_SLICES;
    const int b0 = of_bid_y() * per, b1 = b0 + per < nb ? b0 + per : nb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += a.partials[(size_t)(b + 0) * 2 * a.dim + i];
        s1 += a.partials[(size_t)(b + 1) * 2 * a.dim + i];
        s2 += a.partials[(size_t)(b + 2) * 2 * a.dim + i];
        s3 += a.partials[(size_t)(b + 3) * 2 * a.dim + i];
    }
    for (; b < b1; ++b) s0 += a.partials[(size_t)b * 2 * a.dim + i];
    if (b1 > b0) of_atomic_add(i < a.dim ? a.dw + i : a.db + (i - a.dim), (s0 + s1) + (s2 + s3));
}

// rows per wave: enough workgroups to give every CU two (8 waves), few enough that the per-workgroup dw/db flush
// (2*dim global atomics) stays small next to the row traffic
int pick_rpw(long rows, int cap) {
    int rpw = 1;
    while (rpw < cap && rows / (4L * rpw * 2) >= 512) rpw *= 2;
    return rpw;
}

#define OF_LN_DISPATCH(KERNEL, ...)                                                                  \
    if (a.dim <= 512) return of_launch(KERNEL<1 __VA_ARGS__>, grid, 256, smem, s, a);           \
    if (a.dim <= 1024) return of_launch(KERNEL<2 __VA_ARGS__>, grid, 256, smem, s, a);          \
    if (a.dim <= 1536) return of_launch(KERNEL<3 __VA_ARGS__>, grid, 256, smem, s, a);          \
    if (a.dim <= 2048) return of_launch(KERNEL<4 __VA_ARGS__>, grid, 256, smem, s, a);          \
    if (a.dim <= 2560) return of_launch(KERNEL<5 __VA_ARGS__>, grid, 256, smem, s, a);          \
    if (a.dim <= 4096) return of_launch(KERNEL<8 __VA_ARGS__>, grid, 256, smem, s, a);          \
    return OF_E_SHAPE;

int launch_fwd(LnArgs a, of_stream_t s) {
    // eight resident waves per SIMD take 8192 rows at once; beyond that a wave walks rpw rows with the next row's loads in flight
    int rpw = 1;
    while (rpw < 4 && a.rows > 8192L * rpw) rpw *= 2;
    a.rpw = rpw;
    const long rows_per_block = 4L * rpw;
    of_dim3 grid{(unsigned)((a.rows + rows_per_block - 1) / rows_per_block), 1, 1};
    const size_t smem = (size_t)a.dim * 2 * sizeof(float);
    if (rpw > 1) {
        OF_LN_DISPATCH(of_ln_fwd_kernel, , true)
    }
    OF_LN_DISPATCH(of_ln_fwd_kernel, , false)
}
constexpr int WG_ROWS = 16;          // rows per workgroup of of_ln_bwd_wg_kernel
bool use_wg_bwd(long rows, int dim, bool red) { return red && dim >= 1536 && dim <= 4096 && rows >= 4 * WG_ROWS; }
long bwd_grid(long rows, int dim, bool red) {
    if (use_wg_bwd(rows, dim, red)) return (rows + WG_ROWS - 1) / WG_ROWS;
    const long rows_per_block = 4L * pick_rpw(rows, 16);
    return (rows + rows_per_block - 1) / rows_per_block;
}
int launch_bwd_red(const LnArgs& a, of_dim3 grid, size_t smem, of_stream_t s) {
    OF_LN_DISPATCH(of_ln_bwd_kernel, , true)
}
int launch_bwd_nored(const LnArgs& a, of_dim3 grid, size_t smem, of_stream_t s) {
    OF_LN_DISPATCH(of_ln_bwd_kernel, , false)
}
int launch_bwd_main(const LnArgs& a, of_dim3 grid, size_t smem, of_stream_t s) {
    return a.dw ? launch_bwd_red(a, grid, smem, s) : launch_bwd_nored(a, grid, smem, s);
}
int launch_bwd(LnArgs a, float* workspace, size_t workspace_bytes, of_stream_t s) {
    const bool wg = use_wg_bwd(a.rows, a.dim, a.dw != nullptr);
    a.rpw = wg ? WG_ROWS : pick_rpw(a.rows, 16);
    const long nblk = bwd_grid(a.rows, a.dim, a.dw != nullptr);
    of_dim3 grid{(unsigned)nblk, 1, 1};
    const bool use_ws = a.dw && workspace && nblk > 1 && workspace_bytes >= (size_t)nblk * 2 * a.dim * sizeof(float);
    a.partials = use_ws ? workspace : nullptr;
    int rc;
    if (wg) {
        if (a.dim <= 2048) rc = of_launch(of_ln_bwd_wg_kernel<1, 2>, grid, 256, 2 * 2 * 2 * 4 * sizeof(float), s, a);
        else rc = of_launch(of_ln_bwd_wg_kernel<2, 1>, grid, 256, 2 * 1 * 2 * 4 * sizeof(float), s, a);
    } else {
        rc = launch_bwd_main(a, grid, a.dw ? (size_t)a.dim * 2 * sizeof(float) : 0, s);
    }
    if (rc || !use_ws) return rc;
    a.rpw = (int)nblk;
    return of_launch(of_ln_colsum_kernel, of_dim3{(unsigned)((2 * a.dim + 255) / 256), COLSUM_SLICES, 1}, 256, 0, s, a);
}

int check_common(const void* x, long ldx, long rows, int dim) {
    if (!x || rows <= 0 || dim <= 0) return OF_E_ARG;
    if ((dim & 7) || dim > 4096) return OF_E_SHAPE;
    if ((ldx & 7) || ((uintptr_t)x & 15)) return OF_E_ALIGN;
    return 0;
}

}  // namespace

static int ln_fwd_impl(const void* x, int x_f32, long ldx, const float* w, const float* b, void* y, int y_f32, long ldy,
                       long grp_rows, long grp_stride, bf16_t* y2, float* stats, long rows, int dim, void* stream,
                       const bf16_t* add = nullptr, long ldadd = 0, void* xsum = nullptr, long ldsum = 0) {
    int rc = check_common(x, ldx, rows, dim);
    if (rc) return rc;
    if (!w || !b || !y) return OF_E_ARG;
    if ((ldy & 7) || (grp_stride & 7) || ((uintptr_t)y & 15) || ((uintptr_t)y2 & 15)) return OF_E_ALIGN;
    if ((ldadd & 7) || (ldsum & 7) || ((uintptr_t)add & 15) || ((uintptr_t)xsum & 15)) return OF_E_ALIGN;
    LnArgs a{};
    a.add = add; a.ldadd = ldadd; a.xsum = xsum; a.ldsum = ldsum;
    a.x = x; a.x_f32 = x_f32; a.ldx = ldx; a.w = w; a.b = b; a.y = y; a.y_f32 = y_f32; a.ldy = ldy;
    a.y_grp_rows = grp_rows; a.y_grp_stride = grp_stride; a.y2 = y2;
    a.stats = stats; a.rows = rows; a.dim = dim;
    return launch_fwd(a, (of_stream_t)stream);
}

extern "C" int of_layernorm_fwd_out(const void* x, int x_f32, long ldx, const float* w, const float* b, void* y,
                                    int y_f32, long ldy, float* stats, long rows, int dim, void* stream) {
    return ln_fwd_impl(x, x_f32, ldx, w, b, y, y_f32, ldy, 0, 0, nullptr, stats, rows, dim, stream);
}

extern "C" int of_layernorm_fwd_add(const void* x, int x_f32, long ldx, const uint16_t* add, long ldadd, void* xsum,
                                    long ldsum, const float* w, const float* b, void* y, int y_f32, long ldy,
                                    float* stats, long rows, int dim, void* stream) {
    if (!add || !xsum) return OF_E_ARG;
    return ln_fwd_impl(x, x_f32, ldx, w, b, y, y_f32, ldy, 0, 0, nullptr, stats, rows, dim, stream, add, ldadd, xsum, ldsum);
}

extern "C" int of_layernorm_fwd(const void* x, int x_f32, long ldx, const float* w, const float* b, uint16_t* y,
                                long ldy, float* stats, long rows, int dim, void* stream) {
    return ln_fwd_impl(x, x_f32, ldx, w, b, y, 0, ldy, 0, 0, nullptr, stats, rows, dim, stream);
}

extern "C" int of_layernorm_fwd_grouped(const void* x, int x_f32, long ldx, const float* w, const float* b, uint16_t* y,
                                        long ldy, long grp_rows, long grp_stride, uint16_t* y2, float* stats, long rows,
                                        int dim, void* stream) {
    if (grp_rows <= 0) return OF_E_ARG;
    return ln_fwd_impl(x, x_f32, ldx, w, b, y, 0, ldy, grp_rows, grp_stride, y2, stats, rows, dim, stream);
}

extern "C" int of_layernorm_bwd(const void* dy, int dy_f32, long lddy, long dy_grp_rows, long dy_grp_stride,
                                const uint16_t* dy2, const void* x, int x_f32, long ldx, const float* stats,
                                const float* w, const void* resid, void* dx_out, int out_f32, long lddx,
                                uint16_t* dx_bf16, float* dw, float* db, long rows, int dim, float* workspace,
                                size_t workspace_bytes, void* stream) {
    int rc = check_common(x, ldx, rows, dim);
    if (rc) return rc;
    if (!dy || !stats || !w) return OF_E_ARG;
    if ((dw == nullptr) != (db == nullptr)) return OF_E_ARG;
    if ((lddy & 7) || (lddx & 7) || (dy_grp_stride & 7) || ((uintptr_t)dy & 15) || ((uintptr_t)dy2 & 15)) return OF_E_ALIGN;
    LnArgs a{};
    a.x = x; a.x_f32 = x_f32; a.ldx = ldx; a.w = w; a.stats = const_cast<float*>(stats); a.rows = rows; a.dim = dim;
    a.dy = dy; a.dy_f32 = dy_f32; a.lddy = lddy; a.dy_grp_rows = dy_grp_rows; a.dy_grp_stride = dy_grp_stride;
    a.dy2 = dy2; a.resid = resid; a.dx = dx_out; a.dx_f32 = out_f32; a.lddx = lddx;
    a.dx_bf16 = dx_bf16; a.dw = dw; a.db = db;
    return launch_bwd(a, workspace, workspace_bytes, (of_stream_t)stream);
}

extern "C" size_t of_layernorm_bwd_workspace_bytes(long rows, int dim) {
    if (rows <= 0 || dim <= 0) return 0;
    const long nblk = bwd_grid(rows, dim, true);
    return nblk > 1 ? (size_t)nblk * 2 * dim * sizeof(float) : 0;
}
