// LayerNorm forward/backward (eps 1e-5, affine) for the OpenFlamingo hot path, gfx950.
// Replaces aten::native_layer_norm (+backward) under nn.LayerNorm in open_flamingo/src/helpers.py:18,
// 33-34,105,152.  HBM-bound: one wave per row, 16-byte vector accesses, fp32 statistics.  The forward
// writes the bf16 GEMM operand directly (optionally into a strided destination so the Perceiver's
// cat(LN(x), LN(latents)) of helpers.py:53 is never materialised by a copy); the backward fuses the
// residual-stream add, the bf16 operand copy of the result, and the dw/db column reductions.
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace {

struct LnArgs {
    const void* x; int x_f32; long ldx;
    const float* w; const float* b;
    void* y; int y_f32; long ldy;
    long y_grp_rows, y_grp_stride;   // grp_rows > 0: row r lands at (r / grp_rows) * grp_stride + (r % grp_rows) * ldy
    bf16_t* y2;                      // optional second, contiguous bf16 copy (row stride dim)
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

constexpr int ROWS_PER_WAVE_FWD = 2;

OF_GLOBAL void of_ln_fwd_kernel(LnArgs a) {
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const int nchunk = a.dim >> 3;
    for (int rr = 0; rr < ROWS_PER_WAVE_FWD; ++rr) {
        const long row = ((long)of_bid_x() * 4 + wave) * ROWS_PER_WAVE_FWD + rr;
        if (row >= a.rows) return;  // wave-uniform
        const size_t xo = (size_t)row * a.ldx;
        float sum = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float v[8];
            load8(a.x, a.x_f32, xo + c * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[e];
        }
        const float mean = of_wave_sum(sum) / (float)a.dim;
        float sq = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float v[8];
            load8(a.x, a.x_f32, xo + c * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) sq += (v[e] - mean) * (v[e] - mean);
        }
        const float rstd = of_rsqrt(of_wave_sum(sq) / (float)a.dim + 1e-5f);
        if (lane == 0 && a.stats) {
            a.stats[row * 2] = mean;
            a.stats[row * 2 + 1] = rstd;
        }
        const size_t yo = a.y_grp_rows > 0 ? (size_t)(row / a.y_grp_rows) * a.y_grp_stride + (size_t)(row % a.y_grp_rows) * a.ldy
                                           : (size_t)row * a.ldy;
        for (int c = lane; c < nchunk; c += 64) {
            float v[8], o[8];
            load8(a.x, a.x_f32, xo + c * 8, v);
            const f32x4 w0 = *(const f32x4*)(a.w + c * 8), w1 = *(const f32x4*)(a.w + c * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(a.b + c * 8), b1 = *(const f32x4*)(a.b + c * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[e] - mean) * rstd * w0[e] + b0[e];
                o[4 + e] = (v[4 + e] - mean) * rstd * w1[e] + b1[e];
            }
            store8(a.y, a.y_f32, yo + c * 8, o);
            if (a.y2) store8(a.y2, 0, (size_t)row * a.dim + c * 8, o);
        }
    }
}

constexpr int ROWS_PER_WAVE_BWD = 8;

OF_GLOBAL void of_ln_bwd_kernel(LnArgs a) {
    float* sw = (float*)of_smem();
    float* sb = sw + a.dim;
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const int nchunk = a.dim >> 3;
    const bool red = a.dw != nullptr;
    if (red) {
        for (int c = tid; c < a.dim; c += 256) {
            sw[c] = 0.f;
            sb[c] = 0.f;
        }
        of_sync();
    }
    const float inv_dim = 1.0f / (float)a.dim;
    for (int rr = 0; rr < ROWS_PER_WAVE_BWD; ++rr) {
        const long row = ((long)of_bid_x() * 4 + wave) * ROWS_PER_WAVE_BWD + rr;
        if (row >= a.rows) break;  // wave-uniform
        const float mean = a.stats[row * 2], rstd = a.stats[row * 2 + 1];
        const size_t xo = (size_t)row * a.ldx;
        const size_t go = a.dy_grp_rows > 0 ? (size_t)(row / a.dy_grp_rows) * a.dy_grp_stride + (size_t)(row % a.dy_grp_rows) * a.lddy
                                            : (size_t)row * a.lddy;
        float c1 = 0.f, c2 = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float xv[8], gv[8];
            load8(a.x, a.x_f32, xo + c * 8, xv);
            load8(a.dy, a.dy_f32, go + c * 8, gv);
            if (a.dy2) {
                float g2[8];
                load8(a.dy2, 0, (size_t)row * a.dim + c * 8, g2);
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[e] += g2[e];
            }
            const f32x4 w0 = *(const f32x4*)(a.w + c * 8), w1 = *(const f32x4*)(a.w + c * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wv = e < 4 ? w0[e] : w1[e - 4];
                const float xh = (xv[e] - mean) * rstd;
                c1 += gv[e] * wv;
                c2 += gv[e] * wv * xh;
            }
        }
        c1 = of_wave_sum(c1) * inv_dim;
        c2 = of_wave_sum(c2) * inv_dim;
        const size_t dxo = (size_t)row * a.lddx;
        for (int c = lane; c < nchunk; c += 64) {
            float xv[8], gv[8], o[8];
            load8(a.x, a.x_f32, xo + c * 8, xv);
            load8(a.dy, a.dy_f32, go + c * 8, gv);
            if (a.dy2) {
                float g2[8];
                load8(a.dy2, 0, (size_t)row * a.dim + c * 8, g2);
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[e] += g2[e];
            }
            const f32x4 w0 = *(const f32x4*)(a.w + c * 8), w1 = *(const f32x4*)(a.w + c * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wv = e < 4 ? w0[e] : w1[e - 4];
                const float xh = (xv[e] - mean) * rstd;
                o[e] = rstd * (gv[e] * wv - c1 - xh * c2);
                if (red) {
                    of_atomic_add(sw + c * 8 + e, gv[e] * xh);
                    of_atomic_add(sb + c * 8 + e, gv[e]);
                }
            }
            if (a.dx || a.dx_bf16) {
                if (a.resid) {
                    float rv[8];
                    load8(a.resid, a.dx_f32, dxo + c * 8, rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += rv[e];
                }
                if (a.dx) store8(a.dx, a.dx_f32, dxo + c * 8, o);
                if (a.dx_bf16) store8(a.dx_bf16, 0, dxo + c * 8, o);
            }
        }
    }
    if (red) {
        of_sync();
        for (int c = tid; c < a.dim; c += 256) {
            of_atomic_add(a.dw + c, sw[c]);
            of_atomic_add(a.db + c, sb[c]);
        }
    }
}

int check_common(const void* x, long ldx, long rows, int dim) {
    if (!x || rows <= 0 || dim <= 0) return OF_E_ARG;
    if ((dim & 7) || dim > 8192) return OF_E_SHAPE;
    if ((ldx & 7) || ((uintptr_t)x & 15)) return OF_E_ALIGN;
    return 0;
}

}  // namespace

static int ln_fwd_impl(const void* x, int x_f32, long ldx, const float* w, const float* b, void* y, int y_f32, long ldy,
                       long grp_rows, long grp_stride, bf16_t* y2, float* stats, long rows, int dim, void* stream) {
    int rc = check_common(x, ldx, rows, dim);
    if (rc) return rc;
    if (!w || !b || !y) return OF_E_ARG;
    if ((ldy & 7) || (grp_stride & 7) || ((uintptr_t)y & 15) || ((uintptr_t)y2 & 15)) return OF_E_ALIGN;
    LnArgs a{};
    a.x = x; a.x_f32 = x_f32; a.ldx = ldx; a.w = w; a.b = b; a.y = y; a.y_f32 = y_f32; a.ldy = ldy;
    a.y_grp_rows = grp_rows; a.y_grp_stride = grp_stride; a.y2 = y2;
    a.stats = stats; a.rows = rows; a.dim = dim;
    const long rows_per_block = 4 * ROWS_PER_WAVE_FWD;
    of_dim3 grid{(unsigned)((rows + rows_per_block - 1) / rows_per_block), 1, 1};
    return of_launch(of_ln_fwd_kernel, grid, 256, 0, (of_stream_t)stream, a);
}

extern "C" int of_layernorm_fwd_out(const void* x, int x_f32, long ldx, const float* w, const float* b, void* y,
                                    int y_f32, long ldy, float* stats, long rows, int dim, void* stream) {
    return ln_fwd_impl(x, x_f32, ldx, w, b, y, y_f32, ldy, 0, 0, nullptr, stats, rows, dim, stream);
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
                                uint16_t* dx_bf16, float* dw, float* db, long rows, int dim, void* stream) {
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
    const long rows_per_block = 4 * ROWS_PER_WAVE_BWD;
    of_dim3 grid{(unsigned)((rows + rows_per_block - 1) / rows_per_block), 1, 1};
    const size_t smem = dw ? (size_t)dim * 2 * sizeof(float) : 0;
    return of_launch(of_ln_bwd_kernel, grid, 256, smem, (of_stream_t)stream, a);
}
