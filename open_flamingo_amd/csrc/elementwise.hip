// Small HBM-bound helpers of the OpenFlamingo hot path (gfx950): dtype casts of GEMM operands, the
// latent broadcast "repeat(latents, 'n d -> b T n d')" (open_flamingo/src/helpers.py:128) and its
// gradient (sum over b,T), and the residual-stream add used when gradients of several consumers meet.
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace {

struct EwArgs {
    const void* a; const void* b; void* out;
    long n;          // elements (multiple of 8 handled vectorised, tail scalar)
    int f32;
    // row-structured variants
    long rows; int dim; long ld; int src_rows;
};

constexpr int GRID_CAP = 2048;

OF_GLOBAL void of_cast_f2b_kernel(EwArgs a) {
    const long nv = a.n >> 3;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < nv; i += stride) {
        const f32x4 x0 = *(const f32x4*)((const float*)a.a + i * 8), x1 = *(const f32x4*)((const float*)a.a + i * 8 + 4);
        u32x4 r = {of_pack_bf16(x0[0], x0[1]), of_pack_bf16(x0[2], x0[3]), of_pack_bf16(x1[0], x1[1]), of_pack_bf16(x1[2], x1[3])};
        *(u32x4*)((bf16_t*)a.out + i * 8) = r;
    }
    if (of_bid_x() == 0) {
        for (long i = (nv << 3) + of_tid(); i < a.n; i += 256) ((bf16_t*)a.out)[i] = of_f32_to_bf16(((const float*)a.a)[i]);
    }
}
OF_GLOBAL void of_cast_b2f_kernel(EwArgs a) {
    const long nv = a.n >> 3;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < nv; i += stride) {
        const u32x4 r = *(const u32x4*)((const bf16_t*)a.a + i * 8);
        f32x4 o0, o1;
        o0[0] = of_bf16_to_f32((bf16_t)(r[0] & 0xffff)); o0[1] = of_bf16_to_f32((bf16_t)(r[0] >> 16));
        o0[2] = of_bf16_to_f32((bf16_t)(r[1] & 0xffff)); o0[3] = of_bf16_to_f32((bf16_t)(r[1] >> 16));
        o1[0] = of_bf16_to_f32((bf16_t)(r[2] & 0xffff)); o1[1] = of_bf16_to_f32((bf16_t)(r[2] >> 16));
        o1[2] = of_bf16_to_f32((bf16_t)(r[3] & 0xffff)); o1[3] = of_bf16_to_f32((bf16_t)(r[3] >> 16));
        *(f32x4*)((float*)a.out + i * 8) = o0;
        *(f32x4*)((float*)a.out + i * 8 + 4) = o1;
    }
    if (of_bid_x() == 0) {
        for (long i = (nv << 3) + of_tid(); i < a.n; i += 256) ((float*)a.out)[i] = of_bf16_to_f32(((const bf16_t*)a.a)[i]);
    }
}
// y = x * sigmoid(1.702 x)  ("quick GELU" of CLIP's MLP), bf16 -> bf16, 8 elements per lane
OF_GLOBAL void of_quick_gelu_kernel(EwArgs a) {
    const long nv = a.n >> 3;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < nv; i += stride) {
        const u32x4 r = *(const u32x4*)((const bf16_t*)a.a + i * 8);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
            v[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * of_rcp(1.0f + of_exp(-1.702f * v[e]));
        *(u32x4*)((bf16_t*)a.out + i * 8) = u32x4{of_pack_bf16(v[0], v[1]), of_pack_bf16(v[2], v[3]), of_pack_bf16(v[4], v[5]),
                                                 of_pack_bf16(v[6], v[7])};
    }
    if (of_bid_x() == 0) {
        for (long i = (nv << 3) + of_tid(); i < a.n; i += 256) {
            const float x = of_bf16_to_f32(((const bf16_t*)a.a)[i]);
            ((bf16_t*)a.out)[i] = of_f32_to_bf16(x * of_rcp(1.0f + of_exp(-1.702f * x)));
        }
    }
}
// erf-GELU of a frozen tower's MLP (HF MptMLP: nn.GELU(approximate="none") between up_proj and down_proj), bf16 -> bf16, and
// its backward dx = dy * gelu'(x); MODE 0: y = gelu(a); 1: out = b * gelu'(a) (a = pre-activation, b = upstream gradient);
// 2: out(fp32) = a(fp32) + b(bf16) -- the fp32 residual stream plus a bf16 branch output.  8 elements per lane.
template <int MODE>
OF_GLOBAL void of_ew8_kernel(EwArgs a) {
    const long nv = a.n >> 3;
    const long stride = (long)of_gdim_x() * 256;
    auto unpack = [](const u32x4 r, float (&v)[8]) OF_INLINE_LAMBDA {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
            v[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
        }
    };
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < nv; i += stride) {
        float x[8], y[8];
        if (MODE == 2) {
            const f32x4 x0 = *(const f32x4*)((const float*)a.a + i * 8), x1 = *(const f32x4*)((const float*)a.a + i * 8 + 4);
            unpack(*(const u32x4*)((const bf16_t*)a.b + i * 8), y);
            *(f32x4*)((float*)a.out + i * 8) = f32x4{x0[0] + y[0], x0[1] + y[1], x0[2] + y[2], x0[3] + y[3]};
            *(f32x4*)((float*)a.out + i * 8 + 4) = f32x4{x1[0] + y[4], x1[1] + y[5], x1[2] + y[6], x1[3] + y[7]};
            continue;
        }
        unpack(*(const u32x4*)((const bf16_t*)a.a + i * 8), x);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = of_gelu(x[e]);
        } else {
            unpack(*(const u32x4*)((const bf16_t*)a.b + i * 8), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] *= of_dgelu(x[e]);
        }
        *(u32x4*)((bf16_t*)a.out + i * 8) = u32x4{of_pack_bf16(y[0], y[1]), of_pack_bf16(y[2], y[3]), of_pack_bf16(y[4], y[5]),
                                                 of_pack_bf16(y[6], y[7])};
    }
    if (of_bid_x() == 0) {
        for (long i = (nv << 3) + of_tid(); i < a.n; i += 256) {
            if (MODE == 2) {
                ((float*)a.out)[i] = ((const float*)a.a)[i] + of_bf16_to_f32(((const bf16_t*)a.b)[i]);
            } else {
                const float x = of_bf16_to_f32(((const bf16_t*)a.a)[i]);
                const float y = MODE == 0 ? of_gelu(x) : of_bf16_to_f32(((const bf16_t*)a.b)[i]) * of_dgelu(x);
                ((bf16_t*)a.out)[i] = of_f32_to_bf16(y);
            }
        }
    }
}
OF_GLOBAL void of_add_kernel(EwArgs a) {
    const long stride = (long)of_gdim_x() * 256;
    if (a.f32) {
        const long nv = a.n >> 2;
        for (long i = (long)of_bid_x() * 256 + of_tid(); i < nv; i += stride) {
            const f32x4 x = *(const f32x4*)((const float*)a.a + i * 4), y = *(const f32x4*)((const float*)a.b + i * 4);
            *(f32x4*)((float*)a.out + i * 4) = f32x4{x[0] + y[0], x[1] + y[1], x[2] + y[2], x[3] + y[3]};
        }
        if (of_bid_x() == 0)
            for (long i = (nv << 2) + of_tid(); i < a.n; i += 256) ((float*)a.out)[i] = ((const float*)a.a)[i] + ((const float*)a.b)[i];
    } else {
        for (long i = (long)of_bid_x() * 256 + of_tid(); i < a.n; i += stride)
            ((bf16_t*)a.out)[i] = of_f32_to_bf16(of_bf16_to_f32(((const bf16_t*)a.a)[i]) + of_bf16_to_f32(((const bf16_t*)a.b)[i]));
    }
}
// out[r][c] = src[r % src_rows][c]   (src fp32 parameter, out stream dtype or bf16)
OF_GLOBAL void of_bcast_rows_kernel(EwArgs a) {
    const int cpr = a.dim >> 2;  // 4-element chunks per row
    const long total = a.rows * cpr;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < total; i += stride) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr) * 4;
        const f32x4 v = *(const f32x4*)((const float*)a.a + (size_t)(r % a.src_rows) * a.dim + c);
        if (a.f32) {
            *(f32x4*)((float*)a.out + (size_t)r * a.ld + c) = v;
        } else {
            u32x2 o = {of_pack_bf16(v[0], v[1]), of_pack_bf16(v[2], v[3])};
            *(u32x2*)((bf16_t*)a.out + (size_t)r * a.ld + c) = o;
        }
    }
}
// dst[r][c] += sum_{k} src[k*dst_rows + r][c]
OF_GLOBAL void of_reduce_rows_kernel(EwArgs a) {
    const long total = (long)a.src_rows * a.dim;  // src_rows = dst_rows here
    const long i = (long)of_bid_x() * 256 + of_tid();
    if (i >= total) return;
    const long r = i / a.dim;
    const int c = (int)(i - r * a.dim);
    float s = 0.f;
    for (long k = r; k < a.rows; k += a.src_rows) {
        s += a.f32 ? ((const float*)a.a)[(size_t)k * a.dim + c] : of_bf16_to_f32(((const bf16_t*)a.a)[(size_t)k * a.dim + c]);
    }
    ((float*)a.out)[i] += s;
}

struct EmbArgs {
    const void* x; int f32; void* out;
    const float* e1; long inner1; int outer1;
    const float* e2; long inner2; int outer2;
    long rows; int dim;
    // reduction
    float* dst; long inner; int outer;
};
// out[r][c] = x[r][c] + e1[(r / inner1) % outer1][c] + e2[(r / inner2) % outer2][c]   (each table optional)
OF_GLOBAL void of_add_embs_kernel(EmbArgs a) {
    const long total = a.rows * a.dim;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < total; i += stride) {
        const long r = i / a.dim;
        const int c = (int)(i - r * a.dim);
        float v = a.f32 ? ((const float*)a.x)[i] : of_bf16_to_f32(((const bf16_t*)a.x)[i]);
        if (a.e1) v += a.e1[(size_t)((r / a.inner1) % a.outer1) * a.dim + c];
        if (a.e2) v += a.e2[(size_t)((r / a.inner2) % a.outer2) * a.dim + c];
        if (a.f32) ((float*)a.out)[i] = v;
        else ((bf16_t*)a.out)[i] = of_f32_to_bf16(v);
    }
}
// dst[o][c] += sum over rows r with (r / inner) % outer == o of src[r][c]
OF_GLOBAL void of_reduce_rows_strided_kernel(EmbArgs a) {
    const long total = (long)a.outer * a.dim;
    const long i = (long)of_bid_x() * 256 + of_tid();
    if (i >= total) return;
    const long o = i / a.dim;
    const int c = (int)(i - o * a.dim);
    const long period = a.inner * a.outer;
    float s = 0.f;
    for (long base = o * a.inner; base < a.rows; base += period)
        for (long r = base; r < base + a.inner && r < a.rows; ++r)
            s += a.f32 ? ((const float*)a.x)[(size_t)r * a.dim + c] : of_bf16_to_f32(((const bf16_t*)a.x)[(size_t)r * a.dim + c]);
    a.dst[i] += s;
}

unsigned grid_for(long work_items) {
    long b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > GRID_CAP) b = GRID_CAP;
    return (unsigned)b;
}


// ---- frozen GPT-NeoX blocks (SURVEY.md 8f N1, OF-4B = RedPajama-INCITE-3B: 32 heads x head size 80, rotary embedding)
// HF GPTNeoXAttention lays the fused projection out per head as [q_h | k_h | v_h] (3 * hs columns) and rotates the first
// `rot` columns of q_h and k_h:  out[j] = x[j] cos[j] - x[j + rot/2] sin[j],  out[j + rot/2] = x[j + rot/2] cos[j + rot/2] +
// x[j] sin[j + rot/2]  (apply_rotary_pos_emb / rotate_half; cos = cat(freqs, freqs).cos()).  The attention kernels exist for head
// sizes 64 and 128, so the same pass writes q, k, v as three [rows][heads * pad] matrices with every head zero-padded to `pad`
// columns (80 -> 128: zero key / value columns change neither the scores nor the used output columns).
// INVERSE: the gradient's way back -- padded dq, dk, dv -> d(qkv) in the projection's layout, rotation transposed.
struct RotArgs {
    bf16_t* qkv; long ldqkv;
    const float* cos; const float* sin;      // [L][rot] fp32, position = row % L
    long L;
    bf16_t* q; bf16_t* k; bf16_t* v; long ldo;
    long rows; int heads, hs, rot, pad;
};
template <bool INVERSE>
OF_GLOBAL void of_rotary_neox_kernel(RotArgs a) {
    const int half = a.rot >> 1;
    const long per_row = (long)a.heads * a.pad;
    const long total = a.rows * per_row;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < total; i += stride) {
        const long row = i / per_row;
        const int rem = (int)(i - row * per_row);
        const int h = rem / a.pad, c = rem - h * a.pad;
        const long po = row * a.ldo + (long)h * a.pad + c;
        if (c >= a.hs) {
            if (!INVERSE) a.q[po] = a.k[po] = a.v[po] = 0;
            continue;
        }
        bf16_t* src = a.qkv + row * a.ldqkv + (long)h * 3 * a.hs;
        if (c >= a.rot) {                   // pass-through columns
            if (!INVERSE) { a.q[po] = src[c]; a.k[po] = src[a.hs + c]; a.v[po] = src[2 * a.hs + c]; }
            else { src[c] = a.q[po]; src[a.hs + c] = a.k[po]; src[2 * a.hs + c] = a.v[po]; }
            continue;
        }
        const long pos = row % a.L;
        const int pc = c < half ? c + half : c - half;          // rotate_half partner
        const float cs = a.cos[pos * a.rot + c];
        if (!INVERSE) {
            const float sn = a.sin[pos * a.rot + c] * (c < half ? -1.0f : 1.0f);
            a.q[po] = of_f32_to_bf16(of_bf16_to_f32(src[c]) * cs + of_bf16_to_f32(src[pc]) * sn);
            a.k[po] = of_f32_to_bf16(of_bf16_to_f32(src[a.hs + c]) * cs + of_bf16_to_f32(src[a.hs + pc]) * sn);
            a.v[po] = src[2 * a.hs + c];
        } else {                            // transpose: d x[c] = g[c] cos[c] + g[pc] * (sign of the PARTNER's row) sin[pc]
            const float sn = a.sin[pos * a.rot + pc] * (pc < half ? -1.0f : 1.0f);
            const long pp = po - c + pc;
            src[c] = of_f32_to_bf16(of_bf16_to_f32(a.q[po]) * cs + of_bf16_to_f32(a.q[pp]) * sn);
            src[a.hs + c] = of_f32_to_bf16(of_bf16_to_f32(a.k[po]) * cs + of_bf16_to_f32(a.k[pp]) * sn);
            src[2 * a.hs + c] = a.v[po];
        }
    }
}
// dst[row][h * dhs + c] = c < shs ? src[row][h * shs + c] : 0   for c < dhs: pads (dhs > shs) or trims (dhs < shs) every head
struct RepackArgs {
    const bf16_t* src; long lds; bf16_t* dst; long ldd;
    long rows; int heads, shs, dhs;
};
OF_GLOBAL void of_head_repack_kernel(RepackArgs a) {
    const long per_row = (long)a.heads * a.dhs;
    const long total = a.rows * per_row;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < total; i += stride) {
        const long row = i / per_row;
        const int rem = (int)(i - row * per_row);
        const int h = rem / a.dhs, c = rem - h * a.dhs;
        a.dst[row * a.ldd + rem] = c < a.shs ? a.src[row * a.lds + (long)h * a.shs + c] : (bf16_t)0;
    }
}

// Vector forms (8 bf16 = 16 bytes per lane) for head sizes and half-rotations that are multiples of 8 -- OF-4B: hs 80, rot 80.
OF_DEV void of_unpack8(const u32x4 r, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
        v[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
    }
}
OF_DEV u32x4 of_pack8(const float (&v)[8]) {
    return u32x4{of_pack_bf16(v[0], v[1]), of_pack_bf16(v[2], v[3]), of_pack_bf16(v[4], v[5]), of_pack_bf16(v[6], v[7])};
}
template <bool INVERSE>
OF_GLOBAL void of_rotary_neox_vec_kernel(RotArgs a) {
    const int half = a.rot >> 1, cpr = a.pad >> 3;              // 8-column chunks per padded head
    const long per_row = (long)a.heads * cpr;
    const long total = a.rows * per_row;
    const long stride = (long)of_gdim_x() * 256;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < total; i += stride) {
        const long row = i / per_row;
        const int rem = (int)(i - row * per_row);
        const int h = rem / cpr, c = (rem - h * cpr) << 3;
        const long po = row * a.ldo + (long)h * a.pad + c;
        if (c >= a.hs) {
            if (!INVERSE) {
                *(u32x4*)(a.q + po) = zero;
                *(u32x4*)(a.k + po) = zero;
                *(u32x4*)(a.v + po) = zero;
            }
            continue;
        }
        bf16_t* src = a.qkv + row * a.ldqkv + (long)h * 3 * a.hs;
        if (!INVERSE) *(u32x4*)(a.v + po) = *(const u32x4*)(src + 2 * a.hs + c);
        else *(u32x4*)(src + 2 * a.hs + c) = *(const u32x4*)(a.v + po);
        if (c >= a.rot) {
            if (!INVERSE) {
                *(u32x4*)(a.q + po) = *(const u32x4*)(src + c);
                *(u32x4*)(a.k + po) = *(const u32x4*)(src + a.hs + c);
            } else {
                *(u32x4*)(src + c) = *(const u32x4*)(a.q + po);
                *(u32x4*)(src + a.hs + c) = *(const u32x4*)(a.k + po);
            }
            continue;
        }
        const long pos = row % a.L;
        const int pc = c < half ? c + half : c - half;
        const float* cs = a.cos + pos * a.rot + c;
        // forward: out[c] = x[c] cos[c] + x[pc] * s(c) sin[c];  inverse: dx[c] = g[c] cos[c] + g[pc] * s(pc) sin[pc]
        const float* sn = a.sin + pos * a.rot + (INVERSE ? pc : c);
        const float sg = (INVERSE ? pc : c) < half ? -1.0f : 1.0f;
        float x[8], y[8], o[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16_t* in = INVERSE ? (t == 0 ? a.q : a.k) : src + t * a.hs;
            const long base = INVERSE ? po - c : 0;
            of_unpack8(*(const u32x4*)(in + base + c), x);
            of_unpack8(*(const u32x4*)(in + base + pc), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = x[e] * cs[e] + y[e] * (sg * sn[e]);
            if (!INVERSE) *(u32x4*)((t == 0 ? a.q : a.k) + po) = of_pack8(o);
            else *(u32x4*)(src + t * a.hs + c) = of_pack8(o);
        }
    }
}
OF_GLOBAL void of_head_repack_vec_kernel(RepackArgs a) {
    const int cpr = a.dhs >> 3;
    const long per_row = (long)a.heads * cpr;
    const long total = a.rows * per_row;
    const long stride = (long)of_gdim_x() * 256;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < total; i += stride) {
        const long row = i / per_row;
        const int rem = (int)(i - row * per_row);
        const int h = rem / cpr, c = (rem - h * cpr) << 3;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (c < a.shs) v = *(const u32x4*)(a.src + row * a.lds + (long)h * a.shs + c);
        *(u32x4*)(a.dst + row * a.ldd + (long)h * a.dhs + c) = v;
    }
}
}  // namespace

extern "C" int of_cast_f32_to_bf16(const float* x, uint16_t* y, long n, void* stream) {
    if (!x || !y || n <= 0) return OF_E_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = x; a.out = y; a.n = n;
    return of_launch(of_cast_f2b_kernel, of_dim3{grid_for(n >> 3), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_cast_bf16_to_f32(const uint16_t* x, float* y, long n, void* stream) {
    if (!x || !y || n <= 0) return OF_E_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = x; a.out = y; a.n = n;
    return of_launch(of_cast_b2f_kernel, of_dim3{grid_for(n >> 3), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_add(const void* x, const void* y, void* out, int f32, long n, void* stream) {
    if (!x || !y || !out || n <= 0) return OF_E_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)out & 15)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = x; a.b = y; a.out = out; a.n = n; a.f32 = f32;
    return of_launch(of_add_kernel, of_dim3{grid_for(f32 ? n >> 2 : n), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_broadcast_rows(const float* src, int src_rows, void* y, int y_f32, long ldy, long rows, int dim,
                                 void* stream) {
    if (!src || !y || src_rows <= 0 || rows <= 0 || dim <= 0) return OF_E_ARG;
    if ((dim & 3) || (ldy & 3)) return OF_E_SHAPE;
    if (((uintptr_t)src & 15) || ((uintptr_t)y & 7)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = src; a.out = y; a.f32 = y_f32; a.ld = ldy; a.rows = rows; a.dim = dim; a.src_rows = src_rows;
    return of_launch(of_bcast_rows_kernel, of_dim3{grid_for(rows * (dim >> 2)), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_reduce_rows(const void* src, int src_f32, long rows, int dim, float* dst, int dst_rows, void* stream) {
    if (!src || !dst || rows <= 0 || dim <= 0 || dst_rows <= 0) return OF_E_ARG;
    EwArgs a{};
    a.a = src; a.out = dst; a.f32 = src_f32; a.rows = rows; a.dim = dim; a.src_rows = dst_rows;
    const long total = (long)dst_rows * dim;
    return of_launch(of_reduce_rows_kernel, of_dim3{(unsigned)((total + 255) / 256), 1, 1}, 256, 0, (of_stream_t)stream, a);
}

extern "C" int of_add_embs(const void* x, int x_f32, const float* e1, long inner1, int outer1, const float* e2, long inner2,
                           int outer2, void* out, long rows, int dim, void* stream) {
    if (!x || !out || rows <= 0 || dim <= 0) return OF_E_ARG;
    if ((e1 && (inner1 <= 0 || outer1 <= 0)) || (e2 && (inner2 <= 0 || outer2 <= 0))) return OF_E_ARG;
    EmbArgs a{};
    a.x = x; a.f32 = x_f32; a.out = out; a.e1 = e1; a.inner1 = inner1; a.outer1 = outer1;
    a.e2 = e2; a.inner2 = inner2; a.outer2 = outer2; a.rows = rows; a.dim = dim;
    return of_launch(of_add_embs_kernel, of_dim3{grid_for(rows * dim), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_reduce_rows_strided(const void* src, int src_f32, long rows, int dim, long inner, int outer, float* dst,
                                      void* stream) {
    if (!src || !dst || rows <= 0 || dim <= 0 || inner <= 0 || outer <= 0) return OF_E_ARG;
    EmbArgs a{};
    a.x = src; a.f32 = src_f32; a.rows = rows; a.dim = dim; a.inner = inner; a.outer = outer; a.dst = dst;
    const long total = (long)outer * dim;
    return of_launch(of_reduce_rows_strided_kernel, of_dim3{(unsigned)((total + 255) / 256), 1, 1}, 256, 0,
                     (of_stream_t)stream, a);
}

extern "C" int of_gelu_fwd(const uint16_t* x, uint16_t* y, long n, void* stream) {
    if (!x || !y || n <= 0) return OF_E_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = x; a.out = y; a.n = n;
    return of_launch(of_ew8_kernel<0>, of_dim3{grid_for(n >> 3), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_gelu_bwd(const uint16_t* dy, const uint16_t* x, uint16_t* dx, long n, void* stream) {
    if (!dy || !x || !dx || n <= 0) return OF_E_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = x; a.b = dy; a.out = dx; a.n = n;
    return of_launch(of_ew8_kernel<1>, of_dim3{grid_for(n >> 3), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_add_bf16(const float* x, const uint16_t* y, float* out, long n, void* stream) {
    if (!x || !y || !out || n <= 0) return OF_E_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)out & 15)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = x; a.b = y; a.out = out; a.n = n;
    return of_launch(of_ew8_kernel<2>, of_dim3{grid_for(n >> 3), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_quick_gelu(const uint16_t* x, uint16_t* y, long n, void* stream) {
    if (!x || !y || n <= 0) return OF_E_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return OF_E_ALIGN;
    EwArgs a{};
    a.a = x; a.out = y; a.n = n;
    return of_launch(of_quick_gelu_kernel, of_dim3{grid_for(n >> 3), 1, 1}, 256, 0, (of_stream_t)stream, a);
}

extern "C" int of_rotary_neox(uint16_t* qkv, long ldqkv, const float* cos, const float* sin, long L, uint16_t* q, uint16_t* k,
                              uint16_t* v, long ldo, long rows, int heads, int head_size, int rot_dims, int head_pad, int inverse,
                              void* stream) {
    if (!qkv || !cos || !sin || !q || !k || !v || rows <= 0 || L <= 0 || heads <= 0) return OF_E_ARG;
    if (head_size <= 0 || head_pad < head_size || rot_dims < 0 || rot_dims > head_size || (rot_dims & 1)) return OF_E_SHAPE;
    if (ldqkv < 3L * heads * head_size || ldo < (long)heads * head_pad) return OF_E_SHAPE;
    RotArgs a{qkv, ldqkv, cos, sin, L, q, k, v, ldo, rows, heads, head_size, rot_dims, head_pad};
    const bool vec = !(head_size & 7) && !(head_pad & 7) && !(rot_dims & 15) && !(ldqkv & 7) && !(ldo & 7) &&
                     !(((uintptr_t)qkv | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15);
    if (vec) {
        const of_dim3 grid{grid_for(rows * heads * (head_pad >> 3) / 2 + 1), 1, 1};
        if (inverse) return of_launch(of_rotary_neox_vec_kernel<true>, grid, 256, 0, (of_stream_t)stream, a);
        return of_launch(of_rotary_neox_vec_kernel<false>, grid, 256, 0, (of_stream_t)stream, a);
    }
    const of_dim3 grid{grid_for(rows * heads * head_pad / 4 + 1), 1, 1};
    if (inverse) return of_launch(of_rotary_neox_kernel<true>, grid, 256, 0, (of_stream_t)stream, a);
    return of_launch(of_rotary_neox_kernel<false>, grid, 256, 0, (of_stream_t)stream, a);
}
extern "C" int of_head_repack(const uint16_t* src, long lds, uint16_t* dst, long ldd, long rows, int heads, int src_head_size,
                              int dst_head_size, void* stream) {
    if (!src || !dst || rows <= 0 || heads <= 0 || src_head_size <= 0 || dst_head_size <= 0) return OF_E_ARG;
    if (lds < (long)heads * src_head_size || ldd < (long)heads * dst_head_size) return OF_E_SHAPE;
    RepackArgs a{src, lds, dst, ldd, rows, heads, src_head_size, dst_head_size};
    if (!(src_head_size & 7) && !(dst_head_size & 7) && !(lds & 7) && !(ldd & 7) && !(((uintptr_t)src | (uintptr_t)dst) & 15))
        return of_launch(of_head_repack_vec_kernel, of_dim3{grid_for(rows * heads * (dst_head_size >> 3) / 2 + 1), 1, 1}, 256, 0,
                         (of_stream_t)stream, a);
    return of_launch(of_head_repack_kernel, of_dim3{grid_for(rows * heads * dst_head_size / 4 + 1), 1, 1}, 256, 0, (of_stream_t)stream, a);
}

#if defined(OF_TOOLS_BUILD) && !defined(OF_HOST_EMU)
// tools/libofhip_tools.so only (tools/rehearse_contention.py): `nwg` one-wave workgroups that stay resident for `ticks` of the
// 100-MHz wall clock doing nothing -- a stand-in for the CUs an RCCL collective occupies while the backward's GEMMs run.
namespace {
struct HoldArgs { long long ticks; };
OF_GLOBAL void of_tools_hold_kernel(HoldArgs a) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < a.ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
extern "C" int of_tools_hold_cus(int nwg, long long ticks, void* stream) {
    if (nwg <= 0 || ticks <= 0) return OF_E_ARG;
    return of_launch(of_tools_hold_kernel, of_dim3{(unsigned)nwg, 1, 1}, 64, 0, (of_stream_t)stream, HoldArgs{ticks});
}
// the same with a collective kernel's footprint: 256 threads, 128 registers per lane, `lds_bytes` of LDS per workgroup -- does not
// fit next to a 4-wave GEMM workgroup (448 registers per lane on every SIMD, 128-160 KiB of LDS): the CU is lost to the GEMM
namespace {
OF_GLOBAL void OF_BOUNDS(256, 1) of_tools_hold_heavy_kernel(HoldArgs a) {
    asm volatile("v_mov_b32 v127, 0" ::: "v127");          // forces a 128-register allocation
    extern __shared__ char hold_lds[];
    hold_lds[threadIdx.x] = 0;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < a.ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
extern "C" int of_tools_hold_cus_heavy(int nwg, long long ticks, int lds_bytes, void* stream) {
    if (nwg <= 0 || ticks <= 0 || lds_bytes < 256) return OF_E_ARG;
    return of_launch(of_tools_hold_heavy_kernel, of_dim3{(unsigned)nwg, 1, 1}, 256, (size_t)lds_bytes, (of_stream_t)stream, HoldArgs{ticks});
}
#endif
