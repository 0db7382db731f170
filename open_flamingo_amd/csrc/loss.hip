// Causal-LM loss of the train step (gfx950): token-level cross entropy over the LM head's logits, fused.
//
// The reference's Flamingo.forward hands `labels` to the HF language model (open_flamingo/src/flamingo.py:112-121), whose
// loss is  logits.float() -> log_softmax -> nll_loss(mean over labels != -100)  (transformers ForCausalLMLoss): for
// 8192 x 50435 bf16 logits that chain moves ~12 GB through HBM per step in five eager kernels (rocprofv3,
// profiles/r01_v4_bench_kernel_stats.md).  Here:
//   of_ce_fwd : one pass over the logits; per row an online log-sum-exp in fp32; writes lse[row] and loss[row]
//               (0 for ignored rows).  One workgroup per row, 256 lanes stride the row, wave shuffles + LDS combine.
//   of_ce_bwd : dlogits[row][j] = scale * (exp(logit - lse) - [j == label]),  scale = *gscale (a device scalar:
//               upstream gradient / number of valid rows); ignored rows get zeros.  One pass: reads logits, writes
//               dlogits in the logits' dtype.
// Rows start at arbitrary 2-byte alignment (vocab 50435 is odd): a row is split into a scalar head up to the first 16-byte
// boundary, a body of 16-byte vectors (8 bf16 / 4 fp32 per lane) and a scalar tail.
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace {

struct CeArgs {
    const void* logits; long ld; int f32; long rows; int V;
    const long long* labels; long long ignore_index;
    float* lse; float* loss;
    const float* gscale; void* dlogits; long ldd;
};

OF_DEV float ld_elem(const void* base, int f32, long idx) {
    return f32 ? ((const float*)base)[idx] : of_bf16_to_f32(((const bf16_t*)base)[idx]);
}

// (m, s) <- combine with (m2, s2): running max and sum of exp(x - max)
OF_DEV void lse_merge(float& m, float& s, float m2, float s2) {
    const float mm = m > m2 ? m : m2;
    s = s * of_exp(m - mm) + s2 * of_exp(m2 - mm);
    m = mm;
}

// elements of the scalar head so that (row start + head) is 16-byte aligned, and vector width in elements
OF_DEV int row_head(const void* base, int f32, long off, int V, int& vw) {
    const int es = f32 ? 4 : 2;
    vw = 16 / es;
    const unsigned long long addr = (unsigned long long)base + (unsigned long long)off * es;
    int h = (int)(((16 - (addr & 15)) & 15) / es);
    return h < V ? h : V;
}
OF_DEV void ld_vec(const void* base, int f32, long idx, float (&x)[8]) {     // 8 bf16 or 4 fp32 (x[4..7] untouched)
    if (f32) {
        const f32x4 v = *(const f32x4*)((const float*)base + idx);
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
    } else {
        const u32x4 r = *(const u32x4*)((const bf16_t*)base + idx);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
            x[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
        }
    }
}

template <bool F32IN>      // logits dtype as a template parameter: the per-vector loops unroll over a compile-time width
OF_GLOBAL void OF_BOUNDS(256, 2) of_ce_fwd_kernel(CeArgs a) {
    float* red = (float*)of_smem();
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    for (long row = of_bid_x(); row < a.rows; row += of_gdim_x()) {
        const long long lab = a.labels[row];
        const long base = row * a.ld;
        float m = -3.0e38f, s = 0.f;
        int vw;
        const int head = row_head(a.logits, a.f32, base, a.V, vw);
        const int nvec = (a.V - head) / vw;
        if (tid < head) lse_merge(m, s, ld_elem(a.logits, a.f32, base + tid), 1.0f);
        for (int j = head + nvec * vw + tid; j < a.V; j += 256) lse_merge(m, s, ld_elem(a.logits, a.f32, base + j), 1.0f);
        // body: four 16-byte vectors in flight per lane; per element one v_max, then exp(x - mx) as ONE fma + bare v_exp_f32
        // (exp2 of x * log2 e - mx * log2 e)
        constexpr float LOG2E = 1.4426950408889634f;
        constexpr int VW = F32IN ? 4 : 8;
        auto fold = [&](const float (&x)[8], float& mx) OF_INLINE_LAMBDA {
#pragma unroll
            for (int e = 0; e < VW; ++e) mx = of_max(mx, x[e]);
        };
        auto sum_exp = [&](const float (&x)[8], float nb) OF_INLINE_LAMBDA -> float {
            float p = 0.f;
#pragma unroll
            for (int e = 0; e < VW; ++e) p += of_exp2(x[e] * LOG2E + nb);
            return p;
        };
        int v = tid;
        for (; v + 768 < nvec; v += 1024) {
            float x0[8], x1[8], x2[8], x3[8];
            ld_vec(a.logits, F32IN, base + head + (long)v * vw, x0);
            ld_vec(a.logits, F32IN, base + head + (long)(v + 256) * vw, x1);
            ld_vec(a.logits, F32IN, base + head + (long)(v + 512) * vw, x2);
            ld_vec(a.logits, F32IN, base + head + (long)(v + 768) * vw, x3);
            float mx = x0[0];
            fold(x0, mx);
            fold(x1, mx);
            fold(x2, mx);
            fold(x3, mx);
            const float nb = -mx * LOG2E;
            lse_merge(m, s, mx, (sum_exp(x0, nb) + sum_exp(x1, nb)) + (sum_exp(x2, nb) + sum_exp(x3, nb)));
        }
        for (; v < nvec; v += 256) {
            float x[8];
            ld_vec(a.logits, F32IN, base + head + (long)v * vw, x);
            float mx = x[0];
            fold(x, mx);
            lse_merge(m, s, mx, sum_exp(x, -mx * LOG2E));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float m2 = of_shfl_xor(m, o), s2 = of_shfl_xor(s, o);
            lse_merge(m, s, m2, s2);
        }
        if (lane == 0) {
            red[wave * 2] = m;
            red[wave * 2 + 1] = s;
        }
        of_sync();
        if (tid == 0) {
            float M = red[0], S = red[1];
            for (int w = 1; w < 4; ++w) lse_merge(M, S, red[w * 2], red[w * 2 + 1]);
            const float lse = M + of_log(S);
            a.lse[row] = lse;
            // a label outside [0, V) that is not the ignore index is a caller bug (F.cross_entropy raises a device assert):
            // poison the row's loss so that it surfaces as a NaN loss instead of silently lowering the mean
            const bool in_range = lab >= 0 && lab < a.V;
            a.loss[row] = lab == a.ignore_index ? 0.f : (in_range ? lse - ld_elem(a.logits, a.f32, base + lab) : __builtin_nanf(""));
        }
        of_sync();
    }
}

template <bool F32IN>
OF_GLOBAL void OF_BOUNDS(256, 2) of_ce_bwd_kernel(CeArgs a) {
    const int tid = of_tid();
    const float g = *a.gscale;
    for (long row = of_bid_x(); row < a.rows; row += of_gdim_x()) {
        const long long lab = a.labels[row];
        const bool valid = lab != a.ignore_index && lab >= 0 && lab < a.V;
        const float lse = a.lse[row];
        const long base = row * a.ld, obase = row * a.ldd;
        auto one = [&](int j) {
            const float d = valid ? g * (of_exp(ld_elem(a.logits, a.f32, base + j) - lse) - (j == lab ? 1.0f : 0.0f)) : 0.f;
            if (a.f32) ((float*)a.dlogits)[obase + j] = d;
            else ((bf16_t*)a.dlogits)[obase + j] = of_f32_to_bf16(d);
        };
        int vw;
        const int head = row_head(a.logits, a.f32, base, a.V, vw);
        int vw2;
        // vector stores need the OUTPUT row aligned the same way (it is: dlogits has the logits' shape and a 16-byte
        // aligned base); otherwise everything goes through the scalar path
        const bool vec_ok = row_head(a.dlogits, a.f32, obase, a.V, vw2) == head;
        const int nvec = vec_ok ? (a.V - head) / vw : 0;
        const int body_end = head + nvec * vw;
        if (tid < head) one(tid);
        for (int j = body_end + tid; j < a.V; j += 256) one(j);
        // body: d = g * exp(x - lse) as one fma + bare v_exp_f32 + one multiply per element, two vectors in flight per lane; the
        // "- [j == label]" term is applied by the lane whose vector holds the label (a wave-divergent branch taken once per row)
        constexpr float LOG2E = 1.4426950408889634f;
        const float nb = -lse * LOG2E, gz = valid ? g : 0.f;
        const int lab32 = valid ? (int)lab : -1;
        auto body = [&](int v) OF_INLINE_LAMBDA {
            constexpr int VW = F32IN ? 4 : 8;
            const int j0 = head + v * VW;
            float x[8], d[8];
            ld_vec(a.logits, F32IN, base + j0, x);
#pragma unroll
            for (int e = 0; e < VW; ++e) d[e] = gz * of_exp2(x[e] * LOG2E + nb);
            if ((unsigned)(lab32 - j0) < (unsigned)VW) {
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    if (j0 + e == lab32) d[e] -= gz;
            }
            if (F32IN) *(f32x4*)((float*)a.dlogits + obase + j0) = f32x4{d[0], d[1], d[2], d[3]};
            else *(u32x4*)((bf16_t*)a.dlogits + obase + j0) =
                u32x4{of_pack_bf16(d[0], d[1]), of_pack_bf16(d[2], d[3]), of_pack_bf16(d[4], d[5]), of_pack_bf16(d[6], d[7])};
        };
        int v = tid;
        for (; v + 256 < nvec; v += 512) {
            body(v);
            body(v + 256);
        }
        for (; v < nvec; v += 256) body(v);
    }
}

int grid_for(long rows) { return (int)(rows < 16384 ? rows : 16384); }
}  // namespace

extern "C" int of_ce_fwd(const void* logits, int logits_f32, long ld, const long long* labels, long long ignore_index,
                         long rows, int vocab, float* lse, float* loss_rows, void* stream) {
    if (!logits || !labels || !lse || !loss_rows || rows <= 0 || vocab <= 0 || ld < vocab) return OF_E_ARG;
    CeArgs a{};
    a.logits = logits; a.ld = ld; a.f32 = logits_f32; a.rows = rows; a.V = vocab;
    a.labels = labels; a.ignore_index = ignore_index; a.lse = lse; a.loss = loss_rows;
    if (logits_f32) return of_launch(of_ce_fwd_kernel<true>, of_dim3{(unsigned)grid_for(rows), 1, 1}, 256, 64, (of_stream_t)stream, a);
    return of_launch(of_ce_fwd_kernel<false>, of_dim3{(unsigned)grid_for(rows), 1, 1}, 256, 64, (of_stream_t)stream, a);
}

extern "C" int of_ce_bwd(const void* logits, int logits_f32, long ld, const long long* labels, long long ignore_index,
                         long rows, int vocab, const float* lse, const float* gscale, void* dlogits, long ldd, void* stream) {
    if (!logits || !labels || !lse || !gscale || !dlogits || rows <= 0 || vocab <= 0 || ld < vocab || ldd < vocab) return OF_E_ARG;
    CeArgs a{};
    a.logits = logits; a.ld = ld; a.f32 = logits_f32; a.rows = rows; a.V = vocab;
    a.labels = labels; a.ignore_index = ignore_index; a.lse = const_cast<float*>(lse);
    a.gscale = gscale; a.dlogits = dlogits; a.ldd = ldd;
    if (logits_f32) return of_launch(of_ce_bwd_kernel<true>, of_dim3{(unsigned)grid_for(rows), 1, 1}, 256, 0, (of_stream_t)stream, a);
    return of_launch(of_ce_bwd_kernel<false>, of_dim3{(unsigned)grid_for(rows), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
