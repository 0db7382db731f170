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
        int v = tid;
        for (; v + 256 < nvec; v += 512) {            // two vectors in flight per lane
            float x[8], y[8];
            ld_vec(a.logits, a.f32, base + head + (long)v * vw, x);
            ld_vec(a.logits, a.f32, base + head + (long)(v + 256) * vw, y);
            float mx = x[0];
            for (int e = 1; e < vw; ++e) mx = x[e] > mx ? x[e] : mx;
            for (int e = 0; e < vw; ++e) mx = y[e] > mx ? y[e] : mx;
            float part = 0.f;
            for (int e = 0; e < vw; ++e) part += of_exp(x[e] - mx) + of_exp(y[e] - mx);
            lse_merge(m, s, mx, part);
        }
        for (; v < nvec; v += 256) {
            float x[8];
            ld_vec(a.logits, a.f32, base + head + (long)v * vw, x);
            float mx = x[0];
            for (int e = 1; e < vw; ++e) mx = x[e] > mx ? x[e] : mx;
            float part = 0.f;
            for (int e = 0; e < vw; ++e) part += of_exp(x[e] - mx);
            lse_merge(m, s, mx, part);
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
            const bool valid = lab != a.ignore_index && lab >= 0 && lab < a.V;
            a.loss[row] = valid ? lse - ld_elem(a.logits, a.f32, base + lab) : 0.f;
        }
        of_sync();
    }
}

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
        for (int v = tid; v < nvec; v += 256) {
            const int j0 = head + v * vw;
            float x[8], d[8];
            ld_vec(a.logits, a.f32, base + j0, x);
            for (int e = 0; e < vw; ++e) d[e] = valid ? g * (of_exp(x[e] - lse) - (j0 + e == lab ? 1.0f : 0.0f)) : 0.f;
            if (a.f32) *(f32x4*)((float*)a.dlogits + obase + j0) = f32x4{d[0], d[1], d[2], d[3]};
            else *(u32x4*)((bf16_t*)a.dlogits + obase + j0) =
                u32x4{of_pack_bf16(d[0], d[1]), of_pack_bf16(d[2], d[3]), of_pack_bf16(d[4], d[5]), of_pack_bf16(d[6], d[7])};
        }
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
    return of_launch(of_ce_fwd_kernel, of_dim3{(unsigned)grid_for(rows), 1, 1}, 256, 64, (of_stream_t)stream, a);
}

extern "C" int of_ce_bwd(const void* logits, int logits_f32, long ld, const long long* labels, long long ignore_index,
                         long rows, int vocab, const float* lse, const float* gscale, void* dlogits, long ldd, void* stream) {
    if (!logits || !labels || !lse || !gscale || !dlogits || rows <= 0 || vocab <= 0 || ld < vocab || ldd < vocab) return OF_E_ARG;
    CeArgs a{};
    a.logits = logits; a.ld = ld; a.f32 = logits_f32; a.rows = rows; a.V = vocab;
    a.labels = labels; a.ignore_index = ignore_index; a.lse = const_cast<float*>(lse);
    a.gscale = gscale; a.dlogits = dlogits; a.ldd = ldd;
    return of_launch(of_ce_bwd_kernel, of_dim3{(unsigned)grid_for(rows), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
