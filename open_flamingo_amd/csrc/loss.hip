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
// Rows start at arbitrary 2-byte alignment (vocab 50435 is odd): scalar 16-bit loads, coalesced across the wave, unrolled.
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

OF_GLOBAL void OF_BOUNDS(256, 2) of_ce_fwd_kernel(CeArgs a) {
    float* red = (float*)of_smem();
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    for (long row = of_bid_x(); row < a.rows; row += of_gdim_x()) {
        const long long lab = a.labels[row];
        const long base = row * a.ld;
        float m = -3.0e38f, s = 0.f;
        constexpr int U = 8;
        int j = tid;
        for (; j + (U - 1) * 256 < a.V; j += U * 256) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld_elem(a.logits, a.f32, base + j + u * 256);
            float mx = x[0];
#pragma unroll
            for (int u = 1; u < U; ++u) mx = x[u] > mx ? x[u] : mx;
            float part = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) part += of_exp(x[u] - mx);
            lse_merge(m, s, mx, part);
        }
        for (; j < a.V; j += 256) lse_merge(m, s, ld_elem(a.logits, a.f32, base + j), 1.0f);
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
        constexpr int U = 8;
        int j = tid;
        for (; j + (U - 1) * 256 < a.V; j += U * 256) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld_elem(a.logits, a.f32, base + j + u * 256);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + u * 256;
                const float d = valid ? g * (of_exp(x[u] - lse) - (jj == lab ? 1.0f : 0.0f)) : 0.f;
                if (a.f32) ((float*)a.dlogits)[obase + jj] = d;
                else ((bf16_t*)a.dlogits)[obase + jj] = of_f32_to_bf16(d);
            }
        }
        for (; j < a.V; j += 256) {
            const float d = valid ? g * (of_exp(ld_elem(a.logits, a.f32, base + j) - lse) - (j == lab ? 1.0f : 0.0f)) : 0.f;
            if (a.f32) ((float*)a.dlogits)[obase + j] = d;
            else ((bf16_t*)a.dlogits)[obase + j] = of_f32_to_bf16(d);
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
