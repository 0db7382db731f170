// Step epilogue of the trainable parameters (gfx950, HBM-bound): global-norm gradient clipping + AdamW in two passes
// over flat fp32 buffers, replacing  torch.nn.utils.clip_grad_norm_(params, 1.0); optimizer.step(); zero_grad()
// of open_flamingo/train/train_utils.py:199-216 with the AdamW groups of open_flamingo/train/train.py:392-408.
//   pass 1  of_sumsq_partial: OF_SUMSQ_PARTS per-workgroup partial sums of g^2 per buffer (4 B/element), then ONE
//           of_sumsq_finish:  *acc = the partials of all buffers summed in a fixed order.  No floating-point atomics:
//                             the clip coefficient is bit-identical on every rank and every run, so data-parallel
//                             replicas cannot drift apart by an ulp of the global norm.
//   pass 2  of_adamw_clip:   c = min(1, max_norm / (sqrt(*acc) + 1e-6));  g' = c g
//                            p <- p (1 - lr wd);  m <- b1 m + (1-b1) g';  v <- b2 v + (1-b2) g'^2
//                            p <- p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)           (torch.optim.AdamW, eps outside
//                            the bias-corrected sqrt, decoupled decay first); optionally writes the bf16 operand copy of
//                            p for the next step's GEMMs and zeroes g (16 B read + 12..18 B written per element).
// The clip coefficient is computed on the device from *acc: no host synchronisation anywhere in the step epilogue.
#include "of_platform.h"
#include "../../include/of_hip.h"

namespace {

struct OptArgs {
    float* p; float* g; float* m; float* v; bf16_t* p_bf16;
    long n;
    float* acc;            // sum of squares (device scalar); of_sumsq_partial: the OF_SUMSQ_PARTS partial slots
    float max_norm, lr, beta1, beta2, eps, wd, bc1, bc2, grad_scale;
    int zero_grad;
    int* applied;          // device counter of APPLIED updates (of_step_advance), or NULL: bias correction from the host's step
};

constexpr int OPT_GRID_CAP = 4096;

// One VIRTUAL workgroup of 256 threads per partial slot (OF_SUMSQ_PARTS of them: the slot -> element assignment depends on n alone).
// The classic launch has one physical workgroup per slot; the narrow launch (of_sumsq_partial_w: a few fat workgroups, so that the
// pass occupies only that many CUs and leaves the rest to another stream) walks the slots with 256-thread sub-blocks -- the same
// threads' sums in the same order per slot: bit-identical partials.
OF_GLOBAL void of_sumsq_kernel(OptArgs a) {
    float* red = (float*)of_smem();
    const long nv = a.n >> 2;
    const long stride = (long)OF_SUMSQ_PARTS * 256;
    const int nsub = of_bdim_x() >> 8, sub = of_tid() >> 8, tid = of_tid() & 255;
    for (int base = of_bid_x() * nsub; base < OF_SUMSQ_PARTS; base += of_gdim_x() * nsub) {
        const int part = base + sub;
        float s = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
        if (part < OF_SUMSQ_PARTS) {
            long i = (long)part * 256 + tid;
            for (; i + 7 * stride < nv; i += 8 * stride) {   // eight independent 16-byte loads in flight per lane (the grid is
                f32x4 g[8];                                   // capped at OF_SUMSQ_PARTS workgroups: depth, not width, hides latency)
#pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = *(const f32x4*)(a.g + (i + u * stride) * 4);
#pragma unroll
                for (int u = 0; u < 8; u += 4) {
                    s += g[u][0] * g[u][0] + g[u][1] * g[u][1] + g[u][2] * g[u][2] + g[u][3] * g[u][3];
                    s2 += g[u + 1][0] * g[u + 1][0] + g[u + 1][1] * g[u + 1][1] + g[u + 1][2] * g[u + 1][2] + g[u + 1][3] * g[u + 1][3];
                    s3 += g[u + 2][0] * g[u + 2][0] + g[u + 2][1] * g[u + 2][1] + g[u + 2][2] * g[u + 2][2] + g[u + 2][3] * g[u + 2][3];
                    s4 += g[u + 3][0] * g[u + 3][0] + g[u + 3][1] * g[u + 3][1] + g[u + 3][2] * g[u + 3][2] + g[u + 3][3] * g[u + 3][3];
                }
            }
            for (; i < nv; i += stride) {
                const f32x4 g = *(const f32x4*)(a.g + i * 4);
                s += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
            }
            s += s2 + s3 + s4;
            if (part == 0)
                for (long i = (nv << 2) + tid; i < a.n; i += 256) s += a.g[i] * a.g[i];
        }
        s = of_wave_sum(s);
        if ((tid & 63) == 0) red[sub * 4 + (tid >> 6)] = s;
        of_sync();
        if (tid == 0 && part < OF_SUMSQ_PARTS) a.acc[part] = (red[sub * 4] + red[sub * 4 + 1]) + (red[sub * 4 + 2] + red[sub * 4 + 3]);   // slots without work store 0
        of_sync();          // red is reused by the next round of slots
    }
}

// one workgroup: lane t sums slots t, t+256, ... in index order, then the fixed wave/LDS tree
OF_GLOBAL void of_sumsq_finish_kernel(OptArgs a) {
    float* red = (float*)of_smem();
    float s = 0.f;
    for (long i = of_tid(); i < a.n; i += 256) s += a.g[i];
    s = of_wave_sum(s);
    const int tid = of_tid();
    if ((tid & 63) == 0) red[tid >> 6] = s;
    of_sync();
    if (tid == 0) *a.acc = (red[0] + red[1]) + (red[2] + red[3]);
}

OF_DEV float adamw_one(const OptArgs& a, float coef, float step_size, float inv_sqrt_bc2, float decay, float& p, float g,
                       float& m, float& v) {
    g *= coef;
    p *= decay;
    m = a.beta1 * m + (1.0f - a.beta1) * g;
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
    p -= step_size * (m / denom);
    return p;
}

OF_GLOBAL void of_adamw_kernel(OptArgs a) {
    // gradients arrive as SUMS over ranks: grad_scale = 1/world turns them into the average DDP would have produced;
    // *acc is the squared norm of the unscaled buffers
    const float norm = sqrtf(*a.acc) * a.grad_scale;
    // A non-finite global norm (a NaN / Inf anywhere in the gradients, e.g. from a NaN loss) skips the update on every rank
    // alike -- they all read the same all-reduced norm: the device-side form of the reference's "if torch.isnan(loss): skip
    // the step" (train_utils.py:161-169) without its host synchronisation and without the deadlock a one-rank host decision
    // would cause under a collective.  Parameters, moments and bf16 copies stay untouched; gradients this pass would have
    // cleared are still cleared.
    if (!(norm < 3.0e38f)) {
        if (a.zero_grad) {
            const long nvz = a.n >> 2, strz = (long)of_gdim_x() * of_bdim_x();
            for (long i = (long)of_bid_x() * of_bdim_x() + of_tid(); i < nvz; i += strz) *(f32x4*)(a.g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            if (of_bid_x() == 0)
                for (long i = (nvz << 2) + of_tid(); i < a.n; i += of_bdim_x()) a.g[i] = 0.f;
        }
        return;
    }
    float coef = a.max_norm > 0.f ? a.max_norm / (norm + 1e-6f) : 1.0f;
    coef = (coef < 1.0f ? coef : 1.0f) * a.grad_scale;
    float bc1 = a.bc1, bc2 = a.bc2;
    if (a.applied) {       // Adam's step = updates actually applied (a skipped NaN step does not advance the bias correction)
        const float t = (float)*a.applied;
        bc1 = 1.0f - powf(a.beta1, t);
        bc2 = 1.0f - powf(a.beta2, t);
    }
    const float step_size = a.lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2), decay = 1.0f - a.lr * a.wd;
    const long nv = a.n >> 2;
    const long stride = (long)of_gdim_x() * of_bdim_x();      // element-wise: any grid gives the same results
    auto one = [&](long i, f32x4 p, f32x4 m, f32x4 v, const f32x4 g) OF_INLINE_LAMBDA {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = p[e], me = m[e], ve = v[e];
            adamw_one(a, coef, step_size, inv_sqrt_bc2, decay, pe, g[e], me, ve);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
        *(f32x4*)(a.p + i * 4) = p;
        *(f32x4*)(a.m + i * 4) = m;
        *(f32x4*)(a.v + i * 4) = v;
        if (a.p_bf16) *(u32x2*)(a.p_bf16 + i * 4) = u32x2{of_pack_bf16(p[0], p[1]), of_pack_bf16(p[2], p[3])};
        if (a.zero_grad) *(f32x4*)(a.g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    long i = (long)of_bid_x() * of_bdim_x() + of_tid();
    for (; i + stride < nv; i += 2 * stride) {       // two vectors of each stream per lane: eight 16-byte loads in flight
        const long j = i + stride;
        const f32x4 p0 = *(const f32x4*)(a.p + i * 4), m0 = *(const f32x4*)(a.m + i * 4), v0 = *(const f32x4*)(a.v + i * 4);
        const f32x4 g0 = *(const f32x4*)(a.g + i * 4);
        const f32x4 p1 = *(const f32x4*)(a.p + j * 4), m1 = *(const f32x4*)(a.m + j * 4), v1 = *(const f32x4*)(a.v + j * 4);
        const f32x4 g1 = *(const f32x4*)(a.g + j * 4);
        one(i, p0, m0, v0, g0);
        one(j, p1, m1, v1, g1);
    }
    for (; i < nv; i += stride)
        one(i, *(const f32x4*)(a.p + i * 4), *(const f32x4*)(a.m + i * 4), *(const f32x4*)(a.v + i * 4), *(const f32x4*)(a.g + i * 4));
    if (of_bid_x() == 0) {
        for (long i = (nv << 2) + of_tid(); i < a.n; i += of_bdim_x()) {
            float pe = a.p[i], me = a.m[i], ve = a.v[i];
            adamw_one(a, coef, step_size, inv_sqrt_bc2, decay, pe, a.g[i], me, ve);
            a.p[i] = pe; a.m[i] = me; a.v[i] = ve;
            if (a.p_bf16) a.p_bf16[i] = of_f32_to_bf16(pe);
            if (a.zero_grad) a.g[i] = 0.f;
        }
    }
}

// one thread: the device-side "optimizer step happened" counter
OF_GLOBAL void of_step_advance_kernel(OptArgs a) {
    if (of_tid() == 0 && of_bid_x() == 0 && sqrtf(*a.acc) < 3.0e38f) *a.applied += 1;
}

unsigned opt_grid(long n) {
    long b = ((n >> 2) + 255) / 256;
    if (b < 1) b = 1;
    if (b > OPT_GRID_CAP) b = OPT_GRID_CAP;
    return (unsigned)b;
}

}  // namespace

// Narrow launches (max_workgroups > 0): that many FAT workgroups (1024 threads: one per CU) instead of a grid that covers the chip --
// the pass then holds max_workgroups CUs and leaves the others to whatever runs on another stream (train/optim.py: the next step's
// vision-tower forward).  HBM-bound passes need depth, not width: ~96 CUs with 16 waves of eight 16-byte loads each keep the memory
// pipes as full as 256 do.  Same arithmetic per element / per partial slot: bit-identical results.
#ifdef OF_HOST_EMU
constexpr int OPT_FAT_BLOCK = 512;       // the emulator runs at most 512 fibers per workgroup
#else
constexpr int OPT_FAT_BLOCK = 1024;
#endif

extern "C" int of_sumsq_partial_w(const float* g, long n, float* partials, int max_workgroups, void* stream) {
    if (!g || !partials || n <= 0 || max_workgroups < 0) return OF_E_ARG;
    if ((uintptr_t)g & 15) return OF_E_ALIGN;
    OptArgs a{};
    a.g = const_cast<float*>(g); a.n = n; a.acc = partials;
    if (max_workgroups > 0) {
        constexpr int nsub = OPT_FAT_BLOCK / 256;
        int wg = (OF_SUMSQ_PARTS + nsub - 1) / nsub;
        if (wg > max_workgroups) wg = max_workgroups;
        return of_launch(of_sumsq_kernel, of_dim3{(unsigned)wg, 1, 1}, OPT_FAT_BLOCK, 4 * nsub * sizeof(float), (of_stream_t)stream, a);
    }
    return of_launch(of_sumsq_kernel, of_dim3{OF_SUMSQ_PARTS, 1, 1}, 256, 4 * sizeof(float), (of_stream_t)stream, a);
}

extern "C" int of_sumsq_partial(const float* g, long n, float* partials, void* stream) {
    if (!g || !partials || n <= 0) return OF_E_ARG;
    if ((uintptr_t)g & 15) return OF_E_ALIGN;
    OptArgs a{};
    a.g = const_cast<float*>(g); a.n = n; a.acc = partials;
    // a fixed grid: the slot -> element assignment depends on n alone, never on the device or the launch.  Two
    // workgroups per CU keep the HBM pipes full (round 2: one same-address atomic per workgroup cost 49 us of a 59-us
    // launch at 4096 workgroups, which is how this kernel came to have few, fat workgroups).
    return of_launch(of_sumsq_kernel, of_dim3{OF_SUMSQ_PARTS, 1, 1}, 256, 4 * sizeof(float), (of_stream_t)stream, a);
}

extern "C" int of_sumsq_finish(const float* partials, long count, float* acc, void* stream) {
    if (!partials || !acc || count <= 0) return OF_E_ARG;
    OptArgs a{};
    a.g = const_cast<float*>(partials); a.n = count; a.acc = acc;
    return of_launch(of_sumsq_finish_kernel, of_dim3{1, 1, 1}, 256, 4 * sizeof(float), (of_stream_t)stream, a);
}

extern "C" int of_step_advance(const float* sumsq, int* applied_steps, void* stream) {
    if (!sumsq || !applied_steps) return OF_E_ARG;
    OptArgs a{};
    a.acc = const_cast<float*>(sumsq); a.applied = applied_steps;
    return of_launch(of_step_advance_kernel, of_dim3{1, 1, 1}, 64, 0, (of_stream_t)stream, a);
}

extern "C" int of_adamw_clip_w(float* p, float* g, float* m, float* v, uint16_t* p_bf16, long n, const float* sumsq,
                               float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float grad_scale, int step, int zero_grad, const int* applied_steps, int max_workgroups, void* stream);

extern "C" int of_adamw_clip(float* p, float* g, float* m, float* v, uint16_t* p_bf16, long n, const float* sumsq,
                             float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                             float grad_scale, int step, int zero_grad, const int* applied_steps, void* stream) {
    return of_adamw_clip_w(p, g, m, v, p_bf16, n, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, grad_scale, step, zero_grad,
                           applied_steps, 0, stream);
}

extern "C" int of_adamw_clip_w(float* p, float* g, float* m, float* v, uint16_t* p_bf16, long n, const float* sumsq,
                               float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float grad_scale, int step, int zero_grad, const int* applied_steps, int max_workgroups, void* stream) {
    if (max_workgroups < 0) return OF_E_ARG;
    if (!p || !g || !m || !v || !sumsq || n <= 0 || (step <= 0 && !applied_steps)) return OF_E_ARG;
    if (((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)m & 15) || ((uintptr_t)v & 15) || ((uintptr_t)p_bf16 & 7))
        return OF_E_ALIGN;
    OptArgs a{};
    a.p = p; a.g = g; a.m = m; a.v = v; a.p_bf16 = p_bf16; a.n = n; a.acc = const_cast<float*>(sumsq);
    a.grad_scale = grad_scale; a.max_norm = max_norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = 1.0f - powf(beta1, (float)(step > 0 ? step : 1));
    a.bc2 = 1.0f - powf(beta2, (float)(step > 0 ? step : 1));
    a.zero_grad = zero_grad;
    a.applied = const_cast<int*>(applied_steps);
    if (max_workgroups > 0) {
        long need = ((n >> 2) + OPT_FAT_BLOCK - 1) / OPT_FAT_BLOCK;
        if (need < 1) need = 1;
        const unsigned wg = (unsigned)(need < max_workgroups ? need : max_workgroups);
        return of_launch(of_adamw_kernel, of_dim3{wg, 1, 1}, OPT_FAT_BLOCK, 0, (of_stream_t)stream, a);
    }
    return of_launch(of_adamw_kernel, of_dim3{opt_grid(n), 1, 1}, 256, 0, (of_stream_t)stream, a);
}
