// Skinny GEMM for the decode step: C[M x N] = A[M x K] * B[N x K]^T with M <= 16 rows -- one new
// token per sequence through the nn.Linear layers of a gated cross-attention block (helpers.py:154-156,19-21).
//
// Roofline: HBM.  The weight matrix B (N x K bf16) is streamed exactly once; A (M x K) and C (M x N) are a few KB and
// stay in L2.  Algorithmic bytes = 2*N*K (+ 2*M*K + 2..4*M*N).  The 128x128 MFMA tile kernel runs this shape on
// N/128 workgroups (16 CUs for N = 2048) and takes 70-130 us per launch; here every CU streams.
//
// Work split: a workgroup owns R consecutive weight rows (output columns) for the full K; its four waves take every
// fourth 512-element K chunk, one 16-byte load per lane and row, so a wave-load covers 1 KiB of ONE weight row (whole
// 128-byte lines).  Products are accumulated per lane with v_dot2c_f32_bf16 (2 MACs per instruction, fp32 accumulate:
// at M <= 16 the VALU keeps up with the HBM stream and no LDS staging or MFMA fragment shuffle is needed).  The
// MT*R per-lane partial sums are reduced across the wave by a halving butterfly -- each step exchanges half of the
// live values with the partner lane, so it costs ~MT*R shuffles in total instead of 6*MT*R -- then across the four
// waves through LDS, and thread i < MT*R applies the fused epilogue to output element i.
#include "gemm_common.h"

namespace {

constexpr int SK_WAVES = 4;
constexpr int SK_CHUNK = 512;   // K elements per wave-iteration: 64 lanes x 8 bf16

template <int EPI>
OF_DEV void skinny_store(const OfGemmArgs& p, float v, int m, int n, float sc) {
    const size_t off = (size_t)m * p.ldc + n;
    if (EPI == OF_EPI_STORE_BF16) {
        ((bf16_t*)p.C)[off] = of_f32_to_bf16(sc * v);
    } else if (EPI == OF_EPI_GELU) {
        if (p.C2) ((bf16_t*)p.C2)[off] = of_f32_to_bf16(v);
        ((bf16_t*)p.C)[off] = of_f32_to_bf16(of_gelu(v));
    } else {  // OF_EPI_GATE_RESID
        const size_t aoff = (size_t)m * p.ldaux + n;
        if (p.io_f32)
            ((float*)p.C)[off] = ((const float*)p.aux)[aoff] + sc * v;
        else
            ((bf16_t*)p.C)[off] = of_f32_to_bf16(of_bf16_to_f32(((const bf16_t*)p.aux)[aoff]) + sc * v);
    }
}

template <int MT, int R, int EPI>
OF_GLOBAL void OF_BOUNDS(SK_WAVES * 64, 2) of_gemm_skinny_kernel(const OfGemmArgs p) {
    constexpr int V = MT * R;
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const int n0 = of_bid_x() * R;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;

    for (int k = wave * SK_CHUNK + lane * 8; k < p.K; k += SK_WAVES * SK_CHUNK) {
        u32x4 w[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r < p.N ? n0 + r : p.N - 1;   // clamped rows are computed and dropped
            w[r] = *(const u32x4*)(B + (size_t)n * p.ldb + k);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            u32x4 a = {0u, 0u, 0u, 0u};
            if (m < p.M) a = *(const u32x4*)(A + (size_t)m * p.lda + k);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m * R + r] = of_dot2_bf16(a[e], w[r][e], acc[m * R + r]);
        }
    }

    // ---- wave reduction: after the halving steps lane l holds the total of value index idx(l)
    int live = V, idx = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int bit = (lane >> s) & 1;
        if (live > 1) {
            const int half = live >> 1;
#pragma unroll
            for (int i = 0; i < V / 2; ++i) {
                if (i < half) {
                    const float lo = acc[i], hi = acc[i + half];
                    const float got = of_shfl_xor(bit ? lo : hi, 1 << s);
                    acc[i] = (bit ? hi : lo) + got;
                }
            }
            idx += bit * half;
            live = half;
        } else {
            acc[0] += of_shfl_xor(acc[0], 1 << s);
        }
    }
    float* red = (float*)of_smem();          // [SK_WAVES][V]
    if (lane < V) red[wave * V + idx] = acc[0];
    of_sync();
    if (tid < V) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) v += red[w * V + tid];
        const int m = tid / R, n = n0 + tid % R;
        if (m < p.M && n < p.N) {
            float gv = 1.0f;
            if (p.gate) gv = of_tanh(*p.gate);
            skinny_store<EPI>(p, v, m, n, gv * p.alpha);
        }
    }
}

template <int MT, int R>
int launch_epi(const OfGemmArgs& a, of_stream_t s) {
    const of_dim3 grid{(unsigned)((a.N + R - 1) / R), 1, 1};
    const size_t smem = SK_WAVES * MT * R * sizeof(float);
    switch (a.epi) {
        case OF_EPI_STORE_BF16: return of_launch(of_gemm_skinny_kernel<MT, R, OF_EPI_STORE_BF16>, grid, SK_WAVES * 64, smem, s, a);
        case OF_EPI_GELU: return of_launch(of_gemm_skinny_kernel<MT, R, OF_EPI_GELU>, grid, SK_WAVES * 64, smem, s, a);
        default: return of_launch(of_gemm_skinny_kernel<MT, R, OF_EPI_GATE_RESID>, grid, SK_WAVES * 64, smem, s, a);
    }
}

template <int R>
int launch_rows(const OfGemmArgs& a, of_stream_t s) {
    if (a.M <= 1) return launch_epi<1, R>(a, s);
    if (a.M <= 2) return launch_epi<2, R>(a, s);
    if (a.M <= 4) return launch_epi<4, R>(a, s);
    if (a.M <= 8) return launch_epi<8, R>(a, s);
    return launch_epi<16, R>(a, s);
}

}  // namespace

// OF_E_SHAPE = not a skinny problem (the caller continues with the tile kernels).  Preconditions already checked by
// of_gemm: K % 8 == 0, 16-byte aligned operands and leading dimensions.
int of_gemm_skinny_try(const OfGemmArgs& a, of_stream_t s) {
    if (a.M > 16 || a.a_trans || a.b_trans) return OF_E_SHAPE;
    if (a.epi != OF_EPI_STORE_BF16 && a.epi != OF_EPI_GELU && a.epi != OF_EPI_GATE_RESID) return OF_E_SHAPE;
    // few output columns: two weight rows per workgroup keep >= 2 workgroups per CU in flight up to N = 2048
    return a.N <= 2048 ? launch_rows<2>(a, s) : launch_rows<4>(a, s);
}
