// Fragment-streaming probe (gfx950): how fast can ONE 8-wave workgroup per CU pull a weight matrix that every CU reads (L2 resident)
// straight into MFMA operand registers, 32 activation rows per workgroup?  This is the inner loop a row-complete fused
// "LN -> to_q -> attention -> to_out" kernel would run: with only 32 rows per workgroup the weight bytes per
// FLOP are 8x those of a 256x256 tile, so the kernel is bound by the L2 -> CU path (64 B/clk/CU at best), not by the matrix cores.
// Variants of the weight fetch:
//   0  packed:   the matrix re-laid fragment-major ([n-tile][k-step][lane][8 bf16]); one wave instruction = 1 KiB contiguous
//   1  natural:  row-major [N][K]; lane (n = l & 15, g = l >> 4) loads 16 B of row n: 16 rows x 64 B per instruction
//   2  bpermute: row-major, coalesced quads (lane l: row l >> 2, chunk l & 3) + 4 ds_bpermute_b32 into the fragment layout
//   3  packed + no MFMA (fetch only: the L2 -> CU ceiling)
// Shapes: "q" = N 512 x K 2048 (wave w owns n-tiles 4w..4w+3, 64 k-steps), "o" = N 2048 x K 512 (wave w owns 16 n-tiles, 16 k-steps).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/frag_stream_probe.hip -o tools/probes/frag_stream_probe
// PROFILING TOOL, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8n, a), __builtin_bit_cast(bf16x8n, b), c, 0, 0, 0);
}

// NT = n-tiles per wave, KS = k-steps; W: [8 * NT * 16][KS * 32] bf16 (row-major or packed)
template <int NT, int KS, int VAR, int PF, int UT>
__global__ void __launch_bounds__(512, 2) probe(const unsigned short* __restrict__ W, float* __restrict__ out, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int K = KS * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // activation image: 32 rows x K bf16 (zeros are fine: the clock effect of data is not what this probe asks)
    for (int i = tid; i < 32 * K / 8; i += 512) ((u32x4*)smem)[i] = u32x4{0x3f803f80u + (unsigned)i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int perm_src = 4 * (lane & 15) + (lane >> 4);        // variant 2: fragment lane <- coalesced lane
    // buffer loads: wave-uniform descriptor + ONE per-lane byte offset + a scalar byte offset per tile (no 64-bit per-tile addresses)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, 0xffffffff, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned vo_packed = lane * 16, vo_nat = ((lane & 15) * K + (lane >> 4) * 8) * 2, vo_quad = ((lane >> 2) * K + (lane & 3) * 8) * 2;
    auto fetch = [&](int ks, int t) -> s16x8 {
        const int nt = wv * NT + t;
        if (VAR == 0 || VAR == 3) {
            return __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_packed, (nt * KS + ks) * 1024, 0));
        } else if (VAR == 1) {
            return __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_nat, (nt * 16 * K + ks * 32) * 2, 0));
        } else {
            const u32x4 r = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_quad, (nt * 16 * K + ks * 32) * 2, 0));
            u32x4 p;
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = (unsigned)__builtin_amdgcn_ds_bpermute(perm_src * 4, (int)r[e]);
            return __builtin_bit_cast(s16x8, p);
        }
    };
    // a "unit" = UT n-tiles of one k-step (UT = NT: a whole k-step; UT = NT / 2: half of one -- what the 16-tile shape can hold in
    // registers next to its 128 accumulators); PF units are in flight
    constexpr int H = NT / UT;                      // units per k-step
    constexpr int S = PF / H > 0 ? PF / H : 1;      // k-steps per unrolled super-iteration (S * H is a multiple of PF)
    static_assert((S * H) % PF == 0 && KS % S == 0, "unit ring");
    for (int rep = 0; rep < reps; ++rep) {
        s16x8 buf[PF][UT];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int t = 0; t < UT; ++t) buf[p][t] = fetch(p / H, (p % H) * UT + t);
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS; ks0 += S) {
#pragma unroll
            for (int u = 0; u < S * H; ++u) {
                const int ks = ks0 + u / H, h = u % H, slot = u % PF;
                if (VAR == 3) {
#pragma unroll
                    for (int t = 0; t < UT; ++t) acc[0][h * UT + t][0] += (float)buf[slot][t][0];
                } else {
                    s16x8 ua[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int row = m * 16 + (lane & 15), slot16 = (ks * 4 + (lane >> 4));
                        ua[m] = *(const s16x8*)(smem + (size_t)row * K * 2 + (((slot16 & ~15) | ((slot16 & 15) ^ (row & 15))) << 4));
                    }
#pragma unroll
                    for (int t = 0; t < UT; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][h * UT + t] = mfma(buf[slot][t], ua[m], acc[m][h * UT + t]);
                }
                const int un = u + PF, ksn = ks0 + un / H, hn = un % H;      // refill the slot just consumed: PF - 1 units stay in flight
                if (ksn < KS) {
#pragma unroll
                    for (int t = 0; t < UT; ++t) buf[slot][t] = fetch(ksn, hn * UT + t);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) s += acc[m][t][0] + acc[m][t][1] + acc[m][t][2] + acc[m][t][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int NT, int KS, int VAR, int PF, int UT>
static void run(const char* shape, const unsigned short* W, float* out, int grid) {
    const int reps = 8;
    const size_t smem = (size_t)32 * KS * 32 * 2;
    hipFuncSetAttribute((const void*)probe<NT, KS, VAR, PF, UT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NT, KS, VAR, PF, UT>), dim3(grid), dim3(512), smem, 0, W, out, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = (double)8 * NT * 16 * KS * 32 * 2;        // the matrix, once per workgroup per rep
    const double us = best * 1e3 / reps;
    const char* names[] = {"packed", "natural", "bpermute", "packed_nomfma"};
    printf("{\"probe\": \"frag_stream\", \"shape\": \"%s\", \"variant\": \"%s\", \"units_in_flight\": %d, \"tiles_per_unit\": %d, \"grid\": %d, \"us_per_pass\": %.2f, \"MB_per_wg\": %.2f, "
           "\"GBps_per_cu\": %.1f, \"B_per_clk_at_2.1GHz\": %.1f, \"tflops\": %.1f}\n",
           shape, names[VAR], PF, UT, grid, us, bytes / 1e6, bytes / us / 1e3, bytes / us / 1e3 / 2.1, 2.0 * 32 * (8.0 * NT * 16) * (KS * 32) * grid / us / 1e6);
    fflush(stdout);
}

int main() {
    const size_t elems = (size_t)2048 * 2048;
    std::vector<unsigned short> h(elems);
    unsigned s = 12345u;
    for (size_t i = 0; i < elems; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (unsigned short)(0x3c00u + ((s >> 16) & 0x3ffu) + ((s >> 3) & 0x8000u));
    }
    unsigned short* W;
    float* out;
    hipMalloc(&W, elems * 2);
    hipMalloc(&out, (size_t)256 * 512 * 4);
    hipMemcpy(W, h.data(), elems * 2, hipMemcpyHostToDevice);
    for (int grid : {256, 32}) {
        run<4, 64, 0, 2, 4>("q 512x2048", W, out, grid);
        run<4, 64, 0, 4, 4>("q 512x2048", W, out, grid);
        run<4, 64, 1, 2, 4>("q 512x2048", W, out, grid);
        run<4, 64, 1, 4, 4>("q 512x2048", W, out, grid);
        run<4, 64, 2, 2, 4>("q 512x2048", W, out, grid);
        run<4, 64, 3, 4, 4>("q 512x2048", W, out, grid);
        run<16, 16, 0, 2, 8>("o 2048x512", W, out, grid);
        run<16, 16, 0, 4, 4>("o 2048x512", W, out, grid);
        run<16, 16, 1, 2, 8>("o 2048x512", W, out, grid);
        run<16, 16, 1, 4, 4>("o 2048x512", W, out, grid);
        run<16, 16, 2, 2, 8>("o 2048x512", W, out, grid);
        run<16, 16, 3, 4, 4>("o 2048x512", W, out, grid);
        run<16, 16, 0, 2, 4>("o 2048x512", W, out, grid);
    }
    return 0;
}
