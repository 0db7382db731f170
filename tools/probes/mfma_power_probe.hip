// MFMA-only probe (gfx950): the same FLOPs through v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16, one wave per SIMD,
// 256 accumulator registers, random bf16 operands held in registers -- what the power-limited clock gives each instruction
// shape.  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power_probe.hip -o tools/probes/mfma_power_probe
// PROFILING TOOL, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int ORDER>
__global__ void __launch_bounds__(256, 1) probe(const s16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    s16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(blockIdx.x * 16 + i) * 256 + lane];
        b[i] = src[(blockIdx.x * 16 + 8 + i) * 256 + lane];
    }
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)          // one "stage": 64 MFMAs = 4 k-steps x 4 x 4 fragments, accumulators in place (AGPRs)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 2) + (k & 1) * 4]), "v"(b[(i & 3) + (k >> 1) * 4]));
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[i][e];
    } else {
        f32x4 acc[64];
#pragma unroll
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)          // one "stage": 128 MFMAs = 2 k-steps x 8 x 8 fragments, accumulators in place (AGPRs)
#pragma unroll
                for (int i = 0; i < 64; ++i)
                {
                    if (ORDER == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 3)]), "v"(b[(i & 7)]));
                    else if (ORDER == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i & 7)]), "v"(b[(i >> 3)]));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i & 7)]), "v"(b[((i & 7) + (i >> 3)) & 7]));
                }
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // MFMA results -> VALU reads
#pragma unroll
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += acc[i][e];
    }
    out[blockIdx.x * 256 + lane] = sum;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 128;      // 128 stages = the K = 8192 tile of the GEMM
    const int zeros = argc > 2 ? atoi(argv[2]) : 0;
    const int grid = 256;
    std::vector<unsigned short> h((size_t)grid * 16 * 256 * 8);
    srand(1);
    for (auto& v : h) {
        // random bf16 in roughly N(0, 1): sign + exponent around 127 + random mantissa
        const float f = ((rand() & 0xffff) / 32768.0f - 1.0f) * 2.0f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = zeros ? 0 : (unsigned short)(u >> 16);
    }
    s16x8* d;
    float* o;
    hipMalloc(&d, h.size() * 2);
    hipMalloc(&o, grid * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int shape : {32, 16, 161, 162, 32, 16, 161, 162}) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int j = 0; j < 10; ++j) {
                if (shape == 32) probe<32, 0><<<grid, 256>>>(d, o, iters);
                else if (shape == 16) probe<16, 0><<<grid, 256>>>(d, o, iters);
                else if (shape == 161) probe<16, 1><<<grid, 256>>>(d, o, iters);
                else probe<16, 2><<<grid, 256>>>(d, o, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms / 10 < best) best = ms / 10;
        }
        const double flops = 2.0 * 256 * 256 * 64 * (double)iters * grid;      // one 256x256x64 stage per iteration per workgroup
        printf("{\"mfma\": \"%s\", \"fill\": \"%s\", \"stages\": %d, \"us\": %.1f, \"tflops\": %.0f}\n", shape == 32 ? "32x32x16" : shape == 16 ? "16x16x32 first operand held over 8" : shape == 161 ? "16x16x32 second operand held over 8" : "16x16x32 both operands change",
               zeros ? "zeros" : "random", iters, best * 1e3, flops / best / 1e9);
    }
    return 0;
}
