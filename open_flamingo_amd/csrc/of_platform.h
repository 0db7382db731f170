// Platform layer for the open_flamingo_amd HIP kernels (gfx950 / CDNA4 only).
//
// Every kernel in csrc/ is written against the small of_* vocabulary below.  The product build
// (hipcc --offload-arch=gfx950) maps it 1:1 onto CDNA4 builtins.  Defining OF_HOST_EMU instead
// (tests/emu/, host clang, TEST INFRASTRUCTURE ONLY) maps it onto a fiber-based SIMT emulator so the
// index arithmetic of the kernels (tile maps, swizzles, MFMA fragment layouts, masks) can be checked
// against the oracle on a machine without a GPU.  The product never loads the emulator build.
#pragma once
#include <stdint.h>
#include <stddef.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct of_dim3 {
    unsigned x, y, z;
};
constexpr int OF_NUM_CUS = 256;      // MI355X

#ifndef OF_HOST_EMU
// =============================================================================== gfx950 device build
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "open_flamingo_amd kernels are written for gfx950 (MI355X) only: MFMA shapes, LDS-DMA, permlane swaps and the cache-policy hand-off of the stream-K path are validated there and nowhere else"
#endif
#define OF_DEV __device__ __forceinline__
#define OF_HOSTDEV __host__ __device__ __forceinline__
#define OF_GLOBAL __global__
#define OF_INLINE_LAMBDA __attribute__((always_inline))   // lambdas whose parameters index register arrays must fold
#define OF_BOUNDS(threads, waves_per_simd) __launch_bounds__(threads, waves_per_simd)
typedef hipStream_t of_stream_t;
typedef __bf16 of_bf16x8n __attribute__((ext_vector_type(8)));

OF_DEV int of_tid() { return threadIdx.x; }
OF_DEV int of_bid_x() { return blockIdx.x; }
OF_DEV int of_bid_y() { return blockIdx.y; }
OF_DEV int of_bid_z() { return blockIdx.z; }
OF_DEV int of_gdim_x() { return gridDim.x; }
OF_DEV int of_bdim_x() { return blockDim.x; }
OF_DEV char* of_smem() {
    extern __shared__ __attribute__((aligned(16))) char of_smem_[];
    return of_smem_;
}
OF_DEV void of_sync() { __syncthreads(); }
// D(16x16, f32) += A(16x32 bf16) * B(32x16 bf16).  Lane l supplies A[l&15][8*(l>>4)+0..7] and
// B[8*(l>>4)+0..7][l&15]; it receives D[4*(l>>4)+r][l&15], r=0..3 (cdna_hip_programming.md section 3).
OF_DEV f32x4 of_mfma(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(of_bf16x8n, a),
                                                   __builtin_bit_cast(of_bf16x8n, b), c, 0, 0, 0);
}
// The same accumulating IN PLACE in an accumulation register (AGPR), by inline asm: with 64 such accumulators per wave
// (gemm_w4m.hip) the builtin form leaves hipcc shuttling accumulators between AGPRs and VGPRs around every MFMA
// (390 v_accvgpr_* + 74 s_nop per 128 MFMAs in the cross-compiled loop).  Results are read only after of_mfma_acc_settle().
OF_DEV void of_mfma_acc(s16x8 a, s16x8 b, f32x4& c) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
// pins a value that C code has just computed back into an accumulation register (the compiler would otherwise be free to keep
// the 256 updated accumulators of gemm_w4m.hip's stream-K fix-up in VGPRs until the epilogue reads them: spills)
OF_DEV void of_acc_pin(f32x4& c) { asm volatile("" : "+a"(c)); }
// The compiler does not see the MFMAs inside the asm, so its hazard recognizer does not cover VALU-write -> MFMA-read either: a
// register copy it places right in front of the first MFMA of a loop (fragments that change registers on the way into a loop)
// can still be under way for the upper half of the wave when the MFMA reads it.  Callers put of_mfma_acc_guard() at the points
// where such copies can appear (top of a K stage).
OF_DEV void of_mfma_acc_guard() { asm volatile("s_nop 4" ::: "memory"); }
// ... and cover the MFMA-write -> read hazard before the accumulators are used.  The statement has no operand tie to the 64
// accumulators (an asm statement takes at most 30 operands), so formally nothing stops hipcc from placing a read of one above the
// s_nops: tests/test_isa_lint.py counts the wait states between the last v_mfma and the first read of an accumulation register in
// the cross-compiled ISA of every instantiation (>= 11 for an 8-pass MFMA) and fails the CPU suite otherwise.
OF_DEV void of_mfma_acc_settle() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
// D(32x32, f32) += A(32x16 bf16) * B(16x32 bf16).  Lane l supplies A[l&31][8*(l>>5)+0..7] and B[8*(l>>5)+0..7][l&31];
// it receives D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31], r=0..15 (cdna_hip_programming.md section 3).
OF_DEV f32x16 of_mfma32(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(of_bf16x8n, a),
                                                   __builtin_bit_cast(of_bf16x8n, b), c, 0, 0, 0);
}
// ---- a bank of 32 accumulators (16x16 fp32 MFMA results) in FIXED accumulation registers a[4k : 4k + 3], outside hipcc's register
// allocation (gemm_w4s.hip).  With the accumulators as C++ values ("+a" operands of of_mfma_acc) a kernel whose accumulators live
// across a loop over tiles had hipcc permute them between registers with v_accvgpr_mov in the middle of the MFMA stream -- moves it
// does not know to be reads of MFMA results, issued without the wait states: wrong values on hardware (two registers of 128), nothing
// on the emulator.  Here no C++ value ever is an accumulator: every statement that touches the bank names its registers and lists
// all 128 as clobbered, so hipcc keeps nothing of its own in them across those statements (and counts them into the kernel's
// register budget).  k is a compile-time constant after unrolling (the switch folds).  Reads need of_mfma_acc_settle() first.
#define OF_ACCBANK_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
struct of_accbank_t {};
OF_DEV void of_accbank_mfma(of_accbank_t&, int k, s16x8 a, s16x8 b) {
    switch (k) {
        case 0: asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], %0, %1, a[0:3]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 1: asm volatile("v_mfma_f32_16x16x32_bf16 a[4:7], %0, %1, a[4:7]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 2: asm volatile("v_mfma_f32_16x16x32_bf16 a[8:11], %0, %1, a[8:11]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 3: asm volatile("v_mfma_f32_16x16x32_bf16 a[12:15], %0, %1, a[12:15]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 4: asm volatile("v_mfma_f32_16x16x32_bf16 a[16:19], %0, %1, a[16:19]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 5: asm volatile("v_mfma_f32_16x16x32_bf16 a[20:23], %0, %1, a[20:23]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 6: asm volatile("v_mfma_f32_16x16x32_bf16 a[24:27], %0, %1, a[24:27]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 7: asm volatile("v_mfma_f32_16x16x32_bf16 a[28:31], %0, %1, a[28:31]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 8: asm volatile("v_mfma_f32_16x16x32_bf16 a[32:35], %0, %1, a[32:35]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 9: asm volatile("v_mfma_f32_16x16x32_bf16 a[36:39], %0, %1, a[36:39]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 10: asm volatile("v_mfma_f32_16x16x32_bf16 a[40:43], %0, %1, a[40:43]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 11: asm volatile("v_mfma_f32_16x16x32_bf16 a[44:47], %0, %1, a[44:47]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 12: asm volatile("v_mfma_f32_16x16x32_bf16 a[48:51], %0, %1, a[48:51]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 13: asm volatile("v_mfma_f32_16x16x32_bf16 a[52:55], %0, %1, a[52:55]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 14: asm volatile("v_mfma_f32_16x16x32_bf16 a[56:59], %0, %1, a[56:59]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 15: asm volatile("v_mfma_f32_16x16x32_bf16 a[60:63], %0, %1, a[60:63]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 16: asm volatile("v_mfma_f32_16x16x32_bf16 a[64:67], %0, %1, a[64:67]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 17: asm volatile("v_mfma_f32_16x16x32_bf16 a[68:71], %0, %1, a[68:71]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 18: asm volatile("v_mfma_f32_16x16x32_bf16 a[72:75], %0, %1, a[72:75]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 19: asm volatile("v_mfma_f32_16x16x32_bf16 a[76:79], %0, %1, a[76:79]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 20: asm volatile("v_mfma_f32_16x16x32_bf16 a[80:83], %0, %1, a[80:83]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 21: asm volatile("v_mfma_f32_16x16x32_bf16 a[84:87], %0, %1, a[84:87]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 22: asm volatile("v_mfma_f32_16x16x32_bf16 a[88:91], %0, %1, a[88:91]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 23: asm volatile("v_mfma_f32_16x16x32_bf16 a[92:95], %0, %1, a[92:95]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 24: asm volatile("v_mfma_f32_16x16x32_bf16 a[96:99], %0, %1, a[96:99]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 25: asm volatile("v_mfma_f32_16x16x32_bf16 a[100:103], %0, %1, a[100:103]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 26: asm volatile("v_mfma_f32_16x16x32_bf16 a[104:107], %0, %1, a[104:107]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 27: asm volatile("v_mfma_f32_16x16x32_bf16 a[108:111], %0, %1, a[108:111]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 28: asm volatile("v_mfma_f32_16x16x32_bf16 a[112:115], %0, %1, a[112:115]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 29: asm volatile("v_mfma_f32_16x16x32_bf16 a[116:119], %0, %1, a[116:119]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 30: asm volatile("v_mfma_f32_16x16x32_bf16 a[120:123], %0, %1, a[120:123]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
        case 31: asm volatile("v_mfma_f32_16x16x32_bf16 a[124:127], %0, %1, a[124:127]" ::"v"(a), "v"(b) : OF_ACCBANK_CLOBBERS); break;
    }
}
OF_DEV void of_accbank_zero(of_accbank_t&) {
    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0" ::: OF_ACCBANK_CLOBBERS);
    asm volatile("v_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0" ::: OF_ACCBANK_CLOBBERS);
    asm volatile("v_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0" ::: OF_ACCBANK_CLOBBERS);
    asm volatile("v_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" ::: OF_ACCBANK_CLOBBERS);
}
OF_DEV f32x4 of_accbank_read(of_accbank_t&, int k) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    switch (k) {
        case 0: asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 1: asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 2: asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_accvgpr_read_b32 %2, a10\n\tv_accvgpr_read_b32 %3, a11" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 3: asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\tv_accvgpr_read_b32 %2, a14\n\tv_accvgpr_read_b32 %3, a15" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 4: asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 5: asm volatile("v_accvgpr_read_b32 %0, a20\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a22\n\tv_accvgpr_read_b32 %3, a23" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 6: asm volatile("v_accvgpr_read_b32 %0, a24\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a26\n\tv_accvgpr_read_b32 %3, a27" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 7: asm volatile("v_accvgpr_read_b32 %0, a28\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a30\n\tv_accvgpr_read_b32 %3, a31" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 8: asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 9: asm volatile("v_accvgpr_read_b32 %0, a36\n\tv_accvgpr_read_b32 %1, a37\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a39" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 10: asm volatile("v_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a43" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 11: asm volatile("v_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a47" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 12: asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 13: asm volatile("v_accvgpr_read_b32 %0, a52\n\tv_accvgpr_read_b32 %1, a53\n\tv_accvgpr_read_b32 %2, a54\n\tv_accvgpr_read_b32 %3, a55" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 14: asm volatile("v_accvgpr_read_b32 %0, a56\n\tv_accvgpr_read_b32 %1, a57\n\tv_accvgpr_read_b32 %2, a58\n\tv_accvgpr_read_b32 %3, a59" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 15: asm volatile("v_accvgpr_read_b32 %0, a60\n\tv_accvgpr_read_b32 %1, a61\n\tv_accvgpr_read_b32 %2, a62\n\tv_accvgpr_read_b32 %3, a63" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 16: asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 17: asm volatile("v_accvgpr_read_b32 %0, a68\n\tv_accvgpr_read_b32 %1, a69\n\tv_accvgpr_read_b32 %2, a70\n\tv_accvgpr_read_b32 %3, a71" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 18: asm volatile("v_accvgpr_read_b32 %0, a72\n\tv_accvgpr_read_b32 %1, a73\n\tv_accvgpr_read_b32 %2, a74\n\tv_accvgpr_read_b32 %3, a75" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 19: asm volatile("v_accvgpr_read_b32 %0, a76\n\tv_accvgpr_read_b32 %1, a77\n\tv_accvgpr_read_b32 %2, a78\n\tv_accvgpr_read_b32 %3, a79" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 20: asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 21: asm volatile("v_accvgpr_read_b32 %0, a84\n\tv_accvgpr_read_b32 %1, a85\n\tv_accvgpr_read_b32 %2, a86\n\tv_accvgpr_read_b32 %3, a87" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 22: asm volatile("v_accvgpr_read_b32 %0, a88\n\tv_accvgpr_read_b32 %1, a89\n\tv_accvgpr_read_b32 %2, a90\n\tv_accvgpr_read_b32 %3, a91" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 23: asm volatile("v_accvgpr_read_b32 %0, a92\n\tv_accvgpr_read_b32 %1, a93\n\tv_accvgpr_read_b32 %2, a94\n\tv_accvgpr_read_b32 %3, a95" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 24: asm volatile("v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\tv_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 25: asm volatile("v_accvgpr_read_b32 %0, a100\n\tv_accvgpr_read_b32 %1, a101\n\tv_accvgpr_read_b32 %2, a102\n\tv_accvgpr_read_b32 %3, a103" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 26: asm volatile("v_accvgpr_read_b32 %0, a104\n\tv_accvgpr_read_b32 %1, a105\n\tv_accvgpr_read_b32 %2, a106\n\tv_accvgpr_read_b32 %3, a107" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 27: asm volatile("v_accvgpr_read_b32 %0, a108\n\tv_accvgpr_read_b32 %1, a109\n\tv_accvgpr_read_b32 %2, a110\n\tv_accvgpr_read_b32 %3, a111" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 28: asm volatile("v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\tv_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 29: asm volatile("v_accvgpr_read_b32 %0, a116\n\tv_accvgpr_read_b32 %1, a117\n\tv_accvgpr_read_b32 %2, a118\n\tv_accvgpr_read_b32 %3, a119" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 30: asm volatile("v_accvgpr_read_b32 %0, a120\n\tv_accvgpr_read_b32 %1, a121\n\tv_accvgpr_read_b32 %2, a122\n\tv_accvgpr_read_b32 %3, a123" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
        case 31: asm volatile("v_accvgpr_read_b32 %0, a124\n\tv_accvgpr_read_b32 %1, a125\n\tv_accvgpr_read_b32 %2, a126\n\tv_accvgpr_read_b32 %3, a127" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : : OF_ACCBANK_CLOBBERS); break;
    }
    return r;
}
// shader clock (s_memtime; timing aid of the ablation builds).  The value returns through the scalar-memory counter,
// so reading it also waits for the wave's outstanding LDS operations (lgkmcnt is shared).
OF_DEV unsigned of_cycles() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return (unsigned)t;
}
OF_DEV void of_setprio_hi() { __builtin_amdgcn_s_setprio(1); }
OF_DEV void of_setprio_lo() { __builtin_amdgcn_s_setprio(0); }
// pins the instruction scheduler: nothing moves across this point
// An integer the optimiser must treat as freshly defined here: values derived from it cannot be hoisted out of the enclosing loop (LLVM
// hoists every loop-invariant address term and then SPILLS the lot when registers run out; re-deriving them per iteration is 1-2 VALU).
OF_DEV int of_opaque_i(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
OF_DEV void of_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// asks the scheduler for exactly `n` instructions of class `mask` at this point of a pinned sequence (LLVM SchedGroupMask:
// MFMA 0x8, VMEM read 0x20, DS read 0x100, DS write 0x200); compile-time only, the emulator ignores it
#define OF_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
OF_DEV int of_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a pointer every lane holds the same value of, moved to SGPRs (e.g. one LOADED from global memory behind earlier stores of the
// kernel: such a load is a vector load, its result formally divergent -- a buffer descriptor made from it would be waterfalled)
OF_DEV const void* of_uniform_ptr(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
// Point where the lanes of ONE wave exchange data through LDS: hardware executes a wave in lock-step and its LDS
// operations in program order, so this is only a compiler scheduling fence (the emulator needs a real rendezvous).
OF_DEV void of_wave_sync() { __builtin_amdgcn_wave_barrier(); }
// ds_read_b64_tr_b16: within each 16-lane group, lane i supplies the LDS address of 4 contiguous bf16
// (row i>>2, column chunk i&3 of a 4x16 block) and receives column i of that block (4 rows).
OF_DEV s16x4 of_lds_tr(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
// ---- LDS-DMA (global -> LDS without a VGPR round trip), issued by INLINE ASM on purpose.
// Through the builtins (__builtin_amdgcn_global_load_lds / raw_ptr_buffer_load_lds) hipcc's waitcnt pass knows that LDS-DMA
// writes are pending and puts `s_waitcnt vmcnt(0)` in front of every later LDS read it cannot prove disjoint -- in practice in
// front of every ds_read_b64_tr_b16 (the transposed-fragment reads of K-strided GEMM operands and of the attention kernels):
// the whole ring drains once per stage, which is what made every DMA kernel with a K-strided operand 15-50 % slower than its
// K-contiguous twin (round 3: cross-compiled ISA of of_gemm_mid_kernel<false,true,..>, a vmcnt(0) the source never asked for).
// Ordering between a DMA and the reads of its data is this code's job anyway (the issuing wave's of_wait_vm<N> + a barrier for
// the other waves); as inline asm the compiler neither tracks nor "protects" it.  Its own vmcnt bookkeeping for ordinary
// loads stays correct: vmcnt retires in order, so untracked operations only make its waits conservative.
// M0 = LDS destination of the wave (base + lane*16 is applied by the hardware); s_nop 0 = the M0-write -> LDS-DMA hazard.
// M0 cannot be declared as a clobber: hipcc reserves it ("inline asm clobber list contains reserved registers: m0" -- behaviour
// undefined if listed).  It is sound undeclared because the compiler never holds a value in M0 across statements on gfx950: its own
// uses (the LDS-DMA builtins) write M0 right in front of each use, and LDS instructions need no M0 on gfx9+.
// tests/test_isa_lint.py pins that on the ISA: every mention of m0 is a write, each followed by an LDS-DMA load before the next.
OF_DEV unsigned of_lds_u32(const void* p) { return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p); }
// Kernels WITHOUT transposed-fragment reads (both operands K-contiguous) keep the builtin form: nothing there triggers the
// extra wait and the compiler schedules the M0 set-up better than the asm's fixed s_mov + s_nop (same-box: 1-2 % faster,
// profiles/r03b_gemm_ab_*.jsonl); `TRSAFE = true` selects the inline-asm form.  -DOF_DMA_VIA_BUILTIN (tools/ab builds only)
// forces the builtin everywhere, for A/B of the effect.
template <bool TRSAFE = true>
OF_DEV void of_glds16(const void* gsrc, void* lds_wave_base) {
#ifndef OF_DMA_VIA_BUILTIN
    if (TRSAFE) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(of_lds_u32(lds_wave_base))) : "memory");
        return;
    }
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same with the non-temporal policy (nt): data that is read ONCE by one workgroup (the saved activation a *_DOT epilogue multiplies
// with, 134 MB per launch) should not displace the operand panels every workgroup of the XCD re-reads from its L2.
template <bool TRSAFE = true>
OF_DEV void of_glds16_nt(const void* gsrc, void* lds_wave_base) {
#ifndef OF_DMA_VIA_BUILTIN
    if (TRSAFE) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(of_lds_u32(lds_wave_base))) : "memory");
        return;
    }
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
// Buffer-descriptor loads: wave-uniform 128-bit descriptor (base pointer in SGPRs) + per-lane 32-bit byte offset +
// scalar byte offset -- no 64-bit per-lane address arithmetic.  `base` must be provably wave-uniform (kernel arguments /
// blockIdx-derived), or hipcc wraps every load in a waterfall loop.
struct of_buf_t {
    __amdgpu_buffer_rsrc_t r;      // for the builtin register loads
    u32x4 w;                       // the same descriptor as four dwords, for the inline-asm LDS-DMA
};
OF_DEV of_buf_t of_buf_make(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    return of_buf_t{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000),
                    u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0xffffffffu, 0x00020000u}};
}
OF_DEV u32x4 of_buf_load16(of_buf_t b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, (int)soff, 0));
}
OF_DEV void of_buf_store16(of_buf_t b, unsigned voff, unsigned soff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, b.r, (int)voff, (int)soff, 0);
}
// The same at SYSTEM scope (sc0 sc1: cache-policy bits 0 and 4 on gfx940+): the store writes through this XCD's L2, the load does
// not take a line this XCD's L2 may hold from before -- data handed from one workgroup to another inside a launch (stream-K partial
// tiles) crosses XCDs this way without flushing / invalidating a whole L2 (what a device-scope release / acquire fence costs: every
// other workgroup of the XCD then re-fetches its operand panels).  Completion of the store = s_waitcnt vmcnt(0) of the issuing wave.
OF_DEV void of_buf_store16_sys(of_buf_t b, unsigned voff, unsigned soff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, b.r, (int)voff, (int)soff, 17);
}
OF_DEV u32x4 of_buf_load16_sys(of_buf_t b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, (int)soff, 17));
}
// LDS-DMA through a buffer descriptor: 16 bytes per lane straight into LDS at (wave-uniform base + lane*16); completion is
// tracked only by the issuing wave's vmcnt (+ a barrier for other waves), like of_glds16
template <bool TRSAFE = true>
OF_DEV void of_buf_load16_lds(of_buf_t b, unsigned voff, unsigned soff, void* lds_wave_base) {
#ifndef OF_DMA_VIA_BUILTIN
    if (TRSAFE) {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(b.w), "s"(soff),
                     "s"(__builtin_amdgcn_readfirstlane(of_lds_u32(lds_wave_base)))
                     : "memory");
        return;
    }
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// The same with the destination as a wave-uniform LDS byte address (of_lds_base(smem) + offset): the address arithmetic stays
// on 32-bit scalars -- through a generic pointer every piece pays an address-space cast (null check + two v_readfirstlane)
// in front of its M0 write.
OF_DEV unsigned of_lds_base(const void* smem) { return (unsigned)__builtin_amdgcn_readfirstlane((int)of_lds_u32(smem)); }
template <bool TRSAFE = true>
OF_DEV void of_buf_load16_lds_at(of_buf_t b, unsigned voff, unsigned soff, unsigned lds_addr) {
#ifndef OF_DMA_VIA_BUILTIN
    if (TRSAFE) {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(b.w), "s"(soff), "s"(lds_addr) : "memory");
        return;
    }
#endif
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (lds_ptr_t)(size_t)lds_addr, 16, (int)voff, (int)soff, 0, 0);
}
template <int N>
OF_DEV void of_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
OF_DEV void of_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// vmcnt(0) as an instruction hipcc's wait-count pass SEES (the asm form is opaque to it): registers loaded before it count as arrived,
// so the pass does not add its own, conservative wait (vmcnt(0) again, behind younger loads) at their first use in the next loop iteration
OF_DEV void of_wait_vm0_visible() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) expcnt(7) lgkmcnt(15)
    asm volatile("" ::: "memory");
}
// bare s_barrier (no implicit vmcnt(0) drain, unlike __syncthreads with LDS-DMA in flight) -- fenced for the COMPILER on both
// sides: llvm.amdgcn.s.barrier is IntrNoMem, so nothing but these two empty asm statements tells hipcc that LDS reads must not
// move across it (the kernels' loops happen to have an asm s_waitcnt in front of every barrier; their prologues and epilogues do
// not).  Added in round 3 while chasing wrong results of an unrolled K loop in gemm_w4m.hip: it changed the register allocation
// of the GEMM kernels, not their order, and was not that bug's cause (of_mfma_acc_guard() is) -- kept as the correct contract.
OF_DEV void of_barrier_raw() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// ---- workgroup-to-workgroup hand-off inside ONE launch (stream-K fix-up of gemm_w4m.hip): a producer publishes data it
// wrote to global memory, a consumer on another CU -- possibly another XCD, whose L2 is a different cache -- picks it up.
// Device (agent) scope: the release makes this XCD's dirty L2 lines visible to the others, the acquire drops stale lines
// (LLVM AMDGPU memory model for gfx942 / gfx950: buffer_wbl2 sc1 / buffer_inv sc1 around the flag access).
// The flag itself is a relaxed device-scope atomic; ordering against the data is the caller's: the producer waits for its
// system-scope stores (of_buf_store16_sys + of_wait_vm<0>, then a workgroup barrier) before it raises the flag, the consumer reads
// the data with system-scope loads (of_buf_load16_sys) after it has seen the flag.
OF_DEV void of_flag_publish(int* flag, int value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// No release / acquire FENCE is involved (a device-scope fence writes back / invalidates a whole L2 per hand-off: +15..+26 % on the
// 400-tile launches, profiles/r04f_*): visibility rests on the sc0 sc1 cache policy of the payload accesses, which is validated on
// gfx950 only -- this file refuses to compile device code for anything else (below) -- and by the MI355X GPU tests (the host emulator
// substitutes real acquire / release atomics and cannot see an ordering bug).  The spin is bounded: forward progress depends on
// the producer (a LOWER workgroup id, dispatched earlier) being resident; after ~2^26 polls (tens of seconds) the wave traps
// instead of hanging the device.
OF_DEV void of_flag_await(const int* flag, int value) {       // spin with a short sleep: the producer needs the memory pipes
    unsigned polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) {
        __builtin_amdgcn_s_sleep(8);
        if (++polls == (1u << 26)) __builtin_trap();
    }
}
OF_DEV float of_shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
OF_DEV int of_shfl_xor_i(int v, int m) { return __shfl_xor(v, m, 64); }
OF_DEV float of_shfl(float v, int src) { return __shfl(v, src, 64); }
OF_DEV void of_atomic_add(float* p, float v) { atomicAdd(p, v); }
OF_DEV float of_exp(float x) { return __expf(x); }
// bare v_exp_f32 / v_log_f32 (base 2, no denormal fix-up sequence): softmax in the log2 domain
OF_DEV float of_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
OF_DEV float of_log2(float x) { return __builtin_amdgcn_logf(x); }
OF_DEV float of_max(float a, float b) { return __builtin_fmaxf(a, b); }      // v_max_f32 (a ?: b keeps NaN order: cmp + cndmask)
// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) without the LDS round trips of
// ds_bpermute: v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows between two registers inside the VALU.
// Inline asm: given the same value in both operands the builtin forms are folded away by the compiler (it treats the swap as
// lane-wise pure); the s_nop covers the VALU-write -> permlane-read hazard the compiler would otherwise pad.
OF_DEV void of_swap_rows16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
OF_DEV void of_swap_rows32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// 2 x 2 exchange between the 16-lane rows r and r ^ 1 of a wave: afterwards an even row holds (own a, partner's a) and an odd
// row (partner's b, own b) -- turns "4 columns of block 2m and of block 2m+1 per lane" into 8 consecutive columns of one block
OF_DEV void of_pair_rows16(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
OF_DEV float of_rows_max(float x) {
    float a = x, b = x;
    of_swap_rows16(a, b);        // a = rows {0,0,2,2}, b = rows {1,1,3,3} of x
    x = __builtin_fmaxf(a, b);
    a = x;
    b = x;
    of_swap_rows32(a, b);        // a = lower half twice, b = upper half twice
    return __builtin_fmaxf(a, b);
}
OF_DEV float of_rows_sum(float x) {
    float a = x, b = x;
    of_swap_rows16(a, b);
    x = a + b;
    a = x;
    b = x;
    of_swap_rows32(a, b);
    return a + b;
}
// true if the predicate holds in ANY lane of the wave (the result is wave-uniform: scalar branch)
OF_DEV bool of_wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
// 1-ulp hardware reciprocal (v_rcp_f32) instead of the ~12-instruction IEEE division sequence
OF_DEV float of_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
typedef __bf16 of_bf16x2n __attribute__((ext_vector_type(2)));
// c + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs, fp32 accumulate, ONE v_dot2c_f32_bf16 (gfx950)
OF_DEV float of_dot2_bf16(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(of_bf16x2n, a), __builtin_bit_cast(of_bf16x2n, b), c, false);
}
// two fp32 -> packed bf16, round-to-nearest-even, in ONE v_cvt_pk_bf16_f32 (gfx950)
OF_DEV unsigned of_pack_bf16(float lo, float hi) {
    of_bf16x2n v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
OF_DEV float of_erf(float x) { return erff(x); }
OF_DEV float of_tanh(float x) { return tanhf(x); }
OF_DEV float of_rsqrt(float x) { return rsqrtf(x); }
OF_DEV float of_log(float x) { return __logf(x); }

template <class K, class A>
static inline int of_launch(K kernel, of_dim3 grid, int block, size_t smem, of_stream_t s, const A& args) {
    hipLaunchKernelGGL(kernel, dim3(grid.x, grid.y, grid.z), dim3(block), smem, s, args);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
static inline int of_memset_async(void* p, int v, size_t n, of_stream_t s) {
    hipError_t e = hipMemsetAsync(p, v, n, s);
    return e == hipSuccess ? 0 : (int)e;
}
#else
// =============================================================================== host SIMT emulator
#include "of_emu.h"
#endif

// ------------------------------------------------------------------------------- shared helpers
#ifndef OF_HOSTDEV
#define OF_HOSTDEV OF_DEV
#endif

OF_DEV float of_bf16_to_f32(bf16_t h) {
    unsigned u = ((unsigned)h) << 16;
    return __builtin_bit_cast(float, u);
}
// round-to-nearest-even; NaN stays NaN (quiet)
OF_DEV bf16_t of_f32_to_bf16(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
#ifdef OF_HOST_EMU
OF_DEV float of_dot2_bf16(unsigned a, unsigned b, float c) {
    return c + of_bf16_to_f32((bf16_t)(a & 0xffff)) * of_bf16_to_f32((bf16_t)(b & 0xffff)) +
           of_bf16_to_f32((bf16_t)(a >> 16)) * of_bf16_to_f32((bf16_t)(b >> 16));
}
OF_DEV unsigned of_pack_bf16(float lo, float hi) {
    return (unsigned)of_f32_to_bf16(lo) | ((unsigned)of_f32_to_bf16(hi) << 16);
}
#endif
#ifndef OF_HOST_EMU
// Sum over the 64 lanes, result in every lane, entirely inside the VALU: four DPP adds give every 16-lane row its total
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), two permlane swaps add the four rows.  __shfl_xor would
// make each of the six steps a ds_bpermute round trip through the LDS pipeline (~100+ cycles of latency apiece).
OF_DEV float of_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return of_rows_sum(v);
}
#else
OF_DEV float of_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += of_shfl_xor(v, m);
    return v;
}

#endif
// erf-GELU (nn.GELU() default, reference helpers.py:20) for the GEMM epilogues.  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 rounding of the stored result) so that one exp + one rcp + 6 fma replace
// the ~40-instruction libm erff; the same exp(-a^2/2) also yields the Gaussian term of the derivative.
OF_DEV void of_gelu_parts(float a, float& cdf, float& e) {
    const float x = fabsf(a) * 0.70710678118654752f;
    const float t = of_rcp(1.0f + 0.3275911f * x);
    e = of_exp(-x * x);                                  // = exp(-a^2/2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f * (1.0f + (a < 0.f ? -erf_abs : erf_abs));
}
OF_DEV float of_gelu(float a) {
    float cdf, e;
    of_gelu_parts(a, cdf, e);
    return a * cdf;
}
// d/da gelu(a) = Phi(a) + a * phi(a)
OF_DEV float of_dgelu(float a) {
    float cdf, e;
    of_gelu_parts(a, cdf, e);
    return cdf + a * 0.39894228040143268f * e;
}
OF_DEV void of_gelu_both(float a, float& gelu, float& dgelu) {
    float cdf, e;
    of_gelu_parts(a, cdf, e);
    gelu = a * cdf;
    dgelu = cdf + a * 0.39894228040143268f * e;
}
// ---- the same for TWO elements per lane in packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32: two results per instruction at the
// full issue rate) -- the big-tile GEMM epilogues.  Round 4's tile phase probe showed the erf-GELU epilogues to be bound by VALU
// issue of the one wave a SIMD holds (~90 cycles per element through the scalar forms above, 9 of the 12.6 us a GELU
// tile spends behind its K loop), and the quarter-rate transcendentals to be a third of that.
//   forward (Phi only): Abramowitz-Stegun 7.1.28, erf(x) = 1 - (1 + a1 x + .. + a6 x^6)^-16, |error| <= 3e-7: ONE transcendental
//     (v_rcp_f32) instead of two, 7 packed fma + 4 packed squarings per pair;
//   backward (Phi and phi): 7.1.26 as above with 0.5 folded into the polynomial and exp(-a^2/2) as a bare v_exp_f32 of a
//     pre-scaled square.
// |gelu - exact| <= 9e-7, |gelu' - exact| <= 6e-7 over [-10, 10] in fp32 (tests/test_emu_gemm.py::test_fast_erf_gelu_accuracy).
OF_DEV f32x2 of_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
OF_DEV f32x2 of_splat2(float v) { return f32x2{v, v}; }
OF_DEV f32x2 of_gelu2(f32x2 a) {
    const f32x2 x = __builtin_elementwise_abs(a) * 0.70710678118654752f;
    f32x2 p = of_fma2(x, of_splat2(0.0000430638f), of_splat2(0.0002765672f));
    p = of_fma2(x, p, of_splat2(0.0001520143f));
    p = of_fma2(x, p, of_splat2(0.0092705272f));
    p = of_fma2(x, p, of_splat2(0.0422820123f));
    p = of_fma2(x, p, of_splat2(0.0705230784f));
    p = of_fma2(x, p, of_splat2(1.0f));
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    const f32x2 r = {of_rcp(p[0]), of_rcp(p[1])};
    const f32x2 h = of_fma2(r, of_splat2(-0.5f), of_splat2(0.5f));      // 0.5 erf(|a| / sqrt 2)
    return a * (of_splat2(0.5f) + __builtin_elementwise_copysign(h, a));
}
OF_DEV void of_gelu_both2(f32x2 a, f32x2& gelu, f32x2& dgelu) {
    const f32x2 s = a * 0.84932180028801904f;            // a sqrt(log2(e) / 2): exp2(-s^2) = exp(-a^2 / 2)
    const f32x2 m = s * s;
    const f32x2 e = {of_exp2(-m[0]), of_exp2(-m[1])};
    const f32x2 u = of_fma2(__builtin_elementwise_abs(a), of_splat2(0.23164189f), of_splat2(1.0f));     // 1 + 0.3275911 |a| / sqrt 2
    const f32x2 t = {of_rcp(u[0]), of_rcp(u[1])};
    f32x2 q = of_fma2(t, of_splat2(0.5f * 1.061405429f), of_splat2(0.5f * -1.453152027f));
    q = of_fma2(t, q, of_splat2(0.5f * 1.421413741f));
    q = of_fma2(t, q, of_splat2(0.5f * -0.284496736f));
    q = of_fma2(t, q, of_splat2(0.5f * 0.254829592f));
    q = q * t;
    const f32x2 h = of_fma2(-q, e, of_splat2(0.5f));      // 0.5 erf(|a| / sqrt 2)
    const f32x2 cdf = of_splat2(0.5f) + __builtin_elementwise_copysign(h, a);
    gelu = a * cdf;
    dgelu = of_fma2(a * 0.39894228040143268f, e, cdf);
}
