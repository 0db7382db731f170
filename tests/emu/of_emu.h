// TEST INFRASTRUCTURE ONLY -- host SIMT emulator behind csrc/of_platform.h (OF_HOST_EMU builds).
//
// One workgroup = up to 1024 ucontext fibers (one per lane) scheduled round-robin on one OS thread;
// workgroups of a grid are distributed over OS threads.  Barriers and the wave-level collectives
// (MFMA 16x16x32 bf16, ds_read_b64_tr_b16, shuffles) are implemented with the lane->element maps
// documented in /opt/skills/guides/cdna_hip_programming.md section 2-3, so a kernel whose tile /
// swizzle / fragment index math is wrong fails here, on CPU, against the oracle.  What this CANNOT
// prove is that the documented hardware maps are right -- tests/test_gpu_probes.py does that on a
// real MI355X.  Nothing under open_flamingo_amd/ links or loads this build.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#define OF_DEV static inline
#define OF_GLOBAL
#define OF_INLINE_LAMBDA
#define OF_BOUNDS(threads, waves_per_simd)
typedef void* of_stream_t;

namespace of_emu {
constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 256 * 1024;
struct Block {
    int nthreads = 0;
    of_dim3 bid{0, 0, 0}, gdim{1, 1, 1};
    char* smem = nullptr;
    ucontext_t sched;
    ucontext_t ctx[kMaxThreads];
    char* stacks = nullptr;
    bool done[kMaxThreads];
    int cur = 0;
    int blk_count = 0, blk_gen = 0;
    int wave_count[16] = {}, wave_gen[16] = {};
    alignas(16) char xchg[kMaxThreads][64];
    std::function<void()> body;
};
inline thread_local Block* g_blk = nullptr;

inline void yield() { swapcontext(&g_blk->ctx[g_blk->cur], &g_blk->sched); }
inline void block_barrier() {
    Block* b = g_blk;
    int gen = b->blk_gen;
    if (++b->blk_count == b->nthreads) {
        b->blk_count = 0;
        b->blk_gen++;
    } else {
        while (b->blk_gen == gen) yield();
    }
}
inline void wave_barrier() {
    Block* b = g_blk;
    int w = b->cur >> 6;
    int gen = b->wave_gen[w];
    if (++b->wave_count[w] == 64) {
        b->wave_count[w] = 0;
        b->wave_gen[w]++;
    } else {
        while (b->wave_gen[w] == gen) yield();
    }
}
inline void trampoline() {
    Block* b = g_blk;
    b->body();
    b->done[b->cur] = true;
    swapcontext(&b->ctx[b->cur], &b->sched);
}
inline void run_block(Block* b) {
    g_blk = b;
    for (int t = 0; t < b->nthreads; ++t) {
        b->done[t] = false;
        getcontext(&b->ctx[t]);
        b->ctx[t].uc_stack.ss_sp = b->stacks + (size_t)t * kStack;
        b->ctx[t].uc_stack.ss_size = kStack;
        b->ctx[t].uc_link = nullptr;
        makecontext(&b->ctx[t], (void (*)())trampoline, 0);
    }
    b->blk_count = 0;
    for (int w = 0; w < 16; ++w) b->wave_count[w] = 0;
    int remaining = b->nthreads;
    while (remaining > 0) {
        for (int t = 0; t < b->nthreads; ++t) {
            if (b->done[t]) continue;
            b->cur = t;
            swapcontext(&b->sched, &b->ctx[t]);
            if (b->done[t]) --remaining;
        }
    }
}
template <class F>
inline int launch(of_dim3 grid, int block, size_t smem, F body) {
    if (block <= 0 || block > kMaxThreads || (block & 63)) return -100;
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    std::atomic<size_t> next{0};
    unsigned nworkers = std::thread::hardware_concurrency();
    if (nworkers == 0) nworkers = 4;
    if (nworkers > nblocks) nworkers = (unsigned)nblocks;
    auto worker = [&]() {
        Block* b = new Block();
        b->stacks = (char*)malloc(kStack * (size_t)block);
        b->smem = (char*)aligned_alloc(256, ((smem + 255) / 256 + 1) * 256);
        b->nthreads = block;
        b->gdim = grid;
        b->body = body;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b->bid.x = (unsigned)(i % grid.x);
            b->bid.y = (unsigned)((i / grid.x) % grid.y);
            b->bid.z = (unsigned)(i / ((size_t)grid.x * grid.y));
            memset(b->smem, 0xCD, smem);  // poison: reads of unwritten LDS show up as garbage
            run_block(b);
        }
        free(b->smem);
        free(b->stacks);
        delete b;
        g_blk = nullptr;
    };
    std::vector<std::thread> ts;
    for (unsigned i = 0; i < nworkers; ++i) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
    return 0;
}
}  // namespace of_emu

// workgroup-to-workgroup hand-off (stream-K fix-up): workgroups run on parallel OS threads here, taken in block-id order; a
// waiting fiber yields so that the other fibers of its workgroup (and, on other threads, the producer) keep running
OF_DEV void of_flag_publish(int* flag, int value) { __atomic_store_n(flag, value, __ATOMIC_RELEASE); }
OF_DEV void of_flag_await(const int* flag, int value) {
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != value) {
        of_emu::yield();
        std::this_thread::yield();
    }
}
OF_DEV int of_tid() { return of_emu::g_blk->cur; }
OF_DEV int of_bid_x() { return (int)of_emu::g_blk->bid.x; }
OF_DEV int of_bid_y() { return (int)of_emu::g_blk->bid.y; }
OF_DEV int of_bid_z() { return (int)of_emu::g_blk->bid.z; }
OF_DEV int of_gdim_x() { return (int)of_emu::g_blk->gdim.x; }
OF_DEV int of_bdim_x() { return of_emu::g_blk->nthreads; }
OF_DEV char* of_smem() { return of_emu::g_blk->smem; }
OF_DEV void of_sync() { of_emu::block_barrier(); }

OF_DEV float of_emu_bf16f(short h) {
    unsigned u = ((unsigned)(unsigned short)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
OF_DEV f32x4 of_mfma(s16x8 a, s16x8 b, f32x4 c) {
    of_emu::Block* blk = of_emu::g_blk;
    int t = blk->cur, l = t & 63, w0 = t & ~63;
    memcpy(blk->xchg[t], &a, 16);
    memcpy(blk->xchg[t] + 16, &b, 16);
    of_emu::wave_barrier();
    f32x4 d = c;
    int col = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = g * 4 + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            short av, bv;
            memcpy(&av, blk->xchg[w0 + row + 16 * (k >> 3)] + 2 * (k & 7), 2);
            memcpy(&bv, blk->xchg[w0 + col + 16 * (k >> 3)] + 16 + 2 * (k & 7), 2);
            acc = fmaf(of_emu_bf16f(av), of_emu_bf16f(bv), acc);
        }
        d[r] += acc;
    }
    of_emu::wave_barrier();
    return d;
}
OF_DEV void of_mfma_acc(s16x8 a, s16x8 b, f32x4& c) { c = of_mfma(a, b, c); }
OF_DEV void of_mfma_acc_settle() {}
// the fixed-register accumulator bank of the device build (of_platform.h): here just 32 values
struct of_accbank_t {
    f32x4 v[32];
};
OF_DEV void of_accbank_mfma(of_accbank_t& bank, int k, s16x8 a, s16x8 b) { bank.v[k] = of_mfma(a, b, bank.v[k]); }
OF_DEV void of_accbank_zero(of_accbank_t& bank) {
    for (int k = 0; k < 32; ++k) bank.v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
}
OF_DEV f32x4 of_accbank_read(of_accbank_t& bank, int k) { return bank.v[k]; }
// the 64-tile bank of csrc/of_accbank64.h
struct of_accbank64_t {
    f32x4 v[64];
};
OF_DEV void of_accbank64_mfma(of_accbank64_t& bank, int k, s16x8 a, s16x8 b) { bank.v[k] = of_mfma(a, b, bank.v[k]); }
OF_DEV void of_accbank64_zero(of_accbank64_t& bank) {
    for (int k = 0; k < 64; ++k) bank.v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
}
OF_DEV f32x4 of_accbank64_read(of_accbank64_t& bank, int k) { return bank.v[k]; }
OF_DEV void of_accbank64_fence() {}
OF_DEV f32x4 of_mfma_v0(s16x8 a, s16x8 b) { return of_mfma(a, b, f32x4{0.f, 0.f, 0.f, 0.f}); }
OF_DEV void of_mfma_v(s16x8 a, s16x8 b, f32x4& c) { c = of_mfma(a, b, c); }
OF_DEV void of_mfma_settle4(f32x4&, f32x4&, f32x4&, f32x4&) {}
OF_DEV void of_mfma_settle2(f32x4&, f32x4&) {}
OF_DEV void of_mfma_operands4(s16x8&, s16x8&, s16x8&, s16x8&) {}
OF_DEV void of_mfma_operands2(s16x8&, s16x8&) {}
OF_DEV void of_mfma_guard_nomem() {}
OF_DEV void of_acc_pin(f32x4&) {}
OF_DEV void of_mfma_acc_guard() {}
OF_DEV f32x16 of_mfma32(s16x8 a, s16x8 b, f32x16 c) {
    of_emu::Block* blk = of_emu::g_blk;
    int t = blk->cur, l = t & 63, w0 = t & ~63;
    memcpy(blk->xchg[t], &a, 16);
    memcpy(blk->xchg[t] + 16, &b, 16);
    of_emu::wave_barrier();
    f32x16 d = c;
    int col = l & 31, h = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) {
            short av, bv;
            memcpy(&av, blk->xchg[w0 + row + 32 * (k >> 3)] + 2 * (k & 7), 2);
            memcpy(&bv, blk->xchg[w0 + col + 32 * (k >> 3)] + 16 + 2 * (k & 7), 2);
            acc = fmaf(of_emu_bf16f(av), of_emu_bf16f(bv), acc);
        }
        d[r] += acc;
    }
    of_emu::wave_barrier();
    return d;
}
OF_DEV unsigned of_cycles() { return 0; }
OF_DEV void of_setprio_hi() {}
OF_DEV void of_setprio_lo() {}
OF_DEV void of_sched_fence() {}
OF_DEV int of_opaque_i(int v) { return v; }
#define OF_SCHED_GROUP(mask, n) ((void)0)
OF_DEV int of_uniform(int v) { return v; }
OF_DEV const void* of_uniform_ptr(const void* p) { return p; }
OF_DEV void of_wave_sync() { of_emu::wave_barrier(); }
OF_DEV s16x4 of_lds_tr(const void* p) {
    of_emu::Block* blk = of_emu::g_blk;
    int t = blk->cur, i = t & 15, g0 = t & ~15;
    if (((uintptr_t)p) & 7) {
        fprintf(stderr, "of_emu: ds_read_b64_tr_b16 address not 8-byte aligned\n");
        abort();
    }
    memcpy(blk->xchg[t], p, 8);
    of_emu::wave_barrier();
    s16x4 r;
    for (int j = 0; j < 4; ++j) {
        short v;
        memcpy(&v, blk->xchg[g0 + j * 4 + (i >> 2)] + 2 * (i & 3), 2);
        r[j] = v;
    }
    of_emu::wave_barrier();
    return r;
}
// LDS-DMA executes synchronously here: layout (lane -> LDS address) is emulated, asynchrony is not.
template <bool TRSAFE = true>
OF_DEV void of_glds16(const void* gsrc, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + 16 * (of_emu::g_blk->cur & 63), gsrc, 16);
}
template <bool TRSAFE = true>
OF_DEV void of_glds16_nt(const void* gsrc, void* lds_wave_base) { of_glds16<TRSAFE>(gsrc, lds_wave_base); }
template <int N>
OF_DEV void of_wait_vm() {}
OF_DEV void of_wait_lgkm0() {}
OF_DEV void of_wait_vm0_visible() {}
struct of_buf_t {
    const char* base;
};
OF_DEV of_buf_t of_buf_make(const void* base) { return of_buf_t{(const char*)base}; }
OF_DEV u32x4 of_buf_load16(of_buf_t b, unsigned voff, unsigned soff) { return *(const u32x4*)(b.base + voff + soff); }
OF_DEV void of_buf_store16(of_buf_t b, unsigned voff, unsigned soff, u32x4 v) { *(u32x4*)(const_cast<char*>(b.base) + voff + soff) = v; }
OF_DEV void of_buf_store16_sys(of_buf_t b, unsigned voff, unsigned soff, u32x4 v) { of_buf_store16(b, voff, soff, v); }
OF_DEV u32x4 of_buf_load16_sys(of_buf_t b, unsigned voff, unsigned soff) { return of_buf_load16(b, voff, soff); }
template <bool TRSAFE = true>
OF_DEV void of_buf_load16_lds(of_buf_t b, unsigned voff, unsigned soff, void* lds_wave_base) {
    *(u32x4*)((char*)lds_wave_base + (of_emu::g_blk->cur & 63) * 16) = *(const u32x4*)(b.base + voff + soff);
}
OF_DEV unsigned of_lds_base(const void* smem) { return (unsigned)((const char*)smem - of_emu::g_blk->smem); }
template <bool TRSAFE = true>
OF_DEV void of_buf_load16_lds_at(of_buf_t b, unsigned voff, unsigned soff, unsigned lds_addr) {
    *(u32x4*)(of_emu::g_blk->smem + lds_addr + (of_emu::g_blk->cur & 63) * 16) = *(const u32x4*)(b.base + voff + soff);
}
OF_DEV void of_barrier_raw() { of_emu::block_barrier(); }
OF_DEV float of_shfl(float v, int src) {
    of_emu::Block* blk = of_emu::g_blk;
    int t = blk->cur;
    memcpy(blk->xchg[t], &v, 4);
    of_emu::wave_barrier();
    float r;
    memcpy(&r, blk->xchg[(t & ~63) | (src & 63)], 4);
    of_emu::wave_barrier();
    return r;
}
OF_DEV float of_shfl_xor(float v, int m) { return of_shfl(v, (of_emu::g_blk->cur & 63) ^ m); }
OF_DEV int of_shfl_xor_i(int v, int m) {
    float f;
    memcpy(&f, &v, 4);
    f = of_shfl_xor(f, m);
    memcpy(&v, &f, 4);
    return v;
}
OF_DEV bool of_wave_any(bool p) {
    int v = p ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v |= of_shfl_xor_i(v, m);
    return v != 0;
}
OF_DEV void of_pair_rows16(unsigned& a, unsigned& b) {
    const unsigned pa = (unsigned)of_shfl_xor_i((int)a, 16), pb = (unsigned)of_shfl_xor_i((int)b, 16);
    if (((of_emu::g_blk->cur >> 4) & 1) == 0) b = pa;
    else a = pb;
}
OF_DEV float of_rows_max(float x) {
    x = fmaxf(x, of_shfl_xor(x, 16));
    return fmaxf(x, of_shfl_xor(x, 32));
}
OF_DEV float of_rows_sum(float x) {
    x += of_shfl_xor(x, 16);
    return x + of_shfl_xor(x, 32);
}
OF_DEV float of_exp2(float x) { return exp2f(x); }
OF_DEV float of_log2(float x) { return log2f(x); }
OF_DEV float of_max(float a, float b) { return fmaxf(a, b); }
OF_DEV void of_atomic_add(float* p, float v) {
    std::atomic<unsigned>* a = (std::atomic<unsigned>*)p;
    unsigned old = a->load();
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        f += v;
        unsigned nw;
        memcpy(&nw, &f, 4);
        if (a->compare_exchange_weak(old, nw)) break;
    }
}
OF_DEV float of_exp(float x) { return expf(x); }
OF_DEV float of_rcp(float x) { return 1.0f / x; }
OF_DEV float of_erf(float x) { return erff(x); }
OF_DEV float of_tanh(float x) { return tanhf(x); }
OF_DEV float of_rsqrt(float x) { return 1.0f / sqrtf(x); }
OF_DEV float of_log(float x) { return logf(x); }

template <class K, class A>
static inline int of_launch(K kernel, of_dim3 grid, int block, size_t smem, of_stream_t, const A& args) {
    A copy = args;
    return of_emu::launch(grid, block, smem, [kernel, copy]() { kernel(copy); });
}
static inline int of_memset_async(void* p, int v, size_t n, of_stream_t) {
    memset(p, v, n);
    return 0;
}
