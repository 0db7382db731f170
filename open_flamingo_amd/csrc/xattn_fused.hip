// The attention branch of GatedCrossAttentionBlock as ONE row-complete kernel (gfx950):
//
//     y1 = x + tanh(attn_gate) * to_out( softmax_window( to_q(LN(x)) K^T ) V ),   u2 = LN_ff(y1)
//
// reference open_flamingo/src/helpers.py:184-194 (norm, to_q), :192-231 (masked attention over the media window), :231-233 (to_out),
// :267-276 (tanh gate + residual) and the LayerNorm that opens the FeedForward of the same block (helpers.py:18).  Unfused this is
// of_layernorm_fwd -> of_gemm(to_q) -> of_attn_fwd -> of_gemm(to_out, GATE_RESID) -> of_layernorm_fwd: five launches that read the
// fp32 stream three times and write / re-read q, o and y1 (399 MB at OF-3B, B 32, L 256); here x is read once for the statistics and
// once more (from the Infinity Cache) for the residual, y1 / u2 / the tensors the backward needs are written once (216 MB).
//
// One workgroup = 32 consecutive text positions of one sequence x all 8 heads, 8 waves:
//   1. LayerNorm, a wave per row (4 rows per wave), bit-for-bit the arithmetic of of_ln_fwd_kernel; LN(x) goes to HBM (the to_q
//      weight gradient needs it) and into an LDS image [32][d] bf16;
//   2. to_q: wave w computes head w's 64 query columns for all 32 rows: q^T tiles = Wq-fragment x LN(x)^T-fragment.  The weight
//      fragments come STRAIGHT from L2 into registers, from a fragment-major copy of the matrix (of_pack_frag16: one wave load =
//      1 KiB contiguous) -- 32 rows per workgroup move 8x the weight bytes per FLOP of a 256x256 tile, so this phase is bound by the
//      L2 -> CU path, and row-major fragments (16 rows x 64 B per instruction) take 2.7x as long (profiles/r06b_frag_stream_probe.jsonl);
//   3. the windowed softmax(QK^T)V of attention.hip on the wave's own head: q never leaves the accumulators' lane layout (the k index
//      of the QK^T MFMA is permuted to match it), K / V of the head's 64-key block go through a wave-private LDS image, no barrier;
//   4. to_out: wave w owns d / 8 output columns of all 32 rows, K = 512 from the LDS image of o, weight fragments as in 2.;
//   5. epilogue: y1 = x + tanh(gate) * acc in the accumulator registers, row statistics across the 8 waves through LDS, u2 = LN_ff(y1).
//
// What it is bound by (round 6, phase stamps of the tools build, profiles/r06*_xattn_fused_probe*.jsonl; 256 workgroups = one per CU):
// LayerNorm 19-24 us and epilogue 25 us are HBM / fabric phases (100 and 167 MB: 5-7 TB/s), to_q 22 and to_out 19 us are L2 phases
// (every CU streams the two 2-MB weight matrices: 1 GB over the eight L2s, 17 us each at the measured ceiling), attention 7-10 us.  All
// workgroups are in the same phase at the same time, so the phases ADD: ~105 us in the step against ~119 us for the five launches.  A
// workgroup of more rows would halve the weight traffic but its LN(x) image does not fit the LDS (64 rows x 2048 bf16 = 256 KiB).
#include "attn_core.h"

namespace {
using namespace ofa;

#if defined(OF_TOOLS_BUILD) && !defined(OF_HOST_EMU)
// tools/libofhip_tools.so only (tools/probes/xattn_fused_probe.py): wave 0 of every workgroup stamps the 100-MHz wall clock at the
// phase boundaries -- entry, LayerNorm done, to_q done, attention done, to_out's K loop done, last store issued, stores acknowledged
__device__ unsigned long long* of_xf_stamps = nullptr;
}
extern "C" int of_tools_set_xf_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(of_xf_stamps), &p, sizeof(p)); }
namespace {
#define XF_STAMP_DECL() unsigned long long xf_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define XF_STAMP(i) (xf_t[i] = wall_clock64())
#define XF_STAMP_FLUSH()                                                                   \
    do {                                                                                   \
        if (of_xf_stamps) {                                                                \
            of_wait_vm<0>();                                                               \
            XF_STAMP(6);                                                                   \
            if (of_tid() == 0)                                                             \
                for (int i_ = 0; i_ < 8; ++i_) of_xf_stamps[(size_t)of_bid_x() * 8 + i_] = xf_t[i_]; \
        }                                                                                  \
    } while (0)
#else
#define XF_STAMP_DECL()
#define XF_STAMP(i)
#define XF_STAMP_FLUSH()
#endif

constexpr int XR = 32;                      // rows per workgroup
constexpr int X_INNER = 512;                // heads * dim_head = 8 * 64
constexpr int X_REGION_A = 128 * 1024;      // LN(x) image (64 * d bytes), then 8 wave-private K | V images (16 KiB each), then row statistics
constexpr int X_OIMG = 32 * 1024;           // first gamma / beta of the LayerNorm (8 * d bytes), then the o image [32][512] bf16
constexpr int X_SMEM = X_REGION_A + X_OIMG; // 160 KiB: one workgroup per CU

OF_DEV void x_load8(const void* rowp, int is_f32, unsigned eo, float (&v)[8]) {
    if (is_f32) {
#ifdef OF_XF_X_NT         // tools/ab builds only (round 6 A/B: the LayerNorm's read of x with the non-temporal policy)
        const f32x4 a = __builtin_nontemporal_load((const f32x4*)((const float*)rowp + eo));
        const f32x4 b = __builtin_nontemporal_load((const f32x4*)((const float*)rowp + eo + 4));
#else
        const f32x4 a = *(const f32x4*)((const float*)rowp + eo);
        const f32x4 b = *(const f32x4*)((const float*)rowp + eo + 4);
#endif
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        const u32x4 r = *(const u32x4*)((const bf16_t*)rowp + eo);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = of_bf16_to_f32((bf16_t)(r[e] & 0xffff));
            v[2 * e + 1] = of_bf16_to_f32((bf16_t)(r[e] >> 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// fragment-major copy of a row-major [N][K] bf16 matrix: P[(nt * (K / 32) + ks) * 64 + lane] (16 bytes) = W[16 nt + (lane & 15)][32 ks +
// 8 (lane >> 4) .. + 7] -- what lane `lane` supplies as the A (or B) operand of v_mfma_f32_16x16x32_bf16 for n-tile nt, k-step ks
struct PackArgs {
    const bf16_t* W;
    bf16_t* P;
    int N, K;
    long ldw;
};
OF_DEV void pack_piece(const bf16_t* W, bf16_t* P, int K, long ldw, long idx) {
    const int KS = K / 32;
    const int lane = (int)(idx & 63);
    const long tile = idx >> 6;
    const int ks = (int)(tile % KS);
    const long nt = tile / KS;
    *(u32x4*)(P + idx * 8) = *(const u32x4*)(W + (size_t)(nt * 16 + (lane & 15)) * ldw + ks * 32 + (lane >> 4) * 8);
}
OF_GLOBAL void of_pack_frag16_kernel(PackArgs a) {
    const long idx = (long)of_bid_x() * 256 + of_tid();
    if (idx >= (long)(a.N / 16) * (a.K / 32) * 64) return;
    pack_piece(a.W, a.P, a.K, a.ldw, idx);
}
// several matrices in one launch (the step epilogue re-packs every gated block's to_q / to_out weight once per optimizer step)
struct PackBatchArgs {
    int n;
    long end[OF_PACK_BATCH_MAX];                 // cumulative 16-byte pieces
    OfPackDesc d[OF_PACK_BATCH_MAX];
};
OF_GLOBAL void of_pack_frag16_batch_kernel(PackBatchArgs a) {
    const long idx = (long)of_bid_x() * 256 + of_tid();
    if (idx >= a.end[a.n - 1]) return;
    int i = 0;
    while (idx >= a.end[i]) ++i;
    const long first = i ? a.end[i - 1] : 0;
    pack_piece(a.d[i].W, a.d[i].P, a.d[i].K, a.d[i].ldw, idx - first);
}

// acc[mt][t] (+)= P-fragments x image-fragments: the 32 rows of a [32][512] bf16 LDS image (ximg_off layout) times the d / 8 output columns
// this wave owns of a fragment-major weight copy `pk` ([d / 16 n-tiles][16 k-steps][64 lanes][8]); acc[mt][t][r] = out[row 16 mt + (lane &
// 15)][wave * 16 NT + 16 t + 4 (lane >> 4) + r].  The weight fragments come straight from L2 into registers, two units of <= 4 n-tiles in flight.
template <int NT>
OF_DEV void xf_rows32_times_packed(const bf16_t* pk, const char* img, int wave, int lane, f32x4 (&acc)[2][NT]) {
    constexpr int KSO = X_INNER / 32;        // 16 k-steps
    constexpr int OROW = X_INNER * 2;
    constexpr int UT = NT < 4 ? NT : 4;      // n-tiles per unit of the fragment ring
    constexpr int H = NT / UT;               // units per k-step
    constexpr int PFO = 2;                   // units in flight
    constexpr int S = PFO / H > 0 ? PFO / H : 1;
    static_assert((S * H) % PFO == 0 && KSO % S == 0 && NT % UT == 0, "fragment ring");
    const int g = lane >> 4, i16 = lane & 15;
    const of_buf_t bo = of_buf_make(pk);
    const unsigned vo = (unsigned)lane * 16;
    const int obase = i16 * OROW;            // image fragment of row tile mt, k-step ks: + mt * 16 * OROW + (ks >> 2) * 256 + swizzled slot
    u32x4 wb[PFO][UT];
#pragma unroll
    for (int pp = 0; pp < PFO; ++pp)
#pragma unroll
        for (int t = 0; t < UT; ++t) wb[pp][t] = of_buf_load16(bo, vo, (unsigned)(((wave * NT + (pp % H) * UT + t) * KSO + pp / H) * 1024));
#pragma unroll 1
    for (int ks0 = 0; ks0 < KSO; ks0 += S) {
#pragma unroll
        for (int u = 0; u < S * H; ++u) {
            const int ks = ks0 + u / H, h = u % H, slot = u % PFO;
            s16x8 ob[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                ob[mt] = *(const s16x8*)(img + obase + mt * 16 * OROW + (ks >> 2) * 256 + (((((ks & 3) << 2) | g) ^ i16) << 4));
#pragma unroll
            for (int t = 0; t < UT; ++t)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][h * UT + t] = of_mfma(__builtin_bit_cast(s16x8, wb[slot][t]), ob[mt], acc[mt][h * UT + t]);
            const int un = u + PFO, ksn = ks0 + un / H, hn = un % H;
            if (ksn < KSO) {
#pragma unroll
                for (int t = 0; t < UT; ++t) wb[slot][t] = of_buf_load16(bo, vo, (unsigned)(((wave * NT + hn * UT + t) * KSO + ksn) * 1024));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int D>
OF_GLOBAL void OF_BOUNDS(512, 2) of_xattn_fused_fwd_kernel(OfXattnFusedArgs p) {
    constexpr int CPL = (D + 511) / 512;        // 8-column chunks per lane of a row (wave per row)
    constexpr int KSQ = D / 32;                 // k-steps of to_q
    constexpr int NTO = D / 128;                // 16-column output tiles of to_out per wave
    constexpr int UROW = D * 2;                 // bytes per row of the LN(x) image
    constexpr int OROW = X_INNER * 2;           // bytes per row of the o image
    char* smem = of_smem();
    char* u_img = smem;
    char* o_img = smem + X_REGION_A;
    float* s_gamma = (float*)o_img;
    float* s_beta = s_gamma + D;
    const int tid = of_tid(), lane = tid & 63, g = lane >> 4, i16 = lane & 15;
    const int wave = of_uniform(tid >> 6);
    const long row0 = (long)of_bid_x() * XR;    // L % 32 == 0: the 32 rows belong to one sequence
    const long batch = row0 / p.L;
    const int pos0 = (int)(row0 - batch * p.L);
    const unsigned lo = (unsigned)lane * 8;
    XF_STAMP_DECL();
    XF_STAMP(0);

    // ---- the key windows of this wave's 2 x 16 rows and K | V of the first key block they see (head `wave`): requested at
    // kernel entry, they land under the LayerNorm (helpers.py:196-229: attention.hip's header)
    Window w[2];
    int kb_lo[2], kb_hi[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int pos = pos0 + mt * 16 + i16;
        w[mt] = p.text_time ? media_window(p.text_time[batch * p.L + pos], p.n_per_media, p.T_img, p.only_immediate, p.Lk) : Window{0, p.Lk, 0};
        int rlo = w[mt].hi > w[mt].lo ? w[mt].lo : 0x7fffffff, rhi = w[mt].hi > w[mt].lo ? w[mt].hi : 0;
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            const int olo = of_shfl_xor_i(rlo, m), ohi = of_shfl_xor_i(rhi, m);
            rlo = olo < rlo ? olo : rlo;
            rhi = ohi > rhi ? ohi : rhi;
        }
        kb_lo[mt] = of_uniform(rhi > rlo ? rlo / 64 : 0);
        kb_hi[mt] = of_uniform(rhi > rlo ? (rhi + 63) / 64 : 0);
    }
    const int kb0 = kb_hi[0] > kb_lo[0] ? (kb_hi[1] > kb_lo[1] ? (kb_lo[0] < kb_lo[1] ? kb_lo[0] : kb_lo[1]) : kb_lo[0]) : kb_lo[1];
    const int kb1 = kb_hi[0] > kb_hi[1] ? kb_hi[0] : kb_hi[1];
    const bf16_t* kb_ptr = p.k + (size_t)batch * p.Lk * p.ldk + wave * 64;
    const bf16_t* vb_ptr = p.v + (size_t)batch * p.Lk * p.ldv + wave * 64;
    u32x4 kpre[8];                               // 64 keys x 128 B of K: 8 lanes per row (V follows at the start of the attention phase: both
                                                 // prefetched would not fit next to the four rows of x the LayerNorm holds)
    auto kv_load = [&](const bf16_t* base, long ld, int kb, u32x4 (&dst)[8]) OF_INLINE_LAMBDA {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int id = c * 64 + lane, row = id >> 3, cs = id & 7;
            const long key = (long)kb * 64 + row;
            dst[c] = u32x4{0u, 0u, 0u, 0u};
            if (key < p.Lk) dst[c] = *(const u32x4*)(base + (size_t)key * ld + cs * 8);
        }
    };
    // ------------------------------------------------------------------------------------------------------ 1. LayerNorm (helpers.py:184)
    {
        float v[4][CPL][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const void* xr = of_uniform_ptr((const char*)p.x + (size_t)(row0 + wave * 4 + i) * p.ldx * (p.x_f32 ? 4 : 2));
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (lo + j * 512 < (unsigned)D) x_load8(xr, p.x_f32, lo + j * 512, v[i][j]);
        }
        // L2 warm-up: the packed weights have not been touched since the last optimizer step.  The 32 workgroups of an XCD (block b runs
        // on XCD b % 8: placement only, any other map is merely slower) each pull 1 / 32 of the to_q matrix now, under the HBM-bound
        // LayerNorm, so that the K loop below streams it from this XCD's L2 from its first k-step on; to_out's matrix follows under to_q
        if (kb0 < kb1) kv_load(kb_ptr, p.ldk, kb0, kpre);
        constexpr int NWARM = D / 256;           // 16-byte loads per lane: 1 / 32 of a 512 x D bf16 matrix per workgroup
        const unsigned warm_soff = (((unsigned)of_bid_x() >> 3) & 31u) * (unsigned)(X_INNER * D * 2 / 32);
        unsigned wx = 0;                         // (folded at once: in-order return puts these loads behind x's, which the statistics wait for anyway)
        {
            const of_buf_t bq = of_buf_make(p.wq_pk);
#pragma unroll
            for (int c = 0; c < NWARM; ++c) {
                const u32x4 t = of_buf_load16(bq, (unsigned)tid * 16 + c * 8192, warm_soff);
                wx ^= t[0] ^ t[1] ^ t[2] ^ t[3];
            }
        }
        for (unsigned c = tid * 4; c < (unsigned)D; c += 2048) {
            *(f32x4*)(s_gamma + c) = *(const f32x4*)(p.ln_w + c);
            *(f32x4*)(s_beta + c) = *(const f32x4*)(p.ln_b + c);
        }
        of_sync();
        const float inv_dim = 1.0f / (float)D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = wave * 4 + i;
            const long row = row0 + rl;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (lo + j * 512 < (unsigned)D) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum += v[i][j][e];
                }
            const float mean = of_wave_sum(sum) * inv_dim;
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (lo + j * 512 < (unsigned)D) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sq += (v[i][j][e] - mean) * (v[i][j][e] - mean);
                }
            const float rstd = of_rsqrt(of_wave_sum(sq) * inv_dim + 1e-5f);
            if (lane == 0 && p.stats) {
                p.stats[row * 2] = mean;
                p.stats[row * 2 + 1] = rstd;
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const unsigned eo = lo + j * 512;
                if (eo < (unsigned)D) {
                    const f32x4 w0 = *(const f32x4*)(s_gamma + eo), w1 = *(const f32x4*)(s_gamma + eo + 4);
                    const f32x4 b0 = *(const f32x4*)(s_beta + eo), b1 = *(const f32x4*)(s_beta + eo + 4);
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (v[i][j][e] - mean) * rstd * w0[e] + b0[e];
                        o[4 + e] = (v[i][j][4 + e] - mean) * rstd * w1[e] + b1[e];
                    }
                    *(u32x4*)(u_img + ximg_off(rl, (int)(eo >> 3), UROW)) =
                        u32x4{of_pack_bf16(o[0], o[1]), of_pack_bf16(o[2], o[3]), of_pack_bf16(o[4], o[5]), of_pack_bf16(o[6], o[7])};
                }
            }
        }
        if (wx == 0x9e3779b9u && p.B < 0) p.stats[0] = 0.f;            // (never true: keeps the warm-up loads alive)
    }
    of_sync();                                   // the LN(x) image is complete (and gamma / beta are dead)
    XF_STAMP(1);
    // ------------------------------------------------------------------------------------------------------ 2. to_q (helpers.py:186): head `wave`
    // qacc[mt][t][r] = q[row 16 mt + i16][64 wave + 16 t + 4 g + r]
    f32x4 qacc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) qacc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        constexpr int PFQ = 4;                   // k-steps of weight fragments in flight
        static_assert(KSQ % PFQ == 0, "to_q ring");
        const of_buf_t bq = of_buf_make(p.wq_pk);
        const unsigned vo = (unsigned)lane * 16;
        int uoff[2][4];                          // LN(x) fragment of row tile mt, k-step 4 c + q: + c * 256
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) uoff[mt][q] = (mt * 16 + i16) * UROW + (((4 * q + g) ^ i16) << 4);
        u32x4 wb[PFQ][4];
#pragma unroll
        for (int pp = 0; pp < PFQ; ++pp)
#pragma unroll
            for (int t = 0; t < 4; ++t) wb[pp][t] = of_buf_load16(bq, vo, (unsigned)(((wave * 4 + t) * KSQ + pp) * 1024));
#pragma unroll 1
        for (int ks0 = 0; ks0 < KSQ; ks0 += PFQ) {
#pragma unroll
            for (int pp = 0; pp < PFQ; ++pp) {
                const int ks = ks0 + pp;
                s16x8 ua[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) ua[mt] = *(const s16x8*)(u_img + uoff[mt][pp] + (ks0 >> 2) * 256);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) qacc[mt][t] = of_mfma(__builtin_bit_cast(s16x8, wb[pp][t]), ua[mt], qacc[mt][t]);
                if (ks + PFQ < KSQ) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) wb[pp][t] = of_buf_load16(bq, vo, (unsigned)(((wave * 4 + t) * KSQ + ks + PFQ) * 1024));
                }
            }
        }
    }
    // LN(x) of this wave's four rows, image -> HBM (the to_q weight gradient needs it): stored HERE, behind the K loop's last load -- on
    // gfx950 stores count into vmcnt like loads, so stores issued in front of the loop hold its first counted wait until HBM has taken
    // them (measured: to_q 21.8 -> 31 us) -- and they drain under the attention / to_out phases, which leave HBM idle
    if (p.xn) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = wave * 4 + i;
            bf16_t* xnr = (bf16_t*)of_uniform_ptr(p.xn + (size_t)(row0 + rl) * p.ldxn);
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (lo + j * 512 < (unsigned)D) {
#ifdef OF_SAVED_NT
                    __builtin_nontemporal_store(*(const u32x4*)(u_img + ximg_off(rl, (int)((lo + j * 512) >> 3), UROW)), (u32x4*)(xnr + lo + j * 512));
#else
                    *(u32x4*)(xnr + lo + j * 512) = *(const u32x4*)(u_img + ximg_off(rl, (int)((lo + j * 512) >> 3), UROW));
#endif
                }
        }
    }
    // q as the B operand of S^T = K Q^T, straight from the accumulators' lane layout: k-slot (g, j) of k-step ks' is head column
    // 32 ks' + 16 (j >> 2) + 4 g + (j & 3) -- the K fragments below are read with the same assignment
    s16x8 qf[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int ksp = 0; ksp < 2; ++ksp) qf[mt][ksp] = pack8(qacc[mt][2 * ksp], qacc[mt][2 * ksp + 1]);
        if (p.q) {
            u32x2 qv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) qv[t] = u32x2{of_pack_bf16(qacc[mt][t][0], qacc[mt][t][1]), of_pack_bf16(qacc[mt][t][2], qacc[mt][t][3])};
            store_row_blocks(p.q + (size_t)(row0 + mt * 16 + i16) * p.ldq + wave * 64, qv, g, true);
        }
    }
    of_sync();                                   // every wave has finished reading the LN(x) image: region A becomes the K | V images
    XF_STAMP(2);

    // ------------------------------------------------------------------------------------------------------ 3. windowed attention (helpers.py:192-231)
    {
        // to_out's matrix into this XCD's L2 (see the LayerNorm phase): requested here, not waited for before the end of the phase
        constexpr int NWARM = D / 256;
        u32x4 wv[NWARM];
        {
            const of_buf_t bo = of_buf_make(p.wout_pk);
            const unsigned warm_soff = (((unsigned)of_bid_x() >> 3) & 31u) * (unsigned)(X_INNER * D * 2 / 32);
#pragma unroll
            for (int c = 0; c < NWARM; ++c) wv[c] = of_buf_load16(bo, (unsigned)tid * 16 + c * 8192, warm_soff);
        }
        char* k_img = smem + wave * 16384;       // normal image [64 keys][64] of head `wave`
        char* v_img = k_img + 8192;              // transpose image
        const FragOff<64> fo = make_frag_off<64>(lane);
        RowCtx rc[2];
        TileRange tr[2];
        float m_i[2], l_i[2];
        f32x4 oacc[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            rc[mt] = make_row_ctx(w[mt].lo, w[mt].hi, w[mt].uni, 0, p.scale, 0.f);
            tr[mt] = make_tile_range(w[mt].lo, w[mt].hi);
            m_i[mt] = NEG_BIG;
            l_i[mt] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[mt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // K fragment with the permuted k assignment: the two 8-byte halves (head columns 32 ks' + 4 g .. and 32 ks' + 16 + 4 g ..) of key row i16
        int koff[2][2];
#pragma unroll
        for (int ksp = 0; ksp < 2; ++ksp)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) koff[ksp][hh] = img_n_off<64>(i16, 4 * ksp + 2 * hh + (g >> 1)) + 8 * (g & 1);
        for (int kb = kb0; kb < kb1; ++kb) {
            of_wave_sync();                      // the previous block's fragment reads are done
            u32x4 vpre[8];
            kv_load(vb_ptr, p.ldv, kb, vpre);    // lands while K goes to LDS
            if (kb > kb0) kv_load(kb_ptr, p.ldk, kb, kpre);     // (a tile that straddles media: the first block came in under the LayerNorm)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int id = c * 64 + lane, row = id >> 3, cs = id & 7;
                *(u32x4*)(k_img + img_n_off<64>(row, cs)) = kpre[c];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int id = c * 64 + lane, row = id >> 3, cs = id & 7;
                *(u32x4*)(v_img + img_t_off<64>(row, cs * 8)) = vpre[c];
            }
            of_wave_sync();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if (kb < kb_lo[mt] || kb >= kb_hi[mt]) continue;      // wave-uniform
                f32x4 s[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ksp = 0; ksp < 2; ++ksp) {
                        const u32x2 klo = *(const u32x2*)(k_img + koff[ksp][0] + t * 16 * 128), khi = *(const u32x2*)(k_img + koff[ksp][1] + t * 16 * 128);
                        const u32x4 kf = {klo[0], klo[1], khi[0], khi[1]};
                        s[t] = of_mfma(__builtin_bit_cast(s16x8, kf), qf[mt][ksp], s[t]);
                    }
                }
                const float mb = score_block_any(s, rc[mt], tr[mt], kb * 64, 64, g, false);
                softmax_pv<64, false, 4>(s, mb, v_img, fo, lane, oacc[mt], m_i[mt], l_i[mt]);
            }
        }
        // o = P V / l: to HBM (the to_out weight gradient and the attention backward need it) and into the o image
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float inv = l_i[mt] > 0.f ? 1.0f / l_i[mt] : 0.f;
            const int rl = mt * 16 + i16;
            u32x2 po[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                po[dt] = u32x2{of_pack_bf16(oacc[mt][dt][0] * inv, oacc[mt][dt][1] * inv), of_pack_bf16(oacc[mt][dt][2] * inv, oacc[mt][dt][3] * inv)};
                *(u32x2*)(o_img + ximg_off(rl, 8 * wave + 2 * dt + (g >> 1), OROW) + 8 * (g & 1)) = po[dt];
            }
            if (p.o) store_row_blocks(p.o + (size_t)(row0 + rl) * p.ldo + wave * 64, po, g, true);
            if (p.lse && g == 0)
                p.lse[((size_t)batch * 8 + wave) * p.L + pos0 + rl] = l_i[mt] > 0.f ? (m_i[mt] + of_log2(l_i[mt])) * LN2 : __builtin_inff();
        }
        unsigned wx = 0;
#pragma unroll
        for (int c = 0; c < NWARM; ++c) wx ^= wv[c][0] ^ wv[c][1] ^ wv[c][2] ^ wv[c][3];
        if (wx == 0x9e3779b9u && p.B < 0) p.lse[0] = 0.f;              // (never true: keeps the warm-up loads alive)
    }
    of_sync();                                   // the o image is complete; the K | V images are dead
    XF_STAMP(3);

    // ------------------------------------------------------------------------------------------------------ 4. to_out (helpers.py:233)
    // yacc[mt][t][r] = y[row 16 mt + i16][n0 + 16 t + 4 g + r], n0 = wave * D / 8
    const int n0 = wave * (D / 8);
    f32x4 yacc[2][NTO];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < NTO; ++t) yacc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    xf_rows32_times_packed<NTO>(p.wout_pk, o_img, wave, lane, yacc);

    XF_STAMP(4);
    // ------------------------------------------------------------------------------------------------------ 5. gate + residual (helpers.py:267-276), LN_ff
    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    float rsum[2] = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long row = row0 + mt * 16 + i16;
        if (p.x_f32) {
            const float* xr = (const float*)p.x + (size_t)row * p.ldx + n0 + 4 * g;
            float* yr = (float*)p.y + (size_t)row * p.ldy + n0 + 4 * g;
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                const f32x4 xv = *(const f32x4*)(xr + 16 * t);
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = xv[r] + gv * yacc[mt][t][r];
                yacc[mt][t] = y;
                *(f32x4*)(yr + 16 * t) = y;
                rsum[mt] += (y[0] + y[1]) + (y[2] + y[3]);
            }
        } else {
            const bf16_t* xr = (const bf16_t*)p.x + (size_t)row * p.ldx + n0 + 4 * g;
            u32x2 yv[NTO];
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                float xv[4];
                const u32x2 xb = *(const u32x2*)(xr + 16 * t);
                xv[0] = of_bf16_to_f32((bf16_t)(xb[0] & 0xffff));
                xv[1] = of_bf16_to_f32((bf16_t)(xb[0] >> 16));
                xv[2] = of_bf16_to_f32((bf16_t)(xb[1] & 0xffff));
                xv[3] = of_bf16_to_f32((bf16_t)(xb[1] >> 16));
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = xv[r] + gv * yacc[mt][t][r];
                yv[t] = u32x2{of_pack_bf16(y[0], y[1]), of_pack_bf16(y[2], y[3])};
                // the bf16 stream's LayerNorm sees the ROUNDED residual stream, like the unfused path (it reads y1 back from HBM)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = of_bf16_to_f32((bf16_t)(r & 1 ? yv[t][r >> 1] >> 16 : yv[t][r >> 1] & 0xffff));
                yacc[mt][t] = y;
                rsum[mt] += (y[0] + y[1]) + (y[2] + y[3]);
            }
            store_row_blocks((bf16_t*)p.y + (size_t)row * p.ldy + n0, yv, g, true);
        }
    }
    if (!p.ln2_w) {                              // (kernel-uniform) no LayerNorm behind the branch
        XF_STAMP(5);
        XF_STAMP_FLUSH();
        return;
    }
    XF_STAMP(7);
    // row statistics: a row's D values sit in 4 lanes (g) of each of the 8 waves
    float* red = (float*)smem;                   // [2][8 waves][32 rows]
    float mean[2], rstd[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const float s = of_rows_sum(rsum[mt]);
        if (g == 0) red[wave * 32 + mt * 16 + i16] = s;
    }
    of_sync();
    const float inv_dim = 1.0f / (float)D;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) s += red[wv * 32 + mt * 16 + i16];
        mean[mt] = s * inv_dim;
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq += (yacc[mt][t][r] - mean[mt]) * (yacc[mt][t][r] - mean[mt]);
        sq = of_rows_sum(sq);
        if (g == 0) red[256 + wave * 32 + mt * 16 + i16] = sq;
    }
    of_sync();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) s += red[256 + wv * 32 + mt * 16 + i16];
        rstd[mt] = of_rsqrt(s * inv_dim + 1e-5f);
        const long row = row0 + mt * 16 + i16;
        if (wave == 0 && g == 0 && p.stats2) {
            p.stats2[row * 2] = mean[mt];
            p.stats2[row * 2 + 1] = rstd[mt];
        }
        u32x2 uv[NTO];
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            const f32x4 w4 = *(const f32x4*)(p.ln2_w + n0 + 16 * t + 4 * g), b4 = *(const f32x4*)(p.ln2_b + n0 + 16 * t + 4 * g);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (yacc[mt][t][r] - mean[mt]) * rstd[mt] * w4[r] + b4[r];
            uv[t] = u32x2{of_pack_bf16(o[0], o[1]), of_pack_bf16(o[2], o[3])};
        }
        store_row_blocks(p.u2 + (size_t)row * p.ldu2 + n0, uv, g, true);
    }
    XF_STAMP(5);
    XF_STAMP_FLUSH();
}

int check(const OfXattnFusedArgs& a) {
    if (!a.x || !a.ln_w || !a.ln_b || !a.wq_pk || !a.k || !a.v || !a.wout_pk || !a.y) return OF_E_ARG;
    if (a.B <= 0 || a.L <= 0 || a.Lk <= 0 || a.d <= 0) return OF_E_ARG;
    if (a.heads != 8 || a.head_dim != 64) return OF_E_SHAPE;
    if (a.d != 256 && a.d != 512 && a.d != 1024 && a.d != 2048) return OF_E_SHAPE;
    if (a.L % XR) return OF_E_SHAPE;
    if (a.text_time && (a.n_per_media <= 0 || a.T_img <= 0)) return OF_E_ARG;
    if (a.ln2_w && (!a.ln2_b || !a.u2)) return OF_E_ARG;
    if ((a.ldx & 7) || (a.ldy & 7) || (a.ldk & 7) || (a.ldv & 7)) return OF_E_ALIGN;
    if ((a.xn && (a.ldxn & 7)) || (a.q && (a.ldq & 7)) || (a.o && (a.ldo & 7)) || (a.ln2_w && (a.ldu2 & 7))) return OF_E_ALIGN;
    const void* ptrs[] = {a.x, a.ln_w, a.ln_b, a.wq_pk, a.k, a.v, a.wout_pk, a.y, a.xn, a.q, a.o, a.u2, a.ln2_w, a.ln2_b};
    for (const void* q : ptrs)
        if ((uintptr_t)q & 15) return OF_E_ALIGN;
    return 0;
}

}  // namespace

extern "C" int of_pack_frag16(const uint16_t* W, int N, int K, long ldw, uint16_t* P, void* stream) {
    if (!W || !P || N <= 0 || K <= 0) return OF_E_ARG;
    if ((N % 16) || (K % 32)) return OF_E_SHAPE;
    if ((ldw & 7) || ((uintptr_t)W & 15) || ((uintptr_t)P & 15)) return OF_E_ALIGN;
    PackArgs a{W, P, N, K, ldw};
    const long total = (long)(N / 16) * (K / 32) * 64;
    return of_launch(of_pack_frag16_kernel, of_dim3{(unsigned)((total + 255) / 256), 1, 1}, 256, 0, (of_stream_t)stream, a);
}

extern "C" int of_pack_frag16_batch(const OfPackDesc* descs, int n, void* stream) {
    if (!descs || n <= 0) return OF_E_ARG;
    for (int first = 0; first < n; first += OF_PACK_BATCH_MAX) {
        PackBatchArgs a{};
        a.n = n - first < OF_PACK_BATCH_MAX ? n - first : OF_PACK_BATCH_MAX;
        long total = 0;
        for (int i = 0; i < a.n; ++i) {
            const OfPackDesc& d = descs[first + i];
            if (!d.W || !d.P || d.N <= 0 || d.K <= 0) return OF_E_ARG;
            if ((d.N % 16) || (d.K % 32)) return OF_E_SHAPE;
            if ((d.ldw & 7) || ((uintptr_t)d.W & 15) || ((uintptr_t)d.P & 15)) return OF_E_ALIGN;
            total += (long)(d.N / 16) * (d.K / 32) * 64;
            a.end[i] = total;
            a.d[i] = d;
        }
        const int rc = of_launch(of_pack_frag16_batch_kernel, of_dim3{(unsigned)((total + 255) / 256), 1, 1}, 256, 0, (of_stream_t)stream, a);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int of_xattn_fused_eligible(const OfXattnFusedArgs* args) { return args && check(*args) == 0; }

extern "C" int of_xattn_fused_fwd(const OfXattnFusedArgs* args, void* stream) {
    if (!args) return OF_E_ARG;
    const int rc = check(*args);
    if (rc) return rc;
    const OfXattnFusedArgs& a = *args;
    const of_dim3 grid{(unsigned)((long)a.B * a.L / XR), 1, 1};
    of_stream_t s = (of_stream_t)stream;
    switch (a.d) {
        case 256: return of_launch(of_xattn_fused_fwd_kernel<256>, grid, 512, X_SMEM, s, a);
        case 512: return of_launch(of_xattn_fused_fwd_kernel<512>, grid, 512, X_SMEM, s, a);
        case 1024: return of_launch(of_xattn_fused_fwd_kernel<1024>, grid, 512, X_SMEM, s, a);
        default: return of_launch(of_xattn_fused_fwd_kernel<2048>, grid, 512, X_SMEM, s, a);
    }
}
