// bf16 MFMA GEMM with fused epilogues for the OpenFlamingo hot path (gfx950).
//
// acc[m][n] = sum_k A(m,k) B(n,k); replaces every nn.Linear matmul (forward, dX, dW) of
// open_flamingo/src/helpers.py:19,21,36-38,154-156.  See include/of_hip.h for the contract.
//
// Structure (v1): 128x128x64 tile, 256 threads = 4 waves in 2x2, each wave a 64x64 sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16 fragments.  Operands are staged global -> VGPR -> LDS (the next K-tile's
// global loads are issued before the current tile's MFMAs and written to the other LDS buffer after
// them: one barrier per K-tile).  Two LDS images exist per operand kind:
//   K-contiguous operand  [128 rows][64 k]  (128 B rows), 16-B slot s of row r stored at slot
//       s ^ ((r>>1)&7): a ds_read_b128 fragment read (16 rows x 4 k-slots per wave) is conflict-free.
//   K-strided operand     [64 k][128 cols]  (256 B rows), 32-B chunk c of k-row r stored at chunk
//       c ^ ((r&3) | ((r>>3)&1)<<2): the ds_read_b64_tr_b16 fragment read (8 k-rows x 32 B per half-wave)
//       is conflict-free; the transpose itself is done by the LDS transpose-read, so dX = dY W and
//       dW = dY^T X need no transposed copies of activations or weights in HBM.
// The MFMA is issued with the operands swapped (D = Bfrag x Afrag) so that a lane ends up with four
// consecutive n of one output row: epilogue loads/stores are 8-byte (bf16) or 16-byte (fp32) vectors.
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;      // BN: the wide tile (of_gemm_kernel's NTW = 4); the narrow one is 64
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB for either image
constexpr int SMEM_BYTES = 4 * TILE_BYTES;

OF_DEV int swz_n(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
OF_DEV int swz_t_f(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
OF_DEV int swz_t(int krow, int col) {  // byte offset of element (krow, col) in the K-strided image
    return krow * 256 + ((((col >> 4)) ^ swz_t_f(krow)) << 5) + ((col & 15) << 1);
}

// ---- global -> registers (4 x 16 B per thread per operand), zero-filled out of range
// EXT = tile extent of the operand (rows of a K-contiguous operand / columns of a K-strided one): 128, or 64 for the B
// operand of the 128 x 64 tile -- the images keep their 128-wide pitch, the narrow tile fills half of them
template <bool TR, int EXT>
OF_DEV void g2r(const bf16_t* __restrict__ base, int ld, int row0, int rows, int k0, int K, int tid, u32x4 (&r)[4]) {
#pragma unroll
    for (int c = 0; c < EXT / 32; ++c) {
        int id = c * 256 + tid;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!TR) {
            int row = id >> 3, slot = id & 7;
            int gr = row0 + row, gk = k0 + slot * 8;
            if (gr < rows && gk < K) v = *(const u32x4*)(base + (size_t)gr * ld + gk);
        } else {
            int krow = id / (EXT / 8), cs = id % (EXT / 8);
            int gk = k0 + krow, gc = row0 + cs * 8;
            if (gk < K && gc < rows) v = *(const u32x4*)(base + (size_t)gk * ld + gc);
        }
        r[c] = v;
    }
}
template <bool TR, int EXT>
OF_DEV void r2s(char* tile, int tid, const u32x4 (&r)[4]) {
#pragma unroll
    for (int c = 0; c < EXT / 32; ++c) {
        int id = c * 256 + tid;
        int off;
        if (!TR) {
            off = swz_n(id >> 3, id & 7);
        } else {
            int krow = id / (EXT / 8), cs = id % (EXT / 8);
            off = swz_t(krow, cs * 8);
        }
        *(u32x4*)(tile + off) = r[c];
    }
}
// ---- LDS -> MFMA fragment for 16 rows (row_base..+15) and the 32-wide k-step kk
template <bool TR, bool SAFE>
OF_DEV s16x8 frag(const char* tile, int row_base, int kk, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!TR) {
        return *(const s16x8*)(tile + swz_n(row_base + i, kk * 4 + g));
    } else if (!SAFE) {
        s16x8 f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int krow = kk * 32 + g * 8 + h * 4 + (i >> 2);
            s16x4 t = of_lds_tr(tile + swz_t(krow, row_base + (i & 3) * 4));
            f[h * 4 + 0] = t[0];
            f[h * 4 + 1] = t[1];
            f[h * 4 + 2] = t[2];
            f[h * 4 + 3] = t[3];
        }
        return f;
    } else {
        s16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = *(const short*)(tile + swz_t(kk * 32 + g * 8 + e, row_base + i));
        return f;
    }
}

// NTW = 16-column fragments per wave along N: 4 -> the 128 x 128 tile, 2 -> a 128 x 64 tile.  The narrow tile is for launches
// whose 128 x 128 grid would leave at most one 4-wave workgroup per CU (to_q, the dX of to_out, the Perceiver's 1024-wide
// projections: 128-256 tiles): a k-tile there costs its LDS write -> barrier -> read -> MFMA chain with nothing else resident
// to cover it; twice the workgroups overlap each other.
template <bool AT, bool BT, int EPI, bool SAFE, int NTW = 4>
OF_GLOBAL void OF_BOUNDS(256, 2) of_gemm_kernel(OfGemmArgs p) {
    constexpr int BN = NTW * 32;
    char* smem = of_smem();
    const int tid = of_tid(), lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int pm, pn;
    ofg::tile_coords(of_bid_x(), of_gdim_x(), tiles_m, tiles_n, pm, pn);
    const int m0 = pm * BM, n0 = pn * BN;

    f32x4 acc[4][NTW];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[4], rb[4];
    // split-K (p.ksplit > 1, OF_EPI_ACC_F32 only): slice of_bid_y() owns k-tiles [kt0, kt0 + nk) and adds its partial
    // sums into C with fp32 atomics (the launcher has zeroed C when beta == 0)
    const int nk_all = (p.K + BK - 1) / BK;
    const int per = (nk_all + p.ksplit - 1) / p.ksplit;
    const int kt0 = of_bid_y() * per;
    const int nk = (nk_all - kt0 < per ? nk_all - kt0 : per);
    if (nk <= 0) return;
    g2r<AT, BM>(p.A, p.lda, m0, p.M, kt0 * BK, p.K, tid, ra);
    g2r<BT, BN>(p.B, p.ldb, n0, p.N, kt0 * BK, p.K, tid, rb);
    r2s<AT, BM>(smem, tid, ra);
    r2s<BT, BN>(smem + TILE_BYTES, tid, rb);
    of_sync();
    for (int kt = 0; kt < nk; ++kt) {
        const char* ta = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* tb = ta + TILE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) {
            g2r<AT, BM>(p.A, p.lda, m0, p.M, (kt0 + kt + 1) * BK, p.K, tid, ra);
            g2r<BT, BN>(p.B, p.ldb, n0, p.N, (kt0 + kt + 1) * BK, p.K, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            s16x8 fa[4], fb[NTW];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = frag<AT, SAFE>(ta, wr * 64 + t * 16, kk, lane);
#pragma unroll
            for (int t = 0; t < NTW; ++t) fb[t] = frag<BT, SAFE>(tb, wc * (NTW * 16) + t * 16, kk, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = of_mfma(fb[nt], fa[mt], acc[mt][nt]);
        }
        if (more) {
            char* na = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            r2s<AT, BM>(na, tid, ra);
            r2s<BT, BN>(na + TILE_BYTES, tid, rb);
        }
        of_sync();
    }

    // ---------------------------------------------------------------- epilogue
    float gv = 1.0f;
    if (p.gate) gv = of_tanh(*p.gate);
    const float sc = gv * p.alpha;
    float dot = 0.f;
    if (EPI == OF_EPI_ACC_F32 && p.ksplit > 1) {
        float* slab = p.workspace ? (float*)p.workspace + (size_t)of_bid_y() * p.M * p.N : nullptr;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int m = m0 + wr * 64 + mt * 16 + i16, n = n0 + wc * (NTW * 16) + nt * 16 + g * 4;
                if (m < p.M && n < p.N) {
                    if (slab) {   // this slice's partial tile, combined by of_splitk_reduce_kernel in slice order
                        *(f32x4*)(slab + (size_t)m * p.N + n) =
                            f32x4{sc * acc[mt][nt][0], sc * acc[mt][nt][1], sc * acc[mt][nt][2], sc * acc[mt][nt][3]};
                    } else {
                        float* c = (float*)p.C + (size_t)m * p.ldc + n;
#pragma unroll
                        for (int e = 0; e < 4; ++e) of_atomic_add(c + e, sc * acc[mt][nt][e]);
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            ofg::epilogue_frag<EPI>(p, acc[mt][nt], m0 + wr * 64 + mt * 16 + i16, n0 + wc * (NTW * 16) + nt * 16 + g * 4, gv, sc, dot);
    ofg::epilogue_finish<EPI>(p, dot, lane, wave, 4, (float*)smem, of_bid_x());   // the K loop ended with a workgroup barrier
}

// C = beta * C + sum over slices of slab[s]  (fixed slice order: deterministic)
OF_GLOBAL void of_splitk_reduce_kernel(OfGemmArgs p) {
    const long nv = ((long)p.M * p.N) >> 2;
    const long stride = (long)of_gdim_x() * 256;
    const float* ws = (const float*)p.workspace;
    for (long i = (long)of_bid_x() * 256 + of_tid(); i < nv; i += stride) {
        const long e = i * 4, m = e / p.N, n = e - m * p.N;
        f32x4 s = *(const f32x4*)(ws + e);
        for (int k = 1; k < p.ksplit; ++k) {
            const f32x4 t = *(const f32x4*)(ws + (size_t)k * p.M * p.N + e);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        float* c = (float*)p.C + (size_t)m * p.ldc + n;
        if (p.beta != 0.f) {
            const f32x4 o = *(const f32x4*)c;
            s[0] += p.beta * o[0]; s[1] += p.beta * o[1]; s[2] += p.beta * o[2]; s[3] += p.beta * o[3];
        }
        *(f32x4*)c = s;
    }
}

// the same for the problems of an of_gemm_batch launch in ONE grid: workgroups [wg_end[i-1], wg_end[i]) combine problem i's slabs
OF_GLOBAL void of_splitk_reduce_batch_kernel(OfGemmBatchArgs m) {
    const int bid = of_bid_x();
    int i = 0, first = 0;
#pragma unroll
    for (int j = 0; j + 1 < OF_GEMM_BATCH_MAX; ++j)
        if (j + 1 < m.n && bid >= m.wg_end[j]) {
            i = j + 1;
            first = m.wg_end[j];
        }
    OfGemmArgs p = m.a[0];
    if (i == 1) p = m.a[1];
    if (i == 2) p = m.a[2];
    if (i == 3) p = m.a[3];
    const int nblk = m.wg_end[i] - first;
    const long nv = ((long)p.M * p.N) >> 2;
    const long stride = (long)nblk * 256;
    const float* ws = (const float*)p.workspace;
    for (long e4 = (long)(bid - first) * 256 + of_tid(); e4 < nv; e4 += stride) {
        const long e = e4 * 4, mm = e / p.N, nn = e - mm * p.N;
        f32x4 s = *(const f32x4*)(ws + e);
        for (int k = 1; k < p.ksplit; ++k) {
            const f32x4 t = *(const f32x4*)(ws + (size_t)k * p.M * p.N + e);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        float* c = (float*)p.C + (size_t)mm * p.ldc + nn;
        if (p.beta != 0.f) {
            const f32x4 o = *(const f32x4*)c;
            s[0] += p.beta * o[0]; s[1] += p.beta * o[1]; s[2] += p.beta * o[2]; s[3] += p.beta * o[3];
        }
        *(f32x4*)c = s;
    }
}

// *dot_out += (1 - tanh(gate)^2) * (partials[0] + ... + partials[n-1]), one workgroup, fixed order (ofg::epilogue_finish)
struct OfDotFinishArgs {
    const float* partials;
    int n;
    const float* gate;
    float* dot_out;
};
OF_GLOBAL void of_dot_finish_kernel(OfDotFinishArgs a) {
    float* red = (float*)of_smem();
    const int tid = of_tid();
    float s = 0.f;
    for (int i = tid; i < a.n; i += 256) s += a.partials[i];
    red[tid] = s;
    of_sync();
    for (int w = 128; w >= 1; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        of_sync();
    }
    if (tid == 0) {
        float gv = 1.0f;
        if (a.gate) gv = of_tanh(*a.gate);
        *a.dot_out += (1.0f - gv * gv) * red[0];
    }
}

template <bool AT, bool BT, int EPI>
int launch_layout(const OfGemmArgs& a, of_dim3 grid, of_stream_t s, bool narrow) {
    int rc;
    if (a.safe == 1 && (AT || BT)) rc = of_launch(of_gemm_kernel<AT, BT, EPI, true>, grid, 256, SMEM_BYTES, s, a);
    else if (narrow) rc = of_launch(of_gemm_kernel<AT, BT, EPI, false, 2>, grid, 256, SMEM_BYTES, s, a);
    else rc = of_launch(of_gemm_kernel<AT, BT, EPI, false>, grid, 256, SMEM_BYTES, s, a);
    if (rc || !of_gemm_has_dot(a)) return rc;
    return of_gemm_dot_finish(a, (int)(grid.x * grid.y), s);
}
// Only the (layout, epilogue) pairs the hot path uses are instantiated (see DESIGN.md kernel table).
int dispatch(const OfGemmArgs& a, of_dim3 grid, of_stream_t s, bool narrow = false) {
    const int layout = a.a_trans * 2 + a.b_trans;
    if (layout == 0) {  // y = x W^T
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_layout<false, false, OF_EPI_STORE_BF16>(a, grid, s, narrow);
            case OF_EPI_GELU: return launch_layout<false, false, OF_EPI_GELU>(a, grid, s, narrow);
            case OF_EPI_GATE_RESID: return launch_layout<false, false, OF_EPI_GATE_RESID>(a, grid, s, narrow);
            case OF_EPI_ACC_F32: return launch_layout<false, false, OF_EPI_ACC_F32>(a, grid, s, narrow);
        }
    } else if (layout == 1) {  // dX = dY W
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_layout<false, true, OF_EPI_STORE_BF16>(a, grid, s, narrow);
            case OF_EPI_DGELU_DOT: return launch_layout<false, true, OF_EPI_DGELU_DOT>(a, grid, s, narrow);
            case OF_EPI_SCALE_DOT: return launch_layout<false, true, OF_EPI_SCALE_DOT>(a, grid, s, narrow);
            case OF_EPI_ACC_F32: return launch_layout<false, true, OF_EPI_ACC_F32>(a, grid, s, narrow);
        }
    } else if (layout == 3) {  // dW = dY^T X
        switch (a.epi) {
            case OF_EPI_STORE_BF16: return launch_layout<true, true, OF_EPI_STORE_BF16>(a, grid, s, narrow);
            case OF_EPI_ACC_F32: return launch_layout<true, true, OF_EPI_ACC_F32>(a, grid, s, narrow);
        }
    }
    return OF_E_SHAPE;
}

}  // namespace

namespace {
// K slices for a weight-gradient GEMM with a small output and a deep K (see of_gemm); 1 = not split
int pick_ksplit(const OfGemmArgs& a, bool with_workspace) {
    const long tiles256 = (long)(a.M / 256) * (a.N / 256);
    const bool pp_ok = !(a.M % 256) && !(a.N % 256) && !(a.K % 64) && tiles256 >= 128;
    const bool forced = a.safe >= 8 && a.safe < 16;          // 8 + log2(split): tuning aid (tools/bench_splitk.py)
    if (!(forced || (a.safe == 0 && !pp_ok)) || a.epi != OF_EPI_ACC_F32) return 1;
    if (!with_workspace && !(a.beta == 1.f || (a.beta == 0.f && a.ldc == a.N))) return 1;
    const long t128 = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    int split = 1;
    // with slabs a slice costs one coalesced fp32 store + read of the tile (cheap): go to two workgroups per CU;
    // with atomics (~4 M fp32 atomics per 1 M outputs and slice) stop at one
    // (the 8-wave LDS-DMA kernel that takes tile-aligned slab launches holds one workgroup per CU: stop at 256 there)
    const bool mid = !(a.M % 128) && !(a.N % 128) && !(a.K % 64) && a.safe == 0;
    const long target = with_workspace ? (mid ? 256 : 512) : 256;
    while (split < 16 && t128 * split < target && a.K / (split * 2) >= 512) split *= 2;
    if (forced) split = 1 << (a.safe - 8);
    return split;
}
}  // namespace

int of_gemm_dot_finish(const OfGemmArgs& a, int nslots, of_stream_t s) {
    OfDotFinishArgs f{(const float*)a.workspace, nslots, a.gate, a.dot_out};
    return of_launch(of_dot_finish_kernel, of_dim3{1, 1, 1}, 256, 256 * sizeof(float), s, f);
}
static size_t dot_slots(const OfGemmArgs& a) { return of_gemm_dot_slots(a); }

// Stream-K scheduling of a big-tile launch (gemm_w4m.hip): the workgroup count, or 0 for the classic one-tile-per-workgroup launch.
// G = the CUs the caller says it may count on (cu_limit, whole groups of 8; default all); stream-K whenever that does not divide
// the tile count -- OF-4B's 400-tile weight gradients (1.56 rounds -> every workgroup 1.5625 tiles), OF-9B's 128-tile launches
// (half a tile per workgroup instead of half the chip idle), any launch next to a collective that holds CUs.
static int sk_grid_for(const OfGemmArgs& a, long tiles256) {
    int G = a.cu_limit > 0 ? (a.cu_limit & ~7) : OF_NUM_CUS;
    if (G > OF_NUM_CUS) G = OF_NUM_CUS;
    if (G < 8) G = 8;
    if (a.safe == 17) return G;                 // forced (tests): persistent even when G divides the tile count
    // Sharing a tile costs its workgroups one 256-KiB partial tile written through to memory and read back (~20-30 us per
    // workgroup): worth it against a K loop of >= 64 stages per tile, not against K = 512 / 1024 (measured: 8192 x 2560 x 512,
    // 320 tiles, 47 -> 83 us; profiles/r04g_gemm_ab_stream_k_v2_OF-4B.jsonl) -- those keep the N-split / partial last round.
    if (a.K < 4096) return 0;
    // More tiles than workgroups and a partial last round (OF-4B's 320- and 400-tile launches): stream-K measured +2..+7 % SLOWER
    // than the N-split / partial round (the fix-up's exposed load latency eats the balance gain) -- it is used where it wins:
    // fewer tiles than workgroups (OF-9B's 128-tile launches: 300 -> 216 us, 296 -> 204 us), and under an explicit cu_limit.
    if (a.cu_limit > 0) return tiles256 % G ? G : 0;
    return tiles256 < G ? G : 0;
}
static bool sk_usable(const OfGemmArgs& a, int G) {
    if (G <= 0 || a.group_kind || !of_gemm_w4m_eligible(a)) return false;
    const size_t need = of_gemm_w4m_sk_bytes(a, G);
    return !need || (a.workspace && a.workspace_bytes >= need && !((uintptr_t)a.workspace & 15));
}

extern "C" size_t of_gemm_workspace_bytes(const OfGemmArgs* args) {
    if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
    const OfGemmArgs& a = *args;
    size_t need = of_gemm_has_dot(a) ? dot_slots(a) * sizeof(float) : 0;
    if (of_gemm_is_skinny(a)) return need;
    const long tiles256 = (long)(a.M / 256) * (a.N / 256);
    const bool big = !(a.M % 256) && !(a.N % 256) && !(a.K % 64) && !a.group_kind && ((a.safe == 0 && tiles256 >= 128) || a.safe == 17);
    if (big && of_gemm_w4m_eligible(a)) {       // stream-K partial tiles + flags (behind the *_DOT partials)
        const size_t sk = of_gemm_w4m_sk_bytes(a, sk_grid_for(a, tiles256));
        if (sk > need) need = sk;
    }
    if (of_gemm_has_dot(a)) return need;
    const int split = pick_ksplit(a, true);
    const size_t slabs = split > 1 ? (size_t)split * a.M * a.N * sizeof(float) : 0;
    return slabs > need ? slabs : need;
}

// Mirror of of_gemm's selection for OfGemmArgs.sumsq_out: the launch is ONE launch of the 256x256 kernel (classic or stream-K) --
// not the skinny kernel, not grouped, not split along K, not split along N (tile quantisation), no forced kernel.
extern "C" size_t of_gemm_sumsq_slots(const OfGemmArgs* args) {
    if (!args || !args->A || !args->B || !args->C || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
    OfGemmArgs a = *args;
    a.sk_grid = 0;
    if (a.epi != OF_EPI_ACC_F32 || a.group_kind || a.safe != 0 || of_gemm_is_skinny(a)) return 0;
    const long tiles256 = (long)(a.M / 256) * (a.N / 256);
    const bool pp_ok = !(a.M % 256) && !(a.N % 256) && !(a.K % 64) && tiles256 >= 128;
    if (!pp_ok || !of_gemm_w4m_eligible(a) || pick_ksplit(a, true) > 1 || pick_ksplit(a, false) > 1) return 0;
    const int G = sk_grid_for(a, tiles256);
    if (G > 0 && sk_usable(a, G)) return (size_t)tiles256;
    const int tm = a.M / 256, tn = a.N / 256;
    if (tiles256 > 256 && tiles256 % 256)
        for (int n = tn - 1; n >= 1; --n)
            if (((long)tm * n) % 256 == 0) {
                const bool mid_ok = !(a.M % 128) && !(a.N % 128) && !(a.K % 64);
                if ((long)tm * (tn - n) <= 128 && mid_ok) return 0;      // (of_gemm would try the N-split: two launches)
                break;
            }
    return (size_t)tiles256;
}

// grouped-B launches (OfGemmArgs.group_kind): big-tile kernels only
static int gemm_grouped(const OfGemmArgs& a_in, of_stream_t s) {
    OfGemmArgs a_own = a_in;
    a_own.sumsq_out = nullptr;            // (single-matrix weight gradients only: of_gemm_sumsq_slots)
    const OfGemmArgs& a = a_own;
    if (!a.groups || a.group_extent <= 0) return OF_E_ARG;
    if ((a.M % 256) || (a.N % 256) || (a.K % 64) || a.a_trans) return OF_E_SHAPE;
    if ((a.lda & 7) || (a.ldb & 7) || (a.ldc & 3) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.C & 15)) return OF_E_ALIGN;
    if (a.group_kind == 1) {              // y[:, gE:(g+1)E] = x W_g^T
        if (a.b_trans || (a.group_extent % 256) || (a.N % a.group_extent)) return OF_E_SHAPE;
        OfGemmArgs w = a;
        w.sk_grid = 0;                    // internal field: a caller's value is ignored
        return of_gemm_w4m_try(w, s);
    }
    if (a.group_kind == 2) {              // dX = sum_g dY[:, gE:(g+1)E] W_g
        if (!a.b_trans || (a.group_extent % 64) || (a.K % a.group_extent)) return OF_E_SHAPE;
        return of_gemm_pp_try(a, s);
    }
    return OF_E_ARG;
}

extern "C" int of_gemm(const OfGemmArgs* args, void* stream) {
    if (!args || !args->A || !args->C) return OF_E_ARG;
    if (args->group_kind) return gemm_grouped(*args, (of_stream_t)stream);
    if (!args->B) return OF_E_ARG;
#if !defined(OF_TOOLS_BUILD) && !defined(OF_HOST_EMU)
    // The product library selects its kernels itself; the only alternative it keeps is the checked path (1: the general kernel with
    // scalar-LDS transposed fragments).  The kernel-forcing selectors the test suite compares kernels with (2 ... 19) exist in
    // tools/libofhip_tools.so (-DOF_TOOLS_BUILD) and in the host emulator build only.
    if (args->safe != 0 && args->safe != 1) return OF_E_ARG;
#endif
    OfGemmArgs a_own = *args;
    a_own.sk_grid = 0;                 // internal field ("callers pass 0"): only the stream-K branches below set it, a caller's value is ignored
    float* const sumsq = a_own.sumsq_out;
    a_own.sumsq_out = nullptr;         // honoured by the single big-tile launch only (of_gemm_sumsq_slots): every other kernel must not see it
    const OfGemmArgs& a = a_own;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return OF_E_ARG;
    // vector-loaded (contiguous) extents must be multiples of 8 elements; outputs are written 4 wide
    const int a_vec = a.a_trans ? a.M : a.K, b_vec = a.b_trans ? a.N : a.K;
    if ((a_vec & 7) || (b_vec & 7) || (a.N & 3)) return OF_E_SHAPE;
    if ((a.lda & 7) || (a.ldb & 7) || (a.ldc & 3)) return OF_E_ALIGN;
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) return OF_E_ALIGN;
    if ((a.epi == OF_EPI_GATE_RESID || a.epi == OF_EPI_DGELU_DOT || a.epi == OF_EPI_SCALE_DOT) &&
        (!a.aux || (a.ldaux & 3) || ((uintptr_t)a.aux & 15)))
        return OF_E_ARG;
    if (of_gemm_has_dot(a) && (!a.workspace || a.workspace_bytes < dot_slots(a) * sizeof(float) || ((uintptr_t)a.workspace & 3)))
        return OF_E_WORKSPACE;      // the gate gradient is reduced from per-workgroup partials: no atomic fallback
    of_stream_t s = (of_stream_t)stream;
    if (of_gemm_is_skinny(a)) {                    // a handful of rows (decode step): stream the weights, no tiles
        const int rc = of_gemm_skinny_try(a, s);
        if (rc != OF_E_SHAPE) return rc;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    of_dim3 grid{(unsigned)(tiles_m * tiles_n), 1, 1};
    OfGemmArgs b = a;
    b.ksplit = 1;
    // Kernel selection (safe == 0).  The 256x256 kernels need >= half of the 256 CUs' worth of tiles to pay (OF-9B's B*L =
    // 2048-row launches, 128 tiles with K = 16384, run 9-14 % faster on half the chip with the big tile than on all of it with
    // 128x128 tiles -- twice the operand bytes per FLOP; same-box A/B profiles/r03b_gemm_ab_OF-9B.jsonl); below that the
    // 128x128 kernels give 4x the workgroups, and weight-gradient GEMMs with a small output and a deep K (K = tokens) are
    // additionally split along K until every CU has work.
    const long tiles256 = (long)(a.M / 256) * (a.N / 256);
    const bool pp_ok = !(a.M % 256) && !(a.N % 256) && !(a.K % 64) && tiles256 >= 128;
    const bool mid_ok = !(a.M % 128) && !(a.N % 128) && !(a.K % 64);      // the 8-wave LDS-DMA 128x128 kernel (gemm_mid.hip)
    {
        int split = pick_ksplit(a, true);
        const bool slabs = split > 1 && a.workspace && a.workspace_bytes >= (size_t)split * a.M * a.N * sizeof(float) &&
                           !((uintptr_t)a.workspace & 15);
        if (!slabs) split = pick_ksplit(a, false);
        if (split > 1) {
            b.ksplit = split;
            grid.y = (unsigned)split;
            if (slabs) {
                int rc = OF_E_SHAPE;
                if (mid_ok && a.safe == 0) rc = of_gemm_mid_try(b, s);
                if (rc == OF_E_SHAPE) rc = dispatch(b, grid, s);
                if (rc) return rc;
                long blocks = (((long)a.M * a.N >> 2) + 255) / 256;
                if (blocks > 2048) blocks = 2048;
                return of_launch(of_splitk_reduce_kernel, of_dim3{(unsigned)blocks, 1, 1}, 256, 0, s, b);
            }
            b.workspace = nullptr;
            if (a.beta == 0.f) {
                const int rc = of_memset_async(a.C, 0, (size_t)a.M * a.N * sizeof(float), s);
                if (rc) return rc;
            }
            return dispatch(b, grid, s);
        }
    }
#if defined(OF_TOOLS_BUILD) || defined(OF_HOST_EMU)
    // the two big-tile kernels of rounds 3 / 5 that of_gemm never selects (gemm_w4.hip: 32x32x16 MFMAs; gemm_w4s.hip: wave-specialised):
    // built into the tools / emulator libraries only, with their tests (records: profiles/r03*, r05*)
    if (a.safe == 6 || a.safe == 7) {              // force the 4-wave 128x128-per-wave kernel (6: register staged, 7: LDS-DMA)
        const int rc = of_gemm_w4_try(a, s);
        if (rc != OF_E_SHAPE) return rc;
    }
    if (a.safe == 19) {                            // force the persistent wave-specialised 256x128 kernel (gemm_w4s.hip)
        const int rc = of_gemm_w4s_try(a, s);
        if (rc != OF_E_SHAPE) return rc;
    }
#endif
    if (a.safe == 16) {                            // force the 4-wave kernel on 16x16x32 MFMAs (gemm_w4m.hip)
        const int rc = of_gemm_w4m_try(a, s);
        if (rc != OF_E_SHAPE) return rc;
    }
    if (a.safe == 17) {                            // force the stream-K schedule of that kernel (persistent workgroups; tests)
        OfGemmArgs w = a;
        w.sk_grid = sk_grid_for(a, (long)(a.M / 256) * (a.N / 256));
        const int rc = of_gemm_w4m_try(w, s);
        if (rc != OF_E_SHAPE) return rc;
    }
    if (a.safe == 18) {                            // force the two-workgroups-per-CU 256x128 kernel (gemm_w4h.hip)
        const int rc = of_gemm_w4h_try(a, s);
        if (rc != OF_E_SHAPE) return rc;
    }
    if (a.safe >= 20) return OF_E_ARG;
    const bool pp_forced = a.safe == 4;
    // Big-tile selection (measured on MI355X, random operands, same box): every layout -> the 4-wave LDS-DMA kernel ON 16x16x32
    // MFMAs (gemm_w4m.hip).  With every CU busy the K loop is bound by the chip's power budget and the 16x16x32 shape spends
    // less energy per FLOP than 32x32x16: -7.4..-7.9 % on the K = 8192 launches, -2..-4 % on the K = 2048 ones
    // (profiles/r03r_gemm_ab_w4m_OF-3B.jsonl).  The same kernel on 32x32x16 stays as safe = 7 (6: register
    // staged), the 8-wave ping-pong kernel keeps the K-grouped B launches (gemm_grouped) and safe = 4.  History: round 2 sent
    // layouts with a K-strided operand to the ping-pong kernel because the 4-wave DMA schedule lost 15-25 % there -- that was
    // hipcc draining the DMA ring in front of every transposed-fragment read (of_platform.h).
    if (a.safe == 0 && pp_ok) {
        // *_DOT epilogues (aux tile + erf-GELU derivative + gate-gradient dot: 10-17 us of VALU / LDS issue per 256x256 tile behind a
        // K loop of 32 stages) over >= 4 rounds of big tiles: two workgroups per CU on 256x128 tiles (gemm_w4h.hip) -- one runs its
        // epilogue under the other one's K loop.  Same box, interleaved, four boxes: NN 8192 x 8192 x 2048 DGELU_DOT -2.4..-3.6 %,
        // SCALE_DOT -3.8..-4.8 %, bit-identical outputs; the plain-store and GELU launches do NOT gain (a K loop alone on a CU runs at
        // 68 % of the MFMA rate there, and the erf-GELU epilogue takes the K loop's issue slots) and stay on the 256x256 kernel
        // (profiles/r05a..d_w4h_probe.jsonl).  K = 4096 (OF-9B): +-2 % -> the 256x256 kernel (profiles/r05e_w4h_family_probe.jsonl).
        if ((a.epi == OF_EPI_DGELU_DOT || a.epi == OF_EPI_SCALE_DOT) && tiles256 >= 1024 && a.K <= 3072 && a.cu_limit <= 0 && of_gemm_w4h_eligible(a))
            return of_gemm_w4h_try(a, s);
        // Stream-K first: a tile count the workgroup count does not divide is shared out evenly (needs the workspace; without
        // it the N-split / partial-round forms below).
        const int G = sk_grid_for(a, tiles256);
        if (G > 0 && sk_usable(a, G)) {
            OfGemmArgs w = a;
            w.sk_grid = G;
            if (a.epi == OF_EPI_ACC_F32) w.sumsq_out = sumsq;      // a shared tile is finished -- and summed -- by exactly one workgroup
            return of_gemm_w4m_try(w, s);
        }
        // Tile quantisation: a grid whose last round of 256x256 tiles would be under half full (OF-4B: M = 8192, N = 2560 ->
        // 320 tiles = 1.25 rounds of 256 CUs, 790-870 TFLOP/s where full rounds reach 1250) is split along N into whole rounds
        // of big tiles + a remainder strip on the 128x128 kernel (8192 x 512 -> 256 small tiles: one round).  Two launches on
        // disjoint output columns; every epilogue is column-local, the gate-gradient partials of the two finish in order.
        const int tm = a.M / 256, tn = a.N / 256;
        int n1 = 0;
        if (tiles256 > 256 && tiles256 % 256)
            for (int n = tn - 1; n >= 1 && !n1; --n)
                if (((long)tm * n) % 256 == 0) n1 = n;
        if (n1 && (long)tm * (tn - n1) <= 128 && mid_ok) {
            const long off = (long)n1 * 256;
            const int c_bytes = a.epi == OF_EPI_ACC_F32 ? 4 : (a.epi == OF_EPI_GATE_RESID ? (a.io_f32 ? 4 : 2) : 2);
            const int aux_bytes = a.epi == OF_EPI_GATE_RESID ? (a.io_f32 ? 4 : 2) : 2;
            OfGemmArgs left = a, right = b;
            left.N = (int)off;
            right.N = a.N - (int)off;
            right.B = a.b_trans ? a.B + off : a.B + off * a.ldb;
            right.C = (char*)a.C + off * c_bytes;
            if (a.C2) right.C2 = (char*)a.C2 + off * 2;
            if (a.aux) right.aux = (const char*)a.aux + off * aux_bytes;
            // Both halves are checked BEFORE anything is launched: once the left part has run, falling through to
            // the whole-problem launch below would apply its columns twice for the accumulating epilogues (ACC beta = 1, *_DOT,
            // in-place GATE_RESID).
            if (of_gemm_w4m_eligible(left) && of_gemm_mid_eligible(right)) {
                const int rc = of_gemm_w4m_try(left, s);
                return rc ? rc : of_gemm_mid_try(right, s);
            }
        }
        OfGemmArgs w = a;
        if (a.epi == OF_EPI_ACC_F32) w.sumsq_out = sumsq;
        const int rc = of_gemm_w4m_try(w, s);
        if (rc != OF_E_SHAPE) return rc;
    }
    if ((a.safe == 0 && pp_ok) || pp_forced) {   // 4 = force the ping-pong kernel whenever the shape is eligible
        const int rc = of_gemm_pp_try(a, s);   // 256x256 ping-pong LDS-DMA kernel for tile-aligned shapes
        if (rc != OF_E_SHAPE) return rc;
    }
    // tile-aligned shapes that do not fill the chip with 256x256 tiles: the 8-wave LDS-DMA 128x128 kernel (5 forces it)
    if ((a.safe == 0 || a.safe == 5) && mid_ok) {
        const int rc = of_gemm_mid_try(b, s);
        if (rc != OF_E_SHAPE) return rc;
    }
    // 128 x 64 tiles when the 128 x 128 grid leaves at most one workgroup per CU (safe = 3 forces them: self-check)
    if ((a.safe == 0 && (long)tiles_m * tiles_n <= 256 && a.N > 64) || a.safe == 3) {
        grid.x = (unsigned)(tiles_m * ((a.N + 63) / 64));
        return dispatch(b, grid, s, true);
    }
    return dispatch(b, grid, s);
}


// Several independent weight-gradient GEMMs (TN, OF_EPI_ACC_F32) as ONE grid on the 128x128 LDS-DMA kernel + ONE reduce grid: every
// problem is split along K as of_gemm would split it alone (fp32 slabs in ITS workspace, combined in slice order: the bits of the
// separate launches), the launch boundaries -- where the chip runs half empty -- are gone.  Anything the batch form does not cover
// (other layouts / epilogues, shapes the 128x128 kernel does not take, a missing workspace, more than OF_GEMM_BATCH_MAX problems)
// runs as the separate of_gemm launches: same results either way.
extern "C" int of_gemm_batch(const OfGemmArgs* args, int n, void* stream) {
    if (!args || n <= 0) return OF_E_ARG;
    of_stream_t s = (of_stream_t)stream;
    bool batched = n >= 2 && n <= OF_GEMM_BATCH_MAX;
    OfGemmBatchArgs m{};
    OfGemmBatchArgs r{};
    int total = 0, rtotal = 0;
    for (int i = 0; batched && i < n; ++i) {
        const OfGemmArgs& a = args[i];
        if (!a.A || !a.B || !a.C || a.group_kind || a.safe || !(a.a_trans && a.b_trans) || a.epi != OF_EPI_ACC_F32) batched = false;
        else if ((a.lda & 7) || (a.ldb & 7) || (a.ldc & 3) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15)) batched = false;
        else {
            OfGemmArgs b = a;
            b.sumsq_out = nullptr;
            b.ksplit = pick_ksplit(a, true);
            const size_t need = b.ksplit > 1 ? (size_t)b.ksplit * a.M * a.N * sizeof(float) : 0;
            const long tiles256 = (long)(a.M / 256) * (a.N / 256);
            const bool big = !(a.M % 256) && !(a.N % 256) && !(a.K % 64) && tiles256 >= 128;       // of_gemm would take the big tile
            if (big || b.ksplit < 2 || !a.workspace || a.workspace_bytes < need || ((uintptr_t)a.workspace & 15) || !of_gemm_mid_eligible(b))
                batched = false;
            else {
                m.a[i] = b;
                total += (b.M / 128) * (b.N / 128) * b.ksplit;
                m.wg_end[i] = total;
                r.a[i] = b;
                long blocks = (((long)a.M * a.N >> 2) + 255) / 256;
                if (blocks > 1024) blocks = 1024;
                rtotal += (int)blocks;
                r.wg_end[i] = rtotal;
            }
        }
    }
    if (!batched) {
        for (int i = 0; i < n; ++i) {
            const int rc = of_gemm(&args[i], stream);
            if (rc) return rc;
        }
        return 0;
    }
    m.n = r.n = n;
    for (int i = n; i < OF_GEMM_BATCH_MAX; ++i) {
        m.wg_end[i] = total;
        r.wg_end[i] = rtotal;
    }
    int rc = of_gemm_mid_batch_launch(m, total, s);
    if (rc) return rc;
    return of_launch(of_splitk_reduce_batch_kernel, of_dim3{(unsigned)rtotal, 1, 1}, 256, 0, s, r);
}
