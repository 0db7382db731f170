/* libofhip -- C ABI of the MI355X-native OpenFlamingo visual-conditioning path.
 *
 * The reference (mlfoundations/open_flamingo) is pure Python and has no FFI layer; its boundary for this
 * path is the module API in open_flamingo/src/helpers.py.  This header is the native boundary that the
 * drop-in modules in open_flamingo_amd/src/helpers.py bind through ctypes, and that a reference
 * maintainer would bind the same way.  Each entry point names the reference code it
 * replaces (file:line relative to the reference checkout).
 *
 * Conventions
 *  - Plain C: raw device pointers, ints, floats and a hipStream_t passed as void*.  No torch types.
 *  - The caller owns every buffer (inputs, outputs, saved-for-backward, workspace).  The library allocates
 *    nothing, keeps no global mutable state, and never synchronises the device: work is enqueued on the
 *    given stream and the call returns.
 *  - Return value: 0 ok; negative = argument/shape error (OF_E_*); positive = hipError_t from a launch.
 *  - bf16 = raw uint16 bfloat16.  "stream dtype" (io_f32 = 1 fp32 / 0 bf16) is the dtype of the residual
 *    stream tensors the reference keeps in fp32 under amp_bf16 (train_utils.py:34-41): x, y, their grads.
 *  - All matrices row-major; ld* in elements.
 */
#ifndef OF_HIP_H
#define OF_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OF_ABI_VERSION 11
#define OF_E_ARG (-1)      /* null pointer / negative size */
#define OF_E_SHAPE (-2)    /* shape not supported by the kernels (see each function) */
#define OF_E_ALIGN (-3)    /* pointer or leading dimension not 16-byte aligned */
#define OF_E_WORKSPACE (-4)/* workspace too small */

int of_abi_version(void);
/* 1 when built for gfx950, 2 for the host emulator build used by tests (never shipped). */
int of_build_kind(void);

/* ---------------------------------------------------------------------------------------------------
 * GEMM with fused epilogues: acc[m][n] = sum_k A(m,k) * B(n,k), bf16 operands, fp32 accumulate (MFMA).
 *   a_trans = 0: A stored [M][K] (lda >= K);  1: A stored [K][M] (lda >= M)
 *   b_trans = 0: B stored [N][K] (ldb >= K);  1: B stored [K][N] (ldb >= N)
 * i.e. nn.Linear forward y = x W^T is (a_trans=0,b_trans=0) with B = W; dX = dY W is (0,1) with B = W;
 * dW = dY^T X is (1,1).  Replaces the aten::mm calls under nn.Linear in helpers.py:19,21,36-38,154-156
 * and their autograd backward.  Dimensions that are vector-loaded (the contiguous one of each operand)
 * must be multiples of 8; N must be a multiple of 4.
 * g = tanh(*gate) if gate != NULL else 1.
 *   OF_EPI_STORE_BF16  C(bf16)  = g*alpha*acc
 *   OF_EPI_GELU        C(bf16)  = gelu_erf(acc); if C2: C2(bf16) = acc            (helpers.py:20)
 *   OF_EPI_GATE_RESID  C(T)     = aux(T) + g*alpha*acc, T = stream dtype          (helpers.py:267-277,130-131)
 *   OF_EPI_DGELU_DOT   C(bf16)  = g*alpha*acc*gelu'(aux);  *dot_out += (1-g^2)*sum(gelu(aux)*acc)
 *   OF_EPI_SCALE_DOT   C(bf16)  = g*alpha*acc;             *dot_out += (1-g^2)*sum(aux*acc)
 *   OF_EPI_ACC_F32     C(f32)   = g*alpha*acc + beta*C
 * The two *_DOT epilogues produce the tanh-gate gradients of GatedCrossAttentionBlock
 *  without materialising the un-gated branch output.
 */
enum { OF_EPI_STORE_BF16 = 0, OF_EPI_GELU = 1, OF_EPI_GATE_RESID = 2, OF_EPI_DGELU_DOT = 3,
       OF_EPI_SCALE_DOT = 4, OF_EPI_ACC_F32 = 5 };

typedef struct OfGemmArgs {
    const uint16_t* A;
    const uint16_t* B;
    int M, N, K;
    int lda, ldb;
    int a_trans, b_trans;
    int epi;
    void* C;
    int ldc;
    void* C2;          /* OF_EPI_GELU: optional pre-activation output (bf16, ldc) */
    const void* aux;   /* residual (stream dtype) or saved bf16 activation, [M][ldaux] */
    int ldaux;
    const float* gate; /* device pointer to the raw gate parameter, or NULL */
    float alpha, beta;
    float* dot_out;    /* *_DOT epilogues: device scalar the gate-gradient sum is ADDED to, or NULL.  Deterministic: every workgroup
                          writes one partial to `workspace` (of_gemm_workspace_bytes(args) bytes, required: OF_E_WORKSPACE
                          otherwise) and a second one-workgroup launch adds the partials in a fixed order -- no fp atomics */
    int io_f32;        /* OF_EPI_GATE_RESID: 1 = fp32 stream, 0 = bf16 stream */
    int safe;          /* 0 = production: of_gemm selects the kernel (M <= 16 untransposed -> weight-streaming skinny kernel; tile-aligned
                          shapes that fill the chip -> the 4-wave 256x256 kernel on 16x16x32 MFMAs, two workgroups per CU on 256x128 tiles
                          for the *_DOT epilogues; smaller tile-aligned shapes -> the 8-wave 128x128 LDS-DMA kernel; otherwise the general
                          128x128 kernel, split along K when the output is small).  1 = cross-check: the general kernel on its scalar-LDS
                          fragment path (any shape, slow) -- the one alternative the library keeps, so that a caller can check a result
                          against an independent kernel.  Anything else: OF_E_ARG. */
    int ksplit;        /* internal: filled in by of_gemm (number of K slices of a split-K launch); callers pass 0 */
    void* workspace;   /* optional scratch for split-K partial sums (fp32 slabs): with at least
                          of_gemm_workspace_bytes(args) bytes the K slices are combined by a second pass in a fixed
                          order (deterministic); without it they are combined with fp32 atomics */
    size_t workspace_bytes;
    /* Grouped B operand (0 / NULL = off): `groups` is a DEVICE array of n pointers, each a B matrix of its own.
     *   group_kind 1 (b_trans = 0): B is grouped along N -- output columns [g*E, (g+1)*E) use groups[g] (an [E][K] matrix):
     *       one launch computes y[:, gE:(g+1)E] = x W_g^T for every g (all gated blocks' to_kv of the same media tensor,
     *       helpers.py:189 called from 24 blocks);
     *   group_kind 2 (b_trans = 1): B is grouped along K -- k in [g*E, (g+1)*E) uses groups[g] (an [E][N] matrix):
     *       one launch computes dX = sum_g dY[:, gE:(g+1)E] W_g (the media gradient of all blocks).
     * E = group_extent, a multiple of 256 (kind 1) / 64 (kind 2); shapes must be big-tile eligible (M, N % 256, K % 64);
     * args->B is ignored, ldb applies to every group. */
    const void* const* groups;
    int group_kind;
    int group_extent;
    /* Stream-K scheduling of the big-tile kernel (ABI v8).  cu_limit: how many CUs this launch may count on, 0 = all 256 -- a
     * caller that knows another kernel holds CUs for the duration (an RCCL collective on a side stream) passes what is left, and the
     * launch is laid out for that many workgroups instead of running a second round for the displaced tiles.  Whenever the number
     * of 256x256 tiles is not a multiple of the workgroup count (OF-4B's 400-tile weight gradients, OF-9B's 128-tile launches, any
     * launch under a cu_limit) and `workspace` holds of_gemm_workspace_bytes(args) bytes, every workgroup takes the same share of
     * (tile, K-stage) units: whole tiles first, then a fraction of the remaining ones; a tile shared between workgroups is finished
     * by the one that holds its last K stages, which adds the others' fp32 partial sums (through `workspace`) in ascending K
     * order -- deterministic for a given (shape, workgroup count).  Without the workspace the launch falls back to one tile per
     * workgroup.  sk_grid: internal, filled in by of_gemm; callers pass 0. */
    int cu_limit;
    int sk_grid;
    /* OF_EPI_ACC_F32 (ABI v9): optional sum of squares of the FINAL output values (after alpha, gate and beta), one fp32 partial per
     * 256x256 output tile, m-major tile order, WRITTEN (not added) to sumsq_out[tile]: a weight gradient's share of the global gradient
     * norm (train_utils.py:199: clip_grad_norm_) leaves with the GEMM that produces the gradient instead of in another pass over it.
     * Honoured only where of_gemm_sumsq_slots(args) > 0 -- a single launch of the 256x256 kernel: no split along K or N -- and ignored
     * elsewhere (the slots stay untouched): callers ask first.  NULL = off. */
    float* sumsq_out;
} OfGemmArgs;

/* Reproducibility: the same call (shape, operands, workspace, cu_limit) gives the same bits launch after launch, on every kernel.  The
 * ORDER of the fp32 additions along K belongs to the kernel of_gemm selects: launches it sends to the 256x256 kernel by itself (safe = 0)
 * start each tile's K loop at a stage that depends on the XCD the tile runs on and wrap around (round 5: spreads the requests of
 * operands that come from HBM over its channels) -- so two rows with the same operand values in DIFFERENT tiles agree to summation order
 * (1e-6 relative), not bit for bit.  The cross-check kernel (safe = 1) walks K in stage order 0, 1, 2, ... */
int of_gemm(const OfGemmArgs* args, void* stream);
/* n independent problems in ONE launch (ABI v8).  For 2..4 weight-gradient problems (a_trans = b_trans = 1, OF_EPI_ACC_F32) that
 * of_gemm would each run split along K on the 128x128 kernel -- the 512-wide projections' gradients of a gated block: to_q, to_out,
 * to_kv -- and whose `workspace` fields each hold of_gemm_workspace_bytes() bytes (distinct regions), the problems share one GEMM grid
 * and one reduce grid: the bits of the separate launches without their launch boundaries.  Everything else runs as n of_gemm calls. */
int of_gemm_batch(const OfGemmArgs* args, int n, void* stream);
/* Bytes of workspace of_gemm would use for these arguments: split-K slabs (optional, see `workspace`), the per-workgroup
 * partials of a *_DOT launch with dot_out (required), the stream-K partial tiles + flags of a big-tile launch whose tile count
 * is not a multiple of its workgroup count (optional, see cu_limit), else 0. */
size_t of_gemm_workspace_bytes(const OfGemmArgs* args);
/* Number of fp32 slots of_gemm would write to args->sumsq_out for these arguments (the launch's 256x256 tiles), or 0 if the launch
 * does not take the form that honours it (ABI v9). */
size_t of_gemm_sumsq_slots(const OfGemmArgs* args);

/* ---------------------------------------------------------------------------------------------------
 * LayerNorm (eps 1e-5, affine) -- nn.LayerNorm in helpers.py:18,33-34,105,152.
 * x: rows x dim in stream dtype (x_f32=1 fp32, 0 bf16), row stride ldx; y: bf16, row stride ldy (so the
 * Perceiver can write media rows and latent rows into one [N][v+n][D] buffer, helpers.py:53);
 * stats: rows x 2 fp32 (mean, rstd) saved for backward.  dim % 8 == 0, dim <= 8192.
 * of_layernorm_fwd_out writes the stream dtype instead of bf16 (PerceiverResampler.norm, helpers.py:132).
 */
int of_layernorm_fwd(const void* x, int x_f32, long ldx, const float* w, const float* b, uint16_t* y, long ldy,
                     float* stats, long rows, int dim, void* stream);
int of_layernorm_fwd_out(const void* x, int x_f32, long ldx, const float* w, const float* b, void* y, int y_f32,
                         long ldy, float* stats, long rows, int dim, void* stream);
/* Residual add + LayerNorm in one pass: s = x + add (add: bf16 branch output, row stride ldadd) is written to xsum in x's
 * dtype (row stride ldsum; may alias x) and y = LN(s) in bf16 (y_f32 = 0) or fp32 -- "hidden = hidden + f(...); LN(hidden)"
 * between two sub-layers of a frozen tower (HF MptBlock / CLIPEncoderLayer residual adds) without the
 * separate mixed-dtype add kernel. */
int of_layernorm_fwd_add(const void* x, int x_f32, long ldx, const uint16_t* add, long ldadd, void* xsum, long ldsum,
                         const float* w, const float* b, void* y, int y_f32, long ldy, float* stats, long rows, int dim,
                         void* stream);
/* Grouped destination: row r is written at y + (r / grp_rows) * grp_stride + (r % grp_rows) * ldy (elements), so
 * LN_media(x) and LN_latents(latents) land directly in the [N][v+n][D] key/value input of helpers.py:53 without
 * a torch.cat copy.  y2 (optional) receives a second contiguous bf16 copy (the to_q operand, helpers.py:52). */
int of_layernorm_fwd_grouped(const void* x, int x_f32, long ldx, const float* w, const float* b, uint16_t* y, long ldy,
                             long grp_rows, long grp_stride, uint16_t* y2, float* stats, long rows, int dim,
                             void* stream);
/* Backward: dy (bf16 if dy_f32==0 else fp32, row stride lddy, optionally grouped like the forward destination)
 * plus an optional second upstream gradient dy2 (bf16, contiguous); x + stats as saved.
 *   dx_out(T) = (resid ? resid(T) : 0) + LN_bwd(dy + dy2)      T = stream dtype (out_f32)
 *   dx_bf16   = optional bf16 copy of dx_out (feeds the next GEMMs as an operand), may be NULL
 *   dw, db   += column reductions (fp32, accumulated atomically; caller zero-initialises or accumulates)
 * dx_out and dx_bf16 may both be NULL when only dw/db are needed (norm_media: the ViT features normally carry
 * no gradient). */
int of_layernorm_bwd(const void* dy, int dy_f32, long lddy, long dy_grp_rows, long dy_grp_stride, const uint16_t* dy2,
                     const void* x, int x_f32, long ldx, const float* stats, const float* w, const void* resid,
                     void* dx_out, int out_f32, long lddx, uint16_t* dx_bf16, float* dw, float* db, long rows, int dim,
                     float* workspace, size_t workspace_bytes, void* stream);
/* Optional scratch for of_layernorm_bwd: with this many bytes the dw/db column sums are combined from per-workgroup
 * partials (16 adds per column instead of one per workgroup); without it every workgroup adds its column sums
 * with fp32 atomics. */
size_t of_layernorm_bwd_workspace_bytes(long rows, int dim);

/* ---------------------------------------------------------------------------------------------------
 * Windowed multi-head attention core, head dim 64 (or 128), flash-style (scores never reach HBM).
 * One kernel family serves both hot-path attentions:
 *   - PerceiverAttention core (helpers.py:55-64): text_time == NULL, every query sees all keys.
 *   - MaskedCrossAttention core (helpers.py:192-231): per-query key window derived from text_time
 *     (helpers.py:196-218) instead of a materialised (B,1,L,T*n) mask:
 *        only_immediate=1:  1 <= tt <= T -> keys [(tt-1)*n, tt*n);  tt == 0 -> output 0 (helpers.py:223-229);
 *                           tt > T -> every key is masked with -finfo.max, softmax is uniform over all T*n keys
 *        only_immediate=0:  tt >= 1 -> keys [0, min(tt,T)*n);  tt == 0 -> uniform over all keys
 * Layout: q[(batch*Lq + i)*ldq + h*head_dim + d], k/v[(batch*Lk + j)*ldk + h*head_dim + d] (k and v may point into one
 * fused kv buffer), o like q with ldo.  scale = dim_head^-0.5 is applied to the fp32 scores.
 * lse (batch,H,Lq) fp32 = row log-sum-exp saved for backward (+inf marks zeroed rows).
 */
typedef struct OfAttnArgs {
    const uint16_t* q; const uint16_t* k; const uint16_t* v;
    uint16_t* o;
    float* lse;
    const int32_t* text_time; /* (batch, Lq) or NULL */
    int batch, heads, Lq, Lk;
    long ldq, ldk, ldv, ldo;
    int n_per_media, T_img, only_immediate;
    float scale;
    /* backward only */
    const uint16_t* dout; long lddo;
    uint16_t* dq; long lddq;
    uint16_t* dk; uint16_t* dv; long lddk, lddv;
    float* delta;             /* (batch,H,Lq) fp32 scratch: rowsum(dO*O), written by the dq pass */
    int safe;                 /* DEBUG / SELF-CHECK ONLY -- production callers pass 0 (auto: forward with K and V of a (batch, head)
                                 resident in LDS where that pays, else one workgroup per 64-query tile).  1 = tiled kernels with the
                                 scalar-LDS transposed-fragment path, 2 = tiled kernels (two-pass backward), 3 = resident-K/V forward whenever the
                                 two images fit 160 KB and the single-pass backward whenever it applies (Lq, Lk <= 256, no text_time): every
                                 value gives correct results, tests compare the forms with each other. */
    /* causal self-attention with ALiBi (the frozen MPT blocks); all zero/NULL for the two hot-path uses */
    int head_dim;             /* 0 or 64: 64;  128 */
    int causal;               /* 1: query i sees keys [0, i + 1 + Lk - Lq); text_time must be NULL */
    const float* alibi_slopes;/* (heads) fp32 or NULL: score += slope[h] * (j - (i + Lk - Lq)) before the softmax */
    const int32_t* kv_len;    /* (batch) or NULL, causal only: number of real (non right-padding) keys per sequence */
    int head_valid;           /* ABI v11, compact heads: 0 or head_dim = every head owns head_dim columns.  Else a multiple of 8 with
                                 8 <= head_valid < head_dim (OF_E_SHAPE otherwise; safe = 1: OF_E_ARG): head h owns columns [h * head_valid, (h + 1) * head_valid) of q, k, v, o,
                                 dout, dq, dk, dv; the kernels run at head_dim with the missing columns read as zeros and never stored
                                 (GPT-NeoX head size 80 -- RedPajama-INCITE-3B behind OF-4B -- at head_dim 128 without padded copies
                                 in HBM).  `scale` stays the caller's (head_valid ** -0.5 for the reference's attention). */
} OfAttnArgs;

int of_attn_fwd(const OfAttnArgs* args, void* stream);
/* Backward = two passes (dq: one workgroup per query tile; dk/dv: one per key block), or -- short self-attention without
 * text_time, Lq and Lk <= 256: the frozen MPT blocks -- one pass per (batch, head) (attn_bwd_res.hip); both deterministic,
 * no atomics.  dq/dk/dv are bf16 (they feed the projection-weight GEMMs as operands).  `delta` is scratch of the two-pass form.
 * Alignment: q, k, v, o, dout, dq, dk, dv 16-byte aligned, every leading dimension a multiple of 8 elements (all loads and
 * stores are 16 bytes wide), else OF_E_ALIGN. */
int of_attn_bwd(const OfAttnArgs* args, void* stream);

/* text_time (helpers.py:199-208): cumsum of media_locations along the sequence, or (cached decode branch)
 * the per-sequence count broadcast to Lq new tokens.  media_locations: uint8 (B, Lm). */
int of_text_time(const uint8_t* media_locations, int32_t* text_time, int B, int Lm, int Lq, int use_cached,
                 void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The attention branch of GatedCrossAttentionBlock as ONE kernel (ABI v10; helpers.py:184-194 norm + to_q, :192-231 masked
 * attention, :231-233 to_out, :267-276 tanh gate + residual, and the LayerNorm that opens the block's FeedForward, helpers.py:18):
 *     xn = LN(x);  q = xn Wq^T;  o = softmax_window(q K^T * scale) V;  y = x + tanh(*gate) * (o Wout^T);  u2 = LN2(y)
 * A workgroup owns 32 consecutive text positions of one sequence and all 8 heads.  Shapes it takes (OF_E_SHAPE otherwise -- callers
 * then run the separate launches, of_xattn_fused_eligible() asks without launching): heads = 8, head_dim = 64, d in {256, 512,
 * 1024, 2048}, L a multiple of 32.  x / y: (B*L) x d in the stream dtype (x_f32); k, v: (B*Lk) x 512 bf16 views (row strides ldk,
 * ldv) of the projected media; text_time (B, L) int32 or NULL (no mask); the key window of a position is of_attn_fwd's.
 * wq_pk / wout_pk: FRAGMENT-MAJOR copies (of_pack_frag16) of to_q.weight (512 x d) and to_out.weight (d x 512): with 32 rows per
 * workgroup the weights are streamed from L2 once per workgroup straight into MFMA operand registers, and a wave's 16-byte-per-lane
 * load has to be 1 KiB contiguous for that to run at the L2's rate.
 * Optional outputs (NULL = not written): xn, stats (rows x 2: mean, rstd), q, o, lse -- exactly what of_layernorm_fwd / of_gemm /
 * of_attn_fwd would have saved for the backward; ln2_w = NULL: no second LayerNorm (u2, stats2 unused). */
typedef struct OfXattnFusedArgs {
    const void* x; int x_f32; long ldx;
    const float* ln_w; const float* ln_b;
    const uint16_t* wq_pk;
    const uint16_t* k; const uint16_t* v; long ldk, ldv;
    const int32_t* text_time;
    const uint16_t* wout_pk;
    const float* gate;             /* device pointer to attn_gate (raw), or NULL: 1 */
    const float* ln2_w; const float* ln2_b;
    uint16_t* xn; long ldxn;
    float* stats;
    uint16_t* q; long ldq;
    uint16_t* o; long ldo;
    float* lse;                    /* (B, 8, L) */
    void* y; long ldy;
    uint16_t* u2; long ldu2;
    float* stats2;
    int B, L, Lk, d, heads, head_dim;
    int n_per_media, T_img, only_immediate;
    float scale;
} OfXattnFusedArgs;
int of_xattn_fused_eligible(const OfXattnFusedArgs* args);
int of_xattn_fused_fwd(const OfXattnFusedArgs* args, void* stream);
/* P[((nt * (K / 32) + ks) * 64 + lane) * 8 .. + 7] = W[16 nt + (lane & 15)][32 ks + 8 (lane >> 4) .. + 7]: the fragment-major copy of a
 * row-major N x K bf16 matrix (row stride ldw) -- the 16 bytes lane `lane` of a wave feeds v_mfma_f32_16x16x32_bf16 for n-tile nt, k-step
 * ks, one wave load = 1 KiB contiguous.  N % 16 == 0, K % 32 == 0; P holds N * K elements. */
int of_pack_frag16(const uint16_t* W, int N, int K, long ldw, uint16_t* P, void* stream);
/* n matrices in one launch (descs: a HOST array, copied into the kernel arguments): the step epilogue re-packs the to_q / to_out
 * weights of every gated block once per optimizer step, behind the AdamW pass that rewrites their bf16 copies. */
#define OF_PACK_BATCH_MAX 64
typedef struct OfPackDesc {
    const uint16_t* W;
    uint16_t* P;
    int N, K;
    long ldw;
} OfPackDesc;
int of_pack_frag16_batch(const OfPackDesc* descs, int n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Small element-wise helpers of the path. */
int of_cast_f32_to_bf16(const float* x, uint16_t* y, long n, void* stream);
int of_cast_bf16_to_f32(const uint16_t* x, float* y, long n, void* stream);
/* y[r][c] (bf16, ldy) = src[r % src_rows][c] for r < rows: "repeat(latents, 'n d -> b T n d')" helpers.py:128 */
int of_broadcast_rows(const float* src, int src_rows, void* y, int y_f32, long ldy, long rows, int dim, void* stream);
/* dst[r % dst_rows][c] += src[r][c]: gradient of the repeat above (sum over b,T). dst fp32. */
int of_reduce_rows(const void* src, int src_f32, long rows, int dim, float* dst, int dst_rows, void* stream);
/* Perceiver position tables (helpers.py:117-119,123-124): out[r][c] = x[r][c] + e1[(r / inner1) % outer1][c]
 * + e2[(r / inner2) % outer2][c] (either table may be NULL); x/out stream dtype, tables fp32.  With rows indexed
 * ((b*T + t)*F + f)*v + i: frame_embs use inner = v, outer = F; media_time_embs inner = F*v, outer = T. */
int of_add_embs(const void* x, int x_f32, const float* e1, long inner1, int outer1, const float* e2, long inner2,
                int outer2, void* out, long rows, int dim, void* stream);
/* Gradient of one table: dst[o][c] += sum over rows r with (r / inner) % outer == o of src[r][c].  dst fp32. */
int of_reduce_rows_strided(const void* src, int src_f32, long rows, int dim, long inner, int outer, float* dst,
                           void* stream);
/* y = x * sigmoid(1.702 x), bf16 -> bf16: the "quick GELU" of the frozen CLIP tower's MLP (HF runs it as
 * three element-wise passes).  Forward only: the vision tower runs under no_grad (flamingo.py:194-195). */
int of_quick_gelu(const uint16_t* x, uint16_t* y, long n, void* stream);
/* Element-wise pieces of a frozen MPT block's MLP and residual stream (HF MptMLP: up_proj -> nn.GELU(exact) ->
 * down_proj -> + residual), bf16 in / out like the eager chain under autocast:
 *   of_gelu_fwd: y = gelu_erf(x);   of_gelu_bwd: dx = dy * gelu_erf'(x);   of_add_bf16: out(fp32) = x(fp32) + y(bf16). */
int of_gelu_fwd(const uint16_t* x, uint16_t* y, long n, void* stream);
int of_gelu_bwd(const uint16_t* dy, const uint16_t* x, uint16_t* dx, long n, void* stream);
int of_add_bf16(const float* x, const uint16_t* y, float* out, long n, void* stream);
/* Frozen GPT-NeoX blocks (OF-4B = RedPajama-INCITE-3B; HF GPTNeoXAttention.forward): rotary embedding +
 * head padding in one pass.  qkv: [rows][heads * 3 * head_size] bf16 in HF's per-head [q_h | k_h | v_h] layout (row stride
 * ldqkv); cos / sin: [L][rot_dims] fp32 (HF's position_embeddings of one sequence; position = row % L); q, k, v: three
 * [rows][heads * head_pad] bf16 matrices (row stride ldo), every head zero-padded from head_size to head_pad columns so that the
 * attention kernels' head sizes (64 / 128) cover head size 80; the first rot_dims columns of q and k are rotated
 * (apply_rotary_pos_emb).  inverse = 1 is the gradient's way back: q, k, v hold padded dq, dk, dv and qkv receives d(qkv).
 * of_head_repack: dst[row][h * dst_hs + c] = c < src_hs ? src[row][h * src_hs + c] : 0 -- pads or trims every head (the
 * attention output back to head_size columns in front of the dense projection, its gradient forward to head_pad). */
int of_rotary_neox(uint16_t* qkv, long ldqkv, const float* cos, const float* sin, long L, uint16_t* q, uint16_t* k, uint16_t* v,
                   long ldo, long rows, int heads, int head_size, int rot_dims, int head_pad, int inverse, void* stream);
int of_head_repack(const uint16_t* src, long lds, uint16_t* dst, long ldd, long rows, int heads, int src_head_size,
                   int dst_head_size, void* stream);
/* out(T) = a(T) + b(T) */
int of_add(const void* a, const void* b, void* out, int f32, long n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Step epilogue over flat fp32 buffers: global-norm clip + AdamW, replacing
 * clip_grad_norm_ / optimizer.step / zero_grad of open_flamingo/train/train_utils.py:199-216 with the parameter
 * groups of open_flamingo/train/train.py:392-408.
 *   of_sumsq_partial: partials[0..OF_SUMSQ_PARTS) = per-workgroup sums of g^2 (every slot written; one call per gradient
 *                   buffer, each with its own OF_SUMSQ_PARTS slots);  of_sumsq_finish: *acc = sum of `count` partial slots in a
 *                   fixed order.  Together they are the reference's clip_grad_norm_ total, with no floating-point atomics: the
 *                   same gradients give the same bits on every rank and run, so replicas stay identical.
 *   of_adamw_clip:  the effective gradient is grad_scale * g (grad_scale = 1/world_size when g holds the all-reduced SUM);
 *                   coef = min(1, max_norm / (grad_scale * sqrt(*sumsq) + 1e-6)) (max_norm <= 0: no clipping);
 *                   torch.optim.AdamW update with gradient coef * grad_scale * g at 1-based `step`; p_bf16 (optional) receives the bf16 copy of the new
 *                   parameters; zero_grad != 0 clears g in the same pass.  A non-finite *sumsq (NaN / Inf anywhere in the
 *                   gradients) skips the update -- p, m, v, p_bf16 untouched, g still cleared if asked: the reference's skip-on-NaN
 *                   (train_utils.py:161-169) decided on the device, identically on every rank.  No host synchronisation.
 *                   applied_steps (optional device int): Adam's bias correction then uses *applied_steps instead of `step`;
 *   of_step_advance: *applied_steps += 1 iff *sumsq is finite -- launched once per optimizer step between of_sumsq_finish and
 *                   the of_adamw_clip launches, so the count of APPLIED updates lives on the device and a skipped (NaN) step
 *                   does not advance the bias correction (the reference `continue`s before optimizer.step()). */
#define OF_SUMSQ_PARTS 512
int of_sumsq_partial(const float* g, long n, float* partials, void* stream);
int of_sumsq_finish(const float* partials, long count, float* acc, void* stream);
int of_adamw_clip(float* p, float* g, float* m, float* v, uint16_t* p_bf16, long n, const float* sumsq,
                  float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  int step, int zero_grad, const int* applied_steps, void* stream);
int of_step_advance(const float* sumsq, int* applied_steps, void* stream);
/* ABI v8: the two streaming passes as NARROW launches -- max_workgroups fat workgroups (1024 threads, one per CU) instead of a grid
 * that covers the chip, so that the pass holds that many CUs and leaves the others to work on another stream (the next step's frozen
 * vision-tower forward, train/step.py: next_vision_x).  Same arithmetic per element and per partial slot: bit-identical results;
 * max_workgroups = 0 is the plain launch. */
int of_sumsq_partial_w(const float* g, long n, float* partials, int max_workgroups, void* stream);
int of_adamw_clip_w(float* p, float* g, float* m, float* v, uint16_t* p_bf16, long n, const float* sumsq,
                    float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                    int step, int zero_grad, const int* applied_steps, int max_workgroups, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Token-level cross entropy of the causal-LM loss (the reference's Flamingo.forward passes `labels` to the HF language
 * model, open_flamingo/src/flamingo.py:112-121; transformers computes logits.float() -> log_softmax -> nll, mean over
 * labels != ignore_index).  logits: rows x vocab (bf16, or fp32 when logits_f32), row stride ld elements, any 2-byte
 * alignment; labels: rows int64 (already shifted by the caller).
 *   of_ce_fwd: lse[row] = log sum_j exp(logit[row][j]) (fp32); loss_rows[row] = lse - logit[row][label], 0 for ignored rows,
 *              NaN for a label outside [0, vocab) that is not ignore_index (a caller bug: surfaces as a NaN loss)
 *   of_ce_bwd: dlogits[row][j] = *gscale * (exp(logit - lse[row]) - [j == label]) in the logits' dtype, 0 for ignored rows
 *              (*gscale = upstream gradient / number of valid rows, a device scalar: no host sync)
 */
int of_ce_fwd(const void* logits, int logits_f32, long ld, const long long* labels, long long ignore_index, long rows,
              int vocab, float* lse, float* loss_rows, void* stream);
int of_ce_bwd(const void* logits, int logits_f32, long ld, const long long* labels, long long ignore_index, long rows,
              int vocab, const float* lse, const float* gscale, void* dlogits, long ldd, void* stream);

#ifdef __cplusplus
}
#endif
#endif
