"""Thin marshalling of torch tensors onto the C ABI of include/of_hip.h.

``Ops`` only turns tensors into (pointer, stride, size) arguments and checks return codes; it holds a library
handle and a way to obtain the current stream.  The product instance (``Ops.default()``) is bound to
``libofhip.so`` + the current HIP stream and is the only one ``open_flamingo_amd.src`` ever uses.
"""
import ctypes as C

import torch

from . import abi

BF16 = torch.bfloat16
F32 = torch.float32

_ERR = {-1: "OF_E_ARG", -2: "OF_E_SHAPE", -3: "OF_E_ALIGN", -4: "OF_E_WORKSPACE"}


def _p(t):
    return None if t is None else t.data_ptr()


def _is_f32(t):
    if t.dtype == F32:
        return 1
    if t.dtype == BF16:
        return 0
    raise TypeError(f"stream tensors must be float32 or bfloat16, got {t.dtype}")


class Ops:
    def __init__(self, lib, stream_fn):
        self.lib = lib
        self._stream_fn = stream_fn
        # CUs a big-tile GEMM launch may count on (OfGemmArgs.cu_limit; 0 = all): train/reducer.py lowers it while its collectives
        # hold CUs on the side stream, so that the GEMMs of the backward are laid out (stream-K) for the CUs that are left
        self.cu_limit = 0
        self.batch_dw = True      # gemm_batch_dw really batches (bench.py --no-batched-dw clears it for the same-box A/B)
        self.gemm_timing = None   # bench.py sets this to a list to collect (key, flops, start_evt, end_evt) per launch
        self.gemm_timing_only = None   # optional set of (ta, tb, epi, kernel label) keys: only those launches are bracketed by events

    _default = None

    @classmethod
    def default(cls):
        """The product instance: libofhip.so (gfx950) on the current HIP stream.  No fallback."""
        if cls._default is None:
            from . import lib as _lib
            handle = _lib.load()
            cls._default = cls(handle, lambda: torch.cuda.current_stream().cuda_stream)
        return cls._default

    def _stream(self):
        s = self._stream_fn()
        return None if s is None else C.c_void_p(s)

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"libofhip {what} failed: {_ERR.get(rc, 'hipError ' + str(rc))} ({rc})")

    # ------------------------------------------------------------------ GEMM
    @staticmethod
    def kernel_label(M, N, K, ta, tb, epi=None, cu_limit=0):
        """Mirror of of_gemm's kernel selection (csrc/gemm.hip): which HIP kernel a launch ends up in -- only used to label
        timing records (bench.py's roofline names ONE kernel so that it can be checked against rocprofv3)."""
        if M <= 16 and not ta and not tb:
            return "skinny"
        if M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and (M // 256) * (N // 256) >= 128:
            if (epi in (abi.EPI_DGELU_DOT, abi.EPI_SCALE_DOT) and not ta and tb and (M // 256) * (N // 256) >= 1024 and K <= 3072
                    and cu_limit <= 0):
                return "w4h256x128"       # two workgroups per CU (csrc/gemm_w4h.hip)
            return "w4m256"
        if M % 128 == 0 and N % 128 == 0 and K % 64 == 0:
            return "mid128"
        return "general128"

    def gemm(self, A, B, out, *, ta=False, tb=False, epi=abi.EPI_STORE_BF16, out2=None, aux=None, gate=None,
             alpha=1.0, beta=0.0, dot=None, safe=0, sumsq=None):
        """acc[m][n] = sum_k A(m,k) B(n,k); A is (M,K) or, with ta, (K,M); B is (N,K) or, with tb, (K,N).
        sumsq (EPI_ACC_F32): fp32 tensor that receives the sum of squares of the final output, one partial per 256x256 tile
        (OfGemmArgs.sumsq_out) -- returns True iff the launch honoured it (it is left untouched otherwise)."""
        assert A.dim() == 2 and B.dim() == 2 and out.dim() == 2
        assert A.dtype == BF16 and B.dtype == BF16 and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1
        M, K = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
        N, Kb = (B.shape[1], B.shape[0]) if tb else (B.shape[0], B.shape[1])
        assert K == Kb, f"K mismatch {K} vs {Kb}"
        assert out.shape[0] == M and out.shape[1] == N, f"out {tuple(out.shape)} != ({M},{N})"
        a = abi.OfGemmArgs()
        a.A, a.B = A.data_ptr(), B.data_ptr()
        a.M, a.N, a.K = M, N, K
        a.lda, a.ldb = A.stride(0), B.stride(0)
        a.a_trans, a.b_trans, a.epi = int(ta), int(tb), epi
        a.C, a.ldc = out.data_ptr(), out.stride(0)
        a.C2 = _p(out2)
        if out2 is not None:
            assert out2.stride(0) == out.stride(0) and out2.dtype == BF16
        a.aux = _p(aux)
        a.ldaux = aux.stride(0) if aux is not None else 0
        a.gate = _p(gate)
        a.alpha, a.beta = float(alpha), float(beta)
        a.dot_out = _p(dot)
        a.io_f32 = _is_f32(out) if epi == abi.EPI_GATE_RESID else 0
        if epi == abi.EPI_GATE_RESID:
            assert aux is not None and aux.dtype == out.dtype
        elif epi == abi.EPI_ACC_F32:
            assert out.dtype == F32
        else:
            assert out.dtype == BF16
        a.safe = safe
        a.cu_limit = self.cu_limit
        took_sumsq = False
        if sumsq is not None and epi == abi.EPI_ACC_F32:
            need = self.lib.of_gemm_sumsq_slots(C.byref(a))
            if need and sumsq.dtype == F32 and sumsq.is_contiguous() and sumsq.numel() >= need and sumsq.device == out.device:
                a.sumsq_out = sumsq.data_ptr()
                took_sumsq = True
        # scratch: split-K fp32 slabs (EPI_ACC_F32), the per-workgroup gate-gradient partials of a *_DOT launch (summed in
        # a fixed order by a second launch: deterministic), or the partial tiles + flags of a stream-K big-tile launch (tile count
        # not a multiple of the workgroup count: csrc/gemm.hip sk_grid_for); one grow-only buffer, reused in stream order
        sk = False
        if safe in (0, 17) and M % 256 == 0 and N % 256 == 0 and K % 64 == 0:
            tiles, grid = (M // 256) * (N // 256), ((self.cu_limit & ~7) or 256)
            grid = min(max(grid, 8), 256)
            sk = safe == 17 or (tiles >= 128 and K >= 4096 and (tiles % grid != 0 if self.cu_limit else tiles < grid))
        if epi == abi.EPI_ACC_F32 or dot is not None or sk:
            need = self.lib.of_gemm_workspace_bytes(C.byref(a))
            if need:
                ws = self.__dict__.get("_gemm_ws")
                if ws is None or ws.numel() * 4 < need or ws.device != out.device:
                    ws = torch.empty((need + 3) // 4, dtype=F32, device=out.device)
                    self._gemm_ws = ws
                a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        if self.gemm_timing is not None:
            key = (int(ta), int(tb), epi, self.kernel_label(M, N, K, ta, tb, epi, self.cu_limit))
            if self.gemm_timing_only is None or key in self.gemm_timing_only:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._chk(self.lib.of_gemm(C.byref(a), self._stream()), "of_gemm")
                e1.record()
                self.gemm_timing.append((key, 2.0 * M * N * K, (M, N, K), e0, e1))
                return took_sumsq if sumsq is not None else out
        self._chk(self.lib.of_gemm(C.byref(a), self._stream()), "of_gemm")
        return took_sumsq if sumsq is not None else out

    def gemm_batch_dw(self, problems):
        """Several independent weight-gradient GEMMs out_i (M_i, N_i) = [gate_i] A_i^T B_i (+ beta_i out_i) in ONE launch
        (of_gemm_batch: the 512-wide projections' gradients of a gated block).  problems: [(A (K, M), B (K, N), out fp32 (M, N), beta,
        gate or None)].  Same bits as the separate ``gemm(A, B, out, ta=True, tb=True, epi=EPI_ACC_F32, ...)`` calls."""
        if not self.batch_dw:          # (constant True; A/B tooling clears it: the separate launches)
            for A, B, out, beta, gate in problems:
                self.gemm(A, B, out, ta=True, tb=True, epi=abi.EPI_ACC_F32, gate=gate, beta=beta)
            return
        n = len(problems)
        arr = (abi.OfGemmArgs * n)()
        need = []
        for a, (A, B, out, beta, gate) in zip(arr, problems):
            assert A.dtype == BF16 and B.dtype == BF16 and out.dtype == F32 and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1
            assert A.shape[0] == B.shape[0] and tuple(out.shape) == (A.shape[1], B.shape[1])
            a.A, a.B = A.data_ptr(), B.data_ptr()
            a.M, a.N, a.K = A.shape[1], B.shape[1], A.shape[0]
            a.lda, a.ldb = A.stride(0), B.stride(0)
            a.a_trans, a.b_trans, a.epi = 1, 1, abi.EPI_ACC_F32
            a.C, a.ldc = out.data_ptr(), out.stride(0)
            a.gate = _p(gate)
            a.alpha, a.beta = 1.0, float(beta)
            need.append((self.lib.of_gemm_workspace_bytes(C.byref(a)) + 255) // 256 * 256)
        total = sum(need)
        if total:
            ws = self.__dict__.get("_gemm_ws")
            if ws is None or ws.numel() * 4 < total or ws.device != problems[0][2].device:
                ws = torch.empty((total + 3) // 4, dtype=F32, device=problems[0][2].device)
                self._gemm_ws = ws
            off = 0
            for a, nb in zip(arr, need):
                if nb:
                    a.workspace, a.workspace_bytes = ws.data_ptr() + off, nb
                off += nb
        if self.gemm_timing is not None:
            key = (1, 1, abi.EPI_ACC_F32, "mid128batch")
            if self.gemm_timing_only is None or key in self.gemm_timing_only:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._chk(self.lib.of_gemm_batch(arr, n, self._stream()), "of_gemm_batch")
                e1.record()
                self.gemm_timing.append((key, sum(2.0 * a.M * a.N * a.K for a in arr), tuple((a.M, a.N, a.K) for a in arr), e0, e1))
                return
        self._chk(self.lib.of_gemm_batch(arr, n, self._stream()), "of_gemm_batch")

    def gemm_grouped(self, A, Bs, table, out, *, kind, epi=abi.EPI_STORE_BF16, beta=0.0):
        """One launch over several B matrices (OfGemmArgs.group_kind): kind 1 -- out[:, gE:(g+1)E] = A @ Bs[g]^T with
        Bs[g] (E, K); kind 2 -- out = sum_g A[:, gE:(g+1)E] @ Bs[g] with Bs[g] (E, N).  ``table``: int64 device tensor
        holding the data pointers of Bs (built by the caller, kept alive with them)."""
        E = Bs[0].shape[0]
        assert all(b.dtype == BF16 and b.shape == Bs[0].shape and b.stride() == Bs[0].stride() and b.stride(1) == 1 for b in Bs)
        assert A.dtype == BF16 and A.stride(1) == 1 and out.stride(1) == 1 and table.dtype == torch.int64 and table.numel() == len(Bs)
        a = abi.OfGemmArgs()
        a.A = A.data_ptr()
        a.M, a.lda = A.shape[0], A.stride(0)
        if kind == 1:
            a.N, a.K, a.b_trans = E * len(Bs), A.shape[1], 0
            assert Bs[0].shape[1] == a.K
        else:
            a.N, a.K, a.b_trans = Bs[0].shape[1], A.shape[1], 1
            assert a.K == E * len(Bs)
        assert tuple(out.shape) == (a.M, a.N)
        a.ldb = Bs[0].stride(0)
        a.epi, a.C, a.ldc = epi, out.data_ptr(), out.stride(0)
        a.alpha, a.beta = 1.0, float(beta)
        a.groups, a.group_kind, a.group_extent = table.data_ptr(), kind, E
        self._chk(self.lib.of_gemm(C.byref(a), self._stream()), "of_gemm(grouped)")
        return out

    # ------------------------------------------------------------------ LayerNorm
    def ln_fwd(self, x, w, b, y, stats):
        rows, dim = x.shape
        self._chk(self.lib.of_layernorm_fwd(x.data_ptr(), _is_f32(x), x.stride(0), w.data_ptr(), b.data_ptr(),
                                            y.data_ptr(), y.stride(0), _p(stats), rows, dim, self._stream()),
                  "of_layernorm_fwd")

    def ln_fwd_out(self, x, w, b, y, stats):
        rows, dim = x.shape
        self._chk(self.lib.of_layernorm_fwd_out(x.data_ptr(), _is_f32(x), x.stride(0), w.data_ptr(), b.data_ptr(),
                                                y.data_ptr(), _is_f32(y), y.stride(0), _p(stats), rows, dim,
                                                self._stream()), "of_layernorm_fwd_out")

    def ln_fwd_add(self, x, add, xsum, w, b, y, stats):
        """xsum = x + add (add bf16; xsum in x's dtype, may be x itself); y = LN(xsum) in y's dtype (bf16 / fp32)."""
        rows, dim = x.shape
        assert add.dtype == BF16 and add.shape == x.shape and xsum.dtype == x.dtype and xsum.shape == x.shape
        assert add.stride(1) == 1 and xsum.stride(1) == 1 and y.stride(1) == 1
        self._chk(self.lib.of_layernorm_fwd_add(x.data_ptr(), _is_f32(x), x.stride(0), add.data_ptr(), add.stride(0),
                                                xsum.data_ptr(), xsum.stride(0), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                                _is_f32(y), y.stride(0), _p(stats), rows, dim, self._stream()),
                  "of_layernorm_fwd_add")

    def ln_fwd_grouped(self, x, w, b, y_base, ldy, grp_rows, grp_stride, y2, stats):
        """y_base: bf16 tensor whose data_ptr is the first destination row."""
        rows, dim = x.shape
        self._chk(self.lib.of_layernorm_fwd_grouped(x.data_ptr(), _is_f32(x), x.stride(0), w.data_ptr(), b.data_ptr(),
                                                    y_base.data_ptr(), ldy, grp_rows, grp_stride, _p(y2), _p(stats),
                                                    rows, dim, self._stream()), "of_layernorm_fwd_grouped")

    def ln_bwd(self, dy, x, stats, w, *, lddy=None, dy_grp_rows=0, dy_grp_stride=0, dy2=None, resid=None, dx=None,
               dx_bf16=None, dw=None, db=None):
        rows, dim = x.shape
        lddy = dy.stride(0) if lddy is None else lddy
        out_f32 = _is_f32(dx) if dx is not None else (_is_f32(resid) if resid is not None else 1)
        lddx = dx.stride(0) if dx is not None else dim
        if dx_bf16 is not None:
            assert dx_bf16.stride(0) == lddx
        if resid is not None:
            assert resid.stride(0) == lddx
        ws, ws_bytes = None, 0
        if dw is not None:
            need = self.lib.of_layernorm_bwd_workspace_bytes(rows, dim)
            if need:
                ws = self.__dict__.get("_ln_ws")
                if ws is None or ws.numel() * 4 < need or ws.device != x.device:
                    ws = torch.empty((need + 3) // 4, dtype=F32, device=x.device)
                    self._ln_ws = ws
                ws_bytes = ws.numel() * 4
        self._chk(self.lib.of_layernorm_bwd(dy.data_ptr(), _is_f32(dy), lddy, dy_grp_rows, dy_grp_stride, _p(dy2),
                                            x.data_ptr(), _is_f32(x), x.stride(0), stats.data_ptr(), w.data_ptr(),
                                            _p(resid), _p(dx), out_f32, lddx, _p(dx_bf16), _p(dw), _p(db), rows, dim,
                                            _p(ws), ws_bytes, self._stream()), "of_layernorm_bwd")

    # ------------------------------------------------------------------ attention core
    def _attn_args(self, q, k, v, o, lse, batch, Lq, Lk, heads, text_time, n_per_media, T_img, only_immediate, scale,
                   safe, head_dim=64, causal=False, alibi_slopes=None, kv_len=None, head_valid=0):
        a = abi.OfAttnArgs()
        a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
        a.text_time = _p(text_time)
        a.batch, a.heads, a.Lq, a.Lk = batch, heads, Lq, Lk
        a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
        a.n_per_media, a.T_img, a.only_immediate = n_per_media, T_img, int(only_immediate)
        a.scale = scale
        a.safe = safe
        a.head_dim, a.causal, a.alibi_slopes, a.kv_len = head_dim, int(causal), _p(alibi_slopes), _p(kv_len)
        a.head_valid = head_valid
        return a

    def attn_fwd(self, q, k, v, o, lse, *, batch, Lq, Lk, heads, text_time=None, n_per_media=0, T_img=0,
                 only_immediate=True, scale=0.125, safe=0, head_dim=64, causal=False, alibi_slopes=None, kv_len=None, head_valid=0):
        """q,o: (batch*Lq, >=heads*head_dim) rows; k,v: (batch*Lk, ...) rows (views into a fused kv buffer are fine).
        head_valid (compact heads, include/of_hip.h): heads of head_valid < head_dim columns each, side by side."""
        a = self._attn_args(q, k, v, o, lse, batch, Lq, Lk, heads, text_time, n_per_media, T_img, only_immediate,
                            scale, safe, head_dim, causal, alibi_slopes, kv_len, head_valid)
        self._chk(self.lib.of_attn_fwd(C.byref(a), self._stream()), "of_attn_fwd")

    def attn_bwd(self, q, k, v, o, lse, dout, dq, dk, dv, delta, *, batch, Lq, Lk, heads, text_time=None,
                 n_per_media=0, T_img=0, only_immediate=True, scale=0.125, safe=0, head_dim=64, causal=False,
                 alibi_slopes=None, kv_len=None, head_valid=0):
        a = self._attn_args(q, k, v, o, lse, batch, Lq, Lk, heads, text_time, n_per_media, T_img, only_immediate,
                            scale, safe, head_dim, causal, alibi_slopes, kv_len, head_valid)
        a.dout, a.lddo = dout.data_ptr(), dout.stride(0)
        a.dq, a.lddq = dq.data_ptr(), dq.stride(0)
        a.dk, a.dv, a.lddk, a.lddv = dk.data_ptr(), dv.data_ptr(), dk.stride(0), dv.stride(0)
        a.delta = delta.data_ptr()
        self._chk(self.lib.of_attn_bwd(C.byref(a), self._stream()), "of_attn_bwd")

    # ------------------------------------------------------------------ fused attention branch of a gated block
    def pack_frag16(self, W, out=None):
        """Fragment-major copy of a row-major (N, K) bf16 matrix (of_pack_frag16): the weight operand of xattn_fused_fwd."""
        assert W.dtype == BF16 and W.dim() == 2 and W.stride(1) == 1
        N, K = W.shape
        out = torch.empty(N * K, dtype=BF16, device=W.device) if out is None else out
        assert out.dtype == BF16 and out.numel() >= N * K and out.is_contiguous()
        self._chk(self.lib.of_pack_frag16(W.data_ptr(), N, K, W.stride(0), out.data_ptr(), self._stream()), "of_pack_frag16")
        return out

    def pack_frag16_batch(self, pairs):
        """[(W (N, K) bf16 row-major, out flat bf16 of N * K elements)]: every fragment-major copy in one launch."""
        n = len(pairs)
        arr = (abi.OfPackDesc * n)()
        for d, (W, out) in zip(arr, pairs):
            assert W.dtype == BF16 and W.dim() == 2 and W.stride(1) == 1 and out.dtype == BF16 and out.is_contiguous() and out.numel() >= W.numel()
            d.W, d.P, d.N, d.K, d.ldw = W.data_ptr(), out.data_ptr(), W.shape[0], W.shape[1], W.stride(0)
        self._chk(self.lib.of_pack_frag16_batch(arr, n, self._stream()), "of_pack_frag16_batch")

    def xattn_fused_fwd(self, x, ln_w, ln_b, wq_pk, k, v, tt, wout_pk, gate, y, *, B, L, Lk, heads, head_dim, n_per_media, T_img,
                        only_immediate, scale, ln2_w=None, ln2_b=None, u2=None, st2=None, xn=None, st=None, q=None, o=None, lse=None,
                        probe_only=False):
        """The attention branch of a gated block as ONE launch (include/of_hip.h: of_xattn_fused_fwd).  Returns False -- nothing
        launched -- when the kernel does not take these shapes (the caller then runs the separate launches)."""
        a = abi.OfXattnFusedArgs()
        a.x, a.x_f32, a.ldx = x.data_ptr(), _is_f32(x), x.stride(0)
        a.ln_w, a.ln_b = ln_w.data_ptr(), ln_b.data_ptr()
        a.wq_pk, a.wout_pk = _p(wq_pk), _p(wout_pk)
        a.k, a.v, a.ldk, a.ldv = k.data_ptr(), v.data_ptr(), k.stride(0), v.stride(0)
        a.text_time = _p(tt)
        a.gate = _p(gate)
        a.ln2_w, a.ln2_b = _p(ln2_w), _p(ln2_b)
        a.xn, a.ldxn = _p(xn), (xn.stride(0) if xn is not None else 0)
        a.stats = _p(st)
        a.q, a.ldq = _p(q), (q.stride(0) if q is not None else 0)
        a.o, a.ldo = _p(o), (o.stride(0) if o is not None else 0)
        a.lse = _p(lse)
        a.y, a.ldy = y.data_ptr(), y.stride(0)
        a.u2, a.ldu2 = _p(u2), (u2.stride(0) if u2 is not None else 0)
        a.stats2 = _p(st2)
        a.B, a.L, a.Lk, a.d, a.heads, a.head_dim = B, L, Lk, x.shape[1], heads, head_dim
        a.n_per_media, a.T_img, a.only_immediate = n_per_media, T_img, int(only_immediate)
        a.scale = scale
        if probe_only:
            a.wq_pk = a.wout_pk = x.data_ptr()        # (any aligned non-null pointer: eligibility does not read them)
            return bool(self.lib.of_xattn_fused_eligible(C.byref(a)))
        if not self.lib.of_xattn_fused_eligible(C.byref(a)):
            return False
        self._chk(self.lib.of_xattn_fused_fwd(C.byref(a), self._stream()), "of_xattn_fused_fwd")
        return True

    def text_time(self, media_locations_u8, out_i32, Lq, use_cached):
        B, Lm = media_locations_u8.shape
        self._chk(self.lib.of_text_time(media_locations_u8.data_ptr(), out_i32.data_ptr(), B, Lm, Lq, int(use_cached),
                                        self._stream()), "of_text_time")

    # ------------------------------------------------------------------ element-wise
    def to_bf16(self, x, out=None):
        """Contiguous fp32 -> bf16 copy (GEMM operand); bf16 input is returned unchanged."""
        if x.dtype == BF16:
            return x
        assert x.dtype == F32 and x.is_contiguous()
        out = torch.empty(x.shape, dtype=BF16, device=x.device) if out is None else out
        self._chk(self.lib.of_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), self._stream()),
                  "of_cast_f32_to_bf16")
        return out

    def to_f32(self, x):
        assert x.dtype == BF16 and x.is_contiguous()
        out = torch.empty(x.shape, dtype=F32, device=x.device)
        self._chk(self.lib.of_cast_bf16_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), self._stream()),
                  "of_cast_bf16_to_f32")
        return out

    def broadcast_rows(self, src, out, rows):
        self._chk(self.lib.of_broadcast_rows(src.data_ptr(), src.shape[0], out.data_ptr(), _is_f32(out), out.stride(0),
                                             rows, src.shape[1], self._stream()), "of_broadcast_rows")

    def reduce_rows(self, src, dst):
        self._chk(self.lib.of_reduce_rows(src.data_ptr(), _is_f32(src), src.shape[0], src.shape[1], dst.data_ptr(),
                                          dst.shape[0], self._stream()), "of_reduce_rows")

    def add_embs(self, x, e1, inner1, outer1, e2, inner2, outer2, out):
        rows, dim = x.shape
        assert x.is_contiguous() and out.is_contiguous() and out.dtype == x.dtype
        self._chk(self.lib.of_add_embs(x.data_ptr(), _is_f32(x), _p(e1), inner1, outer1, _p(e2), inner2, outer2,
                                       out.data_ptr(), rows, dim, self._stream()), "of_add_embs")
        return out

    def reduce_rows_strided(self, src, inner, outer, dst):
        assert src.is_contiguous() and dst.dtype == F32 and dst.is_contiguous()
        self._chk(self.lib.of_reduce_rows_strided(src.data_ptr(), _is_f32(src), src.shape[0], src.shape[1], inner, outer,
                                                  dst.data_ptr(), self._stream()), "of_reduce_rows_strided")

    # ------------------------------------------------------------------ step epilogue
    SUMSQ_PARTS = abi.OF_SUMSQ_PARTS

    def sumsq_partial(self, g, partials, max_workgroups=0):
        """partials[0:SUMSQ_PARTS] = per-workgroup sums of g*g (deterministic: no floating-point atomics).  max_workgroups > 0: a
        narrow launch of that many fat workgroups (one per CU; the same bits)."""
        assert g.dtype == F32 and g.is_contiguous() and partials.dtype == F32 and partials.is_contiguous()
        assert partials.numel() >= self.SUMSQ_PARTS
        self._chk(self.lib.of_sumsq_partial_w(g.data_ptr(), g.numel(), partials.data_ptr(), int(max_workgroups), self._stream()),
                  "of_sumsq_partial")

    def sumsq_finish(self, partials, acc):
        """acc[0] = sum(partials) in a fixed order."""
        assert partials.dtype == F32 and partials.is_contiguous() and acc.dtype == F32
        self._chk(self.lib.of_sumsq_finish(partials.data_ptr(), partials.numel(), acc.data_ptr(), self._stream()), "of_sumsq_finish")

    def sumsq(self, bufs, acc, scratch=None, max_workgroups=0):
        """acc[0] = sum over the buffers of sum(g*g): one partial launch per buffer + one finish."""
        bufs = list(bufs)
        if scratch is None or scratch.numel() < len(bufs) * self.SUMSQ_PARTS:
            scratch = torch.empty(len(bufs) * self.SUMSQ_PARTS, dtype=F32, device=acc.device)
        for i, g in enumerate(bufs):
            self.sumsq_partial(g, scratch[i * self.SUMSQ_PARTS:(i + 1) * self.SUMSQ_PARTS], max_workgroups)
        self.sumsq_finish(scratch[:len(bufs) * self.SUMSQ_PARTS], acc)
        return scratch

    def step_advance(self, sumsq, applied):
        """applied[0] += 1 iff the global gradient norm is finite (the update will be applied): Adam's step count on the device."""
        assert applied.dtype == torch.int32 and sumsq.dtype == F32
        self._chk(self.lib.of_step_advance(sumsq.data_ptr(), applied.data_ptr(), self._stream()), "of_step_advance")

    def adamw_clip(self, p, g, m, v, sumsq, *, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=1.0,
                   p_bf16=None, zero_grad=True, grad_scale=1.0, applied=None, max_workgroups=0):
        """applied: optional device int32 count of applied updates (step_advance) that replaces ``step`` in the bias correction.
        max_workgroups > 0: a narrow launch of that many fat workgroups (one per CU; the same bits)."""
        n = p.numel()
        assert all(t.dtype == F32 and t.is_contiguous() and t.numel() == n for t in (p, g, m, v))
        self._chk(self.lib.of_adamw_clip_w(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(p_bf16), n,
                                           sumsq.data_ptr(), max_norm, lr, betas[0], betas[1], eps, weight_decay,
                                           grad_scale, step, int(zero_grad), _p(applied), int(max_workgroups), self._stream()),
                  "of_adamw_clip")

    # ------------------------------------------------------------------ causal-LM loss
    def ce_fwd(self, logits, labels, lse, loss_rows, ignore_index=-100):
        rows, V = logits.shape
        assert logits.stride(1) == 1 and labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == rows
        self._chk(self.lib.of_ce_fwd(logits.data_ptr(), _is_f32(logits), logits.stride(0), labels.data_ptr(), ignore_index,
                                     rows, V, lse.data_ptr(), loss_rows.data_ptr(), self._stream()), "of_ce_fwd")

    def ce_bwd(self, logits, labels, lse, gscale, dlogits, ignore_index=-100):
        rows, V = logits.shape
        assert dlogits.dtype == logits.dtype and dlogits.stride(1) == 1 and gscale.dtype == F32
        self._chk(self.lib.of_ce_bwd(logits.data_ptr(), _is_f32(logits), logits.stride(0), labels.data_ptr(), ignore_index,
                                     rows, V, lse.data_ptr(), gscale.data_ptr(), dlogits.data_ptr(), dlogits.stride(0),
                                     self._stream()), "of_ce_bwd")

    def quick_gelu(self, x, out=None):
        assert x.dtype == BF16 and x.is_contiguous()
        out = torch.empty_like(x) if out is None else out
        self._chk(self.lib.of_quick_gelu(x.data_ptr(), out.data_ptr(), x.numel(), self._stream()), "of_quick_gelu")
        return out

    def gelu_fwd(self, x, out=None):
        assert x.dtype == BF16 and x.is_contiguous()
        out = torch.empty_like(x) if out is None else out
        self._chk(self.lib.of_gelu_fwd(x.data_ptr(), out.data_ptr(), x.numel(), self._stream()), "of_gelu_fwd")
        return out

    def gelu_bwd(self, dy, x, out=None):
        """dy * gelu'(x); ``out`` may be ``dy`` (in place)."""
        assert x.dtype == BF16 and dy.dtype == BF16 and x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape
        out = torch.empty_like(x) if out is None else out
        self._chk(self.lib.of_gelu_bwd(dy.data_ptr(), x.data_ptr(), out.data_ptr(), x.numel(), self._stream()), "of_gelu_bwd")
        return out

    def add_bf16(self, a, b, out=None):
        """fp32 stream + bf16 branch output -> fp32."""
        assert a.dtype == F32 and b.dtype == BF16 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
        out = torch.empty_like(a) if out is None else out
        self._chk(self.lib.of_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), self._stream()), "of_add_bf16")
        return out

    def rotary_neox(self, qkv, cos, sin, q, k, v, *, L, heads, head_size, rot_dims, head_pad, inverse=False):
        """HF GPT-NeoX rotary embedding + per-head zero padding (forward: qkv -> q, k, v; inverse: padded dq, dk, dv -> d(qkv))."""
        rows = qkv.shape[0]
        assert qkv.dtype == BF16 and qkv.stride(1) == 1 and all(t.dtype == BF16 and t.stride(1) == 1 and t.stride(0) == q.stride(0)
                                                              for t in (q, k, v))
        assert cos.dtype == F32 and sin.dtype == F32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (L, rot_dims)
        self._chk(self.lib.of_rotary_neox(qkv.data_ptr(), qkv.stride(0), cos.data_ptr(), sin.data_ptr(), L, q.data_ptr(), k.data_ptr(),
                                          v.data_ptr(), q.stride(0), rows, heads, head_size, rot_dims, head_pad, int(inverse),
                                          self._stream()), "of_rotary_neox")

    def head_repack(self, src, dst, *, heads, src_head_size, dst_head_size):
        assert src.dtype == BF16 and dst.dtype == BF16 and src.stride(1) == 1 and dst.stride(1) == 1 and src.shape[0] == dst.shape[0]
        self._chk(self.lib.of_head_repack(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), src.shape[0], heads,
                                          src_head_size, dst_head_size, self._stream()), "of_head_repack")
        return dst

    def add(self, a, b, out):
        assert a.dtype == b.dtype == out.dtype and a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
        self._chk(self.lib.of_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), _is_f32(a), a.numel(), self._stream()),
                  "of_add")
        return out
