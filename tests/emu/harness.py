"""TEST-ONLY loader for the host-emulator build of the kernels (tests/emu/libofhip_emu.so).

Builds it on demand with the host clang and exposes the C ABI on CPU tensors, so kernel index logic can be
checked against the oracle in the no-GPU container.  Never imported from open_flamingo_amd/."""
import ctypes as C
import os

import numpy as np
import torch

from open_flamingo_amd.csrc import build as _build
from open_flamingo_amd.hip import abi

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build(emu=True)
        _lib = C.CDLL(path)
        abi.declare(_lib, require_all=False)
        assert _lib.of_build_kind() == 2
    return _lib


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def bf16(t):
    return t.to(torch.bfloat16).contiguous()


def gemm(A, B, *, a_trans=0, b_trans=0, epi=abi.EPI_STORE_BF16, C_out=None, C2=None, aux=None, gate=None,
         alpha=1.0, beta=0.0, dot_out=None, io_f32=0, safe=0, M=None, N=None, K=None, workspace=None, cu_limit=0, sumsq=None):
    if M is None:
        M = A.shape[1] if a_trans else A.shape[0]
        K = A.shape[0] if a_trans else A.shape[1]
        N = B.shape[1] if b_trans else B.shape[0]
    a = abi.OfGemmArgs()
    a.A, a.B = A.data_ptr(), B.data_ptr()
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb = A.stride(0), B.stride(0)
    a.a_trans, a.b_trans, a.epi = a_trans, b_trans, epi
    a.C, a.ldc = C_out.data_ptr(), C_out.stride(0)
    a.C2 = C2.data_ptr() if C2 is not None else None
    a.aux = aux.data_ptr() if aux is not None else None
    a.ldaux = aux.stride(0) if aux is not None else 0
    a.gate = gate.data_ptr() if gate is not None else None
    a.alpha, a.beta = alpha, beta
    a.dot_out = dot_out.data_ptr() if dot_out is not None else None
    a.io_f32, a.safe, a.cu_limit = io_f32, safe, cu_limit
    if sumsq is not None:
        assert lib().of_gemm_sumsq_slots(C.byref(a)) <= sumsq.numel()
        a.sumsq_out = sumsq.data_ptr()
    if workspace is None and (dot_out is not None or cu_limit or safe == 17):
        # per-workgroup gate-gradient partials (deterministic finish); stream-K partial tiles + flags
        workspace = torch.empty(max(1, lib().of_gemm_workspace_bytes(C.byref(a)) // 4))
    if workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    rc = lib().of_gemm(C.byref(a), None)
    assert rc == 0, f"of_gemm rc={rc}"
    return C_out


def attn_args(q, k, v, o, lse, text_time=None, n_per_media=0, T_img=0, only_immediate=1, heads=None, safe=0,
              dout=None, dq=None, dk=None, dv=None, delta=None, head_dim=64, causal=0, alibi_slopes=None, kv_len=None, head_valid=0,
              scale=None):
    """q (batch,Lq,H*64) bf16; k,v (batch,Lk,H*64) bf16 (may be views into a fused kv buffer)."""
    a = abi.OfAttnArgs()
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    a.text_time = text_time.data_ptr() if text_time is not None else None
    a.batch, a.Lq, a.Lk = q.shape[0], q.shape[1], k.shape[1]
    a.heads = heads if heads is not None else q.shape[2] // (head_valid or head_dim)
    a.head_valid = head_valid
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1), k.stride(1), v.stride(1), o.stride(1)
    a.n_per_media, a.T_img, a.only_immediate = n_per_media, T_img, only_immediate
    a.scale = scale if scale is not None else head_dim ** -0.5
    a.safe = safe
    a.head_dim, a.causal = head_dim, causal
    a.alibi_slopes = alibi_slopes.data_ptr() if alibi_slopes is not None else None
    a.kv_len = kv_len.data_ptr() if kv_len is not None else None
    if dout is not None:
        a.dout, a.lddo = dout.data_ptr(), dout.stride(1)
        a.dq, a.lddq = dq.data_ptr(), dq.stride(1)
        a.dk, a.dv, a.lddk, a.lddv = dk.data_ptr(), dv.data_ptr(), dk.stride(1), dv.stride(1)
        a.delta = delta.data_ptr()
    return a


def attn_fwd(a):
    rc = lib().of_attn_fwd(C.byref(a), None)
    assert rc == 0, f"of_attn_fwd rc={rc}"


def attn_bwd(a):
    rc = lib().of_attn_bwd(C.byref(a), None)
    assert rc == 0, f"of_attn_bwd rc={rc}"


def emu_ops():
    """Ops bound to the emulator library (CPU tensors, no stream).  Tests only."""
    from open_flamingo_amd.hip.ops import Ops
    return Ops(lib(), lambda: None)
