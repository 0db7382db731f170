"""ctypes mirror of include/of_hip.h (struct layouts + prototypes).

Pure declarations: nothing here loads a library.  ``open_flamingo_amd.hip.lib`` applies them to the gfx950
``libofhip.so``; the test-only emulator harness applies them to its own build.
"""
import ctypes as C

OF_ABI_VERSION = 11
OF_SUMSQ_PARTS = 512
EPI_STORE_BF16, EPI_GELU, EPI_GATE_RESID, EPI_DGELU_DOT, EPI_SCALE_DOT, EPI_ACC_F32 = range(6)

vp = C.c_void_p


class OfGemmArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int),
        ("a_trans", C.c_int), ("b_trans", C.c_int),
        ("epi", C.c_int),
        ("C", vp), ("ldc", C.c_int),
        ("C2", vp),
        ("aux", vp), ("ldaux", C.c_int),
        ("gate", vp),
        ("alpha", C.c_float), ("beta", C.c_float),
        ("dot_out", vp),
        ("io_f32", C.c_int), ("safe", C.c_int), ("ksplit", C.c_int),
        ("workspace", vp), ("workspace_bytes", C.c_size_t),
        ("groups", vp), ("group_kind", C.c_int), ("group_extent", C.c_int),
        ("cu_limit", C.c_int), ("sk_grid", C.c_int),
        ("sumsq_out", vp),
    ]


class OfAttnArgs(C.Structure):
    _fields_ = [
        ("q", vp), ("k", vp), ("v", vp), ("o", vp), ("lse", vp), ("text_time", vp),
        ("batch", C.c_int), ("heads", C.c_int), ("Lq", C.c_int), ("Lk", C.c_int),
        ("ldq", C.c_long), ("ldk", C.c_long), ("ldv", C.c_long), ("ldo", C.c_long),
        ("n_per_media", C.c_int), ("T_img", C.c_int), ("only_immediate", C.c_int),
        ("scale", C.c_float),
        ("dout", vp), ("lddo", C.c_long),
        ("dq", vp), ("lddq", C.c_long),
        ("dk", vp), ("dv", vp), ("lddk", C.c_long), ("lddv", C.c_long),
        ("delta", vp),
        ("safe", C.c_int),
        ("head_dim", C.c_int), ("causal", C.c_int), ("alibi_slopes", vp), ("kv_len", vp),
        ("head_valid", C.c_int),
    ]


class OfXattnFusedArgs(C.Structure):
    _fields_ = [
        ("x", vp), ("x_f32", C.c_int), ("ldx", C.c_long),
        ("ln_w", vp), ("ln_b", vp),
        ("wq_pk", vp),
        ("k", vp), ("v", vp), ("ldk", C.c_long), ("ldv", C.c_long),
        ("text_time", vp),
        ("wout_pk", vp),
        ("gate", vp),
        ("ln2_w", vp), ("ln2_b", vp),
        ("xn", vp), ("ldxn", C.c_long),
        ("stats", vp),
        ("q", vp), ("ldq", C.c_long),
        ("o", vp), ("ldo", C.c_long),
        ("lse", vp),
        ("y", vp), ("ldy", C.c_long),
        ("u2", vp), ("ldu2", C.c_long),
        ("stats2", vp),
        ("B", C.c_int), ("L", C.c_int), ("Lk", C.c_int), ("d", C.c_int), ("heads", C.c_int), ("head_dim", C.c_int),
        ("n_per_media", C.c_int), ("T_img", C.c_int), ("only_immediate", C.c_int),
        ("scale", C.c_float),
    ]


class OfPackDesc(C.Structure):
    _fields_ = [("W", vp), ("P", vp), ("N", C.c_int), ("K", C.c_int), ("ldw", C.c_long)]


PROTOTYPES = {
    "of_abi_version": (C.c_int, []),
    "of_build_kind": (C.c_int, []),
    "of_gemm": (C.c_int, [C.POINTER(OfGemmArgs), vp]),
    "of_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(OfGemmArgs)]),
    "of_gemm_sumsq_slots": (C.c_size_t, [C.POINTER(OfGemmArgs)]),
    "of_gemm_batch": (C.c_int, [C.POINTER(OfGemmArgs), C.c_int, vp]),
    "of_layernorm_fwd": (C.c_int, [vp, C.c_int, C.c_long, vp, vp, vp, C.c_long, vp, C.c_long, C.c_int, vp]),
    "of_layernorm_fwd_out": (C.c_int, [vp, C.c_int, C.c_long, vp, vp, vp, C.c_int, C.c_long, vp, C.c_long,
                                       C.c_int, vp]),
    "of_layernorm_fwd_add": (C.c_int, [vp, C.c_int, C.c_long, vp, C.c_long, vp, C.c_long, vp, vp, vp, C.c_int, C.c_long, vp,
                                       C.c_long, C.c_int, vp]),
    "of_layernorm_fwd_grouped": (C.c_int, [vp, C.c_int, C.c_long, vp, vp, vp, C.c_long, C.c_long, C.c_long, vp, vp,
                                           C.c_long, C.c_int, vp]),
    "of_layernorm_bwd": (C.c_int, [vp, C.c_int, C.c_long, C.c_long, C.c_long, vp, vp, C.c_int, C.c_long, vp, vp, vp,
                                   vp, C.c_int, C.c_long, vp, vp, vp, C.c_long, C.c_int, vp, C.c_size_t, vp]),
    "of_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_long, C.c_int]),
    "of_attn_fwd": (C.c_int, [C.POINTER(OfAttnArgs), vp]),
    "of_attn_bwd": (C.c_int, [C.POINTER(OfAttnArgs), vp]),
    "of_xattn_fused_eligible": (C.c_int, [C.POINTER(OfXattnFusedArgs)]),
    "of_xattn_fused_fwd": (C.c_int, [C.POINTER(OfXattnFusedArgs), vp]),
    "of_pack_frag16": (C.c_int, [vp, C.c_int, C.c_int, C.c_long, vp, vp]),
    "of_pack_frag16_batch": (C.c_int, [C.POINTER(OfPackDesc), C.c_int, vp]),
    "of_text_time": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "of_cast_f32_to_bf16": (C.c_int, [vp, vp, C.c_long, vp]),
    "of_cast_bf16_to_f32": (C.c_int, [vp, vp, C.c_long, vp]),
    "of_broadcast_rows": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_long, C.c_long, C.c_int, vp]),
    "of_reduce_rows": (C.c_int, [vp, C.c_int, C.c_long, C.c_int, vp, C.c_int, vp]),
    "of_add": (C.c_int, [vp, vp, vp, C.c_int, C.c_long, vp]),
    "of_quick_gelu": (C.c_int, [vp, vp, C.c_long, vp]),
    "of_gelu_fwd": (C.c_int, [vp, vp, C.c_long, vp]),
    "of_gelu_bwd": (C.c_int, [vp, vp, vp, C.c_long, vp]),
    "of_add_bf16": (C.c_int, [vp, vp, vp, C.c_long, vp]),
    "of_sumsq_partial": (C.c_int, [vp, C.c_long, vp, vp]),
    "of_sumsq_finish": (C.c_int, [vp, C.c_long, vp, vp]),
    "of_adamw_clip": (C.c_int, [vp, vp, vp, vp, vp, C.c_long, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_int, C.c_int, vp, vp]),
    "of_step_advance": (C.c_int, [vp, vp, vp]),
    "of_sumsq_partial_w": (C.c_int, [vp, C.c_long, vp, C.c_int, vp]),
    "of_adamw_clip_w": (C.c_int, [vp, vp, vp, vp, vp, C.c_long, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_int, C.c_int, vp, C.c_int, vp]),
    "of_rotary_neox": (C.c_int, [vp, C.c_long, vp, vp, C.c_long, vp, vp, vp, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, vp]),
    "of_head_repack": (C.c_int, [vp, C.c_long, vp, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, vp]),
    "of_add_embs": (C.c_int, [vp, C.c_int, vp, C.c_long, C.c_int, vp, C.c_long, C.c_int, vp, C.c_long, C.c_int, vp]),
    "of_ce_fwd": (C.c_int, [vp, C.c_int, C.c_long, vp, C.c_longlong, C.c_long, C.c_int, vp, vp, vp]),
    "of_ce_bwd": (C.c_int, [vp, C.c_int, C.c_long, vp, C.c_longlong, C.c_long, C.c_int, vp, vp, vp, C.c_long, vp]),
    "of_reduce_rows_strided": (C.c_int, [vp, C.c_int, C.c_long, C.c_int, C.c_long, C.c_int, vp, vp]),
}


def declare(lib, require_all=True):
    """Attach restype/argtypes for every symbol of include/of_hip.h; raise if one is missing."""
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and require_all:
        raise RuntimeError(f"libofhip is missing symbols declared in include/of_hip.h: {missing}")
    return missing
