"""ctypes binding of the C ABI in include/qeft_hip.h (qeft_amd/lib/libqeft_hip.so).

There is no CPU fallback: if the library is missing, or an entry point fails, the call raises.
"""
import ctypes
import os

# torch must be in the process BEFORE the library is loaded: torch ships its own libamdhip64 and the library links
# the system one under the same soname.  Whichever loads first serves both; if it is the system copy, torch's
# streams / allocations and this library's kernels end up in two HIP runtimes and every launch fails
# (hipErrorNoDevice).  Loading torch first makes its runtime the only one.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# QEFT_HIP_LIB: load another build of the same ABI (A/B timing of two builds on one GPU box); never a fallback
LIB_PATH = os.environ.get("QEFT_HIP_LIB") or os.path.join(_HERE, "lib", "libqeft_hip.so")

_p, _i = ctypes.c_void_p, ctypes.c_int

# name -> argtypes, exactly as declared in include/qeft_hip.h
SIGNATURES = {
    "qeft_abi_version": [],
    "qeft_error_string": [_i],
    "qeft_last_hip_error": [],
    "qeft_last_variant": [],
    "qeft_gemv_w4": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "qeft_gemv_w4_qeft": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_gemv_w4_fused": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_pack_scales": [_p, _p, _p, _i, _i, _i, _p],
    "qeft_gemm_w4": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_gemm_w4_workspace_bytes": [_i, _i, _i, _i],
    "qeft_gemm_w4_silu_mul": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_gemm_w4_gateup_supported": [_i, _i, _i, _i, _i],
    "qeft_gemm_w4_gateup": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_gemm_w4_ws": [_p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_longlong, _i, _i, _i, _i, _i, _p],
    "qeft_gemm_w4_dx": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_gemm_w4_dx_workspace_bytes": [_i, _i, _i],
    "qeft_gemm_w4_dx_ws": [_p, _p, _p, _p, _p, _p, _p, ctypes.c_longlong, _i, _i, _i, _i, _i, _p],
    "qeft_grad_oweight": [_p, _p, _p, _i, _i, _i, _i, _p],
    "qeft_dequant_w4": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "qeft_pack_oweight": [_p, _p, _i, _i, _p],
    "qeft_gemv_w4_group": [_p, _p, ctypes.c_float, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "qeft_gemv_w4_silu": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "qeft_rmsnorm": [_p, _p, _p, _p, _p, _i, _i, ctypes.c_float, _p],
    "qeft_silu_mul": [_p, _p, _p, _i, _p],
    "qeft_gemv_w3": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_gemv_w3_group": [_p, _p, ctypes.c_float, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "qeft_gemv_w3_silu": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "qeft_expand_w3": [_p, _p, _i, _i, _i, _p],
    "qeft_gemm_w3_supported": [_i, _i, _i, _i, _i],
    "qeft_gemm_w3": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_gemm_w3_dx_supported": [_i, _i, _i, _i, _i],
    "qeft_gemm_w3_dx": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "qeft_attn_workspace_bytes": [_i, _i],
    "qeft_token_begin": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "qeft_token_end": [_p, _p, _p, _i, _i, _p],
    "qeft_decode_linear_blocks": [_i],
    "qeft_decode_linear": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, ctypes.c_float, _p, _p, _p, _p],
    "qeft_decode_linear_w3": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, ctypes.c_float, _p, _p, _p, _p],
    "qeft_gemv_v3_check_extents": [_i, _i, _i, _i, _i, _i],
    "qeft_gemv_v3_check_extents_ckpt": [_i, _i, _i, _i, _i, _i, _i],
    "qeft_decode_linear_hnorm": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, ctypes.c_float, _p],
    "qeft_token_begin_norm_blocks": [_i],
    "qeft_token_begin_norm": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "qeft_rmsnorm_f32": [_p, _p, _p, _i, _i, ctypes.c_float, _p],
    "qeft_lm_head_f16": [_p, _p, _p, _p, _i, _i, ctypes.c_float, _p],
    "qeft_rope_rows": [_p, _p, _p, _i, _i, _i, _p],
    "qeft_residual_norm": [_p, _p, _p, _p, _p, _p, _i, _p],
    "qeft_single_query_attention": [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p],
    "qeft_single_query_attention_alibi": [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p],
    "qeft_single_query_attention_generic": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "qeft_rope_attn_decode": [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "qeft_oneshot_mailbox_bytes": [_i, _i],
    "qeft_oneshot_max_world": [],
    "qeft_oneshot_mailbox_alloc": [_i, _i, ctypes.POINTER(ctypes.c_void_p)],
    "qeft_oneshot_mailbox_free": [_p],
    "qeft_oneshot_ipc_export": [_p, _p],
    "qeft_oneshot_ipc_open": [_p, ctypes.POINTER(ctypes.c_void_p)],
    "qeft_oneshot_ipc_close": [_p],
    "qeft_oneshot_allreduce_f32": [_p, _i, ctypes.POINTER(ctypes.c_void_p), _i, _i, _p, _p, _p],
}

_lib = None


class QeftHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the CDLL.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QeftHipError(
                f"{LIB_PATH} not found: build it with `python -m qeft_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the quantized linear.")
        l = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name, None)
            if fn is None:
                # only an explicitly selected A/B build (QEFT_HIP_LIB) may be older than this table; the in-tree library must
                # export every entry (tests/test_cabi.py checks it against include/qeft_hip.h as well)
                if os.environ.get("QEFT_HIP_LIB"):
                    continue
                raise QeftHipError(f"{LIB_PATH} does not export {name}: rebuild it (python -m qeft_amd.build)")
            fn.argtypes = argtypes
            fn.restype = (ctypes.c_char_p if name in ("qeft_error_string", "qeft_last_variant") else
                          ctypes.c_longlong if name in ("qeft_gemm_w4_workspace_bytes", "qeft_gemm_w4_dx_workspace_bytes",
                                                       "qeft_gemv_v3_check_extents", "qeft_gemv_v3_check_extents_ckpt",
                                                       "qeft_oneshot_mailbox_bytes") else _i)
        _lib = l
    return _lib


def last_variant():
    """Kernel variant the calling thread's last compute entry dispatched to (qeft_last_variant)."""
    return lib().qeft_last_variant().decode()


def check(code):
    if code != 0:
        l = lib()
        msg = l.qeft_error_string(code).decode()
        if code == 5:
            msg += f" (hipError_t={l.qeft_last_hip_error()})"
        raise QeftHipError(msg)
