"""Python operator module with the reference extension's surface (`import qeft_cuda`).

Mirrors the functions the reference binds in qeft/kernel/qeft_cuda.cpp:10-27 for the quantized
linear — same names, positional arguments and error behaviour — implemented by ctypes calls into
the gfx950 C ABI (include/qeft_hip.h).  Outputs are freshly allocated on the input's device with
the input's dtype (gemv_cuda_qeft.cu:404-414, gemm_cuda.cu:935-950).  Unlike the reference the
kernels are launched on torch's *current* stream of the input's device, and shapes / dtypes /
contiguity are validated (the reference validates nothing).
"""
import torch

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_common(in_feats, kernel, scales, zeros):
    _need(in_feats.is_cuda, "in_feats must be a GPU tensor (no CPU fallback)")
    _need(in_feats.dtype == torch.float16, "expected scalar type Half for in_feats")
    _need(kernel.dtype == torch.int16, "expected scalar type Short for kernel")
    _need(scales.dtype == torch.float16 and zeros.dtype == torch.float16, "expected scalar type Half for scales/zeros")
    for t, n in ((kernel, "kernel"), (scales, "scales"), (zeros, "zeros")):
        _need(t.device == in_feats.device, f"{n} is on {t.device}, in_feats on {in_feats.device}")
        _need(t.is_contiguous(), f"{n} must be contiguous")


# ---- the packed-scale shadow of the reference's gemv entries.  gemv_4bit[_qeft] receive the checkpoint's scales / scaled_zeros
# [K/g][N]; the MFMA GEMV wants (scale | scaled_zero << 16) words per row set.  Without a shadow every call re-packs them in LDS
# (+6..7 % at m = 1).  The shim keeps one derived buffer per (scales, zeros) PAIR OF TENSOR OBJECTS, valid
# while both objects are alive, are the same Python objects and carry the same in-place version counters -- a module's buffers
# are exactly that across calls, and anything else (a new view, an in-place update, a freed and re-used address) misses and
# re-packs.  QEFT_SHIM_SZP_CACHE=0 disables it.  Never saved: state_dict is untouched.
import os as _os
import weakref as _weakref

_SZP_CACHE = {}
_SZP_CACHE_ON = _os.environ.get("QEFT_SHIM_SZP_CACHE", "1") != "0"
_SZP_CACHE_MAX = 4096


def _szp_shadow(scales, zeros, n, k, group_size):
    if not _SZP_CACHE_ON:
        return None
    key = (id(scales), id(zeros))
    stamp = (scales._version, zeros._version, scales.data_ptr(), zeros.data_ptr(), n, k, group_size)      # (.data = ... keeps object and version)
    ent = _SZP_CACHE.get(key)
    if ent is not None and ent[0]() is scales and ent[1]() is zeros and ent[2] == stamp:
        return ent[3]
    if torch.cuda.is_current_stream_capturing():
        return None     # (a miss inside a capture would bake a one-off pack kernel into the graph: that launch packs in LDS instead)
    szp = pack_scales(scales, zeros, n, k, group_size)
    if szp is None:
        return None
    if len(_SZP_CACHE) >= _SZP_CACHE_MAX:
        _SZP_CACHE.clear()

    def _drop(_ref, key=key):
        _SZP_CACHE.pop(key, None)
    _SZP_CACHE[key] = (_weakref.ref(scales, _drop), _weakref.ref(zeros, _drop), stamp, szp)
    return szp


def gemv_4bit(in_feats, kernel, scaling_factors, zeros, m, n, k, group_size):
    """gemv_cuda.cu:358-525.  m in 1..7 else RuntimeError("Unsupported batch size for gemv kernel.")."""
    _need(1 <= int(m) <= 7, "Unsupported batch size for gemv kernel.")
    _check_common(in_feats, kernel, scaling_factors, zeros)
    x = in_feats.contiguous()
    _need(x.numel() == m * k, f"in_feats has {x.numel()} elements, expected m*k = {m * k}")
    _need(kernel.shape == (n // 4, k), f"kernel shape {tuple(kernel.shape)} != ({n // 4}, {k})")
    out = torch.empty(*in_feats.shape[:-1], n, dtype=in_feats.dtype, device=in_feats.device)
    szp = _szp_shadow(scaling_factors, zeros, n, k, group_size)
    with torch.cuda.device(in_feats.device):
        if szp is not None:       # the same launch with the shadow in place of the per-call LDS packing
            _lib.check(_lib.lib().qeft_gemv_w4_fused(x.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(),
                                                     None, None, None, None, szp.data_ptr(), out.data_ptr(), m, n, k, group_size, 0,
                                                     _stream(x)))
        else:
            _lib.check(_lib.lib().qeft_gemv_w4(x.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(),
                                               zeros.data_ptr(), out.data_ptr(), m, n, k, group_size, _stream(x)))
    return out


def gemv_4bit_qeft(in_feats, kernel, scaling_factors, zeros, oweight, m, n, k, group_size):
    """gemv_cuda_qeft.cu:392-513.  `oweight` is the interleaved [N/2, 2r] buffer; r = oweight.size(1)/2 (:424).
    m outside 1..7: RuntimeError("Unsupported batch size for gemv kernel.") (:466)."""
    _need(1 <= int(m) <= 7, "Unsupported batch size for gemv kernel.")
    _check_common(in_feats, kernel, scaling_factors, zeros)
    _need(oweight.dtype == torch.float16 and oweight.is_contiguous() and oweight.device == in_feats.device,
          "oweight_interleaved must be a contiguous Half tensor on the input's device")
    x = in_feats.contiguous()
    _need(x.numel() == m * k, f"in_feats has {x.numel()} elements, expected m*k = {m * k}")
    _need(kernel.shape == (n // 4, k), f"kernel shape {tuple(kernel.shape)} != ({n // 4}, {k})")
    n_out = oweight.shape[1] // 2
    _need(oweight.shape[0] == n // 2, f"oweight_interleaved has {oweight.shape[0]} rows, expected {n // 2}")
    out = torch.empty(*in_feats.shape[:-1], n, dtype=in_feats.dtype, device=in_feats.device)
    szp = _szp_shadow(scaling_factors, zeros, n, k, group_size)
    with torch.cuda.device(in_feats.device):
        if szp is not None:
            _lib.check(_lib.lib().qeft_gemv_w4_fused(x.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(),
                                                     oweight.data_ptr() if n_out else None, None, None, None, szp.data_ptr(),
                                                     out.data_ptr(), m, n, k, group_size, n_out, _stream(x)))
        else:
            _lib.check(_lib.lib().qeft_gemv_w4_qeft(x.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(),
                                                    zeros.data_ptr(), oweight.data_ptr(), out.data_ptr(), m, n, k,
                                                    group_size, n_out, _stream(x)))
    return out


def gemm_4bit(in_feats, kernel, scales, zeros):
    """gemm_cuda.cu:929-1033: N = kernel.size(0)*4 (:936), M = numel/K (:937); every column from the nibbles."""
    return gemm_4bit_qeft(in_feats, kernel, scales, zeros, None)


def gemm_4bit_qeft(in_feats, kernel, scales, zeros, oweights, bias=None):
    """The fused GEMM gemm_cuda_qeft.cu:1003 declared but never shipped: the last r = oweights.size(1) input
    columns come from the plain fp16 `oweights` [N, r] inside the same launch (qlinear.py:266 adds them with a
    second F.linear).  oweights=None gives gemm_4bit.  Extension: optional fused bias."""
    _check_common(in_feats, kernel, scales, zeros)
    x = in_feats.contiguous()
    k = x.shape[-1]
    n = kernel.shape[0] * 4
    m = x.numel() // k
    _need(kernel.shape[1] == k, f"kernel has K={kernel.shape[1]}, in_feats has K={k}")
    group = k // scales.shape[0]
    n_out = 0
    if oweights is not None:
        _need(oweights.dtype == torch.float16 and oweights.is_contiguous() and oweights.shape[0] == n,
              "oweights must be a contiguous Half [N, r] tensor")
        n_out = oweights.shape[1]
    out = torch.empty(*in_feats.shape[:-1], n, dtype=in_feats.dtype, device=in_feats.device)
    if m == 0:
        return out
    with torch.cuda.device(in_feats.device):
        lib = _lib.lib()
        st = _stream(x)
        need = lib.qeft_gemm_w4_workspace_bytes(m, n, k, n_out)
        if need > 0:
            # mid-size m: split-K through a scratch buffer (one per device and stream: launches that share it are
            # ordered by that stream)
            key = (x.device.index, st)
            ws = _GEMM_WS.get(key)
            if ws is None or ws.numel() * 4 < need:
                ws = _GEMM_WS[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
            _lib.check(lib.qeft_gemm_w4_ws(x.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                           oweights.data_ptr() if n_out else None,
                                           bias.data_ptr() if bias is not None else None, out.data_ptr(), ws.data_ptr(),
                                           ws.numel() * 4, m, n, k, group, n_out, st))
        else:
            _lib.check(lib.qeft_gemm_w4(x.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                        oweights.data_ptr() if n_out else None,
                                        bias.data_ptr() if bias is not None else None, out.data_ptr(), m, n, k,
                                        group, n_out, st))
    return out


def gemm_gateup_supported(m, op):
    """Whether gemm_4bit_gateup takes `m` rows of the fuse.pair64_gemm_operand `op`."""
    return bool(_lib.lib().qeft_gemm_w4_gateup_supported(m, op.outfeatures, op.infeatures, op.group_size, op.outlierfeatures))


def gemm_4bit_gateup(in_feats, op):
    """silu(gate_proj(x)) * up_proj(x) from ONE launch: `op` = fuse.pair64_gemm_operand(gate_proj, up_proj).  Extension."""
    x = in_feats.contiguous()
    k = x.shape[-1]
    _need(x.dtype == torch.float16 and k == op.infeatures, "in_feats must be Half [..., K] of the operand's K")
    m = x.numel() // k
    out = torch.empty(*in_feats.shape[:-1], op.outfeatures // 2, dtype=in_feats.dtype, device=in_feats.device)
    if m == 0:
        return out
    with torch.cuda.device(in_feats.device):
        _lib.check(_lib.lib().qeft_gemm_w4_gateup(x.data_ptr(), op.qweight.data_ptr(), op.scales.data_ptr(),
                                                  op.scaled_zeros.data_ptr(),
                                                  op.oweight.data_ptr() if op.oweight is not None else None,
                                                  op.bias.data_ptr() if op.bias is not None else None, out.data_ptr(), m,
                                                  op.outfeatures, k, op.group_size, op.outlierfeatures, _stream(x)))
    return out


_GEMM_WS = {}   # (device index, stream) -> fp32 scratch of the split-K GEMM


def gemm_4bit_qeft_silu_mul(in_feats, kernel, scales, zeros, oweights, gate, bias=None):
    """silu(gate) * gemm_4bit_qeft(in_feats, ...): the MLP's act_fn(gate_proj(x)) * up_proj(x) with the activation in the
    up_proj GEMM's epilogue (one launch on the 256-row tier; GEMM + qeft_silu_mul in place otherwise).  Extension."""
    _check_common(in_feats, kernel, scales, zeros)
    x = in_feats.contiguous()
    k = x.shape[-1]
    n = kernel.shape[0] * 4
    m = x.numel() // k
    _need(kernel.shape[1] == k, f"kernel has K={kernel.shape[1]}, in_feats has K={k}")
    _need(gate.dtype == torch.float16 and gate.is_contiguous() and gate.numel() == m * n and gate.device == x.device,
          "gate must be a contiguous Half [..., N] tensor of the output's shape")
    group = k // scales.shape[0]
    n_out = 0
    if oweights is not None:
        _need(oweights.dtype == torch.float16 and oweights.is_contiguous() and oweights.shape[0] == n,
              "oweights must be a contiguous Half [N, r] tensor")
        n_out = oweights.shape[1]
    out = torch.empty(*in_feats.shape[:-1], n, dtype=in_feats.dtype, device=in_feats.device)
    if m == 0:
        return out
    with torch.cuda.device(in_feats.device):
        _lib.check(_lib.lib().qeft_gemm_w4_silu_mul(x.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                                    oweights.data_ptr() if n_out else None,
                                                    bias.data_ptr() if bias is not None else None, gate.data_ptr(),
                                                    out.data_ptr(), m, n, k, group, n_out, _stream(x)))
    return out


# ---- entry points beyond the reference's module (used by QuantLinear's fused paths and the backward) ----

def gemv_4bit_fused(in_feats, kernel, scaling_factors, zeros, oweight_il, bias, reorder_ids, residual, m, n, k,
                    group_size, sz_packed=None):
    """One launch for everything QuantLinear.forward_* does around the GEMV (qlinear.py:244-330).
    sz_packed: optional derived buffer from pack_scales() (block-contiguous scales for the MFMA GEMV)."""
    _check_common(in_feats, kernel, scaling_factors, zeros)
    x = in_feats.contiguous()
    _need(x.numel() == m * k, f"in_feats has {x.numel()} elements, expected m*k = {m * k}")
    n_out = oweight_il.shape[1] // 2 if oweight_il is not None else 0
    if reorder_ids is not None:
        _need(reorder_ids.dtype == torch.int32 and reorder_ids.numel() == k and reorder_ids.is_contiguous(),
              "reorder_ids must be a contiguous int32 [K] tensor")
    if residual is not None:
        residual = residual.contiguous()
        _need(residual.numel() == m * n and residual.dtype == torch.float16, "residual must be Half [m, N]")
    out = torch.empty(*in_feats.shape[:-1], n, dtype=in_feats.dtype, device=in_feats.device)
    with torch.cuda.device(in_feats.device):
        _lib.check(_lib.lib().qeft_gemv_w4_fused(
            x.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(),
            oweight_il.data_ptr() if n_out else None, bias.data_ptr() if bias is not None else None,
            reorder_ids.data_ptr() if reorder_ids is not None else None,
            residual.data_ptr() if residual is not None else None,
            sz_packed.data_ptr() if sz_packed is not None else None, out.data_ptr(), m, n, k, group_size, n_out,
            _stream(x)))
    return out


def pack_scales(scales, zeros, n, k, group_size):
    """Derived buffer int32 [N/16, K/g, 16] = scale | scaled_zero << 16 (qeft_pack_scales).  None when the layer
    does not qualify (group size other than 128 / per-channel, N % 16 != 0, not on a GPU)."""
    if not scales.is_cuda or n % 16 != 0 or group_size not in (128, k):
        return None
    out = torch.empty(n // 16, k // group_size, 16, dtype=torch.int32, device=scales.device)
    with torch.cuda.device(scales.device):
        _lib.check(_lib.lib().qeft_pack_scales(scales.data_ptr(), zeros.data_ptr(), out.data_ptr(), n, k, group_size,
                                               _stream(scales)))
    return out


def gemm_4bit_dx(grad_out, kernel, scales, zeros, oweights):
    """dX[M,K] = dY[M,N] . Wdeq[N,K] with the outlier columns taken from oweights."""
    _check_common(grad_out, kernel, scales, zeros)
    dy = grad_out.contiguous()
    n = kernel.shape[0] * 4
    k = kernel.shape[1]
    m = dy.numel() // n
    _need(dy.shape[-1] == n, f"grad_out has N={dy.shape[-1]}, kernel has N={n}")
    group = k // scales.shape[0]
    n_out = oweights.shape[1] if oweights is not None else 0
    out = torch.empty(*grad_out.shape[:-1], k, dtype=grad_out.dtype, device=grad_out.device)
    with torch.cuda.device(dy.device):
        lib = _lib.lib()
        st = _stream(dy)
        need = lib.qeft_gemm_w4_dx_workspace_bytes(m, n, k)
        ws = None
        if need > 0:      # few output tiles, long contraction: split over n through the per-(device, stream) scratch
            key = (dy.device.index, st)
            ws = _GEMM_WS.get(key)
            if ws is None or ws.numel() * 4 < need:
                ws = _GEMM_WS[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=dy.device)
        _lib.check(lib.qeft_gemm_w4_dx_ws(dy.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                          oweights.data_ptr() if n_out else None, out.data_ptr(),
                                          ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0,
                                          m, n, k, group, n_out, st))
    return out


def grad_oweight(grad_out, in_feats, n_out):
    """d_oweight[N, r] (fp32) = dY^T . x[..., K-r:]  (qlinear.py:41-42)."""
    dy = grad_out.contiguous()
    x = in_feats.contiguous()
    _need(dy.dtype == torch.float16 and x.dtype == torch.float16, "expected Half tensors")
    n, k = dy.shape[-1], x.shape[-1]
    m = dy.numel() // n
    _need(x.numel() // k == m, "grad_out and in_feats disagree on the number of tokens")
    out = torch.empty(n, n_out, dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.lib().qeft_grad_oweight(dy.data_ptr(), x.data_ptr(), out.data_ptr(), m, n, k, n_out,
                                                _stream(dy)))
    return out


def dequantize_weight_4bit_qeft(kernel, scales, zeros, oweights=None):
    """Dense fp16 Wdeq[N, K] (role of the reference's uncompiled dequantize_weight_4bit_qeft, qeft_cuda.cpp:14)."""
    n, k = kernel.shape[0] * 4, kernel.shape[1]
    group = k // scales.shape[0]
    n_out = oweights.shape[1] if oweights is not None else 0
    out = torch.empty(n, k, dtype=torch.float16, device=kernel.device)
    with torch.cuda.device(kernel.device):
        _lib.check(_lib.lib().qeft_dequant_w4(kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                              oweights.data_ptr() if n_out else None, out.data_ptr(), n, k, group,
                                              n_out, _stream(kernel)))
    return out


def pack_oweight_device(oweights):
    """oweight [N, r] -> oweight_interleaved [N/2, 2r] on device (pack_oweight, qlinear.py:70-79)."""
    ow = oweights.contiguous()
    n, r = ow.shape
    out = torch.empty(n // 2, 2 * r, dtype=ow.dtype, device=ow.device)
    with torch.cuda.device(ow.device):
        _lib.check(_lib.lib().qeft_pack_oweight(ow.data_ptr(), out.data_ptr(), n, r, _stream(ow)))
    return out


# ---- 3-bit extension (include/qeft_hip.h, "3-bit EXTENSION"); the reference has no counterpart --------------------
def gemv_3bit(x, qweight3, scales, scaled_zeros, oweight_il, bias, residual, m, n, k, group_size, sz_packed=None):
    """y[m, n] = x . W3^T (+ bias, + residual) on the 3-bit stream; any m >= 1."""
    _need(x.is_cuda and x.dtype == torch.float16, "x must be a Half GPU tensor (no CPU fallback)")
    _need(qweight3.dtype == torch.int32 and qweight3.is_contiguous(), "qweight3 must be a contiguous Int tensor")
    x = x.contiguous()
    _need(x.numel() == m * k, f"x has {x.numel()} elements, expected m*k = {m * k}")
    n_out = 0 if oweight_il is None else oweight_il.shape[1] // 2
    _need(qweight3.shape == (n // 16, (k - n_out) // 128 * 192),
          f"qweight3 shape {tuple(qweight3.shape)} != ({n // 16}, {(k - n_out) // 128 * 192})")
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().qeft_gemv_w3(x.data_ptr(), qweight3.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                           oweight_il.data_ptr() if oweight_il is not None else None,
                                           bias.data_ptr() if bias is not None else None,
                                           residual.data_ptr() if residual is not None else None,
                                           sz_packed.data_ptr() if sz_packed is not None else None, out.data_ptr(), m, n,
                                           k, group_size, n_out, _stream(x)))
    return out


def gemm_3bit_qeft(in_feats, qweight3, scales, zeros, oweights, bias=None):
    """gemm_4bit_qeft for a 3-bit layer: the loader-wave GEMM tiers read the 3-bit stream directly (qeft_gemm_w3); shapes
    they do not take are expanded to the 4-bit layout into a buffer of THIS call (no shared scratch) and use the 4-bit entry."""
    _need(in_feats.is_cuda and in_feats.dtype == torch.float16, "in_feats must be a Half GPU tensor (no CPU fallback)")
    _need(qweight3.dtype == torch.int32 and qweight3.is_contiguous(), "qweight3 must be a contiguous Int tensor")
    x = in_feats.contiguous()
    k = x.shape[-1]
    n = qweight3.shape[0] * 16
    m = x.numel() // k
    n_out = oweights.shape[1] if oweights is not None else 0
    _need(qweight3.shape[1] == (k - n_out) // 128 * 192, f"qweight3 has {qweight3.shape[1]} columns, expected {(k - n_out) // 128 * 192}")
    group = k // scales.shape[0]
    lib = _lib.lib()
    if m == 0 or not lib.qeft_gemm_w3_supported(m, n, k, group, n_out):
        return gemm_4bit_qeft(in_feats, expand_3bit(qweight3, n, k, n_out), scales, zeros, oweights, bias)
    if oweights is not None:
        _need(oweights.dtype == torch.float16 and oweights.is_contiguous() and oweights.shape[0] == n,
              "oweights must be a contiguous Half [N, r] tensor")
    out = torch.empty(*in_feats.shape[:-1], n, dtype=in_feats.dtype, device=in_feats.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.qeft_gemm_w3(x.data_ptr(), qweight3.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                    oweights.data_ptr() if n_out else None, bias.data_ptr() if bias is not None else None,
                                    out.data_ptr(), m, n, k, group, n_out, _stream(x)))
    return out


def gemm_3bit_dx(grad_out, qweight3, scales, zeros, oweights, k):
    """gemm_4bit_dx for a 3-bit layer (k: in_features; the 3-bit buffer has no columns for the fp16 slice)."""
    dy = grad_out.contiguous()
    n = qweight3.shape[0] * 16
    m = dy.numel() // n
    n_out = oweights.shape[1] if oweights is not None else 0
    # (the C entry walks (k - n_out) / 128 steps of 192 ints per 16-row set: a wrong k, or oweights=None on a layer packed with
    #  r > 0, would read past the 3-bit buffer)
    _need(qweight3.dtype == torch.int32 and qweight3.is_contiguous(), "qweight3 must be a contiguous Int tensor")
    _need(qweight3.shape[1] == (k - n_out) // 128 * 192, f"qweight3 has {qweight3.shape[1]} columns, expected {(k - n_out) // 128 * 192}")
    _need(scales.shape[0] > 0 and k % scales.shape[0] == 0, f"scales has {scales.shape[0]} groups for k = {k}")
    group = k // scales.shape[0]
    _need(tuple(scales.shape) == (k // group, n) and tuple(zeros.shape) == (k // group, n),
          f"scales / zeros must be [{k // group}, {n}], got {tuple(scales.shape)} / {tuple(zeros.shape)}")
    _need(qweight3.data_ptr() % 16 == 0, "qweight3 must be 16-byte aligned")
    lib = _lib.lib()
    if m == 0 or not lib.qeft_gemm_w3_dx_supported(m, n, k, group, n_out):
        return gemm_4bit_dx(grad_out, expand_3bit(qweight3, n, k, n_out), scales, zeros, oweights)
    _need(dy.dtype == torch.float16 and dy.shape[-1] == n, "grad_out must be Half [..., N]")
    out = torch.empty(*grad_out.shape[:-1], k, dtype=grad_out.dtype, device=grad_out.device)
    with torch.cuda.device(dy.device):
        _lib.check(lib.qeft_gemm_w3_dx(dy.data_ptr(), qweight3.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                       oweights.data_ptr() if n_out else None, out.data_ptr(), m, n, k, group, n_out, _stream(dy)))
    return out


def expand_3bit(qweight3, n, k, n_out, out=None):
    """3-bit stream -> int16 [n/4, k] in the 4-bit checkpoint layout (for the GEMM / backward / dequant kernels)."""
    _need(qweight3.is_cuda and qweight3.dtype == torch.int32 and qweight3.is_contiguous(),
          "qweight3 must be a contiguous Int GPU tensor")
    if out is None:
        out = torch.empty(n // 4, k, dtype=torch.int16, device=qweight3.device)
    with torch.cuda.device(qweight3.device):
        _lib.check(_lib.lib().qeft_expand_w3(qweight3.data_ptr(), out.data_ptr(), n, k, n_out, _stream(qweight3)))
    return out


# ---- the two FasterTransformer-derived entries of the reference module (qeft_cuda.cpp:22-26), imported unconditionally
# ---- by qeft/monkeypatch/ftllama_modeling.py:18.  Not on the packed-weight path; adapters over the decode helpers.
def layernorm_forward_cuda(x, gamma, out, eps):
    """RMSNorm `out = x * rsqrt(mean(x^2) + eps) * gamma` (qeft/kernel/layernorm/layernorm.cu:94-110: input [b, n, c],
    m = b*n rows of c; T5 layer norm = no mean subtraction, no beta).  Writes into `out`, returns None like the reference."""
    _need(x.is_cuda and x.dtype == torch.float16 and gamma.dtype == torch.float16 and out.dtype == torch.float16,
          "expected Half GPU tensors")
    _need(x.dim() == 3, "input must be [b, n, c]")
    _need(x.is_contiguous() and out.is_contiguous() and gamma.is_contiguous(), "tensors must be contiguous")
    _need(out.shape == x.shape and gamma.numel() == x.shape[2], "shape mismatch")
    m, c = x.shape[0] * x.shape[1], x.shape[2]
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().qeft_rmsnorm(x.data_ptr(), None, gamma.data_ptr(), None, out.data_ptr(), m, c, float(eps),
                                           _stream(x)))


_ROPE_TABLES = {}   # (device, rotary_dim, base, rows) -> fp32 [2][rows][64]: cos table, sin table


def _rope_table(device, dim, base, rows):
    key = (device, dim, float(base), rows)
    tab = _ROPE_TABLES.get(key)
    if tab is None:
        half = 64
        if dim == 0:       # no rotary: identity rotation
            tab = torch.stack([torch.ones(rows, half), torch.zeros(rows, half)])
        else:
            inv = 1.0 / (float(base) ** (torch.arange(0, half, dtype=torch.float64) / half))
            ang = torch.arange(rows, dtype=torch.float64)[:, None] * inv[None, :]
            tab = torch.stack([ang.cos(), ang.sin()])
        tab = _ROPE_TABLES[key] = tab.float().contiguous().to(device)
    return tab


def single_query_attention(q, k, v, k_cache, v_cache, length_per_sample_, alibi_slopes_, timestep,
                           rotary_embedding_dim=0, rotary_base=10000.0, neox_rotary_style=True):
    """ft_attention.cpp:110-181 with the reference's positional signature and cache layouts:
        q [B, H, 128], k / v [B, Hkv, 128] fp16 (last-dim stride 1, head stride 128)
        k_cache [B, Hkv, 128/8, L, 8], v_cache [B, Hkv, L, 128] fp16 contiguous
    Rotates q / k at position `timestep`, appends k / v to the caches at that position and returns
    softmax(q K^T / sqrt(128) [+ alibi]) V as a new tensor shaped like q.  length_per_sample_ (int32 [B], optional): per-sample
    position instead of `timestep` (ft_attention.cpp:143-149).  alibi_slopes_ (fp32 [H], optional): slope * (key - query
    position) added to the scaled scores (decoder_masked_multihead_attention_template.hpp:1335-1345).
    Rotary: the neox style over the whole head (Llama: rotary_embedding_dim 128) or none (0) runs inside the kernel; a partial
    rotary_embedding_dim (even, < 128) and the interleaved GPT-J style (neox_rotary_style=False) are rotated here with a few
    torch ops (fp32 math, rounded to half as the reference's kernel does) in front of a launch without rotary.
    Head sizes other than 128 (any multiple of 8 up to 256: the reference instantiates 32 .. 256, ft_attention.cpp:110-181) take
    the generic kernel (qeft_single_query_attention_generic): rotary always applied here first, ALiBi inside.
    Not supported (RuntimeError): fp32 / bf16, head sizes that are not a multiple of 8 or exceed 256."""
    _need(q.is_cuda and q.dtype == torch.float16, "single_query_attention: only Half GPU tensors are supported")
    B, Hkv, L, D = v_cache.shape
    H = q.shape[1]
    _need(D % 8 == 0 and 8 <= D <= 256, "single_query_attention: head_dim must be a multiple of 8 in [8, 256]")
    rot = int(rotary_embedding_dim)
    _need(0 <= rot <= D and rot % 2 == 0, "single_query_attention: rotary_embedding_dim must be an even number in [0, head_dim]")
    if alibi_slopes_ is not None:
        _need(alibi_slopes_.is_cuda and alibi_slopes_.dtype == torch.float32 and alibi_slopes_.is_contiguous()
              and alibi_slopes_.numel() == H, "alibi_slopes_ must be a contiguous Float [n_heads] GPU tensor")
    _need(tuple(q.shape) == (B, H, D) and tuple(k.shape) == (B, Hkv, D) and tuple(v.shape) == (B, Hkv, D), "bad q/k/v shape")
    _need(tuple(k_cache.shape) == (B, Hkv, D // 8, L, 8), "k_cache must be [B, Hkv, Dh/8, L, 8]")
    _need(k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.dtype == torch.float16
          and v_cache.dtype == torch.float16, "caches must be contiguous Half tensors")
    for t in (q, k, v):
        _need(t.stride(2) == 1 and t.stride(1) == D, "q/k/v: last-dim stride 1 and head stride head_dim")
    _need(L % 16 == 0 and 16 <= L <= 32768, "cache length must be a multiple of 16 in [16, 32768]")
    _need(H % Hkv == 0, "n_heads must be a multiple of n_kv_heads")
    if length_per_sample_ is not None:
        _need(length_per_sample_.dtype == torch.int32 and length_per_sample_.is_cuda and length_per_sample_.numel() == B
              and length_per_sample_.is_contiguous(), "length_per_sample_ must be a contiguous int32 [B] GPU tensor")
        pos = length_per_sample_
    else:
        _need(0 <= int(timestep) < L, f"timestep {timestep} outside the cache (length {L})")
        pos = torch.full((B,), int(timestep), dtype=torch.int32, device=q.device)
    in_kernel = rot == 0 or (D == 128 and rot == D and neox_rotary_style)
    if not in_kernel:
        # rotate here: pairs (i, i + rot/2) (neox) or (2i, 2i + 1) (GPT-J) of the first `rot` dims, angle pos * base^(-2i/rot)
        half = rot // 2
        inv = 1.0 / (float(rotary_base) ** (torch.arange(0, half, dtype=torch.float64, device=q.device) * 2.0 / rot))
        ang = pos.to(torch.float64)[:, None] * inv[None, :]
        c, sn = ang.cos().float()[:, None, :], ang.sin().float()[:, None, :]

        def rotate(t):
            tf = t.float()
            a, b_ = (tf[..., :half], tf[..., half:rot]) if neox_rotary_style else (tf[..., 0:rot:2], tf[..., 1:rot:2])
            ra, rb = a * c - b_ * sn, b_ * c + a * sn
            o = tf.clone()
            if neox_rotary_style:
                o[..., :half], o[..., half:rot] = ra, rb
            else:
                o[..., 0:rot:2], o[..., 1:rot:2] = ra, rb
            return o.half().contiguous()
        q, k = rotate(q), rotate(k)
    out = torch.empty_like(q)
    if D != 128:
        with torch.cuda.device(q.device):
            lib, st = _lib.lib(), _stream(q)
            for b in range(B):
                _lib.check(lib.qeft_single_query_attention_generic(
                    q[b].data_ptr(), k[b].data_ptr(), v[b].data_ptr(), k_cache[b].data_ptr(), v_cache[b].data_ptr(),
                    pos.data_ptr() + 4 * b, out[b].data_ptr(), H, Hkv, L, D,
                    alibi_slopes_.data_ptr() if alibi_slopes_ is not None else None, st))
        return out
    tab = _rope_table(q.device, rot if in_kernel else 0, rotary_base, L)
    with torch.cuda.device(q.device):
        lib, st = _lib.lib(), _stream(q)
        for b in range(B):      # the kernel serves one sequence per launch (decode harness); the batch is a host loop
            args = (q[b].data_ptr(), k[b].data_ptr(), v[b].data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), L,
                    k_cache[b].data_ptr(), v_cache[b].data_ptr(), pos.data_ptr() + 4 * b, out[b].data_ptr(), H, Hkv, L)
            if alibi_slopes_ is None:
                _lib.check(lib.qeft_single_query_attention(*args, st))
            else:
                _lib.check(lib.qeft_single_query_attention_alibi(*args, alibi_slopes_.data_ptr(), st))
    return out


# ---- v3 decode linear (include/qeft_hip.h, "v3 decode linear"): batch-1 GEMV with producer-side epilogues ----------------
V3_PLAIN, V3_PAIR = 0, 1


def decode_linear_blocks(n_rows):
    """Number of partial sums (= blocks) a decode_linear launch over n_rows operand rows writes to ssq_out."""
    return _lib.lib().qeft_decode_linear_blocks(n_rows)


def decode_linear(x, op, mode=V3_PLAIN, residual=None, ssq_in=None, eps=0.0, gamma_out=None, out=None):
    """y = Wdeq . x for one packed operand and the fp16 vector x[K], one launch of the v3 GEMV (qeft_decode_linear).

    op: an object with qweight / sz_packed / oweight (plain fp16 [n, r]) / bias / outfeatures / infeatures / group_size /
        outlierfeatures (/ bits) -- a QuantLinear after set_kernel(), or a derived operand from qeft_amd.fuse (q|k|v
        concatenated, gate|up pair-interleaved).  bits == 3: the 3-bit extension layout (qeft_decode_linear_w3).
    mode V3_PAIR: op is fuse.pair_interleave(gate, up) -> returns silu(gate) * up (fp16 [n/2]).
    residual (fp32 [n]): returns y32 = Wx + residual (fp32; `out` may alias residual for an in-place update).
    ssq_in (fp32 partial sums): x is (v * gamma) and ssq_in the partial sums of v^2: y is scaled by rsqrt(sum / K + eps).
    gamma_out (fp16 [n], with residual): also returns (y_norm = fp16(y * gamma_out), ssq partials of y) for the next launch.
    Returns y, or (y, y_norm, ssq) with gamma_out."""
    n, k, g, r = op.outfeatures, op.infeatures, op.group_size, op.outlierfeatures
    _need(x.is_cuda and x.dtype == torch.float16 and x.numel() == k and x.is_contiguous(), "x must be a contiguous Half [K] GPU tensor")
    szp = getattr(op, "sz_packed", None)
    _need(szp is not None, "decode_linear needs the sz_packed buffer (QuantLinear.set_kernel on a GPU)")
    dev = x.device
    ow = None
    if r:
        ow = op.oweight if op.oweight.dtype == torch.float16 else op.oweight.to(torch.float16)
        _need(ow.is_contiguous() and tuple(ow.shape) == (n, r), "oweight must be a contiguous [n, r] tensor")
    if residual is not None:
        _need(mode == V3_PLAIN and residual.dtype == torch.float32 and residual.numel() == n, "residual must be Float [n]")
        y = out if out is not None else torch.empty(n, dtype=torch.float32, device=dev)
    else:
        y = out if out is not None else torch.empty(n // 2 if mode == V3_PAIR else n, dtype=torch.float16, device=dev)
    y_norm = ssq = None
    if gamma_out is not None:
        y_norm = torch.empty(n, dtype=torch.float16, device=dev)
        ssq = torch.zeros((decode_linear_blocks(n) + 3) // 4 * 4, dtype=torch.float32, device=dev)[:decode_linear_blocks(n)]
    w3 = getattr(op, "bits", 4) == 3           # 3-bit extension layout: qweight int32 [n/16, ((k - r)/128) * 192]
    _need(op.qweight.dtype == (torch.int32 if w3 else torch.int16), "qweight dtype does not match the operand's bit width")
    with torch.cuda.device(dev):
        entry = _lib.lib().qeft_decode_linear_w3 if w3 else _lib.lib().qeft_decode_linear
        _lib.check(entry(
            x.data_ptr(), op.qweight.data_ptr(), szp.data_ptr(), ow.data_ptr() if r else None,
            op.bias.data_ptr() if op.bias is not None else None, y.data_ptr(), n, k, g, r, mode,
            residual.data_ptr() if residual is not None else None,
            ssq_in.data_ptr() if ssq_in is not None else None, ssq_in.numel() if ssq_in is not None else 0, float(eps),
            gamma_out.data_ptr() if gamma_out is not None else None,
            y_norm.data_ptr() if y_norm is not None else None, ssq.data_ptr() if ssq is not None else None, _stream(x)))
    return (y, y_norm, ssq) if gamma_out is not None else y


def decode_linear_hnorm(h32, gamma, op, mode=V3_PLAIN, eps=1e-5, out=None):
    """y = op . rmsnorm(h32) * gamma with the whole norm inside the launch (qeft_decode_linear_hnorm): h32 fp32 [K], gamma
    fp16 [K]; y fp16 [N] (V3_PLAIN) or [N/2] (V3_PAIR, SiLU(gate) * up)."""
    n, k, r = op.outfeatures, op.infeatures, op.outlierfeatures
    _need(h32.dtype == torch.float32 and h32.numel() == k and gamma.dtype == torch.float16 and gamma.numel() == k,
          "h32 must be float32 [K] and gamma float16 [K]")
    y = out if out is not None else torch.empty(n // 2 if mode == V3_PAIR else n, dtype=torch.float16, device=h32.device)
    with torch.cuda.device(h32.device):
        _lib.check(_lib.lib().qeft_decode_linear_hnorm(h32.data_ptr(), gamma.data_ptr(), op.qweight.data_ptr(),
                                                       op.sz_packed.data_ptr(), op.oweight.data_ptr() if r else None,
                                                       op.bias.data_ptr() if op.bias is not None else None, y.data_ptr(), n, k,
                                                       op.group_size, r, mode, eps, _stream(h32)))
    return y


def residual_norm(h32, add=None, gamma=None, out=None):
    """h_out (fp32) = h32 (+ add fp16); with gamma also (fp16(h_out * gamma), partial sums of h_out^2) (qeft_residual_norm)."""
    n = h32.numel()
    h_out = out if out is not None else torch.empty_like(h32)
    hn = torch.empty(n, dtype=torch.float16, device=h32.device) if gamma is not None else None
    nb = _lib.lib().qeft_token_begin_norm_blocks(n)
    ssq = torch.zeros((nb + 3) // 4 * 4, dtype=torch.float32, device=h32.device)[:nb] if gamma is not None else None
    with torch.cuda.device(h32.device):
        _lib.check(_lib.lib().qeft_residual_norm(h32.data_ptr(), add.data_ptr() if add is not None else None,
                                                 gamma.data_ptr() if gamma is not None else None, h_out.data_ptr(),
                                                 hn.data_ptr() if hn is not None else None,
                                                 ssq.data_ptr() if ssq is not None else None, n, _stream(h32)))
    return (h_out, hn, ssq) if gamma is not None else h_out
