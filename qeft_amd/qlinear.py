"""QuantLinear: the packed W4 (+ fp16 outlier columns) linear, MI355X implementation.

Host-side mirror of the reference operator module (qeft/qlinear.py): same constructor, buffer names /
state_dict keys, pack(), set_kernel(), set_for_wct() and forward dispatch, so a packed checkpoint produced
by the reference loads unchanged and the reference's drivers can use this class as-is.  The arithmetic runs
in hand-written gfx950 kernels behind the C ABI (include/qeft_hip.h); there is no CPU fallback.

Differences from the reference, all deliberate:
  * forward paths are fused by default (`fused=True`): decode = one GEMV launch that also does the o_proj
    gather and the bias; prefill = one MFMA GEMM launch that also contracts the fp16 outlier slice and adds
    the bias (reference: gemm + F.linear + add, qlinear.py:265-268).  `fused=False` reproduces the reference's
    call sequence kernel by kernel.
  * the training backward computes dX = dY . Wdeq (the released code re-applies the forward operator,
    qlinear.py:38-39).
  * `refresh_interleaved()` re-derives oweight_interleaved after oweight changed (fixes the stale-copy quirk
    of modelutils.py:192).
"""
import torch
import torch.nn as nn

from . import qeft_cuda
from .reorder import sparse_to_dense_ids


# ----------------------------------------------------------------------------------------------------
# Checkpoint layout (torch, any device).  Index arithmetic instead of the reference's chain of
# reshapes/transposes; bit-exact against it on the golden vectors (tests/test_layout.py).
# ----------------------------------------------------------------------------------------------------
def pack_intweight(unpacked_qweight, interleave=4, kstride=64):
    """int [N, K] in 0..15 -> int16 [N/4, K]  (reference pack_intweight, qlinear.py:81-121).

    Nibble b of int16 [n//4, (k//64)*64 + (n%4)*16 + ((k%64)//32)*8 + j] holds weight (n, k) with
    k%32 = 8*b + j.  No clamping, like the reference.
    """
    assert interleave == 4 and kstride == 64, "only the checkpoint's interleave=4 / kstride=64 layout exists"
    q = unpacked_qweight
    n, k = q.shape
    assert n % 4 == 0 and k % 64 == 0
    v = q.to(torch.int32).reshape(n // 4, 4, k // 64, 2, 4, 8)          # n4, r, t, chunk, b, j
    word = v[..., 0, :] | (v[..., 1, :] << 4) | (v[..., 2, :] << 8) | (v[..., 3, :] << 12)
    word = word.permute(0, 2, 1, 3, 4).reshape(n // 4, k)                 # n4, t, r, chunk, j
    word = torch.where(word >= 32768, word - 65536, word)                 # two's complement into int16
    return word.to(torch.int16).contiguous()


def unpack_intweight(qweight):
    """Inverse of pack_intweight: int16 [N/4, K] -> uint8 [N, K]."""
    n4, k = qweight.shape
    word = qweight.to(torch.int32) & 0xFFFF
    word = word.reshape(n4, k // 64, 4, 2, 8).permute(0, 2, 1, 3, 4)      # n4, r, t, chunk, j
    nib = torch.stack([(word >> (4 * b)) & 0xF for b in range(4)], dim=-2)  # n4, r, t, chunk, b, j
    return nib.reshape(n4 * 4, k).to(torch.uint8)


def pack_oweight(oweight, interleave=4):
    """fp16 [N, r] -> [N/2, 2r]: rows n and n+4 interleaved per 32-column chunk (qlinear.py:70-79)."""
    assert interleave == 4
    n, r = oweight.shape
    assert n % 8 == 0 and r % 32 == 0
    v = oweight.reshape(n // 8, 2, 4, r // 32, 32)                        # blk, h, rr, c, jj
    return v.permute(0, 2, 3, 4, 1).reshape(n // 2, 2 * r).contiguous()


def unpack_oweight(oweight_interleaved):
    n2, r2 = oweight_interleaved.shape
    n, r = n2 * 2, r2 // 2
    v = oweight_interleaved.reshape(n // 8, 4, r // 32, 32, 2)           # blk, rr, c, jj, h
    return v.permute(0, 4, 1, 2, 3).reshape(n, r).contiguous()


def pack_w3(q):
    """int [N, Kq] in 0..7 -> int32 [N/16, (Kq/128)*192]: the 3-bit EXTENSION layout (the reference has none; spec and
    closed form in oracle/qeft_oracle.py: pack_w3 / w3_position, C ABI notes in include/qeft_hip.h)."""
    n, kq = q.shape
    assert n % 16 == 0 and kq % 128 == 0
    v = q.to(torch.int64).reshape(n // 16, 16, kq // 128, 4, 16, 2)      # rs, row, step, chunk, pair e, half h
    words = torch.zeros(n // 16, 16, kq // 128, 4, 3, dtype=torch.int64, device=q.device)
    for e in range(15):
        for h in range(2):
            words[..., e // 5] |= (v[..., e, h] & 7) << (16 * h + 3 * (e % 5))
    for h in range(2):
        for b in range(3):
            words[..., b] |= ((v[..., 15, h] >> b) & 1) << (15 + 16 * h)
    words = words.permute(0, 2, 3, 1, 4).reshape(n // 16, kq // 128 * 192)  # rs, step, chunk, row, i
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words)
    return words.to(torch.int32).contiguous()


def unpack_w3(qweight3):
    """Inverse of pack_w3: int32 [N/16, S*192] -> uint8 [N, S*128]."""
    nrs, cols = qweight3.shape
    steps = cols // 192
    w = (qweight3.to(torch.int64) & 0xFFFFFFFF).reshape(nrs, steps, 4, 16, 3).permute(0, 3, 1, 2, 4)
    out = torch.zeros(nrs, 16, steps, 4, 16, 2, dtype=torch.int64, device=qweight3.device)
    for e in range(15):
        for h in range(2):
            out[..., e, h] = (w[..., e // 5] >> (16 * h + 3 * (e % 5))) & 7
    for h in range(2):
        for b in range(3):
            out[..., 15, h] |= ((w[..., b] >> (15 + 16 * h)) & 1) << b
    return out.reshape(nrs * 16, steps * 128).to(torch.uint8)


# ----------------------------------------------------------------------------------------------------
# autograd wrappers (reference QuantMatMulQEFT / QuantMatMul, qlinear.py:13-68)
# ----------------------------------------------------------------------------------------------------
class QuantMatMulQEFT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oweight, qweight, scales, scaled_zeros, n_out, bias, name):
        dtype = scales.dtype
        x16 = x.to(dtype)
        ow16 = oweight.to(dtype).contiguous()
        # (an int32 qweight is the 3-bit extension layout: the same kernels read it directly where their tiers apply)
        gemm = qeft_cuda.gemm_3bit_qeft if qweight.dtype == torch.int32 else qeft_cuda.gemm_4bit_qeft
        y = gemm(x16, qweight, scales, scaled_zeros, ow16, bias)
        ctx.save_for_backward(x16, ow16, qweight, scales, scaled_zeros)
        ctx.n_out = n_out
        ctx.in_dtype = x.dtype
        ctx.ow_dtype = oweight.dtype
        return y

    @staticmethod
    def backward(ctx, grad_output):
        x16, ow16, qweight, scales, scaled_zeros = ctx.saved_tensors
        dy = grad_output.to(scales.dtype).contiguous()
        grad_input = grad_oweight = None
        if ctx.needs_input_grad[0]:
            if qweight.dtype == torch.int32:
                grad_input = qeft_cuda.gemm_3bit_dx(dy, qweight, scales, scaled_zeros, ow16, x16.shape[-1]).to(ctx.in_dtype)
            else:
                grad_input = qeft_cuda.gemm_4bit_dx(dy, qweight, scales, scaled_zeros, ow16).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            grad_oweight = qeft_cuda.grad_oweight(dy, x16, ctx.n_out).to(ctx.ow_dtype)
        return grad_input, grad_oweight, None, None, None, None, None, None


class QuantMatMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, qweight, scales, scaled_zeros, n_out, bias, name):
        dtype = scales.dtype
        gemm = qeft_cuda.gemm_3bit_qeft if qweight.dtype == torch.int32 else qeft_cuda.gemm_4bit_qeft
        y = gemm(x.to(dtype), qweight, scales, scaled_zeros, None, bias)
        ctx.save_for_backward(qweight, scales, scaled_zeros)
        ctx.in_dtype = x.dtype
        ctx.k = x.shape[-1]
        return y

    @staticmethod
    def backward(ctx, grad_output):
        qweight, scales, scaled_zeros = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            dy = grad_output.to(scales.dtype).contiguous()
            if qweight.dtype == torch.int32:
                grad_input = qeft_cuda.gemm_3bit_dx(dy, qweight, scales, scaled_zeros, None, ctx.k).to(ctx.in_dtype)
            else:
                grad_input = qeft_cuda.gemm_4bit_dx(dy, qweight, scales, scaled_zeros, None).to(ctx.in_dtype)
        return grad_input, None, None, None, None, None, None


class QuantLinear(nn.Module):
    """Signature and buffers as reference QuantLinear (qlinear.py:123-178)."""

    def __init__(self, bits, infeatures, outfeatures, bias, dtype, outlierfeatures, group_size, reorder, name):
        super().__init__()
        # the reference supports 4 bits only (qlinear.py:127); 3 is this build's extension (own layout, pack_w3)
        assert bits in [3, 4], "Only 4 bits (reference layout) and 3 bits (extension layout) are supported."
        assert dtype == torch.float16, "Only fp16 is supported."
        self.bits = bits
        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.outlierfeatures = outlierfeatures
        self.group_size = group_size if group_size != -1 else infeatures
        self.interleave = 4
        assert infeatures % self.group_size == 0
        if bits == 3:
            assert outfeatures % 16 == 0 and infeatures % 128 == 0 and outlierfeatures % 128 == 0 \
                and outlierfeatures < infeatures and self.group_size in (128, infeatures), \
                "3-bit layout: N % 16 == 0, K % 128 == 0, r % 128 == 0, group size 128 or K"
            self.register_buffer("qweight", torch.empty(
                (outfeatures // 16, (infeatures - outlierfeatures) // 128 * 192), dtype=torch.int32))
        else:
            assert outfeatures % (32 // self.bits) == 0
            int16_pack_num = 16 // self.bits
            self.register_buffer("qweight", torch.empty(
                (outfeatures // self.interleave, infeatures // int16_pack_num * self.interleave), dtype=torch.int16))
        numgroup = infeatures // self.group_size
        self.register_buffer("scales", torch.empty((numgroup, outfeatures), dtype=dtype))
        self.register_buffer("scaled_zeros", torch.empty((numgroup, outfeatures), dtype=dtype))
        if bias:
            self.register_buffer("bias", torch.empty((outfeatures), dtype=torch.float16))
        else:
            self.bias = None
        if outlierfeatures > 0:
            self.register_buffer("oweight", torch.empty((outfeatures, outlierfeatures), dtype=dtype))
            self.register_buffer("oweight_interleaved",
                                 torch.empty((outfeatures // 2, outlierfeatures * 2), dtype=dtype))
            self.register_buffer("outlieridx", torch.zeros((outlierfeatures), dtype=torch.int))

        self.faster = True
        self.dtype = dtype
        self.name = name
        self.reorder = reorder
        self.training = False
        self.fused = True

    # ------------------------------------------------------------------ pack (qlinear.py:180-215)
    def pack(self, linear, scales, zeros, outlieridx, sym=False):
        dtype = self.dtype
        self.sym = sym
        if sym:
            zeros = zeros + 2 ** (self.bits - 1)
        if linear.bias is not None:
            self.bias = linear.bias.detach().to(dtype)

        scale_zeros = zeros * scales
        rep = 1 if self.group_size == self.infeatures and scales.shape[1] == self.infeatures else self.group_size
        s_full = torch.repeat_interleave(scales, rep, dim=1)
        sz_full = torch.repeat_interleave(scale_zeros, rep, dim=1)
        intweight = torch.round((linear.weight.data + sz_full) / s_full).to(torch.int32)
        if self.outlierfeatures > 0:
            cols = torch.arange(self.infeatures - self.outlierfeatures, self.infeatures)
            intweight[:, cols] = zeros[:, cols // self.group_size].to(torch.int32)   # dead nibbles hold z
        if self.bits == 3:
            self.qweight = pack_w3(intweight[:, :self.infeatures - self.outlierfeatures])
        else:
            self.qweight = pack_intweight(intweight, interleave=4, kstride=64)
        self.scales = scales.t().contiguous().to(dtype)
        self.scaled_zeros = -scale_zeros.t().contiguous().to(dtype)
        if self.outlierfeatures > 0:
            oweight = linear.weight.data[:, -self.outlierfeatures:].clone()
            self.oweight = oweight
            self.oweight_interleaved = pack_oweight(oweight, interleave=4)
            self.outlieridx = outlieridx

    # ------------------------------------------------------------------ set_kernel (qlinear.py:217-237)
    def set_kernel(self, training=False):
        self.training = training
        # derived, never saved: block-contiguous (scale | scaled_zero) words for the decode GEMV
        self.sz_packed = qeft_cuda.pack_scales(self.scales, self.scaled_zeros, self.outfeatures, self.infeatures,
                                               self.group_size)
        if self.outlierfeatures > 0:
            # The reference left-pads oweight to a multiple of 64 columns here (qlinear.py:221-222) for its cuBLAS
            # F.linear and thereby breaks its own GEMM path for r % 64 != 0 (shape error).  No kernel of this build
            # consumes a padded slice -- the fused GEMM / dX / d(oweight) kernels take r = oweight.shape[1] as the
            # number of fp16 columns -- so oweight keeps its [N, r] shape (a padded one would silently turn live INT4
            # columns into zero-weight outlier columns in the training path).
            if self.oweight.shape[1] != self.outlierfeatures:     # a checkpoint saved after the reference's set_kernel
                self.oweight = self.oweight[:, -self.outlierfeatures:].contiguous()
            self.gemv = qeft_cuda.gemv_4bit_qeft
            self.gemm = qeft_cuda.gemm_4bit
            self.forward = self.forward_outlier
            if "o_proj" in self.name or "out_proj" in self.name:
                ids = sparse_to_dense_ids(self.outlieridx, self.infeatures)
                if "reorder_ids" in self._buffers:       # set_kernel() called again (e.g. switching to training)
                    self.reorder_ids, self.reorder_ids32 = ids, ids.to(torch.int32)
                else:
                    self.register_buffer("reorder_ids", ids)
                    self.register_buffer("reorder_ids32", ids.to(torch.int32), persistent=False)
                self.forward = self.forward_outlier_out_proj
            if training:
                self.matmul = QuantMatMulQEFT.apply
        else:
            self.gemv = qeft_cuda.gemv_4bit
            self.gemm = qeft_cuda.gemm_4bit
            self.forward = self.forward_normal
            if training:
                self.matmul = QuantMatMul.apply

    def set_for_wct(self):
        self.qweight = torch.nn.Parameter(self.qweight, requires_grad=False)
        if self.outlierfeatures > 0:
            self.oweight = torch.nn.Parameter(self.oweight.to(dtype=torch.float), requires_grad=True)

    def refresh_interleaved(self):
        """Re-derive oweight_interleaved from oweight (after fine-tuning / replace_oweight)."""
        if self.outlierfeatures > 0:
            ow = self.oweight.detach().to(self.dtype)
            if ow.is_cuda:
                self.oweight_interleaved = qeft_cuda.pack_oweight_device(ow)
            else:
                self.oweight_interleaved = pack_oweight(ow)

    def _szp(self, x):
        """The derived scale buffer, (re)built lazily if the module moved to another device after set_kernel()."""
        szp = getattr(self, "sz_packed", None)
        if szp is None or szp.device != x.device or self.scales.device != x.device:
            szp = self.sz_packed = qeft_cuda.pack_scales(self.scales, self.scaled_zeros, self.outfeatures,
                                                         self.infeatures, self.group_size)
        return szp

    # ------------------------------------------------------------------ 3-bit extension
    def _qweight4(self):
        """The layer's weights expanded to the 4-bit checkpoint layout, a fresh buffer per call (dense dequantisation for the
        tests / the dense reference, the fused-activation GEMM).  The GEMM forward and backward read the 3-bit stream directly
        where their loader-wave tiers apply (qeft_cuda.gemm_3bit_qeft / gemm_3bit_dx expand per call otherwise), so no shared
        scratch and no per-layer expanded copy exist any more."""
        return qeft_cuda.expand_3bit(self.qweight, self.outfeatures, self.infeatures, self.outlierfeatures)

    def _forward_w3(self, x, gather):
        r = self.outlierfeatures
        inputs = torch.index_select(x, -1, self.reorder_ids) if gather else x
        if self.training:
            if r > 0:
                return self.matmul(inputs, self.oweight, self.qweight, self.scales, self.scaled_zeros, r, self.bias,
                                   self.name)
            return self.matmul(inputs, self.qweight, self.scales, self.scaled_zeros, r, self.bias, self.name)
        seq_len = x.numel() // x.shape[-1]
        if 0 < seq_len <= 16:   # decode / few rows: the 3-bit stream is read directly
            return qeft_cuda.gemv_3bit(inputs, self.qweight, self.scales, self.scaled_zeros,
                                       self.oweight_interleaved if r > 0 else None, self.bias, None, seq_len,
                                       self.outfeatures, self.infeatures, self.group_size, self._szp(x))
        return qeft_cuda.gemm_3bit_qeft(inputs, self.qweight, self.scales, self.scaled_zeros,
                                        self._outlier_weight_f16() if r > 0 else None, self.bias)

    # ------------------------------------------------------------------ forwards (qlinear.py:244-330)
    def _outlier_weight_f16(self):
        ow = self.oweight
        if ow.dtype != self.dtype:
            ow = ow.to(self.dtype)
        return ow[:, -self.outlierfeatures:].contiguous() if ow.shape[1] != self.outlierfeatures else ow

    def forward_outlier(self, x):
        if self.bits == 3:
            return self._forward_w3(x, False)
        if self.training:
            return self.matmul(x, self.oweight, self.qweight, self.scales, self.scaled_zeros,
                               self.outlierfeatures, self.bias, self.name)
        seq_len = x.numel() // x.shape[-1]
        if self.fused:
            if seq_len < 8:
                return qeft_cuda.gemv_4bit_fused(x, self.qweight, self.scales, self.scaled_zeros,
                                                 self.oweight_interleaved, self.bias, None, None, seq_len,
                                                 self.outfeatures, self.infeatures, self.group_size, self._szp(x))
            return qeft_cuda.gemm_4bit_qeft(x, self.qweight, self.scales, self.scaled_zeros,
                                            self._outlier_weight_f16(), self.bias)
        if seq_len < 8:
            y = self.gemv(x, self.qweight, self.scales, self.scaled_zeros, self.oweight_interleaved, seq_len,
                          self.outfeatures, self.infeatures, self.group_size)
        else:
            y = self.gemm(x, self.qweight, self.scales, self.scaled_zeros)
            y += torch.nn.functional.linear(x[..., -self.outlierfeatures:], self._outlier_weight_f16())
        return y + self.bias if self.bias is not None else y

    def forward_outlier_out_proj(self, x):
        if self.bits == 3:
            return self._forward_w3(x, True)
        if self.training:
            inputs = torch.index_select(x, -1, self.reorder_ids)
            return self.matmul(inputs, self.oweight, self.qweight, self.scales, self.scaled_zeros,
                               self.outlierfeatures, self.bias, self.name)
        seq_len = x.numel() // x.shape[-1]
        if self.fused and seq_len < 8:
            # gather folded into the GEMV's x staging
            return qeft_cuda.gemv_4bit_fused(x, self.qweight, self.scales, self.scaled_zeros,
                                             self.oweight_interleaved, self.bias, self.reorder_ids32, None, seq_len,
                                             self.outfeatures, self.infeatures, self.group_size, self._szp(x))
        inputs = torch.index_select(x, -1, self.reorder_ids)
        if self.fused:
            return qeft_cuda.gemm_4bit_qeft(inputs, self.qweight, self.scales, self.scaled_zeros,
                                            self._outlier_weight_f16(), self.bias)
        if seq_len < 8:
            y = self.gemv(inputs, self.qweight, self.scales, self.scaled_zeros, self.oweight_interleaved, seq_len,
                          self.outfeatures, self.infeatures, self.group_size)
        else:
            y = self.gemm(inputs, self.qweight, self.scales, self.scaled_zeros)
            y += torch.nn.functional.linear(inputs[..., -self.outlierfeatures:], self._outlier_weight_f16())
        return y + self.bias if self.bias is not None else y

    def forward_silu_mul(self, x, gate):
        """silu(gate) * self(x), inference: the MLP's act_fn(gate_proj(x)) * up_proj(x) with the activation formed in this
        layer's GEMM epilogue where the launch has one (>= 8 rows on the fused path; same rounding as the unfused pair).
        Extension: the reference multiplies in torch (modeling_llama's LlamaMLP around QuantLinear.forward)."""
        seq_len = x.numel() // x.shape[-1]
        if self.training or not self.fused or seq_len < 8 or self.forward == self.forward_outlier_out_proj:
            y = self.forward(x)
            return torch.nn.functional.silu(gate.float()).mul_(y.float()).to(y.dtype)
        r = self.outlierfeatures
        qw = self._qweight4() if self.bits == 3 else self.qweight
        return qeft_cuda.gemm_4bit_qeft_silu_mul(x, qw, self.scales, self.scaled_zeros,
                                                 self._outlier_weight_f16() if r > 0 else None, gate.contiguous(), self.bias)

    def forward_normal(self, x):
        if self.bits == 3:
            return self._forward_w3(x, False)
        if self.training:
            return self.matmul(x, self.qweight, self.scales, self.scaled_zeros, self.outlierfeatures, self.bias,
                               self.name)
        seq_len = x.numel() // x.shape[-1]
        if self.fused:
            if seq_len < 8:
                return qeft_cuda.gemv_4bit_fused(x, self.qweight, self.scales, self.scaled_zeros, None, self.bias,
                                                 None, None, seq_len, self.outfeatures, self.infeatures,
                                                 self.group_size, self._szp(x))
            return qeft_cuda.gemm_4bit_qeft(x, self.qweight, self.scales, self.scaled_zeros, None, self.bias)
        if seq_len < 8:
            y = self.gemv(x, self.qweight, self.scales, self.scaled_zeros, seq_len, self.outfeatures,
                          self.infeatures, self.group_size)
        else:
            y = self.gemm(x, self.qweight, self.scales, self.scaled_zeros)
        return y + self.bias if self.bias is not None else y
