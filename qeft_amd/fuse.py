"""Derived operands of the v3 decode GEMV (csrc/gemv_v3.h), built once at load time and never saved.

The kernel streams ONE packed operand per launch.  Two fusions of the decoder are therefore expressed in the data:

* q|k|v (any linears that read the same x): their buffers concatenated along the output rows -- `concat_linears`.
* gate|up with SiLU(gate) * up in the epilogue: rows pair-interleaved so that every 16-row MFMA set holds 8 gate rows and
  the same 8 rows of up -- `pair_interleave`.  The checkpoint interleaves 4 rows per qweight row ("row group"), so this is
  a permutation of whole qweight rows / 8-row halves of the scale shadow / whole oweight rows: no nibble is touched.

The checkpoint buffers of the modules are left alone (state_dict round-trips unchanged).
"""
from types import SimpleNamespace

import torch


def _operand(qweight, sz_packed, oweight, bias, n, k, g, r, bits=4):
    return SimpleNamespace(qweight=qweight.contiguous(), sz_packed=sz_packed.contiguous(),
                           oweight=oweight.contiguous() if oweight is not None else None,
                           bias=bias.contiguous() if bias is not None else None,
                           outfeatures=n, infeatures=k, group_size=g, outlierfeatures=r, bits=bits)


def _bits(l):
    return getattr(l, "bits", 4)


def _plain_oweight(l):
    ow = l.oweight.detach()
    return (ow if ow.dtype == torch.float16 else ow.to(torch.float16))[:, -l.outlierfeatures:]


def _szp(l):
    szp = l._szp(l.scales) if hasattr(l, "_szp") else l.sz_packed
    assert szp is not None, "the layer has no sz_packed shadow (group size 128 / per-channel on a GPU, N % 16 == 0)"
    return szp


def _check(layers):
    l0 = layers[0]
    for l in layers:
        assert (l.infeatures, l.group_size, l.outlierfeatures, _bits(l)) == (l0.infeatures, l0.group_size, l0.outlierfeatures, _bits(l0))
        assert l.outfeatures % 16 == 0
        assert (l.bias is None) == (l0.bias is None)
    return l0.infeatures, l0.group_size, l0.outlierfeatures


def single(layer):
    """One linear as a v3 operand (shares the module's buffers)."""
    k, g, r = _check([layer])
    return _operand(layer.qweight, _szp(layer), _plain_oweight(layer) if r else None, layer.bias, layer.outfeatures, k, g, r, _bits(layer))


def concat_linears(layers):
    """q|k|v: rows of the linears one after the other (qweight [N/4, K], sz_packed [N/16, K/g, 16], oweight [N, r])."""
    k, g, r = _check(layers)
    return _operand(torch.cat([l.qweight for l in layers], 0), torch.cat([_szp(l) for l in layers], 0),
                    torch.cat([_plain_oweight(l) for l in layers], 0) if r else None,
                    torch.cat([l.bias for l in layers], 0) if layers[0].bias is not None else None,
                    sum(l.outfeatures for l in layers), k, g, r, _bits(layers[0]))


def pair_interleave(gate, up):
    """gate|up: operand row 16 i + j = gate row 8 i + j (j < 8), up row 8 i + j - 8 (j >= 8); 2 N rows."""
    k, g, r = _check([gate, up])
    n = gate.outfeatures
    assert up.outfeatures == n
    if _bits(gate) == 3:
        # 3-bit layout int32 [n/16][steps][4 chunks][16 rows][3]: a lane record (row, chunk) is self-contained, so the
        # interleave moves whole 12-byte records -- operand set i = rows 8i..8i+7 of gate (rows 0..7) and of up (rows 8..15)
        steps = gate.qweight.shape[1] // 192
        halves3 = lambda q: q.view(n // 16, steps, 4, 2, 8, 3).permute(0, 3, 1, 2, 4, 5).reshape(n // 8, steps, 4, 8, 3)   # noqa: E731
        qw = torch.cat([halves3(gate.qweight), halves3(up.qweight)], 3).reshape(n // 8, steps * 192)
    else:
        qw = torch.cat([gate.qweight.view(n // 8, 2, -1), up.qweight.view(n // 8, 2, -1)], 1).reshape(n // 2, -1)
    sg, su = _szp(gate), _szp(up)                       # [n/16, groups, 16]
    ng = sg.shape[1]
    halves = lambda s: s.view(n // 16, ng, 2, 8).permute(0, 2, 1, 3).reshape(n // 8, ng, 8)   # noqa: E731
    szp = torch.cat([halves(sg), halves(su)], -1)       # [n/8, groups, 16]
    ow = torch.cat([_plain_oweight(gate).view(n // 8, 8, r), _plain_oweight(up).view(n // 8, 8, r)], 1).reshape(2 * n, r) if r else None
    bias = None
    if gate.bias is not None:
        bias = torch.cat([gate.bias.view(n // 8, 8), up.bias.view(n // 8, 8)], 1).reshape(2 * n)
    return _operand(qw, szp, ow, bias, 2 * n, k, g, r, _bits(gate))


def column_shard(layer, owned_cols):
    """The K-shard of `layer` for a tensor-parallel rank that owns the input columns `owned_cols` (kernel column order, i.e.
    after the layer's reorder; any order, typically a contiguous run of INT4 columns plus some of the fp16 outlier columns).

    Returns (operand, local_pos): the operand keeps every output row and, of the input columns, the whole 128-column groups
    that hold an owned INT4 column plus the complete fp16 outlier slice; local_pos[i] is where owned column i sits in the
    operand's x vector.  x positions of columns the rank does NOT own (the rest of a boundary group, foreign outlier
    columns) must hold zeros: every column is then counted exactly once in the sum of the ranks' partial outputs, which is
    what the all-reduce after o_proj / down_proj forms (Megatron pairing: the row-sharded producer feeds the column-sharded
    consumer with no collective in between).  A boundary group's weights are streamed by both neighbours (<= 2 groups of
    K / P per rank) and the outlier slice by every rank (N x 128 fp16)."""
    from . import qeft_cuda
    k, g, r = _check([layer])
    assert g == 128 and r in (0, 128), "column sharding cuts at the 128-column groups of the v3 GEMV"
    n, kq = layer.outfeatures, k - r
    owned = torch.as_tensor(owned_cols, dtype=torch.long, device=layer.qweight.device)
    q_cols = owned[owned < kq]
    g0, g1 = (int(q_cols.min()) // 128, int(q_cols.max()) // 128 + 1) if q_cols.numel() else (0, 0)
    if g1 == g0 and r == 0:
        g1 = g0 + 1                    # an operand needs at least one step
    kp_q = 128 * (g1 - g0)
    qw = torch.cat([layer.qweight[:, 128 * g0:128 * g1], layer.qweight[:, kq:]], 1)
    sc = torch.cat([layer.scales[g0:g1], layer.scales[kq // g:]], 0).contiguous()
    sz = torch.cat([layer.scaled_zeros[g0:g1], layer.scaled_zeros[kq // g:]], 0).contiguous()
    szp = qeft_cuda.pack_scales(sc, sz, n, kp_q + r, g) if sc.is_cuda else None
    op = SimpleNamespace(qweight=qw.contiguous(), sz_packed=szp, scales=sc, scaled_zeros=sz,
                         oweight=_plain_oweight(layer).contiguous() if r else None, bias=None,
                         outfeatures=n, infeatures=kp_q + r, group_size=g, outlierfeatures=r, first_group=g0)
    local = torch.where(owned < kq, owned - 128 * g0, kp_q + owned - kq).to(torch.int32)
    return op, local


# ---------------------------------------------------------------------------------------------------------------------------
# GEMM-side operands of the batched prompt pass (llama.prefill): the checkpoint layout itself (qweight [N/4, K] int16,
# scales / scaled_zeros [K/g, N], oweight [N, r]), several linears in one launch.  Derived at load time, never saved.

def _gemm_operand(qweight, scales, zeros, oweight, bias, n, k, g, r):
    return SimpleNamespace(qweight=qweight.contiguous(), scales=scales.contiguous(), scaled_zeros=zeros.contiguous(),
                           oweight=oweight.contiguous() if oweight is not None else None,
                           bias=bias.contiguous() if bias is not None else None,
                           outfeatures=n, infeatures=k, group_size=g, outlierfeatures=r)


def _check_gemm(layers):
    k, g, r = _check(layers)
    assert all(_bits(l) == 4 for l in layers), "GEMM-side fusions read the 4-bit checkpoint layout"
    return k, g, r


def concat_gemm_operand(layers):
    """q|k|v for the GEMM: the linears' rows one after the other (they read the same x): one launch, y [M, sum N]."""
    k, g, r = _check_gemm(layers)
    return _gemm_operand(torch.cat([l.qweight for l in layers], 0), torch.cat([l.scales for l in layers], 1),
                         torch.cat([l.scaled_zeros for l in layers], 1),
                         torch.cat([_plain_oweight(l) for l in layers], 0) if r else None,
                         torch.cat([l.bias for l in layers], 0) if layers[0].bias is not None else None,
                         sum(l.outfeatures for l in layers), k, g, r)


def pair64_gemm_operand(gate, up):
    """gate|up for qeft_gemm_w4_gateup: rows interleaved in blocks of 64 -- operand rows 128 b .. 128 b + 63 = gate rows
    64 b .., rows 128 b + 64 .. = up rows 64 b .. -- so that a 128-column tile of the GEMM holds the same 64 columns of both
    linears and SiLU(gate) * up is its epilogue.  64 rows = 16 qweight rows: whole checkpoint rows move, no nibble is touched."""
    k, g, r = _check_gemm([gate, up])
    n = gate.outfeatures
    assert up.outfeatures == n and n % 64 == 0
    il_rows = lambda a, b, rows: torch.stack([a.reshape(n // 64, rows, -1), b.reshape(n // 64, rows, -1)], 1).reshape(2 * a.shape[0], -1)   # noqa: E731
    il_cols = lambda a, b: torch.stack([a.reshape(a.shape[0], n // 64, 64), b.reshape(b.shape[0], n // 64, 64)], 2).reshape(a.shape[0], 2 * n)   # noqa: E731
    return _gemm_operand(il_rows(gate.qweight, up.qweight, 16), il_cols(gate.scales, up.scales),
                         il_cols(gate.scaled_zeros, up.scaled_zeros),
                         il_rows(_plain_oweight(gate), _plain_oweight(up), 64) if r else None,
                         il_rows(gate.bias[:, None], up.bias[:, None], 64).reshape(-1) if gate.bias is not None else None,
                         2 * n, k, g, r)
