"""Host logic of the product (qeft_amd.qlinear): checkpoint layout and QuantLinear.pack are bit-exact
against the reference-generated golden vectors; module API mirrors the reference."""
import numpy as np
import pytest
import torch

from conftest import golden_cases
from qeft_amd.qlinear import QuantLinear, pack_intweight, pack_oweight, unpack_intweight, unpack_oweight
from qeft_amd.reorder import sparse_to_dense_ids

CASES = golden_cases()


@pytest.mark.parametrize("path", CASES)
def test_packers_bit_exact(path):
    d = np.load(path)
    n, k, r, g, sym, bias = [int(v) for v in d["case"]]
    got = pack_intweight(torch.from_numpy(d["qraw"].astype(np.int32)), 4, 64)
    assert got.dtype == torch.int16 and np.array_equal(got.numpy(), d["qraw_packed"])
    assert np.array_equal(unpack_intweight(torch.from_numpy(d["qraw_packed"])).numpy(), d["qraw"])
    if r:
        il = pack_oweight(torch.from_numpy(d["ow_rand"]), 4)
        assert np.array_equal(il.numpy().view(np.uint16), d["ow_rand_packed"].view(np.uint16))
        assert np.array_equal(unpack_oweight(il).numpy().view(np.uint16), d["ow_rand"].view(np.uint16))
        ids = sparse_to_dense_ids(torch.from_numpy(d["outlieridx"]), k)
        assert np.array_equal(ids.numpy(), d["reorder_ids"])


@pytest.mark.parametrize("path", CASES)
def test_quantlinear_pack_state_dict_matches_reference(path):
    d = np.load(path)
    n, k, r, g, sym, bias = [int(v) for v in d["case"]]
    ql = QuantLinear(4, k, n, bool(bias), torch.float16, r, g, True, "model.layers.0.self_attn.o_proj")
    lin = torch.nn.Linear(k, n, bias=bool(bias), dtype=torch.float16)
    lin.weight.data = torch.from_numpy(d["w_fake"])
    if bias:
        lin.bias.data = torch.from_numpy(d["sd_bias"])
    ql.pack(lin, torch.from_numpy(d["scale"]), torch.from_numpy(d["zero"]), torch.from_numpy(d["outlieridx"]),
            sym=bool(sym))
    sd = ql.state_dict()
    ref_keys = sorted(kk[3:] for kk in d.files if kk.startswith("sd_"))
    assert sorted(sd.keys()) == ref_keys
    for key, val in sd.items():
        ref = d["sd_" + key]
        assert val.numpy().dtype == ref.dtype and tuple(val.shape) == ref.shape, key
        assert np.array_equal(val.numpy().view(np.uint8), ref.view(np.uint8)), key


def test_constructor_contract():
    with pytest.raises(AssertionError, match="Only 4 bits"):     # the reference: bits in [4] (qlinear.py:127)
        QuantLinear(2, 128, 8, False, torch.float16, 0, 128, False, "x")
    with pytest.raises(AssertionError, match="3-bit layout"):    # 3 bits is the extension: own shape rules
        QuantLinear(3, 128, 8, False, torch.float16, 0, 128, False, "x")
    q3 = QuantLinear(3, 512, 64, False, torch.float16, 128, 128, True, "model.layers.0.mlp.up_proj")
    assert q3.qweight.shape == (4, 3 * 192) and q3.qweight.dtype == torch.int32
    with pytest.raises(AssertionError, match="Only fp16"):
        QuantLinear(4, 128, 8, False, torch.bfloat16, 0, 128, False, "x")
    ql = QuantLinear(4, 4096, 4096, False, torch.float16, 128, 128, True, "model.layers.0.self_attn.q_proj")
    assert ql.qweight.shape == (1024, 4096) and ql.qweight.dtype == torch.int16
    assert ql.scales.shape == (32, 4096) and ql.scaled_zeros.shape == (32, 4096)
    assert ql.oweight.shape == (4096, 128) and ql.oweight_interleaved.shape == (2048, 256)
    assert ql.outlieridx.shape == (128,) and ql.outlieridx.dtype == torch.int32
    assert ql.bias is None and ql.interleave == 4
    pc = QuantLinear(4, 256, 8, True, torch.float16, 0, -1, False, "x")   # group_size -1 -> per channel
    assert pc.group_size == 256 and pc.scales.shape == (1, 8) and pc.bias.shape == (8,)


def test_set_kernel_binds_forward_like_reference():
    ql = QuantLinear(4, 256, 16, False, torch.float16, 64, 128, True, "model.layers.0.self_attn.o_proj")
    ql.outlieridx = torch.arange(10, 74, dtype=torch.int32)
    ql.set_kernel()
    assert ql.forward == ql.forward_outlier_out_proj
    assert ql.reorder_ids.dtype == torch.int64 and ql.reorder_ids.shape == (256,)
    assert "reorder_ids" in ql.state_dict() and "reorder_ids32" not in ql.state_dict()
    q2 = QuantLinear(4, 256, 16, False, torch.float16, 64, 128, True, "model.layers.0.mlp.up_proj")
    q2.set_kernel()
    assert q2.forward == q2.forward_outlier
    q3 = QuantLinear(4, 256, 16, False, torch.float16, 0, 128, False, "lm")
    q3.set_kernel(training=True)
    assert q3.forward == q3.forward_normal and q3.training and hasattr(q3, "matmul")
    q2.set_for_wct()
    assert isinstance(q2.oweight, torch.nn.Parameter) and q2.oweight.dtype == torch.float32 and q2.oweight.requires_grad
    assert isinstance(q2.qweight, torch.nn.Parameter) and not q2.qweight.requires_grad


def _ns(bufs, n, k, r, g):
    import types
    import torch
    t = {a: torch.from_numpy(np.ascontiguousarray(v)) for a, v in bufs.items() if a != "fake_weight"}
    return types.SimpleNamespace(qweight=t["qweight"], scales=t["scales"], scaled_zeros=t["scaled_zeros"], oweight=t.get("oweight"),
                                 bias=t.get("bias"), outfeatures=n, infeatures=k, group_size=g, outlierfeatures=r, bits=4)


@pytest.mark.parametrize("n,k,r,g,bias", [(128, 256, 64, 64, True), (192, 384, 128, 128, False), (64, 128, 0, 128, True)])
def test_gemm_side_fused_operands_are_row_permutations(n, k, r, g, bias):
    """fuse.concat_gemm_operand / pair64_gemm_operand (prefill: q|k|v and gate|up as one GEMM each): dequantising the derived
    buffers with the oracle gives exactly the source layers' dense weights, rows concatenated / interleaved in blocks of 64."""
    from oracle import qeft_oracle as O
    from qeft_amd import fuse
    a, b = O.make_layer(n, k, r, g, seed=1, bias=bias), O.make_layer(n, k, r, g, seed=2, bias=bias)
    la, lb = _ns(a, n, k, r, g), _ns(b, n, k, r, g)
    dense = lambda x: O.dequant_dense(x["qweight"], x["scales"], x["scaled_zeros"], x.get("oweight") if r else None, g)   # noqa: E731
    wa, wb = dense(a), dense(b)

    def dense_op(op):
        return O.dequant_dense(op.qweight.numpy(), op.scales.numpy(), op.scaled_zeros.numpy(),
                               op.oweight.numpy() if op.oweight is not None else None, g)
    cat = fuse.concat_gemm_operand([la, lb])
    assert cat.outfeatures == 2 * n
    assert np.array_equal(dense_op(cat), np.concatenate([wa, wb], 0))
    p64 = fuse.pair64_gemm_operand(la, lb)
    want = np.stack([wa.reshape(n // 64, 64, k), wb.reshape(n // 64, 64, k)], 1).reshape(2 * n, k)
    assert p64.outfeatures == 2 * n
    assert np.array_equal(dense_op(p64), want)
    if bias:
        assert np.array_equal(cat.bias.numpy(), np.concatenate([a["bias"], b["bias"]]))
        assert np.array_equal(p64.bias.numpy(), np.stack([a["bias"].reshape(-1, 64), b["bias"].reshape(-1, 64)], 1).reshape(-1))
