"""GPU parity of the 3-bit extension (BASELINE config 5) against the oracle, all through the C ABI: decode GEMV on the
3-bit stream, the fused decode variants, the 3-bit -> 4-bit expansion that feeds the GEMM / backward kernels, the
QuantLinear module at bits = 3 (forward at any row count, fine-tune backward) and the decode engine on a w3 model."""
import ctypes
import dataclasses

import numpy as np
import pytest
import torch

from oracle import qeft_oracle as O
from util import REL_TOL, elem_err_ok, layer_to_torch, rel_err

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _st():
    return torch.cuda.current_stream(DEV).cuda_stream


def _arr(ts):
    a = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        a[i] = t.data_ptr() if t is not None else None
    return a


def _ref(bufs, x, g):
    return O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight"), None, g)


@pytest.mark.parametrize("n,k,r,g", [(4096, 4096, 128, 128), (11008, 4096, 128, 128), (4096, 11008, 128, 128),
                                     (64, 512, 128, 128), (48, 256, 0, 128), (32, 384, 128, 384), (5120, 5120, 128, 128)])
@pytest.mark.parametrize("m", [1, 3, 7, 16, 21])
def test_gemv_w3_vs_oracle(n, k, r, g, m):
    from qeft_amd import qeft_cuda
    if n * k > 2 ** 24 and m not in (1, 7):
        pytest.skip("full-size layers: batch 1 and 7 only (oracle time)")
    bufs = O.make_layer(n, k, r, g, seed=n + k, bits=3)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=m)
    szp = qeft_cuda.pack_scales(t["scales"], t["scaled_zeros"], n, k, g)
    for shadow in (szp, None):
        y = qeft_cuda.gemv_3bit(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                t.get("oweight_interleaved"), None, None, m, n, k, g, shadow)
        torch.cuda.synchronize()
        yref = _ref(bufs, x, g)
        assert rel_err(y.cpu().numpy(), yref) < REL_TOL
        assert elem_err_ok(y.cpu().numpy(), yref, rtol=2e-3, atol_scale=2e-3)


def test_gemv_w3_bias_residual_and_errors():
    from qeft_amd import _lib, qeft_cuda
    n, k, r, g = 256, 512, 128, 128
    bufs = O.make_layer(n, k, r, g, seed=2, bits=3)
    t = layer_to_torch(bufs, DEV)
    x = torch.from_numpy(O.make_activation(2, k, r, seed=9)).to(DEV)
    bias = torch.randn(n, device=DEV).half()
    res = torch.randn(2, n, device=DEV).half()
    y0 = qeft_cuda.gemv_3bit(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"], None, None, 2, n, k, g)
    y1 = qeft_cuda.gemv_3bit(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"], bias, res, 2, n, k, g)
    torch.cuda.synchronize()
    want = (y0.float() + bias.float() + res.float())
    assert (y1.float() - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 1e-3
    lib = _lib.lib()
    y = torch.empty(1, n, device=DEV, dtype=torch.float16)
    # K not a multiple of 128 / outlier slice not whole steps: no 3-bit layout exists for them
    assert lib.qeft_gemv_w3(x.data_ptr(), t["qweight"].data_ptr(), t["scales"].data_ptr(), t["scaled_zeros"].data_ptr(),
                            None, None, None, None, y.data_ptr(), 1, n, 448, 64, 0, _st()) != 0
    assert lib.qeft_expand_w3(t["qweight"].data_ptr(), y.data_ptr(), n, k, 64, _st()) != 0


@pytest.mark.parametrize("n,k,r", [(256, 512, 128), (4096, 4096, 128), (64, 256, 0)])
def test_expand_w3_is_the_4bit_layout_bit_for_bit(n, k, r):
    from qeft_amd import qeft_cuda
    rng = np.random.default_rng(n)
    q = rng.integers(0, 8, (n, k - r)).astype(np.int32)
    q3 = torch.from_numpy(O.pack_w3(q)).to(DEV)
    got = qeft_cuda.expand_3bit(q3, n, k, r)
    torch.cuda.synchronize()
    full = np.concatenate([q, np.zeros((n, r), dtype=np.int32)], axis=1)      # dead zero nibbles under the fp16 columns
    assert got.dtype == torch.int16 and np.array_equal(got.cpu().numpy(), O.pack_intweight(full))


@pytest.mark.parametrize("ns,k", [((512, 256, 256), 512), ((11008, 11008), 4096)])
def test_w3_group_norm_and_silu_variants(ns, k):
    from qeft_amd import _lib, qeft_cuda
    lib, r, g = _lib.lib(), 128, 128
    bufs = [O.make_layer(n, k, r, g, seed=40 + i, bits=3) for i, n in enumerate(ns)]
    layers = [layer_to_torch(b, DEV) for b in bufs]
    torch.manual_seed(7)
    x = (torch.randn(1, k, device=DEV) * 2).half()
    gamma = (1 + 0.1 * torch.randn(k, device=DEV)).half()
    ys = [torch.empty(1, n, device=DEV, dtype=torch.float16) for n in ns]
    szp = [qeft_cuda.pack_scales(l["scales"], l["scaled_zeros"], n, k, g) for l, n in zip(layers, ns)]
    nn_ = (ctypes.c_int * len(ns))(*ns)
    _lib.check(lib.qeft_gemv_w3_group(x.data_ptr(), gamma.data_ptr(), 1e-5, len(ns), _arr([l["qweight"] for l in layers]),
                                      _arr([l["scales"] for l in layers]), _arr([l["scaled_zeros"] for l in layers]),
                                      _arr([l["oweight_interleaved"] for l in layers]), None, _arr(szp), _arr(ys), nn_,
                                      k, g, r, _st()))
    torch.cuda.synchronize()
    x64 = x.cpu().numpy().astype(np.float64)[0]
    xn = x64 / np.sqrt((x64 ** 2).mean() + 1e-5) * gamma.cpu().numpy().astype(np.float64)
    for y, bf in zip(ys, bufs):
        w = O.dequant_dense(bf["qweight"], bf["scales"], bf["scaled_zeros"], bf["oweight"], g).astype(np.float64)
        assert rel_err(y.cpu().numpy()[0], w @ xn) < REL_TOL
    # silu(gate) * up folded into the staging == the unfused sequence on the same 3-bit layer, bit for bit
    l, n = layers[0], ns[0]
    gate = (torch.randn(k, device=DEV) * 2).half()
    up = torch.randn(k, device=DEV).half()
    res = torch.randn(1, n, device=DEV).half()
    act = torch.empty(1, k, device=DEV, dtype=torch.float16)
    _lib.check(lib.qeft_silu_mul(gate.data_ptr(), up.data_ptr(), act.data_ptr(), k, _st()))
    y0 = qeft_cuda.gemv_3bit(act, l["qweight"], l["scales"], l["scaled_zeros"], l["oweight_interleaved"], None, res, 1, n, k, g, szp[0])
    y1 = torch.empty(1, n, device=DEV, dtype=torch.float16)
    _lib.check(lib.qeft_gemv_w3_silu(gate.data_ptr(), up.data_ptr(), l["qweight"].data_ptr(), l["scales"].data_ptr(),
                                     l["scaled_zeros"].data_ptr(), l["oweight_interleaved"].data_ptr(), None,
                                     res.data_ptr(), szp[0].data_ptr(), y1.data_ptr(), n, k, g, r, _st()))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("name", ["model.layers.0.mlp.up_proj", "model.layers.0.self_attn.o_proj"])
def test_quantlinear_w3_forward_and_finetune_backward(name):
    """BASELINE config 5: w3 layer, forward at decode / few-row / GEMM row counts and one fwd/bwd step with the fp16
    outlier slice trainable (the batched dequant-GEMM backward), against the oracle."""
    from qeft_amd.llama import synthetic_quantlinear
    k, n, r, g = 512, 384, 128, 128
    oidx = torch.randperm(k, generator=torch.Generator().manual_seed(3))[:r].sort().values.to(torch.int32) \
        if "o_proj" in name else None
    ql = synthetic_quantlinear(name, k, n, r, g, 77, DEV, oidx, bits=3)
    assert ql.bits == 3 and ql.qweight.dtype == torch.int32
    sd = {kk: v.cpu().numpy() for kk, v in ql.state_dict().items()}
    ids = sd.get("reorder_ids")
    for m in (1, 5, 16, 40, 300):
        x = O.make_activation(m, k, r, seed=m)
        y = ql(torch.from_numpy(x).to(DEV))
        torch.cuda.synchronize()
        yref = O.quant_linear(x, sd["qweight"], sd["scales"], sd["scaled_zeros"], sd["oweight"], None, g, reorder_ids=ids)
        assert y.shape == (m, n) and rel_err(y.cpu().numpy(), yref) < REL_TOL, m
    # fine-tune step: d(loss)/dx and d(loss)/d(oweight)
    ql.set_kernel(training=True)
    ql.set_for_wct()
    m = 96
    x = torch.from_numpy(O.make_activation(m, k, r, seed=5)).to(DEV).requires_grad_(True)
    dy = torch.randn(m, n, device=DEV).half()
    y = ql(x)
    y.backward(dy)
    torch.cuda.synchronize()
    xin = x.detach().cpu().numpy()
    xg = xin[:, ids] if ids is not None else xin
    dx_ref, dow_ref = O.quant_linear_backward(dy.cpu().numpy(), xg, sd["qweight"], sd["scales"], sd["scaled_zeros"],
                                              sd["oweight"], g)
    if ids is not None:                       # gradient w.r.t. the un-gathered input
        full = np.zeros_like(dx_ref)
        full[:, ids] = dx_ref
        dx_ref = full
    assert rel_err(x.grad.cpu().numpy(), dx_ref) < 2e-3
    assert rel_err(ql.oweight.grad.cpu().numpy(), dow_ref) < 2e-3


@pytest.mark.parametrize("use_graph", [False, True])
def test_decode_engine_w3_matches_dense_model(use_graph):
    from qeft_amd.llama import DecodeEngine, QuantLlama, nll_from_logits, tiny_shape
    shape = dataclasses.replace(tiny_shape(n_layers=2, hidden=256, inter=512, n_heads=2, vocab=384, max_seq=64), bits=3)
    model = QuantLlama(shape, DEV, seed=4)
    assert model.model.layers[0].mlp.down_proj.bits == 3
    eng = DecodeEngine(model, use_graph=use_graph)
    tokens = torch.randint(0, shape.vocab, (24,), generator=torch.Generator().manual_seed(0)).to(DEV)
    got = eng.teacher_forced_logits(tokens)
    ref = model.forward_dense_reference(tokens)
    torch.cuda.synchronize()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-2
    assert abs(nll_from_logits(got, tokens) - nll_from_logits(ref, tokens)) < 5e-3
    # a w3 layer streams 3/4 of the packed-weight bytes of its w4 twin
    w4 = DecodeEngine(QuantLlama(dataclasses.replace(shape, bits=4), DEV, seed=4), use_graph=False)
    assert eng.weight_bytes_per_token() < w4.weight_bytes_per_token()


def _w3_operand(n, k, r, g, seed):
    import types
    from qeft_amd import qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=seed, bits=3)
    t = layer_to_torch(bufs, DEV)
    l = types.SimpleNamespace(qweight=t["qweight"], scales=t["scales"], scaled_zeros=t["scaled_zeros"], oweight=t.get("oweight"),
                              bias=None, outfeatures=n, infeatures=k, group_size=g, outlierfeatures=r, bits=3)
    l.sz_packed = qeft_cuda.pack_scales(l.scales, l.scaled_zeros, n, k, g)
    return l, bufs


@pytest.mark.parametrize("n,k,r", [(16, 256, 128), (16, 128, 0), (256, 1024, 128), (4096, 4096, 128), (11008, 4096, 128),
                                   (4096, 11008, 128), (5120, 13824, 128), (16 * 513, 384, 128), (12288, 4096, 128)])
def test_v3_gemv_on_the_3bit_stream(n, k, r):
    """qeft_decode_linear_w3 (the round-2 GEMV on the 3-bit layout: raw x, every field shifted to one scale class) vs oracle."""
    from qeft_amd import _lib, qeft_cuda
    l, bufs = _w3_operand(n, k, r, 128, seed=n + k)
    x = O.make_activation(1, k, r, seed=4)
    y = qeft_cuda.decode_linear(torch.from_numpy(x[0]).to(DEV), l)
    assert _lib.last_variant() == "gemv_v3_w3"
    torch.cuda.synchronize()
    yref = _ref(bufs, x, 128)
    assert rel_err(y.cpu().numpy()[None], yref) < REL_TOL
    assert elem_err_ok(y.cpu().numpy()[None], yref, rtol=2e-3, atol_scale=2e-3)


@pytest.mark.parametrize("n,k", [(48, 1024), (1376, 4096), (11008, 4096)])
def test_v3_w3_concat_and_pair(n, k):
    """Derived 3-bit operands: q|k|v-style concatenation and the gate|up pair interleave (12-byte lane records moved whole)
    with the SiLU epilogue, plus the fp32 residual / producer-norm forms."""
    from qeft_amd import _lib, fuse, qeft_cuda
    r, g = 128, 128
    (lg, bg), (lu, bu) = _w3_operand(n, k, r, g, seed=n), _w3_operand(n, k, r, g, seed=n + 1)
    x = O.make_activation(1, k, r, seed=5)
    xt = torch.from_numpy(x[0]).to(DEV)
    cat = qeft_cuda.decode_linear(xt, fuse.concat_linears([lg, lu]))
    act = qeft_cuda.decode_linear(xt, fuse.pair_interleave(lg, lu), mode=qeft_cuda.V3_PAIR)
    assert _lib.last_variant() == "gemv_v3_w3_pair"
    torch.cuda.synchronize()
    g64, u64 = _ref(bg, x, g)[0].astype(np.float64), _ref(bu, x, g)[0].astype(np.float64)
    assert rel_err(cat[:n].cpu().numpy(), g64) < REL_TOL and rel_err(cat[n:].cpu().numpy(), u64) < REL_TOL
    assert rel_err(act.cpu().numpy(), g64 / (1 + np.exp(-g64)) * u64) < 2e-3
    if n == k:
        return
    # residual + producer norm on a square-free shape is covered by the 4-bit tests; here: the fp32 residual form
    h0 = np.random.default_rng(1).standard_normal(n).astype(np.float32)
    y32 = qeft_cuda.decode_linear(xt, lg, residual=torch.from_numpy(h0).to(DEV))
    torch.cuda.synchronize()
    assert rel_err(y32.cpu().numpy(), h0.astype(np.float64) + g64) < REL_TOL


@pytest.mark.parametrize("m,n,k,r,g,tile", [(2048, 4096, 4096, 128, 128, "256x128"), (2048, 11008, 4096, 128, 128, "256x128"),
                                            (2048, 4096, 11008, 128, 128, "256x128"), (1030, 5904, 1024, 0, 256, "256x128"),
                                            (600, 4096, 1024, 128, 128, "128x128"), (513, 6160, 640, 128, 64, "128x128")])
def test_gemm_forward_on_the_3bit_stream(m, n, k, r, g, tile):
    """BASELINE config 5 (round 3): the prefill / fine-tune forward of a 3-bit layer WITHOUT the 3 -> 4-bit expansion pass -- the
    loader-wave GEMM tiers read the 12-byte lane records of the 3-bit stream (qeft_gemm_w3).  M = 2048 on the three 7B shapes,
    ragged tiles, no outlier slice / group 256 / 64, both tile sizes; sampled columns vs the oracle, variant asserted."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=n // 16 + k + 3, bits=3, bias=True)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=21)
    y = qeft_cuda.gemm_3bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                 t.get("oweight") if r else None, t["bias"])
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == f"gemm_v3_{tile}_w3", variant
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None, g)
    rows = np.unique(np.concatenate([np.arange(0, 48), np.arange(n - 48, n), np.random.default_rng(1).integers(0, n, 200)]))
    yref = x.astype(np.float64) @ w[rows].astype(np.float64).T + bufs["bias"][rows].astype(np.float64)
    y = y.cpu().numpy()
    assert y.shape == (m, n)
    assert rel_err(y[:, rows], yref) < REL_TOL
    assert elem_err_ok(y[:, rows], yref)
    # and bit-equal to the expansion route (the same fp16 weights, the same kernel arithmetic)
    y4 = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), qeft_cuda.expand_3bit(t["qweight"], n, k, r), t["scales"],
                                  t["scaled_zeros"], t.get("oweight") if r else None, t["bias"])
    torch.cuda.synchronize()
    assert np.array_equal(y, y4.cpu().numpy())


@pytest.mark.parametrize("m,n,k,r,g", [(2048, 4096, 4096, 128, 128), (2048, 11008, 4096, 128, 128), (2048, 4096, 11008, 128, 128),
                                       (1100, 448, 6144, 128, 128), (1025, 320, 6144, 0, 256)])
def test_gemm_dx_on_the_3bit_stream(m, n, k, r, g):
    """dX = dY . Wdeq of a 3-bit layer from the 3-bit stream (qeft_gemm_w3_dx; the loader waves unpack 12-byte records into the
    fp16 weight tile): M = 2048 on the 7B shapes, ragged M, a short n loop (7 / 5 tiles: the register ring's remainders)."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=n // 16 + k + 4, bits=3)
    t = layer_to_torch(bufs, DEV)
    dy = (np.random.default_rng(5).standard_normal((m, n)) * 0.1).astype(np.float16)
    dx = qeft_cuda.gemm_3bit_dx(torch.from_numpy(dy).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                t.get("oweight") if r else None, k)
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == "dx256_w3", variant
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None, g)
    cols = np.unique(np.concatenate([np.arange(0, 64), np.arange(k - 192, k), np.random.default_rng(2).integers(0, k, 200)]))
    ref = dy.astype(np.float64) @ w[:, cols].astype(np.float64)
    dx = dx.cpu().numpy()
    assert dx.shape == (m, k)
    assert rel_err(dx[:, cols], ref) < REL_TOL
    assert elem_err_ok(dx[:, cols], ref)
