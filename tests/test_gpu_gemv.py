"""GPU parity: decode GEMV (C ABI via the qeft_cuda shim) vs the CPU oracle.  fp16 tolerance 1e-3 relative."""
import numpy as np
import pytest
import torch

from oracle import qeft_oracle as O
from util import REL_TOL, elem_err_ok, layer_to_torch, oracle_forward, rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def v3_takes(n, k, r, g, residual=False):
    """Shapes the reference's gemv entries route to the round-2 kernel (include/qeft_hip.h, "Kernel reached by ...")."""
    return n % 16 == 0 and k % 128 == 0 and g in (128, k) and (r == 0 or (r == 128 and k > 128)) and not residual


def run_case(n, k, r, g, m, seed, bias=False, gather=False, residual=False, fused=True, szp=False):
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=seed, bias=bias)
    x = O.make_activation(m, k, r, seed=seed)
    t = layer_to_torch(bufs, DEV)
    xt = torch.from_numpy(x).to(DEV)
    ids = None
    if gather:
        rng = np.random.default_rng(seed)
        out_idx = np.sort(rng.choice(k, size=max(r, 1), replace=False))
        ids = O.sparse_to_dense_ids(out_idx, k)
    res = None
    if residual:
        res = (np.random.default_rng(seed + 5).standard_normal((m, n)) * 0.5).astype(np.float16)
    yref = oracle_forward(bufs, x, r, g, reorder_ids=ids).astype(np.float64)
    if res is not None:
        yref = (yref + res.astype(np.float64))
    if fused:
        y = qeft_cuda.gemv_4bit_fused(xt, t["qweight"], t["scales"], t["scaled_zeros"],
                                      t.get("oweight_interleaved") if r else None, t.get("bias"),
                                      torch.from_numpy(ids.astype(np.int32)).to(DEV) if ids is not None else None,
                                      torch.from_numpy(res).to(DEV) if res is not None else None, m, n, k, g,
                                      qeft_cuda.pack_scales(t["scales"], t["scaled_zeros"], n, k, g) if szp else None)
    elif r:
        y = qeft_cuda.gemv_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"],
                                     m, n, k, g)
    else:
        y = qeft_cuda.gemv_4bit(xt, t["qweight"], t["scales"], t["scaled_zeros"], m, n, k, g)
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    assert y.shape == (m, n) and y.dtype == np.float16
    assert rel_err(y, yref) < REL_TOL, (rel_err(y, yref), n, k, r, g, m, variant)
    assert elem_err_ok(y, yref), (n, k, r, g, m, variant)
    if v3_takes(n, k, r, g, residual):
        # (a batch whose x rows do not fit the block's LDS reads x from global memory -- _xg -- or, with a gather, goes as several
        #  launches; the last one names the variant)
        assert variant in ("gemv_v3", "gemv_v3_mb", "gemv_v3_mb_xg") and (m > 1 or variant == "gemv_v3"), (variant, n, k, r, g, m)
    else:
        assert variant in ("gemv_mfma", "gemv_valu"), (variant, n, k, r, g, m)
    return y


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 6, 7])
def test_gemv_qeft_small_batches(m):
    run_case(256, 1024, 128, 128, m, seed=m, fused=False)


@pytest.mark.parametrize("m", [1, 4, 7])
def test_gemv_plain_no_outliers(m):
    run_case(256, 1024, 0, 128, m, seed=10 + m, fused=False)


@pytest.mark.parametrize("n,k,r,g", [
    (8, 128, 0, 128), (8, 128, 64, 64), (16, 256, 128, 128), (64, 512, 64, 128), (24, 384, 0, 128),
    (40, 576, 32, 64), (512, 4096, 128, 128), (1376, 4096, 128, 128), (512, 11008, 128, 128),
    (64, 2048, 128, 2048),  # per-channel (group == K)
    (128, 640, 96, 32),
])
def test_gemv_shapes_and_groups(n, k, r, g):
    run_case(n, k, r, g, 1, seed=n + k)
    run_case(n, k, r, g, 3, seed=n + k + 1)


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008), (5120, 5120), (13824, 5120),
                                 (5120, 13824)])
def test_gemv_llama_shapes(n, k):
    run_case(n, k, 128, 128, 1, seed=7)


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008), (5120, 5120), (13824, 5120),
                                 (5120, 13824)])
def test_reference_entry_batches_1_to_7_on_llama_shapes(n, k):
    """gemv_4bit_qeft as the reference calls it (qlinear.py:253-263; m switch gemv_cuda_qeft.cu:433-466), m = 1..7 on the six
    Llama-2 7B / 13B shapes: one packed layer, every batch size, the round-2 kernel asserted (run_case)."""
    from qeft_amd import _lib, qeft_cuda
    r, g = 128, 128
    bufs = O.make_layer(n, k, r, g, seed=n // 16 + k)
    t = layer_to_torch(bufs, DEV)
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], g).astype(np.float32)
    for m in range(1, 8):
        x = O.make_activation(m, k, r, seed=100 + m)
        y = qeft_cuda.gemv_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                     t["oweight_interleaved"], m, n, k, g)
        variant = _lib.last_variant()
        torch.cuda.synchronize()
        yref = x.astype(np.float64) @ w.astype(np.float64).T
        # (K = 11008 / 13824: seven rows of x exceed the block's LDS: the lanes read their x fragments from global memory, _xg)
        assert variant == "gemv_v3" if m == 1 else variant in ("gemv_v3_mb", "gemv_v3_mb_xg"), (variant, m, k)
        assert variant != "gemv_v3_mb_xg" or k > 8192, (variant, m, k)             # short rows always fit
        assert variant == "gemv_v3_mb_xg" or m * k * 2 <= 160 * 1024, (variant, m, k)
        assert rel_err(y.cpu().numpy(), yref) < REL_TOL, (m, rel_err(y.cpu().numpy(), yref))
        assert elem_err_ok(y.cpu().numpy(), yref), m


@pytest.mark.parametrize("m", [1, 2, 5, 7])
def test_reference_entries_no_outliers_and_per_channel(m):
    """gemv_4bit (no outlier slice) and a per-channel layer (group == K: scales [1][N]) on the round-2 kernel."""
    run_case(4096, 4096, 0, 128, m, seed=20 + m, fused=False)
    run_case(512, 2048, 128, 2048, m, seed=30 + m, fused=False)
    run_case(16 * 513, 256, 128, 128, m, seed=40 + m, fused=False)     # row sets that do not divide over the blocks, one INT4 step
    run_case(48, 384, 0, 128, m, seed=50 + m, fused=False)


@pytest.mark.parametrize("m", [1, 3, 7])
def test_fused_entry_gather_bias_and_shadow(m):
    """QuantLinear.forward_outlier_out_proj's launch (qlinear.py:273-300): the reorder_ids gather inside the v3 launch, with
    the bias, with and without the sz_packed shadow; N = 11008 puts several row sets on a block."""
    run_case(4096, 4096, 128, 128, m, seed=60 + m, bias=True, gather=True)
    run_case(4096, 4096, 128, 128, m, seed=61 + m, gather=True, szp=True)
    run_case(11008, 4096, 128, 128, m, seed=62 + m, bias=True, szp=True)
    run_case(1376, 5120, 128, 128, m, seed=63 + m, bias=True, gather=True)


def test_gemv_fused_bias_gather_residual():
    run_case(256, 1024, 128, 128, 1, seed=3, bias=True)
    run_case(256, 1024, 128, 128, 2, seed=4, bias=True, gather=True)
    run_case(512, 4096, 128, 128, 1, seed=5, gather=True, residual=True)
    run_case(256, 1024, 0, 128, 5, seed=6, bias=True, gather=True, residual=True)


def test_gemv_batch_out_of_range_raises():
    """m outside 1..7 -> RuntimeError with the reference's message (gemv_cuda_qeft.cu:466)."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(64, 512, 128, 128, seed=0)
    t = layer_to_torch(bufs, DEV)
    for m in (8, 9):
        x = torch.zeros(m, 512, dtype=torch.float16, device=DEV)
        with pytest.raises(RuntimeError, match="Unsupported batch size for gemv kernel"):
            qeft_cuda.gemv_4bit_qeft(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"],
                                     m, 64, 512, 128)
        with pytest.raises(RuntimeError, match="Unsupported batch size for gemv kernel"):
            qeft_cuda.gemv_4bit(x, t["qweight"], t["scales"], t["scaled_zeros"], m, 64, 512, 128)
    x = torch.zeros(1, 512, dtype=torch.float16, device=DEV)
    code = _lib.lib().qeft_gemv_w4(x.data_ptr(), t["qweight"].data_ptr(), t["scales"].data_ptr(),
                                   t["scaled_zeros"].data_ptr(), x.data_ptr(), 0, 64, 512, 128, None)
    assert code == 1


def test_gemv_rows_are_independent_of_sharding():
    """N-sharding property: a row's result does not depend on which rows share its launch,
    as long as the block shape (rows per block) is the same -> shard outputs concatenate bit-exactly."""
    from qeft_amd import qeft_cuda
    n, k, r, g = 2048, 4096, 128, 128
    bufs = O.make_layer(n, k, r, g, seed=11)
    x = torch.from_numpy(O.make_activation(1, k, r, seed=11)).to(DEV)
    t = layer_to_torch(bufs, DEV)
    full = qeft_cuda.gemv_4bit_qeft(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"],
                                    1, n, k, g)
    parts = []
    for s in range(2):
        n0, n1 = s * n // 2, (s + 1) * n // 2
        parts.append(qeft_cuda.gemv_4bit_qeft(
            x, t["qweight"][n0 // 4:n1 // 4].contiguous(), t["scales"][:, n0:n1].contiguous(),
            t["scaled_zeros"][:, n0:n1].contiguous(), t["oweight_interleaved"][n0 // 2:n1 // 2].contiguous(),
            1, n1 - n0, k, g))
    torch.cuda.synchronize()
    got = torch.cat(parts, dim=-1)
    # fp32 sums are order-dependent only through the block shape; allow 1 fp16 ulp
    assert rel_err(got.cpu().numpy(), full.cpu().numpy()) < 1e-3


def test_shim_scale_shadow_follows_the_tensors():
    """gemv_4bit_qeft keeps a packed-scale shadow per (scales, zeros) pair of tensor objects (qeft_cuda._szp_shadow): the result is
    the one of the shadow-free launch bit for bit, an in-place update of the scales is seen at the next call (version counter),
    and a dead tensor leaves no entry behind."""
    import gc
    from qeft_amd import _lib, qeft_cuda
    n, k, R, G = 512, 1024, 128, 128
    bufs = O.make_layer(n, k, R, G, seed=21)
    t = layer_to_torch(bufs, DEV)
    x = torch.from_numpy(O.make_activation(1, k, R, seed=3)).to(DEV)
    qeft_cuda._SZP_CACHE.clear()
    y1 = qeft_cuda.gemv_4bit_qeft(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"], 1, n, k, G)
    assert len(qeft_cuda._SZP_CACHE) == 1 and _lib.last_variant() == "gemv_v3"
    y2 = qeft_cuda.gemv_4bit_qeft(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"], 1, n, k, G)      # hit
    lib = _lib.lib()
    y0 = torch.empty_like(y1)
    _lib.check(lib.qeft_gemv_w4_qeft(x.data_ptr(), t["qweight"].data_ptr(), t["scales"].data_ptr(), t["scaled_zeros"].data_ptr(),
                                     t["oweight_interleaved"].data_ptr(), y0.data_ptr(), 1, n, k, G, R, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(y1, y0) and torch.equal(y2, y0)
    t["scales"].mul_(2.0)                                 # in place: the shadow is stale now, the version counter says so
    y3 = qeft_cuda.gemv_4bit_qeft(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"], 1, n, k, G)
    yref = O.quant_linear(x.cpu().numpy(), bufs["qweight"], t["scales"].cpu().numpy(), bufs["scaled_zeros"], bufs["oweight"], None, G)
    assert rel_err(y3.cpu().numpy(), yref.astype(np.float64)) < REL_TOL
    # `.data = ...` keeps the tensor object and its version counter: the storage address in the stamp catches it
    t["scales"].data = (t["scales"].data * 0.5).clone()
    y4 = qeft_cuda.gemv_4bit_qeft(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight_interleaved"], 1, n, k, G)
    yref4 = O.quant_linear(x.cpu().numpy(), bufs["qweight"], t["scales"].cpu().numpy(), bufs["scaled_zeros"], bufs["oweight"], None, G)
    assert rel_err(y4.cpu().numpy(), yref4.astype(np.float64)) < REL_TOL
    del t, y1, y2, y3, y4
    gc.collect()
    assert len(qeft_cuda._SZP_CACHE) == 0
