"""GPU parity at the sizes BASELINE.json's configs run at -- the kernels a prefill (M = 2048), a fine-tune step
(M = 2048 backward) and a row-sharded 13B decode actually dispatch to.  Every case asserts the variant the launch took
(qeft_last_variant), so coverage cannot fall off a routing threshold unnoticed.  Oracle time is kept small by checking a
sample of output columns (first / last / random), never by shrinking the launch."""
import functools

import numpy as np
import pytest
import torch

from oracle import qeft_oracle as O
from util import REL_TOL, elem_err_ok, layer_to_torch, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
R, G, M = 128, 128, 2048
SHAPES = [(4096, 4096), (11008, 4096), (4096, 11008)]


@functools.lru_cache(maxsize=2)
def _layer(n, k):
    bufs = O.make_layer(n, k, R, G, seed=n // 7 + k)
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], G)   # fp32 [N, K]
    return bufs, w


def _sample(n, count=256, seed=0):
    return np.unique(np.concatenate([np.arange(0, 64), np.arange(n - 64, n),
                                     np.random.default_rng(seed).integers(0, n, count)]))


@pytest.mark.parametrize("n,k", SHAPES)
def test_gemm_forward_m2048(n, k):
    """BASELINE config 3 (prefill seq = 2048): the tier the reference calls T2 (gemm_cuda.cu:1005-1029)."""
    from qeft_amd import _lib, qeft_cuda
    bufs, w = _layer(n, k)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(M, k, R, seed=5)
    bias = (np.random.default_rng(1).standard_normal(n) * 0.1).astype(np.float16)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"],
                                 torch.from_numpy(bias).to(DEV))
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant in ("gemm_v3_256x128", "gemm_v2_128x256"), variant       # the M >= 2048 tiers, never the 128x128 tile
    rows = _sample(n)
    yref = x.astype(np.float64) @ w[rows].astype(np.float64).T + bias[rows].astype(np.float64)
    y = y.cpu().numpy()
    assert y.shape == (M, n)
    assert rel_err(y[:, rows], yref) < REL_TOL
    assert elem_err_ok(y[:, rows], yref)          # element-wise: an error confined to small outputs does not hide behind the max norm
    # every M tile and every N tile carries data: no all-zero 128 x 128 block anywhere in the output
    blocks = np.abs(y.astype(np.float32)).reshape(M // 128, 128, n // 128, 128).max(axis=(1, 3))
    assert (blocks > 0).all()


@pytest.mark.parametrize("n,k", SHAPES)
def test_gemm_dx_and_grad_oweight_m2048(n, k):
    """BASELINE config 5: the backward of one fine-tune step at M = 2048 (QuantMatMulQEFT.backward, qlinear.py:30-44)."""
    from qeft_amd import _lib, qeft_cuda
    bufs, w = _layer(n, k)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(M, k, R, seed=6)
    dy = (np.random.default_rng(7).standard_normal((M, n)) * 0.1).astype(np.float16)
    dyt, xt = torch.from_numpy(dy).to(DEV), torch.from_numpy(x).to(DEV)
    dx = qeft_cuda.gemm_4bit_dx(dyt, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"])
    v_dx = _lib.last_variant()
    dow = qeft_cuda.grad_oweight(dyt, xt, R)
    v_dow = _lib.last_variant()
    torch.cuda.synchronize()
    assert v_dx in ("dx256", "dx128"), v_dx
    assert v_dow == ("grad_oweight_mfma_n64" if n > 8192 else "grad_oweight_mfma"), v_dow     # 64-column blocks for wide layers
    cols = np.unique(np.concatenate([_sample(k - R), np.arange(k - R, k)]))      # sample of INT4 columns + the whole fp16 slice
    dx_ref = dy.astype(np.float64) @ w[:, cols].astype(np.float64)
    dx = dx.cpu().numpy()
    assert dx.shape == (M, k)
    assert rel_err(dx[:, cols], dx_ref) < REL_TOL
    assert elem_err_ok(dx[:, cols], dx_ref)
    dow_ref = dy.astype(np.float64).T @ x[:, k - R:].astype(np.float64)
    assert rel_err(dow.cpu().numpy(), dow_ref) < REL_TOL


@pytest.mark.parametrize("n,k", [(640, 5120), (1728, 5120), (640, 13824), (2560, 5120), (6912, 5120), (1376, 4096), (512, 11008)])
@pytest.mark.parametrize("m", [1, 3])
def test_gemv_row_shard_shapes(n, k, m):
    """BASELINE config 4: the per-rank row shards of Llama-2-13B at 8 and 2 GPUs (and of 7B at 8)."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, R, G, seed=n + k + m)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, R, seed=m)
    y = qeft_cuda.gemv_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                 t["oweight_interleaved"], m, n, k, G)
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == ("gemv_v3" if m == 1 else "gemv_v3_mb"), variant
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], None, G)
    assert rel_err(y.cpu().numpy(), yref.astype(np.float64)) < REL_TOL


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008), (5120, 5120), (13824, 5120)])
@pytest.mark.parametrize("m", [8, 11, 16])
def test_gemm_entry_8_to_16_rows_on_the_decode_gemv(n, k, m):
    """QuantLinear.forward sends 8 and more rows to gemm_4bit (+ F.linear on the outlier slice, qlinear.py:251-266).  Up to 16
    rows ride as A rows of the decode GEMV's MFMAs (one weight stream; the plain oweight rows, checkpoint-layout scales): every
    output vs the oracle, element-wise, and the variant the routing must take -- x rows staged in LDS where they fit, read by the
    lanes from global memory (gemv_v3_mb_xg) where they do not."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, R, G, seed=n + k + m, bias=True)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, R, seed=m)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"], t["bias"])
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    # (x rows that fit the block's LDS are staged there -- two launches for the widest blocks; longer ones are read from global memory)
    xg = k > 5120 or (k == 5120 and (m == 16 or (n == 13824 and m >= 11))) or (n == 11008 and m == 16)      # (the LDS plan of v3_lds)
    assert variant == ("gemv_v3_mb_xg" if xg else "gemv_v3_mb"), (variant, n, k, m)
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], bufs["bias"], G).astype(np.float64)
    got = y.cpu().numpy()
    assert got.shape == (m, n)
    assert rel_err(got, yref) < REL_TOL
    assert elem_err_ok(got, yref), (variant, n, k, m)
    # no outlier slice: gemm_4bit's own semantics (N from the qweight rows)
    if n == 4096 and k == 4096:
        b0 = O.make_layer(n, k, 0, G, seed=5)
        t0 = layer_to_torch(b0, DEV)
        y0 = qeft_cuda.gemm_4bit(torch.from_numpy(x).to(DEV), t0["qweight"], t0["scales"], t0["scaled_zeros"])
        assert _lib.last_variant() == "gemv_v3_mb"
        ref0 = O.quant_linear(x, b0["qweight"], b0["scales"], b0["scaled_zeros"], None, None, G).astype(np.float64)
        assert elem_err_ok(y0.cpu().numpy(), ref0)


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008), (5120, 5120), (13824, 5120)])
@pytest.mark.parametrize("m", [17, 24, 33, 48, 64])
def test_gemm_entry_17_to_64_rows_weight_stationary(n, k, m):
    """17 .. 64 rows at the GEMM entries (benchmark.py:118's 64-token prompt; the reference's M <= 32 / M <= 64 tiers,
    gemm_cuda.cu:952-978): ONE launch of the weight-stationary kernel (gemm_ws.hip) -- every output vs the oracle, element-wise,
    with and without a split-K workspace in reach, and the variant asserted."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, R, G, seed=n + k + m, bias=True)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, R, seed=m)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"], t["bias"])
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == "gemm_ws", (variant, n, k, m)
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], bufs["bias"], G).astype(np.float64)
    got = y.cpu().numpy()
    assert got.shape == (m, n)
    assert rel_err(got, yref) < REL_TOL
    assert elem_err_ok(got, yref), (variant, n, k, m)
    # the plain C entry without a workspace (no test covered m = 17 .. 64 there) reaches the same kernel
    lib = _lib.lib()
    y2 = torch.empty(m, n, dtype=torch.float16, device=DEV)
    xt = torch.from_numpy(x).to(DEV)
    _lib.check(lib.qeft_gemm_w4(xt.data_ptr(), t["qweight"].data_ptr(), t["scales"].data_ptr(), t["scaled_zeros"].data_ptr(),
                                t["oweight"].data_ptr(), t["bias"].data_ptr(), y2.data_ptr(), m, n, k, G, R,
                                torch.cuda.current_stream().cuda_stream))
    assert _lib.last_variant() == "gemm_ws"
    torch.cuda.synchronize()
    assert torch.equal(y2, y)
    if n == 4096 and k == 4096:
        # no outlier slice: gemm_4bit's own semantics (every column from the nibbles)
        b0 = O.make_layer(n, k, 0, G, seed=5)
        t0 = layer_to_torch(b0, DEV)
        y0 = qeft_cuda.gemm_4bit(xt, t0["qweight"], t0["scales"], t0["scaled_zeros"])
        assert _lib.last_variant() == "gemm_ws"
        ref0 = O.quant_linear(x, b0["qweight"], b0["scales"], b0["scaled_zeros"], None, None, G).astype(np.float64)
        assert elem_err_ok(y0.cpu().numpy(), ref0)


def test_weight_stationary_tier_ragged_and_small_shapes():
    """Row-set counts that do not divide over the blocks (short blocks repeat their last set), a single row set, K = 256 (one full
    step + the outlier step: most waves have no step at all), n_out = 0 with K = 128 k."""
    from qeft_amd import _lib, qeft_cuda
    for (n, k, r, m) in [(16, 256, R, 17), (48, 384, R, 40), (4112, 1024, R, 64), (1040 * 16, 512, R, 33), (272, 1152, 0, 25), (8208, 2048, R, 64)]:
        bufs = O.make_layer(n, k, r, G, seed=n + k + m)
        t = layer_to_torch(bufs, DEV)
        x = O.make_activation(m, k, max(r, 1), seed=m)
        y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"] if r else None)
        assert _lib.last_variant() == "gemm_ws", (_lib.last_variant(), n, k, r, m)
        torch.cuda.synchronize()
        yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"] if r else None, None, G).astype(np.float64)
        got = y.cpu().numpy()
        assert rel_err(got, yref) < REL_TOL, (n, k, r, m)
        assert elem_err_ok(got, yref), (n, k, r, m)


def test_variant_names_follow_the_routing():
    """The routing tiers below the BASELINE sizes keep their own names (and their own tests in test_gpu_gemm.py)."""
    from qeft_amd import _lib, qeft_cuda
    n, k = 512, 1024
    bufs = O.make_layer(n, k, R, G, seed=3)
    t = layer_to_torch(bufs, DEV)
    seen = {}
    for m in (8, 200, 1024):
        x = torch.from_numpy(O.make_activation(m, k, R, seed=m)).to(DEV)
        qeft_cuda.gemm_4bit_qeft(x, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"])
        seen[m] = _lib.last_variant()
    torch.cuda.synchronize()
    assert seen[8] == "gemv_v3_mb", seen                         # up to 16 rows: A rows of the decode GEMV's MFMAs
    assert seen[200] == "gemm_v3_128x128+splitk", seen           # 8 tiles: two blocks per tile through the split-K workspace
    dy = torch.zeros(64, n, dtype=torch.float16, device=DEV)
    qeft_cuda.gemm_4bit_dx(dy, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"])
    assert _lib.last_variant().startswith("dx64")
    torch.cuda.synchronize()


@pytest.mark.parametrize("m,n,k,r,g", [(1100, 5896, 512, 64, 64), (2148, 3584, 1024, 0, 128), (1024, 7168, 640, 128, 128)])
def test_gemm_large_m_tile_edges(m, n, k, r, g):
    """The 256 x 128 tile of the M >= 2048 tier with everything ragged: M not a multiple of 256, N not a multiple of 128,
    group size 64, 64 / 0 outlier columns, a K loop barely longer than the DMA ring -- full output vs the oracle."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=m + n, bias=True)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=9)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                 t.get("oweight") if r else None, t["bias"])
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == "gemm_v3_256x128", variant
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None,
                          bufs["bias"], g).astype(np.float64)
    assert rel_err(y.cpu().numpy(), yref) < REL_TOL


@pytest.mark.parametrize("m,n,k,r,g", [(1100, 256, 6144, 128, 128), (1100, 320, 6144, 0, 256), (1280, 384, 6144, 128, 6144),
                                       (1100, 448, 6144, 128, 128), (1025, 512, 6144, 0, 128), (1100, 704, 6144, 128, 128)])
def test_dx_large_m_tile_edges(m, n, k, r, g):
    """The 256 x 128 tile of dX with everything ragged: M not a multiple of 256, every remainder of the n-tile count modulo the
    loader's 4-deep weight ring (4, 5, 6, 7, 8, 11 n-tiles), no outlier slice, group 256 and per-channel -- full output vs oracle."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=m + n)
    t = layer_to_torch(bufs, DEV)
    dy = (np.random.default_rng(n).standard_normal((m, n)) * 0.1).astype(np.float16)
    dx = qeft_cuda.gemm_4bit_dx(torch.from_numpy(dy).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                t.get("oweight") if r else None)
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == "dx256", variant
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None, g)
    ref = dy.astype(np.float64) @ w.astype(np.float64)
    assert rel_err(dx.cpu().numpy(), ref) < REL_TOL


def test_gemm_one_int4_ktile_keeps_the_128_row_kernels():
    """K - n_out = 64: exactly ONE INT4 k-tile.  The 256-row tier's k loop needs two (its odd-count prologue dequantises k-tile
    1, here an fp16 outlier tile), so the launcher must leave this shape -- otherwise inside the tier's domain (M = 1024,
    56 column tiles) -- to the 128-row kernels; full output vs the oracle (round-2 advisory)."""
    from qeft_amd import _lib, qeft_cuda
    m, n, k, r, g = 1024, 7168, 512, 448, 64
    bufs = O.make_layer(n, k, r, g, seed=17, bias=True)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=3)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"], t["bias"])
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant.startswith("gemm_v2"), variant
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], bufs["bias"], g).astype(np.float64)
    assert rel_err(y.cpu().numpy(), yref) < REL_TOL
    assert elem_err_ok(y.cpu().numpy(), yref)
    # two INT4 k-tiles: back on the 256-row tier
    bufs2 = O.make_layer(n, k, 384, g, seed=18)
    t2 = layer_to_torch(bufs2, DEV)
    x2 = O.make_activation(m, k, 384, seed=4)
    y2 = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x2).to(DEV), t2["qweight"], t2["scales"], t2["scaled_zeros"], t2["oweight"])
    assert _lib.last_variant() == "gemm_v3_256x128"
    torch.cuda.synchronize()
    yref2 = O.quant_linear(x2, bufs2["qweight"], bufs2["scales"], bufs2["scaled_zeros"], bufs2["oweight"], None, g).astype(np.float64)
    assert rel_err(y2.cpu().numpy(), yref2) < REL_TOL


@pytest.mark.parametrize("m,n,k,r,g", [(1024, 4096, 4096, 128, 128), (700, 5000, 1024, 64, 64), (513, 6144, 640, 128, 128),
                                       (1000, 4096, 1536, 0, 256), (960, 3200, 2048, 128, 2048), (130, 7168, 512, 128, 128)])
def test_gemm_mid_m_tier_128_row_tiles(m, n, k, r, g):
    """The 128 x 128 form of the loader-wave GEMM (round 3; the reference's tuned tiers for mid-size M: gemm_cuda.cu:952-1004):
    M = 512 .. 1024 on wide layers -- whole tiles, ragged M and N (N % 128 != 0, M % 128 != 0), group 64 / 256 / per-channel,
    64 / 0 outlier columns, a K loop barely longer than the rings (10 k-tiles).  Full output vs the oracle, variant asserted."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=m + n + k, bias=True)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=13)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                 t.get("oweight") if r else None, t["bias"])
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == "gemm_v3_128x128", variant
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None,
                          bufs["bias"], g).astype(np.float64)
    y = y.cpu().numpy()
    assert y.shape == (m, n)
    assert rel_err(y, yref) < REL_TOL
    assert elem_err_ok(y, yref)


@pytest.mark.parametrize("m,n,k,r,g", [(96, 4096, 4096, 128, 128), (200, 520, 1024, 128, 128), (100, 11008, 4096, 128, 128),
                                       (512, 4096, 11008, 128, 128), (17, 1000, 2048, 0, 256), (300, 4096, 1536, 64, 64)])
def test_gemm_split_k_on_the_loader_wave_tile(m, n, k, r, g):
    """Fewer than 192 tiles of 128 x 128: S blocks per tile contract K / S each (the fp16 outlier k-tiles fall to the last one),
    fp32 partial tiles through the workspace, the ordered reduce launch (round 3; deterministic).  M = 17 .. 512 (17 .. 64 rows of a
    group-128 shape go to the weight-stationary tier since round 4: the 17-row case here has group 256), ragged M and N, S = 2 .. 8, no outlier slice / 64 columns; full output vs the oracle."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=m + n + k, bias=True)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=31)
    xt = torch.from_numpy(x).to(DEV)
    y = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None, t["bias"])
    variant = _lib.last_variant()
    y2 = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None, t["bias"])
    torch.cuda.synchronize()
    assert variant == "gemm_v3_128x128+splitk", variant
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None,
                          bufs["bias"], g).astype(np.float64)
    y = y.cpu().numpy()
    assert y.shape == (m, n)
    assert rel_err(y, yref) < REL_TOL
    assert elem_err_ok(y, yref)
    assert np.array_equal(y, y2.cpu().numpy())          # ordered sum of the partials: run to run bit-identical


@pytest.mark.parametrize("m,n,k,r,g,bits", [(1024, 4096, 4096, 128, 128, 4), (600, 448, 6144, 128, 128, 4), (513, 320, 6144, 0, 256, 4),
                                            (1000, 11008, 4096, 128, 128, 4), (1024, 4096, 4096, 128, 128, 3), (530, 704, 6144, 128, 128, 3)])
def test_dx_mid_m_tier_128_row_tiles(m, n, k, r, g, bits):
    """dX on 128 x 128 loader-wave tiles (round 3): M = 1024 on K = 4096 was SLOWER than M = 2048 (128 tiles of 256 rows fell
    back to the older kernels: 80 vs 72 us).  Whole and ragged M, every remainder class of the loader's weight ring (5 / 7 / 11 /
    64 / 172 n-tiles), no outlier slice, group 256, the 3-bit stream; sampled columns vs the oracle, variant asserted."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=m + n + bits, bits=bits)
    t = layer_to_torch(bufs, DEV)
    dy = (np.random.default_rng(n).standard_normal((m, n)) * 0.1).astype(np.float16)
    dyt = torch.from_numpy(dy).to(DEV)
    ow = t.get("oweight") if r else None
    if bits == 3:
        dx = qeft_cuda.gemm_3bit_dx(dyt, t["qweight"], t["scales"], t["scaled_zeros"], ow, k)
    else:
        dx = qeft_cuda.gemm_4bit_dx(dyt, t["qweight"], t["scales"], t["scaled_zeros"], ow)
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == ("dx128v3_w3" if bits == 3 else "dx128v3"), variant
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None, g)
    cols = np.unique(np.concatenate([np.arange(0, 64), np.arange(k - 192, k), np.random.default_rng(3).integers(0, k, 256)]))
    ref = dy.astype(np.float64) @ w[:, cols].astype(np.float64)
    dx = dx.cpu().numpy()
    assert dx.shape == (m, k)
    assert rel_err(dx[:, cols], ref) < REL_TOL
    assert elem_err_ok(dx[:, cols], ref)
