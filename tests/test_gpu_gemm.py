"""GPU parity: MFMA GEMM forward / backward and dense dequant vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import qeft_oracle as O
from util import REL_TOL, elem_err_ok, layer_to_torch, oracle_forward, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,k,r,g", [(8, 128, 0, 128), (16, 256, 128, 128), (64, 512, 64, 128), (256, 1024, 128, 128),
                                     (24, 384, 0, 128), (128, 640, 96, 32), (64, 2048, 128, 2048)])
def test_dequant_dense_bit_exact(n, k, r, g):
    from qeft_amd import qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=n)
    t = layer_to_torch(bufs, DEV)
    w = qeft_cuda.dequantize_weight_4bit_qeft(t["qweight"], t["scales"], t["scaled_zeros"],
                                              t.get("oweight") if r else None)
    torch.cuda.synchronize()
    ref = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None,
                          g, round_fp16=True).astype(np.float16)
    assert np.array_equal(w.cpu().numpy().view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("m", [1, 8, 33, 64, 128, 200, 300])
@pytest.mark.parametrize("n,k,r,g", [(256, 1024, 128, 128), (136, 512, 0, 128), (384, 640, 64, 64)])
def test_gemm_forward(m, n, k, r, g):
    from qeft_amd import qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=m + n, bias=True)
    x = O.make_activation(m, k, r, seed=m)
    t = layer_to_torch(bufs, DEV)
    xt = torch.from_numpy(x).to(DEV)
    y = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None,
                                 t["bias"])
    torch.cuda.synchronize()
    yref = oracle_forward(bufs, x, r, g).astype(np.float64)
    y = y.cpu().numpy()
    assert rel_err(y, yref) < REL_TOL
    assert elem_err_ok(y, yref)


def test_gemm_4bit_reference_semantics_uses_dead_nibbles():
    """gemm_4bit ignores oweight (reference gemm_cuda.cu): every column from the nibbles; the reference's
    forward then adds F.linear on the outlier slice (qlinear.py:265-266).  Both routes must agree."""
    from qeft_amd import qeft_cuda
    n, k, r, g, m = 256, 1024, 128, 128, 64
    bufs = O.make_layer(n, k, r, g, seed=1)
    x = O.make_activation(m, k, r, seed=1)
    t = layer_to_torch(bufs, DEV)
    xt = torch.from_numpy(x).to(DEV)
    y0 = qeft_cuda.gemm_4bit(xt, t["qweight"], t["scales"], t["scaled_zeros"])
    w_all = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], None, g)
    assert rel_err(y0.cpu().numpy(), O.linear(x, w_all).astype(np.float64)) < REL_TOL
    y1 = y0 + torch.nn.functional.linear(xt[..., -r:], t["oweight"])
    y2 = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"])
    torch.cuda.synchronize()
    yref = oracle_forward(bufs, x, r, g).astype(np.float64)
    assert rel_err(y2.cpu().numpy(), yref) < REL_TOL
    assert rel_err(y1.cpu().numpy(), yref) < 2e-3   # includes the dead-nibble residual the reference also has


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_gemm_llama_shapes_sampled(n, k):
    """Full-size layer, M=256: check against the oracle on a sample of output columns."""
    from qeft_amd import qeft_cuda
    r, g, m = 128, 128, 256
    bufs = O.make_layer(n, k, r, g, seed=2)
    x = O.make_activation(m, k, r, seed=2)
    t = layer_to_torch(bufs, DEV)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                 t["oweight"])
    torch.cuda.synchronize()
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], g)
    rows = np.unique(np.concatenate([np.arange(0, 64), np.arange(n - 64, n),
                                     np.random.default_rng(0).integers(0, n, 256)]))
    yref = x.astype(np.float64) @ w[rows].astype(np.float64).T
    assert rel_err(y.cpu().numpy()[:, rows], yref) < REL_TOL


@pytest.mark.parametrize("m", [1, 16, 130])
@pytest.mark.parametrize("n,k,r,g", [(256, 1024, 128, 128), (136, 512, 0, 128), (384, 640, 64, 64)])
def test_backward_dx_and_doweight(m, n, k, r, g):
    from qeft_amd import qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=m + k)
    x = O.make_activation(m, k, r, seed=m + 1)
    dy = (np.random.default_rng(m).standard_normal((m, n)) * 0.1).astype(np.float16)
    t = layer_to_torch(bufs, DEV)
    dx = qeft_cuda.gemm_4bit_dx(torch.from_numpy(dy).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                t.get("oweight") if r else None)
    dx_ref, dow_ref = O.quant_linear_backward(dy, x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"],
                                              bufs.get("oweight") if r else None, g)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu().numpy(), dx_ref.astype(np.float64)) < REL_TOL
    if r:
        dow = qeft_cuda.grad_oweight(torch.from_numpy(dy).to(DEV), torch.from_numpy(x).to(DEV), r)
        torch.cuda.synchronize()
        assert rel_err(dow.cpu().numpy(), dow_ref) < REL_TOL


@pytest.mark.parametrize("m,n,k,r", [(130, 8456, 256, 72), (1, 8200, 128, 8), (300, 16384, 192, 128), (257, 264, 320, 200)])
def test_doweight_block_widths_ragged(m, n, k, r):
    """d(oweight) = dY^T . x[:, K-r:] alone, against numpy fp64, on shapes that reach both block widths of the kernel (64
    columns of dy per block once N / 32 * ceil(r / 64) > 512) with N not a multiple of the block, r not a multiple of 64
    (a partly filled second j block, r > 128: a third) and M not a multiple of the 128-row slab."""
    from qeft_amd import _lib, qeft_cuda
    rng = np.random.default_rng(m + n)
    x = rng.standard_normal((m, k)).astype(np.float16)
    dy = (rng.standard_normal((m, n)) * 0.1).astype(np.float16)
    dow = qeft_cuda.grad_oweight(torch.from_numpy(dy).to(DEV), torch.from_numpy(x).to(DEV), r)
    variant = _lib.last_variant()
    torch.cuda.synchronize()
    assert variant == ("grad_oweight_mfma_n64" if (n + 31) // 32 * ((r + 63) // 64) > 512 else "grad_oweight_mfma"), variant
    ref = dy.astype(np.float64).T @ x[:, k - r:].astype(np.float64)
    assert dow.shape == (n, r)
    assert rel_err(dow.cpu().numpy(), ref) < REL_TOL


@pytest.mark.parametrize("m,n,k,r,fused", [(2048, 3584, 512, 128, True), (2000, 3592, 384, 0, True), (1100, 5896, 448, 64, True),
                                            (1100, 7176, 448, 64, False),      # 285 tiles of 256 rows = two rounds: another tier is cheaper
                                            (64, 512, 512, 128, False), (300, 1024, 256, 0, False)])
def test_gemm_silu_mul_epilogue(m, n, k, r, fused):
    """silu(gate) * (x . W^T + bias) from one launch on the 256-row tier == the GEMM followed by qeft_silu_mul, bit for bit
    (the product is rounded to fp16 before the activation in both), and within tolerance of float64; other tiers take the
    two-launch form inside the same entry."""
    from qeft_amd import _lib, qeft_cuda
    bufs = O.make_layer(n, k, r, 128 if k % 128 == 0 else 64, seed=m + n, bias=True)
    g = 128 if k % 128 == 0 else 64
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=m)
    gate = np.random.default_rng(n).standard_normal((m, n)).astype(np.float16)
    xt, gt = torch.from_numpy(x).to(DEV), torch.from_numpy(gate).to(DEV)
    ow = t.get("oweight") if r else None
    y = qeft_cuda.gemm_4bit_qeft_silu_mul(xt, t["qweight"], t["scales"], t["scaled_zeros"], ow, gt, t["bias"])
    variant = _lib.last_variant()
    up = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], ow, t["bias"])
    two = torch.empty_like(up)
    _lib.check(_lib.lib().qeft_silu_mul(gt.data_ptr(), up.data_ptr(), two.data_ptr(), up.numel(),
                                        torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert ("+silu" in variant) == fused, variant
    assert torch.equal(y, two)
    ref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None,
                         bufs["bias"], g).astype(np.float64)
    g64 = gate.astype(np.float64)
    assert rel_err(y.cpu().numpy(), g64 / (1 + np.exp(-g64)) * ref) < REL_TOL


@pytest.mark.parametrize("m,n,k,r", [(2048, 1792, 512, 128), (300, 256, 384, 0), (1030, 3520, 448, 64), (8, 64, 384, 128)])
def test_gemm_gateup_one_launch(m, n, k, r):
    """gate_proj and up_proj as one GEMM over the 64-row-interleaved operand with SiLU(gate) * up as its epilogue == the two
    GEMMs followed by qeft_silu_mul: bit for bit where those take the same kernel, within fp16 rounding where the separate
    launches fall to another tier; at any row count the entry says it supports."""
    import types
    from qeft_amd import _lib, fuse, qeft_cuda
    g = 128 if k % 128 == 0 else 64
    ls = []
    for seed in (1, 2):
        b = O.make_layer(n, k, r, g, seed=seed + n, bias=True)
        t = layer_to_torch(b, DEV)
        ls.append(types.SimpleNamespace(qweight=t["qweight"], scales=t["scales"], scaled_zeros=t["scaled_zeros"], oweight=t.get("oweight"),
                                        bias=t["bias"], outfeatures=n, infeatures=k, group_size=g, outlierfeatures=r, bits=4))
    op = fuse.pair64_gemm_operand(*ls)
    x = torch.from_numpy(O.make_activation(m, k, r, seed=m)).to(DEV)
    assert qeft_cuda.gemm_gateup_supported(m, op)
    y = qeft_cuda.gemm_4bit_gateup(x, op)
    assert _lib.last_variant() == "gemm_v3_256x128+silu_pair"
    gate, up = (qeft_cuda.gemm_4bit_qeft(x, l.qweight, l.scales, l.scaled_zeros, l.oweight if r else None, l.bias) for l in ls)
    same_kernel = _lib.last_variant() == "gemm_v3_256x128"        # smaller problems: another tier, another summation order
    two = torch.empty_like(up)
    _lib.check(_lib.lib().qeft_silu_mul(gate.data_ptr(), up.data_ptr(), two.data_ptr(), up.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert y.shape == (m, n)
    if same_kernel:
        assert torch.equal(y, two)
    else:
        assert rel_err(y.cpu().numpy(), two.float().cpu().numpy()) < 2e-3


def test_pack_oweight_device_bit_exact():
    from qeft_amd import qeft_cuda
    ow = (np.random.default_rng(0).standard_normal((64, 128))).astype(np.float16)
    got = qeft_cuda.pack_oweight_device(torch.from_numpy(ow).to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint16), O.pack_oweight(ow).view(np.uint16))


@pytest.mark.parametrize("m,n,k,r", [(200, 512, 4096, 128), (130, 256, 2048, 0), (520, 1024, 11008, 128)])
def test_gemm_split_k_path_vs_oracle(m, n, k, r):
    """Mid-size M: the K loop is cut into S parts through an fp32 workspace and summed in a fixed order."""
    from qeft_amd import _lib, qeft_cuda
    lib, g = _lib.lib(), 128
    need = lib.qeft_gemm_w4_workspace_bytes(m, n, k, r)
    assert need >= 2 * m * n * 4, "shape chosen so that the split path is taken"
    assert lib.qeft_gemm_w4_workspace_bytes(4096, 4096, 4096, 128) == 0      # enough tiles: no split
    assert lib.qeft_gemm_w4_workspace_bytes(8, 4096, 4096, 128) == 0         # few rows: the GEMV route
    bufs = O.make_layer(n, k, r, g, seed=m)
    t = layer_to_torch(bufs, DEV)
    x = O.make_activation(m, k, r, seed=3)
    xt = torch.from_numpy(x).to(DEV)
    bias = torch.randn(n, device=DEV).half()
    y = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight"), bias)
    y2 = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight"), bias)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)                                                  # fixed summation order
    yref = O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight"),
                          bias.cpu().numpy(), g)
    assert rel_err(y.cpu().numpy(), yref) < REL_TOL
    # a workspace that only fits 2 parts, and none at all: same result within tolerance (other summation order)
    out = torch.empty_like(y)
    ws = torch.empty(2 * m * n, dtype=torch.float32, device=DEV)
    for wsp, nbytes in ((ws.data_ptr(), ws.numel() * 4), (None, 0)):
        _lib.check(lib.qeft_gemm_w4_ws(xt.data_ptr(), t["qweight"].data_ptr(), t["scales"].data_ptr(),
                                       t["scaled_zeros"].data_ptr(), t["oweight"].data_ptr() if r else None,
                                       bias.data_ptr(), out.data_ptr(), wsp, nbytes, m, n, k, g, r,
                                       torch.cuda.current_stream(DEV).cuda_stream))
        torch.cuda.synchronize()
        assert rel_err(out.cpu().numpy(), yref) < REL_TOL


@pytest.mark.parametrize("m,n,k,r", [(300, 4096, 512, 128), (130, 2560, 256, 0)])
def test_dx_split_over_n_vs_oracle(m, n, k, r):
    """Backward wrt the input with few output tiles and a long contraction: the n loop is cut into parts (fp32 partials
    in a workspace, fixed summation order)."""
    from qeft_amd import _lib, qeft_cuda
    g = 128
    assert _lib.lib().qeft_gemm_w4_dx_workspace_bytes(m, n, k) >= 2 * m * k * 4
    assert _lib.lib().qeft_gemm_w4_dx_workspace_bytes(4096, 4096, 4096) == 0
    bufs = O.make_layer(n, k, r, g, seed=m + 1)
    t = layer_to_torch(bufs, DEV)
    dy = (torch.randn(m, n, generator=torch.Generator().manual_seed(2)) * 0.1).half()
    x = O.make_activation(m, k, r, seed=4)
    dx = qeft_cuda.gemm_4bit_dx(dy.to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight"))
    dx2 = qeft_cuda.gemm_4bit_dx(dy.to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight"))
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
    dx_ref, _ = O.quant_linear_backward(dy.numpy(), x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"],
                                        bufs.get("oweight"), g)
    assert rel_err(dx.cpu().numpy(), dx_ref) < 2e-3
