"""GPU parity of the v3 decode GEMV (qeft_decode_linear, csrc/gemv_v3.h) vs the CPU oracle: plain, concatenated (q|k|v as one
operand) and pair-interleaved (gate|up with SiLU) operands, the producer-side RMSNorm (ssq partials + deferred 1/rms), the
fp32 residual epilogue, row-set counts that do not divide over the blocks."""
import types

import numpy as np
import pytest
import torch

from oracle import qeft_oracle as O
from util import REL_TOL, elem_err_ok, layer_to_torch, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make(n, k, r, g, seed, bias=False):
    """A packed layer on the GPU with the attributes decode_linear reads (what QuantLinear.set_kernel provides)."""
    from qeft_amd import qeft_cuda
    bufs = O.make_layer(n, k, r, g, seed=seed, bias=bias)
    t = layer_to_torch(bufs, DEV)
    l = types.SimpleNamespace(qweight=t["qweight"], scales=t["scales"], scaled_zeros=t["scaled_zeros"],
                              oweight=t.get("oweight"), bias=t.get("bias"), outfeatures=n, infeatures=k, group_size=g,
                              outlierfeatures=r)
    l.sz_packed = qeft_cuda.pack_scales(l.scales, l.scaled_zeros, n, k, g)
    return l, bufs


def ref(bufs, x, r, g):
    return O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs.get("oweight") if r else None,
                          bufs.get("bias"), g).astype(np.float64)


@pytest.mark.parametrize("n,k,r,g", [
    (16, 256, 128, 128), (16, 128, 0, 128), (32, 384, 128, 128), (256, 1024, 128, 128), (256, 1024, 0, 128),
    (512, 4096, 128, 128), (4096, 4096, 128, 128), (11008, 4096, 128, 128), (4096, 11008, 128, 128),
    (5120, 5120, 128, 128), (13824, 5120, 128, 128), (5120, 13824, 128, 128), (640, 5120, 128, 128), (1728, 5120, 128, 128),
    (64, 2048, 128, 2048), (16 * 513, 256, 128, 128), (16 * 769, 256, 0, 128),
    (8192, 8192, 128, 128), (28672, 8192, 128, 128), (8192, 28672, 128, 128),      # Llama-2-70B shapes (not a BASELINE config)
])
def test_single_linear(n, k, r, g):
    from qeft_amd import _lib, qeft_cuda
    l, bufs = make(n, k, r, g, seed=n + k, bias=(n % 32 == 0))
    x = O.make_activation(1, k, r, seed=3)
    y = qeft_cuda.decode_linear(torch.from_numpy(x[0]).to(DEV), l)
    assert _lib.last_variant() == "gemv_v3"
    torch.cuda.synchronize()
    yref = ref(bufs, x, r, g)
    assert y.shape == (n,) and y.dtype == torch.float16
    assert rel_err(y.cpu().numpy()[None], yref) < REL_TOL
    assert elem_err_ok(y.cpu().numpy()[None], yref)


@pytest.mark.parametrize("ns,k", [((4096, 4096, 4096), 4096), ((512, 64, 64), 1024), ((5120, 5120, 5120), 5120),
                                   ((16, 32, 16), 256), ((4096, 1024, 1024), 4096), ((11008, 11008), 4096)])
def test_concatenated_linears_share_x(ns, k):
    """q|k|v as ONE operand (fuse.concat_linears) == the same linears one by one == oracle.  (Bit-equal only when both
    launches take the same waves-per-block variant: the split of a row's K steps over the waves is the summation order.)"""
    from qeft_amd import fuse, qeft_cuda
    r, g = 128, 128
    ls = [make(n, k, r, g, seed=10 * i + n) for i, n in enumerate(ns)]
    x = O.make_activation(1, k, r, seed=4)
    xt = torch.from_numpy(x[0]).to(DEV)
    y = qeft_cuda.decode_linear(xt, fuse.concat_linears([l for l, _ in ls]))
    torch.cuda.synchronize()
    assert y.numel() == sum(ns)
    o = 0
    for (l, bufs), n in zip(ls, ns):
        assert rel_err(y[o:o + n].cpu().numpy()[None], ref(bufs, x, r, g)) < REL_TOL
        one = qeft_cuda.decode_linear(xt, l)
        assert rel_err(y[o:o + n].cpu().numpy()[None], one.float().cpu().numpy()[None]) < 2e-3
        o += n


@pytest.mark.parametrize("n,k", [(16, 256), (48, 1024), (1376, 4096), (11008, 4096), (13824, 5120), (1728, 5120)])
@pytest.mark.parametrize("bias", [False, True])
def test_gate_up_pair_silu(n, k, bias):
    """PAIR mode: gate and up rows interleaved 8 + 8 per MFMA set (fuse.pair_interleave), silu(gate) * up in the epilogue
    == the unfused sequence (two linears, then silu * mul on their fp16 outputs) and within tolerance of float64."""
    from qeft_amd import _lib, fuse, qeft_cuda
    r, g = 128, 128
    (lg, bg), (lu, bu) = make(n, k, r, g, seed=n, bias=bias), make(n, k, r, g, seed=n + 1, bias=bias)
    x = O.make_activation(1, k, r, seed=5)
    xt = torch.from_numpy(x[0]).to(DEV)
    act = qeft_cuda.decode_linear(xt, fuse.pair_interleave(lg, lu), mode=qeft_cuda.V3_PAIR)
    assert _lib.last_variant() == "gemv_v3_pair" and act.numel() == n
    gate, up = qeft_cuda.decode_linear(xt, lg), qeft_cuda.decode_linear(xt, lu)
    torch.cuda.synchronize()
    gf, uf = gate.float(), up.float()
    unfused = (gf * torch.sigmoid(gf) * uf)
    assert (act.float() - unfused).abs().max().item() <= 2e-3 * unfused.abs().max().item()   # rcp-based silu: 1-2 ulp of fp16
    g64, u64 = ref(bg, x, r, g)[0], ref(bu, x, r, g)[0]
    want = g64 / (1 + np.exp(-g64)) * u64
    assert rel_err(act.cpu().numpy(), want) < 2e-3


@pytest.mark.parametrize("n,k", [(4096, 4096), (1024, 1024), (5120, 5120), (256, 256)])
def test_producer_side_rmsnorm_and_fp32_residual_chain(n, k):
    """o_proj-like launch emits (h32 += W x, fp16(h32 gamma), partial sums of h32^2); the next launch consumes them:
    together  y = W2 . rmsnorm(h) * gamma  within tolerance of the float64 chain."""
    from qeft_amd import qeft_cuda
    assert n == k
    r, g, eps = 128, 128, 1e-5
    (l1, b1), (l2, b2) = make(n, k, r, g, seed=1), make(n, k, r, g, seed=2)
    rng = np.random.default_rng(0)
    x = O.make_activation(1, k, r, seed=6)
    h0 = rng.standard_normal(n).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(n)).astype(np.float16)
    h32 = torch.from_numpy(h0).to(DEV)
    y32, hn, ssq = qeft_cuda.decode_linear(torch.from_numpy(x[0]).to(DEV), l1, residual=h32, out=h32,
                                           gamma_out=torch.from_numpy(gamma).to(DEV))
    torch.cuda.synchronize()
    assert y32.data_ptr() == h32.data_ptr() and y32.dtype == torch.float32
    h_ref = h0.astype(np.float64) + ref(b1, x, r, g)[0]
    assert rel_err(y32.cpu().numpy(), h_ref) < REL_TOL
    assert ssq.numel() == qeft_cuda.decode_linear_blocks(n)
    assert abs(ssq.sum().item() - float((y32.double() ** 2).sum())) <= 1e-4 * float((y32.double() ** 2).sum())
    assert torch.equal(hn, (y32 * torch.from_numpy(gamma).to(DEV).float()).half())
    # the stand-alone producer kernel gives the same (h gamma) and the same total
    h2, hn2, ssq2 = qeft_cuda.residual_norm(y32, None, torch.from_numpy(gamma).to(DEV))
    assert torch.equal(hn2, hn) and torch.equal(h2, y32)
    assert abs(ssq2.sum().item() - ssq.sum().item()) <= 1e-5 * ssq.sum().item()
    # consumer: deferred 1/rms
    y = qeft_cuda.decode_linear(hn, l2, ssq_in=ssq, eps=eps)
    # a residual launch without gamma_out: only the fp32 stream is written
    y_only = qeft_cuda.decode_linear(torch.from_numpy(x[0]).to(DEV), l1, residual=torch.from_numpy(h0).to(DEV))
    assert rel_err(y_only.cpu().numpy(), h_ref) < REL_TOL
    torch.cuda.synchronize()
    hv = y32.cpu().numpy().astype(np.float64)
    xn = hv / np.sqrt((hv ** 2).mean() + eps) * gamma.astype(np.float64)
    w2 = O.dequant_dense(b2["qweight"], b2["scales"], b2["scaled_zeros"], b2["oweight"], g).astype(np.float64)
    assert rel_err(y.cpu().numpy(), w2 @ xn) < 2e-3       # x gamma is rounded to fp16 before the product (one more rounding than the oracle)


def test_token_begin_norm_and_final_norm():
    from qeft_amd import _lib
    lib = _lib.lib()
    hidden, vocab, max_seq = 4096, 100, 64
    torch.manual_seed(0)
    embed = torch.randn(vocab, hidden, device=DEV).half()
    gamma = (1 + 0.1 * torch.randn(hidden, device=DEV)).half()
    tab = torch.randn(max_seq, 128, device=DEV)
    tok = torch.tensor([37], device=DEV)
    pos = torch.tensor([5], dtype=torch.int32, device=DEV)
    h = torch.empty(hidden, dtype=torch.float32, device=DEV)
    hn = torch.empty(hidden, dtype=torch.float16, device=DEV)
    row = torch.empty(128, dtype=torch.float32, device=DEV)
    nb = lib.qeft_token_begin_norm_blocks(hidden)
    ssq = torch.empty(nb, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.qeft_token_begin_norm(embed.data_ptr(), tok.data_ptr(), tab.data_ptr(), pos.data_ptr(), h.data_ptr(),
                                         row.data_ptr(), gamma.data_ptr(), hn.data_ptr(), ssq.data_ptr(), hidden, vocab,
                                         max_seq, st))
    torch.cuda.synchronize()
    assert torch.equal(h, embed[37].float()) and torch.equal(row, tab[5])
    assert torch.equal(hn, (embed[37].float() * gamma.float()).half())
    assert abs(ssq.sum().item() - float((embed[37].double() ** 2).sum())) < 1e-3
    y = torch.empty(1, hidden, dtype=torch.float16, device=DEV)
    _lib.check(lib.qeft_rmsnorm_f32(h.data_ptr(), gamma.data_ptr(), y.data_ptr(), 1, hidden, 1e-5, st))
    torch.cuda.synchronize()
    want = h * torch.rsqrt((h ** 2).mean() + 1e-5) * gamma.float()
    assert (y[0].float() - want).abs().max().item() <= 2e-3 * want.abs().max().item()


@pytest.mark.parametrize("hidden,vocab", [(4096, 32000), (512, 1000), (5120, 777), (1024, 8)])
def test_fused_final_norm_lm_head(hidden, vocab):
    """qeft_lm_head_f16 == rmsnorm_f32 followed by the fp16 matmul (fp32 accumulation), row counts that do not divide
    over the blocks / waves included."""
    from qeft_amd import _lib
    lib = _lib.lib()
    torch.manual_seed(hidden + vocab)
    h = torch.randn(hidden, device=DEV, dtype=torch.float32) * 3
    gamma = (1 + 0.1 * torch.randn(hidden, device=DEV)).half()
    w = (torch.randn(vocab, hidden, device=DEV) * 0.02).half()
    logits = torch.full((vocab,), float("nan"), dtype=torch.float16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.qeft_lm_head_f16(h.data_ptr(), gamma.data_ptr(), w.data_ptr(), logits.data_ptr(), hidden, vocab, 1e-5, st))
    hn = torch.empty(1, hidden, dtype=torch.float16, device=DEV)
    _lib.check(lib.qeft_rmsnorm_f32(h.data_ptr(), gamma.data_ptr(), hn.data_ptr(), 1, hidden, 1e-5, st))
    torch.cuda.synchronize()
    want = hn[0].double() @ w.double().t()
    assert torch.isfinite(logits).all()
    assert (logits.double() - want).abs().max().item() <= 2e-3 * want.abs().max().item()


@pytest.mark.parametrize("n,k,pair", [(4096, 4096, False), (22016 // 2, 4096, True), (512, 1024, False), (1728, 5120, True), (48, 640, False)])
def test_consumer_side_whole_rmsnorm(n, k, pair):
    """qeft_decode_linear_hnorm: the launch normalises the fp32 vector itself == residual_norm producer + deferred-1/rms
    consumer (same fp16(h * gamma) staging, another summation order of h^2) and the float64 chain."""
    from qeft_amd import fuse, qeft_cuda
    r, g, eps = 128, 128, 1e-5
    rng = np.random.default_rng(n + k)
    h = (rng.standard_normal(k) * 2).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    ht, gt = torch.from_numpy(h).to(DEV), torch.from_numpy(gamma).to(DEV)
    if pair:
        (lg, bg), (lu, bu) = make(n, k, r, g, seed=n), make(n, k, r, g, seed=n + 1)
        op, mode = fuse.pair_interleave(lg, lu), qeft_cuda.V3_PAIR
    else:
        l, b = make(n, k, r, g, seed=n)
        op, mode = fuse.single(l), qeft_cuda.V3_PLAIN
    y = qeft_cuda.decode_linear_hnorm(ht, gt, op, mode=mode, eps=eps)
    _, hn, ssq = qeft_cuda.residual_norm(ht, None, gt)
    y2 = qeft_cuda.decode_linear(hn, op, mode=mode, ssq_in=ssq, eps=eps)
    torch.cuda.synchronize()
    assert y.shape == y2.shape
    assert (y.float() - y2.float()).abs().max().item() <= 2e-3 * y2.float().abs().max().item()
    xn = h.astype(np.float64) / np.sqrt((h.astype(np.float64) ** 2).mean() + eps) * gamma.astype(np.float64)
    if pair:
        wg = O.dequant_dense(bg["qweight"], bg["scales"], bg["scaled_zeros"], bg["oweight"], g).astype(np.float64) @ xn
        wu = O.dequant_dense(bu["qweight"], bu["scales"], bu["scaled_zeros"], bu["oweight"], g).astype(np.float64) @ xn
        want = wg / (1 + np.exp(-wg)) * wu
    else:
        want = O.dequant_dense(b["qweight"], b["scales"], b["scaled_zeros"], b["oweight"], g).astype(np.float64) @ xn
    assert rel_err(y.cpu().numpy(), want) < 3e-3
