"""Guard for the fault class of the two round-1 development aborts: UNCONDITIONAL prefetch / staging
loads whose address was not clamped into the operand -- a wave without ring steps re-reading "step 0" of a row group
shorter than one wave-wide load (K < one step), and a ring prefetch running past a wave's last step.  Such reads are
never consumed, so no numeric check sees them; they fault only when the operand ends where its mapping ends.

The cases below run in a child process with PYTORCH_NO_CUDA_MEMORY_CACHING=1 (every tensor its own hipMalloc, so an
operand ends at the end of its allocation instead of somewhere inside a 2 MiB pool segment), on the shapes where the
clamps bite: one 16-row set, K below / at / just above one step and one staging pass, last blocks with fewer row sets
than rs_cap, batch slices, ragged GEMM tiles, and (round 2) the 256-row GEMM / dX tiles with their loader waves, the
round-2 decode launches on their smallest shapes, the fused token tail.  Every result is also checked against the oracle, so a clamp that
"fixes" a fault by reading the wrong bytes fails too.  Runs last (file name) and once."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np, torch
torch.manual_seed(20261004)
from oracle import qeft_oracle as O
from util import layer_to_torch, rel_err
from qeft_amd import qeft_cuda, _lib
DEV = "cuda:0"

def gemv(n, k, r, g, m, gather=False):
    b = O.make_layer(n, k, r, g, seed=n + k + m)
    t = layer_to_torch(b, DEV)
    x = O.make_activation(m, k, r, seed=m)
    ids = None
    if gather:
        ids = O.sparse_to_dense_ids(np.sort(np.random.default_rng(1).choice(k, size=max(r, 1), replace=False)), k)
    y = qeft_cuda.gemv_4bit_fused(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                  t.get("oweight_interleaved") if r else None, None,
                                  torch.from_numpy(ids.astype(np.int32)).to(DEV) if ids is not None else None, None, m, n, k, g)
    torch.cuda.synchronize()
    ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, None, g, reorder_ids=ids)
    assert rel_err(y.cpu().numpy(), ref.astype(np.float64)) < 1e-3, ("gemv", n, k, r, g, m)

def gemm(n, k, r, g, m):
    b = O.make_layer(n, k, r, g, seed=n + k + m)
    t = layer_to_torch(b, DEV)
    x = O.make_activation(m, k, r, seed=m)
    dy = (np.random.default_rng(m).standard_normal((m, n)) * 0.1).astype(np.float16)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None)
    dx = qeft_cuda.gemm_4bit_dx(torch.from_numpy(dy).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None)
    torch.cuda.synchronize()
    ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, None, g)
    assert rel_err(y.cpu().numpy(), ref.astype(np.float64)) < 1e-3, ("gemm", n, k, r, g, m)
    dx_ref, dow_ref = O.quant_linear_backward(dy, x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, g)
    assert rel_err(dx.cpu().numpy(), dx_ref.astype(np.float64)) < 2e-3, ("dx", n, k, r, g, m)
    if r:
        dow = qeft_cuda.grad_oweight(torch.from_numpy(dy).to(DEV), torch.from_numpy(x).to(DEV), r)
        torch.cuda.synchronize()
        assert rel_err(dow.cpu().numpy(), dow_ref) < 1e-3, ("dow", n, k, r, g, m)

# decode GEMV: one row set; K below one 128-k step of INT4 work (only the fp16 slice), exactly one, many; VALU-path shapes
for (n, k, r, g) in [(16, 128, 0, 128), (16, 256, 128, 128), (16, 384, 128, 128), (8, 128, 64, 64), (8, 64, 0, 32),
                     (24, 384, 0, 128), (16, 2048, 128, 2048), (16, 4096, 128, 128), (16, 4224, 128, 128), (32, 8192 + 128, 128, 128)]:
    for m in (1, 2, 7):
        gemv(n, k, r, g, m)
gemv(16, 4224, 128, 128, 1, gather=True)
# row sets that do not divide over the blocks (rs_cap = 3, some blocks own 2) and the grouped / tail-block arithmetic
for n in (16 * 513, 16 * 769, 16 * 1025):
    gemv(n, 256, 128, 128, 1)
# batch slices: 7 rows of K = 11008 do not fit the staging area in one pass
gemv(64, 11008, 128, 128, 7)
# GEMM / dX / d(oweight): ragged M and N tiles, the minimum K of the DMA ring, the small-M route, split-K
for (n, k, r, g, m) in [(136, 192, 0, 64, 129), (8, 64, 0, 64, 1), (264, 256, 128, 128, 17), (128, 1024, 128, 128, 130),
                        (520, 2048, 64, 128, 257), (256, 4096, 128, 128, 200)]:
    gemm(n, k, r, g, m)
# ---- round 2: the 256-row GEMM / dX tiles (loader waves: DMA pieces, the register ring of packed weights, the outlier k-tile's
# direct DMA), ragged M / N, every remainder class of the dX loader's ring
import types
from qeft_amd import fuse
for (n, k, r, g, m) in [(5896, 512, 64, 64, 1100), (5888, 384, 0, 128, 1025)]:
    b = O.make_layer(n, k, r, g, seed=n + m)
    t = layer_to_torch(b, DEV)
    x = O.make_activation(m, k, r, seed=m)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None)
    assert _lib.last_variant() == "gemm_v3_256x128", _lib.last_variant()
    torch.cuda.synchronize()
    ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, None, g)
    assert rel_err(y.cpu().numpy(), ref.astype(np.float64)) < 1e-3, ("gemm v3", n, k, r, g, m)
for (n, k, r, g, m) in [(256, 6144, 128, 128, 1100), (320, 6144, 0, 128, 1025), (448, 6144, 128, 128, 1100)]:
    b = O.make_layer(n, k, r, g, seed=n + m)
    t = layer_to_torch(b, DEV)
    dy = (np.random.default_rng(n).standard_normal((m, n)) * 0.1).astype(np.float16)
    dx = qeft_cuda.gemm_4bit_dx(torch.from_numpy(dy).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None)
    assert _lib.last_variant() == "dx256", _lib.last_variant()
    torch.cuda.synchronize()
    w = O.dequant_dense(b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, g)
    assert rel_err(dx.cpu().numpy(), dy.astype(np.float64) @ w.astype(np.float64)) < 2e-3, ("dx256", n, k, r, g, m)
# ---- round 2 decode launches on their smallest shapes: plain / pair / consumer-side norm / 3-bit stream, fused token tail, rotary rows
def operand(n, k, r, bits=4, seed=0):
    b = O.make_layer(n, k, r, 128, seed=seed, bits=bits)
    t = layer_to_torch(b, DEV)
    l = types.SimpleNamespace(qweight=t["qweight"], scales=t["scales"], scaled_zeros=t["scaled_zeros"], oweight=t.get("oweight"), bias=None,
                              outfeatures=n, infeatures=k, group_size=128, outlierfeatures=r, bits=bits)
    l.sz_packed = qeft_cuda.pack_scales(l.scales, l.scaled_zeros, n, k, 128)
    return l, b
for (n, k, r) in [(16, 128, 0), (16, 256, 128), (48, 640, 128), (16 * 257, 256, 128)]:
    for bits in (4, 3):
        l, b = operand(n, k, r, bits, seed=n + k + bits)
        x = O.make_activation(1, k, r, seed=3)
        y = qeft_cuda.decode_linear(torch.from_numpy(x[0]).to(DEV), l)
        torch.cuda.synchronize()
        ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, None, 128)
        assert rel_err(y.cpu().numpy()[None], ref.astype(np.float64)) < 1e-3, ("v3", n, k, r, bits)
    lg, bg = operand(n, k, r, 4, seed=n); lu, bu = operand(n, k, r, 4, seed=n + 1)
    h = (np.random.default_rng(n).standard_normal(k) * 2).astype(np.float32)
    gam = np.ones(k, dtype=np.float16)
    act = qeft_cuda.decode_linear_hnorm(torch.from_numpy(h).to(DEV), torch.from_numpy(gam).to(DEV), fuse.pair_interleave(lg, lu), mode=qeft_cuda.V3_PAIR)
    torch.cuda.synchronize()
    xn = h.astype(np.float64) / np.sqrt((h.astype(np.float64) ** 2).mean() + 1e-5)
    wg = O.dequant_dense(bg["qweight"], bg["scales"], bg["scaled_zeros"], bg.get("oweight") if r else None, 128).astype(np.float64) @ xn
    wu = O.dequant_dense(bu["qweight"], bu["scales"], bu["scaled_zeros"], bu.get("oweight") if r else None, 128).astype(np.float64) @ xn
    assert rel_err(act.cpu().numpy(), wg / (1 + np.exp(-wg)) * wu) < 4e-3, ("hnorm pair", n, k, r)
# ---- round 3: the 128 x 128 GEMM tier on ragged tiles (M, N not multiples of 128, the minimum K of the rings); the reference's gemv
# entries on the v3 kernel with the operands as the checkpoint holds them (raw scale rows, interleaved outlier slab, the
# reorder_ids gather, several batch rows), on the shapes where their clamps bite
for (n, k, r, g, m) in [(7176, 384, 64, 64, 129), (3592, 512, 128, 128, 530)]:
    b = O.make_layer(n, k, r, g, seed=n + m)
    t = layer_to_torch(b, DEV)
    x = O.make_activation(m, k, r, seed=m)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None)
    assert _lib.last_variant() == "gemm_v3_128x128", _lib.last_variant()
    torch.cuda.synchronize()
    ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, None, g)
    assert rel_err(y.cpu().numpy(), ref.astype(np.float64)) < 1e-3, ("gemm v3 128", n, k, r, g, m)
for (n, k, r, g, m) in [(256, 3072, 128, 128, 530), (320, 3072, 0, 128, 513)]:       # the 128-row dX tile: 4 / 5 n-tiles, ragged M
    b = O.make_layer(n, k, r, g, seed=n + m)
    t = layer_to_torch(b, DEV)
    dy = (np.random.default_rng(n).standard_normal((m, n)) * 0.1).astype(np.float16)
    dx = qeft_cuda.gemm_4bit_dx(torch.from_numpy(dy).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None)
    assert _lib.last_variant() == "dx128v3", _lib.last_variant()
    torch.cuda.synchronize()
    w = O.dequant_dense(b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, g)
    assert rel_err(dx.cpu().numpy(), dy.astype(np.float64) @ w.astype(np.float64)) < 2e-3, ("dx128v3", n, k, r, g, m)
for (n, k, r, g, m, gather) in [(16, 256, 128, 128, 1, True), (16, 256, 128, 128, 7, True), (48, 384, 0, 128, 5, False),
                                (16 * 257, 256, 128, 128, 3, False), (32, 4096, 128, 4096, 2, True), (16, 11008, 128, 128, 7, False)]:
    gemv(n, k, r, g, m, gather=gather)
    assert _lib.last_variant().startswith("gemv_v3"), (_lib.last_variant(), n, k, r, g, m)
# ---- round 3: 8..16 rows at the GEMM entries ride on the decode GEMV -- x rows staged in LDS (gemv_v3_mb) or read by the lanes from
#      global memory (gemv_v3_mb_xg: rows * K * 2 bytes exceed the LDS); one row set, a last block with fewer sets, per-channel scales
for (n, k, r, g, m, want) in [(16, 256, 128, 128, 16, "gemv_v3_mb"), (48, 384, 0, 128, 9, "gemv_v3_mb"), (16 * 257, 256, 128, 128, 12, "gemv_v3_mb"),
                              (32, 11008, 128, 128, 16, "gemv_v3_mb_xg"), (16, 13824, 0, 128, 8, "gemv_v3_mb_xg"),
                              (64, 4096, 128, 4096, 16, "gemv_v3_mb"), (16 * 9, 11008, 128, 11008, 11, "gemv_v3_mb_xg")]:
    b = O.make_layer(n, k, r, g, seed=n + k + m)
    t = layer_to_torch(b, DEV)
    x = O.make_activation(m, k, r, seed=m)
    y = qeft_cuda.gemm_4bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None)
    assert _lib.last_variant() == want, (_lib.last_variant(), n, k, r, g, m)
    torch.cuda.synchronize()
    ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, None, g)
    assert rel_err(y.cpu().numpy(), ref.astype(np.float64)) < 1e-3, ("rows16", n, k, r, g, m)
lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
hh = torch.randn(512, device=DEV); gg = torch.ones(512, device=DEV).half(); ww = (torch.randn(7, 512, device=DEV) * 0.05).half()
lo = torch.empty(7, dtype=torch.float16, device=DEV)
_lib.check(lib.qeft_lm_head_f16(hh.data_ptr(), gg.data_ptr(), ww.data_ptr(), lo.data_ptr(), 512, 7, 1e-5, st))
rr = torch.randn(3, 1, 128, device=DEV).half(); cc = torch.randn(3, 64, device=DEV); ss = torch.randn(3, 64, device=DEV)
want = torch.cat([rr[..., :64].float() * cc[:, None] - rr[..., 64:].float() * ss[:, None], rr[..., 64:].float() * cc[:, None] + rr[..., :64].float() * ss[:, None]], -1).half()
_lib.check(lib.qeft_rope_rows(rr.data_ptr(), cc.data_ptr(), ss.data_ptr(), 3, 1, 128, st))
torch.cuda.synchronize()
hn = (hh * torch.rsqrt((hh ** 2).mean() + 1e-5)).half()
assert (lo.float() - hn.float() @ ww.float().t()).abs().max().item() < 2e-2
# (the kernel's a * c - b * s may be one fused multiply-add where torch rounds two products: equal up to one fp16 ulp on random data)
assert (rr.float() - want.float()).abs().max().item() <= 2e-3 * max(1.0, want.float().abs().max().item())
print("GUARD-OK")
'''


def test_exact_size_allocations_no_fault_and_correct():
    env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "GUARD-OK" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
