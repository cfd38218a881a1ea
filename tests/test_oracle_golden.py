"""Pin the CPU oracle against golden vectors produced by the reference's own packer
(tests/golden/make_golden.py).  Integer/byte work: bit-exact."""
import numpy as np
import pytest

from conftest import golden_cases
from oracle import qeft_oracle as O

CASES = golden_cases()


def load(path):
    d = np.load(path)
    n, k, r, g, sym, bias = [int(v) for v in d["case"]]
    return d, n, k, r, g, bool(sym), bool(bias)


def test_have_fixtures():
    assert len(CASES) >= 8


@pytest.mark.parametrize("path", CASES)
def test_pack_intweight_bit_exact(path):
    d, *_ = load(path)
    got = O.pack_intweight(d["qraw"])
    assert got.dtype == np.int16
    assert np.array_equal(got, d["qraw_packed"])
    assert np.array_equal(O.unpack_intweight(d["qraw_packed"]), d["qraw"])


@pytest.mark.parametrize("path", CASES)
def test_nibble_position_closed_form(path):
    d, n, k, *_ = load(path)
    qw = d["qraw_packed"].view(np.uint16)
    rng = np.random.default_rng(0)
    for _ in range(200):
        i, j = int(rng.integers(n)), int(rng.integers(k))
        row, col, nib = O.nibble_position(i, j)
        assert (int(qw[row, col]) >> (4 * nib)) & 0xF == int(d["qraw"][i, j])


@pytest.mark.parametrize("path", CASES)
def test_pack_oweight_bit_exact(path):
    d, n, k, r, *_ = load(path)
    if r == 0:
        pytest.skip("no outlier slice")
    got = O.pack_oweight(d["ow_rand"])
    assert np.array_equal(got.view(np.uint16), d["ow_rand_packed"].view(np.uint16))
    assert np.array_equal(O.unpack_oweight(d["ow_rand_packed"]).view(np.uint16), d["ow_rand"].view(np.uint16))


@pytest.mark.parametrize("path", CASES)
def test_minmax_and_fakequant_match_reference_quantizer(path):
    d, n, k, r, g, sym, _ = load(path)
    scale, zero = O.minmax_params(d["w_orig"], g, sym=bool(sym))      # both branches of quant.py:142-158
    assert np.array_equal(scale, d["scale"])
    assert np.array_equal(zero, d["zero"])
    wq = O.fake_quantize(d["w_orig"], scale, zero, g, sym=bool(sym)).astype(np.float16)
    if r:
        wq[:, k - r:] = d["w_orig"][:, k - r:]
    assert np.array_equal(wq.view(np.uint16), d["w_fake"].view(np.uint16))


@pytest.mark.parametrize("path", CASES)
def test_pack_layer_matches_reference_state_dict(path):
    d, n, k, r, g, sym, bias = load(path)
    out = O.pack_layer(d["w_fake"], d["scale"], d["zero"], r, g, sym=sym)
    assert np.array_equal(out["qweight"], d["sd_qweight"])
    assert np.array_equal(out["scales"].view(np.uint16), d["sd_scales"].view(np.uint16))
    assert np.array_equal(out["scaled_zeros"].view(np.uint16), d["sd_scaled_zeros"].view(np.uint16))
    if r:
        assert np.array_equal(out["oweight"].view(np.uint16), d["sd_oweight"].view(np.uint16))
        assert np.array_equal(out["oweight_interleaved"].view(np.uint16),
                              d["sd_oweight_interleaved"].view(np.uint16))
        assert np.array_equal(O.sparse_to_dense_ids(d["outlieridx"], k), d["reorder_ids"])


@pytest.mark.parametrize("path", CASES)
def test_dequant_recovers_fake_quant_weight(path):
    """Dequantised checkpoint == the fake-quant nn.Linear weight the reference evaluates
    (recon.py:573) up to fp16 rounding of scale / scaled-zero (~5e-5 abs)."""
    d, n, k, r, g, sym, _ = load(path)
    ow = d["sd_oweight"] if r else None
    w = O.dequant_dense(d["sd_qweight"], d["sd_scales"], d["sd_scaled_zeros"], ow, g)
    ref = d["w_fake"].astype(np.float32)
    assert np.abs(w - ref).max() <= 2e-4
    if r:
        assert np.array_equal(w[:, k - r:], ref[:, k - r:])
        # dead nibbles under the outlier columns dequantise to ~0 (qlinear.py:200-202)
        dead = O.dequant_dense(d["sd_qweight"], d["sd_scales"], d["sd_scaled_zeros"], None, g)[:, k - r:]
        assert np.abs(dead).max() <= 1e-4


@pytest.mark.parametrize("path", CASES)
def test_linear_vs_fake_quant_linear(path):
    d, n, k, r, g, sym, bias = load(path)
    ow = d["sd_oweight"] if r else None
    b = d["sd_bias"] if bias else None
    x = O.make_activation(3, k, r, seed=1)
    y = O.quant_linear(x, d["sd_qweight"], d["sd_scales"], d["sd_scaled_zeros"], ow, b, g).astype(np.float64)
    yref = x.astype(np.float64) @ d["w_fake"].astype(np.float64).T
    if bias:
        yref = yref + b.astype(np.float64)
    rel = np.abs(y - yref).max() / max(np.abs(yref).max(), 1e-6)
    assert rel < 5e-3
