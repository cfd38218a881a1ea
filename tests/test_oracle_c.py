"""The plain-C restatement agrees with the numpy oracle and with the reference-generated golden vectors."""
import ctypes

import numpy as np
import pytest

from conftest import golden_cases
from oracle import build_c
from oracle import qeft_oracle as O

CASES = golden_cases()


@pytest.fixture(scope="module")
def lib():
    return build_c.load()


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("path", CASES)
def test_c_packers_match_golden(lib, path):
    d = np.load(path)
    n, k, r = [int(v) for v in d["case"][:3]]
    q = np.ascontiguousarray(d["qraw"])
    out = np.zeros((n // 4, k), np.uint16)
    lib.qeft_oracle_pack_intweight(ptr(q), ptr(out), n, k)
    assert np.array_equal(out.view(np.int16), d["qraw_packed"])
    if r:
        ow = np.ascontiguousarray(d["ow_rand"]).view(np.uint16)
        il = np.zeros((n // 2, 2 * r), np.uint16)
        lib.qeft_oracle_pack_oweight(ptr(ow), ptr(il), n, r)
        assert np.array_equal(il, d["ow_rand_packed"].view(np.uint16))


@pytest.mark.parametrize("path", CASES)
@pytest.mark.parametrize("rounded", [0, 1])
def test_c_dequant_and_linear_match_numpy_oracle(lib, path, rounded):
    d = np.load(path)
    n, k, r, g, sym, bias = [int(v) for v in d["case"]]
    qw = np.ascontiguousarray(d["sd_qweight"]).view(np.uint16)
    sc = np.ascontiguousarray(d["sd_scales"]).view(np.uint16)
    sz = np.ascontiguousarray(d["sd_scaled_zeros"]).view(np.uint16)
    ow = np.ascontiguousarray(d["sd_oweight"]).view(np.uint16) if r else None
    w = np.zeros((n, k), np.float32)
    lib.qeft_oracle_dequant(ptr(qw), ptr(sc), ptr(sz), ptr(ow) if r else None, ptr(w), n, k, g, r, rounded)
    wref = O.dequant_dense(d["sd_qweight"], d["sd_scales"], d["sd_scaled_zeros"], d["sd_oweight"] if r else None, g,
                           round_fp16=bool(rounded))
    assert np.array_equal(w, wref)
    x = O.make_activation(2, k, r, seed=3)
    b = np.ascontiguousarray(d["sd_bias"]).view(np.uint16) if bias else None
    y = np.zeros((2, n), np.uint16)
    lib.qeft_oracle_linear(ptr(np.ascontiguousarray(x).view(np.uint16)), ptr(w), ptr(b) if bias else None, ptr(y), 2, n, k)
    yref = O.linear(x, wref, d["sd_bias"] if bias else None)
    diff = np.abs(y.view(np.float16).astype(np.float64) - yref.astype(np.float64))
    assert diff.max() <= 2 * np.spacing(np.abs(yref).max().astype(np.float16)).astype(np.float64)
