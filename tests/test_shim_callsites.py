"""The drop-in boundary pinned to the reference's own call sites, mechanically.

`tests/golden/qeft_cuda_callsites.json` lists every `qeft_cuda.<fn>(...)` call the reference's Python makes (qlinear.py:20,38,51,
66,253-263,265,285-295,297,314-323,325; monkeypatch/ftllama_modeling.py:42,134) and every `m.def` of qeft_cuda.cpp:18-26, as
written by tests/golden/make_callsites.py (an `ast` / regex pass over the reference's text; names and counts only).  The shim a
reference user imports (`qeft_cuda` at the repo root) must accept each of those calls exactly as made.  Where /root/reference
exists (the build container) the fixture itself is re-derived and must be current."""
import inspect
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)

with open(os.path.join(GOLD, "qeft_cuda_callsites.json")) as _fh:
    SITES = json.load(_fh)


def _shim():
    import qeft_cuda
    return qeft_cuda


def _binds(fn, n_positional, keywords):
    """True when a call with that many positional arguments and those keywords is accepted by fn's signature."""
    sig = inspect.signature(fn)
    try:
        sig.bind(*([None] * n_positional), **{k: None for k in keywords})
        return True
    except TypeError:
        return False


def test_every_bound_function_exists_with_the_pybind_argument_names():
    shim = _shim()
    assert set(SITES["bound"]) == {"gemm_4bit", "gemv_4bit", "gemv_4bit_qeft", "layernorm_forward_cuda", "single_query_attention"}
    for name, args in SITES["bound"].items():
        assert hasattr(shim, name), f"qeft_cuda.{name} (qeft_cuda.cpp m.def) is missing from the shim"
        params = inspect.signature(getattr(shim, name)).parameters
        if not args:
            continue
        # py::arg names are part of the interface (callable by keyword), and so is which of them has a default
        assert [a["name"] for a in args] == list(params), name
        for a in args:
            has_default = params[a["name"]].default is not inspect.Parameter.empty
            assert has_default == a["has_default"], (name, a["name"])


def test_every_direct_call_site_binds():
    shim = _shim()
    assert SITES["calls"], "no call sites in the fixture"
    for c in SITES["calls"]:
        fn = getattr(shim, c["fn"])
        assert _binds(fn, c["n_positional"], c["keywords"]), f"{c['file']}:{c['line']} qeft_cuda.{c['fn']} with {c['n_positional']} positional"
        # no silently optional extras: one argument fewer must NOT bind (the reference passes every operand positionally)
        if not c["keywords"] and c["fn"] != "single_query_attention":
            assert not _binds(fn, c["n_positional"] - 1, []), f"{c['fn']} accepts fewer operands than {c['file']}:{c['line']} passes"


def test_calls_through_self_gemv_and_self_gemm_bind_to_their_targets():
    """qlinear.py:223-234 stores the entry points in self.gemv / self.gemm; each later call through the alias must be accepted by
    at least one function the alias can hold, and every function it can hold must be reached by some call."""
    shim = _shim()
    for alias, rec in SITES["aliases"].items():
        targets = sorted({t["fn"] for t in rec["targets"]})
        assert targets and rec["calls"], alias
        reached = set()
        for c in rec["calls"]:
            ok = [t for t in targets if _binds(getattr(shim, t), c["n_positional"], c["keywords"])
                  and not _binds(getattr(shim, t), c["n_positional"] - 1, c["keywords"])]
            assert ok, f"{c['file']}:{c['line']} self.{alias}(...) with {c['n_positional']} positional fits none of {targets}"
            reached.update(ok)
        assert reached == set(targets), (alias, reached, targets)


def test_gemv_entries_keep_the_reference_error_for_unsupported_batch():
    """gemv_cuda_qeft.cu:466 / gemv_cuda.cu: m outside 1..7 raises RuntimeError("Unsupported batch size for gemv kernel.")."""
    import torch
    shim = _shim()
    x = torch.zeros(8, 128, dtype=torch.float16)
    qw = torch.zeros(2, 128, dtype=torch.int16)
    s = torch.zeros(1, 8, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="Unsupported batch size"):
        shim.gemv_4bit(x, qw, s, s, 8, 8, 128, 128)
    with pytest.raises(RuntimeError, match="Unsupported batch size"):
        shim.gemv_4bit_qeft(x, qw, s, s, torch.zeros(4, 256, dtype=torch.float16), 0, 8, 128, 128)


@pytest.mark.skipif(not os.path.isdir("/root/reference/qeft"), reason="the reference tree only exists in the build container")
def test_fixture_is_current_with_the_reference_tree():
    import make_callsites
    assert make_callsites.collect() == SITES, "tests/golden/qeft_cuda_callsites.json is stale: rerun tests/golden/make_callsites.py"
