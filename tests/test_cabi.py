"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol that
include/qeft_hip.h declares; argument validation that needs no GPU; the product has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from qeft_amd import _lib, build
    build.build(verbose=False)
    return _lib.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qeft_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qeft_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/qeft_hip.h but not exported"
    from qeft_amd import _lib
    assert set(syms) == set(_lib.SIGNATURES), "ctypes table and header disagree"


def test_abi_version_and_error_strings(lib):
    assert lib.qeft_abi_version() == 1
    assert lib.qeft_error_string(1).decode() == "Unsupported batch size for gemv kernel."
    assert lib.qeft_error_string(0).decode() == "ok"


def test_argument_validation_without_gpu(lib):
    """Validation happens before any HIP call, so these run on a GPU-less host."""
    buf = ctypes.create_string_buffer(4096 + 16)
    p = (ctypes.addressof(buf) + 15) & ~15
    g = lib.qeft_gemv_w4
    assert g(p, p, p, p, p, 8, 64, 512, 128, None) == 1       # batch
    assert g(p, p, p, p, p, 0, 64, 512, 128, None) == 1
    assert g(p, p, p, p, p, 1, 62, 512, 128, None) == 2       # N % 4
    assert g(p, p, p, p, p, 1, 64, 500, 128, None) == 2       # K % 64
    assert g(p, p, p, p, p, 1, 64, 512, 48, None) == 3        # group
    assert g(None, p, p, p, p, 1, 64, 512, 128, None) == 4    # null
    assert g(p + 2, p, p, p, p, 1, 64, 512, 128, None) == 6   # alignment
    q = lib.qeft_gemv_w4_qeft
    assert q(p, p, p, p, None, p, 1, 64, 512, 128, 128, None) == 4
    assert q(p, p, p, p, p, p, 1, 64, 512, 128, 100, None) == 2
    assert q(p, p, p, p, p, p, 1, 60, 512, 128, 128, None) == 2   # outliers need N % 8
    assert lib.qeft_gemm_w4(p, p, p, p, None, None, p, 0, 64, 512, 128, 0, None) == 2


def test_no_cpu_fallback():
    """A CPU tensor must raise, never silently compute on the host."""
    from qeft_amd import qeft_cuda
    x = torch.zeros(1, 128, dtype=torch.float16)
    qw = torch.zeros(2, 128, dtype=torch.int16)
    s = torch.zeros(1, 8, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        qeft_cuda.gemv_4bit(x, qw, s, s, 1, 8, 128, 128)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        qeft_cuda.gemm_4bit(x, qw, s, s)


def test_missing_library_fails_loudly(monkeypatch):
    from qeft_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libqeft_hip.so")
    with pytest.raises(_lib.QeftHipError, match="no CPU fallback"):
        _lib.lib()


def test_product_does_not_import_oracle():
    import glob
    for path in glob.glob(os.path.join(ROOT, "qeft_amd", "**", "*.py"), recursive=True) + \
            [os.path.join(ROOT, "qeft_cuda.py")]:
        src = open(path).read()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f"{path} references the oracle"


def test_binding_puts_torch_hip_runtime_first():
    """The library links the system libamdhip64, torch ships its own: torch has to be loaded before the library or the
    process ends up with two HIP runtimes (launches then fail with hipErrorNoDevice, seen as build() + smoke() in one
    process).  The binding module therefore imports torch itself."""
    import subprocess
    import sys
    code = "import sys; from qeft_amd import _lib; assert 'torch' in sys.modules; print('ok')"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_shim_exports_the_reference_module_surface():
    """Every function the reference binds in qeft_cuda.cpp:10-27 exists in the module a reference caller imports, with
    the reference's positional parameters (the two FT entries keep their pybind defaults, qeft_cuda.cpp:23-26)."""
    import inspect
    import qeft_cuda
    want = {
        "gemm_4bit": ["in_feats", "kernel", "scales", "zeros"],
        "gemv_4bit": ["in_feats", "kernel", "scaling_factors", "zeros", "m", "n", "k", "group_size"],
        "gemv_4bit_qeft": ["in_feats", "kernel", "scaling_factors", "zeros", "oweight", "m", "n", "k", "group_size"],
        "layernorm_forward_cuda": ["x", "gamma", "out", "eps"],
        "single_query_attention": ["q", "k", "v", "k_cache", "v_cache", "length_per_sample_", "alibi_slopes_", "timestep",
                                   "rotary_embedding_dim", "rotary_base", "neox_rotary_style"],
    }
    for name, params in want.items():
        sig = inspect.signature(getattr(qeft_cuda, name))
        assert list(sig.parameters) == params, name
    d = inspect.signature(qeft_cuda.single_query_attention).parameters
    assert d["rotary_embedding_dim"].default == 0 and d["rotary_base"].default == 10000.0 and d["neox_rotary_style"].default is True
    x = torch.zeros(1, 2, 128, dtype=torch.float16)
    with pytest.raises(RuntimeError):        # CPU tensors raise: no host fallback for the FT entries either
        qeft_cuda.layernorm_forward_cuda(x, torch.ones(128, dtype=torch.float16), torch.empty_like(x), 1e-5)


def test_round2_entries_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any HIP call: NULL operands, misaligned pointers and shapes the kernels do not
    take come back as error codes (no launch, no GPU needed)."""
    from qeft_amd import _lib
    lib = _lib.lib()
    ok_ptr = 1 << 20            # a non-NULL, 16-byte aligned address: validation must fail before anything dereferences it
    # decode GEMVs: K not a multiple of 128 / n not a multiple of 16 / an outlier width the kernel does not take
    for entry in (lib.qeft_decode_linear, lib.qeft_decode_linear_w3):
        assert entry(ok_ptr, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4000, 128, 128, 0, None, None, 0, 0.0, None, None, None, None) != 0
        assert entry(ok_ptr, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4090, 4096, 128, 128, 0, None, None, 0, 0.0, None, None, None, None) != 0
        assert entry(ok_ptr, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4096, 128, 64, 0, None, None, 0, 0.0, None, None, None, None) != 0
        assert entry(None, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4096, 128, 128, 0, None, None, 0, 0.0, None, None, None, None) != 0
        assert entry(ok_ptr + 2, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4096, 128, 128, 0, None, None, 0, 0.0, None, None, None, None) != 0
        # gamma_out without a residual, and more partial sums than a consumer accepts
        assert entry(ok_ptr, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4096, 128, 128, 0, None, None, 0, 0.0, ok_ptr, ok_ptr, ok_ptr, None) != 0
        assert entry(ok_ptr, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4096, 128, 128, 0, None, ok_ptr, 513, 1e-5, None, None, None, None) != 0
    assert lib.qeft_decode_linear_hnorm(ok_ptr, None, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4096, 128, 128, 0, 1e-5, None) != 0
    assert lib.qeft_decode_linear_hnorm(ok_ptr, ok_ptr, ok_ptr, ok_ptr, ok_ptr, None, ok_ptr, 4096, 4100, 128, 128, 0, 1e-5, None) != 0
    assert lib.qeft_lm_head_f16(ok_ptr, ok_ptr, ok_ptr, ok_ptr, 4000, 32000, 1e-5, None) != 0      # hidden not a multiple of 512
    assert lib.qeft_lm_head_f16(ok_ptr, ok_ptr, None, ok_ptr, 4096, 32000, 1e-5, None) != 0
    assert lib.qeft_rope_rows(None, ok_ptr, ok_ptr, 4, 2, 256, None) != 0
    assert lib.qeft_rope_rows(ok_ptr, ok_ptr, ok_ptr, 0, 2, 256, None) != 0
    assert lib.qeft_rope_rows(ok_ptr, ok_ptr, ok_ptr, 4, 2, 128, None) != 0      # rows narrower than their heads
    assert lib.qeft_decode_linear_blocks(4096) == 256 and lib.qeft_decode_linear_blocks(22016) == 459
    assert lib.qeft_decode_linear_blocks(5120) == 160          # between 256 and 512 row sets: two per block


def test_gemm_workspace_follows_the_row_routing(lib):
    """No compute: the split-K workspace the shim asks for matches where gemm_impl will send the rows -- up to 16 rows of a shape
    the decode GEMV serves ride on it (no workspace), 17 .. 64 rows take the weight-stationary tier (gemm_ws.hip, one launch, no
    workspace), from 65 rows on the split-K tier wants its partial-sum buffer, enough tiles need none."""
    f = lib.qeft_gemm_w4_workspace_bytes
    for (n, k) in ((4096, 4096), (11008, 4096), (4096, 11008), (13824, 5120)):
        for m in (8, 12, 16):
            assert f(m, n, k, 128) == 0, (m, n, k)                # gemv_v3_mb / gemv_v3_mb_xg
        for m in (17, 24, 64):
            assert f(m, n, k, 128) == 0, (m, n, k)                # gemm_ws
        assert f(96, n, k, 128) >= 2 * 96 * n * 4, (n, k)           # split-K partial sums
    assert f(24, 4096, 4160, 128) >= 2 * 24 * 4096 * 4              # K % 128 != 0: not a shape the weight-stationary tier takes
    assert f(4096, 4096, 4096, 128) == 0
    assert f(16, 4096, 4160, 128) == 0                              # K % 128 != 0: the round-1 small-M route, no workspace either
