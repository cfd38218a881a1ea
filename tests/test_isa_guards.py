"""ISA-level guards (CPU only: hipcc cross-compiles to gfx950 assembly, nothing runs).

1. gemm_w4_dx_kernel_v3 keeps in-flight global loads in accumulation registers a0..a29 NAMED in its asm text (the loader
   waves' weight ring, D3_RING_SET in csrc/gemm_w4.hip).  The clobber lists tell hipcc those registers are overwritten, not
   that they stay live between the load and the read -- so nothing in the COMPILER's part of the kernel may touch an AGPR:
   a spill, a copy or a v_accvgpr_* there would be silent corruption that the numeric tests only catch on this exact build.
   (cdna_hip_programming.md 5.7 item 4: "audit after every edit".)
2. No kernel of the decode GEMV (csrc/gemv_v3.h) spills: a scratch access would sit in the vmcnt queue of the weight ring.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qeft_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-S",
         "--cuda-device-only"]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(tmp_path_factory, src):
    out = tmp_path_factory.mktemp("isa") / (src + ".s")
    subprocess.run([HIPCC, *FLAGS, "-o", str(out), os.path.join(CSRC, src)], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel_body(text, mangled_prefix):
    m = re.search(r"^(%s\w*):" % re.escape(mangled_prefix), text, re.M)
    assert m, mangled_prefix
    start = m.end()
    end = text.index(".Lfunc_end", start)
    return m.group(1), text[start:end]


def _metadata(text, name):
    i = text.index(".name:           " + name)
    blk = text[i:i + 1200]
    return {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")
            if re.search(r"\.%s:\s+(\d+)" % k, blk)}


@pytest.fixture(scope="module")
def gemm_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "gemm_w4.hip")


def test_dx_v3_agpr_ring_is_untouched_by_compiler_code(gemm_asm):
    """The ring lives in the LOADER waves (waves 4..7: `if (wave >= 4) { ... return; }`, one contiguous stretch of the kernel's
    text); the compute waves' MFMA accumulators may use the same register numbers in their own stretch.  Between the first and
    the last asm statement that names a ring register, no compiler-generated instruction may name an AGPR below 32."""
    for name in sorted(set(re.findall(r"^(_ZN4qeft20gemm_w4_dx_kernel_v3\w*):", gemm_asm, re.M))):      # every instantiation (bits, tile rows)
        _check_dx_ring(gemm_asm, name)


def _check_dx_ring(gemm_asm, name):
    _, body = _kernel_body(gemm_asm, name)
    agpr = re.compile(r"(?<![\w.])a(\d+|\[\d+:\d+\])(?![\w])")
    rows, in_asm = [], False                   # (in_asm, low AGPRs named, text)
    for ln in body.split("\n"):
        l = ln.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        code = l.split(";")[0]
        regs = []
        for m in agpr.finditer(code):
            t = m.group(1)
            regs += range(int(t[1:].split(":")[0]), int(t[:-1].split(":")[1]) + 1) if t.startswith("[") else [int(t)]
        rows.append((in_asm, [r for r in regs if r < 32], code))
    ring = [i for i, (a, low, _) in enumerate(rows) if a and low]
    loads = sum(rows[i][2].startswith("global_load") for i in ring)
    reads = sum(rows[i][2].startswith("v_accvgpr_read") for i in ring)
    assert loads >= 8 and reads >= 8, (loads, reads)          # the ring is really there (the guard is not vacuous)
    assert all(max(rows[i][1]) < 31 for i in ring)            # a0..a29: four sets of six registers on even tuples; a30: the L2 touch's sink
    offenders = [c for a, low, c in rows[ring[0]:ring[-1] + 1] if not a and low]
    assert not offenders, offenders[:5]
    # and no MFMA at all in that stretch: it is the loader waves' code
    assert not any("v_mfma" in c for _, _, c in rows[ring[0]:ring[-1] + 1])
    md = _metadata(gemm_asm, name)
    assert md.get("vgpr_spill_count", 0) == 0 and md.get("private_segment_fixed_size", 0) == 0, md


@pytest.mark.parametrize("src,at_least", [("gemv_v3.hip", 64), ("gemv_v3_plain.hip", 64)])
def test_decode_gemv_kernels_do_not_spill(tmp_path_factory, src, at_least):
    text = _asm(tmp_path_factory, src)
    names = re.findall(r"\.name:\s+(_ZN4qeft14gemv_v3_kernel\w+)", text)
    assert len(names) >= at_least
    bad = {}
    for n in set(names):
        md = _metadata(text, n)
        if md.get("vgpr_spill_count", 0) or md.get("private_segment_fixed_size", 0):
            bad[n] = md
    assert not bad, bad
