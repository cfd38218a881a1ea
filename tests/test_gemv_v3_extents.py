"""CPU-only: every address the v3 decode GEMV can form stays inside its operand.

The kernel (qeft_amd/csrc/gemv_v3.h) issues all of its loads unconditionally with clamped addresses -- the pattern behind
both device faults of round 1.  Its address arithmetic is a set of __host__ __device__ functions;
qeft_gemv_v3_check_extents walks every block x wave x lane x {DMA piece, ring issue incl. the clamped ones past the end,
epilogue operand, output row} of a configuration with those same functions on the host and counts accesses outside the
operand sizes."""
import pytest


@pytest.fixture(scope="module")
def lib():
    from qeft_amd import _lib, build
    build.build(verbose=False)
    return _lib.lib()


ROWS = (16, 32, 48, 512, 640, 1376, 1728, 4096, 16 * 513, 16 * 769, 8192, 11008, 12288, 16 * 1025, 22016, 27648, 28672, 16 * 3001, 57344)


@pytest.mark.parametrize("k", [128, 256, 384, 1024, 4096, 4224, 5120, 8192, 11008, 13824, 28672])
@pytest.mark.parametrize("r", [0, 128])
def test_all_row_counts(lib, k, r):
    if r and k == 128:
        assert lib.qeft_gemv_v3_check_extents(16, k, 128, r, 0, 0) == -1      # no INT4 column left: not a v3 configuration
        return
    for n in ROWS:
        assert lib.qeft_gemv_v3_check_extents(n, k, 128, r, 0, 0) == 0, (n, k, r)
    assert lib.qeft_gemv_v3_check_extents(4096, k, k, r, 0, 0) == 0          # per-channel: one group


@pytest.mark.parametrize("n_ssq", [1, 2, 3, 4, 5, 255, 256, 257, 320, 511, 512])
def test_partial_sums_of_squares_piece(lib, n_ssq):
    assert lib.qeft_gemv_v3_check_extents(4096, 4096, 128, 128, n_ssq, 0) == 0
    assert lib.qeft_gemv_v3_check_extents(16, 256, 128, 128, n_ssq, 0) == 0


def test_rejected_configurations(lib):
    assert lib.qeft_gemv_v3_check_extents(24, 256, 128, 0, 0, 0) == -1            # rows not a multiple of 16
    assert lib.qeft_gemv_v3_check_extents(16, 192, 64, 0, 0, 0) == -1             # K not a multiple of 128 / group 64
    assert lib.qeft_gemv_v3_check_extents(16, 256, 128, 64, 0, 0) == -1           # r other than 0 / 128
    assert lib.qeft_gemv_v3_check_extents(16, 256, 128, 128, 513, 0) == -1        # too many partial sums


def test_the_enumerator_sees_an_operand_that_is_too_short(lib):
    """Negative control: with the operands one 16-row set shorter than the geometry, the accesses of the last set must count."""
    for n in (16, 4096, 22016):
        assert lib.qeft_gemv_v3_check_extents(n, 4096, 128, 128, 0, 16) > 0
        assert lib.qeft_gemv_v3_check_extents(n, 4096, 128, 0, 0, 16) > 0


# ---- the launches the reference's gemv entries make on the checkpoint-layout operands (round 3)
@pytest.mark.parametrize("k", [256, 1024, 4096, 5120, 11008, 13824])
@pytest.mark.parametrize("m", [1, 2, 4, 7, 8, 9, 16])
@pytest.mark.parametrize("gather", [0, 1])
def test_checkpoint_layout_launches(lib, k, m, gather):
    for n in (16, 48, 640, 4096, 16 * 513, 11008, 13824):
        for r in (0, 128):
            assert lib.qeft_gemv_v3_check_extents_ckpt(n, k, 128, r, m, gather, 0) == 0, (n, k, r, m, gather)
    assert lib.qeft_gemv_v3_check_extents_ckpt(4096, k, k, 128, m, gather, 0) == 0      # per-channel scales [1][n]


def test_checkpoint_layout_negative_control(lib):
    """Operands one 16-row set shorter than the geometry: the scale rows' stride and the last set's pieces must count."""
    for n in (16, 4096, 11008):
        assert lib.qeft_gemv_v3_check_extents_ckpt(n, 4096, 128, 128, 3, 1, 16) > 0
        assert lib.qeft_gemv_v3_check_extents_ckpt(n, 4096, 128, 0, 1, 0, 16) > 0
    assert lib.qeft_gemv_v3_check_extents_ckpt(4096, 4096, 128, 128, 17, 0, 0) == -1      # m outside 1..16
    assert lib.qeft_gemv_v3_check_extents_ckpt(4096, 4096, 64, 128, 1, 0, 0) == -1        # group size the v3 launch does not take
