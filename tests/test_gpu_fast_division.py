"""ggq_selftest_fast_division on the GPU at hand: the short division sequences of the quantizers (constant divisor;
reciprocal from v_rcp_f32 and two fused correction steps) against the compiler's general IEEE division, for every
float in their range.  Bit-exactness of the quantizers for ordinary blocks rests on this property of the hardware's
reciprocal unit; the parity tests can only sample it."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def test_short_divisions_equal_ieee_division_for_every_input(torch_cuda):
    from gguf_amd import _lib
    bad = (ctypes.c_uint64 * 5)()
    tested = ctypes.c_uint64(0)
    rc = _lib.lib.ggq_selftest_fast_division(bad, ctypes.byref(tested))
    assert rc == 0, _lib.lib.ggq_last_error()
    # 201 binades (2^-100 .. 2^100 exclusive of the top binade's interior) x 2^23 mantissas x 2 signs, plus 2^100 itself
    assert tested.value == 2 * (200 * (1 << 23) + 1)
    assert list(bad) == [0, 0, 0, 0, 0], dict(zip(("a/15", "a/31", "a/127", "a/-127", "1/d"), bad))
