"""ggq_selftest_fast_division on the GPU at hand: the short division sequences of the quantizers (constant divisor;
reciprocal from v_rcp_f32 and two fused correction steps) against the compiler's general IEEE division, for every
float in their range.  Bit-exactness of the quantizers for ordinary blocks rests on this property of the hardware's
reciprocal unit; the parity tests can only sample it."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def test_short_divisions_equal_ieee_division_for_every_input(torch_cuda):
    import gguf_amd as g
    bad, tested = g.selftest_fast_division()
    # 200 binades (2^-100 up to, not including, 2^100) x 2^23 mantissas, plus 2^100 itself; both signs
    assert tested == 2 * (200 * (1 << 23) + 1)
    assert bad == {"a/15": 0, "a/31": 0, "a/127": 0, "a/-127": 0, "1/d": 0}
