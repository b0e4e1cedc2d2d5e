"""Property-based parity: hypothesis draws (block type, output type, block count, source byte
offset, benign or adversarial payload, entry point) and the device result must equal the CPU
oracle's, bit for bit (NaN must meet NaN).  Derandomized, so every run checks the same cases."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from test_gpu_parity import ALL_TYPES, check, gpu_decode
from util import same_bits

pytestmark = pytest.mark.gpu

QUANT_TYPES = [1, 30, 2, 3, 6, 7, 8, 9, 15]
import os  # noqa: E402

# GGQ_FUZZ_EXAMPLES=2000 for a longer one-off run
_cfg = dict(max_examples=int(os.environ.get("GGQ_FUZZ_EXAMPLES", "120")), deadline=None, derandomize=True,
            suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@settings(**_cfg)
@given(ty=st.sampled_from(ALL_TYPES), out=st.sampled_from(["f32", "f16", "bf16"]),
       blocks=st.one_of(st.integers(1, 70), st.integers(100, 3000)), offset=st.integers(0, 31),
       adversarial=st.booleans(), seed=st.integers(0, 2 ** 31))
def test_decode_matches_oracle(oracle, ty, out, blocks, offset, adversarial, seed):
    import torch
    from gguf_amd import synth
    if ty in (1, 30):
        blocks *= 37                      # 1-element "blocks": make the vector path and its tail both run
        offset &= ~1                      # element alignment
    packed = synth.random_blocks(ty, blocks, seed, adversarial=adversarial)
    check(oracle, ty, packed, gpu_decode(torch, ty, packed, out, offset), out)


@settings(**_cfg)
@given(ty=st.sampled_from(QUANT_TYPES), in_ty=st.sampled_from(["f32", "f16", "bf16"]),
       blocks=st.one_of(st.integers(1, 70), st.integers(100, 1500)), kind=st.sampled_from(["normal", "wide", "special"]),
       seed=st.integers(0, 2 ** 31))
def test_quantize_matches_oracle(oracle, ty, in_ty, blocks, kind, seed):
    import torch
    import gguf_amd as g
    rng = np.random.default_rng(seed)
    n = blocks * oracle.block_elems(ty)
    if ty in (1, 30):
        n *= 19
    x = rng.standard_normal(n).astype(np.float32)
    if kind == "wide":
        x *= np.float32(10.0) ** rng.integers(-30, 30, n).astype(np.float32)
    elif kind == "special":
        pick = rng.integers(0, 12, n)
        x[pick == 0] = 0.0
        x[pick == 1] = -0.0
        x[pick == 2] = np.inf
        x[pick == 3] = -np.inf
        x[pick == 4] = np.nan
        x[pick == 5] = x[0]                # ties of the maximum
        x[pick == 6] = -x[0]
    tx = torch.from_numpy(x).cuda()
    if in_ty == "f16":
        tx = tx.to(torch.float16)
        host, o_in = tx.view(torch.int16).cpu().numpy().view(np.uint16), oracle.F16
    elif in_ty == "bf16":
        tx = tx.to(torch.bfloat16)
        host, o_in = tx.view(torch.int16).cpu().numpy().view(np.uint16), oracle.BF16
    else:
        host, o_in = x, oracle.F32
    if (ty, in_ty) in ((1, "f16"), (30, "bf16")):
        return                             # same type: the reference has no such cast
    want = oracle.quantize(ty, host, o_in)
    dst = torch.empty(want.size, dtype=torch.uint8, device="cuda")
    g.quantize_slice(ty, dst, tx)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    if ty in (1, 30):                      # element formats: a NaN may differ in payload
        a, b = got.view(np.uint16), want.view(np.uint16)
        nan = (lambda v: (v & 0x7C00 == 0x7C00) & (v & 0x3FF != 0)) if ty == 1 else \
              (lambda v: (v & 0x7F80 == 0x7F80) & (v & 0x7F != 0))
        ok = (a == b) | (nan(a) & nan(b))
        assert ok.all(), np.flatnonzero(~ok)[:5]
    else:
        assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
