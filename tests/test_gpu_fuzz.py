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
       blocks=st.one_of(st.integers(1, 70), st.integers(100, 1500)), kind=st.sampled_from(["normal", "wide", "special", "extreme"]),
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
        x.view(np.uint32)[pick == 7] = 0x7FA00000 | rng.integers(0, 1 << 20, int((pick == 7).sum())).astype(np.uint32)  # signalling
    elif kind == "extreme":                # one power of two per block, over the whole exponent range: blocks on both
        if ty not in (1, 30):              # sides of the limits between which the encoders take their short divisions
            xb = x.reshape(blocks, -1)
            with np.errstate(over="ignore"):       # products beyond f32 become infinities: wanted
                xb *= (np.float32(2.0) ** rng.integers(-126, 127, blocks).astype(np.float32))[:, None]
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


@settings(**_cfg)
@given(out=st.sampled_from(["f32", "f16", "bf16"]),
       plan=st.lists(st.tuples(st.sampled_from(ALL_TYPES), st.one_of(st.integers(0, 40), st.integers(100, 2500)),
                               st.integers(0, 15), st.booleans()), min_size=1, max_size=12),
       seed=st.integers(0, 2 ** 31))
def test_launch_plan_matches_oracle(oracle, out, plan, seed):
    """A drawn launch plan: up to 12 tensors of drawn types, sizes (also empty ones), source offsets and payload
    kinds, one output type; every tensor must come out as the oracle decodes it."""
    import torch as t
    import gguf_amd as g
    from gguf_amd import synth
    from util import same_bits16, first_diff
    out_dt = {"f32": t.float32, "f16": t.float16, "bf16": t.bfloat16}[out]
    o_out = {"f32": oracle.F32, "f16": oracle.F16, "bf16": oracle.BF16}[out]
    tensors, packed = [], []
    for i, (ty, blocks, off, adversarial) in enumerate(plan):
        if ty in (1, 30):
            blocks *= 41
            off &= ~7                     # the widening kernel wants its source 8-byte aligned... or takes the scalar path
            off |= (i & 1) * 2            # so exercise both
        p = synth.random_blocks(ty, blocks, seed + i, adversarial=adversarial) if blocks else np.zeros(0, dtype=np.uint8)
        buf = t.empty(p.size + off + 16, dtype=t.uint8, device="cuda")
        src = buf[off:off + p.size]
        if p.size:
            src.copy_(t.from_numpy(p))
        dst = t.full((blocks * oracle.block_elems(ty),), -1.0, dtype=out_dt, device="cuda")
        tensors.append((ty, src, dst))
        packed.append(p)
    b = g.Batch(tensors, {"f32": g.GGmlType.F32, "f16": g.GGmlType.F16, "bf16": g.GGmlType.BF16}[out])
    b.run()
    t.cuda.synchronize()
    for (ty, _, d), p in zip(tensors, packed):
        if p.size == 0:
            continue
        want = oracle.dequantize(ty, p, o_out)
        if out == "f32":
            assert same_bits(d.cpu().numpy(), want), (ty, p.size, first_diff(d.cpu().numpy(), want))
        else:
            got = d.view(t.int16).cpu().numpy().view(np.uint16)
            assert same_bits16(got, want, out), (ty, p.size, first_diff(got, want))
    b.close()


@settings(**_cfg)
@given(ty=st.sampled_from([12, 2, 14, 0, 1, 8, 24, 10]), blocks_per_row=st.integers(1, 40), nh=st.integers(1, 9),
       half=st.integers(1, 70), seed=st.integers(0, 2 ** 31))
def test_permute_qk_matches_numpy(ty, blocks_per_row, nh, half, seed):
    """Drawn row sizes (so pieces per row of 16, 4, 2 or 1 bytes, powers of two or not) and head sizes."""
    import torch
    from gguf_amd import rearrange as R
    from gguf_amd._lib import lib
    from test_gpu_rearrange import np_permute_qk
    row_elems = blocks_per_row * lib.ggq_gguf_type_elems(ty)
    rows = nh * 2 * half
    rb = R.row_bytes(ty, row_elems)
    src = np.random.default_rng(seed).integers(0, 256, rows * rb, dtype=np.uint8)
    got = R.permute_qk(ty, row_elems, rows, nh, torch.from_numpy(src).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), np_permute_qk(src, rb, rows, nh)), (ty, row_elems, nh, half)


@settings(**_cfg)
@given(ty=st.sampled_from([0, 1, 30, 28]), n=st.one_of(st.integers(1, 300), st.integers(1000, 200000)),
       offset=st.integers(0, 7), factor=st.floats(allow_nan=False, allow_infinity=False, width=64),
       wide=st.booleans(), seed=st.integers(0, 2 ** 31))
def test_scale_matches_oracle(oracle, ty, n, offset, factor, wide, seed):
    """Drawn element types, lengths, element-aligned start offsets and f64 factors (any finite double, so the
    factor's own narrowing to f32 / f16 / bf16 is drawn too), against the oracle's scale_data."""
    import torch
    from gguf_amd import rearrange as R
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    if wide:
        x *= 10.0 ** rng.integers(-12, 12, n)
    if ty == 0:
        host = x.astype(np.float32)
    elif ty == 28:
        host = x
    elif ty == 1:
        with np.errstate(over="ignore"):
            host = x.astype(np.float16).view(np.uint16)
    else:
        host = (x.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    eb = host.dtype.itemsize
    want = oracle.scale(ty, host, factor)
    dev = torch.empty(host.nbytes + 64, dtype=torch.uint8, device="cuda")
    view = dev[offset * eb:offset * eb + host.nbytes]
    view.copy_(torch.from_numpy(host.view(np.uint8).copy()))
    got = R.scale(ty, view, factor).cpu().numpy()
    w8 = want.view(np.uint8)
    if not np.array_equal(got, w8):
        # NaN results (inf * 0) must be NaN on both sides; every other element bit for bit
        g = got.view(host.dtype if ty in (0, 28) else np.uint16)
        if ty in (0, 28):
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(g), nan) and np.array_equal(g[~nan].view(np.uint8), want[~nan].view(np.uint8))
        else:
            f = (lambda b: b.view(np.float16)) if ty == 1 else (lambda b: (b.astype(np.uint32) << 16).view(np.float32))
            nan = np.isnan(f(want))
            assert np.array_equal(np.isnan(f(g)), nan) and np.array_equal(g[~nan], want[~nan])
