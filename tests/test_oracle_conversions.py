"""The oracle's f16 / bf16 conversions against numpy's IEEE binary16 and a
bit-level bf16 model.  The reference delegates these to the crate `half` 2.6.0
(call sites ggml-quants/src/structs/half.rs:16,20,32,36 and lib.rs:68,72,83,88)."""
import numpy as np


def test_f16_widen_all_patterns(oracle):
    bits = np.arange(65536, dtype=np.uint16)
    got = oracle.dequantize(oracle.F16, bits.view(np.uint8))
    want = bits.view(np.float16).astype(np.float32)
    nan = np.isnan(want)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    assert np.isnan(got[nan]).all()
    # NaNs keep sign and payload and come out quiet (half's documented behaviour)
    g = got.view(np.uint32)[nan]
    b = bits[nan].astype(np.uint32)
    assert np.array_equal(g, ((b & 0x8000) << 16) | 0x7FC00000 | ((b & 0x3FF) << 13))


def test_f16_narrow_rne(oracle):
    rng = np.random.default_rng(7)
    x = rng.standard_normal(300_000).astype(np.float32) * \
        np.float32(10.0) ** rng.integers(-9, 6, 300_000).astype(np.float32)
    # every f16 value, the midpoint to its successor, and the floats either side of the midpoint
    h = np.arange(0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mid = ((h[:-1] + h[1:]) / 2).astype(np.float32)
    edge = np.concatenate([h.astype(np.float32), mid, np.nextafter(mid, np.float32(np.inf)),
                           np.nextafter(mid, np.float32(-np.inf))])
    special = np.array([0.0, -0.0, 65504, 65519.996, 65520, 1e-8, 2.9802322e-8, 2.9802326e-8,
                        np.inf, -np.inf, 1e38, -1e38, 1e-45], dtype=np.float32)
    x = np.concatenate([x, edge, -edge, special])
    got = oracle.quantize(oracle.F16, x).view(np.uint16)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, want)


def test_f16_narrow_nan(oracle):
    x = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0x7FFFE000, 0xFF802000], dtype=np.uint32).view(np.float32)
    got = oracle.quantize(oracle.F16, x).view(np.uint16)
    assert list(got) == [0x7E00, 0xFE00, 0x7E00, 0x7FFF, 0xFE01]


def test_bf16_roundtrip_and_rne(oracle):
    bits = np.arange(65536, dtype=np.uint16)
    wide = oracle.dequantize(oracle.BF16, bits.view(np.uint8))
    assert np.array_equal(wide.view(np.uint32), bits.astype(np.uint32) << 16)
    rng = np.random.default_rng(9)
    u = rng.integers(0, 2 ** 32, 400_000, dtype=np.uint64).astype(np.uint32)
    tie = (rng.integers(0, 2 ** 16, 5000, dtype=np.uint64).astype(np.uint32) << 16) | 0x8000
    u = np.concatenate([u, tie, tie + 1, tie - 1])
    nan = ((u & 0x7F800000) == 0x7F800000) & ((u & 0x007FFFFF) != 0)
    got = oracle.quantize(oracle.BF16, u.view(np.float32)).view(np.uint16)
    rnd = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    assert np.array_equal(got[~nan], rnd[~nan])
    assert np.array_equal(got[nan], ((u[nan] >> 16) | 0x40).astype(np.uint16))


def test_narrowed_outputs_match_f32_then_round(oracle):
    """lib.rs:70-73 / 86-89: f16 and bf16 outputs are the f32 result narrowed per element."""
    import sys
    from gguf_amd import synth
    for ty in (oracle.Q4_0, oracle.Q4K, oracle.Q6K):
        p = synth.random_blocks(ty, 8, 42)
        y = oracle.dequantize(ty, p)
        assert np.array_equal(oracle.dequantize(ty, p, oracle.F16), oracle.quantize(oracle.F16, y).view(np.uint16))
        assert np.array_equal(oracle.dequantize(ty, p, oracle.BF16), oracle.quantize(oracle.BF16, y).view(np.uint16))
