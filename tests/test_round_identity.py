"""The quantize kernels compute f32::round (half away from zero, q8_0.rs:36 and the like) as
trunc(x + copysign(pred(0.5), x)).  The identity was checked over every float with |x| < 512 when
it was introduced (about a minute of numpy); this keeps the ties, their neighbours and a random
sample in the suite."""
import numpy as np


def _both(x):
    x = x.astype(np.float32)
    half_below = np.nextafter(np.float32(0.5), np.float32(0))
    got = np.trunc(x + np.copysign(half_below, x))
    x64 = x.astype(np.float64)
    want = (np.sign(x64) * np.floor(np.abs(x64) + 0.5)).astype(np.float32)
    return got, want


def test_ties_and_neighbours():
    k = np.arange(-600, 601, dtype=np.float32)
    base = np.concatenate([k, k + np.float32(0.5), k - np.float32(0.5), k + np.float32(0.25)])
    pts = np.concatenate([base, np.nextafter(base, np.float32(np.inf)), np.nextafter(base, np.float32(-np.inf)),
                          np.array([0.0, -0.0, 0.49999997, -0.49999997, 1e-30, -1e-30], dtype=np.float32)])
    got, want = _both(pts)
    assert np.array_equal(got, want)


def test_random_sample():
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 1 << 32, 4_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x) & (np.abs(x) < 512)]
    x = np.concatenate([x, (rng.standard_normal(2_000_000) * 60).astype(np.float32)])
    got, want = _both(x)
    assert np.array_equal(got, want)
