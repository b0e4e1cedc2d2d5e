"""The reference's own tests for this path, restated against the oracle.

They are all the reference holds: tolerance round trips on one block of U[0,1)
values (ggml-quants/src/lib.rs:173-194, thresholds in each structs/*.rs test), and
the slice-API cases instantiated for Q8_0 (lib.rs:274-369).  The reference draws
unseeded values; here 200 seeded blocks per format are drawn instead of one.
"""
import numpy as np
import pytest

# (type, elements, abs threshold): q4_0.rs:62, q4_1.rs:65, q5_0.rs:78, q5_1.rs:82,
# q8_0.rs:52, q8_1.rs:66, q8_k.rs:65, half.rs:42, half.rs:47
TOL = [(2, 32, 8e-2), (3, 32, 4e-2), (6, 32, 4e-2), (7, 32, 2e-2), (8, 32, 4.5e-3),
       (9, 32, 4.5e-3), (15, 256, 4.5e-3), (1, 1, 4e-3), (30, 1, 8e-3)]


@pytest.mark.parametrize("ty,n,tol", TOL)
def test_tolerance_round_trip(oracle, ty, n, tol):
    rng = np.random.default_rng(1000 + ty)
    x = rng.random(200 * n, dtype=np.float32)          # rand::rng().fill(&mut [f32]) -> U[0,1)
    y = oracle.dequantize(ty, oracle.quantize(ty, x))
    # lib.rs:237 with rel threshold 0: an outlier is abs > tol (and rel > 0)
    assert float(np.max(np.abs(x - y))) <= tol


N = 32
Q8_0 = 8


def _deq_status(oracle, dst_elems, blocks):
    dst = np.zeros(max(dst_elems, 1), dtype=np.float32)
    src = np.zeros(max(blocks, 1) * 34, dtype=np.uint8)
    return oracle.dequantize_status(Q8_0, oracle.F32, dst, dst_elems, src, blocks)


def _q_status(oracle, src_elems, blocks):
    src = np.zeros(max(src_elems, 1), dtype=np.float32)
    dst = np.zeros(max(blocks, 1) * 34, dtype=np.uint8)
    return oracle.quantize_status(Q8_0, oracle.F32, dst, blocks, src, src_elems)


def test_quant_dequant_slice_ok(oracle):            # lib.rs:276-291
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, 2 * N).astype(np.float32)
    y = oracle.dequantize(Q8_0, oracle.quantize(Q8_0, x))
    assert np.all(np.abs(x - y) < 4.5e-3)


def test_quant_slice_indivisible(oracle):           # lib.rs:293-301
    assert _q_status(oracle, N - 1, 1) == oracle.INDIVISIBLE


def test_quant_slice_length_mismatch(oracle):       # lib.rs:303-311
    assert _q_status(oracle, 2 * N, 3) == oracle.LENGTH_MISMATCH


def test_dequant_slice_indivisible(oracle):         # lib.rs:313-321
    assert _deq_status(oracle, N - 1, 1) == oracle.INDIVISIBLE


def test_dequant_slice_length_mismatch(oracle):     # lib.rs:323-331
    assert _deq_status(oracle, 2 * N, 3) == oracle.LENGTH_MISMATCH


def test_indivisible_is_checked_first(oracle):      # lib.rs:136-141 order
    assert _deq_status(oracle, N - 1, 3) == oracle.INDIVISIBLE


def test_block_zeros_and_count(oracle):             # lib.rs:333-338
    assert oracle.block_elems(Q8_0) == N
    blk = oracle.quantize(Q8_0, np.zeros(N, dtype=np.float32))
    assert not blk.any()
    y = oracle.dequantize(Q8_0, blk)
    assert np.array_equal(y.view(np.uint32), np.zeros(N, dtype=np.uint32))


@pytest.mark.parametrize("in_ty,kind", [(1, "f16"), (30, "bf16")])
def test_quantize_half_inputs(oracle, in_ty, kind):  # lib.rs:340-364
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, N).astype(np.float32)
    xin = oracle.quantize(in_ty, x).view(np.uint16)            # f16::from_f32 per element
    blk = oracle.quantize(Q8_0, xin, in_ty=in_ty)
    y16 = oracle.dequantize(Q8_0, blk, in_ty)
    back = oracle.dequantize(in_ty, y16.view(np.uint8))
    src = oracle.dequantize(in_ty, xin.view(np.uint8))
    assert np.all(np.abs(src - back) < 4.5e-3 * (2 if kind == "bf16" else 1))


def test_layout_names():                            # types.rs:35-62
    import json, os
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "type_table.json")))
    names = {t["id"]: t["name"] for t in doc["types"]}
    assert names[2] == "q40" and names[12] == "q4k" and names[20] == "iq4nl" and names[15] == "q8k"


@pytest.mark.parametrize("ty", [2, 3, 6, 7, 8, 9, 15])
def test_signalling_nan_elements_fold_like_quiet_ones(oracle, ty):
    """structs.rs:91-107: `acc.max(x.abs())`, `x.abs() > acc.abs()` and f32::min / f32::max skip a NaN element; the
    compiled reference compares and selects, so a signalling NaN is skipped like a quiet one.  (C's fmaxf answers a
    signalling NaN with a NaN and the fold then forgets what it held: the oracle once did that.)  Every position of
    a block, each signalling pattern against the same block with a quiet NaN in its place."""
    n = oracle.block_elems(ty)
    rng = np.random.default_rng(5 + ty)
    base = (rng.standard_normal(n) * 0.7).astype(np.float32)
    for pos in range(n):
        for pat in (0x7FA00000, 0xFF800001, 0x7F800400):
            s = base.copy(); s.view(np.uint32)[pos] = pat
            q = base.copy(); q[pos] = np.nan
            a, b = oracle.quantize(ty, s), oracle.quantize(ty, q)
            assert np.array_equal(a, b), (pos, hex(pat))
