"""The C oracle against the committed golden vectors and the naive Python
restatement (tests/pyref.py)."""
import os

import numpy as np
import pytest

import hand_kats
import pyref
from util import same_bits, same_bits16, first_diff

G = os.path.join(os.path.dirname(__file__), "golden")
BLOCK_TYPES = [2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 20, 23]


@pytest.fixture(scope="module")
def seeded():
    return np.load(os.path.join(G, "seeded.npz"))


@pytest.mark.parametrize("ty", BLOCK_TYPES + [1, 30])
def test_seeded_vectors(oracle, seeded, ty):
    p = seeded[f"t{ty}_packed"]
    assert same_bits(oracle.dequantize(ty, p), seeded[f"t{ty}_f32"])
    assert same_bits16(oracle.dequantize(ty, p, oracle.F16), seeded[f"t{ty}_f16"], "f16")
    assert same_bits16(oracle.dequantize(ty, p, oracle.BF16), seeded[f"t{ty}_bf16"], "bf16")
    if f"t{ty}_input" in seeded.files:   # the quantizer reproduces the packed bytes
        assert np.array_equal(oracle.quantize(ty, seeded[f"t{ty}_input"]), p)


@pytest.mark.parametrize("ty", BLOCK_TYPES)
def test_kat_blocks(oracle, ty):
    kat = np.load(os.path.join(G, "kat.npz"))
    got = oracle.dequantize(ty, kat[f"t{ty}_packed"])
    assert same_bits(got, kat[f"t{ty}_f32"]), first_diff(got, kat[f"t{ty}_f32"])


def test_kat_hand_values(oracle):
    """Values worked out by hand from the formulas, d = 1, min = 0.5."""
    # Q4_0: qs[i] = ((i+1)&15)<<4 | (i&15)  ->  y[i] = i-8, y[i+16] = ((i+1)&15)-8
    blk = np.zeros(18, dtype=np.uint8)
    blk[0:2] = [0x00, 0x3C]
    for i in range(16):
        blk[2 + i] = (((i + 1) & 15) << 4) | (i & 15)
    y = oracle.dequantize(2, blk)
    assert list(y[:16]) == [i - 8 for i in range(16)]
    assert list(y[16:]) == [((i + 1) & 15) - 8 for i in range(16)]
    # Q4_1 same payload: y = nib + 0.5
    blk = np.concatenate([np.array([0x00, 0x3C, 0x00, 0x38], dtype=np.uint8), blk[2:]])
    y = oracle.dequantize(3, blk)
    assert list(y[:16]) == [i + 0.5 for i in range(16)]
    # Q8_0: quants -16..15
    blk = np.concatenate([np.array([0x00, 0x3C], dtype=np.uint8), np.arange(-16, 16, dtype=np.int8).view(np.uint8)])
    assert list(oracle.dequantize(8, blk)) == list(range(-16, 16))
    # Q5_0: all ql nibbles 0, qh = 0xFFFF0000 -> low half -16, high half 0
    blk = np.zeros(22, dtype=np.uint8)
    blk[0:2] = [0x00, 0x3C]
    blk[2:6] = [0, 0, 0xFF, 0xFF]
    y = oracle.dequantize(6, blk)
    assert list(y[:16]) == [-16.0] * 16 and list(y[16:]) == [0.0] * 16
    # Q4_K: d=1, dmin=0.5, scales: sc[j]=j+1 (j<4), mn[j]=2 ; sub-blocks 4..7 sc=1, mn=0
    blk = np.zeros(144, dtype=np.uint8)
    blk[0:4] = [0x00, 0x3C, 0x00, 0x38]
    blk[4:8] = [1, 2, 3, 4]
    blk[8:12] = [2, 2, 2, 2]
    blk[12:16] = [1, 1, 1, 1]            # sc[4..7] = 1 (low nibble), mn[4..7] = 0
    blk[16:] = 0x21                      # low nibble 1, high nibble 2
    y = oracle.dequantize(12, blk)
    assert list(y[0:32]) == [1 * 1 - 1.0] * 32      # d*sc0*1 - dmin*2
    assert list(y[32:64]) == [2 * 2 - 1.0] * 32     # high nibbles with sc1 = 2
    assert list(y[64:96]) == [3 * 1 - 1.0] * 32
    assert list(y[128:160]) == [1.0] * 32           # sc4 = 1, mn4 = 0, low nibble
    assert list(y[160:192]) == [2.0] * 32
    # Q6_K: ql=0, qh=0 -> q=-32 ; scales = 1 -> y = -32 ; scale -2 -> +64
    blk = np.zeros(210, dtype=np.uint8)
    blk[192:208] = 1
    blk[192] = 0xFE                      # int8 -2 for elements 0..15
    blk[208:210] = [0x00, 0x3C]
    y = oracle.dequantize(14, blk)
    assert list(y[:16]) == [64.0] * 16 and list(y[16:32]) == [-32.0] * 16


@pytest.mark.parametrize("case", hand_kats.cases(), ids=lambda c: f"type{c[0]}")
def test_hand_derived_blocks(oracle, case):
    """Expected values typed in by hand (tests/hand_kats.py): the 6-bit scale repacks with non-zero top bits,
    qh / hmask bit walks, negative scales, all 16 IQ4 codes.  Compared bit for bit (also the sign of zero).
    The same blocks must also pass through the naive Python restatement."""
    ty, blk, want = case
    got = oracle.dequantize(ty, blk)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), first_diff(got, want)
    for out_ty, dt in ((oracle.F16, np.float16),):
        got16 = oracle.dequantize(ty, blk, out_ty)
        assert np.array_equal(got16, want.astype(dt).view(np.uint16))     # numpy narrows with round-to-nearest-even too
    b = pyref.decode(ty, blk, oracle.block_bytes(ty))
    assert np.array_equal(b.view(np.uint32), want.view(np.uint32)), first_diff(b, want)


@pytest.mark.parametrize("ty", BLOCK_TYPES)
@pytest.mark.parametrize("adversarial", [False, True])
def test_oracle_vs_naive_python(oracle, ty, adversarial):
    from gguf_amd import synth
    p = synth.random_blocks(ty, 5, 77 + ty, adversarial=adversarial)
    a = oracle.dequantize(ty, p)
    b = pyref.decode(ty, p, oracle.block_bytes(ty))
    assert same_bits(a, b), first_diff(a, b)


@pytest.mark.parametrize("ty", BLOCK_TYPES)
def test_edge_scales(oracle, ty):
    e = np.load(os.path.join(G, "edges.npz"))
    p = e[f"t{ty}_packed"]
    assert same_bits(oracle.dequantize(ty, p), e[f"t{ty}_f32"])
    bb = oracle.block_bytes(ty)
    some = p[: 12 * bb]
    assert same_bits(oracle.dequantize(ty, some), pyref.decode(ty, some, bb))


def test_threads_do_not_change_results(oracle):
    from gguf_amd import synth
    p = synth.random_blocks(12, 1000, 3)
    assert np.array_equal(oracle.dequantize(12, p, threads=1), oracle.dequantize(12, p, threads=7))


@pytest.mark.parametrize("case", hand_kats.quantizer_cases(), ids=lambda c: f"type{c[0]}")
def test_quantizer_hand_kats(oracle, case):
    """Packed bytes typed in by hand (tests/hand_kats.py), not produced by any quantizer: header bits, the
    max_by_abs sign rule (Q4_0 / Q5_0: the block's extreme is negative), nibble order, fifth-bit word."""
    ty, x, want = case
    got = oracle.quantize(ty, x)
    assert got.tolist() == want.tolist()
