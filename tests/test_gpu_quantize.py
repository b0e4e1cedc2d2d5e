"""Parity of the GPU quantizers with the CPU oracle, through the C ABI: packed bytes
must be identical (integer codes, f16 headers, Q8_1 / Q8K sums)."""
import numpy as np
import pytest

from util import same_bits

pytestmark = pytest.mark.gpu

QUANT = [2, 3, 6, 7, 8, 9, 15]          # Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q8_1 Q8K
IN = {"f32": 0, "f16": 1, "bf16": 30}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def to_in(oracle, x, kind):
    """float32 values -> the array handed to the quantizer (uint16 patterns for 16-bit inputs)."""
    if kind == "f32":
        return x
    return oracle.quantize(oracle.F16 if kind == "f16" else oracle.BF16, x).view(np.uint16)


def gpu_quantize(torch, ty, arr, kind):
    import gguf_amd as g
    n = arr.size
    blocks = n // g.block_elems(ty)
    if kind == "f32":
        src = torch.from_numpy(arr).cuda()
    else:
        dt = torch.float16 if kind == "f16" else torch.bfloat16
        src = torch.from_numpy(arr.view(np.int16)).cuda().view(dt)
    dst = torch.full((blocks * g.block_bytes(ty) + 16,), 0xAB, dtype=torch.uint8, device="cuda")
    g.quantize_slice(ty, dst[: blocks * g.block_bytes(ty)], src)
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    assert (out[blocks * g.block_bytes(ty):] == 0xAB).all(), "wrote past the end"
    return out[: blocks * g.block_bytes(ty)]


def check(oracle, ty, got, arr, kind):
    want = oracle.quantize(ty, arr, in_ty=IN[kind])
    if not np.array_equal(got, want):
        bb = oracle.block_bytes(ty)
        bad = np.nonzero(got != want)[0]
        b = int(bad[0]) // bb
        raise AssertionError(f"type {ty} in {kind}: {bad.size} bytes differ, first in block {b} at byte "
                             f"{int(bad[0]) % bb}: got {got[b*bb:(b+1)*bb].tolist()} want {want[b*bb:(b+1)*bb].tolist()}")


@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_hand_kats(torch_cuda, oracle, kind):
    """The quantizer blocks of tests/hand_kats.py (expected bytes typed in by hand): alone, and repeated so that
    whole waves of them run together; their inputs are small integers, exact in f16 and bf16 as well."""
    import hand_kats
    for ty, x, want in hand_kats.quantizer_cases():
        for reps in (1, 200):
            arr = to_in(oracle, np.tile(x, reps), kind)
            got = gpu_quantize(torch_cuda, ty, arr, kind)
            assert got.tolist() == np.tile(want, reps).tolist(), (ty, kind, reps)


@pytest.mark.parametrize("ty", QUANT)
@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_random_blocks(torch_cuda, oracle, ty, kind):
    n = oracle.block_elems(ty)
    rng = np.random.default_rng(100 + ty)
    tile = 4096 if kind == "f32" else 8192
    for blocks in (1, 2, tile // n - 1, tile // n, tile // n + 1, 5 * tile // n + 3):
        x = (rng.standard_normal(blocks * n) * rng.choice([1e-3, 0.02, 1.0, 300.0])).astype(np.float32)
        arr = to_in(oracle, x, kind)
        check(oracle, ty, gpu_quantize(torch_cuda, ty, arr, kind), arr, kind)


@pytest.mark.parametrize("ty", QUANT)
def test_uniform_01_like_the_reference_tests(torch_cuda, oracle, ty):
    """U[0,1) blocks as in lib.rs:173-194, then the reference's tolerance on the GPU round trip."""
    import gguf_amd as g
    n = oracle.block_elems(ty)
    tol = {2: 8e-2, 3: 4e-2, 6: 4e-2, 7: 2e-2, 8: 4.5e-3, 9: 4.5e-3, 15: 4.5e-3}[ty]
    x = np.random.default_rng(ty).random(300 * n, dtype=np.float32)
    packed = gpu_quantize(torch_cuda, ty, x, "f32")
    check(oracle, ty, packed, x, "f32")
    t = torch_cuda
    dst = t.empty(x.size, dtype=t.float32, device="cuda")
    g.dequantize_slice(ty, dst, t.from_numpy(packed).cuda())
    assert float(np.max(np.abs(dst.cpu().numpy() - x))) <= tol


@pytest.mark.parametrize("ty", QUANT)
def test_order_and_tie_rules(torch_cuda, oracle, ty):
    """max_by_abs keeps the first element of maximal magnitude and its sign; zero blocks;
    all-equal blocks (min == max); signed zeros; values that clamp."""
    n = oracle.block_elems(ty)
    rng = np.random.default_rng(7)
    rows = []
    base = rng.standard_normal(n).astype(np.float32)
    for first, second in ((3, n - 2), (n - 2, 3), (0, n - 1), (n // 2 - 1, n // 2)):
        for s1, s2 in ((1, -1), (-1, 1)):
            r = base.copy() * 0.1
            r[first], r[second] = s1 * 2.5, s2 * 2.5       # equal magnitudes, opposite signs
            rows.append(r)
    rows.append(np.zeros(n, dtype=np.float32))
    rows.append(np.full(n, -0.0, dtype=np.float32))
    z = np.zeros(n, dtype=np.float32); z[1::2] = -0.0
    rows.append(z)
    rows.append(np.full(n, 0.7311, dtype=np.float32))                   # min == max
    rows.append(np.full(n, -3.0, dtype=np.float32))
    r = base.copy(); r[5] = 1e30; rows.append(r)                         # huge outlier
    r = base.copy() * 1e-30; rows.append(r)                              # tiny values (f16 delta underflows)
    r = base.copy() * 1e-42; rows.append(r)                              # f32 subnormals
    r = np.linspace(-1, 1, n).astype(np.float32); rows.append(r)
    r = np.linspace(1, -1, n).astype(np.float32); rows.append(r)
    x = np.concatenate(rows)
    for kind in ("f32", "f16", "bf16"):
        arr = to_in(oracle, x, kind)
        check(oracle, ty, gpu_quantize(torch_cuda, ty, arr, kind), arr, kind)


@pytest.mark.parametrize("ty", [2, 3, 6, 7, 8, 15])
def test_nan_elements_are_ignored_by_the_folds(torch_cuda, oracle, ty):
    """NaN never wins max_by_abs / min / max; its own code comes from the saturating cast
    (NaN -> 0, or 15 / 31 / 127 after `.min(c)`).  Q8_1 is left out: its `sum` header is
    0 * inf-free only without infinities, and the default NaN differs between x86 and the GPU."""
    n = oracle.block_elems(ty)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(6 * n).astype(np.float32)
    x[3] = np.nan; x[n + 0] = np.nan; x[2 * n + n - 1] = np.nan
    x[3 * n: 4 * n] = np.nan                       # a block of nothing but NaN
    x[4 * n + 7] = np.nan; x[4 * n + 8] = -np.nan
    check(oracle, ty, gpu_quantize(torch_cuda, ty, x, "f32"), x, "f32")


@pytest.mark.parametrize("ty", QUANT)
@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_signalling_nan_elements_are_ignored_like_quiet_ones(torch_cuda, oracle, ty, kind):
    """f32::max / f32::min / `>` of the reference make no difference between quiet and signalling NaN elements
    (structs.rs:91-107); the GPU's v_max_f32 / v_min_f32 do unless the operand was quieted first.  Signalling
    patterns of the input type at the start, in the middle and at the end of ordinary blocks."""
    n = oracle.block_elems(ty)
    rng = np.random.default_rng(17 + ty)
    x = (rng.standard_normal(40 * n) * 0.7).astype(np.float32)
    arr = to_in(oracle, x, kind).copy()
    snan = {"f32": [0x7FA00000, 0xFF800001, 0x7F800400], "f16": [0x7D00, 0xFC01, 0x7C40],
            "bf16": [0x7FA0, 0xFF81, 0x7F90]}[kind]
    bits = arr.view(np.uint32) if kind == "f32" else arr
    for b, pos in enumerate([0, 1, n // 2, n - 1, 3, n - 2, 7, 8]):
        bits[(3 * b) * n + pos] = snan[b % 3]
    bits[30 * n + 4] = snan[0]; bits[30 * n + 5] = snan[1]            # two in one lane's elements
    check(oracle, ty, gpu_quantize(torch_cuda, ty, arr, kind), arr, kind)


@pytest.mark.parametrize("ty", [3, 7, 8, 9])
def test_every_bf16_pattern_through_the_quieting_conversion(torch_cuda, oracle, ty):
    """The encoders that fold with v_min_f32 / v_max_f32 widen bf16 input with the hardware conversion (which quiets
    a signalling NaN) instead of a 16-bit shift.  One block per bf16 pattern p: [p, the power of two above |p|, 0 ...]
    -- p's code tells its value to 7 bits; subnormal patterns alone in their block (a conversion that flushed them
    would give code 0 instead of a saturated one); NaN patterns are ignored, infinities take the degenerate path (left out for Q8_1, whose `sum` header is
    then 0 * inf, a NaN whose bits differ between x86 and the GPU)."""
    n = oracle.block_elems(ty)
    pats = np.arange(65536, dtype=np.uint32)
    exp = (pats >> 7) & 0xFF
    if ty == 9:
        pats = pats[exp != 0xFF]
        exp = (pats >> 7) & 0xFF
    # subnormal patterns stand alone in their block: 1 / delta overflows there, every non-zero element saturates, and
    # a flushed element would show as code 0
    above = np.where(exp >= 0xFE, 0x7F7F, np.where(exp == 0, 0, (exp + 1) << 7)).astype(np.uint16)
    arr = np.zeros((pats.size, n), dtype=np.uint16)
    arr[:, 0] = pats.astype(np.uint16)
    arr[:, 1] = above
    arr[:, 5] = np.where(above != 0, 0x8000 | above, 0)   # a negative companion so that min/max formats see a span
    arr = arr.reshape(-1)
    check(oracle, ty, gpu_quantize(torch_cuda, ty, arr, "bf16"), arr, "bf16")


@pytest.mark.parametrize("ty", QUANT)
@pytest.mark.parametrize("kind", ["f32", "bf16"])
def test_blocks_at_the_ends_of_the_ordinary_range(torch_cuda, oracle, ty, kind):
    """The encoders take short division sequences when every block of a wave has its magnitude (largest element, or
    max - min) within 2^-100 ..= 2^100, and the general division otherwise.  Blocks scaled to either side of both
    limits, a few exactly on them, alone in their wave and mixed with ordinary blocks, must give the reference's
    bytes on both paths.  (f16 input cannot reach the limits.)"""
    n = oracle.block_elems(ty)
    rng = np.random.default_rng(23 + ty)
    per_wave = 64 * (4 if kind == "f32" else 8) // n or 1          # blocks one wave encodes per load
    rows = []

    def block(scale):
        b = rng.standard_normal(n).astype(np.float32)
        b /= np.abs(b).max()
        if ty in (3, 7):                                             # min/max formats: the range is on max - min
            b = (b - b.min()) / np.float32(b.max() - b.min())
        return (b * np.float32(scale)).astype(np.float32)

    scales = [2.0 ** e for e in (-104, -101, -100, -99, -96, 96, 99, 100, 101, 104)]
    scales += [float(np.nextafter(np.float32(2.0 ** -100), np.float32(0))), float(np.nextafter(np.float32(2.0 ** 100), np.float32(np.inf)))]
    for s in scales:                                                 # a whole wave (and more) of one scale
        rows += [block(s) for _ in range(2 * per_wave + 1)]
    for s in scales:                                                 # single extreme blocks among ordinary ones
        rows += [block(1.0) for _ in range(per_wave)] + [block(s)] + [block(0.01) for _ in range(3)]
    x = np.concatenate(rows)
    arr = to_in(oracle, x, kind)
    check(oracle, ty, gpu_quantize(torch_cuda, ty, arr, kind), arr, kind)


@pytest.mark.parametrize("ty", [2, 3, 6, 7, 8, 15])
def test_degenerate_blocks(torch_cuda, oracle, ty):
    """The blocks for which the reference leaves its general path -- all zero (`max == 0`), all equal (`min == max`)
    -- mixed with NaN elements, made of infinities, or of -0, and blocks that hold an infinity among finite values:
    the GPU encoders have no branch for most of these cases and must still produce the reference's bytes.  (Q8_1 is
    left out as in the NaN test: its `sum` header becomes 0 * inf, whose NaN differs between x86 and the GPU.)"""
    n = oracle.block_elems(ty)
    rng = np.random.default_rng(13)
    rows = []
    z = np.zeros(n, dtype=np.float32); z[2] = np.nan; z[n - 1] = np.nan; rows.append(z)                # zeros and NaN
    z = np.full(n, -0.0, dtype=np.float32); z[0] = np.nan; rows.append(z)
    rows.append(np.full(n, np.nan, dtype=np.float32))
    f = np.full(n, 0.4375, dtype=np.float32); f[5] = np.nan; f[6] = -np.nan; rows.append(f)          # flat and NaN
    rows.append(np.full(n, np.inf, dtype=np.float32))
    rows.append(np.full(n, -np.inf, dtype=np.float32))
    i = np.full(n, np.inf, dtype=np.float32); i[3] = np.nan; rows.append(i)
    r = (rng.standard_normal(n) * 0.1).astype(np.float32); r[7] = np.inf; rows.append(r)              # one infinity
    r = (rng.standard_normal(n) * 0.1).astype(np.float32); r[0] = -np.inf; r[9] = np.inf; rows.append(r)
    z = np.zeros(n, dtype=np.float32); z[1::2] = -0.0; z[4] = np.nan; rows.append(z)
    rows.append(np.full(n, 1e-45, dtype=np.float32))                                                  # flat subnormal
    x = np.concatenate(rows)
    check(oracle, ty, gpu_quantize(torch_cuda, ty, x, "f32"), x, "f32")


@pytest.mark.parametrize("ty,kind", [(1, "f32"), (30, "f32"), (30, "f16"), (1, "bf16"), (1, "f16"), (30, "bf16")])
def test_narrowing(torch_cuda, oracle, ty, kind):
    """f16::from_f32 / bf16::from_f32 per element (half.rs:10-17, 26-33), incl. ties and specials."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(70001) * np.float32(10.0) ** rng.integers(-9, 6, 70001)).astype(np.float32)
    h = np.arange(0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mid = ((h[:-1] + h[1:]) / 2).astype(np.float32)
    x = np.concatenate([x, mid, -mid, np.array([0, -0.0, 65504, 65520, 1e-8, np.inf, -np.inf, 1e38], dtype=np.float32)])
    arr = to_in(oracle, x, kind)
    got = gpu_quantize(torch_cuda, ty, arr, kind).view(np.uint16)
    want = oracle.quantize(ty, arr, in_ty=IN[kind]).view(np.uint16)
    assert np.array_equal(got, want)


def test_slice_errors(torch_cuda):
    """lib.rs:293-311 through the device entry point."""
    import gguf_amd as g
    t = torch_cuda
    src = t.zeros(64, dtype=t.float32, device="cuda")
    dst = t.zeros(34 * 3, dtype=t.uint8, device="cuda")
    with pytest.raises(g.Indivisible):
        g.Q8_0.quantize_slice(dst[:34], src[:31])
    with pytest.raises(g.LengthMismatch):
        g.Q8_0.quantize_slice(dst, src)
    with pytest.raises(g.Indivisible):
        g.Q8_0.quantize_slice(dst, src[:31])
    with pytest.raises(g.Unsupported):
        g.Q4K.quantize_slice(t.zeros(144, dtype=t.uint8, device="cuda"), t.zeros(256, dtype=t.float32, device="cuda"))
    g.Q8_0.quantize_slice(dst[:68], src)           # zeros -> ZEROS blocks (lib.rs:333-338)
    t.cuda.synchronize()
    assert not dst[:68].cpu().numpy().any()


@pytest.mark.parametrize("ty", [8, 2, 15])
def test_host_entry_point(torch_cuda, oracle, ty):
    import gguf_amd as g
    n = oracle.block_elems(ty)
    blocks = (9 << 20) // n + 5
    x = (np.random.default_rng(5).standard_normal(blocks * n) * 0.02).astype(np.float32)
    hold = np.zeros(x.size + 1, dtype=np.float32)[1:]   # 4-byte, not 16-byte, aligned
    hold[:] = x
    dst = np.zeros(blocks * oracle.block_bytes(ty) + 1, dtype=np.uint8)
    g.quantize_slice(ty, dst[1:], hold)
    assert np.array_equal(dst[1:], oracle.quantize(ty, x))


def test_tinyllama_f16_to_q8_0_round_trip(torch_cuda, oracle):
    """The benchmark's CPU-runnable configuration on the GPU: F16 tensor -> Q8_0 -> F32, on one TinyLlama-shaped
    tensor (attn_output [2048, 2048]) with seeded F16 payload (the shipped file has none)."""
    import gguf_amd as g
    t = torch_cuda
    rng = np.random.default_rng(0xC0FFEE01)
    w16 = (rng.standard_normal(2048 * 2048) * 0.02).astype(np.float16)
    src = t.from_numpy(w16.view(np.int16)).cuda().view(t.float16)
    q = t.empty(w16.size // 32 * 34, dtype=t.uint8, device="cuda")
    g.Q8_0.quantize_slice(q, src)
    out = t.empty(w16.size, dtype=t.float32, device="cuda")
    g.Q8_0.dequantize_slice(out, q)
    t.cuda.synchronize()
    want_q = oracle.quantize(8, w16.view(np.uint16), in_ty=oracle.F16)
    assert np.array_equal(q.cpu().numpy(), want_q)
    want = oracle.dequantize(8, want_q, threads=8)
    assert same_bits(out.cpu().numpy(), want)
    assert float(np.max(np.abs(want - w16.astype(np.float32)))) < 4.5e-3


@pytest.mark.parametrize("ty", QUANT)
@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_large_tensor_every_block(torch_cuda, oracle, ty, kind):
    """32 Mi elements of N(0, 0.02^2) weights (a [4096, 8192] matrix) per encoder and input type: every packed
    byte against the oracle's quantizer."""
    import gguf_amd as g
    t = torch_cuda
    n = 32 << 20
    gen = t.Generator(device="cuda")
    gen.manual_seed(4000 + ty)
    x = t.randn(n, device="cuda", generator=gen) * 0.02
    if kind != "f32":
        x = x.to(t.float16 if kind == "f16" else t.bfloat16)
    q = t.empty(n // g.block_elems(ty) * g.block_bytes(ty), dtype=t.uint8, device="cuda")
    g.quantize_slice(ty, q, x)
    t.cuda.synchronize()
    host = x.cpu().numpy() if kind == "f32" else x.view(t.int16).cpu().numpy().view(np.uint16)
    want = oracle.quantize(ty, host, in_ty=IN[kind])
    got = q.cpu().numpy()
    if not np.array_equal(got, want):
        bad = np.nonzero(got != want)[0]
        raise AssertionError(f"type {ty} from {kind}: {bad.size} bytes differ, first at {int(bad[0])}")


@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_quantize_plan_mixed(torch_cuda, oracle, kind):
    """ggq_quant_batch_*: many tensors, every encoder plus the F16 / BF16 narrowing, in one plan: whole tiles, short
    last tiles, single blocks, empty tensors, and tensors that follow each other in memory (one run).  Every packed
    byte against the oracle; nothing written past a tensor's end."""
    import gguf_amd as g
    t = torch_cuda
    rng = np.random.default_rng(77)
    dt = {"f32": t.float32, "f16": t.float16, "bf16": t.bfloat16}[kind]
    tile = 4096 if kind == "f32" else 8192
    spec = [(8, 3 * tile + 32), (2, tile), (2, 32), (8, 0), (3, 5 * tile - 64), (6, 2 * tile + 96), (7, 64), (9, tile + 32),
            (15, 256 * 37), (1, 1000), (30, 8 * 513), (8, 32 * 1001), (15, 256), (2, 7 * tile + 32 * 5), (1, 24), (9, 32)]
    tensors, inputs = [], []
    for i, (ty, n) in enumerate(spec):
        x = (rng.standard_normal(n) * (0.02 if i % 3 else 3.0)).astype(np.float32)
        if n >= 64:
            x[5] = 0.0
            x[33:40] = x[32]          # ties
        arr = to_in(oracle, x, kind)
        src = t.from_numpy(arr).cuda() if kind == "f32" else t.from_numpy(arr.view(np.int16)).cuda().view(dt)
        nb = n // g.block_elems(ty) * g.block_bytes(ty)
        dst = t.full((nb + 16,), 0xAB, dtype=t.uint8, device="cuda")
        tensors.append((ty, src, dst[:nb]))
        inputs.append((arr, dst))
    # two tensors of one type back to back in one element buffer and one packed buffer: a single run
    xa = (rng.standard_normal(32 * 300) * 0.5).astype(np.float32)
    run_arr = to_in(oracle, xa, kind)
    both = t.from_numpy(run_arr).cuda() if kind == "f32" else t.from_numpy(run_arr.view(np.int16)).cuda().view(dt)
    packed = t.full((300 * 34 + 16,), 0xAB, dtype=t.uint8, device="cuda")
    tensors += [(8, both[:32 * 100], packed[:100 * 34]), (8, both[32 * 100:], packed[100 * 34:300 * 34])]
    b = g.QuantBatch(tensors)
    # one launch per block type in the plan, one per run of F16 / BF16 targets
    assert b.launches == len({ty for ty, n in spec if n and ty not in (1, 30)}) + sum(1 for ty, n in spec if ty in (1, 30))
    for _ in range(2):
        b.run()
    t.cuda.synchronize()
    for (ty, n), (arr, dst) in zip(spec, inputs):
        nb = n // g.block_elems(ty) * g.block_bytes(ty)
        out = dst.cpu().numpy()
        assert (out[nb:] == 0xAB).all(), ("wrote past the end", ty, n)
        check(oracle, ty, out[:nb], arr, kind)
    out = packed.cpu().numpy()
    assert (out[300 * 34:] == 0xAB).all()
    check(oracle, 8, out[:300 * 34], run_arr, kind)
    b.close()


def test_quantize_plan_errors(torch_cuda):
    import gguf_amd as g
    t = torch_cuda
    x = t.zeros(256, device="cuda")
    q = t.zeros(144, dtype=t.uint8, device="cuda")
    with pytest.raises(g.Unsupported):
        g.QuantBatch([(12, x, q)])                                   # Q4_K has no quantizer in the reference
    with pytest.raises(g.LengthMismatch):
        g.QuantBatch([(8, x, t.zeros(34 * 7, dtype=t.uint8, device="cuda"))])
    with pytest.raises(g.Indivisible):
        g.QuantBatch([(8, x[:31], t.zeros(34, dtype=t.uint8, device="cuda"))])
    with pytest.raises(g.Misaligned):
        g.QuantBatch([(8, x[1:33], t.zeros(34, dtype=t.uint8, device="cuda"))])
    with pytest.raises(ValueError):
        g.QuantBatch([(8, x, t.zeros(34 * 8, dtype=t.uint8, device="cuda")),
                      (8, x.half(), t.zeros(34 * 8, dtype=t.uint8, device="cuda"))])
    g.QuantBatch([], g.GGmlType.F32).close()                          # an empty plan is fine
