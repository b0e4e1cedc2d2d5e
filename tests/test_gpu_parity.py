"""Parity of the HIP kernels with the CPU oracle, through the C ABI.

Bar: bit-exact for every finite result (the unpack is integer work and every float
step is a single IEEE operation on both sides); a NaN must be a NaN on both sides
(payloads are not compared: x86 and the GPU pick different default NaNs for inf*0).
f16 / bf16 outputs: bit-exact as well (tolerance 0 ULP, tighter than the 1 ULP
the target statement allows).
"""
import os

import numpy as np
import pytest

import hand_kats
from util import same_bits, same_bits16, first_diff

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
BLOCK_TYPES = [2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 20, 23]
ALL_TYPES = BLOCK_TYPES + [1, 30]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def gpu_decode(torch, ty, packed, out="f32", offset=0):
    """Decode on the device; `offset` shifts the source pointer to test misaligned inputs."""
    import gguf_amd as g
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[out]
    n = packed.size // g.block_bytes(ty) * g.block_elems(ty)
    buf = torch.empty(packed.size + offset + 64, dtype=torch.uint8, device="cuda")
    src = buf[offset:offset + packed.size]
    src.copy_(torch.from_numpy(packed))
    dst = torch.empty(n, dtype=dt, device="cuda")
    g.dequantize_slice(ty, dst, src)
    torch.cuda.synchronize()
    if out == "f32":
        return dst.cpu().numpy()
    return dst.view(torch.int16).cpu().numpy().view(np.uint16)


def check(oracle, ty, packed, got, out):
    o_out = {"f32": oracle.F32, "f16": oracle.F16, "bf16": oracle.BF16}[out]
    want = oracle.dequantize(ty, packed, o_out, threads=8)
    if out == "f32":
        assert same_bits(got, want), first_diff(got, want)
    else:
        assert same_bits16(got, want, out), first_diff(got, want)


@pytest.mark.parametrize("ty", ALL_TYPES)
def test_seeded_golden(torch_cuda, oracle, ty):
    s = np.load(os.path.join(G, "seeded.npz"))
    p = s[f"t{ty}_packed"]
    assert same_bits(gpu_decode(torch_cuda, ty, p), s[f"t{ty}_f32"])
    assert same_bits16(gpu_decode(torch_cuda, ty, p, "f16"), s[f"t{ty}_f16"], "f16")
    assert same_bits16(gpu_decode(torch_cuda, ty, p, "bf16"), s[f"t{ty}_bf16"], "bf16")


@pytest.mark.parametrize("ty", BLOCK_TYPES)
def test_kat_and_edge_scales(torch_cuda, oracle, ty):
    k = np.load(os.path.join(G, "kat.npz"))
    got = gpu_decode(torch_cuda, ty, k[f"t{ty}_packed"])
    assert same_bits(got, k[f"t{ty}_f32"]), first_diff(got, k[f"t{ty}_f32"])
    e = np.load(os.path.join(G, "edges.npz"))
    got = gpu_decode(torch_cuda, ty, e[f"t{ty}_packed"])
    assert same_bits(got, e[f"t{ty}_f32"]), first_diff(got, e[f"t{ty}_f32"])
    got = gpu_decode(torch_cuda, ty, e[f"t{ty}_packed"], "f16")
    assert same_bits16(got, e[f"t{ty}_f16"], "f16")


@pytest.mark.parametrize("case", hand_kats.cases(), ids=lambda c: f"type{c[0]}")
def test_hand_derived_blocks(torch_cuda, case):
    """Expected values typed in by hand (tests/hand_kats.py), not produced by any decoder; bit for bit,
    through the single-tensor kernel and, repeated over several tiles, through the batch kernel."""
    import gguf_amd as g
    t = torch_cuda
    ty, blk, want = case
    got = gpu_decode(t, ty, blk)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), first_diff(got, want)
    got16 = gpu_decode(t, ty, blk, "f16")
    assert np.array_equal(got16, want.astype(np.float16).view(np.uint16))
    reps = 3 * 8192 // g.block_elems(ty) + 1
    src = t.from_numpy(np.tile(blk, reps)).cuda()
    dst = t.empty(reps * want.size, dtype=t.float32, device="cuda")
    b = g.Batch([(ty, src, dst)])
    b.run()
    t.cuda.synchronize()
    got = dst.cpu().numpy().reshape(reps, want.size)
    assert np.array_equal(got.view(np.uint32), np.tile(want.view(np.uint32), (reps, 1)))
    b.close()


# block counts around the tile sizes (a workgroup tile is 4096 elements for f32
# output and 8192 for 2-byte outputs), one block, and many tiles
def counts_for(n):
    tb = 4096 // n
    return [1, 2, tb - 1, tb, tb + 1, 2 * tb - 1, 2 * tb, 2 * tb + 1, 3 * tb + 5, 40 * tb + 7]


@pytest.mark.parametrize("ty", BLOCK_TYPES)
@pytest.mark.parametrize("out", ["f32", "f16", "bf16"])
def test_ragged_sizes(torch_cuda, oracle, ty, out):
    from gguf_amd import synth
    for blocks in counts_for(oracle.block_elems(ty)):
        p = synth.random_blocks(ty, blocks, 9000 + blocks)
        check(oracle, ty, p, gpu_decode(torch_cuda, ty, p, out), out)


@pytest.mark.parametrize("ty", BLOCK_TYPES)
def test_adversarial_bytes(torch_cuda, oracle, ty):
    from gguf_amd import synth
    p = synth.random_blocks(ty, 3000, 31337 + ty, adversarial=True)
    for out in ("f32", "f16", "bf16"):
        check(oracle, ty, p, gpu_decode(torch_cuda, ty, p, out), out)


@pytest.mark.parametrize("ty", BLOCK_TYPES)
@pytest.mark.parametrize("offset", [2, 4, 6, 8, 10, 1, 15])
def test_source_alignment(torch_cuda, oracle, ty, offset):
    """Blocks are only 2-byte aligned in the reference (f16 header); any byte offset must decode."""
    from gguf_amd import synth
    blocks = 5 * (4096 // oracle.block_elems(ty)) + 3
    p = synth.random_blocks(ty, blocks, 555 + offset)
    check(oracle, ty, p, gpu_decode(torch_cuda, ty, p, "f32", offset), "f32")


@pytest.mark.parametrize("ty", [1, 30])
@pytest.mark.parametrize("out", ["f32", "f16", "bf16"])
def test_widen_all_patterns(torch_cuda, oracle, ty, out):
    bits = np.arange(65536, dtype=np.uint16)
    for n, off in ((65536, 0), (65533, 0), (1001, 2), (7, 0), (65536, 6)):
        p = bits[:n].view(np.uint8)
        check(oracle, ty, p, gpu_decode(torch_cuda, ty, p, out, off), out)


@pytest.mark.parametrize("ty", [12, 14, 2, 8])
@pytest.mark.parametrize("out", ["f32", "f16"])
def test_beyond_prefetch_distance(torch_cuda, oracle, ty, out):
    """More than 2048 tiles in one tensor, so workgroups touch the tile 2048 ahead
    (the L2 prefetch), with a ragged tail and an odd source alignment."""
    from gguf_amd import synth
    n = oracle.block_elems(ty)
    tile = 4096 if out == "f32" else 8192
    blocks = (2048 + 37) * tile // n + 3
    p = synth.random_blocks(ty, blocks, 2048 + ty)
    check(oracle, ty, p, gpu_decode(torch_cuda, ty, p, out, 6), out)


def test_slice_errors_on_device(torch_cuda):
    """lib.rs:293-338 through the device entry point."""
    import gguf_amd as g
    t = torch_cuda
    src = t.zeros(34 * 3, dtype=t.uint8, device="cuda")
    dst = t.zeros(96, dtype=t.float32, device="cuda")
    with pytest.raises(g.Indivisible):
        g.Q8_0.dequantize_slice(dst[:31], src[:34])
    with pytest.raises(g.LengthMismatch):
        g.Q8_0.dequantize_slice(dst[:64], src)
    with pytest.raises(g.Indivisible):
        g.Q8_0.dequantize_slice(dst[:31], src)
    with pytest.raises(g.Misaligned):
        g.Q8_0.dequantize_slice(dst[1:33], src[:34])
    dst.fill_(7.0)
    g.Q8_0.dequantize_slice(dst, src)          # zero blocks -> exact +0.0
    t.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), np.zeros(96, dtype=np.uint32))


@pytest.mark.parametrize("ty", [8, 2, 12, 14, 1])
def test_host_buffer_entry_point(torch_cuda, oracle, ty):
    """ggq_dequantize_slice: host pointers in and out, odd host alignment, several chunks."""
    import gguf_amd as g
    from gguf_amd import synth
    n = oracle.block_elems(ty)
    blocks = (9 << 20) // n + 3            # > 2 staged chunks of 4 Mi elements
    p = synth.random_blocks(ty, blocks, 4242)
    hold = np.zeros(p.size + 3, dtype=np.uint8)
    hold[3:] = p                            # misaligned host source
    dst = np.full(blocks * n + 1, 9.0, dtype=np.float32)
    g.dequantize_slice(ty, dst[1:], hold[3:])
    want = oracle.dequantize(ty, p, threads=8)
    assert same_bits(dst[1:], want), first_diff(dst[1:], want)
    assert dst[0] == 9.0


def test_batch_matches_single(torch_cuda, oracle):
    import gguf_amd as g
    from gguf_amd import synth
    t = torch_cuda
    # > 2048 tiles of Q4_K in total, with short tiles in between (prefetch must skip those)
    specs = [(12, 700), (14, 33), (12, 1), (2, 5000), (14, 260), (12, 16), (8, 129), (1, 5001), (23, 40),
             (12, 16 * 1500 + 5), (12, 16 * 900), (12, 7), (12, 16 * 300 + 1)]
    tensors, packed = [], []
    for i, (ty, blocks) in enumerate(specs):
        p = synth.random_blocks(ty, blocks, 800 + i)
        pad = (i * 2) % 16                  # different source alignments inside one launch
        buf = t.empty(p.size + pad, dtype=t.uint8, device="cuda")
        src = buf[pad:]
        src.copy_(t.from_numpy(p))
        dst = t.empty(blocks * oracle.block_elems(ty), dtype=t.float32, device="cuda")
        tensors.append((ty, src, dst))
        packed.append(p)
    b = g.Batch(tensors)
    assert b.launches == 6                  # one per block type present (Q4K, Q6K, Q4_0, Q8_0, IQ4XS) + the f16 tensor
    for _ in range(2):
        for _, _, d in tensors:
            d.fill_(-1.0)
        b.run()
        t.cuda.synchronize()
        for (ty, _, d), p in zip(tensors, packed):
            want = oracle.dequantize(ty, p, threads=8)
            assert same_bits(d.cpu().numpy(), want), (ty, first_diff(d.cpu().numpy(), want))
    b.close()


def _batch_case(t, oracle, ty, sizes, seed, out_dt, offsets=True):
    """Builds tensors of `sizes` blocks of one type on the device; returns (tensor list, packed list)."""
    from gguf_amd import synth
    tensors, packed = [], []
    for i, blocks in enumerate(sizes):
        p = synth.random_blocks(ty, blocks, seed + i)
        pad = (2 * i + 2 * (i // 8)) % 16 if offsets else 0       # different source alignments inside one launch
        buf = t.empty(p.size + pad, dtype=t.uint8, device="cuda")
        src = buf[pad:]
        src.copy_(t.from_numpy(p))
        dst = t.full((blocks * oracle.block_elems(ty),), -1.0, dtype=out_dt, device="cuda")
        tensors.append((ty, src, dst))
        packed.append(p)
    return tensors, packed


@pytest.mark.parametrize("ty", BLOCK_TYPES)
@pytest.mark.parametrize("out", ["f32", "f16", "bf16"])
def test_batch_every_format_and_output(torch_cuda, oracle, ty, out):
    """The batch kernel (the one the benchmark times) for every block type and output type: more tiles
    than lead + period of every burst-prefetch schedule (<= 2048 + 1536), so duty workgroups fire and look
    far tiles up in the table; short tiles (a tensor's last) between whole ones, which the prefetch must
    skip; mixed source alignments."""
    import gguf_amd as g
    t = torch_cuda
    n = oracle.block_elems(ty)
    tile = (4096 if out == "f32" else 8192) // n            # blocks per tile
    sizes = [1500 * tile + 5, 1, 900 * tile, 7, tile - 1, 1300 * tile + 1, tile + 1, 33, 400 * tile, 2 * tile]
    out_dt = {"f32": t.float32, "f16": t.float16, "bf16": t.bfloat16}[out]
    o_out = {"f32": oracle.F32, "f16": oracle.F16, "bf16": oracle.BF16}[out]
    tensors, packed = _batch_case(t, oracle, ty, sizes, 8100 + ty, out_dt)
    b = g.Batch(tensors, {"f32": g.GGmlType.F32, "f16": g.GGmlType.F16, "bf16": g.GGmlType.BF16}[out])
    assert b.launches == 1 and b.table_launches == 1        # ten separate tensors: the tile-table kernel
    b.run()
    t.cuda.synchronize()
    for (_, _, d), p in zip(tensors, packed):
        want = oracle.dequantize(ty, p, o_out, threads=8)
        if out == "f32":
            got = d.cpu().numpy()
            assert same_bits(got, want), (ty, p.size, first_diff(got, want))
        else:
            got = d.view(t.int16).cpu().numpy().view(np.uint16)
            assert same_bits16(got, want, out), (ty, p.size, first_diff(got, want))
    b.close()


@pytest.mark.parametrize("out", ["f32", "f16", "bf16"])
def test_batch_all_types_in_one_plan(torch_cuda, oracle, out):
    """One plan holding every decodable type (14 block types + F16 + BF16 sources), interleaved."""
    import gguf_amd as g
    from gguf_amd import synth
    t = torch_cuda
    out_dt = {"f32": t.float32, "f16": t.float16, "bf16": t.bfloat16}[out]
    o_out = {"f32": oracle.F32, "f16": oracle.F16, "bf16": oracle.BF16}[out]
    tensors, packed = [], []
    for rnd in range(3):
        for ty in ALL_TYPES:
            n = oracle.block_elems(ty)
            blocks = (3 + rnd) * 8192 // n + (rnd * 5 + ty) % 7 if n > 1 else 50001 + 4 * rnd + ty
            ts, ps = _batch_case(t, oracle, ty, [blocks], 8300 + 31 * rnd + ty, out_dt, offsets=(n > 1))
            tensors += ts
            packed += ps
    b = g.Batch(tensors, {"f32": g.GGmlType.F32, "f16": g.GGmlType.F16, "bf16": g.GGmlType.BF16}[out])
    assert b.launches == len(BLOCK_TYPES) + 6            # one per block type + one per F16 / BF16 tensor
    for _ in range(2):
        for _, _, d in tensors:
            d.fill_(-1.0)
        b.run()
        t.cuda.synchronize()
        for (ty, _, d), p in zip(tensors, packed):
            want = oracle.dequantize(ty, p, o_out, threads=8)
            if out == "f32":
                assert same_bits(d.cpu().numpy(), want), (ty, first_diff(d.cpu().numpy(), want))
            else:
                got = d.view(t.int16).cpu().numpy().view(np.uint16)
                assert same_bits16(got, want, out), (ty, first_diff(got, want))
    b.close()


def test_streamed_context(torch_cuda, oracle):
    import gguf_amd as g
    from gguf_amd import synth
    t = torch_cuda
    ctx = g.Context(0, staging_bytes=2 << 20)     # small staging -> many chunks
    work = []
    for i, (ty, blocks) in enumerate([(12, 20000), (14, 9000), (2, 70000), (30, 300001)]):
        p = synth.random_blocks(ty, blocks, 600 + i)
        dst = t.empty(blocks * oracle.block_elems(ty), dtype=t.float32, device="cuda")
        ctx.dequantize_async(ty, dst, p)
        work.append((ty, p, dst))
    ctx.sync()
    for ty, p, dst in work:
        want = oracle.dequantize(ty, p, threads=8)
        assert same_bits(dst.cpu().numpy(), want), (ty, first_diff(dst.cpu().numpy(), want))
    ctx.close()


def test_q4_0_4096_square(torch_cuda, oracle):
    """Benchmark configuration: a single 4096 x 4096 Q4_0 tensor, packed by the reference's quantizer rule."""
    rng = np.random.default_rng(0xC0FFEE02)
    x = (rng.standard_normal(4096 * 4096) * 0.02).astype(np.float32)
    p = oracle.quantize(2, x)
    assert p.size == 9_437_184
    got = gpu_decode(torch_cuda, 2, p)
    want = oracle.dequantize(2, p, threads=8)
    assert same_bits(got, want)
    assert float(np.max(np.abs(got - x))) < 8e-2        # the reference's own tolerance (q4_0.rs:62)


def test_full_size_properties(torch_cuda, oracle):
    """At benchmark size the oracle is too slow to decode everything; check instead
    (a) a sampled set of blocks bit-for-bit, (b) the decode is a pure function of the
    block: decoding a permutation of the blocks gives the permuted output."""
    import gguf_amd as g
    from gguf_amd import synth
    t = torch_cuda
    for ty in (12, 14):
        bb, n = g.block_bytes(ty), g.block_elems(ty)
        blocks = 4096 * 11008 // n                      # one 7B FFN matrix
        src = synth.random_blocks_device(ty, blocks, 77, "cuda")
        dst = t.empty(blocks * n, dtype=t.float32, device="cuda")
        g.dequantize_slice(ty, dst, src)
        perm = t.randperm(blocks, device="cuda")
        src2 = src.view(blocks, bb)[perm].contiguous().view(-1)
        dst2 = t.empty_like(dst)
        g.dequantize_slice(ty, dst2, src2)
        t.cuda.synchronize()
        assert t.equal(dst.view(blocks, n)[perm].view(t.int32), dst2.view(blocks, n).view(t.int32))
        idx = t.randint(0, blocks, (4096,), device="cuda")
        sample = src.view(blocks, bb)[idx].cpu().numpy().reshape(-1)
        want = oracle.dequantize(ty, sample, threads=8)
        got = dst.view(blocks, n)[idx].cpu().numpy().reshape(-1)
        assert same_bits(got, want)


def test_batch_plan_replayed_from_a_graph(torch_cuda, oracle):
    """A launch plan is plain kernel launches on the caller's stream, so it can be captured into a HIP graph;
    replays, with the inputs rewritten and eager work in between, must each give the oracle's bits."""
    import gguf_amd as g
    from gguf_amd import synth
    t = torch_cuda
    specs = [(12, 16 * 40 + 3), (14, 300), (2, 5000), (20, 999), (1, 70001)]
    srcs = [t.empty(b * g.block_bytes(ty), dtype=t.uint8, device="cuda") for ty, b in specs]
    dsts = [t.empty(b * g.block_elems(ty), dtype=t.float16, device="cuda") for ty, b in specs]
    plan = g.Batch([(ty, s, d) for (ty, _), s, d in zip(specs, srcs, dsts)], g.GGmlType.F16)
    graph = t.cuda.CUDAGraph()
    with t.cuda.graph(graph, capture_error_mode="thread_local"):
        plan.run(t.cuda.current_stream().cuda_stream)
    for rep in range(3):
        packed = [synth.random_blocks(ty, b, 7000 + 10 * rep + i) for i, (ty, b) in enumerate(specs)]
        for s, p in zip(srcs, packed):
            s.copy_(t.from_numpy(p))
        for d in dsts:
            d.fill_(-1.0)
        t.cuda.synchronize()
        graph.replay()
        t.cuda.synchronize()
        for (ty, _), d, p in zip(specs, dsts, packed):
            want = oracle.dequantize(ty, p, oracle.F16, threads=4)
            got = d.view(t.int16).cpu().numpy().view(np.uint16)
            assert same_bits16(got, want, "f16"), (rep, ty, first_diff(got, want))
    plan.close()


@pytest.mark.parametrize("ty,out", [(8, "f32"), (14, "f16"), (12, "f16"), (13, "bf16"), (9, "f32"), (12, "f32")])
def test_result_does_not_depend_on_what_the_destination_held(torch_cuda, oracle, ty, out):
    """Some kernels load a word of the destination before they overwrite it (the group start).  Whatever the
    destination held -- zeros, all ones, NaN patterns -- the result is the oracle's."""
    import gguf_amd as g
    from gguf_amd import synth
    t = torch_cuda
    # 67 Mi elements: from 64 Mi on the single-tensor kernels of these formats use the group start
    blocks = (67 << 20) // g.block_elems(ty)
    packed = synth.random_blocks(ty, blocks, 4242 + ty)
    src = t.from_numpy(packed).cuda()
    dt = {"f32": t.float32, "f16": t.float16, "bf16": t.bfloat16}[out]
    o_out = {"f32": oracle.F32, "f16": oracle.F16, "bf16": oracle.BF16}[out]
    want = oracle.dequantize(ty, packed, o_out, threads=16)
    for fill in (0x00, 0xFF, 0x7F):
        dst = t.empty(blocks * g.block_elems(ty), dtype=dt, device="cuda")
        dst.view(t.uint8).fill_(fill)
        g.dequantize_slice(ty, dst, src)
        t.cuda.synchronize()
        if out == "f32":
            assert same_bits(dst.cpu().numpy(), want), (ty, out, fill)
        else:
            assert same_bits16(dst.view(t.int16).cpu().numpy().view(np.uint16), want, out), (ty, out, fill)
