"""ggq_dequantize_noalloc: the entry point that allocates nothing (fixed device staging array).

Checked through the C ABI against ggq_dequantize_async (the entry point it is a variant of) and
against the CPU oracle, bit for bit, on calls in a row that share the staging array -- the later
ones smaller than the first, so stale bytes of an earlier call lie behind the new input.
"""
import ctypes

import numpy as np
import pytest

from util import same_bits, same_bits16, first_diff

pytestmark = pytest.mark.gpu

MAX_BYTES = 32 << 20


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def pinned_blocks(torch, packed):
    h = torch.empty(packed.size, dtype=torch.uint8).pin_memory()
    h.numpy()[:] = packed
    return h


def run_noalloc(torch, ty, out_dt, h_src, blocks, stream=None):
    import gguf_amd as g
    from gguf_amd._lib import lib
    dst = torch.full((blocks * g.block_elems(ty),), -3.0, dtype=out_dt, device="cuda")
    out_ty = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 30}[out_dt]
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    rc = lib.ggq_dequantize_noalloc(int(ty), out_ty, dst.data_ptr(), dst.numel(), h_src.data_ptr(), blocks, st)
    return rc, dst


def run_async(torch, ctx, ty, out_dt, h_src, blocks):
    import gguf_amd as g
    dst = torch.full((blocks * g.block_elems(ty),), -5.0, dtype=out_dt, device="cuda")
    ctx.dequantize_async(ty, dst, h_src.numpy()[: blocks * g.block_bytes(ty)])
    ctx.sync()
    return dst


def bits(torch, t):
    return t.view(torch.int32 if t.element_size() == 4 else torch.int16).cpu().numpy()


# (type, blocks of the first call, blocks of the second, smaller call)
CASES = [(12, 70001, 1037), (14, 40000, 17), (2, 300007, 5000), (8, 100000, 129), (10, 9000, 8999),
         (11, 5000, 1), (13, 6000, 33), (3, 4100, 127), (6, 4100, 128), (7, 4100, 129), (9, 999, 3),
         (15, 3000, 15), (20, 123457, 1000), (23, 30000, 16), (1, 4000003, 70001), (30, 1000001, 5)]


@pytest.mark.parametrize("ty,first,second", CASES)
def test_matches_async_two_calls_in_a_row(torch_cuda, oracle, ty, first, second):
    import gguf_amd as g
    from gguf_amd import synth
    t = torch_cuda
    ctx = g.Context(0, staging_bytes=4 << 20)
    try:
        for call, blocks in enumerate((first, second)):
            p = synth.random_blocks(ty, blocks, 4100 + 7 * ty + call)
            h = pinned_blocks(t, p)
            for out_dt, kind in ((t.float32, "f32"), (t.float16, "f16"), (t.bfloat16, "bf16")):
                rc, got = run_noalloc(t, ty, out_dt, h, blocks)
                assert rc == 0, (rc, g._lib.lib.ggq_last_error())
                t.cuda.synchronize()
                want = run_async(t, ctx, ty, out_dt, h, blocks)
                assert np.array_equal(bits(t, got), bits(t, want)), (ty, kind, call)
            # and the oracle, f32
            rc, got = run_noalloc(t, ty, t.float32, h, blocks)
            assert rc == 0
            t.cuda.synchronize()
            ref = oracle.dequantize(ty, p, threads=8)
            assert same_bits(got.cpu().numpy(), ref), (ty, call, first_diff(got.cpu().numpy(), ref))
    finally:
        ctx.close()


def test_back_to_back_without_sync(torch_cuda, oracle):
    """Calls on one stream are ordered by the stream: no host synchronization between them."""
    from gguf_amd import synth
    t = torch_cuda
    work = []
    for i, (ty, blocks) in enumerate([(12, 50000), (14, 300), (12, 20), (2, 70000), (23, 7)]):
        p = synth.random_blocks(ty, blocks, 5200 + i)
        h = pinned_blocks(t, p)
        rc, dst = run_noalloc(t, ty, t.float32, h, blocks)
        assert rc == 0
        work.append((ty, p, h, dst))
    t.cuda.synchronize()
    for ty, p, _, dst in work:
        ref = oracle.dequantize(ty, p, threads=8)
        assert same_bits(dst.cpu().numpy(), ref), (ty, first_diff(dst.cpu().numpy(), ref))


def test_side_stream(torch_cuda, oracle):
    from gguf_amd import synth
    t = torch_cuda
    s = t.cuda.Stream()
    p = synth.random_blocks(13, 7777, 5300)
    h = pinned_blocks(t, p)
    with t.cuda.stream(s):
        rc, dst = run_noalloc(t, 13, t.float16, h, 7777, s.cuda_stream)
    assert rc == 0
    s.synchronize()
    ref = oracle.dequantize(13, p, oracle.F16, threads=8)
    assert same_bits16(dst.view(t.int16).cpu().numpy(), ref, "f16")


def test_limits_and_errors(torch_cuda):
    import gguf_amd as g
    from gguf_amd import _lib
    from gguf_amd._lib import lib
    t = torch_cuda
    # exactly at the limit: accepted (Q4_K: 144-byte blocks)
    blocks = MAX_BYTES // 144
    h = t.zeros(blocks * 144 + 144, dtype=t.uint8).pin_memory()
    rc, dst = run_noalloc(t, 12, t.float16, h, blocks)
    assert rc == _lib.OK
    t.cuda.synchronize()
    assert not bits(t, dst).any()            # zero blocks decode to +0.0
    # one block more: refused, nothing written
    dst = t.full(((blocks + 1) * 256,), -3.0, dtype=t.float16, device="cuda")
    rc = lib.ggq_dequantize_noalloc(12, 1, dst.data_ptr(), dst.numel(), h.data_ptr(), blocks + 1, None)
    assert rc == _lib.TOO_LARGE
    assert b"GGQ_NOALLOC_MAX_BYTES" in lib.ggq_last_error()
    t.cuda.synchronize()
    assert bool((dst == -3.0).all())
    with pytest.raises(ValueError):
        g.quants.check(rc)
    # pageable source: refused
    pageable = np.zeros(144 * 4, dtype=np.uint8)
    dst = t.empty(1024, dtype=t.float32, device="cuda")
    rc = lib.ggq_dequantize_noalloc(12, 0, dst.data_ptr(), 1024, pageable.ctypes.data, 4, None)
    assert rc == _lib.INVALID
    # the reference's check order, then alignment
    hp = t.zeros(144 * 4, dtype=t.uint8).pin_memory()
    assert lib.ggq_dequantize_noalloc(12, 0, dst.data_ptr(), 1000, hp.data_ptr(), 4, None) == _lib.INDIVISIBLE
    assert lib.ggq_dequantize_noalloc(12, 0, dst.data_ptr(), 1024, hp.data_ptr(), 3, None) == _lib.LENGTH_MISMATCH
    assert lib.ggq_dequantize_noalloc(12, 0, dst.data_ptr(), 1000, hp.data_ptr(), 3, None) == _lib.INDIVISIBLE
    assert lib.ggq_dequantize_noalloc(16, 0, dst.data_ptr(), 1024, hp.data_ptr(), 4, None) == _lib.UNSUPPORTED
    assert lib.ggq_dequantize_noalloc(12, 0, dst.data_ptr() + 4, 768, hp.data_ptr(), 3, None) == _lib.MISALIGNED
    assert lib.ggq_dequantize_noalloc(12, 0, dst.data_ptr(), 0, hp.data_ptr(), 0, None) == _lib.OK


def test_under_graph_capture(torch_cuda, oracle):
    """What the entry point is for: the call is captured into a HIP graph (a host-to-device copy node and a
    kernel node, nothing allocated) and replayed; the pinned source's CONTENT changes between replays, the
    program's own eager work and synchronizations sit between them, and every replay must give the oracle's
    bits."""
    from gguf_amd import synth
    from gguf_amd._lib import lib
    t = torch_cuda
    ty, blocks = 12, 30001
    h = t.empty(blocks * 144, dtype=t.uint8).pin_memory()
    dst = t.empty(blocks * 256, dtype=t.float32, device="cuda")
    # one-time per-device set-up outside the capture (an empty call does just that)
    assert lib.ggq_dequantize_noalloc(ty, 0, dst.data_ptr(), 0, h.data_ptr(), 0, None) == 0
    h.numpy()[:] = synth.random_blocks(ty, blocks, 6000)
    t.cuda.synchronize()
    graph = t.cuda.CUDAGraph()
    with t.cuda.graph(graph, capture_error_mode="thread_local"):
        rc = lib.ggq_dequantize_noalloc(ty, 0, dst.data_ptr(), dst.numel(), h.data_ptr(), blocks,
                                        t.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.ggq_last_error()
    for rep in range(4):
        p = synth.random_blocks(ty, blocks, 6001 + rep)
        h.numpy()[:] = p
        dst.fill_(-7.0)                                   # eager work between the replays
        t.cuda.synchronize()
        graph.replay()
        t.cuda.synchronize()
        want = oracle.dequantize(ty, p, threads=8)
        assert same_bits(dst.cpu().numpy(), want), (rep, first_diff(dst.cpu().numpy(), want))
        # an eager call of another type in between shares the staging array; the next replay must not care
        p2 = synth.random_blocks(14, 777, 6100 + rep)
        rc, d2 = run_noalloc(t, 14, t.float32, pinned_blocks(t, p2), 777)
        assert rc == 0
        t.cuda.synchronize()
        assert same_bits(d2.cpu().numpy(), oracle.dequantize(14, p2, threads=8))
