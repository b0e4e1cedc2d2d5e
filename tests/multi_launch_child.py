"""Run by test_gpu_multi_launch.py in a child process whose environment sets GGQ_MAX_GRID=3 (or leaves the grid
limit alone), so that every entry point that splits large inputs over several kernel launches does so on small
inputs, and GGQ_HYBRID_MIN_ELEMS=20000, so that launch plans send runs of adjacent tensors with 20000 elements and
more through launches of their own and the rest through the tile table, as they do for runs of 64 Mi elements by
default; GGQ_PREFETCH may force a schedule (with or without group start) onto every kernel.
Compares everything with the CPU oracle / numpy and prints OK."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch  # noqa: E402
import gguf_amd as g  # noqa: E402
import oracle as o  # noqa: E402
from gguf_amd import rearrange as R, synth  # noqa: E402
from test_gpu_rearrange import np_permute_qk  # noqa: E402
import util  # noqa: E402
from util import same_bits, same_bits16  # noqa: E402

MAX_GRID = int(os.environ.get("GGQ_MAX_GRID", str(1 << 22)))
assert MAX_GRID in (3, 1 << 22) and os.environ.get("GGQ_HYBRID_MIN_ELEMS") == "20000"
FORCED = [int(x) for x in os.environ["GGQ_PREFETCH"].split(",")] if os.environ.get("GGQ_PREFETCH") else None


def plan_entries(tensors, ob):
    return util.plan_entries(tensors, ob, forced=FORCED, min_elems=20000)


o.build()
t = torch


def bits16(x):
    return x.view(t.int16).cpu().numpy().view(np.uint16)


# single-tensor decode: 10 tiles + a ragged tail -> 4 launches of <= 3 workgroups
for ty in (12, 14, 2, 8, 10, 11, 13, 20, 23):
    n = g.block_elems(ty)
    for out_dt, o_out, tile in ((t.float32, o.F32, 4096), (t.float16, o.F16, 8192)):
        blocks = 10 * tile // n + 3
        p = synth.random_blocks(ty, blocks, 60 + ty)
        dst = t.empty(blocks * n, dtype=out_dt, device="cuda")
        g.dequantize_slice(ty, dst, t.from_numpy(p).cuda())
        t.cuda.synchronize()
        want = o.dequantize(ty, p, o_out, threads=4)
        ok = same_bits(dst.cpu().numpy(), want) if out_dt == t.float32 else same_bits16(bits16(dst), want, "f16")
        assert ok, ("single", ty, out_dt)

# the same single-tensor decode from a captured graph, replayed with new inputs in the same buffers (under a
# forced schedule with a group start, the kernel's probe reads a line of `dst` that the previous replay wrote; the
# consumer between the replays is an ordinary reduction over `dst`)
for ty, out_dt, o_out in ((12, t.float16, o.F16), (8, t.float32, o.F32), (14, t.bfloat16, o.BF16)):
    n = g.block_elems(ty)
    blocks = 9 * 8192 // n + 5
    src = t.empty(blocks * g.block_bytes(ty), dtype=t.uint8, device="cuda")
    dst = t.empty(blocks * n, dtype=out_dt, device="cuda")
    side = t.cuda.Stream()
    with t.cuda.stream(side):
        g.dequantize_slice(ty, dst, src)            # warm-up outside the capture
    side.synchronize()
    gr = t.cuda.CUDAGraph()
    with t.cuda.graph(gr):
        g.dequantize_slice(ty, dst, src)
        g.dequantize_slice(ty, dst, src)            # back to back into the same buffer
    for rep in range(3):
        p = synth.random_blocks(ty, blocks, 900 + 10 * ty + rep)
        src.copy_(t.from_numpy(p))
        gr.replay()
        t.cuda.synchronize()
        want = o.dequantize(ty, p, o_out, threads=4)
        ok = same_bits(dst.cpu().numpy(), want) if out_dt == t.float32 else \
            same_bits16(bits16(dst), want, "f16" if out_dt == t.float16 else "bf16")
        assert ok, ("graph replay", ty, rep)
        assert bool(t.isfinite(dst.float().sum()))   # eager work on the buffer between the replays
    del gr

# the single-tensor quantizers (their groups of tiles start behind a probe of the packed output, as the decode's do)
# from a captured graph, replayed with new inputs in the same buffers, two launches back to back per replay
for ty, in_dt in ((8, t.float32), (3, t.float16), (15, t.bfloat16), (6, t.float32)):
    n = g.block_elems(ty)
    elems = 70 * 2048 + 3 * n                       # many groups of tiles and a ragged last tile
    x = t.empty(elems, dtype=in_dt, device="cuda")
    q = t.empty(elems // n * g.block_bytes(ty), dtype=t.uint8, device="cuda")
    x.normal_()
    side = t.cuda.Stream()
    with t.cuda.stream(side):
        g.quantize_slice(ty, q, x)                  # warm-up outside the capture
    side.synchronize()
    gr = t.cuda.CUDAGraph()
    with t.cuda.graph(gr):
        g.quantize_slice(ty, q, x)
        g.quantize_slice(ty, q, x)
    gen = t.Generator(device="cuda")
    for rep in range(3):
        gen.manual_seed(1234 + 10 * ty + rep)
        x.copy_((t.randn(elems, device="cuda", generator=gen) * (0.02 + rep)).to(in_dt))
        gr.replay()
        t.cuda.synchronize()
        host = x.cpu().numpy() if in_dt == t.float32 else x.view(t.int16).cpu().numpy().view(np.uint16)
        want = o.quantize(ty, host, in_ty={t.float32: o.F32, t.float16: o.F16, t.bfloat16: o.BF16}[in_dt])
        assert np.array_equal(q.cpu().numpy(), want), ("quantizer graph replay", ty, rep)
        assert int(q.sum(dtype=t.int64).item()) >= 0   # eager work on the buffer between the replays
    del gr

# batch decode: one group of 11 tiles in 4 launches, several tensors with short tiles
tensors, packed = [], []
for i, blocks in enumerate((16 * 4 + 1, 7, 16 * 5, 16 + 3)):
    p = synth.random_blocks(12, blocks, 70 + i)
    tensors.append((12, t.from_numpy(p).cuda(), t.empty(blocks * 256, dtype=t.float32, device="cuda")))
    packed.append(p)
b = g.Batch(tensors)
b.run()
t.cuda.synchronize()
for (_, _, d), p in zip(tensors, packed):
    assert same_bits(d.cpu().numpy(), o.dequantize(12, p, threads=4)), "batch"
b.close()

# hybrid plans: tensors of every route in one plan (tile table, launches of their own, widening), several types,
# 2- and 4-byte outputs, odd source offsets, ragged last tiles, empty tensors; sizes straddle the limit (20000
# elements) on both sides.  Third field: the tensor sits right behind the previous tensor of its type, in the packed
# arena and in the output arena, so the two form one run (runs reach the limit that their tensors alone do not).
SPEC = ((12, 19968), (12, 20224), (14, 8192 * 5 + 256), (14, 256), (8, 20000 - 32), (8, 20000), (2, 32 * 2000),
        (12, 0), (13, 256 * 400), (1, 30000), (20, 32 * 625), (12, 256 * 3), (12, 8192 * 3, True), (14, 256 * 33, True),
        (12, 256, True), (12, 256 * 40, True), (12, 256 * 31), (12, 256 * 90, True), (12, 512), (12, 256 * 77, True),
        (12, 256 * 5, True), (8, 32 * 700), (8, 32, True), (15, 256 * 100), (23, 256 * 81), (6, 32 * 1000),
        (7, 32 * 999), (3, 32 * 640), (9, 32 * 641), (3, 32 * 3, True), (3, 32 * 77, True))
for out_dt, o_out, o_ty in ((t.float32, o.F32, g.GGmlType.F32), (t.float16, o.F16, g.GGmlType.F16),
                            (t.bfloat16, o.BF16, g.GGmlType.BF16)):
    ob = 4 if out_dt == t.float32 else 2
    tensors, packed, tail = [], [], {}          # tail[type] = (packed arena, bytes used, output arena, elements used)
    for i, spec in enumerate(SPEC):
        ty, elems, adj = (spec + (False,))[:3]
        blocks = elems // g.block_elems(ty)
        p = synth.random_blocks(ty, blocks, 170 + i)
        if adj:
            raw, used_b, out, used_e = tail[ty]
        else:
            off = (3 * i) % 16 if ty not in (1, 30) else 8
            raw, used_b = t.empty(off + (1 << 20), dtype=t.uint8, device="cuda"), off
            out, used_e = t.full((1 << 20,), float("nan"), dtype=out_dt, device="cuda"), 0
        src = raw[used_b:used_b + p.size]
        src.copy_(t.from_numpy(p))
        dst = out[used_e:used_e + elems]
        tail[ty] = (raw, used_b + p.size, out, used_e + elems)
        tensors.append((ty, src, dst))
        packed.append(p)
    b = g.Batch(tensors, o_ty)
    want_entries = plan_entries(SPEC, ob)
    assert b.launches == want_entries, ("hybrid plan entries", b.launches, want_entries, FORCED, MAX_GRID)
    for rep in range(2):
        b.run()
    t.cuda.synchronize()
    for (ty, _, d), p in zip(tensors, packed):
        want = o.dequantize(ty, p, o_out, threads=4)
        ok = same_bits(d.cpu().numpy(), want) if out_dt == t.float32 else \
            same_bits16(bits16(d), want, "f16" if out_dt == t.float16 else "bf16")
        assert ok, ("hybrid plan", ty, d.numel(), out_dt)
    for ty, (raw, used_b, out, used_e) in tail.items():         # nothing written behind a run's last element
        rest = out[used_e:used_e + 4096]
        assert bool(t.isnan(rest.float()).all()), ("hybrid plan wrote past a run", ty)
    b.close()

# whole-file decode: the loader places a group's tensors of one type back to back (packed and decoded), so every
# type is one run; runs above the limit get launches of their own
import tempfile  # noqa: E402
from gguf_amd import gguf as G  # noqa: E402
fspec = [("a.weight", 12, (2048, 6)), ("n0.weight", 0, (2048,)), ("b.weight", 14, (2048, 3)), ("c.weight", 12, (2048, 5)),
         ("d.weight", 8, (2048, 4)), ("e.weight", 14, (2048, 9)), ("f.weight", 1, (2048, 2)), ("g.weight", 12, (256, 3)),
         ("h.weight", 8, (2048, 7)), ("i.weight", 10, (2048, 12)), ("j.weight", 20, (2048, 3))]
ft, fpacked = [], {}
for i, (name, ty, shape) in enumerate(fspec):
    n = int(np.prod(shape))
    data = np.random.default_rng(i).standard_normal(n).astype(np.float32).view(np.uint8) if ty == 0 else \
        synth.random_blocks(ty, n // g.block_elems(ty), 300 + i)
    fpacked[name] = (ty, data)
    ft.append((name, ty, shape, data))
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "typed.gguf")
    open(path, "wb").write(G.build_gguf({"general.architecture": "llama"}, ft))
    with G.GGuf(path) as f:
        for out, o_out in (("f32", o.F32), ("f16", o.F16)):
            stats = {}
            res = G.dequantize_file(f, out=out, staging_bytes=1 << 20, stats=stats)
            per_type = {}
            for name, ty, shape in fspec:
                if ty != 0:
                    per_type[ty] = per_type.get(ty, 0) + int(np.prod(shape))
            want_entries = plan_entries([(ty, n) for ty, n in per_type.items()], 4 if out == "f32" else 2)
            assert (stats["groups"], stats["launches"]) == (1, want_entries), (stats, want_entries, FORCED)
            for name, (ty, data) in fpacked.items():
                if ty == 0:
                    assert np.array_equal(res[name].cpu().numpy().view(np.uint8), data)
                    continue
                want = o.dequantize(ty, data, o_out, threads=4)
                ok = same_bits(res[name].cpu().numpy(), want) if out == "f32" else same_bits16(bits16(res[name]), want, "f16")
                assert ok, ("file decode", name, out)
    G.close_thread_contexts()

# a group of more than 3 * 256 tiles: the tile table itself is built by several launches
p = synth.random_blocks(12, 16 * 1700 + 3, 79)
d = t.empty((16 * 1700 + 3) * 256, dtype=t.float32, device="cuda")
b = g.Batch([(12, t.from_numpy(p).cuda(), d)])
b.run()
t.cuda.synchronize()
assert same_bits(d.cpu().numpy(), o.dequantize(12, p, threads=4)), "batch, table built in pieces"
b.close()

# widen (F16 / BF16 sources) and narrow / quantize
for ty in (1, 30):
    nel = 10 * 4096 + 5
    p = synth.random_blocks(ty, nel, 80 + ty)
    dst = t.empty(nel, dtype=t.float32, device="cuda")
    g.dequantize_slice(ty, dst, t.from_numpy(p).cuda())
    t.cuda.synchronize()
    assert same_bits(dst.cpu().numpy(), o.dequantize(ty, p, threads=4)), ("widen", ty)
x = (np.random.default_rng(5).standard_normal(10 * 8192 + 64) * 0.1).astype(np.float32)
for ty in (1, 30, 2, 3, 6, 7, 8, 9):
    q = t.empty(x.size // g.block_elems(ty) * g.block_bytes(ty), dtype=t.uint8, device="cuda")
    g.quantize_slice(ty, q, t.from_numpy(x).cuda())
    t.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), o.quantize(ty, x)), ("quantize", ty)

# permute_qk: 4096-wide Q4_K rows (2304 bytes = 144 pieces of 16), 64 rows -> 9 workgroups -> 3 launches
rb = R.row_bytes(12, 4096)
src = np.random.default_rng(6).integers(0, 256, 64 * rb, dtype=np.uint8)
got = R.permute_qk(12, 4096, 64, 4, t.from_numpy(src).cuda())
t.cuda.synchronize()
assert np.array_equal(got.cpu().numpy(), np_permute_qk(src, rb, 64, 4)), "permute"
# and with divisors that are not powers of two: 25 pieces a row (F32 rows of 100), heads of 6 and of 10 rows
for nh, dh in ((8, 6), (3, 10), (5, 24)):
    rb = R.row_bytes(0, 100)
    src = np.random.default_rng(nh).integers(0, 256, nh * dh * rb, dtype=np.uint8)
    got = R.permute_qk(0, 100, nh * dh, nh, t.from_numpy(src).cuda())
    t.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), np_permute_qk(src, rb, nh * dh, nh)), ("permute", nh, dh)

# scale: 10 workgroups of 16-byte pieces plus a tail of single elements, aligned and from an odd address
for ty, npdt in ((0, np.float32), (1, np.uint16), (30, np.uint16), (28, np.float64)):
    eb = np.dtype(npdt).itemsize
    nel = 10 * 256 * (16 // eb) + 3 * 256 + 5
    v = np.random.default_rng(7 + ty).standard_normal(nel)
    host = v.astype(npdt) if ty in (0, 28) else v.astype(np.float16).view(np.uint16) if ty == 1 else \
        (v.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    want = o.scale(ty, host, 1.4).view(np.uint8)
    dev = t.empty(host.nbytes + 16, dtype=t.uint8, device="cuda")
    for off in (0, eb):
        view = dev[off:off + host.nbytes]
        view.copy_(t.from_numpy(host.view(np.uint8).copy()))
        got = R.scale(ty, view, 1.4)
        t.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), want), ("scale", ty, off)
print("OK")
