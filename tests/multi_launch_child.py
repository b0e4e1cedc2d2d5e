"""Run by test_gpu_multi_launch.py in a child process whose environment sets GGQ_MAX_GRID=3, so that
every entry point that splits large inputs over several kernel launches does so on small inputs.
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
from util import same_bits, same_bits16  # noqa: E402

assert os.environ.get("GGQ_MAX_GRID") == "3"
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
