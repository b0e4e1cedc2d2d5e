"""The host entry points under concurrent callers, as the reference's per-shard writer threads use
them (xtask/src/utils/write.rs:64-68): every thread borrows its own pipeline, the staging copies
share the helper threads; results must still equal the oracle's."""
import threading

import numpy as np
import pytest

from util import same_bits

pytestmark = pytest.mark.gpu


def test_concurrent_host_callers(oracle):
    import gguf_amd as g
    from gguf_amd import synth
    jobs = []
    for i, (ty, blocks) in enumerate([(12, 30000), (14, 20011), (8, 150001), (2, 99999), (12, 7), (13, 40000),
                                      (10, 25000), (23, 18000)]):
        packed = synth.random_blocks(ty, blocks, 100 + i)
        jobs.append((ty, packed, oracle.dequantize(ty, packed, threads=8)))
    errors = []

    def worker(k):
        try:
            for rep in range(3):
                ty, packed, want = jobs[(k + rep) % len(jobs)]
                dst = np.empty(want.size, dtype=np.float32)
                g.dequantize_slice(ty, dst, packed)
                if not same_bits(dst, want):
                    errors.append((k, rep, ty))
                # and back: quantize what a legacy format decoded to
                if ty in (2, 8):
                    q = np.empty(packed.size, dtype=np.uint8)
                    g.quantize_slice(ty, q, want)
                    if not np.array_equal(q, oracle.quantize(ty, want)):
                        errors.append((k, rep, ty, "quantize"))
        except BaseException as e:
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_one_context_per_thread(oracle):
    """A ggq_ctx belongs to one caller; different contexts on one device are independent.  Two threads,
    each with its own context on device 0, stream different block types at the same time."""
    import torch
    import gguf_amd as g
    from gguf_amd import synth
    plans = [[(12, 40000), (14, 9001), (12, 33), (10, 20000)], [(2, 150000), (23, 12000), (8, 70001), (20, 5)]]
    errors = []

    def worker(k):
        try:
            ctx = g.Context(0, staging_bytes=2 << 20)        # small staging: many chunks, slots reused often
            work = []
            for rep in range(2):
                for i, (ty, blocks) in enumerate(plans[k]):
                    p = synth.random_blocks(ty, blocks, 900 + 10 * k + i + rep)
                    dst = torch.empty(blocks * g.block_elems(ty), dtype=torch.float32, device="cuda:0")
                    ctx.dequantize_async(ty, dst, p)
                    work.append((ty, p, dst))
            ctx.sync()
            for ty, p, dst in work:
                if not same_bits(dst.cpu().numpy(), oracle.dequantize(ty, p, threads=4)):
                    errors.append((k, ty))
            ctx.close()
        except BaseException as e:
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
