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
