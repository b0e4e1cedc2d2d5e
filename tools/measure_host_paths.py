"""Rates of the two entry points that cross PCIe, (reported beside the kernel numbers, never the benchmark's `value`):
  host -> host   ggq_dequantize_slice   (packed in, decoded out, both over PCIe)
  host -> device ggq_dequantize_async   (only the packed bytes cross; decoded tensor stays in HBM)
Each host -> host rate twice: into a pre-faulted destination, and into a FRESH anonymous mapping per call, which
is how the reference's caller allocates its output (xtask/src/utils/operator/cast.rs:159-161) -- there the
first-touch page faults of the destination are part of the call.  The CPU port of the reference's decode is timed
beside it under the same two conditions (bench.cpu_baseline, the benchmark's own CPU leg).
Run on the GPU box: python tools/measure_host_paths.py
"""
import json
import mmap
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import gguf_amd as g
    from gguf_amd import synth
    T = g.GGmlType
    out = {}
    for ty, elems in ((T.Q4K, 1 << 28), (T.Q6K, 1 << 28)):
        blocks = elems // g.block_elems(ty)
        chunk = synth.random_blocks(ty, 4096, 1)
        packed = np.tile(chunk, blocks // 4096 + 1)[: blocks * g.block_bytes(ty)].copy()
        dst = np.zeros(elems, dtype=np.float32)          # pre-faulted
        g.dequantize_slice(ty, dst[: 1 << 22], packed[: (1 << 22) // g.block_elems(ty) * g.block_bytes(ty)])   # warm the pool
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            g.dequantize_slice(ty, dst, packed)
            best = min(best, time.perf_counter() - t0)
        out[f"{ty.name}_host_to_host"] = {"decoded_GBps": round(elems * 4 / best / 1e9, 2),
                                          "melem_per_s": round(elems / best / 1e6, 1), "seconds": round(best, 4)}
        fresh = []
        for _ in range(3):
            t0 = time.perf_counter()
            m = mmap.mmap(-1, elems * 4)
            f = np.frombuffer(m, dtype=np.float32)
            g.dequantize_slice(ty, f, packed)
            fresh.append(time.perf_counter() - t0)
            del f
            m.close()
        fresh.sort()
        out[f"{ty.name}_host_to_host_fresh_mapping"] = {"decoded_GBps": round(elems * 4 / fresh[1] / 1e9, 2),
                                                        "melem_per_s": round(elems / fresh[1] / 1e6, 1),
                                                        "seconds": round(fresh[1], 4), "note": "median of 3"}
        d = torch.empty(elems, dtype=torch.float32, device="cuda")
        pinned = torch.from_numpy(packed).pin_memory()
        for label, src in (("pageable", packed), ("pinned", pinned.numpy())):
            for staging in (64 << 20,):
                ctx = g.Context(0, staging)
                ctx.dequantize_async(ty, d[: 1 << 22], src[: (1 << 22) // g.block_elems(ty) * g.block_bytes(ty)])
                ctx.sync()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    ctx.dequantize_async(ty, d, src)
                    ctx.sync()
                    best = min(best, time.perf_counter() - t0)
                ctx.close()
                out[f"{ty.name}_host_{label}_to_device"] = {
                    "decoded_GBps": round(elems * 4 / best / 1e9, 2), "packed_GBps": round(packed.size / best / 1e9, 2),
                    "melem_per_s": round(elems / best / 1e6, 1), "seconds": round(best, 4)}
    # the CPU port under the same two conditions (pre-faulted / fresh mapping), on this box's cores
    import bench
    for ty, name in ((T.Q4K, "Q4K"), (T.Q6K, "Q6K")):
        rec = bench.cpu_baseline([("w", ty, 1 << 28)], "f32", 6.0, min_elems=1)
        out[f"{name}_cpu_port"] = {"prefaulted_GBps": rec["value"], "fresh_mapping_GBps": rec["fresh_mapping_value"],
                                   "cores": rec["cores"], "build": rec["build"]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
