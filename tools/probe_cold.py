"""Probe: single-tensor decode of a COLD input (caches evicted by READING 2 GiB before every launch; a
fill would leave dirty lines whose write-back lands inside the timed launch), with
the burst prefetch on and off (GGQ_NO_PREFETCH=1 in a second process), for tensor sizes from 4 Mi to 512 Mi
elements (PROBE_TYPES, PROBE_OUT, PROBE_SIZES select others).  Repeated launches over the same small input would find it in the memory-side cache and
say nothing about a tensor that is decoded once.
Run on the GPU box: python tools/probe_cold.py ; GGQ_NO_PREFETCH=1 python tools/probe_cold.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gguf_amd as g  # noqa: E402
from gguf_amd import synth  # noqa: E402
from gguf_amd.quants import GGmlType  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    evict = torch.zeros(2 << 28, dtype=torch.int32, device=dev)    # 2 GiB
    tag = "prefetch off" if os.environ.get("GGQ_NO_PREFETCH") else "prefetch " + os.environ.get("GGQ_PREFETCH", "on ")
    sizes = [int(x) for x in os.environ.get("PROBE_SIZES", "4,16,45,128,512").split(",")]
    types = [GGmlType[x] for x in os.environ.get("PROBE_TYPES", "Q4K,Q6K").split(",")]
    out_dt = {"f32": torch.float32, "f16": torch.float16}[os.environ.get("PROBE_OUT", "f32")]
    for ty in types:
        for mi in sizes:
            n = mi << 20
            src = synth.random_blocks_device(ty, n // g.block_elems(ty), 3, dev)
            dst = torch.empty(n, dtype=out_dt, device=dev)
            for _ in range(20):                      # clocks up
                g.dequantize_slice(ty, dst, src)
            ts = []
            for _ in range(15):
                sink = evict.sum()                   # leaves only clean lines in the caches
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.dequantize_slice(ty, dst, src)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            med = ts[len(ts) // 2]
            per = g.block_bytes(ty) / g.block_elems(ty) + dst.element_size()
            print(f"{tag} {ty.name:4s} {mi:4d} Mi elements  {med * 1e3:9.1f} us  {n * per / med / 1e6:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
