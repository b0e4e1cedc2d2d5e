#!/usr/bin/env python3
"""How the decode rate of ONE long run (one tensor) depends on its size: Batch over a single tensor of 2^k Mi
elements, timed with HIP events.  With the default settings a run with a group-start schedule goes through the
single-tensor kernel (deq_one); with GGQ_HYBRID_MIN_ELEMS=0 in the environment through the tile table (deq_batch).
Prints one line per (type, output, size): fraction of 8 TB/s (algorithmic bytes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gguf_amd as g  # noqa: E402
from gguf_amd.quants import GGmlType, block_bytes, block_elems  # noqa: E402

out_name = sys.argv[1] if len(sys.argv) > 1 else "f16"
sizes_mi = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "256,1024,4096,16384,32768,49152").split(",")]
types = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "12,14").split(",")]
out_dt = {"f32": torch.float32, "f16": torch.float16}[out_name]
out_ty = {"f32": GGmlType.F32, "f16": GGmlType.F16}[out_name]
ob = 4 if out_name == "f32" else 2
mode = ("table" if os.environ.get("GGQ_HYBRID_MIN_ELEMS") == "0" else "own") + " " + os.environ.get("GGQ_PREFETCH", "default")
for ty in types:
    for mi in sizes_mi:
        n = mi << 20
        blocks = n // block_elems(ty)
        nbytes = blocks * block_bytes(ty)
        from gguf_amd import synth
        if blocks <= (1 << 24):
            src = synth.random_blocks_device(ty, blocks, 1, "cuda")
        else:       # benign blocks, generated in pieces
            src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            step = 1 << 24
            for k, b0 in enumerate(range(0, blocks, step)):
                nb = min(step, blocks - b0)
                src[b0 * block_bytes(ty):(b0 + nb) * block_bytes(ty)] = synth.random_blocks_device(ty, nb, 1 + k, "cuda")
        dst = torch.empty(n, dtype=out_dt, device="cuda")
        b = g.Batch([(ty, src, dst)], out_ty)
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:          # clock ramp
            b.run()
            torch.cuda.synchronize()
        reps = max(8, min(40, (16 << 30) // n))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        alg = n * (block_bytes(ty) / block_elems(ty) + ob)
        print(f"{mode} {GGmlType(ty).name}->{out_name} {mi} Mi elements: {ms:.3f} ms, {alg / ms / 1e6:.0f} GB/s, "
              f"{alg / ms / 1e6 / 8000:.4f} of 8 TB/s, launches {b.launches}", flush=True)
        b.close()
        del src, dst
        torch.cuda.empty_cache()
