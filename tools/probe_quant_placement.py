#!/usr/bin/env python3
"""quant_one<EncQ4_0, InF32> on 1 Gi elements in several "placements": the same kernel, the same input values,
different buffers and different predecessors.  Every launch is timed on its own with HIP events (30 per placement),
so each placement gets a distribution, not one average.  Under `rocprofv3 --kernel-trace` / `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` the dispatches come out in the order printed here (placement k = dispatches 30k .. 30k+29 of the
kernel), which is how profiles/summarize_quant_placement.py attaches durations and counters to placements.
Prints one JSON object."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gguf_amd as g  # noqa: E402
from gguf_amd.quants import GGmlType as T  # noqa: E402

n = 1 << 30
reps = 30
ty = T.Q4_0
nb = n // 32 * g.block_bytes(ty)
alg = n * 4 + nb
x0 = torch.randn(n, device="cuda") * 0.02
t_end = time.perf_counter() + 0.3                      # clock ramp
while time.perf_counter() < t_end:
    x0.mul_(1.0)
    torch.cuda.synchronize()


def timed(dst, src):
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.quantize_slice(ty, dst, src)
        e1.record()
        out.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in out]


def spoil():
    """What the kernels before a placement leave behind: 4 GiB written (dirty lines in the memory-side cache)."""
    junk = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
    junk.fill_(1.0)
    torch.cuda.synchronize()
    del junk


placements = []


def run(label, dst, src, before=None):
    if before:
        before()
    ms = timed(dst, src)
    fr = [alg / (m * 1e-3) / 8e12 for m in ms]
    placements.append({"placement": label, "src": hex(src.data_ptr()), "dst": hex(dst.data_ptr()),
                       "frac_median": round(statistics.median(fr), 4), "frac_min": round(min(fr), 4),
                       "frac_max": round(max(fr), 4), "frac_first": round(fr[0], 4),
                       "ms": [round(m, 4) for m in ms]})


d0 = torch.empty(nb, dtype=torch.uint8, device="cuda")
run("A: first use, right after the warm-up", d0, x0)
run("B: same buffers again", d0, x0)
d1 = torch.empty(nb, dtype=torch.uint8, device="cuda")
run("C: another destination buffer", d1, x0)
run("D: after 4 GiB of unrelated writes", d1, x0, spoil)
x1 = x0.clone()
run("E: another source buffer (a copy)", d1, x1)
big = torch.empty(nb + (4 << 20), dtype=torch.uint8, device="cuda")
run("F: destination 2 MiB + 4 KiB into an allocation", big[(2 << 20) + 4096:(2 << 20) + 4096 + nb], x1)
xb = torch.empty(n + (1 << 20), dtype=torch.float32, device="cuda")
xs = xb[1028:1028 + n]
xs.copy_(x0)
run("G: source 4112 bytes into an allocation", d0, xs)
run("H: the first placement's buffers once more", d0, x0)
print(json.dumps({"kernel": "quant_one<EncQ4_0, InF32>", "elements": n, "algorithmic_bytes": alg, "launches_per_placement": reps,
                  "placements": placements}))
