"""The first benchmark configuration on the GPU: the TinyLlama-1.1B F16 tensor table (tests/golden/tinyllama_table.json,
seeded payload; the file ships none) quantized F16 -> Q8_0 and decoded Q8_0 -> F32, the round trip the reference runs
on the CPU (xtask `convert --steps cast:q8_0`, then `cast:f32`; ggml-quants/src/lib.rs:121-147).  Both directions as
launch plans over the 90 two-dimensional tensors, laid out back to back (one run per direction), timed with events
over `--reps` runs after a settle period; rates are priced against 8 TB/s with the algorithmic bytes
(2 + 34/32 per element for the quantizer, 34/32 + 4 for the decode).  Measurement only: the bytes of both steps are
checked at full size by tests/test_gpu_full_size.py.  Run on the GPU box:
    python tools/measure_roundtrip.py [--reps 30]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gguf_amd as g  # noqa: E402
from gguf_amd.quants import GGmlType as T  # noqa: E402


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), float(min(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "tinyllama_table.json")))["tensors"]
    mats = [x for x in table if x["type"] == 1 and len(x["shape"]) == 2]
    sizes = [x["shape"][0] * x["shape"][1] for x in mats]
    total = sum(sizes)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0xC0FFEE01)
    f16 = (torch.randn(total, device="cuda", generator=gen) * 0.02).to(torch.float16)      # the F16 payload, one arena
    q8 = torch.empty(total // 32 * 34, dtype=torch.uint8, device="cuda")
    f32 = torch.empty(total, dtype=torch.float32, device="cuda")
    quant, decode, at = [], [], 0
    for n in sizes:
        quant.append((T.Q8_0, f16[at:at + n], q8[at // 32 * 34:(at + n) // 32 * 34]))
        decode.append((T.Q8_0, q8[at // 32 * 34:(at + n) // 32 * 34], f32[at:at + n]))
        at += n
    qplan = g.QuantBatch(quant)
    dplan = g.Batch(decode, T.F32)
    t_end = time.perf_counter() + 0.3            # steady clocks first
    while time.perf_counter() < t_end:
        qplan.run()
        dplan.run()
        torch.cuda.synchronize()
    q_ms, q_best = timed(qplan.run, args.reps)
    d_ms, d_best = timed(dplan.run, args.reps)
    both_ms, _ = timed(lambda: (qplan.run(), dplan.run()), args.reps)
    out = {
        "workload": "tinyllama_f16 -> q8_0 -> f32 (90 tensors)", "elements": total,
        "quantize": {"launches": qplan.launches, "ms": round(q_ms, 4), "best_ms": round(q_best, 4),
                     "algorithmic_GBps": round(total * (2 + 34 / 32) / q_ms / 1e6, 1),
                     "frac_of_8TBps": round(total * (2 + 34 / 32) / q_ms / 1e6 / 8000, 4)},
        "decode": {"launches": dplan.launches, "ms": round(d_ms, 4), "best_ms": round(d_best, 4),
                   "algorithmic_GBps": round(total * (4 + 34 / 32) / d_ms / 1e6, 1),
                   "frac_of_8TBps": round(total * (4 + 34 / 32) / d_ms / 1e6 / 8000, 4)},
        "round_trip_ms": round(both_ms, 4), "melem_per_s_round_trip": round(total / both_ms / 1e3, 1),
    }
    qplan.close()
    dplan.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
