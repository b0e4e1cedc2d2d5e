"""Probe: what costs the benchmark's batch launches bandwidth against one contiguous tensor?
Variants of the same Q4_K tiles: separately allocated or carved from one arena; outputs contiguous or
interleaved with the other type's outputs; the launch alone or alternating with the Q6_K launch.
Run on the GPU box: python tools/probe_arena.py <workload>
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
    base = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "tinyllama_q4_k_m")
    srcs = [synth.random_blocks_device(ty, ne // 256, 7 + i, dev) for i, (n, ty, ne) in enumerate(base)]
    total = sum(ne for _, _, ne in base)
    dst_all = torch.empty(total, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def plan(ty_want, interleaved):
        at, ts = 0, []
        for (n, ty, ne), s in zip(base, srcs):
            if ty == ty_want:
                ts.append((ty, s, dst_all[at:at + ne]))
            if ty == ty_want or interleaved:
                at += ne
        return g.Batch(ts, GGmlType.F32), sum(t[2].numel() for t in ts)

    def time_it(label, fn, elems_bytes, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        print(f"{label:58s} {best:.4f} ms  {elems_bytes / best / 1e6:7.0f} GB/s", flush=True)
        return best

    for inter in (False, True):
        b4, n4 = plan(GGmlType.Q4K, inter)
        b6, n6 = plan(GGmlType.Q6K, inter)
        tag = "outputs interleaved" if inter else "outputs contiguous per type"
        t4 = time_it(f"Q4_K launch alone, {tag}", lambda: b4.run(st), n4 * 4.5625)
        t6 = time_it(f"Q6_K launch alone, {tag}", lambda: b6.run(st), n6 * 4.8203125)

        def both():
            b4.run(st)
            b6.run(st)
        tb = time_it(f"Q4_K then Q6_K alternating, {tag}", both, n4 * 4.5625 + n6 * 4.8203125)
        print(f"    sum of the two alone {t4 + t6:.4f} ms, alternating {tb:.4f} ms", flush=True)
        b4.close()
        b6.close()


if __name__ == "__main__":
    main()
