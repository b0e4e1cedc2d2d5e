"""A GGUF file on disk -> device tensors, two ways, for the per-tensor-versus-plan comparison:

  --mode batch       gguf_amd.gguf.dequantize_file: packed bytes streamed up, ONE launch plan per group
                     (one kernel launch per block type present)
  --mode per-tensor  one ggq_dequantize_async per tensor (a kernel launch per staged chunk of every tensor)

and, with --resident, the same tensors, now resident in HBM, decoded (a) by one ggq_dequantize_device call per tensor,
back to back on one stream, and (b) by one launch plan -- timed with HIP events.

The file decode is PCIe-bound end to end; what differs between the modes is the kernel side.  Run under
`rocprofv3 --kernel-trace --stats -- python3 tools/measure_file_decode.py ...` to get the kernel durations
(profiles/summarize_file_decode.py turns them into algorithmic GB/s).
Run on the GPU box.  Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="tinyllama_q4_k_m")
    ap.add_argument("--mode", default="batch", choices=["batch", "per-tensor"])
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--resident", action="store_true",
                    help="also time the resident per-tensor-launch / one-plan comparison (leave out under rocprofv3, "
                         "so that every decode kernel in its trace belongs to the file decode)")
    args = ap.parse_args()
    import torch
    import gguf_amd as g
    from gguf_amd import gguf as G, synth
    from gguf_amd.quants import block_bytes, block_elems
    items = synth.workload(args.workload)
    path = os.path.join(args.dir, f"ggq_probe_{args.workload}.gguf")
    w = G.GGufFileWriter(path, len(items), 1)
    w.write_meta("general.architecture", "llama")
    for i, (name, ty, n) in enumerate(items):
        data = synth.random_blocks_device(ty, n // block_elems(ty), 0xC0FFEE03 + i, "cuda").cpu().numpy()
        w.write_tensor(name, int(ty), [n], data)
    file_bytes = w.finish()
    pin, pout, elems = synth.workload_bytes(items, 4)
    out = {"workload": args.workload, "mode": args.mode, "tensors": len(items), "elements": elems,
           "packed_bytes": pin, "decoded_bytes": pout, "file_bytes": file_bytes,
           "algorithmic_bytes_by_type": {}}
    for _, ty, n in items:
        k = g.GGmlType(ty).name
        out["algorithmic_bytes_by_type"][k] = out["algorithmic_bytes_by_type"].get(k, 0) + \
            n * (block_bytes(ty) / block_elems(ty) + 4)
    with G.GGuf(path) as f:
        walls = []
        for rep in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if args.mode == "batch":
                stats = {}
                res = G.dequantize_file(f, stats=stats)
                out["launches_per_pass"] = stats["launches"]
                out["groups"] = stats["groups"]
            else:
                ctx = g.Context(0, 64 << 20)
                res = {}
                for t in f.tensors:
                    dst = torch.empty(t.numel, dtype=torch.float32, device="cuda")
                    ctx.dequantize_async(t.ty, dst, t.bytes())
                    res[t.name] = dst
                ctx.sync()
                ctx.close()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
            if rep + 1 < args.reps:
                del res
        out["file_decode"] = {"seconds_best": round(min(walls), 4), "packed_GBps": round(pin / min(walls) / 1e9, 2),
                              "decoded_GBps": round(pout / min(walls) / 1e9, 2), "passes": args.reps}
        if not args.resident:
            os.remove(path)
            print(json.dumps(out))
            return
        # resident: per-tensor launches against one plan, same tensors
        srcs = []
        for t in f.tensors:
            srcs.append((t.ty, torch.from_numpy(t.bytes()).cuda(), res[t.name]))
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    plan = g.Batch(srcs)

    def per_tensor():
        for ty, s, d in srcs:
            g.dequantize_slice(ty, d, s)

    def one_plan():
        plan.run(stream)

    # the per-tensor loop again as ONE captured graph: replay has no per-call host work, so what remains is the
    # GPU's own cost of 150-230 dependent launches
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        cap = torch.cuda.current_stream().cuda_stream
        for ty, s, d in srcs:
            g.dequantize_slice(ty, d, s, stream=cap)

    for label, fn in (("per_tensor_launches", per_tensor), ("per_tensor_launches_in_a_graph", graph.replay),
                      ("one_plan", one_plan)):
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:      # clocks up
            fn()
            torch.cuda.synchronize()
        times = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        ms = sorted(times)[len(times) // 2]
        out[f"resident_{label}"] = {"ms": round(ms, 4), "launches": len(srcs) if label.startswith("per") else plan.launches,
                                    "algorithmic_GBps": round((pin + pout) / ms / 1e6, 1),
                                    "frac_of_8TBps": round((pin + pout) / ms / 1e6 / 8000, 4)}
    plan.close()
    os.remove(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
