#!/usr/bin/env python3
"""Benchmark of the GGUF dequantize hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--out f32|f16|bf16]

One "step" decodes one whole synthetic tensor set that is already resident in HBM
(packed blocks in, f32 out), through the C ABI (ggq_batch_run: one kernel launch
per block type present).  Default workload: the Llama-7B-shaped tensor set under
the Q4_K_M type mixture (6.74 G elements, 226 tensors, Q4_K + Q6_K).

N > 1 (launched by torch.distributed.run, one rank per GPU): the job is N such sets
(weak scaling), dealt to the ranks by the index partition of gguf_amd.shard; no
collective takes part in the data path, the process group is used for the barriers
around the timed region and for the max-over-ranks of the time.

Prints ONE JSON line on rank 0.  `value` = decoded (output) bytes of the whole job
per second.  `roofline` prices the dominant kernel (the Q4_K launch) in algorithmic
bytes = elements x (block_bytes/block_elems + sizeof(out)) against 8 TB/s, with the
launch duration measured by HIP events on the launch stream inside the timed region.
`cpu_baseline` times the CPU oracle (a port of the reference's per-block decode) on
this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed steps run for this long before the warm-up steps (clock ramp)")
    ap.add_argument("--workload", default="llama7b_q4_k_m")
    ap.add_argument("--out", default="f32", choices=["f32", "f16", "bf16"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank's share is one whole workload (default); strong: ONE workload "
                         "split over the ranks (for sets that do not fit one GPU, e.g. llama70b_q4_k_m_iq4nl)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(items, out_ty_name, budget_s):
    """Oracle (C port of the reference decode, blocks split over a thread pool like the
    reference's rayon loop) on a bounded sample: the leading tensors of the workload up
    to ~0.4 G elements, decoded repeatedly for about `budget_s` seconds; best pass."""
    import numpy as np
    import oracle as o
    from gguf_amd import synth
    from gguf_amd.quants import block_bytes, block_elems
    o.build()
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    out_ty = {"f32": o.F32, "f16": o.F16, "bf16": o.BF16}[out_ty_name]
    ob = 4 if out_ty_name == "f32" else 2
    sample, elems = [], 0
    for name, ty, n in items:
        sample.append((name, ty, n))
        elems += n
        if elems >= 400_000_000:
            break
    bufs = []
    for i, (name, ty, n) in enumerate(sample):
        blocks = n // block_elems(ty)
        # payload content does not change the decode's cost; tile a 4 MiB random chunk
        chunk = synth.random_blocks(ty, min(blocks, (4 << 20) // block_bytes(ty)), 0xC0FFEE04 + i)
        reps = -(-blocks * block_bytes(ty) // chunk.size)
        packed = np.tile(chunk, reps)[: blocks * block_bytes(ty)]
        dst = np.zeros(n, dtype=np.float32 if ob == 4 else np.uint16)   # pre-faulted
        bufs.append((int(ty), packed, dst, blocks))
    best = float("inf")
    passes = 0
    t_end = time.perf_counter() + budget_s
    while passes < 2 or (time.perf_counter() < t_end and passes < 20):
        t0 = time.perf_counter()
        for ty, packed, dst, blocks in bufs:
            st = o.dequantize_status(ty, out_ty, dst, dst.size, packed, blocks, threads=cores)
            assert st == 0
        best = min(best, time.perf_counter() - t0)
        passes += 1
    # what the reference's caller does (cast.rs:159-161): a fresh anonymous mapping per tensor,
    # so the decode also pays the first-touch page faults of its output
    import mmap
    t0 = time.perf_counter()
    for ty, packed, dst, blocks in bufs:
        m = mmap.mmap(-1, dst.nbytes)
        fresh = np.frombuffer(m, dtype=dst.dtype)
        st = o.dequantize_status(ty, out_ty, fresh, fresh.size, packed, blocks, threads=cores)
        assert st == 0
        del fresh
        m.close()
    fresh_s = time.perf_counter() - t0
    types = sorted({o_ty for o_ty, _, _, _ in bufs})
    return {
        "value": round(elems * ob / best / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
        "melem_per_s": round(elems / best / 1e6, 1),
        "fresh_mapping_value": round(elems * ob / fresh_s / 1e9, 3),
        "sample": f"first {len(sample)} tensors of the workload ({elems} elements, ggml types {types}), "
                  f"pre-faulted output, best of {passes} passes",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import gguf_amd as g
    from gguf_amd import shard, synth
    from gguf_amd.quants import GGmlType, block_bytes, block_elems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    ndev = torch.cuda.device_count()
    # Normal case: one rank per GPU over RCCL.  Rehearsal case (fewer GPUs than ranks,
    # e.g. two ranks on a one-GPU box): ranks share devices and the process group is gloo
    # on host tensors -- the group only carries the barriers and the max of the times.
    shared = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- the job: `world` tensor sets, dealt to ranks by the index partition -------
    base = synth.workload(args.workload)
    sets = world if args.scaling == "weak" else 1
    items = [(f"set{r}.{n}", ty, ne) for r in range(sets) for (n, ty, ne) in base]
    mine = shard.partition(items, world)[rank]
    out_dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.out]
    ob = 4 if args.out == "f32" else 2
    out_ty = {"f32": GGmlType.F32, "f16": GGmlType.F16, "bf16": GGmlType.BF16}[args.out]

    my_elems = sum(p.elems for p in mine)
    need = my_elems * ob + sum(p.blocks * block_bytes(p.ggml_type) for p in mine)
    free = torch.cuda.mem_get_info(dev)[0]
    if need > 0.97 * free:
        sys.exit(f"rank {rank}: this share needs {need / 2**30:.1f} GiB of HBM, {free / 2**30:.1f} GiB are free; "
                 f"use more GPUs with --scaling strong, or --out f16")
    dst_all = torch.empty(my_elems, dtype=out_dt, device=dev)
    by_type = {}
    at = 0
    for i, p in enumerate(mine):
        src = synth.random_blocks_device(p.ggml_type, p.blocks, 0xC0FFEE04 + 1000 * rank + i, dev)
        dst = dst_all[at:at + p.elems]
        at += p.elems
        by_type.setdefault(p.ggml_type, []).append((p.ggml_type, src, dst))
    # one launch plan per block type, so each kernel can be timed on its own
    order = sorted(by_type, key=lambda t: -sum(x[2].numel() for x in by_type[t]))
    plans = [(ty, g.Batch(by_type[ty], out_ty)) for ty in order]
    launches = sum(b.launches for _, b in plans)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        for _, b in plans:
            b.run(stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The GPU reaches its steady clocks only after some tens of milliseconds of load (measured:
    # the same launch runs 9 % slower in the first few milliseconds after an idle period), so
    # untimed steps run until `settle_ms` of wall time have passed, then the W warm-up steps.
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    # HIP events on the launch stream around every kernel, inside the timed region.  An event
    # between two launches costs the GPU a few microseconds of idle time (measured: 2 % of a
    # 4.7 ms step, 10 % of a 0.85 ms one), so only every `every`-th step carries them; the
    # kernel durations reported are the averages over those steps.
    every = max(1, args.steps // 5)
    sampled = [s for s in range(args.steps) if s % every == 0]
    ev = {s: [torch.cuda.Event(enable_timing=True) for _ in range(len(plans) + 1)] for s in sampled}
    t0 = time.perf_counter()
    for s in range(args.steps):
        if s in ev:
            ev[s][0].record()
            for k, (_, b) in enumerate(plans):
                b.run(stream)
                ev[s][k + 1].record()
        else:
            for _, b in plans:
                b.run(stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel launch durations (ms), averaged over the timed steps
    kern_ms = []
    for k in range(len(plans)):
        kern_ms.append(sum(ev[s][k].elapsed_time(ev[s][k + 1]) for s in sampled) / len(sampled))

    if rank == 0:
        total_elems = sum(ne for _, _, ne in items)
        pin, pout, _ = synth.workload_bytes(items, ob)
        ms = elapsed / args.steps * 1e3
        dom_ty = order[0]
        dom_elems = sum(x[2].numel() for x in by_type[dom_ty])
        per_elem = block_bytes(dom_ty) / block_elems(dom_ty) + ob
        dom_bytes = dom_elems * per_elem
        achieved = dom_bytes / (kern_ms[0] * 1e-3) / 1e9
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path))
                key = f"{args.workload}:{args.out}:{GGmlType(dom_ty).name}"
                traffic = tr.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        rec = {
            "metric": "dequantized_GBps", "value": round(pout / (elapsed / args.steps) / 1e9, 2), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_ms": args.settle_ms,
            "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "io": "u8->" + args.out, "data": "synthetic",
            "config": {"workload": (args.workload if sets == 1 else f"{sets} x {args.workload}")
                       + (f" (REHEARSAL: {world} ranks share {ndev} GPU)" if shared else ""),
                       "tensors": len(items), "elements": total_elems, "packed_bytes": pin,
                       "decoded_bytes": pout, "partition": "index (LPT), no collective",
                       "kernel_launches_per_step": launches},
            "melem_per_s": round(total_elems / (elapsed / args.steps) / 1e6, 1),
            "packed_GBps": round(pin / (elapsed / args.steps) / 1e9, 2),
            "algorithmic_GBps": round((pin + pout) / (elapsed / args.steps) / 1e9, 2),
            "kernels": [{"type": GGmlType(ty).name, "elements": sum(x[2].numel() for x in by_type[ty]),
                         "avg_ms": round(kern_ms[k], 4)} for k, ty in enumerate(order)],
            "roofline": {"bound": "hbm", "kernel": f"deq_batch<{GGmlType(dom_ty).name},{args.out}> (rank 0)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(kern_ms[0], 4),
                         "timed_launches": len(sampled)},
        }
        if world == 1 and not args.no_cpu_baseline:
            for _, b in plans:
                b.close()
            del dst_all, by_type, plans
            torch.cuda.empty_cache()
            rec["cpu_baseline"] = cpu_baseline(base, args.out, args.cpu_seconds)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
