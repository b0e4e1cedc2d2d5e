#!/usr/bin/env python3
"""Benchmark of the GGUF dequantize hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--out f32|f16|bf16]
                    [--scaling strong|weak] [--streamed-steps S]

One "step" decodes one whole synthetic tensor set that is already resident in HBM
(packed blocks in, f32 out), through the C ABI (ggq_batch_run: one kernel launch
per block type present).  Default workload: the Llama-7B-shaped tensor set under
the Q4_K_M type mixture (6.74 G elements, 226 tensors, Q4_K + Q6_K).

N > 1, one rank (process) per GPU.  Either an external launcher starts the ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or plain `python bench.py --gpus N`
starts them itself: with WORLD_SIZE unset the parent -- which never touches the GPU -- spawns N
fresh children with those variables set, relays rank 0's JSON line and exits with the ranks' code
(one process fanning out over workers, like the reference's writer threads,
xtask/src/utils/write.rs:64-99).  By default ONE tensor set is dealt to the ranks by the index
partition of gguf_amd.shard (strong scaling: the way a whole-model conversion shards its
tensors); `--scaling weak` gives every rank a whole set.  No collective takes part in the data
path; the process group (gloo, host tensors) carries only the barriers around the timed regions
and the max-over-ranks of the times.

Prints ONE JSON line on rank 0.  `value` = decoded (output) bytes of the whole job
per second, inputs resident in HBM.  `roofline` prices the dominant kernel (the Q4_K
launch) in algorithmic bytes = elements x (block_bytes/block_elems + sizeof(out))
against 8 TB/s, with the launch duration measured by HIP events on the launch stream
inside the timed region.  `verified`: after the timed region, random blocks of every launch
and their outputs are pulled back and compared bit for bit with the CPU oracle.
`streamed` (a leg of its own, never `value`): the same share decoded from page-locked HOST
memory through ggq_dequantize_async (H2D copy and decode overlapping on two streams) --
end to end, PCIe-bound.  `cpu_baseline` times the CPU oracle (a port of the reference's
per-block decode) on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0
VERIFY_BLOCKS = 4096          # per launch plan (block type)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed steps run for this long before the warm-up steps (clock ramp)")
    ap.add_argument("--workload", default="llama7b_q4_k_m")
    ap.add_argument("--out", default="f32", choices=["f32", "f16", "bf16"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default): ONE workload split over the ranks by the index partition; "
                         "weak: every rank's share is one whole workload")
    ap.add_argument("--streamed-steps", type=int, default=3,
                    help="timed steps of the streamed (host -> device, PCIe-inclusive) leg; 0 = skip it")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N > 1 with strong scaling: skip the secondary weak-scaling measurement")
    ap.add_argument("--gather", type=int, default=0, metavar="G",
                    help="N > 1: after the decode, time G all-gathers of the decoded shares (every rank ends with every "
                         "tensor; RCCL over xGMI).  Opt-in and reported on its own: the decode needs no exchange")
    ap.add_argument("--layout", default="typed", choices=["typed", "scattered"],
                    help="typed (default): a rank keeps the packed tensors of one block type back to back in one arena, "
                         "and their outputs likewise, the way the library's file loader places them, so a launch plan sees "
                         "one long run per type; scattered: every packed tensor is an allocation of its own")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def job_items(workload, scaling, world):
    """The job's tensor table [(name, ggml type, elements)] and its label."""
    from gguf_amd import synth
    base = synth.workload(workload)
    sets = world if scaling == "weak" else 1
    items = [(f"set{r}.{n}" if sets > 1 else n, ty, ne) for r in range(sets) for (n, ty, ne) in base]
    label = workload if sets == 1 else f"{sets} x {workload}"
    return base, items, label


# --------------------------------------------------------------------------------------
# CPU baseline
# --------------------------------------------------------------------------------------
def _native_oracle():
    """The oracle rebuilt on THIS machine with -march=native (the shipped library is built
    elsewhere for a generic x86-64).  Returns (ctypes lib or None, note)."""
    import ctypes
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "ggq_oracle.c")
    try:
        d = tempfile.mkdtemp(prefix="ggq_oracle_native_")
        so = os.path.join(d, "libggq_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
                               "-fPIC", "-pthread", "-shared", "-o", so, src, "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        L = ctypes.CDLL(so)
        L.ggo_dequantize_slice_mt.restype = ctypes.c_int
        L.ggo_dequantize_slice_mt.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        return L, "-O3 -march=native"
    except Exception as e:      # no compiler on the box, or the build failed: time the shipped build
        return None, f"-O3 generic x86-64 (native rebuild unavailable: {type(e).__name__})"


def cpu_baseline(items, out_ty_name, budget_s, min_elems=1_000_000_000):
    """Oracle (C port of the reference decode, blocks split over a thread pool like the
    reference's rayon loop) on a bounded sample: the leading tensors of the workload up to
    `min_elems` elements, every byte of input drawn at random (no repeated chunk, so the reads
    come from DRAM as they would from a file mapping); median of up to 10 passes within about
    `budget_s` seconds.  Two outputs: a pre-faulted buffer, and -- what the reference's caller
    does (cast.rs:159-161) -- a fresh anonymous mapping per tensor."""
    import mmap
    import numpy as np
    import oracle as o
    from gguf_amd.quants import block_bytes, block_elems
    from gguf_amd.synth import HEADER_F16
    o.build()
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    out_ty = {"f32": o.F32, "f16": o.F16, "bf16": o.BF16}[out_ty_name]
    ob = 4 if out_ty_name == "f32" else 2
    sample, elems = [], 0
    for name, ty, n in items:
        sample.append((name, ty, n))
        elems += n
        if elems >= min_elems:
            break
    native, flags = _native_oracle()
    rng = np.random.default_rng(0xC0FFEE04)
    bufs = []
    for name, ty, n in sample:
        blocks = n // block_elems(ty)
        bb = block_bytes(ty)
        packed = rng.integers(0, 256, size=blocks * bb, dtype=np.uint8)
        view = packed.reshape(blocks, bb)
        for off in HEADER_F16.get(ty, ()):                  # finite, normal scales
            view[:, off] = rng.integers(0, 256, size=blocks, dtype=np.uint8)
            view[:, off + 1] = rng.integers(0x20, 0x30, size=blocks, dtype=np.uint8)
        dst = np.empty(n, dtype=np.float32 if ob == 4 else np.uint16)
        dst.fill(0)                                          # pre-faulted
        bufs.append((int(ty), packed, dst, blocks))

    def decode(ty, dst, packed, blocks):
        if native is not None:
            st = native.ggo_dequantize_slice_mt(ty, out_ty, dst.ctypes.data, dst.size, packed.ctypes.data, blocks, cores)
        else:
            st = o.dequantize_status(ty, out_ty, dst, dst.size, packed, blocks, threads=cores)
        assert st == 0

    if native is not None:
        # the native build must give the shipped build's bits before it may stand in for it
        for ty, packed, dst, blocks in bufs[:4]:
            nb = min(blocks, 4096)
            a = np.empty(nb * block_elems(ty), dtype=dst.dtype)
            b = np.empty_like(a)
            native.ggo_dequantize_slice_mt(ty, out_ty, a.ctypes.data, a.size, packed.ctypes.data, nb, 1)
            o.dequantize_status(ty, out_ty, b, b.size, packed, nb)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), "native oracle build differs from the shipped one"

    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 10):
        t0 = time.perf_counter()
        for ty, packed, dst, blocks in bufs:
            decode(ty, dst, packed, blocks)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    fresh = []
    for _ in range(3):
        t0 = time.perf_counter()
        for ty, packed, dst, blocks in bufs:
            m = mmap.mmap(-1, dst.nbytes)
            f = np.frombuffer(m, dtype=dst.dtype)
            decode(ty, f, packed, blocks)
            del f
            m.close()
        fresh.append(time.perf_counter() - t0)
    fresh_s = statistics.median(fresh)
    types = sorted({ty for ty, _, _, _ in bufs})
    return {
        "value": round(elems * ob / med / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
        "melem_per_s": round(elems / med / 1e6, 1),
        "best_value": round(elems * ob / min(times) / 1e9, 3),
        "fresh_mapping_value": round(elems * ob / fresh_s / 1e9, 3),
        "build": flags,
        "sample": f"first {len(sample)} tensors of the workload ({elems} elements, ggml types {types}), random "
                  f"non-repeating input, pre-faulted output, median of {len(times)} passes; fresh_mapping_value: "
                  f"output in a fresh anonymous mapping per tensor, median of {len(fresh)} passes",
    }


# --------------------------------------------------------------------------------------
# GPU legs
# --------------------------------------------------------------------------------------
def bind_near_gpu(index):
    """N > 1: run this rank on the CPUs of the NUMA node its GPU hangs off, before any page-locked memory is
    allocated (first touch places it there), so that the streamed leg's H2D reads do not cross sockets.
    Best effort: returns a note for the JSON line, changes nothing when the topology cannot be read."""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return f"gpu {bdf}: no NUMA node reported"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        mine = cpus & os.sched_getaffinity(0)
        if not mine:
            return f"gpu {bdf}: node {node} has none of this process's CPUs"
        os.sched_setaffinity(0, mine)
        return f"gpu {bdf}: bound to NUMA node {node} ({len(mine)} CPUs)"
    except Exception as e:
        return f"not bound ({type(e).__name__})"



class Share:
    """One rank's pieces of the job: packed blocks and decoded output resident in HBM, one launch
    plan per block type (largest first)."""

    def __init__(self, pieces, out_name, dev, seed_base, arena_bytes=None, typed=True):
        import torch
        import gguf_amd as g
        from gguf_amd import synth
        from gguf_amd.quants import GGmlType, block_bytes
        self.out_name = out_name
        self.ob = 4 if out_name == "f32" else 2
        out_dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[out_name]
        self.out_ty = {"f32": GGmlType.F32, "f16": GGmlType.F16, "bf16": GGmlType.BF16}[out_name]
        self.pieces = pieces
        self.elems = sum(p.elems for p in pieces)
        self.packed_bytes = sum(p.blocks * block_bytes(p.ggml_type) for p in pieces)
        need = max(self.elems * self.ob, arena_bytes or 0) + self.packed_bytes
        free = torch.cuda.mem_get_info(dev)[0]
        if need > 0.97 * free:
            raise MemoryError(f"this share needs {need / 2**30:.1f} GiB of HBM, {free / 2**30:.1f} GiB are free; "
                              f"use more GPUs (strong scaling) or --out f16")
        # decoded pieces in share order, each 256-byte aligned, in one arena (gguf_amd.shard.arena_layout);
        # `arena_bytes` pads it to the largest share of the job when the shares are to be gathered
        offs, at = [], 0
        for p in pieces:
            offs.append(at)
            at += (p.elems * self.ob + 255) & ~255
        self.arena = torch.empty(max(at, arena_bytes or 0), dtype=torch.uint8, device=dev)
        self.dst_all = self.arena[:at].view(out_dt)
        self.by_type = {}
        # typed layout: one packed arena per block type, the type's pieces back to back in share order (the caller
        # passes the pieces sorted by type, so their outputs in `arena` are back to back as well)
        packed_arena, packed_at = {}, {}
        if typed:
            for p in pieces:
                packed_at[int(p.ggml_type)] = packed_at.get(int(p.ggml_type), 0) + p.blocks * block_bytes(p.ggml_type)
            packed_arena = {ty: torch.empty(n, dtype=torch.uint8, device=dev) for ty, n in packed_at.items()}
            packed_at = {ty: 0 for ty in packed_at}
        for i, (p, o) in enumerate(zip(pieces, offs)):
            src = synth.random_blocks_device(p.ggml_type, p.blocks, seed_base + i, dev)
            if typed:
                ty, n = int(p.ggml_type), src.numel()
                slot = packed_arena[ty][packed_at[ty]:packed_at[ty] + n]
                slot.copy_(src)
                packed_at[ty] += n
                src = slot
            dst = self.arena[o:o + p.elems * self.ob].view(out_dt)
            self.by_type.setdefault(int(p.ggml_type), []).append((p.ggml_type, src, dst))
        self.order = sorted(self.by_type, key=lambda t: -sum(x[2].numel() for x in self.by_type[t]))
        self.plans = [(ty, g.Batch(self.by_type[ty], self.out_ty)) for ty in self.order]
        self.launches = sum(b.launches for _, b in self.plans)

    def step(self, stream):
        for _, b in self.plans:
            b.run(stream)

    def poison(self):
        import torch
        if self.ob == 2:
            self.dst_all.view(torch.int16).fill_(0x7FC1)
        else:
            self.dst_all.view(torch.int32).fill_(0x7FC00001)

    def verify(self, seed):
        """Random blocks of every launch plan and their outputs against the CPU oracle, bit for
        bit (a NaN matches any NaN).  Returns (blocks checked, ok)."""
        import numpy as np
        import oracle as o
        from gguf_amd import synth
        o.build()
        o_out = {"f32": o.F32, "f16": o.F16, "bf16": o.BF16}[self.out_name]
        checked, ok = 0, True
        for ty in self.order:
            packed, got = synth.sample_blocks(self.by_type[ty], VERIFY_BLOCKS, seed + ty)
            want = o.dequantize(ty, packed, o_out, threads=8)
            if self.ob == 4:
                same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            else:
                same = got == want        # benign blocks give no NaN
            ok = ok and bool(same.all())
            checked += VERIFY_BLOCKS
        return checked, ok

    def close(self):
        for _, b in self.plans:
            b.close()
        self.plans = []


def timed_resident(share, steps, warmup, settle_ms, fence, with_events=True):
    """W warm-up steps, then exactly `steps` timed steps between fences.  Returns (seconds on this
    rank, per-plan average launch ms or None)."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    # The GPU reaches its steady clocks only after some tens of milliseconds of load (measured:
    # the same launch runs 9 % slower in the first few milliseconds after an idle period), so
    # untimed steps run until `settle_ms` of wall time have passed, then the W warm-up steps.
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        share.step(stream)
        torch.cuda.synchronize()
    for _ in range(warmup):
        share.step(stream)
    fence()
    # HIP events on the launch stream around every kernel, inside the timed region.  An event
    # between two launches costs the GPU a few microseconds of idle time (measured: 2 % of a
    # 4.7 ms step, 10 % of a 0.85 ms one), so only every `every`-th step carries them; the
    # kernel durations reported are the averages over those steps.
    every = max(1, steps // 5)
    sampled = [s for s in range(steps) if s % every == 0] if with_events else []
    nplans = len(share.plans)
    ev = {s: [torch.cuda.Event(enable_timing=True) for _ in range(nplans + 1)] for s in sampled}
    t0 = time.perf_counter()
    for s in range(steps):
        if s in ev:
            ev[s][0].record()
            for k, (_, b) in enumerate(share.plans):
                b.run(stream)
                ev[s][k + 1].record()
        else:
            share.step(stream)
    # this rank's clock stops when ITS GPU has drained; the barrier that follows is not part of anybody's time (the
    # caller takes the maximum over the ranks, i.e. the time of the slowest one)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    kern_ms = None
    if sampled:
        kern_ms = [sum(ev[s][k].elapsed_time(ev[s][k + 1]) for s in sampled) / len(sampled) for k in range(nplans)]
    return elapsed, kern_ms, len(sampled)


def streamed_leg(share, local, steps, fence, max_over_ranks):
    """The same share decoded from page-locked host memory: every piece goes through
    ggq_dequantize_async (chunked H2D copy and decode kernel overlapping on the context's two
    streams), the decoded tensors stay in HBM.  One step = the whole share once."""
    import torch
    import gguf_amd as g
    staging = 128 << 20
    arena = torch.empty(share.packed_bytes, dtype=torch.uint8, pin_memory=True)
    work, at = [], 0
    for ty in share.order:
        for (t, src, dst) in share.by_type[ty]:
            n = src.numel()
            arena[at:at + n].copy_(src)
            work.append((t, dst, arena[at:at + n]))
            at += n
    torch.cuda.synchronize()
    ctx = g.Context(local, staging_bytes=staging)

    def step():
        for t, dst, h in work:
            ctx.dequantize_async(t, dst, h)
        ctx.sync()

    share.poison()
    step()                                  # warm-up (also the pass that gets verified)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()                              # ends with ctx.sync(): this rank's copies and decodes are done
    elapsed = time.perf_counter() - t0
    fence()
    elapsed = max_over_ranks(elapsed)
    ctx.close()
    return elapsed, staging


def self_launch(n):
    """Plain `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as fresh child
    processes (same command line; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), wait for them, pass rank 0's
    standard output (the one JSON line) through, and return the exit code: rank 0's, or the first non-zero one.
    This process imports neither torch nor the library and makes no HIP call, so every child initialises the GPU
    in a process of its own.  A rank that fails takes the others down after a grace period, so that nobody waits
    in a barrier for it until gloo's timeout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GGQ_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    codes = [None] * n
    failed_at = None
    out0 = None
    try:
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None and p.poll() is not None:
                    codes[r] = p.returncode
                    if r == 0:
                        out0 = p.stdout.read()
                    if p.returncode != 0 and failed_at is None:
                        failed_at = time.perf_counter()
            if failed_at is not None and time.perf_counter() - failed_at > 20.0:
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.terminate()
                failed_at = time.perf_counter() + 1e9          # once
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return codes[0] if codes[0] else bad[0][1]
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    affinity0 = os.sched_getaffinity(0)
    import torch
    import torch.distributed as dist
    import gguf_amd as g  # noqa: F401
    from gguf_amd import shard, synth
    from gguf_amd.quants import GGmlType, block_bytes, block_elems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world} set by the launcher (plain `python bench.py "
                 f"--gpus N` without WORLD_SIZE starts its own ranks; under torch.distributed.run pass "
                 f"--nproc-per-node N and --gpus N)")
    if not torch.cuda.is_available():
        sys.exit(f"rank {rank}: bench.py needs a GPU (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    # Normal case: one rank per GPU.  Rehearsal case (fewer GPUs than ranks, e.g. two ranks on a
    # one-GPU box): ranks share devices.  Either way the process group is gloo on host tensors: it
    # carries nothing but barriers and the max of the times, the data path has no exchange step.
    shared = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    numa_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        numa_note = bind_near_gpu(local)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(blocks, ok):
        if world == 1:
            return blocks, ok
        t = torch.tensor([blocks, 0 if ok else 1], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t[0]), int(t[1]) == 0

    # how many ranks actually took part, and on how many distinct GPUs (counted, not assumed)
    ranks_seen, gpus_seen = 1, 1
    if world > 1:
        ids = [None] * world
        p = torch.cuda.get_device_properties(local)
        dist.all_gather_object(ids, (rank, f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"))
        ranks_seen, gpus_seen = len({i[0] for i in ids}), len({i[1] for i in ids})

    # ---- the job, dealt to ranks by the index partition -----------------------------
    base, items, label = job_items(args.workload, args.scaling, world)
    shares = shard.partition(items, world, out_bytes=4 if args.out == "f32" else 2)
    typed = args.layout == "typed"
    if typed:
        shares = [shard.by_type(s) for s in shares]
    gathering = args.gather > 0 and world > 1
    where, padded = shard.arena_layout(shares, 4 if args.out == "f32" else 2)
    try:
        share = Share(shares[rank], args.out, dev, 0xC0FFEE04 + 100000 * rank, padded if gathering else None, typed)
    except MemoryError as e:
        sys.exit(f"rank {rank}: {e}")
    ob = share.ob

    elapsed, kern_ms, timed_launches = timed_resident(share, args.steps, args.warmup, args.settle_ms, fence)
    elapsed = max_over_ranks(elapsed)

    verified = None
    if not args.no_verify:
        share.poison()
        share.step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        blocks, ok = share.verify(4242 + rank)
        blocks, ok = all_ok(blocks, ok)
        verified = {"blocks": blocks, "ok": ok, "how": f"{VERIFY_BLOCKS} random blocks per launch plan and rank, "
                    "outputs of one more (untimed) step after the timed region, bit for bit against the CPU oracle"}

    streamed = None
    if args.streamed_steps > 0:
        try:
            s_elapsed, staging = streamed_leg(share, local, args.streamed_steps, fence, max_over_ranks)
            s_blocks, s_ok = (0, True) if args.no_verify else share.verify(777 + rank)
            s_blocks, s_ok = all_ok(s_blocks, s_ok)
            pin, pout, nel = synth.workload_bytes(items, ob)
            per = s_elapsed / args.streamed_steps
            streamed = {
                "steps": args.streamed_steps, "ms_per_step": round(per * 1e3, 3),
                "end_to_end_GBps": round(pout / per / 1e9, 2), "packed_pcie_GBps": round(pin / per / 1e9, 2),
                "packed_pcie_GBps_per_gpu": round(pin / per / 1e9 / world, 2),
                "melem_per_s": round(nel / per / 1e6, 1), "bound": "pcie (H2D)",
                "path": "ggq_dequantize_async per tensor piece: page-locked host source read in place, "
                        f"{staging >> 20} MiB device staging in 2 slots, copy and decode on 2 streams; decoded tensors stay in HBM",
                "verified": None if args.no_verify else {"blocks": s_blocks, "ok": s_ok},
            }
            if numa_note:
                streamed["rank0_cpu_binding"] = numa_note
        except Exception as e:          # never lose the main line to the secondary leg
            streamed = {"error": f"{type(e).__name__}: {e}"}
        if world > 1:
            dist.barrier()

    # opt-in: every rank ends with every rank's decoded share (one large all-gather)
    gather = None
    if gathering:
        try:
            share.poison()
            share.step(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if shared:      # rehearsal: ranks share a GPU, which RCCL refuses; host copies over gloo carry the logic
                arena, group, how = share.arena.cpu(), None, "gloo on host copies (REHEARSAL)"
            else:
                arena, group = share.arena, dist.new_group(backend="nccl")
                how = f"RCCL all_gather_into_tensor over xGMI, communicator of {dist.get_world_size(group)} ranks"
            full = shard.gather_arenas(arena, world, group)          # warm-up: communicator set-up
            fence()
            t0 = time.perf_counter()
            for _ in range(args.gather):
                full = shard.gather_arenas(arena, world, group)
            torch.cuda.synchronize()
            g_dt = time.perf_counter() - t0
            fence()
            g_elapsed = max_over_ranks(g_dt) / args.gather
            # every rank's row must be that rank's arena: compare 64-bit sums of the rows with the owners' own
            mine = share.arena.view(torch.int64).sum().cpu().reshape(1)
            sums = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(sums, mine)
            rows = full.view(world, -1).view(torch.int64).sum(dim=1).cpu()
            g_ok = all(int(rows[r]) == int(sums[r]) for r in range(world))
            _, g_ok = all_ok(0, g_ok)
            gather = {"collective": how, "steps": args.gather, "ms": round(g_elapsed * 1e3, 3),
                      "arena_bytes_per_rank": int(padded), "received_GBps_per_gpu": round((world - 1) * padded / g_elapsed / 1e9, 2),
                      "whole_job_GBps": round(world * (world - 1) * padded / g_elapsed / 1e9, 2), "verified": g_ok}
            del full
        except Exception as e:
            gather = {"error": f"{type(e).__name__}: {e}"}
        if world > 1:
            dist.barrier()

    order = share.order
    dom_elems_by_type = {ty: sum(x[2].numel() for x in share.by_type[ty]) for ty in share.order}
    launches = share.launches
    plan_launches = [b.launches for _, b in share.plans]
    dom_name = GGmlType(share.order[0]).name
    dom_table, dom_all = share.plans[0][1].table_launches, share.plans[0][1].launches
    if dom_all == 1:
        dom_kernel = (f"deq_batch<{dom_name},{args.out}>" if dom_table else
                      f"deq_one<{dom_name},{args.out}> over the type's tensors as one run") + " (rank 0)"
    else:
        dom_kernel = (f"the {dom_name} -> {args.out} launch plan (rank 0): {dom_table} deq_batch launch over the tile table + "
                      f"{dom_all - dom_table} deq_one launches for runs of 64 Mi elements and more")

    # secondary: weak scaling (every rank one whole set), when the headline is strong scaling
    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_secondary:
        try:
            share.close()
            share = None
            torch.cuda.empty_cache()
            _, w_items, w_label = job_items(args.workload, "weak", world)
            w_pieces = shard.partition(w_items, world, out_bytes=ob)[rank]
            w_share = Share(shard.by_type(w_pieces) if typed else w_pieces, args.out, dev,
                            0xC0FFEE04 + 100000 * rank, None, typed)
            w_elapsed, _, _ = timed_resident(w_share, args.steps, args.warmup, args.settle_ms, fence, with_events=False)
            w_elapsed = max_over_ranks(w_elapsed)
            _, w_pout, w_nel = synth.workload_bytes(w_items, ob)
            weak = {"scaling": "weak", "workload": w_label, "value": round(w_pout / (w_elapsed / args.steps) / 1e9, 2),
                    "unit": "GB/s", "ms_per_step": round(w_elapsed / args.steps * 1e3, 4),
                    "melem_per_s": round(w_nel / (w_elapsed / args.steps) / 1e6, 1)}
            w_share.close()
            del w_share
        except Exception as e:
            weak = {"error": f"{type(e).__name__}: {e}"}

    rc = 0
    if rank == 0:
        total_elems = sum(ne for _, _, ne in items)
        pin, pout, _ = synth.workload_bytes(items, ob)
        per = elapsed / args.steps
        dom_ty = order[0]
        dom_elems = dom_elems_by_type[dom_ty]
        per_elem = block_bytes(dom_ty) / block_elems(dom_ty) + ob
        dom_bytes = dom_elems * per_elem
        achieved = dom_bytes / (kern_ms[0] * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path))
                key = f"{args.workload}:{args.out}:{'' if typed else 'scattered:'}{GGmlType(dom_ty).name}"
                if key in tr and world == 1:
                    traffic = tr[key].get("hbm_bytes_per_launch")
                    traffic_source = ("static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch recorded in "
                                      f"profiles/traffic.json ({tr[key].get('round', 'r01')}), not measured in this run")
                elif key in tr:
                    # a rank's launch covers 1/N of the set: the counters were read on the whole-set launch, so only
                    # their ratio to that launch's algorithmic bytes carries over, not the byte count
                    whole = sum(ne for _, ty, ne in base if int(ty) == int(dom_ty)) * per_elem
                    ratio = tr[key]["hbm_bytes_per_launch"] / whole
                    traffic_source = (f"none for this launch size; the one-GPU launch of the same kernel moved {ratio:.3f} x "
                                      f"its algorithmic bytes (profiles/traffic.json, {tr[key].get('round', 'r01')})")
            except Exception:
                traffic = None
        rec = {
            "metric": "dequantized_GBps", "value": round(pout / per / 1e9, 2), "unit": "GB/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "gpus_seen": gpus_seen,
            "launcher": "self (bench.py started the ranks)" if os.environ.get("GGQ_BENCH_SELF_LAUNCHED") else
                        ("external" if world > 1 else "none"),
            "steps": args.steps, "warmup": args.warmup, "settle_ms": args.settle_ms,
            "ms_per_step": round(per * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "io": "u8->" + args.out, "data": "synthetic",
            "config": {"workload": label + (f" (REHEARSAL: {world} ranks share {ndev} GPU)" if shared else ""),
                       "tensors": len(items), "elements": total_elems, "packed_bytes": pin,
                       "decoded_bytes": pout, "partition": "index (LPT), no collective",
                       "layout": "typed: per rank, the packed tensors of a block type back to back in HBM and their outputs "
                                 "likewise (one run per type)" if typed else "scattered: one allocation per packed tensor",
                       "kernel_launches_per_step": launches},
            "melem_per_s": round(total_elems / per / 1e6, 1),
            "packed_GBps": round(pin / per / 1e9, 2),
            "algorithmic_GBps": round((pin + pout) / per / 1e9, 2),
            "kernels": [{"type": GGmlType(ty).name, "elements": dom_elems_by_type[ty], "launches": plan_launches[k],
                         "avg_ms": round(kern_ms[k], 4)} for k, ty in enumerate(order)],
            "roofline": {"bound": "hbm", "kernel": dom_kernel,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(kern_ms[0], 4),
                         "timed_launches": timed_launches},
        }
        if verified is not None:
            rec["verified"] = verified
            if not verified["ok"]:
                rc = 1
        if streamed is not None:
            rec["streamed"] = streamed
            if streamed.get("verified") and not streamed["verified"]["ok"]:
                rc = 1
        if gather is not None:
            rec["gather"] = gather
            if gather.get("verified") is False:
                rc = 1
        if weak is not None:
            rec["secondary"] = weak
        if not args.no_cpu_baseline:
            if share is not None:
                share.close()
                share = None
            torch.cuda.empty_cache()
            # rank 0 alone, after every GPU leg, on all the CPUs this process started with (the other ranks wait in
            # the barrier below; bind_near_gpu had narrowed the affinity for the streamed leg)
            os.sched_setaffinity(0, affinity0)
            if world > 1 and os.environ.get("OMP_NUM_THREADS") == "1":
                os.environ.pop("OMP_NUM_THREADS")       # torch.distributed.run's per-rank default, not a core count
            rec["cpu_baseline"] = cpu_baseline(base, args.out, args.cpu_seconds)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        sys.exit("bench.py: the outputs of the timed kernels differ from the CPU oracle")


if __name__ == "__main__":
    main()
