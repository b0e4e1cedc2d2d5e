"""Index partition over ranks (no collective in the data path), including a
world_size-2 run over gloo on the CPU."""
import os
import socket

import numpy as np
import pytest

from gguf_amd import shard, synth
from gguf_amd.quants import GGmlType, block_elems


@pytest.mark.parametrize("ws", [1, 2, 4, 8])
@pytest.mark.parametrize("wl", ["llama7b_q4_k_m", "llama70b_q4_k_m_iq4nl", "q4_0_4096"])
def test_partition_covers_everything_once(ws, wl):
    items = synth.workload(wl)
    shares = shard.partition(items, ws)
    assert len(shares) == ws
    seen = {}
    for s in shares:
        for p in s:
            seen.setdefault(p.tensor, []).append((p.block_begin, p.blocks))
    for i, (name, ty, n) in enumerate(items):
        ranges = sorted(seen[i])
        at = 0
        for b, c in ranges:
            assert b == at and c > 0
            at += c
        assert at == n // block_elems(ty)
    if len(items) > 8 * ws:
        assert shard.imbalance(shares) < 1.05


def test_partition_cuts_on_rows():
    items = [("w", GGmlType.Q4K, 4096 * 64)]
    shares = shard.partition(items, 4, row_elems=[4096])
    rows = 4096 // 256
    for s in shares:
        for p in s:
            assert p.block_begin % rows == 0 and p.blocks % rows == 0


def test_partition_rejects_bad_input():
    with pytest.raises(ValueError):
        shard.partition([("w", GGmlType.Q4K, 100)], 2)
    with pytest.raises(ValueError):
        shard.partition([("w", GGmlType.IQ2XXS, 256)], 2)
    with pytest.raises(ValueError):
        shard.partition([("w", GGmlType.Q4K, 256)], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, ws, port, q):
    import torch.distributed as dist
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import oracle as o     # the checker: stands in for the kernels on a GPU-less box
        items = [(f"t{i}", ty, n) for i, (ty, n) in enumerate(
            [(GGmlType.Q4K, 256 * 300), (GGmlType.Q6K, 256 * 77), (GGmlType.Q4_0, 32 * 1000),
             (GGmlType.Q4K, 256 * 5), (GGmlType.Q8_0, 32 * 64)])]
        mine = shard.partition(items, ws)[rank]
        # every rank derives the same inputs from the same seeds, decodes only its pieces
        total = 0
        digest = 0
        for p in mine:
            full = synth.random_blocks(p.ggml_type, items[p.tensor][2] // block_elems(p.ggml_type), 100 + p.tensor)
            bb = o.block_bytes(p.ggml_type)
            part = full[p.block_begin * bb:(p.block_begin + p.blocks) * bb]
            y = o.dequantize(p.ggml_type, part)
            total += y.size
            digest += int(y.view(np.uint32).astype(np.uint64).sum())
        t = torch.tensor([total, digest % (2 ** 62)], dtype=torch.int64)
        dist.all_reduce(t)       # bookkeeping only: the data path has no collective
        q.put((rank, int(t[0]), int(t[1])))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_cover_the_model(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want_elems = 256 * 300 + 256 * 77 + 32 * 1000 + 256 * 5 + 32 * 64
    # single-process digest of the whole model
    digest = 0
    tys = [GGmlType.Q4K, GGmlType.Q6K, GGmlType.Q4_0, GGmlType.Q4K, GGmlType.Q8_0]
    ns = [256 * 300, 256 * 77, 32 * 1000, 256 * 5, 32 * 64]
    for i, (ty, n) in enumerate(zip(tys, ns)):
        y = oracle.dequantize(ty, synth.random_blocks(ty, n // block_elems(ty), 100 + i))
        digest += int(y.view(np.uint32).astype(np.uint64).sum())
    for _, total, dg in res:
        assert total == want_elems
        assert dg == digest % (2 ** 62)


def _gather_main(rank, ws, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        items = [(f"t{i}", ty, n) for i, (ty, n) in enumerate(
            [(GGmlType.Q4K, 256 * 300), (GGmlType.Q6K, 256 * 77), (GGmlType.Q4_0, 32 * 1000),
             (GGmlType.Q4K, 256 * 5), (GGmlType.Q8_0, 32 * 64)])]
        shares = shard.partition(items, ws)
        where, padded = shard.arena_layout(shares, 4)
        # stand-in for the decode: element k of tensor t is the float 1000 * t + (k % 997)
        arena = torch.zeros(padded, dtype=torch.uint8)
        for p in shares[rank]:
            r, off, n = where[(p.tensor, p.block_begin)]
            assert r == rank
            first = p.block_begin * block_elems(p.ggml_type)
            vals = (torch.arange(first, first + p.elems) % 997 + 1000 * p.tensor).to(torch.float32)
            arena[off:off + n] = vals.view(torch.uint8)
        full = shard.gather_arenas(arena, ws)
        ok = True
        for (t, begin), (r, off, n) in where.items():
            ty = items[t][1]
            first = begin * block_elems(ty)
            got = full[r, off:off + n].view(torch.float32)
            want = (torch.arange(first, first + n // 4) % 997 + 1000 * t).to(torch.float32)
            ok = ok and bool(torch.equal(got, want))
        q.put((rank, ok, len(where)))
    finally:
        dist.destroy_process_group()


def test_opt_in_gather_two_ranks_gloo():
    """After the opt-in gather every rank can find every piece of every tensor (host tensors over gloo here;
    device tensors over an nccl group take the same call)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and all(n >= 5 for _, _, n in res)


def test_arena_layout_is_disjoint_and_aligned():
    items = synth.workload("llama7b_q4_k_m")
    shares = shard.partition(items, 8)
    where, padded = shard.arena_layout(shares, 4)
    assert padded % 256 == 0
    per_rank = {}
    for (t, b), (r, off, n) in where.items():
        assert off % 256 == 0 and off + n <= padded
        per_rank.setdefault(r, []).append((off, n))
    for r, spans in per_rank.items():
        spans.sort()
        for (a, n), (b, _) in zip(spans, spans[1:]):
            assert a + n <= b
    assert sum(n for _, _, n in where.values()) == 6_738_149_376 * 4
