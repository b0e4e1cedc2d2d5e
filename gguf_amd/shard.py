"""Index partition of a tensor table over the GPUs of one node.

Blocks are independent, so whole-model conversion needs no exchange step: every
rank derives the same partition from the same table and decodes only its share.
No collective takes part in the data path.

Rule: pieces are sorted by decoded size and dealt to the least-loaded rank
(longest-processing-time first); a tensor whose decoded size exceeds a quarter
of a rank's fair share is first cut into contiguous block ranges that end on row
boundaries (a row = shape[0] elements, the contiguous dimension) -- the same way
the reference plans output shards by accumulated tensor size
(xtask/src/utils/write.rs:23-51), but with a balancing deal instead of a running
cut, because here the shares run concurrently.
"""
from dataclasses import dataclass

from .quants import block_bytes, block_elems


@dataclass(frozen=True)
class Piece:
    tensor: int        # index into the table
    name: str
    ggml_type: int
    block_begin: int   # first block of the tensor in this piece
    blocks: int

    @property
    def elems(self):
        return self.blocks * block_elems(self.ggml_type)

    @property
    def packed_bytes(self):
        return self.blocks * block_bytes(self.ggml_type)


def partition(items, world_size, row_elems=None, out_bytes=4):
    """items: [(name, ggml_type, elements)]; returns a list of Piece lists, one per rank.

    row_elems: optional list with each tensor's contiguous-dimension length, so that
    cuts fall on row boundaries; default is one block per row (any block boundary).
    """
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    cost = lambda ty, blocks: blocks * (block_elems(ty) * out_bytes + block_bytes(ty))
    total = 0
    table = []
    for i, (name, ty, n) in enumerate(items):
        be = block_elems(ty)
        if be == 0:
            raise ValueError(f"{name}: unsupported type {ty}")
        if n % be:
            raise ValueError(f"{name}: {n} elements is not a whole number of blocks")
        row = row_elems[i] if row_elems else be
        if row % be or n % row:
            raise ValueError(f"{name}: row of {row} elements does not tile the tensor")
        table.append((i, name, int(ty), n // be, row // be))
        total += cost(ty, n // be)
    limit = max(1, total // world_size // 4)

    pieces = []
    for i, name, ty, blocks, row_blocks in table:
        c = cost(ty, blocks)
        parts = 1 if world_size == 1 else min(-(-c // limit), blocks // row_blocks)
        parts = max(1, parts)
        rows = blocks // row_blocks
        at = 0
        for p in range(parts):
            r = rows // parts + (1 if p < rows % parts else 0)
            if r == 0:
                continue
            pieces.append(Piece(i, name, ty, at, r * row_blocks))
            at += r * row_blocks
        assert at == blocks

    order = sorted(range(len(pieces)), key=lambda k: (-cost(pieces[k].ggml_type, pieces[k].blocks), k))
    load = [0] * world_size
    shares = [[] for _ in range(world_size)]
    for k in order:
        r = min(range(world_size), key=lambda j: (load[j], j))
        shares[r].append(pieces[k])
        load[r] += cost(pieces[k].ggml_type, pieces[k].blocks)
    for s in shares:
        s.sort(key=lambda p: (p.tensor, p.block_begin))
    return shares


def by_type(share):
    """A share's pieces with those of one block type next to each other (types in order of first appearance,
    tensor order kept inside a type).  A rank that lays its packed pieces and its outputs out in THIS order has
    every type's blocks and every type's outputs contiguous in memory, which a launch plan decodes as one long
    tensor per type (ggq_batch_create merges adjacent tensors into runs)."""
    first = {}
    for p in share:
        first.setdefault(p.ggml_type, len(first))
    return sorted(share, key=lambda p: (first[p.ggml_type], p.tensor, p.block_begin))


def imbalance(shares, out_bytes=4):
    """max rank load / mean rank load (1.0 = perfect)."""
    loads = [sum(p.blocks * (block_elems(p.ggml_type) * out_bytes + block_bytes(p.ggml_type)) for p in s)
             for s in shares]
    mean = sum(loads) / len(loads)
    return max(loads) / mean if mean else 1.0


# ---- opt-in gather ---------------------------------------------------------------------
# The decode needs no exchange; a caller that wants every rank to hold every decoded tensor asks for
# it explicitly.  Each rank keeps its pieces, in share order, in ONE contiguous arena padded to the
# largest share, so the whole exchange is a single large all-gather (over RCCL / xGMI when the group
# is an nccl group and the arenas are device tensors; xGMI is point to point, a ring all-gather is
# bound by one link per GPU, ~150 GB/s, some 40 times below the decode rate -- hence opt-in).

def arena_layout(shares, out_bytes=4):
    """Where every piece lives after a gather: ({(tensor, block_begin): (rank, byte offset, bytes)},
    padded arena bytes per rank).  Pieces sit in share order, each aligned to 256 bytes."""
    where, sizes = {}, []
    for r, s in enumerate(shares):
        at = 0
        for p in s:
            n = p.elems * out_bytes
            where[(p.tensor, p.block_begin)] = (r, at, n)
            at += (n + 255) & ~255
        sizes.append(at)
    return where, (max(sizes) + 255) & ~255 if sizes else 0


def gather_arenas(arena, world, group=None):
    """All-gather of the ranks' padded arenas: returns a [world, arena.numel()] tensor that holds rank
    r's arena in row r, on every rank.  `arena`: this rank's uint8 arena, the same (padded) length on
    every rank; device tensors need an nccl group, host tensors a gloo group."""
    import torch
    import torch.distributed as dist
    if arena.dtype != torch.uint8 or arena.dim() != 1:
        raise ValueError("arena must be a flat uint8 tensor")
    full = torch.empty((world, arena.numel()), dtype=torch.uint8, device=arena.device)
    dist.all_gather_into_tensor(full.view(-1), arena, group=group)
    return full
