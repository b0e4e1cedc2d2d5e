import numpy as np


def same_bits(a, b):
    """Bit-exact equality, except that a NaN matches any NaN."""
    a = np.ascontiguousarray(a).reshape(-1)
    b = np.ascontiguousarray(b).reshape(-1)
    assert a.dtype == b.dtype and a.size == b.size, (a.dtype, b.dtype, a.size, b.size)
    if a.dtype == np.float32:
        ua, ub = a.view(np.uint32), b.view(np.uint32)
        na, nb = np.isnan(a), np.isnan(b)
    else:  # uint16 bit patterns of f16 or bf16: caller passes nan masks through kind
        raise TypeError("use same_bits16 for 16-bit patterns")
    return bool(np.all((ua == ub) | (na & nb)))


def nan16(bits, kind):
    bits = bits.astype(np.uint32)
    if kind == "f16":
        return ((bits & 0x7C00) == 0x7C00) & ((bits & 0x03FF) != 0)
    return ((bits & 0x7F80) == 0x7F80) & ((bits & 0x007F) != 0)


def same_bits16(a, b, kind):
    a = np.ascontiguousarray(a).reshape(-1).view(np.uint16)
    b = np.ascontiguousarray(b).reshape(-1).view(np.uint16)
    assert a.size == b.size
    return bool(np.all((a == b) | (nan16(a, kind) & nan16(b, kind))))


def first_diff(a, b):
    a = np.ascontiguousarray(a).reshape(-1)
    b = np.ascontiguousarray(b).reshape(-1)
    ua = a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint16)
    ub = b.view(ua.dtype)
    idx = np.nonzero(ua != ub)[0]
    if idx.size == 0:
        return "no raw-bit difference"
    i = int(idx[0])
    return f"{idx.size} differing elements; first at {i}: {a[i]!r} ({ua[i]:#x}) vs {b[i]!r} ({ub[i]:#x})"


def plan_entries(tensors, ob, forced=None, min_elems=64 << 20):
    """The routing rule of ggq_batch_create restated: how many entries (one table launch per type with several
    runs in the table, launches of their own, widenings) a plan with `ob`-byte outputs has.  tensors: [(type, elements)] or [(type, elements,
    adjacent)], adjacent = this tensor's blocks and outputs start where those of the previous tensor of its type
    end (such tensors form one run).  forced: the GGQ_PREFETCH triple if that variable is set; min_elems mirrors
    GGQ_HYBRID_MIN_ELEMS (default = the library's)."""
    def grouped(ty, n):
        if forced is not None:
            return len(forced) == 3 and forced[2] > 0
        # schedule.h: the types whose single-tensor schedule has a group start, which it has from 64 Mi elements
        return n >= 64 << 20 and (ty not in (10, 11) if ob == 2 else ty in (8, 9, 15))
    entries, runs, last = 0, [], {}
    for t in tensors:
        ty, n, adj = (t + (False,))[:3]
        if n == 0:
            continue
        if ty in (1, 30):
            entries += 1
        elif adj and ty in last:
            runs[last[ty]][1] += n
        else:
            last[ty] = len(runs)
            runs.append([ty, n])
    table = set()
    for ty, n in runs:
        only = sum(1 for t2, _ in runs if t2 == ty) == 1           # a type's only run needs no table
        if min_elems and (only or (grouped(ty, n) and n >= min_elems)):
            entries += 1
        else:
            table.add(ty)
    return entries + len(table)


def with_adjacency(tensors):
    """[(type, src, dst)] of device tensors -> [(type, elements, adjacent)] for plan_entries: adjacent = src and
    dst both start exactly where those of the previous tensor of the same type end (torch's allocator can place
    separate allocations back to back, and the library merges whatever is adjacent)."""
    out, last = [], {}
    for ty, src, dst in tensors:
        ty = int(ty)
        s0, d0 = src.data_ptr(), dst.data_ptr()
        adj = ty in last and last[ty] == (s0, d0)
        out.append((ty, dst.numel(), adj))
        if dst.numel():
            last[ty] = (s0 + src.numel() * src.element_size(), d0 + dst.numel() * dst.element_size())
    return out
