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
