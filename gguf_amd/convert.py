"""The reference's `convert` pipeline, with the per-tensor work on the GPU:

    read    Content::new / merge_file        xtask/src/utils/read.rs:4-63
    cast    Operator::cast, Content::cast    xtask/src/utils/operator/cast.rs:11-138, 180-216
    merge   Content::merge_linear            xtask/src/utils/operator/merge.rs:22-268
    permute Content::permute_qk              xtask/src/utils/operator/permute_qk.rs:12-69
    scale   Content::convert_to_llama        xtask/src/utils/operator/to_llama.rs:26-125
    filter, sort, set-meta (host only)       xtask/src/utils/operator/mod.rs:58-71, sort.rs, set_meta.rs
    steps   the `-x` grammar                 xtask/src/convert.rs:38-51
    write   Content::write_files             xtask/src/utils/write.rs:5-100

Input shards are mapped, not read; a tensor that changes type is produced only when the writer of
its shard reaches it (the reference's lazy DataPromise): packed bytes go to the device, are decoded /
re-encoded / rearranged there, and come back through a pinned buffer that the C++ writer hands to write(2).  One
producer thread per shard keeps the GPU one tensor ahead of the file.  Output file names are the
caller's business (`stem`); the reference's name grammar is not part of this library.
"""
import functools
import math
import os
import queue
import re
import struct
import threading

import numpy as np

from ._lib import lib
from .gguf import GGuf, GGufError, GGufFileWriter, plan_shards
from . import rearrange
from .quants import GGmlType, Indivisible, Unsupported, block_bytes, block_elems, dequantize_slice, quantize_slice

_FLOATS = (GGmlType.F32, GGmlType.F16, GGmlType.BF16)
_QUANTIZABLE = (GGmlType.F16, GGmlType.BF16, GGmlType.Q4_0, GGmlType.Q4_1, GGmlType.Q5_0, GGmlType.Q5_1,
                GGmlType.Q8_0, GGmlType.Q8_1, GGmlType.Q8K)

# cast.rs:180-216: names are GGmlType variant names, case-insensitive
_NAMES = {t.name.upper(): t for t in GGmlType}


def parse_cast(types):
    """'embd:f16 mat:q8_0, norm:f32' -> {'embd': F16, ...} (cast.rs:13-25)."""
    out = {}
    for key, val in re.findall(r"(\w+):(\w+)", types):
        if val.upper() not in _NAMES:
            raise Unsupported(f"unknown type name {val}")
        out[key] = _NAMES[val.upper()]
    return out


def target_type(arch, name, shape, types):
    """The type a tensor is cast to, or None to leave it (cast.rs:28-71)."""
    linear, embd, norm, else_ = (types.get(k) for k in ("linear", "embd", "norm", "else"))
    if arch in ("llama", "gpt2", "qwen2"):
        if name in ("token_embd.weight", "output.weight"):
            return embd
        if name.endswith("_norm.weight") or name.endswith("_norm.bias"):
            return norm
        if len(shape) > 1 or name.endswith(".bias"):
            return linear
        return else_
    if arch == "clip":
        if name.startswith("v."):
            rest = name[2:]
            return embd if "embd" in rest else norm if "ln" in rest else linear
        if name.startswith("resampler."):
            return norm if name[len("resampler."):].startswith("ln_") else linear
        return else_
    raise Unsupported(f"Unsupported architecture: {arch}")


def _torch_dtype(ty):
    import torch
    return {GGmlType.F32: torch.float32, GGmlType.F16: torch.float16, GGmlType.BF16: torch.bfloat16}[ty]


def cast_device(row, src, from_ty, to_ty):
    """cast(row, data, from, to) of cast.rs:94-138 on a uint8 device tensor of packed data; returns a
    uint8 device tensor.  Every pair decodable-source x encodable-target is served (the reference
    implements the float sources, Q8_0 -> F32, and routes the rest through F32 in principle)."""
    import torch
    from_ty, to_ty = GGmlType(int(from_ty)), GGmlType(int(to_ty))
    if from_ty in _FLOATS:
        elems = src.view(_torch_dtype(from_ty))
        n = elems.numel()
    else:
        bb = block_bytes(from_ty)
        if bb == 0:
            raise Unsupported()
        if src.numel() % bb:
            raise Indivisible()
        elems, n = None, src.numel() // bb * block_elems(from_ty)
    if to_ty in _FLOATS and (from_ty not in _FLOATS or to_ty == GGmlType.F32):
        # dequantize::<Blk, T, N> (cast.rs:150-156); 16-bit T = f32 result narrowed, as lib.rs:62-90
        out = torch.empty(n, dtype=_torch_dtype(to_ty), device=src.device)
        dequantize_slice(from_ty, out, src)
        return out.view(torch.uint8)
    if to_ty not in _QUANTIZABLE:
        raise Unsupported()
    # quantize::<Blk, T, N> (cast.rs:140-148)
    be = block_elems(to_ty)
    if n % row or row % be:
        raise Indivisible()
    if elems is None:
        elems = torch.empty(n, dtype=torch.float32, device=src.device)
        dequantize_slice(from_ty, elems, src)
    out = torch.empty(n // be * block_bytes(to_ty), dtype=torch.uint8, device=src.device)
    quantize_slice(to_ty, out, elems)
    return out


# ---- where a tensor's bytes come from: the reference's DataPromise (xtask/src/utils/mod.rs:98-118) as
# a small expression tree that is evaluated on the device when the writer reaches the tensor ----------

class Borrowed:
    """Bytes of an input file (a view of the mapping); None when the file holds none."""
    lazy = False

    def __init__(self, arr, nbytes):
        self.arr, self.nbytes = arr, nbytes

    def device(self, dev, up=None):
        """`up`: host array -> uint8 device tensor (the producer's pinned, streamed upload); without one
        the bytes go up with a plain blocking copy."""
        import torch
        if self.arr is None:
            raise ValueError("the input file holds no data for this tensor")
        if up is not None:
            return up(self.arr)
        return torch.from_numpy(self.arr).to(dev)


class Cast:
    lazy = True

    def __init__(self, src, row, from_ty, to_ty, nbytes):
        self.src, self.row, self.from_ty, self.to_ty, self.nbytes = src, row, from_ty, to_ty, nbytes

    def device(self, dev, up=None):
        return cast_device(self.row, self.src.device(dev, up), self.from_ty, self.to_ty)

    def is_decode(self):
        """A decode (or widening) of bytes that come straight from an input file: the casts that can share
        one launch plan with others of their kind."""
        return (isinstance(self.src, Borrowed) and self.to_ty in _FLOATS and self.from_ty not in (GGmlType.F32,)
                and (self.from_ty not in _FLOATS or self.to_ty == GGmlType.F32) and block_bytes(self.from_ty) != 0)

    def is_encode(self):
        """A quantization (or narrowing) of float elements that come straight from an input file: the casts that
        can share one quantize plan with others of their kind (quantize::<Blk, T, N>, cast.rs:140-148)."""
        if not isinstance(self.src, Borrowed) or self.from_ty not in _FLOATS or self.to_ty not in _QUANTIZABLE:
            return False
        if self.to_ty == GGmlType.F32:
            return False
        if self.from_ty == self.to_ty or self.src.arr is None:
            return False
        n = self.src.arr.size // (4 if self.from_ty == GGmlType.F32 else 2)
        be = block_elems(self.to_ty)
        return n % self.row == 0 and self.row % be == 0        # the others raise Indivisible when evaluated alone


class Concat:
    """concat(1, parts) (merge.rs:279-324): rows of the parts one after another."""
    lazy = True

    def __init__(self, ty, row_elems, parts):          # parts: [(node, rows)]
        self.ty, self.row_elems, self.parts = ty, row_elems, parts
        self.nbytes = sum(n.nbytes for n, _ in parts)

    def device(self, dev, up=None):
        return rearrange.concat_rows(self.ty, self.row_elems, [(n.device(dev, up), r) for n, r in self.parts])


class Slice:
    """One output of split(1, tensor, sizes) (merge.rs:326-358): a run of whole rows."""
    lazy = True

    def __init__(self, src, start, nbytes):
        self.src, self.start, self.nbytes = src, start, nbytes

    def device(self, dev, up=None):
        return self.src.device(dev, up)[self.start:self.start + self.nbytes].clone()


class PermuteQK:
    lazy = True

    def __init__(self, src, ty, row_elems, rows, nh):
        self.src, self.ty, self.row_elems, self.rows, self.nh, self.nbytes = src, ty, row_elems, rows, nh, src.nbytes

    def device(self, dev, up=None):
        return rearrange.permute_qk(self.ty, self.row_elems, self.rows, self.nh, self.src.device(dev, up))


class Scale:
    """scale_tensor (to_llama.rs:99-125): every element times a factor, in the tensor's type."""
    lazy = True

    def __init__(self, src, ty, factor):
        self.src, self.ty, self.factor, self.nbytes = src, ty, factor, src.nbytes

    def device(self, dev, up=None):
        t = self.src.device(dev, up)
        # a Borrowed upload may be a view of the producer's arena: never scale that in place
        return rearrange.scale(self.ty, t, self.factor)


def _slice(src, start, nbytes):
    if isinstance(src, Borrowed):                        # stays a view of the mapping
        return Borrowed(None if src.arr is None else src.arr[start:start + nbytes], nbytes)
    return Slice(src, start, nbytes)


def _type_bytes(ty, shape):
    n = 1
    for d in shape:
        n *= d
    return n // lib.ggq_gguf_type_elems(int(ty)) * lib.ggq_gguf_type_bytes(int(ty))


class Tensor:
    """Tensor of Content (xtask/src/utils/mod.rs:92-96): type, shape, and where the bytes come from."""
    __slots__ = ("ty", "shape", "data")

    def __init__(self, ty, shape, data):
        self.ty, self.shape, self.data = GGmlType(ty), list(shape), data

    @property
    def nbytes(self):
        return _type_bytes(self.ty, self.shape)


def _distruct(t):
    """[c, r] with a leading 1 for vectors (merge.rs:271-277)."""
    if len(t.shape) == 1:
        return 1, t.shape[0]
    if len(t.shape) == 2:
        return t.shape[0], t.shape[1]
    raise ValueError(f"invalid tensor shape: {t.shape}")


def _concat(tensors):
    """concat(1, tensors), axis 0 for vectors (merge.rs:279-297)."""
    ty, shape = tensors[0].ty, list(tensors[0].shape)
    axis = 0 if len(shape) == 1 else 1
    for t in tensors[1:]:
        assert t.ty == ty and len(t.shape) == len(shape)
        for i, d in enumerate(t.shape):
            if i == axis:
                shape[i] += d
            else:
                assert shape[i] == d
    if axis == 0:      # vectors: a "row" is one block
        be = lib.ggq_gguf_type_elems(int(ty))
        parts = [(t.data, t.shape[0] // be) for t in tensors]
        row_elems = be
    else:
        parts = [(t.data, _type_bytes(ty, t.shape) // _type_bytes(ty, t.shape[:1])) for t in tensors]
        row_elems = shape[0]
    return Tensor(ty, shape, Concat(ty, row_elems, parts))


def _split(t, sizes):
    """split(1, tensor, sizes), axis 0 for vectors (merge.rs:326-358)."""
    axis = 0 if len(t.shape) == 1 else 1
    assert t.shape[axis] == sum(sizes)
    out, start = [], 0
    for d in sizes:
        shape = list(t.shape)
        shape[axis] = d
        n = _type_bytes(t.ty, shape)
        out.append(Tensor(t.ty, shape, _slice(t.data, start, n)))
        start += n
    return out


def merge_qkv(q, k, v):
    qr, kr, vr = _distruct(q)[1], _distruct(k)[1], _distruct(v)[1]
    assert qr % kr == 0 and qr >= kr and kr == vr          # merge.rs:231-233
    return _concat([q, k, v])


def split_qkv(t, nh, nkvh):
    r = _distruct(t)[1]
    dh = r // (nh + 2 * nkvh)                              # merge.rs:255-257
    return _split(t, [nh * dh, nkvh * dh, nkvh * dh])


def permute_qk_tensor(t, nh):
    """permute_qk (permute_qk.rs:46-69)."""
    c, r = _distruct(t)
    return Tensor(t.ty, t.shape, PermuteQK(t.data, t.ty, c, r, nh))


_MERGE = re.compile(r"(attn_q|attn_k|attn_v|ffn_gate|ffn_up|ffn_gate_exps|ffn_up_exps)\.(weight|bias)$")
# as in the reference, ffn_gate_up_exps is merged but never split: its pattern has no such alternative
_SPLIT = re.compile(r"(attn_qkv|ffn_gate_up)\.(weight|bias)$")
_QK = re.compile(r"(attn_q|attn_k|attn_qkv)\.(weight|bias)$")
# group, slot in the group, pieces a complete group has, merged name (merge.rs:171-211, 224-250)
_GROUPS = {"attn_q": ("attn", 0), "attn_k": ("attn", 1), "attn_v": ("attn", 2), "ffn_gate": ("ffn", 0),
           "ffn_up": ("ffn", 1), "ffn_gate_exps": ("moe", 0), "ffn_up_exps": ("moe", 1)}
_GROUP_SIZE = {"attn": 3, "ffn": 2, "moe": 2}
_GROUP_NAME = {"attn": "attn_qkv", "ffn": "ffn_gate_up", "moe": "ffn_gate_up_exps"}


# ---- host-only operators: name patterns, tensor order, metadata text ---------------------------------

def compile_patterns(patterns):
    """compile_patterns (xtask/src/utils/name_pattern.rs:8-58): every run of word characters, dots and
    stars is one pattern; between dots, an empty piece is one name segment, a piece of stars any number
    of segments, and a star inside a piece any run of word characters.  A name matches if it matches
    one pattern as a whole."""
    def one(pattern):
        pieces = []
        for piece in pattern.split("."):
            if piece == "":
                pieces.append(r"\w+")
            elif set(piece) == {"*"}:
                pieces.append(r"(\w+\.)*\w+")
            else:
                pieces.append(re.sub(r"\*+", r"\\w*", piece))
        return "^" + r"\.".join(pieces) + "$"
    return re.compile("|".join(one(p) for p in re.findall(r"[\w*.]+", patterns)))


# sort.rs:18-49: the order of the middle and last parts of a tensor name
_MID = ("attn_norm", "attn_norm_2", "ln1", "attn_qkv", "attn_q", "attn_k", "attn_v", "attn.q", "attn.k", "attn.v",
        "attn_output", "attn_out", "attn.out", "ffn_norm", "ln2", "ffn_gate_inp", "ffn_gate_up", "ffn_gate_up_exps",
        "ffn_gate_up_exp", "ffn_gate", "ffn_gate_exps", "ffn_gate_exp", "ffn_up", "ffn_up_exps", "ffn_up_exp",
        "ffn_down", "ffn_down_exps", "ffn_down_exp")
_POST = ("weight", "bias")
# alternatives are tried in list order at the leftmost position, as the reference's regex does (so
# "attn_norm_2" is found as "attn_norm" followed by "_2")
_MID_RE = re.compile("|".join(m.replace(".", r"\.") for m in _MID))
_MID_RANK = {m: i for i, m in enumerate(_MID)}
_POST_RANK = {m: i for i, m in enumerate(_POST)}


def _cmp(a, b):
    return (a > b) - (a < b)


def _cmp_by_rank(a, b, rank):
    """cmp_by_map (sort.rs:170-177): listed names first, in list order; the rest as strings."""
    ra, rb = rank.get(a), rank.get(b)
    if ra is not None and rb is None:
        return -1
    if ra is None and rb is not None:
        return 1
    if ra is not None:
        return _cmp(ra, rb)
    return _cmp(a, b)


def _name_key(name):
    """Name::new_key (sort.rs:70-104): (segments before the middle part, middle part, rest)."""
    m = _MID_RE.search(name)
    start, end = (m.start(), m.end()) if m else (len(name), len(name))
    pre = [int(s) if s.isascii() and s.isdigit() else s for s in name[:start].split(".")]
    return pre, name[start:end], name[end:].split(".")


def _cmp_names(x, y):
    """Ord for Name / Pre / Mid / Post (sort.rs:106-168)."""
    (pa, ma, sa), (pb, mb, sb) = x, y
    for a, b in zip(pa, pb):
        if isinstance(a, str) != isinstance(b, str):
            return -1 if isinstance(a, str) else 1           # a string segment sorts before a number
        if a != b:
            return _cmp(a, b)
    if len(pa) != len(pb):
        return _cmp(len(pa), len(pb))
    c = _cmp_by_rank(ma, mb, _MID_RANK)
    if c:
        return c
    for a, b in zip(sa, sb):
        c = _cmp_by_rank(a, b, _POST_RANK)
        if c:
            return c
    return _cmp(len(sa), len(sb))


_META_TYPES = {"u8": 0, "i8": 1, "u16": 2, "i16": 3, "u32": 4, "i32": 5, "f32": 6, "bool": 7, "str": 8, "arr": 9,
               "u64": 10, "i64": 11, "f64": 12}
_INT_RANGE = {0: (0, 2 ** 8), 1: (-2 ** 7, 2 ** 7), 2: (0, 2 ** 16), 3: (-2 ** 15, 2 ** 15), 4: (0, 2 ** 32),
              5: (-2 ** 31, 2 ** 31), 10: (0, 2 ** 64), 11: (-2 ** 63, 2 ** 63)}
_INT_FMT = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 10: "<Q", 11: "<q"}


def _meta_string(text):
    raw = text.encode()
    return 8, struct.pack("<Q", len(raw)) + raw


def _parse_f64(text):
    """str::parse::<f64> of the reference: Python's float() minus what Rust refuses (digit separators, padding)."""
    if re.search(r"[_\s]", text) or text == "":
        raise ValueError(f"invalid float: {text!r}")
    return float(text)


def _parse_f32(text):
    """str::parse::<f32>: ONE correctly rounded step from the decimal text (float() then a cast would round twice:
    to f64, then to f32, which differs for texts within half an f64 step of an f32 tie)."""
    from fractions import Fraction
    d = _parse_f64(text)
    if d != d or d in (float("inf"), float("-inf")) and not re.search(r"\d", text):
        return np.float32(d)                     # "nan", "inf", "infinity" in any case
    try:
        exact = Fraction(text)
    except (ValueError, ZeroDivisionError):
        with np.errstate(over="ignore"):
            return np.float32(d)
    neg = text.lstrip().startswith("-")
    mag = abs(exact)
    if mag == 0:
        return np.float32(-0.0 if neg else 0.0)
    e = mag.numerator.bit_length() - mag.denominator.bit_length()       # 2^(e-1) < mag < 2^(e+1)
    if Fraction(2) ** e > mag:
        e -= 1
    quantum = Fraction(2) ** (max(e, -126) - 23)
    n, rem = divmod(mag, quantum)
    n = int(n)
    if rem * 2 > quantum or (rem * 2 == quantum and n & 1):
        n += 1
    value = n * quantum
    out = float("inf") if value >= Fraction(2) ** 128 else float(value)     # exact: value is an f32
    return np.float32(-out if neg else out)


def _meta_value(ty, text):
    """write_val (set_meta.rs:212-249): the value's bytes from its text."""
    if ty == 8:
        return _meta_string(text)
    if ty in _INT_FMT:
        if not re.fullmatch(r"[+-]?\d+" if _INT_RANGE[ty][0] < 0 else r"\+?\d+", text):      # no sign for unsigned types
            raise ValueError(f"invalid integer: {text!r}")
        v = int(text)
        lo, hi = _INT_RANGE[ty]
        if not lo <= v < hi:
            raise ValueError(f"integer out of range: {text}")
        return ty, struct.pack(_INT_FMT[ty], v)
    if ty in (6, 12):
        return ty, struct.pack("<f", _parse_f32(text)) if ty == 6 else struct.pack("<d", _parse_f64(text))
    if ty == 7:
        if text not in ("true", "false"):
            raise ValueError(f"Invalid bool value: {text}")
        return ty, bytes([text == "true"])
    raise Unsupported("array values are not supported")          # todo!() in the reference (set_meta.rs:106-108)


def parse_meta_cfg(cfg):
    """Operator::set_meta_by_cfg (set_meta.rs:10-145): lines `'key' type value`; `'key' str "text"` for
    a one-line string; `'key' str<sep>` followed by lines that start with `<sep> ` for a string of
    several lines.  Returns {key: (type id, value bytes)} in the order of the text."""
    out = {}
    pending = None                       # [key, separator + " ", lines or None, started]

    def save():
        out[pending[0]] = _meta_string("\n".join(pending[2] or []))

    for line in cfg.split("\n"):
        line = line[:-1] if line.endswith("\r") else line
        if pending is not None:
            if line == "":
                if pending[3]:           # an empty line ends a string that has begun
                    save()
                    pending = None
                continue
            if line.startswith(pending[1]):
                pending[2] = (pending[2] or []) + [line[len(pending[1]):]]
                pending[3] = True
                continue
            save()                       # the string is over; the line is an item of its own
            pending = None
        if line == "":
            continue
        m = re.match(r"^'((?:\w+\.)*\w+)'\s*(\S+)", line)
        if not m:
            raise ValueError(f"Expect meta kv config item: {line!r}")
        key, ty = m.group(1), m.group(2)
        val = line[m.end(2):].strip()
        if ty.startswith("str"):
            sep = ty[3:]
            if sep == "":
                if len(val) < 2 or val[0] != '"' or val[-1] != '"':
                    raise ValueError(f"{key}: a one-line string stands in double quotes")
                out[key] = _meta_string(val[1:-1])
            else:
                if val:
                    raise ValueError(f"{key}: a string of several lines starts on the next line")
                pending = [key, sep + " ", None, False]
        elif re.match(r"^\[(\w+)\](\S+)?$", ty):
            raise Unsupported("array values are not supported")
        else:
            if ty not in _META_TYPES:
                raise ValueError(f"Unknown type: {ty}")
            out[key] = _meta_value(_META_TYPES[ty], val)
    if pending is not None:
        save()
    return out


def parse_steps(steps):
    """The `-x` argument of `convert` (xtask/src/convert.rs:38-51): steps separated by "->"."""
    out = []
    for op in steps.split("->"):
        op = op.strip()
        if op == "sort":
            out.append(("sort",))
        elif op == "permute-qk":
            out.append(("permute-qk",))
        elif op == "merge-linear":
            out.append(("merge-linear", True))
        elif op in ("split-linear", "!merge-linear"):
            out.append(("merge-linear", False))
        elif op == "to-llama":
            out.append(("to-llama", ""))
        else:
            head, sep, rest = op.partition(":")
            if not sep or head not in ("cast", "to-llama", "filter-meta", "filter-tensor"):
                raise Unsupported(f"Unsupported operation: {op}")
            out.append((head, rest))
    return out


class Content:
    """Content (xtask/src/utils/mod.rs:61-66): the merged view of one or more input shards."""

    def __init__(self, paths):
        self._files = [GGuf(p) for p in paths]
        self.alignment = 0
        self.meta_kvs = {}          # key -> (type id, raw value bytes), insertion-ordered
        self.tensors = {}
        for g in self._files:       # merge_file, read.rs:33-62
            self.alignment = max(self.alignment, g.alignment)
            for k in g.keys():
                if k == "general.alignment" or k.startswith("split."):
                    continue
                if k in self.meta_kvs:
                    raise GGufError(8, k)
                self.meta_kvs[k] = g.meta_raw(k)
            for t in g.tensors:
                if t.name in self.tensors:
                    raise GGufError(9, t.name)
                self.tensors[t.name] = Tensor(t.ty, t.shape, Borrowed(t.bytes() if t.ptr else None, t.nbytes))

    def close(self):
        self.tensors = {}
        for g in self._files:
            g.close()
        self._files = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def general_architecture(self):
        ty, raw = self.meta_kvs["general.architecture"]
        return raw[8:].decode()

    def cast(self, types):
        """Content::cast / cast_ (cast.rs:28-90): only marks the tensors; the work happens when written."""
        if isinstance(types, str):
            types = parse_cast(types)
        arch = self.general_architecture()
        for name, t in self.tensors.items():
            to = target_type(arch, name, t.shape, types)
            if to is not None and to != t.ty:
                if to not in _FLOATS and to not in _QUANTIZABLE:
                    raise Unsupported(f"{name}: no encoder for {to.name}")
                if t.ty not in _FLOATS and block_bytes(t.ty) == 0:
                    raise Unsupported(f"{name}: no decoder for {t.ty.name}")
                be = lib.ggq_gguf_type_elems(int(to))
                if t.shape[0] % be:
                    raise Indivisible(f"{name}: row of {t.shape[0]} is not a multiple of {be}")
                t.data = Cast(t.data, t.shape[0], t.ty, to, _type_bytes(to, t.shape))
                t.ty = to

    def _heads(self):
        def u(key):
            if key not in self.meta_kvs:
                return None
            ty, raw = self.meta_kvs[key]
            return int.from_bytes(raw, "little")
        arch = self.general_architecture()
        nh = u(f"{arch}.attention.head_count")
        if nh is None:
            raise KeyError(f"{arch}.attention.head_count")
        nkvh = u(f"{arch}.attention.head_count_kv")
        return nh, nh if nkvh is None else nkvh           # merge.rs:42-47

    def merge_linear(self, merge=True):
        """Content::merge_linear (merge.rs:22-80): merge=True joins attn_q/k/v into attn_qkv and
        ffn_gate/up into ffn_gate_up per name prefix (a merged tensor takes the place of the piece
        that completes it; incomplete groups are appended at the end); merge=False splits them."""
        old, self.tensors = self.tensors, {}
        if merge:
            groups = {}
            for name, t in old.items():
                m = _MERGE.search(name)
                if not m:
                    self.tensors[name] = t
                    continue
                pre, (piece, wb) = name[:m.start()], m.groups()
                group, slot = _GROUPS[piece]
                slots = groups.setdefault((pre, group, wb), {})
                slots[slot] = t
                if len(slots) == _GROUP_SIZE[group]:
                    del groups[(pre, group, wb)]
                    parts = [slots[i] for i in range(_GROUP_SIZE[group])]
                    if group == "attn":
                        merged = merge_qkv(*parts)
                    else:
                        if group == "ffn":
                            assert parts[0].shape[1] == parts[1].shape[1]        # merge.rs:241
                        merged = _concat(parts)
                    self.tensors[f"{pre}{_GROUP_NAME[group]}.{wb}"] = merged
            names = {v: k for k, v in _GROUPS.items()}
            for (pre, group, wb), slots in groups.items():
                for slot in sorted(slots):
                    self.tensors[f"{pre}{names[(group, slot)]}.{wb}"] = slots[slot]
        else:
            nh, nkvh = self._heads()
            for name, t in old.items():
                m = _SPLIT.search(name)
                if not m:
                    self.tensors[name] = t
                    continue
                pre, (piece, wb) = name[:m.start()], m.groups()
                if piece == "attn_qkv":
                    outs = zip(("attn_q", "attn_k", "attn_v"), split_qkv(t, nh, nkvh))
                else:
                    r = t.shape[1] // 2                                            # merge.rs:260-263
                    outs = zip(("ffn_gate", "ffn_up"), _split(t, [r, r]))
                for piece_name, part in outs:
                    self.tensors[f"{pre}{piece_name}.{wb}"] = part

    def permute_qk(self):
        """Content::permute_qk (permute_qk.rs:12-44)."""
        nh, nkvh = self._heads()
        for name, t in list(self.tensors.items()):
            m = _QK.search(name)
            if not m:
                continue
            piece = m.group(1)
            if piece == "attn_q":
                self.tensors[name] = permute_qk_tensor(t, nh)
            elif piece == "attn_k":
                self.tensors[name] = permute_qk_tensor(t, nkvh)
            else:
                q, k, v = split_qkv(t, nh, nkvh)
                self.tensors[name] = merge_qkv(permute_qk_tensor(q, nh), permute_qk_tensor(k, nkvh), v)

    # ---- host-only operators (operator/mod.rs:58-71) --------------------------------------------

    def filter_meta_key(self, patterns):
        r = compile_patterns(patterns)
        self.meta_kvs = {k: v for k, v in self.meta_kvs.items() if r.search(k)}

    def filter_tensor_name(self, patterns):
        r = compile_patterns(patterns)
        self.tensors = {k: v for k, v in self.tensors.items() if r.search(k)}

    def sort_tensors(self):
        """Content::sort_tensors (sort.rs:7-15)."""
        order = sorted(self.tensors, key=functools.cmp_to_key(lambda a, b: _cmp_names(_name_key(a), _name_key(b))))
        self.tensors = {k: self.tensors[k] for k in order}

    def set_meta(self, items):
        """Content::set_meta (set_meta.rs:26-55).  `items`: the configuration text, or {key: (type id,
        value bytes)}.  Keys that exist keep their place; new ones are appended; `general.alignment`
        sets the alignment of the output; `split.*` keys are refused."""
        if isinstance(items, str):
            items = parse_meta_cfg(items)
        items = dict(items)
        for k in self.meta_kvs:
            if k in items:
                self.meta_kvs[k] = items.pop(k)
        for k, (ty, raw) in items.items():
            if k == "general.alignment":
                if ty != 4 or len(raw) != 4:
                    raise GGufError(7, k)
                self.alignment = int.from_bytes(raw, "little")
            elif k.startswith("split."):
                raise ValueError(f"Split is not allowed: {k}")
            else:
                self.meta_kvs[k] = (ty, raw)

    def to_llama(self, extra=""):
        """Content::convert_to_llama (to_llama.rs:26-97).  A llama model stays as it is.  MiniCPM:
        token_embd times the embedding scale, every attn_output and ffn_down times residual scale /
        sqrt(block count), the three scale keys dropped, `minicpm.*` keys renamed to `llama.*`.
        `extra`: "embd_scale=<num> res_scale=<num>" overrides the file's values."""
        arch = self.general_architecture()
        if arch == "llama":
            return
        if arch != "minicpm":
            raise Unsupported(f"Unsupported architecture: {arch}")
        if isinstance(extra, str):
            extra = dict(re.findall(r"(\w+)=(\w+)", extra))
        keys = {n: f"minicpm.{n}" for n in ("embedding_scale", "residual_scale", "logit_scale")}

        def f32_key(key):
            if key not in self.meta_kvs:
                return None
            ty, raw = self.meta_kvs[key]
            if ty != 6:
                raise GGufError(11, key)
            return float(struct.unpack("<f", raw)[0])

        ty, raw = self.meta_kvs.get("minicpm.block_count", (None, None))
        if ty not in _INT_FMT:
            raise KeyError("Missing llm_block_count")
        nblk = struct.unpack(_INT_FMT[ty], raw)[0]
        embd, res = f32_key(keys["embedding_scale"]), f32_key(keys["residual_scale"])
        if "embd_scale" in extra:
            embd = _parse_f64(extra["embd_scale"])
        if "res_scale" in extra:
            res = _parse_f64(extra["res_scale"])
        if embd is None or res is None:
            raise ValueError('convert from minicpm requires extra args "embd_scale=<num> res_scale=<num>"')
        res = res / math.sqrt(float(nblk))
        blk = re.compile(r"^blk\.(\d+)\.(\w+)\.weight$")
        for name, t in self.tensors.items():
            m = blk.match(name)
            factor = embd if name == "token_embd.weight" else \
                res if m and m.group(2) in ("attn_output", "ffn_down") else None
            if factor is not None:
                if t.ty not in (GGmlType.F32, GGmlType.F16, GGmlType.BF16, GGmlType.F64):
                    raise Unsupported(f"unsupported tensor type: {t.ty!r}")
                t.data = Scale(t.data, t.ty, factor)
        # set_arch (to_llama.rs:84-97)
        old, self.meta_kvs = self.meta_kvs, {}
        for k, v in old.items():
            if k == "general.architecture":
                self.meta_kvs[k] = _meta_string("llama")
            elif k not in keys.values():
                self.meta_kvs["llama." + k[len("minicpm."):] if k.startswith("minicpm.") else k] = v

    def apply(self, step):
        """Content::apply (operator/mod.rs:59-71) for one parsed step of `parse_steps`."""
        op = step[0]
        if op == "cast":
            self.cast(step[1])
        elif op == "merge-linear":
            self.merge_linear(step[1])
        elif op == "permute-qk":
            self.permute_qk()
        elif op == "sort":
            self.sort_tensors()
        elif op == "to-llama":
            self.to_llama(step[1] if len(step) > 1 else "")
        elif op == "filter-meta":
            self.filter_meta_key(step[1])
        elif op == "filter-tensor":
            self.filter_tensor_name(step[1])
        elif op == "set-meta":
            self.set_meta(step[1])
        else:
            raise Unsupported(f"unknown step {op}")

    # ---- writing ------------------------------------------------------------------------------

    def write_files(self, out_dir, stem, max_tensor_count=2 ** 63, max_file_size=2 ** 63, no_tensor_first=False,
                    write_data=True, device=0, rank=0, world=1, max_parallel=4):
        """Content::write_files (write.rs:5-100): plan shards, then one writer thread per shard.
        Returns [(path, n_tensors, n_meta_kvs, n_bytes)] (FileInfo).
        With world > 1 (one process per GPU) every rank plans the same shards and writes those with
        index % world == rank on its own device; no data moves between ranks.
        At most `max_parallel` shards are in flight at once (each holds two pinned buffers of its
        largest lazy tensor); the reference starts all shard threads together."""
        names = list(self.tensors)
        kvs = [(k, ty, raw) for k, (ty, raw) in self.meta_kvs.items()]
        shard_of, n_shards = plan_shards(self.alignment, kvs, [(n, self.tensors[n].ty, self.tensors[n].shape)
                                                               for n in names],
                                         max_tensor_count, max_file_size, no_tensor_first)
        shards = [[] for _ in range(n_shards)]
        for name, s in zip(names, shard_of):
            shards[s].append((name, self.tensors[name]))
        os.makedirs(out_dir, exist_ok=True)
        paths = [os.path.join(out_dir, f"{stem}.gguf" if n_shards == 1 else f"{stem}-{i + 1:05}-of-{n_shards:05}.gguf")
                 for i in range(n_shards)]
        results, errors = [None] * n_shards, []

        gate = threading.Semaphore(max(1, max_parallel))

        def run(i):
            with gate:
                try:
                    results[i] = self._write_shard(i, paths[i], shards[i], kvs, write_data, device)
                except BaseException as e:      # reported by the caller's thread
                    errors.append(e)

        threads = [threading.Thread(target=run, args=(i,)) for i in range(n_shards) if i % world == rank]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
        return [r for r in results if r is not None]

    def _write_shard(self, i, path, tensors, kvs, write_data, device):
        n_kv = (len(kvs) if i == 0 else 0) + 1
        w = GGufFileWriter(path, len(tensors), n_kv, write_data=write_data)
        w.write_alignment(self.alignment)
        if i == 0:
            for k, ty, raw in kvs:
                w.write_meta_kv(k, ty, raw)
        lazy = [(name, t) for name, t in tensors if t.data.lazy]
        producer = _Producer(lazy, device) if (write_data and lazy) else None
        try:
            for name, t in tensors:
                if not write_data:
                    w.write_tensor(name, t.ty, t.shape, None)
                elif not t.data.lazy:
                    if t.data.arr is None:
                        raise ValueError(f"{name}: the input file holds no data for this tensor")
                    w.write_tensor(name, t.ty, t.shape, t.data.arr)
                else:
                    w.write_tensor(name, t.ty, t.shape, producer.next)
            n_bytes = w.finish()
        finally:
            w.close()
            if producer:
                producer.stop()
        return path, len(tensors), n_kv, n_bytes


class _Producer:
    """Evaluates a shard's lazy tensors on the GPU in file order, ahead of the writer, into two pinned host
    buffers that alternate.  Input bytes go up through the producer's own Context (pinned staging, read
    from the file mapping by several threads).  Runs of plain decode casts are evaluated as a group: all
    their packed bytes are uploaded, then ONE launch plan decodes them (one kernel launch per block type
    instead of one per tensor), up to `group_bytes` of decoded output at a time."""

    def __init__(self, tensors, device, group_bytes=4 << 30):
        import torch
        self._torch = torch
        self._tensors = tensors
        self._device = device
        self._group_bytes = group_bytes
        size = max(t.nbytes for _, t in tensors)
        self._bufs = [torch.empty(size, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._free = queue.Queue()
        for b in range(2):
            self._free.put(b)
        self._ready = queue.Queue()
        self._held = None
        self._stop = False
        self.launches = 0             # decode kernel launches issued through launch plans
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _units(self):
        """File order, cut into units: ("decode", [casts that share a decode plan]), ("encode", [casts that share
        a quantize plan]) or ("one", (name, tensor))."""
        units, run, kind, held = [], [], None, 0

        def close():
            nonlocal run, kind, held
            if run:
                units.append((kind, run))
            run, kind, held = [], None, 0

        for name, t in self._tensors:
            k = None
            if isinstance(t.data, Cast) and isinstance(t.data.src, Borrowed) and t.data.src.arr is not None:
                k = "decode" if t.data.is_decode() else "encode" if t.data.is_encode() else None
            if k is None:
                close()
                units.append(("one", (name, t)))
                continue
            size = max(t.nbytes, t.data.src.arr.size)
            if run and (k != kind or held + size > self._group_bytes):
                close()
            kind = k
            run.append((name, t))
            held += size
        close()
        return units

    def _emit(self, name, t, out, stream):
        """Device bytes -> the next free pinned buffer -> the writer.  False when asked to stop."""
        b = self._free.get()
        if b is None or self._stop:
            return False
        assert out.numel() == t.nbytes, (name, out.numel(), t.nbytes)
        host = self._bufs[b][:out.numel()]
        host.copy_(out, non_blocking=True)
        stream.synchronize()
        self._ready.put((b, host.numpy()))
        return True

    def _run(self):
        torch = self._torch
        from .quants import Batch, Context, QuantBatch
        dev = f"cuda:{self._device}"
        ctx = None
        try:
            with torch.cuda.device(self._device):
                ctx = Context(self._device, 64 << 20)

                stream = torch.cuda.Stream()

                def up(arr):
                    # The allocator hands out blocks freed on `stream` while kernels queued there may still read
                    # them (safe for later work on `stream` only); the copy runs on the context's streams, so it
                    # is ordered behind `stream` first.
                    d = torch.empty(arr.size, dtype=torch.uint8, device=dev)
                    ctx.wait_stream(stream.cuda_stream)
                    ctx.upload_async(d, arr)
                    ctx.sync()
                    return d

                with torch.cuda.stream(stream):
                    for kind, unit in self._units():
                        if kind == "encode":
                            # the same for quantizing casts: per (element type, target type) pair one element arena
                            # and one packed arena, one quantize plan per element type
                            pairs = {}
                            for k, (name, t) in enumerate(unit):
                                pairs.setdefault((int(t.data.from_ty), int(t.data.to_ty)), []).append(k)
                            ctx.wait_stream(stream.cuda_stream)
                            outs, by_in, keep = [None] * len(unit), {}, []
                            for (from_ty, to_ty), members in pairs.items():
                                eb = 4 if from_ty == GGmlType.F32 else 2
                                nbytes = sum(unit[k][1].data.src.arr.size for k in members)
                                nblocks = nbytes // eb // block_elems(to_ty)
                                elems = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                                packed = torch.empty(nblocks * block_bytes(to_ty), dtype=torch.uint8, device=dev)
                                keep += [elems, packed]
                                at_e = at_b = 0
                                for k in members:
                                    arr = unit[k][1].data.src.arr
                                    nb = arr.size // eb // block_elems(to_ty) * block_bytes(to_ty)
                                    ctx.upload_async(elems[at_e:at_e + arr.size], arr)
                                    outs[k] = packed[at_b:at_b + nb]
                                    by_in.setdefault(from_ty, []).append(
                                        (to_ty, elems[at_e:at_e + arr.size].view(_torch_dtype(GGmlType(from_ty))), outs[k]))
                                    at_e += arr.size
                                    at_b += nb
                            plans = [QuantBatch(v, k) for k, v in by_in.items()]
                            ctx.sync()
                            for p in plans:
                                p.run(stream.cuda_stream)
                                self.launches += p.launches
                            for (name, t), od in zip(unit, outs):
                                if not self._emit(name, t, od, stream):
                                    return
                            for p in plans:
                                p.close()
                            del outs, by_in, keep
                        elif kind == "decode":
                            # Typed layout: the unit's tensors of one (source type, output type) pair sit back to
                            # back in one packed arena and in one output arena, in file order, so that the launch
                            # plan sees one long run per pair (ggq_batch_create merges adjacent tensors).
                            pairs = {}
                            for k, (name, t) in enumerate(unit):
                                pairs.setdefault((int(t.data.from_ty), int(t.data.to_ty)), []).append(k)
                            ctx.wait_stream(stream.cuda_stream)
                            outs, by_out, keep = [None] * len(unit), {}, []
                            for (from_ty, to_ty), members in pairs.items():
                                nbytes = sum(unit[k][1].data.src.arr.size for k in members)
                                nelems = nbytes // block_bytes(from_ty) * block_elems(from_ty)
                                packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                                decoded = torch.empty(nelems, dtype=_torch_dtype(to_ty), device=dev)
                                keep += [packed, decoded]
                                at_b = at_e = 0
                                for k in members:
                                    arr = unit[k][1].data.src.arr
                                    n = arr.size // block_bytes(from_ty) * block_elems(from_ty)
                                    ctx.upload_async(packed[at_b:at_b + arr.size], arr)
                                    outs[k] = decoded[at_e:at_e + n]
                                    by_out.setdefault(to_ty, []).append((from_ty, packed[at_b:at_b + arr.size], outs[k]))
                                    at_b += arr.size
                                    at_e += n
                            plans = [Batch(v, k) for k, v in by_out.items()]     # built while the last copies fly
                            ctx.sync()
                            for p in plans:
                                p.run(stream.cuda_stream)
                                self.launches += p.launches
                            for (name, t), od in zip(unit, outs):
                                if not self._emit(name, t, od.view(torch.uint8), stream):
                                    return
                            for p in plans:
                                p.close()
                            del outs, by_out, keep
                        else:
                            name, t = unit
                            try:
                                out = t.data.device(dev, up)
                            except ValueError as e:
                                raise ValueError(f"{name}: {e}") from None
                            if not self._emit(name, t, out, stream):
                                return
        except BaseException as e:
            self._ready.put(e)
        finally:
            if ctx is not None:
                ctx.close()

    def next(self):
        """The next tensor's bytes; the buffer handed out before it goes back to the producer."""
        if self._held is not None:
            self._free.put(self._held)
            self._held = None
        item = self._ready.get()
        if isinstance(item, BaseException):
            raise item
        self._held, arr = item
        return arr

    def stop(self):
        self._stop = True
        self._free.put(None)
        self._thread.join(timeout=60)


def convert(inputs, out_dir, stem, steps=(), cast=None, **write_args):
    """`convert <inputs> -x <steps>`: read, apply the steps in order, write.  `steps` is the reference's
    step string ("sort->cast:linear:f16->permute-qk", convert.rs:13) or a list of parsed steps:
    ("cast", "linear:f32 ..."), ("merge-linear", True|False), ("permute-qk",), ("sort",), ("to-llama", extra),
    ("filter-meta", patterns), ("filter-tensor", patterns), ("set-meta", text or dict).
    Returns the FileInfo list."""
    steps = parse_steps(steps) if isinstance(steps, str) else list(steps)
    steps += [("cast", cast)] if cast else []
    with Content(inputs) as c:
        for step in steps:
            c.apply(step)
        return c.write_files(out_dir, stem, **write_args)
