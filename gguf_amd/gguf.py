"""GGUF container access: host-side mirror of the reference's `ggus` reader over the C ABI of
include/ggq_gguf.h (`GGuf::new`, ggus/src/file.rs:66-145), a minimal writer for building files
(layout of ggus/src/write/writer.rs:95-128: header, key/values, tensor infos, padding, data), and
the whole-file decode that the reference's `convert ... -x cast:...:f32` performs one tensor at a
time (xtask/src/utils/operator/cast.rs:73-90).
"""
import ctypes
import struct
import threading

import numpy as np

from . import _lib
from ._lib import lib
from .quants import GGmlType, block_bytes, block_elems

OK = 0
ERRORS = {1: "Eos", 2: "Utf8", 3: "Bool", 4: "MagicMismatch", 5: "EndianNotSupport", 6: "VersionNotSupport",
          7: "AlignmentTypeMismatch", 8: "DuplicateMetaKey", 9: "DuplicateTensorName", 10: "Io",
          11: "ValueType", 12: "NotFound", 13: "Invalid", 14: "Indivisible", 15: "TensorRange"}
DEFAULT_ALIGNMENT = 32

_SCALAR = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}


class GGufError(Exception):
    """GGufError / GGufReadError of the reference (file.rs:25-40, read.rs:14-21)."""

    def __init__(self, code, name=""):
        self.code = code
        self.kind = ERRORS.get(code, str(code))
        self.name = name
        super().__init__(lib.ggq_gguf_error_str(code).decode() + (f": {name}" if name else ""))


def _check(code):
    if code != OK:
        name = lib.ggq_gguf_last_name().decode() if code in (7, 8, 9, 14, 15) else ""
        raise GGufError(code, name)


class TensorInfo:
    """GGufTensorInfo (ggus/src/tensor.rs:255-290) plus where its bytes are."""
    __slots__ = ("name", "ty", "shape", "offset", "nbytes", "ptr", "_region")

    def __init__(self, t, region=None):
        self._region = region              # () -> uint8 view of the file's whole data region, or None
        self.name = ctypes.string_at(t.name, t.name_len).decode()
        self.ty = t.ggml_type
        self.shape = tuple(t.shape[i] for i in range(min(t.ndim, 4)))
        self.offset = t.offset
        self.nbytes = t.nbytes
        self.ptr = t.data                  # None when the file does not hold the tensor

    def bytes(self):
        """Zero-copy uint8 view of the packed data (valid while the GGuf is open)."""
        if not self.ptr:
            raise ValueError(f"{self.name}: the file holds no data for this tensor")
        if self._region is not None:       # a slice of one cached view: building a view per tensor costs ~50 us
            return self._region()[self.offset:self.offset + self.nbytes]
        return np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(self.nbytes,))

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


def _decode_value(ty, buf, pos=0):
    if ty in _SCALAR:
        return struct.unpack_from(_SCALAR[ty], buf, pos)[0], pos + struct.calcsize(_SCALAR[ty])
    if ty == 8:
        n = struct.unpack_from("<Q", buf, pos)[0]
        return bytes(buf[pos + 8:pos + 8 + n]).decode("utf-8"), pos + 8 + n
    if ty == 9:
        ety, n = struct.unpack_from("<IQ", buf, pos)
        pos += 12
        out = []
        for _ in range(n):
            v, pos = _decode_value(ety, buf, pos)
            out.append(v)
        return out, pos
    raise ValueError(f"unknown metadata value type {ty}")


class GGuf:
    """A parsed GGUF file.  `GGuf(path)` maps the file; `GGuf.from_bytes(buf)` parses memory."""

    def __init__(self, path=None, _buf=None):
        h = ctypes.c_void_p()
        if _buf is not None:
            self._keep = np.frombuffer(_buf, dtype=np.uint8) if not isinstance(_buf, np.ndarray) else _buf
            _check(lib.ggq_gguf_parse(self._keep.ctypes.data, self._keep.size, ctypes.byref(h)))
        else:
            self._keep = None
            _check(lib.ggq_gguf_open(str(path).encode(), ctypes.byref(h)))
        self._h = h
        info = _lib.ggq_gguf_info()
        _check(lib.ggq_gguf_get_info(h, ctypes.byref(info)))
        self.version = info.version
        self.alignment = info.alignment
        self.data_offset = info.data_offset
        self.data_len = info.data_len
        self.data_avail = info.data_avail
        self.tensors = []
        self._data_view = None
        self._data_base = None
        t = _lib.ggq_gguf_tensor()
        for i in range(info.tensor_count):
            _check(lib.ggq_gguf_tensor_at(h, i, ctypes.byref(t)))
            ti = TensorInfo(t, self._region)
            if ti.ptr and self._data_base is None:
                self._data_base = ti.ptr - ti.offset
            self.tensors.append(ti)
        self._kv_count = info.kv_count

    def _region(self):
        """uint8 view of the whole data region (built once; tensors slice it)."""
        if self._data_view is None:
            self._data_view = np.ctypeslib.as_array(ctypes.cast(self._data_base, ctypes.POINTER(ctypes.c_uint8)),
                                                    shape=(self.data_avail,))
        return self._data_view

    @classmethod
    def from_bytes(cls, buf):
        return cls(_buf=buf)

    def meta_raw(self, key):
        """(value type, raw value bytes) like GGufMetaMap::get (file.rs:58-62); None if absent."""
        kv = _lib.ggq_gguf_kv()
        code = lib.ggq_gguf_find_kv(self._h, key.encode(), ctypes.byref(kv))
        if code == 12:
            return None
        _check(code)
        return kv.type, ctypes.string_at(kv.value, kv.value_len)

    def meta(self, key, default=None):
        raw = self.meta_raw(key)
        if raw is None:
            return default
        return _decode_value(raw[0], raw[1])[0]

    def keys(self):
        kv = _lib.ggq_gguf_kv()
        out = []
        for i in range(self._kv_count):
            _check(lib.ggq_gguf_kv_at(self._h, i, ctypes.byref(kv)))
            out.append(ctypes.string_at(kv.key, kv.key_len).decode())
        return out

    def tensor(self, name):
        t = _lib.ggq_gguf_tensor()
        _check(lib.ggq_gguf_find_tensor(self._h, name.encode(), ctypes.byref(t)))
        return TensorInfo(t, self._region if self._data_base is not None else None)

    def close(self):
        if self._h:
            self._data_view = None
            lib.ggq_gguf_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- writer --------------------------------------------------------------------------------

def _enc_str(s):
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def _enc_value(v):
    """(type id, bytes) for a Python value: bool, int (-> u32 / i32 / u64 / i64), float (f32), str,
    or a (type id, value) pair to force a type; lists become arrays of their first item's type."""
    if isinstance(v, tuple):
        ty, val = v
        if ty in _SCALAR:
            return ty, struct.pack(_SCALAR[ty], val)
        if ty == 8:
            return 8, _enc_str(val)
        raise ValueError(ty)
    if isinstance(v, bool):
        return 7, struct.pack("<?", v)
    if isinstance(v, int):
        if 0 <= v < 2 ** 32:
            return 4, struct.pack("<I", v)
        if -2 ** 31 <= v < 0:
            return 5, struct.pack("<i", v)
        return (10, struct.pack("<Q", v)) if v >= 0 else (11, struct.pack("<q", v))
    if isinstance(v, float):
        return 6, struct.pack("<f", v)
    if isinstance(v, str):
        return 8, _enc_str(v)
    if isinstance(v, (list, tuple)):
        items = [_enc_value(x) for x in v]
        ety = items[0][0] if items else 4
        return 9, struct.pack("<IQ", ety, len(items)) + b"".join(b for _, b in items)
    raise TypeError(type(v))


def build_gguf(meta, tensors, alignment=None, with_data=True):
    """Bytes of a GGUF v3 file.  meta: {key: value}; tensors: [(name, ggml_type, shape, data)], data a
    uint8 array of exactly the tensor's packed bytes (or None with with_data=False for a
    metadata-only file like the reference's test asset)."""
    meta = dict(meta)
    if alignment is not None:
        meta["general.alignment"] = alignment
    align = int(meta.get("general.alignment", DEFAULT_ALIGNMENT)
                if not isinstance(meta.get("general.alignment"), tuple) else meta["general.alignment"][1])
    out = bytearray(b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(meta)))
    for k, v in meta.items():
        ty, b = _enc_value(v)
        out += _enc_str(k) + struct.pack("<I", ty) + b
    offset = 0
    offsets = []
    for name, ty, shape, data in tensors:
        eb, ee = lib.ggq_gguf_type_bytes(int(ty)), lib.ggq_gguf_type_elems(int(ty))
        n = 1
        for s in shape:
            n *= s
        nbytes = n // ee * eb
        if data is not None:
            assert np.asarray(data).nbytes == nbytes, (name, np.asarray(data).nbytes, nbytes)
        out += _enc_str(name) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", s) for s in shape)
        out += struct.pack("<IQ", int(ty), offset)
        offsets.append((offset, nbytes))
        offset += nbytes + (-nbytes) % align
    if with_data and tensors:
        out += b"\0" * ((-len(out)) % align)
        base = len(out)
        for (name, ty, shape, data), (off, nbytes) in zip(tensors, offsets):
            out += b"\0" * (base + off - len(out))
            out += np.ascontiguousarray(data).view(np.uint8).tobytes()
    return bytes(out)


class GGufFileWriter:
    """GGufFileWriter + GGufTensorWriter of the reference (ggus/src/write/file_writer.rs:39-137) in one
    object: key/values, then tensors, then finish().  path=None gives the byte-counting simulator
    (GGufFileSimulator / GGufTensorSimulator, simulator.rs:26-95)."""

    def __init__(self, path, n_tensors=0, n_kv=0, write_data=True, alignment=None):
        h = ctypes.c_void_p()
        _check(lib.ggq_gguf_writer_create(None if path is None else str(path).encode(), n_tensors, n_kv,
                                          int(write_data), ctypes.byref(h)))
        self._h = h
        self._keep = []          # data buffers and callbacks must outlive finish()
        self._error = None
        if alignment is not None:
            self.write_alignment(alignment)

    @property
    def alignment(self):
        return lib.ggq_gguf_writer_alignment_value(self._h)

    def write_alignment(self, alignment):
        _check(lib.ggq_gguf_writer_alignment(self._h, alignment))

    def write_meta_kv(self, key, ty, value):
        """value: the raw bytes of the value as stored (what GGufMetaMap::get returns)."""
        k = key.encode()
        _check(lib.ggq_gguf_writer_kv(self._h, k, len(k), ty, bytes(value), len(value)))

    def write_meta(self, key, value):
        ty, b = _enc_value(value)
        self.write_meta_kv(key, ty, b)

    def write_tensor(self, name, ty, shape, data=None):
        """data: None, a uint8 array / bytes, or a callable () -> uint8 array evaluated when finish()
        reaches the tensor (the reference's lazy DataPromise, xtask/src/utils/mod.rs)."""
        n = name.encode()
        shp = (ctypes.c_uint64 * len(shape))(*shape)
        ptr, fn = None, _lib.ggq_gguf_data_fn(0)
        if callable(data):
            produce = data

            def get(_user, _index, nbytes, produce=produce):
                try:
                    arr = np.ascontiguousarray(produce()).reshape(-1).view(np.uint8)
                except BaseException as e:   # must not unwind through the C caller
                    self._error = e
                    return None
                if arr.size != nbytes:
                    return None
                self._current = arr          # valid until the next call
                return arr.ctypes.data
            fn = _lib.ggq_gguf_data_fn(get)
            self._keep.append(fn)
        elif data is not None:
            arr = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray))
                                       else data).reshape(-1).view(np.uint8)
            self._keep.append(arr)
            ptr = arr.ctypes.data
        _check(lib.ggq_gguf_writer_tensor(self._h, n, len(n), int(ty), shp, len(shape), ptr, fn, None))

    def written_bytes(self):
        return lib.ggq_gguf_writer_written_bytes(self._h)

    def finish(self):
        n = ctypes.c_uint64()
        try:
            code = lib.ggq_gguf_writer_finish(self._h, ctypes.byref(n))
            if self._error is not None:
                raise self._error
            _check(code)
        finally:
            self.close()
        return n.value

    def close(self):
        if self._h:
            lib.ggq_gguf_writer_destroy(self._h)
            self._h = None
            self._keep = []
            self._current = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_shards(alignment, meta_kvs, tensors, max_tensor_count=2 ** 64 - 1, max_file_size=2 ** 64 - 1,
                no_tensor_first=False):
    """Shard index per tensor as Content::write_files plans it (xtask/src/utils/write.rs:21-51).
    meta_kvs: [(key, type id, raw value bytes)]; tensors: [(name, ggml type, shape)]."""
    keep = []
    kvs = (_lib.ggq_gguf_kv * max(1, len(meta_kvs)))()
    for i, (k, ty, v) in enumerate(meta_kvs):
        kb, vb = ctypes.create_string_buffer(k.encode(), len(k.encode())), ctypes.create_string_buffer(bytes(v), len(v))
        keep += [kb, vb]
        kvs[i].key, kvs[i].key_len = ctypes.addressof(kb), len(kb)
        kvs[i].type, kvs[i].value, kvs[i].value_len = ty, ctypes.addressof(vb), len(vb)
    ts = (_lib.ggq_gguf_tensor * max(1, len(tensors)))()
    for i, (name, ty, shape) in enumerate(tensors):
        nb = ctypes.create_string_buffer(name.encode(), len(name.encode()))
        keep.append(nb)
        ts[i].name, ts[i].name_len, ts[i].ggml_type, ts[i].ndim = ctypes.addressof(nb), len(nb), int(ty), len(shape)
        for j, d in enumerate(shape[:4]):
            ts[i].shape[j] = d
    shard_of = (ctypes.c_uint32 * max(1, len(tensors)))()
    n = ctypes.c_size_t()
    _check(lib.ggq_gguf_plan_shards(alignment, kvs, len(meta_kvs), ts, len(tensors), max_tensor_count, max_file_size,
                                    int(no_tensor_first), shard_of, ctypes.byref(n)))
    return list(shard_of[:len(tensors)]), n.value


# ---- whole-file decode ---------------------------------------------------------------------

_ctx_cache = threading.local()


def _thread_context(device, staging_bytes):
    """One streamed-upload context per (thread, device, staging size): creating one pins its staging memory,
    which costs about as long as decoding a 1 B-parameter model.  It stays (staging_bytes of page-locked host
    memory plus as much device memory) until the thread calls close_thread_contexts() or ends."""
    from .quants import Context
    cache = _ctx_cache.__dict__.setdefault("ctx", {})
    key = (int(device), int(staging_bytes))
    if key not in cache:
        cache[key] = Context(device, staging_bytes)
    return cache[key]


def close_thread_contexts():
    """Release the calling thread's cached upload contexts (their pinned and device staging memory)."""
    cache = _ctx_cache.__dict__.pop("ctx", {})
    for c in cache.values():
        c.close()


def dequantize_file(gguf, out="f32", device=0, staging_bytes=64 << 20, only=None, group_bytes=1 << 30,
                    stats=None):
    """Decode every tensor of an open GGuf that holds data and has a block type this library
    decodes; returns {name: device tensor}.  F32 tensors are copied as they are (the reference's
    cast leaves them alone too).

    The packed bytes are streamed host -> device through one Context (pinned staging, helper
    threads); tensors are gathered into groups of about `group_bytes` packed bytes and each group is
    decoded by ONE launch plan while the next group's bytes are still on their way.  Inside a group the
    tensors of one block type are placed back to back, in the packed arena and in the output (the returned
    tensors of a group and type are views of one allocation, which lives as long as any of them), so the
    plan sees one long run per type and decodes it with one launch at the single-tensor kernel's rate.  A
    group's packed bytes are released as soon as its plan is enqueued: peak device memory is the decoded
    tensors plus about two groups of packed bytes.  `stats`, if a dict, receives the counts."""
    import torch
    from .quants import Batch, SUPPORTED
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[out]
    out_ty = {"f32": GGmlType.F32, "f16": GGmlType.F16, "bf16": GGmlType.BF16}[out]
    supported = {int(x) for x in SUPPORTED}
    dev = f"cuda:{device}"
    ctx = _thread_context(device, staging_bytes)
    res, plans = {}, []
    counts = {"tensors": 0, "groups": 0, "launches": 0, "packed_bytes": 0}

    def flush(group):
        if not group:
            return
        # Memory that torch's allocator hands out now may be a block this function released a moment ago (an
        # earlier group's packed arena) that a plan queued on the current stream is still reading: the uploads
        # below run on the context's streams, so they are ordered behind the current stream explicitly.
        ctx.wait_stream()
        first = {}
        for t in group:
            first.setdefault(int(t.ty), len(first))
        group = sorted(group, key=lambda t: first[int(t.ty)])        # stable: file order inside a type
        offs, total, elems = [], 0, {}
        for k, t in enumerate(group):
            if k and int(group[k - 1].ty) != int(t.ty):
                total = (total + 31) & ~31
            offs.append(total)
            total += t.nbytes
            elems[int(t.ty)] = elems.get(int(t.ty), 0) + t.numel
        arena = torch.empty(total, dtype=torch.uint8, device=dev)
        outs = {ty: torch.empty(n, dtype=dt, device=dev) for ty, n in elems.items()}
        at = {ty: 0 for ty in elems}
        table = []
        for t, o in zip(group, offs):
            ctx.upload_async(arena[o:o + t.nbytes], t.bytes())
            ty = int(t.ty)
            dst = outs[ty][at[ty]:at[ty] + t.numel]
            at[ty] += t.numel
            res[t.name] = dst
            table.append((t.ty, arena[o:o + t.nbytes], dst))
        # the plan needs only addresses: it is built while the last copies are still in flight
        b = Batch(table, out_ty)
        ctx.fence()                     # the plan waits on the device for this group's copies; the host goes on
        b.run()                         # staging the next group meanwhile
        plans.append(b)                 # (the plan keeps its tensors -- and with them the arena -- until it is dropped)
        counts["groups"] += 1
        counts["launches"] += b.launches
        counts["packed_bytes"] += total
        b.release_tensors()             # enqueued: the arena goes back to the allocator, stream-ordered

    try:
        with torch.cuda.device(device):
            group, held = [], 0
            for t in gguf.tensors:
                if only is not None and t.name not in only:
                    continue
                if not t.ptr:
                    continue
                if t.ty == GGmlType.F32:
                    ctx.wait_stream()
                    dst = torch.empty(t.numel, dtype=torch.float32, device=dev)
                    ctx.upload_async(dst.view(torch.uint8), t.bytes())
                    res[t.name] = dst
                    continue
                if t.ty not in supported:
                    continue
                group.append(t)
                held += t.nbytes
                counts["tensors"] += 1
                if held >= group_bytes:
                    flush(group)
                    group, held = [], 0
            flush(group)
            ctx.sync()
            torch.cuda.synchronize()
    finally:
        ctx.sync()                      # also on the error path: nothing of this call may still be in flight
        for b in plans:
            b.close()
    if stats is not None:
        stats.update(counts)
    return res
