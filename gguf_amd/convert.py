"""The reference's `convert` pipeline for the cast operator, with the per-tensor work on the GPU:

    read    Content::new / merge_file        xtask/src/utils/read.rs:4-63
    cast    Operator::cast, Content::cast    xtask/src/utils/operator/cast.rs:11-138, 180-216
    write   Content::write_files             xtask/src/utils/write.rs:5-100

Input shards are mapped, not read; a tensor that changes type is produced only when the writer of
its shard reaches it (the reference's lazy DataPromise): packed bytes go to the device, are decoded /
re-encoded there, and come back through a pinned buffer that the C++ writer hands to write(2).  One
producer thread per shard keeps the GPU one tensor ahead of the file.  Output file names are the
caller's business (`stem`); the reference's name grammar is not part of this library.
"""
import os
import queue
import re
import threading

import numpy as np

from ._lib import lib
from .gguf import GGuf, GGufError, GGufFileWriter, plan_shards
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


class Tensor:
    """Tensor of Content (xtask/src/utils/mod.rs:92-96): type, shape, and where the bytes come from."""
    __slots__ = ("name", "ty", "shape", "src_ty", "src")

    def __init__(self, name, ty, shape, src):
        self.name, self.ty, self.shape, self.src_ty, self.src = name, GGmlType(ty), list(shape), GGmlType(ty), src

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self):
        return self.numel // lib.ggq_gguf_type_elems(int(self.ty)) * lib.ggq_gguf_type_bytes(int(self.ty))


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
                self.tensors[t.name] = Tensor(t.name, t.ty, t.shape, t.bytes() if t.ptr else None)

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
        for t in self.tensors.values():
            to = target_type(arch, t.name, t.shape, types)
            if to is not None and to != t.ty:
                if to not in _FLOATS and to not in _QUANTIZABLE:
                    raise Unsupported(f"{t.name}: no encoder for {to.name}")
                if t.ty not in _FLOATS and block_bytes(t.ty) == 0:
                    raise Unsupported(f"{t.name}: no decoder for {t.ty.name}")
                be = lib.ggq_gguf_type_elems(int(to))
                if t.shape[0] % be:
                    raise Indivisible(f"{t.name}: row of {t.shape[0]} is not a multiple of {be}")
                t.ty = to

    # ---- writing ------------------------------------------------------------------------------

    def write_files(self, out_dir, stem, max_tensor_count=2 ** 63, max_file_size=2 ** 63, no_tensor_first=False,
                    write_data=True, device=0):
        """Content::write_files (write.rs:5-100): plan shards, then one writer thread per shard.
        Returns [(path, n_tensors, n_meta_kvs, n_bytes)] (FileInfo)."""
        names = list(self.tensors)
        kvs = [(k, ty, raw) for k, (ty, raw) in self.meta_kvs.items()]
        shard_of, n_shards = plan_shards(self.alignment, kvs, [(n, self.tensors[n].ty, self.tensors[n].shape)
                                                               for n in names],
                                         max_tensor_count, max_file_size, no_tensor_first)
        shards = [[] for _ in range(n_shards)]
        for name, s in zip(names, shard_of):
            shards[s].append(self.tensors[name])
        os.makedirs(out_dir, exist_ok=True)
        paths = [os.path.join(out_dir, f"{stem}.gguf" if n_shards == 1 else f"{stem}-{i + 1:05}-of-{n_shards:05}.gguf")
                 for i in range(n_shards)]
        results, errors = [None] * n_shards, []

        def run(i):
            try:
                results[i] = self._write_shard(i, paths[i], shards[i], kvs, write_data, device)
            except BaseException as e:      # reported by the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=run, args=(i,)) for i in range(n_shards)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
        return results

    def _write_shard(self, i, path, tensors, kvs, write_data, device):
        n_kv = (len(kvs) if i == 0 else 0) + 1
        w = GGufFileWriter(path, len(tensors), n_kv, write_data=write_data)
        w.write_alignment(self.alignment)
        if i == 0:
            for k, ty, raw in kvs:
                w.write_meta_kv(k, ty, raw)
        lazy = [t for t in tensors if t.ty != t.src_ty]
        producer = _Producer(lazy, device) if (write_data and lazy) else None
        try:
            for t in tensors:
                if t.ty == t.src_ty or not write_data:
                    w.write_tensor(t.name, t.ty, t.shape, t.src if write_data else None)
                else:
                    w.write_tensor(t.name, t.ty, t.shape, producer.next)
            n_bytes = w.finish()
        finally:
            w.close()
            if producer:
                producer.stop()
        return path, len(tensors), n_kv, n_bytes


class _Producer:
    """Casts a shard's tensors on the GPU in file order, one tensor ahead of the writer, into two
    pinned host buffers that alternate."""

    def __init__(self, tensors, device):
        import torch
        self._torch = torch
        self._tensors = tensors
        self._device = device
        size = max(t.nbytes for t in tensors)
        self._bufs = [torch.empty(size, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._free = queue.Queue()
        for b in range(2):
            self._free.put(b)
        self._ready = queue.Queue()
        self._held = None
        self._stop = False
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _run(self):
        torch = self._torch
        try:
            with torch.cuda.device(self._device):
                stream = torch.cuda.Stream()
                with torch.cuda.stream(stream):
                    for t in self._tensors:
                        b = self._free.get()
                        if b is None or self._stop:
                            return
                        if t.src is None:
                            raise ValueError(f"{t.name}: the input file holds no data for this tensor")
                        src = torch.from_numpy(t.src).to(f"cuda:{self._device}", non_blocking=False)
                        out = cast_device(t.shape[0], src, t.src_ty, t.ty)
                        host = self._bufs[b][:out.numel()]
                        host.copy_(out, non_blocking=True)
                        stream.synchronize()
                        self._ready.put((b, host.numpy()))
        except BaseException as e:
            self._ready.put(e)

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


def convert(inputs, out_dir, stem, cast=None, **write_args):
    """`convert <inputs> -x cast:<types>`: read, mark, write.  Returns the FileInfo list."""
    with Content(inputs) as c:
        if cast:
            c.cast(cast)
        return c.write_files(out_dir, stem, **write_args)
