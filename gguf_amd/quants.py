"""Host-side mirror of the reference's block-quantization interface.

Mirrors ggml-quants/src/lib.rs: `DataBlock` (ID, COUNT), `QuantExt::dequantize_slice`
with its two errors, and the block types re-exported by structs.rs:27-51 -- with
the decode done by the HIP kernels behind the C ABI of include/ggq.h.

    Q8_0.dequantize_slice(dst, src)       # <Q8_0 as QuantExt<f32, 32>>::dequantize_slice

`src` holds packed blocks (uint8), `dst` the decoded elements (float32, float16
or bfloat16).  Both on the same CUDA/HIP device: the kernel runs on that device on
torch's current stream.  Both in host memory (numpy arrays or CPU tensors): the
data is staged through the GPU by the library's pinned pipeline.
"""
import ctypes
import enum

import numpy as np

from . import _lib
from ._lib import lib


class GGmlType(enum.IntEnum):
    """ggus/src/tensor.rs:14-51 (discriminants verbatim)."""
    F32 = 0
    F16 = 1
    Q4_0 = 2
    Q4_1 = 3
    Q5_0 = 6
    Q5_1 = 7
    Q8_0 = 8
    Q8_1 = 9
    Q2K = 10
    Q3K = 11
    Q4K = 12
    Q5K = 13
    Q6K = 14
    Q8K = 15
    IQ2XXS = 16
    IQ2XS = 17
    IQ3XXS = 18
    IQ1S = 19
    IQ4NL = 20
    IQ3S = 21
    IQ2S = 22
    IQ4XS = 23
    I8 = 24
    I16 = 25
    I32 = 26
    I64 = 27
    F64 = 28
    IQ1M = 29
    BF16 = 30
    Q4_0_4_4 = 31
    Q4_0_4_8 = 32
    Q4_0_8_8 = 33


class QuantizeError(Exception):
    """ggml-quants/src/lib.rs:107-113."""


class Indivisible(QuantizeError):
    pass


class LengthMismatch(QuantizeError):
    pass


class Unsupported(QuantizeError):
    """The reference panics (`todo!()`) for these; this library reports them."""


class DeviceError(RuntimeError):
    pass


class Misaligned(QuantizeError):
    pass


def check(status):
    if status == _lib.OK:
        return
    if status == _lib.INDIVISIBLE:
        raise Indivisible()
    if status == _lib.LENGTH_MISMATCH:
        raise LengthMismatch()
    if status == _lib.UNSUPPORTED:
        raise Unsupported()
    if status == _lib.MISALIGNED:
        raise Misaligned()
    msg = lib.ggq_status_str(status).decode()
    if status == _lib.DEVICE:
        raise DeviceError(f"{msg}: {lib.ggq_last_error().decode()}")
    raise ValueError(msg)


def block_elems(ty):
    return lib.ggq_block_elems(int(ty))


def block_bytes(ty):
    return lib.ggq_block_bytes(int(ty))


def type_name(ty):
    s = lib.ggq_type_name(int(ty))
    return s.decode() if s else None


def elements_to_bytes(ty, shape):
    """GGmlTypeSize::elements_to_bytes (ggus/src/tensor.rs:83-96); shape[0] is contiguous."""
    arr = (ctypes.c_uint64 * max(1, len(shape)))(*shape)
    out = ctypes.c_uint64()
    check(lib.ggq_elements_to_bytes(int(ty), arr, len(shape), ctypes.byref(out)))
    return out.value


def _out_type(dst):
    name = str(dst.dtype)
    if name in ("torch.float32", "float32"):
        return GGmlType.F32
    if name in ("torch.float16", "float16"):
        return GGmlType.F16
    if name == "torch.bfloat16":
        return GGmlType.BF16
    raise TypeError(f"unsupported output element type {name}")


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _describe(x):
    """(pointer, element count, itemsize, on_device)"""
    if _is_torch(x):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr(), x.numel(), x.element_size(), x.is_cuda
    a = np.asarray(x)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be contiguous")
    return a.ctypes.data, a.size, a.itemsize, False


def dequantize_slice(ty, dst, src, stream=None):
    """QuantExt::dequantize_slice (ggml-quants/src/lib.rs:135-147) for block type `ty`.

    `src`: uint8 buffer of packed blocks.  Its block count is len(src) / block_bytes(ty);
    a length that is not a whole number of blocks cannot be expressed as the
    reference's `&[Blk]` and raises ValueError.
    """
    ty = GGmlType(int(ty))
    bb = block_bytes(ty)
    if bb == 0:
        raise Unsupported()
    sp, sn, ss, s_dev = _describe(src)
    dp, dn, _, d_dev = _describe(dst)
    nbytes = sn * ss
    if nbytes % bb:
        raise ValueError(f"{nbytes} bytes is not a whole number of {bb}-byte blocks")
    blocks = nbytes // bb
    out_ty = _out_type(dst)
    if s_dev != d_dev:
        raise ValueError("src and dst must both be on the device or both on the host")
    if s_dev:
        import torch
        if src.device != dst.device:
            raise ValueError("src and dst are on different devices")
        with torch.cuda.device(dst.device):
            st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
            check(lib.ggq_dequantize_device(int(ty), int(out_ty), dp, dn, sp, blocks, st))
    else:
        check(lib.ggq_dequantize_slice(int(ty), int(out_ty), dp, dn, sp, blocks))


def quantize_slice(ty, dst, src, stream=None):
    """QuantExt::quantize_slice (ggml-quants/src/lib.rs:121-133) for block type `ty`.

    `src`: elements (float32, float16 or bfloat16); `dst`: uint8 buffer that receives the
    packed blocks, len(dst) / block_bytes(ty) blocks.
    """
    ty = GGmlType(int(ty))
    bb = block_bytes(ty)
    if bb == 0:
        raise Unsupported()
    sp, sn, _, s_dev = _describe(src)
    dp, dn, ds, d_dev = _describe(dst)
    nbytes = dn * ds
    if nbytes % bb:
        raise ValueError(f"{nbytes} bytes is not a whole number of {bb}-byte blocks")
    blocks = nbytes // bb
    in_ty = _out_type(src)
    if s_dev != d_dev:
        raise ValueError("src and dst must both be on the device or both on the host")
    if s_dev:
        import torch
        if src.device != dst.device:
            raise ValueError("src and dst are on different devices")
        with torch.cuda.device(dst.device):
            st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
            check(lib.ggq_quantize_device(int(ty), int(in_ty), dp, blocks, sp, sn, st))
    else:
        check(lib.ggq_quantize_slice(int(ty), int(in_ty), dp, blocks, sp, sn))


def selftest_fast_division():
    """ggq_selftest_fast_division on the current device: the quantizer kernels' short division sequences against the
    general IEEE division for every float in their range (about a second).  Returns ({"a/15": n, "a/31": n,
    "a/127": n, "a/-127": n, "1/d": n}, inputs tested per constant); every n must be 0 on a GPU the library
    quantizes on (the reference divides with `/`: ggml-quants/src/structs/q8_0.rs:31-32 and the other formats)."""
    bad = (ctypes.c_uint64 * 5)()
    tested = ctypes.c_uint64(0)
    check(lib.ggq_selftest_fast_division(bad, ctypes.byref(tested)))
    return dict(zip(("a/15", "a/31", "a/127", "a/-127", "1/d"), (int(x) for x in bad))), int(tested.value)


class _Block:
    """DataBlock (ggml-quants/src/lib.rs:11-21) + QuantExt for one block type."""
    TYPE = None

    @classmethod
    def ID(cls):
        return type_name(cls.TYPE)

    @classmethod
    def COUNT(cls):
        return block_elems(cls.TYPE)

    @classmethod
    def SIZE(cls):
        return block_bytes(cls.TYPE)

    @classmethod
    def dequantize_slice(cls, dst, src, stream=None):
        return dequantize_slice(cls.TYPE, dst, src, stream)

    @classmethod
    def quantize_slice(cls, dst, src, stream=None):
        return quantize_slice(cls.TYPE, dst, src, stream)


def _mk(name, ty):
    return type(name, (_Block,), {"TYPE": ty, "__doc__": f"Block type {name} (GGmlType {int(ty)})."})


f16 = _mk("f16", GGmlType.F16)
bf16 = _mk("bf16", GGmlType.BF16)
Q4_0 = _mk("Q4_0", GGmlType.Q4_0)
Q4_1 = _mk("Q4_1", GGmlType.Q4_1)
Q5_0 = _mk("Q5_0", GGmlType.Q5_0)
Q5_1 = _mk("Q5_1", GGmlType.Q5_1)
Q8_0 = _mk("Q8_0", GGmlType.Q8_0)
Q8_1 = _mk("Q8_1", GGmlType.Q8_1)
Q2K = _mk("Q2K", GGmlType.Q2K)
Q3K = _mk("Q3K", GGmlType.Q3K)
Q4K = _mk("Q4K", GGmlType.Q4K)
Q5K = _mk("Q5K", GGmlType.Q5K)
Q6K = _mk("Q6K", GGmlType.Q6K)
Q8K = _mk("Q8K", GGmlType.Q8K)
IQ4NL = _mk("IQ4NL", GGmlType.IQ4NL)
IQ4XS = _mk("IQ4XS", GGmlType.IQ4XS)

BLOCK_TYPES = [f16, bf16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1,
               Q2K, Q3K, Q4K, Q5K, Q6K, Q8K, IQ4NL, IQ4XS]
SUPPORTED = [b.TYPE for b in BLOCK_TYPES]


class Batch:
    """A launch plan over many device-resident tensors (ggq_batch_*)."""

    def __init__(self, tensors, out_ty=GGmlType.F32):
        """tensors: iterable of (ggml_type, src_uint8_cuda_tensor, dst_cuda_tensor)."""
        tensors = list(tensors)
        self._keep = tensors
        arr = (_lib.ggq_tensor * max(1, len(tensors)))()
        for i, (ty, src, dst) in enumerate(tensors):
            bb = block_bytes(ty)
            if bb == 0:
                raise Unsupported()
            nbytes = src.numel() * src.element_size()
            if nbytes % bb:
                raise ValueError("src is not a whole number of blocks")
            blocks = nbytes // bb
            if dst.numel() != blocks * block_elems(ty):
                raise LengthMismatch()
            arr[i] = _lib.ggq_tensor(int(ty), 0, src.data_ptr(), dst.data_ptr(), blocks)
        h = ctypes.c_void_p()
        check(lib.ggq_batch_create(arr, len(tensors), int(out_ty), ctypes.byref(h)))
        self._h = h

    @property
    def launches(self):
        return lib.ggq_batch_launches(self._h)

    @property
    def table_launches(self):
        """How many of the launches go through a tile table (the others are runs with a launch of their own)."""
        return lib.ggq_batch_table_launches(self._h)

    def run(self, stream=None):
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        check(lib.ggq_batch_run(self._h, stream))

    def release_tensors(self):
        """Drop the plan's references to its tensors.  For a plan that will not run again, once its last run is
        enqueued: with a stream-ordered allocator (torch's) the memory can then be handed out again for later work
        on the same stream while the plan's kernels are still queued."""
        self._keep = None

    def close(self):
        if self._h:
            lib.ggq_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QuantBatch(Batch):
    """A launch plan that QUANTIZES many device-resident tensors (ggq_quant_batch_create): one kernel launch per
    target block type instead of one per tensor -- the loop over tensors of the reference's cast operator
    (xtask/src/utils/operator/cast.rs:73-90) when its iterations end in quantize_slice (cast.rs:140-148)."""

    def __init__(self, tensors, in_ty=None):
        """tensors: iterable of (target ggml_type, src cuda tensor of elements, dst uint8 cuda tensor); all sources
        share one element type (float32, float16 or bfloat16), which `in_ty` names when the table is empty."""
        tensors = list(tensors)
        self._keep = tensors
        arr = (_lib.ggq_tensor * max(1, len(tensors)))()
        for i, (ty, src, dst) in enumerate(tensors):
            bb, be = block_bytes(ty), block_elems(ty)
            if bb == 0:
                raise Unsupported()
            this_in = _out_type(src)
            if in_ty is None:
                in_ty = this_in
            if int(this_in) != int(in_ty):
                raise ValueError("all sources of one quantize plan must have the same element type")
            nbytes = dst.numel() * dst.element_size()
            if src.numel() % be:
                raise Indivisible()
            if nbytes % bb or nbytes // bb != src.numel() // be:
                raise LengthMismatch()
            arr[i] = _lib.ggq_tensor(int(ty), 0, src.data_ptr(), dst.data_ptr(), nbytes // bb)
        h = ctypes.c_void_p()
        check(lib.ggq_quant_batch_create(arr, len(tensors), int(in_ty if in_ty is not None else GGmlType.F32),
                                         ctypes.byref(h)))
        self._h = h


class Context:
    """Streamed host -> device decode on one GPU (ggq_ctx_*)."""

    def __init__(self, device=0, staging_bytes=64 << 20):
        h = ctypes.c_void_p()
        check(lib.ggq_ctx_create(int(device), int(staging_bytes), ctypes.byref(h)))
        self._h = h

    def dequantize_async(self, ty, dst, src_host):
        sp, sn, ss, s_dev = _describe(src_host)
        if s_dev:
            raise ValueError("src must be host memory")
        bb = block_bytes(ty)
        if bb == 0:
            raise Unsupported()
        nbytes = sn * ss
        if nbytes % bb:
            raise ValueError("src is not a whole number of blocks")
        check(lib.ggq_dequantize_async(self._h, int(ty), int(_out_type(dst)), dst.data_ptr(),
                                       dst.numel(), sp, nbytes // bb))

    def upload_async(self, dst, src_host):
        """Copy host bytes (numpy array / CPU tensor, pageable or pinned) into a uint8 device tensor."""
        sp, sn, ss, s_dev = _describe(src_host)
        dp, dn, ds, d_dev = _describe(dst)
        if s_dev or not d_dev:
            raise ValueError("src must be host memory and dst device memory")
        if sn * ss != dn * ds:
            raise LengthMismatch()
        check(lib.ggq_upload_async(self._h, dp, sp, sn * ss))

    def fence(self, stream=None):
        """Later work on `stream` (default: torch's current stream) waits, on the device, for everything
        enqueued on this context so far."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        check(lib.ggq_ctx_fence(self._h, stream))

    def wait_stream(self, stream=None):
        """Work enqueued on this context from now on waits, on the device, for everything enqueued on `stream`
        (default: torch's current stream) so far.  Call it before uploading into memory that torch's allocator may
        have handed out before: the allocator orders reuse only on the stream the memory was allocated on."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        check(lib.ggq_ctx_wait_stream(self._h, stream))

    def sync(self):
        check(lib.ggq_sync(self._h))

    def close(self):
        if self._h:
            lib.ggq_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
