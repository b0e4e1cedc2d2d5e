"""ctypes binding of oracle/libggq_oracle.so (see ggq_oracle.c for what it restates).

Test infrastructure: the checker, never the thing measured or shipped.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libggq_oracle.so")

OK, INDIVISIBLE, LENGTH_MISMATCH, UNSUPPORTED = 0, 1, 2, 3

# GGmlType discriminants (ggus/src/tensor.rs:14-51)
F32, F16, BF16, F64 = 0, 1, 30, 28
Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 2, 3, 6, 7, 8, 9
Q2K, Q3K, Q4K, Q5K, Q6K, Q8K = 10, 11, 12, 13, 14, 15
IQ4NL, IQ4XS = 20, 23

_NP_OUT = {F32: np.float32, F16: np.uint16, BF16: np.uint16}


def build(force=False):
    """Compile the C restatement (gcc, a second or two)."""
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ggq_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"],
                              stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.ggo_block_elems.restype = ctypes.c_size_t
        L.ggo_block_elems.argtypes = [ctypes.c_uint32]
        L.ggo_block_bytes.restype = ctypes.c_size_t
        L.ggo_block_bytes.argtypes = [ctypes.c_uint32]
        L.ggo_f16_to_f32.restype = ctypes.c_float
        L.ggo_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.ggo_f32_to_f16.restype = ctypes.c_uint16
        L.ggo_f32_to_f16.argtypes = [ctypes.c_float]
        L.ggo_bf16_to_f32.restype = ctypes.c_float
        L.ggo_bf16_to_f32.argtypes = [ctypes.c_uint16]
        L.ggo_f32_to_bf16.restype = ctypes.c_uint16
        L.ggo_f32_to_bf16.argtypes = [ctypes.c_float]
        L.ggo_dequantize_slice_mt.restype = ctypes.c_int
        L.ggo_dequantize_slice_mt.argtypes = [
            ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t,
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        L.ggo_quantize_slice.restype = ctypes.c_int
        L.ggo_quantize_slice.argtypes = [
            ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t,
            ctypes.c_void_p, ctypes.c_size_t]
        L.ggo_scale.restype = ctypes.c_int
        L.ggo_scale.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double]
        L.ggo_f64_to_f16.restype = ctypes.c_uint16
        L.ggo_f64_to_f16.argtypes = [ctypes.c_double]
        L.ggo_f64_to_bf16.restype = ctypes.c_uint16
        L.ggo_f64_to_bf16.argtypes = [ctypes.c_double]
        _lib = L
    return _lib


def block_elems(ty):
    return lib().ggo_block_elems(ty)


def block_bytes(ty):
    return lib().ggo_block_bytes(ty)


def dequantize_status(ty, out_ty, dst, dst_elems, src, src_blocks, threads=1):
    """Raw call: numpy buffers in, status out (no shape fixing)."""
    return lib().ggo_dequantize_slice_mt(
        ty, out_ty, dst.ctypes.data, dst_elems, src.ctypes.data, src_blocks, threads)


def dequantize(ty, packed, out_ty=F32, threads=1):
    """Decode a whole packed byte buffer.  f16/bf16 outputs come back as uint16 bit patterns."""
    packed = np.ascontiguousarray(packed).view(np.uint8).reshape(-1)
    bb, n = block_bytes(ty), block_elems(ty)
    assert n and packed.size % bb == 0, (ty, packed.size)
    blocks = packed.size // bb
    out = np.empty(blocks * n, dtype=_NP_OUT[out_ty])
    st = dequantize_status(ty, out_ty, out, out.size, packed, blocks, threads)
    assert st == OK, st
    return out


def quantize_status(ty, in_ty, dst, dst_blocks, src, src_elems):
    return lib().ggo_quantize_slice(ty, in_ty, dst.ctypes.data, dst_blocks,
                                    src.ctypes.data, src_elems)


def quantize(ty, values, in_ty=F32):
    """Pack values (float32, or uint16 bit patterns for f16/bf16 input) into blocks."""
    values = np.ascontiguousarray(values).reshape(-1)
    bb, n = block_bytes(ty), block_elems(ty)
    assert values.size % n == 0
    blocks = values.size // n
    out = np.empty(blocks * bb, dtype=np.uint8)
    st = quantize_status(ty, in_ty, out, blocks, values, values.size)
    assert st == OK, st
    return out


def f16_to_f32(bits):
    L = lib()
    return np.array([L.ggo_f16_to_f32(int(b)) for b in np.asarray(bits).reshape(-1)],
                    dtype=np.float32)


def f32_to_f16(vals):
    L = lib()
    return np.array([L.ggo_f32_to_f16(float(v)) for v in
                     np.asarray(vals, dtype=np.float32).reshape(-1)], dtype=np.uint16)


def scale(ty, values, factor):
    """Every element times `factor` in the tensor's type (float32 / float64 arrays, or uint16 bit
    patterns for f16 / bf16)."""
    values = np.ascontiguousarray(values).reshape(-1)
    out = np.empty_like(values)
    st = lib().ggo_scale(ty, out.ctypes.data, values.ctypes.data, values.size, float(factor))
    assert st == OK, st
    return out
