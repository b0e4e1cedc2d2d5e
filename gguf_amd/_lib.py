"""ctypes binding of libggq.so, the C-ABI product library (include/ggq.h).

There is no fallback: if the library has not been built, importing this module
raises; if no HIP device is usable, compute calls return GGQ_DEVICE, which the
wrappers in gguf_amd.quants turn into DeviceError.
"""
import ctypes
import os

# torch must be loaded first.  libggq.so needs libamdhip64.so.7; PyTorch ships its
# own copy of that runtime.  The dynamic linker binds libggq to whichever copy is
# already in the process, and the two must be ONE runtime, or torch's device pointers
# and streams mean nothing to the kernels here (a second runtime in the process
# does not even see the GPU).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libggq.so")

OK, INDIVISIBLE, LENGTH_MISMATCH, UNSUPPORTED, DEVICE, MISALIGNED, INVALID, TOO_LARGE = range(8)


class ggq_tensor(ctypes.Structure):
    _fields_ = [("ggml_type", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("d_src", ctypes.c_void_p), ("d_dst", ctypes.c_void_p),
                ("blocks", ctypes.c_uint64)]


class ggq_gguf_info(ctypes.Structure):
    _fields_ = [("version", ctypes.c_uint32), ("tensor_count", ctypes.c_uint64), ("kv_count", ctypes.c_uint64),
                ("alignment", ctypes.c_uint64), ("data_offset", ctypes.c_uint64), ("data_len", ctypes.c_uint64),
                ("data_avail", ctypes.c_uint64)]


class ggq_gguf_tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_void_p), ("name_len", ctypes.c_size_t), ("ggml_type", ctypes.c_uint32),
                ("ndim", ctypes.c_uint32), ("shape", ctypes.c_uint64 * 4), ("offset", ctypes.c_uint64),
                ("nbytes", ctypes.c_uint64), ("data", ctypes.c_void_p)]


class ggq_gguf_kv(ctypes.Structure):
    _fields_ = [("key", ctypes.c_void_p), ("key_len", ctypes.c_size_t), ("type", ctypes.c_uint32),
                ("value", ctypes.c_void_p), ("value_len", ctypes.c_size_t)]


# DataFuture::get as a C callback: (user, tensor index, nbytes) -> pointer to the bytes
ggq_gguf_data_fn = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64)

# every symbol include/ggq.h and include/ggq_gguf.h declare: name -> (restype, argtypes)
_vp, _sz, _u32, _int = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
SYMBOLS = {
    "ggq_block_elems": (_sz, [_u32]),
    "ggq_block_bytes": (_sz, [_u32]),
    "ggq_type_name": (ctypes.c_char_p, [_u32]),
    "ggq_elements_to_bytes": (_int, [_u32, ctypes.POINTER(ctypes.c_uint64), _sz,
                                     ctypes.POINTER(ctypes.c_uint64)]),
    "ggq_status_str": (ctypes.c_char_p, [_int]),
    "ggq_last_error": (ctypes.c_char_p, []),
    "ggq_device_count": (_int, []),
    "ggq_dequantize_slice": (_int, [_u32, _u32, _vp, _sz, _vp, _sz]),
    "ggq_dequantize_device": (_int, [_u32, _u32, _vp, _sz, _vp, _sz, _vp]),
    "ggq_quantize_slice": (_int, [_u32, _u32, _vp, _sz, _vp, _sz]),
    "ggq_quantize_device": (_int, [_u32, _u32, _vp, _sz, _vp, _sz, _vp]),
    "ggq_concat_rows_device": (_int, [_u32, ctypes.c_uint64, _vp, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint64), _sz, _vp]),
    "ggq_split_rows_device": (_int, [_u32, ctypes.c_uint64, _vp, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint64), _sz, _vp]),
    "ggq_scale_device": (_int, [_u32, _vp, _vp, _sz, ctypes.c_double, _vp]),
    "ggq_permute_qk_device": (_int, [_u32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp, _vp]),
    "ggq_batch_create": (_int, [ctypes.POINTER(ggq_tensor), _sz, _u32, ctypes.POINTER(_vp)]),
    "ggq_quant_batch_create": (_int, [ctypes.POINTER(ggq_tensor), _sz, _u32, ctypes.POINTER(_vp)]),
    "ggq_batch_run": (_int, [_vp, _vp]),
    "ggq_batch_launches": (_sz, [_vp]),
    "ggq_batch_table_launches": (_sz, [_vp]),
    "ggq_batch_destroy": (None, [_vp]),
    "ggq_ctx_create": (_int, [_int, _sz, ctypes.POINTER(_vp)]),
    "ggq_dequantize_async": (_int, [_vp, _u32, _u32, _vp, _sz, _vp, _sz]),
    "ggq_upload_async": (_int, [_vp, _vp, _vp, _sz]),
    "ggq_ctx_fence": (_int, [_vp, _vp]),
    "ggq_ctx_wait_stream": (_int, [_vp, _vp]),
    "ggq_dequantize_noalloc": (_int, [_u32, _u32, _vp, _sz, _vp, _sz, _vp]),
    "ggq_selftest_fast_division": (_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "ggq_sync": (_int, [_vp]),
    "ggq_ctx_destroy": (None, [_vp]),
    "ggq_gguf_parse": (_int, [_vp, _sz, ctypes.POINTER(_vp)]),
    "ggq_gguf_open": (_int, [ctypes.c_char_p, ctypes.POINTER(_vp)]),
    "ggq_gguf_close": (None, [_vp]),
    "ggq_gguf_last_name": (ctypes.c_char_p, []),
    "ggq_gguf_error_str": (ctypes.c_char_p, [_int]),
    "ggq_gguf_get_info": (_int, [_vp, ctypes.POINTER(ggq_gguf_info)]),
    "ggq_gguf_tensor_at": (_int, [_vp, _sz, ctypes.POINTER(ggq_gguf_tensor)]),
    "ggq_gguf_find_tensor": (_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ggq_gguf_tensor)]),
    "ggq_gguf_kv_at": (_int, [_vp, _sz, ctypes.POINTER(ggq_gguf_kv)]),
    "ggq_gguf_find_kv": (_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ggq_gguf_kv)]),
    "ggq_gguf_type_bytes": (_sz, [_u32]),
    "ggq_gguf_type_elems": (_sz, [_u32]),
    "ggq_gguf_writer_create": (_int, [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, _int, ctypes.POINTER(_vp)]),
    "ggq_gguf_writer_alignment": (_int, [_vp, _u32]),
    "ggq_gguf_writer_kv": (_int, [_vp, ctypes.c_char_p, _sz, _u32, ctypes.c_char_p, _sz]),
    "ggq_gguf_writer_tensor": (_int, [_vp, ctypes.c_char_p, _sz, _u32, ctypes.POINTER(ctypes.c_uint64), _u32, _vp,
                                      ggq_gguf_data_fn, _vp]),
    "ggq_gguf_writer_written_bytes": (ctypes.c_uint64, [_vp]),
    "ggq_gguf_writer_alignment_value": (ctypes.c_uint64, [_vp]),
    "ggq_gguf_writer_finish": (_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "ggq_gguf_writer_destroy": (None, [_vp]),
    "ggq_gguf_plan_shards": (_int, [_u32, ctypes.POINTER(ggq_gguf_kv), _sz, ctypes.POINTER(ggq_gguf_tensor), _sz,
                                    ctypes.c_uint64, ctypes.c_uint64, _int, ctypes.POINTER(_u32),
                                    ctypes.POINTER(_sz)]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C gguf_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "gguf_amd has no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args
