"""gguf_amd -- GGUF block dequantization on AMD MI355X (gfx950).

Hand-written HIP kernels behind a C ABI (include/ggq.h, built as
gguf_amd/libggq.so), with a host-side mirror of the reference crate's
`DataBlock` / `QuantExt::dequantize_slice` interface in gguf_amd.quants.
"""
from .quants import *  # noqa: F401,F403
from .quants import (GGmlType, QuantizeError, Indivisible, LengthMismatch, Unsupported,  # noqa: F401
                     DeviceError, Misaligned, dequantize_slice, quantize_slice, selftest_fast_division, Batch, QuantBatch, Context)
