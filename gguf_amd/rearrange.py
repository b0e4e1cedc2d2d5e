"""Host-side mirror of the reference's block-granular tensor operators, on device tensors:
merge_qkv / merge_gate_up (concat along rows), split_qkv / split_gate_up, permute_qk
(xtask/src/utils/operator/merge.rs:224-268, permute_qk.rs:46-69), and to-llama's scale_tensor
(to_llama.rs:99-125).  Tensors are uint8 CUDA
tensors holding packed rows; shapes are (row_elems, rows) with row_elems contiguous, as in GGUF.
"""
import ctypes

from ._lib import lib
from .quants import check


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def row_bytes(ty, row_elems):
    be, bb = lib.ggq_gguf_type_elems(int(ty)), lib.ggq_gguf_type_bytes(int(ty))
    if be == 0:
        from .quants import Unsupported
        raise Unsupported()
    if row_elems % be:
        from .quants import Indivisible
        raise Indivisible()
    return row_elems // be * bb


def concat_rows(ty, row_elems, parts):
    """concat(1, parts): parts = [(tensor, rows)]; returns the merged uint8 tensor."""
    import torch
    rb = row_bytes(ty, row_elems)
    total = sum(r for _, r in parts)
    out = torch.empty(total * rb, dtype=torch.uint8, device=parts[0][0].device)
    ptrs = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t, _ in parts])
    rows = (ctypes.c_uint64 * len(parts))(*[r for _, r in parts])
    check(lib.ggq_concat_rows_device(int(ty), row_elems, out.data_ptr(), ptrs, rows, len(parts), _stream()))
    return out


def split_rows(ty, row_elems, src, part_rows):
    """split(1, tensor, part_rows): returns one uint8 tensor per part."""
    import torch
    rb = row_bytes(ty, row_elems)
    outs = [torch.empty(r * rb, dtype=torch.uint8, device=src.device) for r in part_rows]
    ptrs = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
    rows = (ctypes.c_uint64 * len(outs))(*part_rows)
    check(lib.ggq_split_rows_device(int(ty), row_elems, src.data_ptr(), ptrs, rows, len(outs), _stream()))
    return outs


def merge_qkv(ty, row_elems, q, k, v, q_rows, kv_rows):
    return concat_rows(ty, row_elems, [(q, q_rows), (k, kv_rows), (v, kv_rows)])


def split_qkv(ty, row_elems, qkv, rows, nh, nkvh):
    """split_qkv (merge.rs:252-258): head size dh = rows / (nh + 2 nkvh)."""
    dh = rows // (nh + 2 * nkvh)
    return split_rows(ty, row_elems, qkv, [nh * dh, nkvh * dh, nkvh * dh])


def permute_qk(ty, row_elems, rows, n_head, src):
    import torch
    out = torch.empty_like(src)
    check(lib.ggq_permute_qk_device(int(ty), row_elems, rows, n_head, out.data_ptr(), src.data_ptr(), _stream()))
    return out


_SCALABLE = {0: 4, 1: 2, 30: 2, 28: 8}       # F32, F16, BF16, F64: bytes per element


def scale(ty, src, factor, out=None):
    """scale_tensor (to_llama.rs:99-125): every element of a float tensor (uint8 device tensor of F32,
    F16, BF16 or F64 data) times `factor`, computed in the tensor's type.  `out` may be `src`."""
    import torch
    eb = _SCALABLE.get(int(ty))
    if eb is None:
        from .quants import Unsupported
        raise Unsupported(f"unsupported tensor type: {ty!r}")
    if src.numel() % eb:
        from .quants import Indivisible
        raise Indivisible()
    if out is None:
        out = torch.empty_like(src)
    check(lib.ggq_scale_device(int(ty), out.data_ptr(), src.data_ptr(), src.numel() // eb, float(factor), _stream()))
    return out
