"""The GGUF reader (include/ggq_gguf.h) against the reference's own parser tests
(ggus/src/file.rs:148-444, hand-built byte buffers) and against the tensor table of the
reference's test asset (tests/golden/tinyllama_table.json)."""
import json
import os
import struct

import numpy as np
import pytest

from gguf_amd import gguf as G

HERE = os.path.dirname(os.path.abspath(__file__))


def s(x):
    b = x.encode()
    return struct.pack("<Q", len(b)) + b


def pad(n, a):
    return (-n) % a


def minimal():
    """create_minimal_gguf_data (file.rs:155-226)."""
    d = bytearray(b"GGUF" + struct.pack("<IQQ", 3, 2, 3))
    d += s("general.architecture") + struct.pack("<I", 8) + s("llama\0")
    d += s("general.alignment") + struct.pack("<II", 4, 32)
    d += s("llm.context_length") + struct.pack("<II", 4, 4096)
    d += s("tensor1") + struct.pack("<IQQIQ", 2, 3, 4, 0, 0)
    d += s("tensor2") + struct.pack("<IQIQ", 1, 5, 1, 48)
    d += b"\0" * pad(len(d), 32)
    d += np.arange(12, dtype=np.float32).tobytes()
    d += b"\0" * 10
    return bytes(d)


def test_valid_gguf_parsing():                       # file.rs:312-351
    g = G.GGuf.from_bytes(minimal())
    assert g.version == 3 and len(g.tensors) == 2 and len(g.keys()) == 3 and g.alignment == 32
    assert g.meta_raw("general.architecture")[0] == 8
    assert g.meta("llm.context_length") == 4096
    t1, t2 = g.tensor("tensor1"), g.tensor("tensor2")
    assert (t1.ty, t1.shape, t1.offset) == (0, (3, 4), 0)
    assert (t2.ty, t2.shape, t2.offset) == (1, (5,), 48)
    assert t1.nbytes == 48 and t2.nbytes == 10
    assert list(t1.bytes().view(np.float32)) == list(range(12))
    assert [t.name for t in g.tensors] == ["tensor1", "tensor2"]        # file order kept
    assert g.keys() == ["general.architecture", "general.alignment", "llm.context_length"]


def _err(buf):
    with pytest.raises(G.GGufError) as e:
        G.GGuf.from_bytes(buf)
    return e.value


def test_invalid_magic():                            # file.rs:353-358
    assert _err(b"XXXX" + struct.pack("<IQQ", 3, 0, 0)).kind == "MagicMismatch"


def test_invalid_version():                          # file.rs:360-365
    assert _err(b"GGUF" + struct.pack("<IQQ", 99, 0, 0)).kind == "VersionNotSupport"


def test_other_endian():                             # header.rs:50-57
    assert _err(b"GGUF" + struct.pack(">IQQ", 3, 0, 0)).kind == "EndianNotSupport"


def test_duplicate_meta_key():                       # file.rs:367-375
    d = b"GGUF" + struct.pack("<IQQ", 3, 0, 2) + (s("duplicate.key") + struct.pack("<II", 4, 1)) * 2
    e = _err(d)
    assert e.kind == "DuplicateMetaKey" and e.name == "duplicate.key"


def test_duplicate_tensor_name():                    # file.rs:377-385
    d = b"GGUF" + struct.pack("<IQQ", 3, 2, 0) + (s("duplicate_tensor") + struct.pack("<IQIQ", 1, 1, 0, 0)) * 2
    e = _err(d)
    assert e.kind == "DuplicateTensorName" and e.name == "duplicate_tensor"


def test_invalid_alignment_type():                   # file.rs:387-395
    d = b"GGUF" + struct.pack("<IQQ", 3, 0, 1) + s("general.alignment") + struct.pack("<I", 8) + s("not_a_number\0")
    assert _err(d).kind == "AlignmentTypeMismatch"


def test_reading_truncated_data():                   # file.rs:426-435
    d = minimal()
    assert _err(d[: len(d) // 2]).kind == "Eos"
    for cut in (0, 3, 10, 23, 24, 40, 100):
        assert _err(d[:cut]).kind in ("Eos", "MagicMismatch")


def test_extra_data_handling():                      # file.rs:437-443
    g = G.GGuf.from_bytes(minimal() + b"\xAA" * 100)
    assert len(g.tensors) == 2 and g.data_len == 58 and g.data_avail == 158
    assert g.tensor("tensor2").nbytes == 10


def test_bad_bool_and_utf8():                        # read.rs:57-73
    d = b"GGUF" + struct.pack("<IQQ", 3, 0, 1) + s("flag") + struct.pack("<IB", 7, 2)
    assert _err(d).kind == "Bool"
    d = b"GGUF" + struct.pack("<IQQ", 3, 0, 1) + struct.pack("<Q", 2) + b"\xff\xfe" + struct.pack("<II", 4, 1)
    assert _err(d).kind == "Utf8"


def test_value_kinds_round_trip():
    meta = {"a.u32": 7, "a.i32": -5, "a.u64": 2 ** 40, "a.f32": 0.5, "a.bool": True, "a.str": "héllo",
            "a.arr": [1, 2, 3], "a.strs": ["x", "yy", ""], "a.u8": (0, 200), "a.f64": (12, 1.25)}
    g = G.GGuf.from_bytes(G.build_gguf(meta, []))
    for k, v in meta.items():
        want = v[1] if isinstance(v, tuple) else v
        assert g.meta(k) == want, k
    assert g.meta("missing", 42) == 42 and g.alignment == 32


def test_alignment_u64_and_custom():
    x = np.arange(64, dtype=np.uint8)
    g = G.GGuf.from_bytes(G.build_gguf({"general.alignment": (10, 64)}, [("t", 24, (64,), x), ("u", 24, (64,), x)]))
    assert g.alignment == 64 and g.data_offset % 64 == 0
    assert [t.offset for t in g.tensors] == [0, 64]
    assert np.array_equal(g.tensor("u").bytes(), x)


def test_quantized_tensor_sizes():
    """elements_to_bytes (tensor.rs:83-96) for block types, and the file sizes of the types the
    reference's structs get wrong."""
    from gguf_amd import _lib
    for ty, (elems, nbytes) in {2: (32, 18), 12: (256, 144), 14: (256, 210), 20: (32, 18), 23: (256, 136),
                                18: (256, 98), 16: (256, 66), 29: (256, 56), 0: (1, 4), 28: (1, 8)}.items():
        assert _lib.lib.ggq_gguf_type_elems(ty) == elems and _lib.lib.ggq_gguf_type_bytes(ty) == nbytes
    assert _lib.lib.ggq_gguf_type_bytes(4) == 0 and _lib.lib.ggq_gguf_type_bytes(31) == 0
    g = G.GGuf.from_bytes(G.build_gguf({}, [("w", 12, (512, 3), None)], with_data=False))
    assert g.tensor("w").nbytes == 512 * 3 // 256 * 144 and g.tensor("w").ptr is None


def test_tinyllama_table_metadata_only(tmp_path):
    """The reference's test asset is a metadata-only file (135 tensor infos, no data) on which
    GGuf::new panics; rebuilt here from the committed table and read back through the file API."""
    doc = json.load(open(os.path.join(HERE, "golden", "tinyllama_table.json")))
    meta = {k: v for k, v in doc["meta"].items()}
    tensors = [(t["name"], t["type"], tuple(t["shape"]), None) for t in doc["tensors"]]
    p = tmp_path / "tiny.gguf"
    p.write_bytes(G.build_gguf(meta, tensors, with_data=False))
    with G.GGuf(p) as g:
        assert len(g.tensors) == 135 and g.meta("general.architecture") == "llama"
        assert g.meta("llama.block_count") == 22 and g.alignment == 32
        for got, want in zip(g.tensors, doc["tensors"]):
            assert (got.name, got.ty, list(got.shape), got.offset) == (want["name"], want["type"], want["shape"], want["offset"])
            assert got.ptr is None
        assert g.data_avail == 0 and g.data_len == 2177212416 + 5632 * 2048 * 2


def test_open_errors(tmp_path):
    with pytest.raises(G.GGufError) as e:
        G.GGuf(tmp_path / "nope.gguf")
    assert e.value.kind == "Io"
    (tmp_path / "bad.gguf").write_bytes(b"XXXX" + b"\0" * 40)
    with pytest.raises(G.GGufError) as e:
        G.GGuf(tmp_path / "bad.gguf")
    assert e.value.kind == "MagicMismatch"


# ---- hostile input: the parser must answer with an error code, never crash ----------------------

def _valid_file():
    rng = np.random.default_rng(9)
    tensors = [("token_embd.weight", 1, [64, 8], rng.integers(0, 256, 64 * 8 * 2, dtype=np.uint8)),
               ("blk.0.attn_q.weight", 12, [256, 4], rng.integers(0, 256, 4 * 144, dtype=np.uint8)),
               ("blk.0.norm.weight", 0, [64], rng.integers(0, 256, 256, dtype=np.uint8))]
    meta = {"general.architecture": "llama", "llama.block_count": 1, "tokens": ["a", "bc", "def"],
            "scores": [0.5, 1.5], "general.alignment": 64, "flag": True, "big": (10, 2 ** 40)}
    return np.frombuffer(G.build_gguf(meta, tensors), dtype=np.uint8).copy()


def test_truncations_never_crash():
    buf = _valid_file()
    with G.GGuf.from_bytes(buf) as g:
        n_ok = len(g.tensors)
    assert n_ok == 3
    for cut in range(0, buf.size):
        try:
            with G.GGuf.from_bytes(buf[:cut].copy()) as g:
                assert len(g.tensors) == 3          # only the data region may be short
        except G.GGufError as e:
            assert e.kind in ("Eos", "MagicMismatch"), (cut, e.kind)


def test_byte_flips_never_crash():
    from hypothesis import given, settings, strategies as st
    buf = _valid_file()
    header_len = int(np.flatnonzero(buf[:-4] != 0).max()) if False else 600

    @settings(max_examples=400, deadline=None, derandomize=True)
    @given(st.lists(st.tuples(st.integers(0, header_len - 1), st.integers(0, 255)), min_size=1, max_size=6))
    def run(flips):
        b = buf.copy()
        for pos, val in flips:
            b[pos] = val
        try:
            with G.GGuf.from_bytes(b) as g:
                for t in g.tensors:
                    if t.ptr:
                        t.bytes()[:1]              # what it hands out must be readable
                g.keys()
        except G.GGufError:
            pass
        except UnicodeDecodeError:
            pass

    run()


def _hdr(n_tensors, n_kv):
    return b"GGUF" + struct.pack("<IQQ", 3, n_tensors, n_kv)


def _s(b):
    return struct.pack("<Q", len(b)) + b


@pytest.mark.parametrize("blob", [
    _hdr(2 ** 60, 0),                                                     # absurd tensor count, nothing follows
    _hdr(0, 2 ** 63),                                                     # absurd key/value count
    _hdr(0, 1) + struct.pack("<Q", 2 ** 63) + b"k",                       # key length beyond the buffer
    _hdr(0, 1) + _s(b"k") + struct.pack("<I", 8) + struct.pack("<Q", 2 ** 64 - 1),          # string value length
    _hdr(0, 1) + _s(b"k") + struct.pack("<IIQ", 9, 4, 2 ** 62),           # array of 2^62 u32
    _hdr(0, 1) + _s(b"k") + struct.pack("<IIQ", 9, 9, 1) * 40,            # arrays nested 40 deep, then nothing
    _hdr(1, 0) + _s(b"t") + struct.pack("<I", 2 ** 32 - 1),               # 2^32-1 dimensions
    _hdr(1, 0) + _s(b"t") + struct.pack("<IQQIQ", 2, 2 ** 40, 2 ** 40, 0, 0),               # shape product overflows
    _hdr(1, 0) + _s(b"t") + struct.pack("<IQIQ", 1, 32, 2, 2 ** 64 - 8),  # offset near 2^64
])
def test_hostile_headers(blob):
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    try:
        with G.GGuf.from_bytes(buf) as g:
            for t in g.tensors:
                assert t.ptr is None        # nothing of these can be backed by the buffer
    except G.GGufError as e:
        assert e.kind in ("Eos", "ValueType", "Invalid", "TensorRange", "Indivisible")


def _with_data(tensor_meta, data_bytes=4096):
    """One tensor record followed by padding to 32 and a NON-EMPTY data region."""
    head = _hdr(1, 0) + tensor_meta
    head += b"\0" * ((32 - len(head) % 32) % 32)
    return np.frombuffer(head + bytes(range(256)) * (data_bytes // 256), dtype=np.uint8).copy()


@pytest.mark.parametrize("meta,kind", [
    # offset + nbytes wraps to a small number: F16 [32] = 64 bytes at offset 2^64 - 63
    (_s(b"t") + struct.pack("<IQIQ", 1, 32, 1, 2 ** 64 - 63), "TensorRange"),
    (_s(b"t") + struct.pack("<IQIQ", 1, 32, 1, 2 ** 64 - 64), "TensorRange"),
    # shape product wraps: 2^33 x 2^33 F32 -> 2^66 elements
    (_s(b"t") + struct.pack("<IQQIQ", 2, 2 ** 33, 2 ** 33, 0, 0), "TensorRange"),
    # product fits, bytes wrap: 2^62 F32 elements -> 2^64 bytes
    (_s(b"t") + struct.pack("<IQQIQ", 2, 2 ** 31, 2 ** 31, 0, 0), "TensorRange"),
    # three dimensions whose running product wraps back to a small value: 2^32 * 2^32 * 1
    (_s(b"t") + struct.pack("<IQQQIQ", 3, 256, 2 ** 32, 2 ** 32, 12, 0), "TensorRange"),
    # five dimensions
    (_s(b"t") + struct.pack("<IQQQQQIQ", 5, 2, 2, 2, 2, 2, 0, 0), "TensorRange"),
    # shape[0] not a whole number of blocks (Q4_K wants multiples of 256); unknown type id; scalar of a block type
    (_s(b"t") + struct.pack("<IQQIQ", 2, 100, 4, 12, 0), "Indivisible"),
    (_s(b"t") + struct.pack("<IQIQ", 1, 32, 77, 0), "Indivisible"),
    (_s(b"t") + struct.pack("<IIQ", 0, 12, 0), "Indivisible"),
])
def test_hostile_tensor_geometry_with_data_present(meta, kind):
    """With a data region present, geometry that wraps 64 bits must not turn into a pointer: the reference's
    checked arithmetic panics here, the parser answers with an error."""
    with pytest.raises(G.GGufError) as e:
        G.GGuf.from_bytes(_with_data(meta))
    assert e.value.kind == kind and e.value.name == "t"


def test_offset_beyond_data_is_not_backed():
    """In range arithmetically but beyond the bytes present: parsed, ptr is None (metadata-only files)."""
    buf = _with_data(_s(b"t") + struct.pack("<IQIQ", 1, 32, 1, 4096 - 63))      # 64 bytes, 1 byte short
    with G.GGuf.from_bytes(buf) as g:
        assert g.tensors[0].ptr is None
    buf = _with_data(_s(b"t") + struct.pack("<IQIQ", 1, 32, 1, 4096 - 64))      # exactly fits
    with G.GGuf.from_bytes(buf) as g:
        assert g.tensors[0].ptr is not None and bytes(g.tensors[0].bytes()[-1:]) == b"\xff"


def test_extreme_tensor_records_never_yield_bad_pointers():
    """Tensor records drawn from extreme dimension / offset / type values, with a data region present: the parse
    either fails with a reader error or hands out pointers whose whole range lies inside the data region."""
    from hypothesis import given, settings, strategies as st
    dims = st.sampled_from([0, 1, 31, 32, 256, 4096, 2 ** 31, 2 ** 32, 2 ** 33, 2 ** 62, 2 ** 63, 2 ** 64 - 1])
    offs = st.sampled_from([0, 32, 64, 4096 - 64, 4096, 2 ** 32, 2 ** 63, 2 ** 64 - 4096, 2 ** 64 - 63, 2 ** 64 - 1])
    tys = st.sampled_from([0, 1, 2, 8, 12, 14, 20, 24, 28, 30, 31, 77, 2 ** 32 - 1])

    @settings(max_examples=600, deadline=None, derandomize=True)
    @given(st.lists(st.tuples(st.lists(dims, min_size=0, max_size=5), tys, offs), min_size=1, max_size=3))
    def run(records):
        head = _hdr(len(records), 0)
        for i, (shape, ty, off) in enumerate(records):
            head += _s(b"t%d" % i) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", d) for d in shape)
            head += struct.pack("<IQ", ty, off)
        head += b"\0" * ((32 - len(head) % 32) % 32)
        buf = np.frombuffer(head + bytes(4096), dtype=np.uint8).copy()
        try:
            with G.GGuf.from_bytes(buf) as g:
                assert g.data_avail == 4096
                for t in g.tensors:
                    if t.ptr:
                        assert t.nbytes > 0 and t.offset + t.nbytes <= 4096
                        b = t.bytes()
                        assert b[0] == 0 and b[-1] == 0          # readable end to end
                        assert t.ptr == buf.ctypes.data + g.data_offset + t.offset
        except G.GGufError as e:
            assert e.kind in ("Indivisible", "TensorRange", "Eos")

    run()
