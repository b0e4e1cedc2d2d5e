"""GGUF writer, byte-counting simulator and shard planner (C++ behind include/ggq_gguf.h) against
the reference's own expectations (ggus/src/write/simulator.rs:110-118, file_writer.rs:240-291),
an independent Python layout of the same format, and a Python statement of the planning loop of
xtask/src/utils/write.rs:21-51."""
import random
import struct

import numpy as np
import pytest

from gguf_amd import gguf as G

F32, F16, Q4_0, Q8_0, Q4_K = 0, 1, 2, 8, 12


def test_simulator_known_answer():
    # simulator.rs:113-118: one string key/value, alignment 16, one 2x2 F32 tensor -> 160 bytes
    sim = G.GGufFileWriter(None)
    sim.write_meta_kv("test_key", 8, b"test_value")
    sim.write_alignment(16)
    sim.write_tensor("test_tensor", F32, [2, 2])
    assert sim.written_bytes() == 160
    assert sim.alignment == 16
    assert G.GGufFileWriter(None).alignment == 32                 # simulator.rs:121-122
    assert G.GGufFileWriter(None, alignment=64).alignment == 64   # :125-126


def test_alignment_key_value_rules():
    w = G.GGufFileWriter(None)
    w.write_meta_kv("test.key", 4, bytes([1, 0, 0, 0]))
    w.write_meta_kv("general.alignment", 4, bytes([64, 0, 0, 0]))   # file_writer.rs:200-209
    assert w.alignment == 64
    with pytest.raises(G.GGufError) as e:                           # :212-221 (a panic there)
        w.write_meta_kv("general.alignment", 8, b"test\0")
    assert e.value.kind == "AlignmentTypeMismatch"


def test_offsets_and_round_trip(tmp_path):
    # file_writer.rs:262-291: the second tensor starts at 24 + pad(24, 32)
    rng = np.random.default_rng(0)
    d1 = rng.integers(0, 256, 24, dtype=np.uint8)
    d2 = rng.integers(0, 256, 32, dtype=np.uint8)
    d3 = rng.integers(0, 256, 18 * 4, dtype=np.uint8)
    p = tmp_path / "a.gguf"
    w = G.GGufFileWriter(p, n_tensors=3, n_kv=2)
    w.write_meta("general.architecture", "llama")
    w.write_meta("llm.context_length", 4096)
    w.write_tensor("tensor1", F32, [2, 3], d1)
    w.write_tensor("tensor2", F16, [4, 4], d2.tobytes())
    w.write_tensor("tensor3", Q4_0, [64, 2], lambda: d3)
    n = w.finish()
    assert n == p.stat().st_size
    with G.GGuf(p) as g:
        assert [t.offset for t in g.tensors] == [0, 32, 64]
        assert g.meta("general.architecture") == "llama" and g.meta("llm.context_length") == 4096
        for t, d in zip(g.tensors, (d1, d2, d3)):
            assert np.array_equal(t.bytes(), d)
        assert g.data_offset % 32 == 0 and n == g.data_offset + 64 + 72


@pytest.mark.parametrize("alignment", [None, 16, 64, 4096])
def test_bytes_identical_to_python_layout(tmp_path, alignment):
    rng = np.random.default_rng(3)
    tensors = []
    for i, (ty, shape) in enumerate([(Q4_K, [512, 3]), (F32, [7]), (Q8_0, [96, 5, 2]), (F16, [33, 3]), (Q4_0, [32])]):
        eb, ee = G.lib.ggq_gguf_type_bytes(ty), G.lib.ggq_gguf_type_elems(ty)
        n = int(np.prod(shape)) // ee * eb
        tensors.append((f"blk.{i}.w", ty, shape, rng.integers(0, 256, n, dtype=np.uint8)))
    meta = {"general.architecture": "llama", "a.float": 1.5, "a.list": [1, 2, 3], "a.bool": True, "s": "héllo"}
    if alignment is not None:
        meta = {"general.alignment": alignment, **meta}
    want = G.build_gguf(meta, tensors)
    p = tmp_path / "b.gguf"
    w = G.GGufFileWriter(p, len(tensors), len(meta))
    for k, v in meta.items():
        w.write_meta(k, v)
    for name, ty, shape, data in tensors:
        w.write_tensor(name, ty, shape, data)
    assert w.written_bytes() == len(want)          # the simulator's forecast
    assert w.finish() == len(want)
    assert p.read_bytes() == want


def test_metadata_only_file(tmp_path):
    p = tmp_path / "c.gguf"
    w = G.GGufFileWriter(p, 1, 0, write_data=False)
    w.write_tensor("t", Q8_0, [64, 4])
    n = w.finish()
    assert n == p.stat().st_size
    with G.GGuf(p) as g:
        assert g.tensors[0].nbytes == 34 * 8 and g.tensors[0].ptr is None


def test_shape_must_be_whole_blocks():
    w = G.GGufFileWriter(None)
    with pytest.raises(G.GGufError) as e:
        w.write_tensor("t", Q4_K, [100, 2])
    assert e.value.kind == "Indivisible"


def test_lazy_data_of_wrong_size_fails(tmp_path):
    w = G.GGufFileWriter(tmp_path / "d.gguf", 1, 0)
    w.write_tensor("t", F32, [4], lambda: np.zeros(3, dtype=np.uint8))
    with pytest.raises(G.GGufError) as e:
        w.finish()
    assert e.value.kind == "Io"


# ---- shard planner ---------------------------------------------------------------------------

def _pad(pos, a):
    return (a - pos % a) % a


def _kv_bytes(k, v):
    return 8 + len(k.encode()) + 4 + len(v)


def _info_bytes(name, shape):
    return 8 + len(name.encode()) + 4 + 8 * len(shape) + 4 + 8


def _nbytes(ty, shape):
    eb, ee = G.lib.ggq_gguf_type_bytes(ty), G.lib.ggq_gguf_type_elems(ty)
    return int(np.prod(shape)) // ee * eb


class _Sim:
    """GGufFileSimulator::with_alignment(...).finish() (simulator.rs:29-44, 63-94)."""

    def __init__(self, alignment, kvs=()):
        self.a = alignment
        self.meta = 24 + _kv_bytes("general.alignment", b"1234") + sum(_kv_bytes(k, v) for k, _, v in kvs)
        self.lens = []

    def write_tensor(self, name, ty, shape):
        self.meta += _info_bytes(name, shape)
        self.lens.append(_nbytes(ty, shape))

    def written_bytes(self):
        total = self.meta
        for n in self.lens:
            total += _pad(total, self.a) + n
        return total


def plan_py(alignment, kvs, tensors, max_count, max_size, no_tensor_first):
    sim = _Sim(alignment, kvs)
    shards = [[]]
    for i, (name, ty, shape) in enumerate(tensors):
        if len(shards) == 1 and no_tensor_first:
            sim = _Sim(alignment)
            sim.write_tensor(name, ty, shape)
            shards.append([i])
            continue
        sim.write_tensor(name, ty, shape)
        if len(shards[-1]) < max_count and sim.written_bytes() < max_size:
            shards[-1].append(i)
        else:
            sim = _Sim(alignment)
            sim.write_tensor(name, ty, shape)
            shards.append([i])
    out = [0] * len(tensors)
    for s, members in enumerate(shards):
        for i in members:
            out[i] = s
    return out, len(shards)


def test_plan_shards_matches_reference_loop():
    rnd = random.Random(7)
    types = [(F32, 1), (F16, 1), (Q4_0, 32), (Q8_0, 32), (Q4_K, 256)]
    for case in range(200):
        kvs = [(f"k{j}" * rnd.randint(1, 4), 8, bytes(rnd.randint(0, 40))) for j in range(rnd.randint(0, 5))]
        tensors = []
        for j in range(rnd.randint(0, 30)):
            ty, blk = rnd.choice(types)
            shape = [blk * rnd.randint(1, 8)] + [rnd.randint(1, 6) for _ in range(rnd.randint(0, 2))]
            tensors.append((f"blk.{j}.{'x' * rnd.randint(1, 9)}", ty, shape))
        a = rnd.choice([8, 32, 64, 256])
        max_count = rnd.choice([1, 2, 5, 2 ** 63])
        max_size = rnd.choice([1, 300, 1000, 5000, 2 ** 63])
        ntf = rnd.random() < 0.3
        assert G.plan_shards(a, kvs, tensors, max_count, max_size, ntf) == plan_py(a, kvs, tensors, max_count,
                                                                                    max_size, ntf), case


def test_plan_shards_edges():
    assert G.plan_shards(32, [], []) == ([], 1)
    t = [("a", F32, [8]), ("b", F32, [8])]
    assert G.plan_shards(32, [], t) == ([0, 0], 1)
    assert G.plan_shards(32, [], t, no_tensor_first=True) == ([1, 1], 2)
    assert G.plan_shards(32, [], t, max_tensor_count=1) == ([0, 1], 2)
    # a first tensor that is already over the size limit leaves the first shard empty (as the reference does)
    assert G.plan_shards(32, [], t, max_file_size=1) == ([1, 2], 3)


def test_sizes_that_overflow_are_refused():
    w = G.GGufFileWriter(None)
    with pytest.raises(G.GGufError) as e:
        w.write_tensor("t", 0, [2 ** 33, 2 ** 33])
    assert e.value.kind == "TensorRange"
    with pytest.raises(G.GGufError) as e:
        w.write_tensor("t", 12, [100, 4])
    assert e.value.kind == "Indivisible"
    w.close()
