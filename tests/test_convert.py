"""The convert pipeline (read -> cast -> write): host logic on the CPU, casts on the GPU against the
CPU oracle.  Rules restated from xtask/src/utils/operator/cast.rs:28-71, 218-227 and read.rs:33-62."""
import numpy as np
import pytest

from gguf_amd import convert as C
from gguf_amd import gguf as G
from gguf_amd import synth
from gguf_amd.quants import GGmlType as T


def test_parse_cast_reference_case():
    # cast.rs:218-227
    types = C.parse_cast("embd:f16 mat:q8_0, norm:f32")
    assert types == {"embd": T.F16, "mat": T.Q8_0, "norm": T.F32}
    assert C.parse_cast("a:q4k b:IQ4NL c:bf16") == {"a": T.Q4K, "b": T.IQ4NL, "c": T.BF16}


def test_target_type_rules():
    ty = {"linear": T.F32, "embd": T.F16, "norm": T.BF16, "else": T.Q8_0}
    f = C.target_type
    assert f("llama", "token_embd.weight", [8, 8], ty) == T.F16
    assert f("llama", "output.weight", [8, 8], ty) == T.F16
    assert f("llama", "blk.0.attn_norm.weight", [8], ty) == T.BF16
    assert f("llama", "output_norm.bias", [8], ty) == T.BF16
    assert f("qwen2", "blk.0.attn_q.weight", [8, 8], ty) == T.F32
    assert f("qwen2", "blk.0.attn_q.bias", [8], ty) == T.F32
    assert f("gpt2", "rope_freqs", [8], ty) == T.Q8_0
    assert f("llama", "rope_freqs", [8], {}) is None
    assert f("clip", "v.patch_embd.weight", [8, 8], ty) == T.F16
    assert f("clip", "v.blk.0.ln1.weight", [8], ty) == T.BF16
    assert f("clip", "v.blk.0.attn_q.weight", [8, 8], ty) == T.F32
    assert f("clip", "resampler.ln_q.weight", [8], ty) == T.BF16
    assert f("clip", "resampler.proj.weight", [8, 8], ty) == T.F32
    assert f("clip", "mm.0.weight", [8, 8], ty) == T.Q8_0
    with pytest.raises(Exception):
        f("mamba", "x", [1], ty)


def _model(tmp_path, name="in.gguf", linear=T.Q8_0, extra_meta=None, seed=0, alignment=None):
    """A small llama-shaped file: embedding, two layers, norms."""
    tensors = []

    def add(tname, ty, shape):
        n = int(np.prod(shape))
        if ty == T.F32:
            data = np.random.default_rng(seed + len(tensors)).standard_normal(n).astype(np.float32).view(np.uint8)
        else:
            data = synth.random_blocks(ty, n // G.lib.ggq_gguf_type_elems(int(ty)), seed + len(tensors))
        tensors.append((tname, int(ty), shape, data))

    add("token_embd.weight", T.F16, [256, 40])
    for i in range(2):
        add(f"blk.{i}.attn_norm.weight", T.F32, [256])
        add(f"blk.{i}.attn_qkv.weight", linear, [256, 96])
        add(f"blk.{i}.ffn_down.weight", linear, [512, 24])
    add("output_norm.weight", T.F32, [256])
    meta = {"general.architecture": "llama", "llama.block_count": 2}
    meta.update(extra_meta or {})
    p = tmp_path / name
    p.write_bytes(G.build_gguf(meta, tensors, alignment=alignment))
    return p, tensors


def test_copy_without_cast_and_sharding(tmp_path):
    p, tensors = _model(tmp_path)
    with C.Content([p]) as c:
        infos = c.write_files(tmp_path / "out", "m", max_tensor_count=3)
    assert [i[1] for i in infos] == [3, 3, 2]
    assert [i[2] for i in infos] == [3, 1, 1]           # alignment + 2 key/values, then alignment only
    assert infos[0][0].endswith("m-00001-of-00003.gguf")
    got = {}
    for path, n_t, n_kv, n_bytes in infos:
        with G.GGuf(path) as g:
            assert len(g.tensors) == n_t and len(g.keys()) == n_kv
            import os
            assert os.path.getsize(path) == n_bytes
            for t in g.tensors:
                got[t.name] = (t.ty, t.shape, t.bytes().copy())
    assert list(got) == [t[0] for t in tensors]
    for name, ty, shape, data in tensors:
        assert got[name][0] == ty and list(got[name][1]) == shape and np.array_equal(got[name][2], data)
    # the shards merge back into the same content (read.rs:33-62)
    with C.Content([i[0] for i in infos]) as c:
        assert list(c.tensors) == [t[0] for t in tensors]
        assert list(c.meta_kvs) == ["general.architecture", "llama.block_count"]
        assert c.alignment == 32


def test_ranks_split_the_shards(tmp_path):
    p, tensors = _model(tmp_path)
    with C.Content([p]) as c:
        a = c.write_files(tmp_path / "r", "m", max_tensor_count=3, rank=0, world=2)
        b = c.write_files(tmp_path / "r", "m", max_tensor_count=3, rank=1, world=2)
    assert [i[0][-19:] for i in a] == ["00001-of-00003.gguf", "00003-of-00003.gguf"]
    assert [i[0][-19:] for i in b] == ["00002-of-00003.gguf"]
    with C.Content([i[0] for i in a + b]) as c:          # any order: the union is the model
        assert sorted(c.tensors) == sorted(t[0] for t in tensors)


def test_merge_rejects_duplicates_and_takes_max_alignment(tmp_path):
    a, _ = _model(tmp_path, "a.gguf", alignment=64)
    b, _ = _model(tmp_path, "b.gguf")
    with pytest.raises(G.GGufError) as e:
        C.Content([a, b])
    assert e.value.kind == "DuplicateMetaKey"
    with C.Content([a]) as c:
        assert c.alignment == 64


def test_cast_marks_only_changed_tensors(tmp_path):
    p, _ = _model(tmp_path)
    with C.Content([p]) as c:
        c.cast("linear:f32 norm:f32 embd:f16")
        changed = [n for n, t in c.tensors.items() if t.data.lazy]
        assert changed == ["blk.0.attn_qkv.weight", "blk.0.ffn_down.weight", "blk.1.attn_qkv.weight",
                           "blk.1.ffn_down.weight"]
        # metadata-only output needs no device
        infos = c.write_files(tmp_path / "o", "m", write_data=False)
    with G.GGuf(infos[0][0]) as g:
        assert g.tensor("blk.0.attn_qkv.weight").ty == 0 and g.tensor("blk.0.attn_qkv.weight").ptr is None


# ---- casts on the device -----------------------------------------------------------------------

def _oracle_cast(data, from_ty, to_ty):
    from oracle import oracle as O
    floats = {int(T.F32): np.float32, int(T.F16): np.uint16, int(T.BF16): np.uint16}
    from_ty, to_ty = int(from_ty), int(to_ty)
    if to_ty in floats and (from_ty not in floats or to_ty == 0):
        return O.dequantize(from_ty, data, to_ty).view(np.uint8)
    vals = data.view(floats[from_ty]) if from_ty in floats else O.dequantize(from_ty, data, 0)
    return O.quantize(to_ty, vals, from_ty if from_ty in floats else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("linear,spec", [
    (T.Q8_0, "linear:f32"),                       # the reference's one working decode (cast.rs:132-135)
    (T.Q4K, "linear:f32 embd:f32"),
    (T.Q6K, "linear:f16 embd:bf16"),
    (T.F16, "linear:q8_0 embd:q4_0 norm:f16"),    # cast.rs:108-119, 96-107
    (T.BF16, "linear:f16 embd:q5_1"),             # cast.rs:120-131
    (T.Q5K, "linear:q4_1 embd:q8_1"),             # through f32 (cast.rs:136)
])
def test_convert_cast_matches_oracle(tmp_path, linear, spec):
    p, tensors = _model(tmp_path, linear=linear, seed=11)
    infos = C.convert([p], tmp_path / "out", "m", cast=spec, max_tensor_count=4)
    types = C.parse_cast(spec)
    got = {}
    for path, *_ in infos:
        with G.GGuf(path) as g:
            for t in g.tensors:
                got[t.name] = (t.ty, t.bytes().copy())
    assert list(got) == [t[0] for t in tensors]
    n_cast = 0
    for name, ty, shape, data in tensors:
        to = C.target_type("llama", name, shape, types)
        if to is None or to == ty:
            assert got[name][0] == ty and np.array_equal(got[name][1], data)
        else:
            n_cast += 1
            want = _oracle_cast(data, ty, to)
            assert got[name][0] == int(to)
            assert np.array_equal(got[name][1], want), name
    assert n_cast >= 4


# ---- merge-linear / split / permute-qk ----------------------------------------------------------

def _attn_model(tmp_path, ty=T.Q8_0, nh=4, nkvh=2, dh=8, c=64, seed=5):
    tensors = []

    def add(tname, tty, shape):
        n = int(np.prod(shape))
        if tty == T.F32:
            data = np.random.default_rng(seed + len(tensors)).standard_normal(n).astype(np.float32).view(np.uint8)
        else:
            data = synth.random_blocks(tty, n // G.lib.ggq_gguf_type_elems(int(tty)), seed + len(tensors))
        tensors.append((tname, int(tty), shape, data))

    add("token_embd.weight", T.F16, [c, 16])
    for i in range(2):
        add(f"blk.{i}.attn_q.weight", ty, [c, nh * dh])
        add(f"blk.{i}.attn_q.bias", T.F32, [nh * dh])
        add(f"blk.{i}.attn_k.weight", ty, [c, nkvh * dh])
        add(f"blk.{i}.attn_k.bias", T.F32, [nkvh * dh])
        add(f"blk.{i}.attn_norm.weight", T.F32, [c])
        add(f"blk.{i}.attn_v.weight", ty, [c, nkvh * dh])
        add(f"blk.{i}.attn_v.bias", T.F32, [nkvh * dh])
        add(f"blk.{i}.ffn_gate.weight", ty, [c, 96])
        add(f"blk.{i}.ffn_up.weight", ty, [c, 96])
    add("blk.9.ffn_up.weight", ty, [c, 32])             # a group that never completes
    meta = {"general.architecture": "llama", "llama.attention.head_count": nh, "llama.attention.head_count_kv": nkvh}
    p = tmp_path / "attn.gguf"
    p.write_bytes(G.build_gguf(meta, tensors))
    return p, {t[0]: t for t in tensors}


def _np_permute(data, rows, nh):
    rb = data.size // rows
    return np.ascontiguousarray(data.reshape(nh, 2, rows // nh // 2, rb).transpose(0, 2, 1, 3)).reshape(-1)


def _read_all(infos):
    got = {}
    for path, *_ in infos:
        with G.GGuf(path) as g:
            for t in g.tensors:
                got[t.name] = (t.ty, list(t.shape), t.bytes().copy())
    return got


def test_merge_linear_names_order_and_shapes(tmp_path):
    p, src = _attn_model(tmp_path)
    with C.Content([p]) as c:
        c.merge_linear(True)
        names = list(c.tensors)
        # a merged tensor takes the place of the piece that completed it (merge.rs:27-36)
        assert names[:7] == ["token_embd.weight", "blk.0.attn_norm.weight", "blk.0.attn_qkv.weight",
                             "blk.0.attn_qkv.bias", "blk.0.ffn_gate_up.weight", "blk.1.attn_norm.weight",
                             "blk.1.attn_qkv.weight"]
        assert names[-1] == "blk.9.ffn_up.weight"        # leftovers at the end (merge.rs:38-40)
        assert c.tensors["blk.0.attn_qkv.weight"].shape == [64, 64]
        assert c.tensors["blk.0.attn_qkv.bias"].shape == [64]
        assert c.tensors["blk.0.ffn_gate_up.weight"].shape == [64, 192]
        c.merge_linear(False)
        assert set(c.tensors) == set(src)
        for name, t in c.tensors.items():
            assert t.shape == src[name][2] and int(t.ty) == src[name][1]


@pytest.mark.gpu
def test_merge_split_permute_on_device(tmp_path):
    nh, nkvh, dh = 4, 2, 8
    p, src = _attn_model(tmp_path, nh=nh, nkvh=nkvh, dh=dh)
    # merge
    got = _read_all(C.convert([p], tmp_path / "m", "m", steps=[("merge-linear", True)]))
    for i in range(2):
        w = np.concatenate([src[f"blk.{i}.attn_{x}.weight"][3] for x in "qkv"])
        b = np.concatenate([src[f"blk.{i}.attn_{x}.bias"][3] for x in "qkv"])
        g = np.concatenate([src[f"blk.{i}.ffn_{x}.weight"][3] for x in ("gate", "up")])
        assert np.array_equal(got[f"blk.{i}.attn_qkv.weight"][2], w)
        assert np.array_equal(got[f"blk.{i}.attn_qkv.bias"][2], b)
        assert np.array_equal(got[f"blk.{i}.ffn_gate_up.weight"][2], g)
    # merge then split gives the input back
    back = _read_all(C.convert([p], tmp_path / "s", "s", steps=[("merge-linear", True), ("merge-linear", False)]))
    assert set(back) == set(src)
    for name, (_, ty, shape, data) in src.items():
        assert back[name][0] == ty and back[name][1] == shape and np.array_equal(back[name][2], data), name
    # permute on separate and on merged tensors, followed by a cast
    perm = _read_all(C.convert([p], tmp_path / "p", "p", steps=[("permute-qk",)]))
    both = _read_all(C.convert([p], tmp_path / "b", "b", steps=[("merge-linear", True), ("permute-qk",),
                                                               ("cast", "linear:f32")]))
    from oracle import oracle as O
    for i in range(2):
        q = _np_permute(src[f"blk.{i}.attn_q.weight"][3], nh * dh, nh)
        k = _np_permute(src[f"blk.{i}.attn_k.weight"][3], nkvh * dh, nkvh)
        qb = _np_permute(src[f"blk.{i}.attn_q.bias"][3], nh * dh, nh)
        assert np.array_equal(perm[f"blk.{i}.attn_q.weight"][2], q)
        assert np.array_equal(perm[f"blk.{i}.attn_k.weight"][2], k)
        assert np.array_equal(perm[f"blk.{i}.attn_q.bias"][2], qb)
        assert np.array_equal(perm[f"blk.{i}.attn_v.weight"][2], src[f"blk.{i}.attn_v.weight"][3])
        qkv = np.concatenate([q, k, src[f"blk.{i}.attn_v.weight"][3]])
        assert both[f"blk.{i}.attn_qkv.weight"][0] == 0
        assert np.array_equal(both[f"blk.{i}.attn_qkv.weight"][2], O.dequantize(int(T.Q8_0), qkv).view(np.uint8))


@pytest.mark.gpu
def test_concat_of_cast_parts_of_equal_size(tmp_path):
    """cast, then merge: every merged tensor is a Concat of Cast(Borrowed) parts, evaluated one part after the other.
    A part's packed upload is freed as soon as its decode is enqueued, and the next part's upload (k and v have the
    same size, so the allocator offers the very same block) runs on the upload context's streams: it must be ordered
    behind the decode that still reads the block.  Repeated, because the window is a few microseconds."""
    from oracle import oracle as O
    nh, nkvh, dh = 4, 2, 8
    p, src = _attn_model(tmp_path, nh=nh, nkvh=nkvh, dh=dh)
    for rep in range(6):
        got = _read_all(C.convert([p], tmp_path / f"cm{rep}", "cm", steps=[("cast", "linear:f32"), ("merge-linear", True)]))
        for i in range(2):
            parts = [src[f"blk.{i}.attn_{x}.weight"] for x in "qkv"]
            want = np.concatenate([O.dequantize(int(ty), data).view(np.uint8) for _, ty, _, data in parts])
            assert got[f"blk.{i}.attn_qkv.weight"][0] == 0
            assert np.array_equal(got[f"blk.{i}.attn_qkv.weight"][2], want), (rep, i)
