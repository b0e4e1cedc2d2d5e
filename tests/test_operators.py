"""The convert pipeline's remaining operators: filter-meta / filter-tensor, sort, set-meta, to-llama and the
step grammar (xtask/src/utils/operator/mod.rs:58-71, name_pattern.rs, sort.rs, set_meta.rs, to_llama.rs,
xtask/src/convert.rs:38-51).  Host logic on the CPU; the scaling kernel on the GPU against the oracle."""
import math
import struct

import numpy as np
import pytest

from gguf_amd import convert as C
from gguf_amd import gguf as G
from gguf_amd.quants import GGmlType as T


# ---- name patterns (name_pattern.rs:8-58) -----------------------------------------------------------

def test_compile_patterns():
    r = C.compile_patterns("blk.*.attn_q.weight")
    assert r.search("blk.0.attn_q.weight") and r.search("blk.a.b.attn_q.weight")
    assert not r.search("blk.attn_q.weight") and not r.search("xblk.0.attn_q.weight")
    r = C.compile_patterns("general.*, tokenizer..model")            # separators are anything else
    assert r.search("general.name") and r.search("general.a.b") and not r.search("general")
    assert r.search("tokenizer.ggml.model") and not r.search("tokenizer.model") and not r.search("tokenizer.a.b.model")
    r = C.compile_patterns("*_norm.weight attn*.bias")               # a star inside a piece: word characters only
    assert r.search("output_norm.weight") and r.search("_norm.weight") and not r.search("blk.0.attn_norm.weight")
    assert r.search("attn_q.bias") and r.search("attn.bias") and not r.search("attn.q.bias")
    assert C.compile_patterns("a**b").pattern == r"^a\w*b$"


# ---- sort (sort.rs) ----------------------------------------------------------------------------------

def test_sort_tensors_order(tmp_path):
    names = ["output.weight", "blk.10.attn_q.weight", "blk.2.ffn_down.weight", "blk.2.ffn_up.weight",
             "blk.2.attn_norm.weight", "token_embd.weight", "blk.2.attn_q.bias", "blk.2.attn_q.weight",
             "blk.2.attn_qkv.weight", "output_norm.weight", "blk.2.attn_norm_2.weight", "blk.2.ffn_gate.weight",
             "blk.2.ffn_norm.weight", "blk.2.attn_output.weight", "blk.2.attn_k.weight", "rope_freqs.weight",
             "blk.2.zzz.weight", "blk.x.attn_q.weight"]
    tensors = [(n, 0, [4], np.zeros(16, dtype=np.uint8)) for n in names]
    p = tmp_path / "s.gguf"
    p.write_bytes(G.build_gguf({"general.architecture": "llama"}, tensors))
    with C.Content([p]) as c:
        c.sort_tensors()
        got = list(c.tensors)
    assert got == [
        # "blk.x." : a string segment sorts before a number (sort.rs:124-125)
        "blk.x.attn_q.weight",
        # layer 2 in the list order of sort.rs:18-47; "attn_norm_2" is found as attn_norm + "_2.weight"
        "blk.2.attn_norm.weight", "blk.2.attn_norm_2.weight", "blk.2.attn_qkv.weight", "blk.2.attn_q.weight",
        "blk.2.attn_q.bias", "blk.2.attn_k.weight", "blk.2.attn_output.weight", "blk.2.ffn_norm.weight",
        "blk.2.ffn_gate.weight", "blk.2.ffn_up.weight", "blk.2.ffn_down.weight",
        # no known middle part: the whole name is the prefix ["blk", 2, "zzz", "weight"], and "" < "zzz"
        "blk.2.zzz.weight",
        "blk.10.attn_q.weight",
        "output.weight", "output_norm.weight", "rope_freqs.weight", "token_embd.weight"]


# ---- set-meta (set_meta.rs) --------------------------------------------------------------------------

CFG = """'general.name' str "my model"
'llama.block_count' u32 22

'tokenizer.chat_template' str|
| {% for m in messages %}
|  indented
| {% endfor %}
'a.f' f32 1.5
'a.d' f64 -2.25
'a.b' bool true
'a.i' i8 -128
'a.q' u64 18446744073709551615
'empty.text' str#
'general.alignment' u32 64
"""


def test_parse_meta_cfg():
    m = C.parse_meta_cfg(CFG)
    assert list(m) == ["general.name", "llama.block_count", "tokenizer.chat_template", "a.f", "a.d", "a.b", "a.i",
                       "a.q", "empty.text", "general.alignment"]
    s = lambda t: (8, struct.pack("<Q", len(t.encode())) + t.encode())
    assert m["general.name"] == s("my model")
    assert m["tokenizer.chat_template"] == s("{% for m in messages %}\n indented\n{% endfor %}")
    assert m["llama.block_count"] == (4, struct.pack("<I", 22))
    assert m["a.f"] == (6, struct.pack("<f", 1.5)) and m["a.d"] == (12, struct.pack("<d", -2.25))
    assert m["a.b"] == (7, b"\x01") and m["a.i"] == (1, b"\x80") and m["a.q"] == (10, b"\xff" * 8)
    assert m["empty.text"] == s("")
    # a string of several lines ends at an empty line once it has begun, not before (set_meta.rs:115-142)
    m = C.parse_meta_cfg("'k' str>\n\n> one\n> two\n\n'n' u8 7")
    assert m["k"] == s("one\ntwo") and m["n"] == (0, b"\x07")
    for bad in ("'k' u8 256", "'k' u8 1_0", "'k' bool yes", "k u8 1", "'k' str no quotes", "'k' f16 1", "'k' str| x"):
        with pytest.raises(ValueError):
            C.parse_meta_cfg(bad)
    with pytest.raises(C.Unsupported):
        C.parse_meta_cfg("'k' [u8] 1")


def _small(tmp_path, meta, tensors=None, name="m.gguf"):
    tensors = tensors or [("token_embd.weight", 0, [4], np.arange(16, dtype=np.uint8))]
    p = tmp_path / name
    p.write_bytes(G.build_gguf(meta, tensors))
    return p


def test_set_meta_and_filters(tmp_path):
    p = _small(tmp_path, {"general.architecture": "llama", "general.name": "old", "llama.block_count": 2,
                          "tokenizer.ggml.model": "gpt2"},
               [("token_embd.weight", 0, [4], np.arange(16, dtype=np.uint8)),
                ("blk.0.attn_q.weight", 0, [4], np.arange(16, dtype=np.uint8)),
                ("blk.0.attn_q.bias", 0, [4], np.arange(16, dtype=np.uint8))])
    with C.Content([p]) as c:
        c.set_meta(CFG)
        # keys that exist keep their place, new ones follow in the order of the text
        assert list(c.meta_kvs)[:4] == ["general.architecture", "general.name", "llama.block_count",
                                       "tokenizer.ggml.model"]
        assert list(c.meta_kvs)[4:] == ["tokenizer.chat_template", "a.f", "a.d", "a.b", "a.i", "a.q", "empty.text"]
        assert c.meta_kvs["llama.block_count"] == (4, struct.pack("<I", 22))
        assert c.alignment == 64 and "general.alignment" not in c.meta_kvs
        with pytest.raises(ValueError):
            c.set_meta("'split.count' u16 2")
        with pytest.raises(G.GGufError):
            c.set_meta("'general.alignment' u64 64")
        c.filter_meta_key("general.* llama.*")
        assert list(c.meta_kvs) == ["general.architecture", "general.name", "llama.block_count"]
        c.filter_tensor_name("blk.*.weight")
        assert list(c.tensors) == ["blk.0.attn_q.weight"]
        infos = c.write_files(tmp_path / "o", "m")
    with G.GGuf(infos[0][0]) as g:
        assert g.alignment == 64 and g.meta("general.name") == "my model" and g.meta("llama.block_count") == 22
        assert [t.name for t in g.tensors] == ["blk.0.attn_q.weight"]


def test_parse_steps():
    assert C.parse_steps("sort -> cast:embd:f16 linear:q8_0->!merge-linear->split-linear->merge-linear->permute-qk"
                         "->to-llama->to-llama:embd_scale=12 res_scale=1->filter-meta:general.*->filter-tensor:blk.*") == [
        ("sort",), ("cast", "embd:f16 linear:q8_0"), ("merge-linear", False), ("merge-linear", False),
        ("merge-linear", True), ("permute-qk",), ("to-llama", ""), ("to-llama", "embd_scale=12 res_scale=1"),
        ("filter-meta", "general.*"), ("filter-tensor", "blk.*")]
    for bad in ("sorted", "cast", "merge", "set-meta:x"):
        with pytest.raises(C.Unsupported):
            C.parse_steps(bad)


# ---- to-llama (to_llama.rs) --------------------------------------------------------------------------

def _minicpm(tmp_path, tys=(T.F16, T.F32, T.BF16), with_scales=True, n=4096 + 24):
    rng = np.random.default_rng(17)
    tensors, vals = [], {}

    def add(name, ty, count):
        x = rng.standard_normal(count).astype(np.float32)
        if ty == T.F32:
            raw = x.view(np.uint8)
        elif ty == T.F16:
            raw = x.astype(np.float16).view(np.uint8)
        elif ty == T.BF16:
            raw = (x.view(np.uint32) >> 16).astype(np.uint16).view(np.uint8)
        else:
            raw = x.astype(np.float64).view(np.uint8)
        tensors.append((name, int(ty), [count], raw.copy()))
        vals[name] = (ty, raw.copy())

    add("token_embd.weight", tys[0], n)
    for i in range(2):
        add(f"blk.{i}.attn_norm.weight", T.F32, 64)
        add(f"blk.{i}.attn_output.weight", tys[1], n)
        add(f"blk.{i}.ffn_down.weight", tys[2], n)
        add(f"blk.{i}.ffn_up.weight", T.F16, 64)
    add("blk.0.attn_output.bias", T.F32, 64)            # only ".weight" tensors are scaled (to_llama.rs:67-68)
    meta = {"general.architecture": "minicpm", "minicpm.block_count": 2, "minicpm.context_length": 4096}
    if with_scales:
        meta.update({"minicpm.embedding_scale": 12.0, "minicpm.residual_scale": 1.4, "minicpm.logit_scale": 0.25})
    meta["tokenizer.ggml.model"] = "llama"
    return _small(tmp_path, meta, tensors, "cpm.gguf"), vals


def test_to_llama_metadata_and_marks(tmp_path):
    p, vals = _minicpm(tmp_path)
    with C.Content([p]) as c:
        c.to_llama()
        assert list(c.meta_kvs) == ["general.architecture", "llama.block_count", "llama.context_length",
                                    "tokenizer.ggml.model"]
        assert c.general_architecture() == "llama"
        scaled = {n: t.data.factor for n, t in c.tensors.items() if t.data.lazy}
        res = float(np.float32(1.4)) / math.sqrt(2.0)
        assert scaled == {"token_embd.weight": 12.0, "blk.0.attn_output.weight": res, "blk.0.ffn_down.weight": res,
                          "blk.1.attn_output.weight": res, "blk.1.ffn_down.weight": res}
        c.to_llama()                                   # a llama model stays as it is
        assert sum(t.data.lazy for t in c.tensors.values()) == 5
    # the arguments override the file's values; without either there is nothing to scale by
    with C.Content([p]) as c:
        c.to_llama("embd_scale=3 res_scale=2")
        assert c.tensors["token_embd.weight"].data.factor == 3.0
        assert c.tensors["blk.1.ffn_down.weight"].data.factor == 2.0 / math.sqrt(2.0)
    p2, _ = _minicpm(tmp_path, with_scales=False)
    with C.Content([p2]) as c:
        with pytest.raises(ValueError):
            c.to_llama("embd_scale=3")
    with C.Content([_small(tmp_path, {"general.architecture": "mamba"}, name="x.gguf")]) as c:
        with pytest.raises(C.Unsupported):
            c.to_llama()


def test_oracle_factor_conversions():
    """The factor in the tensor's type (to_llama.rs:102-105): f16 through f32 as numpy does it in two steps;
    bf16 by the `half` crate's routine, which rounds the upper 32 bits of the f64 to nearest even."""
    from oracle import oracle as O
    L = O.lib()
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.standard_normal(5000) * 10.0 ** rng.integers(-9, 5, 5000),
                        [0.0, -0.0, 1.0, 12.0, 1.4, 65504.0, 65519.99, 65520.0, 6e-8, 2.9802322387695312e-08, 1e-30,
                         float("inf"), -float("inf")]])
    with np.errstate(over="ignore"):
        want = v.astype(np.float32).astype(np.float16).view(np.uint16)
    assert np.array_equal(np.array([L.ggo_f64_to_f16(float(x)) for x in v], dtype=np.uint16), want)
    # hand cases for bf16: 1.0 -> 0x3F80; 1 + 2^-8 is a tie -> even (0x3F80); 1 + 2^-8 + 2^-20 -> up (0x3F81);
    # 1 + 2^-8 + 2^-40 lies above the tie but only in the dropped low 32 bits -> 0x3F80 (the crate's truncation);
    # 1 + 3 * 2^-8 is a tie with an odd neighbour below -> up to 0x3F82
    cases = [(1.0, 0x3F80), (1 + 2.0 ** -8, 0x3F80), (1 + 2.0 ** -8 + 2.0 ** -20, 0x3F81),
             (1 + 2.0 ** -8 + 2.0 ** -40, 0x3F80), (1 + 3 * 2.0 ** -8, 0x3F82), (-12.0, 0xC140), (0.0, 0x0000),
             (-0.0, 0x8000), (float("inf"), 0x7F80), (3.5e38, 0x7F80), (1e-320, 0x0000), (2.0 ** -127, 0x0040),
             (2.0 ** -133, 0x0001)]
    for x, bits in cases:
        assert L.ggo_f64_to_bf16(x) == bits, (x, hex(L.ggo_f64_to_bf16(x)))
    assert L.ggo_f64_to_bf16(float("nan")) & 0x7FC0 == 0x7FC0


def _np_scale(ty, raw, factor):
    """numpy statement of scale_data for the types numpy has."""
    if ty == T.F32:
        return (raw.view(np.float32) * np.float32(factor)).view(np.uint8)
    if ty == T.F64:
        return (raw.view(np.float64) * np.float64(factor)).view(np.uint8)
    if ty == T.F16:
        f = np.float32(factor).astype(np.float16).astype(np.float32)
        return (raw.view(np.float16).astype(np.float32) * f).astype(np.float16).view(np.uint8)
    raise KeyError(ty)


def test_oracle_scale_against_numpy():
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(4000) * 10.0 ** rng.integers(-6, 4, 4000), [0.0, -0.0, 65504.0, 1e-7]])
    for factor in (12.0, 1.4 / math.sqrt(40.0), -0.0, 3.0000001, 1e-4):
        for ty, raw in ((T.F32, x.astype(np.float32)), (T.F64, x.copy()), (T.F16, x.astype(np.float16))):
            with np.errstate(over="ignore"):
                want = _np_scale(ty, raw.view(np.uint8), factor)
            got = O.scale(int(ty), raw.view(np.uint16) if ty == T.F16 else raw, factor)
            assert np.array_equal(got.view(np.uint8), want), (ty, factor)


# ---- on the device -----------------------------------------------------------------------------------

def _patterns16(kind):
    """Every 16-bit pattern except signalling/any NaN (the payload a product keeps is the hardware's business)."""
    bits = np.arange(65536, dtype=np.uint32)
    if kind == "f16":
        keep = ~(((bits & 0x7C00) == 0x7C00) & ((bits & 0x3FF) != 0))
    else:
        keep = ~(((bits & 0x7F80) == 0x7F80) & ((bits & 0x7F) != 0))
    return bits[keep].astype(np.uint16)


@pytest.mark.gpu
@pytest.mark.parametrize("factor", [12.0, 1.4 / math.sqrt(40.0), -3.0000001, 1e-4, 65000.0, 0.0])
def test_scale_device_matches_oracle(factor):
    import torch
    from gguf_amd import rearrange
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    n = 1_000_003
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 5, n)
    edge = np.array([0.0, -0.0, 1e-45, -1e-45, 3e38, -3e38, np.inf, -np.inf, 1.17549435e-38])
    cases = [
        (T.F32, np.concatenate([x, edge]).astype(np.float32)),
        (T.F64, np.concatenate([x, edge, [1e-320, 1e308]])),
        (T.F16, np.concatenate([_patterns16("f16"), x.astype(np.float16).view(np.uint16)])),
        (T.BF16, np.concatenate([_patterns16("bf16"), (x.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)])),
    ]
    for ty, host in cases:
        raw = np.ascontiguousarray(host).view(np.uint8)
        want = O.scale(int(ty), host, factor).view(np.uint8)
        src = torch.from_numpy(raw.copy()).cuda()
        got = rearrange.scale(ty, src, factor)
        assert np.array_equal(got.cpu().numpy(), want), (ty, factor)
        assert np.array_equal(src.cpu().numpy(), raw)                       # the source is left alone ...
        eb = host.dtype.itemsize
        # ... unless asked to work in place; and from an address that is only element-aligned
        off = torch.empty(raw.size + 16, dtype=torch.uint8, device="cuda")
        view = off[eb:eb + raw.size]
        view.copy_(src)
        rearrange.scale(ty, view, factor, out=view)
        assert np.array_equal(view.cpu().numpy(), want), (ty, factor, "in place, unaligned")
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_scale_device_errors():
    import torch
    import gguf_amd as g
    from gguf_amd import rearrange
    x = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(g.Unsupported):
        rearrange.scale(T.Q8_0, x, 2.0)
    with pytest.raises(g.Indivisible):
        rearrange.scale(T.F32, x[:63], 2.0)
    with pytest.raises(g.Misaligned):
        rearrange.scale(T.F32, x[1:61], 2.0)
    assert rearrange.scale(T.F32, x[:0], 2.0).numel() == 0


@pytest.mark.gpu
def test_to_llama_end_to_end(tmp_path):
    """minicpm -> llama through the whole pipeline, followed by a cast of the scaled tensors."""
    from oracle import oracle as O
    p, vals = _minicpm(tmp_path, tys=(T.F16, T.F32, T.BF16))
    infos = C.convert([p], tmp_path / "o", "m", steps="to-llama->sort")
    res = float(np.float32(1.4)) / math.sqrt(2.0)
    with G.GGuf(infos[0][0]) as g:
        assert g.meta("general.architecture") == "llama" and g.meta("llama.block_count") == 2
        assert g.meta("minicpm.embedding_scale") is None and g.meta("llama.embedding_scale") is None
        assert [t.name for t in g.tensors][:2] == ["blk.0.attn_norm.weight", "blk.0.attn_output.weight"]
        for t in g.tensors:
            ty, raw = vals[t.name]
            factor = 12.0 if t.name == "token_embd.weight" else \
                res if t.name.endswith(("attn_output.weight", "ffn_down.weight")) else None
            if factor is None:
                assert np.array_equal(t.bytes(), raw), t.name
            else:
                host = raw.view(np.float32) if ty == T.F32 else raw.view(np.uint16)
                assert np.array_equal(t.bytes(), O.scale(int(ty), host, factor).view(np.uint8)), t.name
    # a cast after the scaling works on the scaled values (the lazy nodes compose)
    infos = C.convert([p], tmp_path / "o2", "m", steps="to-llama:embd_scale=2 res_scale=3->cast:else:f32 linear:f32 embd:f32")
    with G.GGuf(infos[0][0]) as g:
        t = g.tensor("token_embd.weight")
        ty, raw = vals["token_embd.weight"]
        scaled = O.scale(int(ty), raw.view(np.uint16), 2.0)
        assert t.ty == 0 and np.array_equal(t.bytes(), O.dequantize(int(T.F16), scaled.view(np.uint8)).view(np.uint8))


def test_meta_numbers_parse_like_rust():
    """set-meta values and to-llama overrides go through Rust's str::parse: f32 is rounded ONCE from the decimal text
    (a detour through f64 rounds twice and lands one step off next to a tie), unsigned types take no minus sign, and
    neither digit separators nor padding are numbers."""
    import struct
    from gguf_amd.convert import _meta_value, _parse_f32, _parse_f64
    up = np.nextafter(np.float32(1), np.float32(2))
    assert _parse_f32("1.0000000596046447753906251") == up          # just above the tie between 1 and its successor
    assert np.float32(float("1.0000000596046447753906251")) == np.float32(1)      # ... which the detour misses
    assert _parse_f32("1.00000005960464477539062") == np.float32(1)                # just below the tie
    assert _parse_f32("1.000000059604644775390625") == np.float32(1)               # the tie itself: to even
    rng = np.random.default_rng(3)
    for bits in rng.integers(0, 2 ** 32, 3000, dtype=np.uint64):
        v = np.uint32(bits).view(np.float32)
        if np.isfinite(v):
            assert _parse_f32(repr(float(v))).view(np.uint32) == np.uint32(bits)
    assert _parse_f32("3.40282357e38") == np.inf and _parse_f32("3.4028235e38") == np.finfo(np.float32).max
    assert _parse_f32("7e-46") == 0 and _parse_f32("8e-46").view(np.uint32) == 1 and str(_parse_f32("-0.0")) == "-0.0"
    assert np.isnan(_parse_f32("NaN")) and _parse_f32("-inf") == -np.inf
    assert _meta_value(6, "0.1") == (6, struct.pack("<f", np.float32(0.1)))
    assert _meta_value(0, "+7") == (0, b"\x07") and _meta_value(1, "-0") == (1, b"\x00")
    for ty, bad in ((0, "-0"), (4, "-1"), (10, "-0"), (6, "1_0"), (12, " 1.0"), (12, "1_0.5")):
        with pytest.raises(ValueError):
            _meta_value(ty, bad)
    with pytest.raises(ValueError):
        _parse_f64("1_0")
