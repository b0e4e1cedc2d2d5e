"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/ggq.h declares, and its type table and argument checks (which run before
any device work) agree with the oracle and the reference's rules."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("ggq.h", "ggq_gguf.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ggq_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from gguf_amd import _lib
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(_lib.lib, n), n
    assert sorted(_lib.SYMBOLS) == names


def test_type_table_matches_oracle_and_fixture(oracle):
    import gguf_amd as g
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "type_table.json")))
    for t in doc["types"]:
        assert g.block_elems(t["id"]) == t["elems"] == oracle.block_elems(t["id"])
        assert g.block_bytes(t["id"]) == t["bytes"] == oracle.block_bytes(t["id"])
        assert g.type_name(t["id"]) == t["name"]
    for bad in (0, 4, 5, 16, 17, 18, 19, 21, 22, 24, 29, 31, 99):   # F32 is not a block type; IQ1/2/3, ints
        assert g.block_elems(bad) == 0 and g.block_bytes(bad) == 0 and g.type_name(bad) is None


def test_elements_to_bytes():
    """ggus/src/tensor.rs:83-96 and its tests :305-341."""
    import gguf_amd as g
    T = g.GGmlType
    assert g.elements_to_bytes(T.F32, []) == 4
    assert g.elements_to_bytes(T.F32, [3, 5]) == 60
    assert g.elements_to_bytes(T.Q4K, [4096, 4096]) == 4096 * 4096 // 256 * 144
    assert g.elements_to_bytes(T.Q8_0, [64, 2, 3]) == 6 * 2 * 34
    with pytest.raises(g.Indivisible):
        g.elements_to_bytes(T.Q4K, [100, 4])
    with pytest.raises(g.Indivisible):
        g.elements_to_bytes(T.Q4K, [])


def test_argument_checks_precede_device_work():
    """Order of lib.rs:136-141; these return before any HIP call, so they hold without a GPU."""
    import gguf_amd as g
    dst = np.zeros(96, dtype=np.float32)
    src = np.zeros(34 * 3, dtype=np.uint8)
    with pytest.raises(g.Indivisible):
        g.Q8_0.dequantize_slice(dst[:31], src[:34])
    with pytest.raises(g.LengthMismatch):
        g.Q8_0.dequantize_slice(dst[:64], src)
    with pytest.raises(g.Indivisible):
        g.Q8_0.dequantize_slice(dst[:31], src)
    with pytest.raises(g.Unsupported):
        g.dequantize_slice(g.GGmlType.IQ2XXS, dst, src)
    with pytest.raises(g.Unsupported):
        g.dequantize_slice(g.GGmlType.Q4_0_4_4, dst, src)
    g.Q8_0.dequantize_slice(dst[:0], src[:0])     # empty slices are fine


def test_no_device_is_an_error_not_a_fallback():
    import gguf_amd as g
    from gguf_amd import _lib
    if _lib.lib.ggq_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(g.DeviceError):
        g.Q8_0.dequantize_slice(np.zeros(32, dtype=np.float32), np.zeros(34, dtype=np.uint8))


def test_product_does_not_touch_the_oracle():
    """The shipped package must not import, load or link oracle/."""
    pkg = os.path.join(ROOT, "gguf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libggq.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
