"""Regenerates the committed golden vectors (run from the repo root:
`python tests/golden/make_golden.py`).  Inputs are seeded; expected outputs come
from the C oracle (oracle/ggq_oracle.c) after it has been cross-checked against the
naive Python restatement in tests/pyref.py.  The reference itself holds no golden
vectors (its numeric tests are unseeded tolerance round trips), so these are the
build's own known answers: data only, no reference code.
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as o  # noqa: E402
import pyref  # noqa: E402

LEGACY = [o.Q4_0, o.Q4_1, o.Q5_0, o.Q5_1, o.Q8_0, o.Q8_1]
KQ = [o.Q2K, o.Q3K, o.Q4K, o.Q5K, o.Q6K, o.Q8K, o.IQ4XS]
HEADER_F16 = {o.Q4_0: (0,), o.Q4_1: (0, 2), o.Q5_0: (0,), o.Q5_1: (0, 2), o.Q8_0: (0,), o.Q8_1: (0, 2),
              o.Q2K: (80, 82), o.Q3K: (108,), o.Q4K: (0, 2), o.Q5K: (0, 2), o.Q6K: (208,), o.Q8K: (0,),
              o.IQ4NL: (0,), o.IQ4XS: (0,)}


def seeded():
    """Round-trip vectors: N(0,1) inputs packed by the oracle's quantizer where the
    reference has one, random benign blocks otherwise.  Seed 0xF1C50000 + type id."""
    out = {}
    for ty in LEGACY + KQ + [o.IQ4NL, o.F16, o.BF16]:
        rng = np.random.default_rng(0xF1C50000 + ty)
        n = o.block_elems(ty)
        blocks = 64 if n == 32 else 16 if n == 256 else 2048
        if ty in LEGACY + [o.Q8K, o.F16, o.BF16]:
            x = rng.standard_normal(blocks * n).astype(np.float32)
            packed = o.quantize(ty, x)
            out[f"t{ty}_input"] = x
        else:
            bb = o.block_bytes(ty)
            raw = rng.integers(0, 256, size=(blocks, bb), dtype=np.uint8)
            for off in HEADER_F16[ty]:
                h = (np.abs(rng.standard_normal(blocks)) * 2.0 ** -6 + 2.0 ** -10).astype(np.float16)
                raw[:, off:off + 2] = h.view(np.uint8).reshape(blocks, 2)
            packed = raw.reshape(-1)
        out[f"t{ty}_packed"] = packed
        out[f"t{ty}_f32"] = o.dequantize(ty, packed, o.F32)
        out[f"t{ty}_f16"] = o.dequantize(ty, packed, o.F16)
        out[f"t{ty}_bf16"] = o.dequantize(ty, packed, o.BF16)
    np.savez_compressed(os.path.join(HERE, "seeded.npz"), **out)


def kats():
    """One hand-checkable block per format: d = 1.0 (0x3C00), min = 0.5 (0x3800),
    payload bytes a ramp.  Expected values from the naive Python restatement, so the
    C oracle is checked against something it did not produce."""
    out = {}
    for ty in LEGACY + KQ + [o.IQ4NL]:
        bb = o.block_bytes(ty)
        blk = np.array([(17 * i + 3) & 0xFF for i in range(bb)], dtype=np.uint8)
        offs = HEADER_F16[ty]
        blk[offs[0]:offs[0] + 2] = np.frombuffer(struct.pack("<H", 0x3C00), dtype=np.uint8)
        if len(offs) > 1:
            blk[offs[1]:offs[1] + 2] = np.frombuffer(struct.pack("<H", 0x3800), dtype=np.uint8)
        out[f"t{ty}_packed"] = blk
        out[f"t{ty}_f32"] = pyref.decode(ty, blk, bb)
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)


EDGE_SCALES = [0x0000, 0x8000, 0x0001, 0x03FF, 0x0400, 0x7BFF, 0xFBFF, 0x7C00, 0xFC00, 0x7E00, 0x7D01, 0xFE00]


def edges():
    """Blocks whose f16 header fields walk through zero, subnormal, max, inf and NaN."""
    out = {}
    for ty in LEGACY + KQ + [o.IQ4NL]:
        rng = np.random.default_rng(0xED6E0000 + ty)
        bb = o.block_bytes(ty)
        offs = HEADER_F16[ty]
        rows = []
        for a in EDGE_SCALES:
            for b in (EDGE_SCALES if len(offs) > 1 else [0]):
                blk = rng.integers(0, 256, size=bb, dtype=np.uint8)
                blk[offs[0]:offs[0] + 2] = np.frombuffer(struct.pack("<H", a), dtype=np.uint8)
                if len(offs) > 1:
                    blk[offs[1]:offs[1] + 2] = np.frombuffer(struct.pack("<H", b), dtype=np.uint8)
                rows.append(blk)
        packed = np.concatenate(rows)
        out[f"t{ty}_packed"] = packed
        out[f"t{ty}_f32"] = o.dequantize(ty, packed, o.F32)
        out[f"t{ty}_f16"] = o.dequantize(ty, packed, o.F16)
    np.savez_compressed(os.path.join(HERE, "edges.npz"), **out)


def type_table():
    rows = []
    names = {o.F16: "f16", o.BF16: "bf16", o.Q4_0: "q40", o.Q4_1: "q41", o.Q5_0: "q50", o.Q5_1: "q51",
             o.Q8_0: "q80", o.Q8_1: "q81", o.Q2K: "q2k", o.Q3K: "q3k", o.Q4K: "q4k", o.Q5K: "q5k",
             o.Q6K: "q6k", o.Q8K: "q8k", o.IQ4NL: "iq4nl", o.IQ4XS: "iq4xs"}
    for ty, name in sorted(names.items()):
        rows.append({"id": ty, "name": name, "elems": o.block_elems(ty), "bytes": o.block_bytes(ty)})
    doc = {
        "types": rows,
        "deviations_from_reference_struct_size": {
            "iq4nl": {"reference_struct_bytes": 34, "used": 18,
                      "why": "reference declares qs as [u16;16] (iq4nl.rs:10); GGUF files hold [u8;16]"},
            "iq4xs": {"reference_struct_bytes": 264, "used": 136,
                      "why": "reference declares qs as [u16;128] (iq4xs.rs:14); GGUF files hold [u8;128]"},
        },
        "kept_as_reference": {
            "q8k": {"bytes": 290, "why": "reference implements it with an f16 delta (q8_k.rs:8-15); ggml's own is 292 with f32"},
        },
    }
    with open(os.path.join(HERE, "type_table.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    o.build()
    seeded()
    kats()
    edges()
    type_table()
    print("golden vectors written to", HERE)
