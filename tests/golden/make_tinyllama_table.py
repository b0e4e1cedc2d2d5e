"""Extracts the tensor table (names, shapes, types, offsets) of the metadata-only
GGUF file that the reference ships as its test asset, into tinyllama_table.json.
Run here, where /root/reference exists; the JSON is the committed fixture (data
only).  GGUF v3 layout: magic, version, n_tensors, n_kv, kv pairs, tensor infos.
"""
import json
import os
import struct
import sys

SRC = "/root/reference/test-files/TinyLlama-1.1B-Chat-v1.0-F16.gguf"
HERE = os.path.dirname(os.path.abspath(__file__))

SCALAR = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}


def main():
    buf = open(SRC, "rb").read()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v[0]

    def rstr():
        nonlocal pos
        n = rd("<Q")
        s = buf[pos:pos + n].decode("utf-8", "replace")
        pos += n
        return s

    def rval(t):
        if t in SCALAR:
            return rd(SCALAR[t])
        if t == 8:
            return rstr()
        if t == 9:
            et = rd("<I")
            n = rd("<Q")
            vals = [rval(et) for _ in range(n)]
            return {"array_len": n, "head": vals[:4]}
        raise ValueError(t)

    assert buf[:4] == b"GGUF"
    pos = 4
    version = rd("<I")
    n_tensors = rd("<Q")
    n_kv = rd("<Q")
    kvs = {}
    for _ in range(n_kv):
        k = rstr()
        t = rd("<I")
        kvs[k] = rval(t)
    tensors = []
    for _ in range(n_tensors):
        name = rstr()
        nd = rd("<I")
        shape = [rd("<Q") for _ in range(nd)]
        ty = rd("<I")
        off = rd("<Q")
        tensors.append({"name": name, "shape": shape, "type": ty, "offset": off})
    doc = {"source": "test-files/TinyLlama-1.1B-Chat-v1.0-F16.gguf", "file_bytes": len(buf),
           "version": version, "n_kv": n_kv, "header_end": pos,
           "meta": {k: v for k, v in kvs.items() if not isinstance(v, dict)},
           "tensors": tensors}
    with open(os.path.join(HERE, "tinyllama_table.json"), "w") as f:
        json.dump(doc, f, indent=0)
    print(version, n_tensors, n_kv, pos, len(buf))


if __name__ == "__main__":
    sys.exit(main())
