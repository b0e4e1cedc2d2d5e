"""End-to-end rate of the convert pipeline (map input, decode on the GPU, D2H, write the output file):
a synthetic llama-shaped Q4_K file cast to F32.  File I/O inclusive, so far below the kernel's rate;
reported beside the kernel numbers, never the benchmark's `value`.
Run on the GPU box: python tools/measure_convert.py [out_dir ...]   (default /dev/shm and /tmp)
"""
import json
import os
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch  # noqa: F401  (first, so the library binds to torch's HIP runtime)
    from gguf_amd import convert as C
    from gguf_amd import gguf as G
    from gguf_amd import synth
    from gguf_amd.quants import GGmlType as T
    dirs = sys.argv[1:] or ["/dev/shm", "/tmp"]
    n_layer, c, r = 8, 4096, 4096
    chunk = synth.random_blocks(T.Q4K, c * 64 // 256, 1)
    tensors = []
    for i in range(n_layer):
        for name in ("attn_q", "attn_k", "attn_v", "attn_output"):
            data = np.tile(chunk, r // 64)
            tensors.append((f"blk.{i}.{name}.weight", int(T.Q4K), [c, r], data))
    elems = n_layer * 4 * c * r
    out = {"elements": elems, "packed_bytes": int(sum(t[3].nbytes for t in tensors))}
    for d in dirs:
        work = os.path.join(d, "ggq_convert_probe")
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
        src = os.path.join(work, "in.gguf")
        w = G.GGufFileWriter(src, len(tensors), 1)
        w.write_meta("general.architecture", "llama")
        for name, ty, shape, data in tensors:
            w.write_tensor(name, ty, shape, data)
        w.finish()
        for label, kw in (("one_file", {}), ("four_shards", {"max_tensor_count": n_layer})):
            t0 = time.perf_counter()
            infos = C.convert([src], os.path.join(work, label), "m", cast="linear:f32", **kw)
            dt = time.perf_counter() - t0
            nbytes = sum(i[3] for i in infos)
            out[f"{d}:{label}"] = {"seconds": round(dt, 3), "files": len(infos), "written_GBps": round(nbytes / dt / 1e9, 2),
                                   "melem_per_s": round(elems / dt / 1e6, 1)}
            print(d, label, out[f"{d}:{label}"], flush=True)
        # plain copy of the same number of bytes through the same writer: the I/O floor
        buf = np.zeros(c * r * 4, dtype=np.uint8)
        t0 = time.perf_counter()
        w = G.GGufFileWriter(os.path.join(work, "floor.gguf"), len(tensors), 0)
        for name, ty, shape, data in tensors:
            w.write_tensor(name, 0, shape, buf)
        nbytes = w.finish()
        dt = time.perf_counter() - t0
        out[f"{d}:writer_only"] = {"seconds": round(dt, 3), "written_GBps": round(nbytes / dt / 1e9, 2)}
        print(d, "writer_only", out[f"{d}:writer_only"], flush=True)
        shutil.rmtree(work, ignore_errors=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/convert_rates.json", "w"), indent=1)


if __name__ == "__main__":
    main()
