"""The kernels beside the batch decode -- quantize, widen, narrow, the permute_qk row gather, scale, and four
single-tensor decodes that use the group start -- run a few
times each on 1 Gi elements, for `rocprofv3 --kernel-trace --stats` (profiles/summarize_misc.py prices
the averages against 8 TB/s).  Run on the GPU box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gguf_amd as g  # noqa: E402
from gguf_amd import rearrange as R  # noqa: E402
from gguf_amd.quants import GGmlType as T  # noqa: E402

import time  # noqa: E402

n = 1 << 30
reps = 60          # the summary also prices the second half alone (the first launches after a change of load run slow)
x = torch.randn(n, device="cuda") * 0.02
bytes_per_launch = {}
# the GPU reaches its steady clocks only after tens of milliseconds of load: keep it busy first
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    x.mul_(1.0)
    torch.cuda.synchronize()

for ty in (T.Q4_0, T.Q4_1, T.Q5_0, T.Q5_1, T.Q8_0, T.Q8_1):
    dst = torch.empty(n // 32 * g.block_bytes(ty), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        g.quantize_slice(ty, dst, x)
    bytes_per_launch[f"quant_one<ggq::Enc{ty.name}, ggq::InF32"] = n * 4 + dst.numel()
torch.cuda.synchronize()

h = torch.empty(n, dtype=torch.float16, device="cuda")
for _ in range(reps):
    g.quantize_slice(T.F16, h.view(torch.uint8), x)            # narrow f32 -> f16
for ty in (T.Q4_0, T.Q5_1, T.Q8_0):                            # quantizers from f16 input
    dst = torch.empty(n // 32 * g.block_bytes(ty), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        g.quantize_slice(ty, dst, h)
    bytes_per_launch[f"quant_one<ggq::Enc{ty.name}, ggq::InF16"] = n * 2 + dst.numel()
torch.cuda.synchronize()
bytes_per_launch["narrow_vec<ggq::InF32, ggq::OutF16"] = n * 6
for _ in range(reps):
    g.dequantize_slice(T.F16, x, h.view(torch.uint8))          # widen f16 -> f32
bytes_per_launch["widen_vec<false, ggq::OutF32"] = n * 6
torch.cuda.synchronize()

# permute_qk: a [4096, rows] Q4_K tensor (2304-byte rows), 1 Gi elements
rows = n // 4096
src = torch.randint(0, 255, (rows * 2304,), dtype=torch.uint8, device="cuda")
for _ in range(reps):
    out = R.permute_qk(T.Q4K, 4096, rows, rows // 128, src)
bytes_per_launch["permute_qk_rows<16"] = 2 * rows * 2304
torch.cuda.synchronize()
del src, out

# scale: 1 Gi elements of f16 and of f32, out of place
for ty, buf, key in ((T.F16, h.view(torch.uint8), "scale_vec<1"), (T.F32, x.view(torch.uint8), "scale_vec<0")):
    dst = torch.empty_like(buf)
    for _ in range(reps):
        R.scale(ty, buf, 1.4, out=dst)
    bytes_per_launch[key] = 2 * buf.numel()
    del dst
torch.cuda.synchronize()

# single-tensor decode kernels whose schedule is the group start: 1 Gi elements each
from gguf_amd import synth  # noqa: E402
del x, h
for ty, out_dt, key in ((T.Q8_0, torch.float32, "deq_one<ggq::Q8_0, ggq::OutF32"), (T.Q6K, torch.float16, "deq_one<ggq::Q6K, ggq::OutF16"),
                        (T.Q4K, torch.float16, "deq_one<ggq::Q4K, ggq::OutF16"), (T.Q5_1, torch.float16, "deq_one<ggq::Q5_1, ggq::OutF16")):
    src = synth.random_blocks_device(ty, n // g.block_elems(ty), 5, "cuda")
    dst = torch.empty(n, dtype=out_dt, device="cuda")
    for _ in range(reps):
        g.dequantize_slice(ty, dst, src)
    bytes_per_launch[key] = src.numel() + n * dst.element_size()
    del src, dst
torch.cuda.synchronize()
print(json.dumps(bytes_per_launch))
