"""One Q8_0 and one Q4_1 quantize launch of 256 Mi f32 elements, for counter collection under rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gguf_amd as g  # noqa: E402
from gguf_amd.quants import GGmlType  # noqa: E402

n = 256 << 20
x = torch.randn(n, device="cuda")
for ty in (GGmlType.Q8_0, GGmlType.Q4_1):
    dst = torch.empty(n // 32 * g.block_bytes(ty), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        g.quantize_slice(ty, dst, x)
    torch.cuda.synchronize()
print("done")
