"""Three launches each of the Q4_0, Q4_1, Q5_1, Q8_0 and Q8_1 quantizers on 256 Mi f32 elements (f16 elements with
`--f16`), for counter collection under rocprofv3 (tools/run_quant_traffic.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gguf_amd as g  # noqa: E402
from gguf_amd.quants import GGmlType  # noqa: E402

n = 256 << 20
x = torch.randn(n, device="cuda")
if "--f16" in sys.argv:
    x = x.half()
for ty in (GGmlType.Q4_0, GGmlType.Q4_1, GGmlType.Q5_1, GGmlType.Q8_0, GGmlType.Q8_1):
    dst = torch.empty(n // 32 * g.block_bytes(ty), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        g.quantize_slice(ty, dst, x)
    torch.cuda.synchronize()
print("done")
