"""CPU oracle for the GGUF dequantize path -- test infrastructure only.

Loaded by tests/, by the smoke check and by the benchmark's cpu_baseline leg.
The product package (gguf_amd) never imports this.
"""
from .oracle import *  # noqa: F401,F403
