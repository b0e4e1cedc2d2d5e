"""One-off soak against the CPU restatement used by the tests: random cases through the case bodies of
tests/test_gpu_fuzz.py (quantizers; with a third argument "decode": single-tensor decodes and launch plans), for a
given number of seconds.
Run on the GPU box: python tests/soak.py [seconds] [seed] [quantize|decode]   (not collected by pytest; 120 s = 130,000 cases)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

oracle.build()
body = F.test_quantize_matches_oracle.hypothesis.inner_test
decode = F.test_decode_matches_oracle.hypothesis.inner_test
plan = F.test_launch_plan_matches_oracle.hypothesis.inner_test
MODE = sys.argv[3] if len(sys.argv) > 3 else "quantize"
OUTS = ["f32", "f16", "bf16"]
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
n = 0
while time.time() < t_end and MODE == "decode":
    seed = int(rng.integers(0, 2 ** 31))
    try:
        if rng.random() < 0.7:
            case = (int(rng.choice(F.ALL_TYPES)), str(rng.choice(OUTS)),
                    int(rng.integers(1, 70)) if rng.random() < 0.5 else int(rng.integers(100, 3000)),
                    int(rng.integers(0, 32)), bool(rng.random() < 0.5), seed)
            decode(oracle, *case)
        else:
            case = (str(rng.choice(OUTS)),
                    [(int(rng.choice(F.ALL_TYPES)), int(rng.integers(0, 41)) if rng.random() < 0.5 else int(rng.integers(100, 2500)),
                      int(rng.integers(0, 16)), bool(rng.random() < 0.5)) for _ in range(int(rng.integers(1, 13)))], seed)
            plan(oracle, *case)
    except AssertionError as e:
        print("MISMATCH", case, e, flush=True)
        sys.exit(1)
    n += 1
    if n % 500 == 0:
        print(n, "cases", flush=True)
while time.time() < t_end and MODE == "quantize":
    ty = int(rng.choice(F.QUANT_TYPES))
    in_ty = str(rng.choice(["f32", "f16", "bf16"]))
    blocks = int(rng.integers(1, 70)) if rng.random() < 0.5 else int(rng.integers(100, 3000))
    kind = str(rng.choice(["normal", "wide", "special", "extreme"]))
    seed = int(rng.integers(0, 2 ** 31))
    try:
        body(oracle, ty, in_ty, blocks, kind, seed)
    except AssertionError as e:
        print("MISMATCH", ty, in_ty, blocks, kind, seed, e, flush=True)
        sys.exit(1)
    n += 1
    if n % 500 == 0:
        print(n, "cases", flush=True)
print("OK", n, "cases")
