"""The quantizers' short division by a constant (gguf_amd/csrc/quantize.h, div_const_fast: a multiply and two fused
steps instead of the compiler's eleven-instruction IEEE division) gives the correctly rounded quotient -- the bits of
the reference's `/` (ggml-quants/src/structs/q8_0.rs:31, q4_1.rs:35, q5_1.rs:37, q8_k.rs:37) -- for every magnitude the
kernels apply it to (2^-100 ..= 2^100).  Host restatement with fmaf (tests/native/fastdiv_check.c); every mantissa and
both signs of the binades at the ends and in the middle of the range here (the arithmetic is the same in every
binade as long as nothing leaves the normal range, which is what the end binades probe); the whole range, 201
binades, takes 32 s and was run once: 0 mismatches for 15, 31, 127 and -127.  The GPU's own sequence, including the
reciprocal that depends on the hardware's v_rcp_f32, is checked on the device (tests/test_gpu_fast_division.py)."""
import os
import shutil
import subprocess

import pytest

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "fastdiv_check.c")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    exe = str(tmp_path_factory.mktemp("fastdiv") / "fastdiv_check")
    subprocess.check_call(["gcc", "-O2", "-march=native", "-ffp-contract=off", "-o", exe, SRC, "-lm"])
    return exe


def run(exe, e0, e1):
    out = subprocess.run([exe, str(e0), str(e1)], capture_output=True, text=True, check=True, timeout=600).stdout
    return {float(c): int(n) for c, n in (line.split() for line in out.splitlines())}


@pytest.mark.parametrize("e0, e1", [(-100, -99), (-1, 1), (99, 100)])
def test_constant_division_is_correctly_rounded_in_the_fast_range(checker, e0, e1):
    assert run(checker, e0, e1) == {15.0: 0, 31.0: 0, 127.0: 0, -127.0: 0}


def test_the_check_can_fail(checker):
    """Below the range the residual underflows and the short sequence is no longer exact: the checker sees it (this
    is why the kernels take the general division for such blocks)."""
    assert sum(run(checker, -127, -126).values()) > 0
