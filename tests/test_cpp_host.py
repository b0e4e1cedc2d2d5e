"""The C++ host mirror (gguf_amd/host/ggml_quants.hpp) run through its test program,
which restates the reference's slice-API tests (ggml-quants/src/lib.rs:274-369)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "gguf_amd", "host", "test_slice_api")


def _run(*args):
    assert os.path.exists(EXE), "build it: make -C gguf_amd/host"
    return subprocess.run([EXE, *args], capture_output=True, text=True, timeout=120)


def test_error_cases_without_device():
    r = _run("--no-device")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "slice api ok" in r.stdout


@pytest.mark.gpu
def test_full_program_on_gpu():
    r = _run()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "slice api ok (with device)" in r.stdout


def test_ggus_mirror_program():
    """gguf_amd/host/ggus.hpp: the reference's container tests (simulator.rs:110-131,
    file_writer.rs:150-371, file.rs:313-443) in C++; host only."""
    exe = os.path.join(ROOT, "gguf_amd", "host", "test_ggus_api")
    assert os.path.exists(exe), "build it: make -C gguf_amd/host"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ggus api ok" in r.stdout


def test_container_code_under_host_sanitizers():
    """Reader and writer built with AddressSanitizer + UBSan (host code only): the mirror's test
    program, then 20000 mutated files through the parser."""
    host = os.path.join(ROOT, "gguf_amd", "host")
    r = subprocess.run(["make", "-C", host, "--no-print-directory", "asan"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "fsanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ggus api ok" in r.stdout and "fuzz ok" in r.stdout
