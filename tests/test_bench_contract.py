"""CPU-side checks of bench.py: the defaults the driver relies on and the shape of the
`cpu_baseline` record (the GPU legs run on the GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_defaults(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 1 and a.workload == "llama7b_q4_k_m" and a.out == "f32"
    assert a.scaling == "weak"


def test_cpu_baseline_record():
    sys.path.insert(0, ROOT)
    import bench
    from gguf_amd import synth
    rec = bench.cpu_baseline(synth.workload("q4_0_4096"), "f32", 0.5)
    assert set(rec) >= {"value", "unit", "cores", "kind", "sample"}
    assert rec["kind"] == "port" and rec["unit"] == "GB/s" and rec["value"] > 0 and rec["cores"] >= 1


def test_refuses_to_run_without_gpu_or_with_wrong_world():
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "torch.distributed.run" in r.stderr
