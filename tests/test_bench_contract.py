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
    assert a.scaling == "strong"            # N > 1: ONE tensor set sharded over the ranks
    assert a.streamed_steps >= 1


def test_job_tables():
    """strong: one set whatever the world size; weak: one set per rank; the index partition's shares are
    disjoint and complete."""
    sys.path.insert(0, ROOT)
    import bench
    from gguf_amd import shard
    base, items, label = bench.job_items("llama7b_q4_k_m", "strong", 8)
    assert items == base and label == "llama7b_q4_k_m" and len(items) == 226
    shares = shard.partition(items, 8)
    seen = {}
    for r, sh in enumerate(shares):
        for p in sh:
            for b in (p.block_begin, p.block_begin + p.blocks - 1):
                assert (p.tensor, b) not in seen
                seen[(p.tensor, b)] = r
    assert sum(p.elems for sh in shares for p in sh) == sum(n for _, _, n in items) == 6_738_149_376
    assert shard.imbalance(shares) < 1.02
    _, w_items, w_label = bench.job_items("llama7b_q4_k_m", "weak", 4)
    assert len(w_items) == 4 * 226 and w_label == "4 x llama7b_q4_k_m"


def test_cpu_baseline_record():
    sys.path.insert(0, ROOT)
    import bench
    from gguf_amd import synth
    rec = bench.cpu_baseline(synth.workload("q4_0_4096"), "f32", 0.5, min_elems=1)
    assert set(rec) >= {"value", "unit", "cores", "kind", "sample", "fresh_mapping_value", "build"}
    assert rec["kind"] == "port" and rec["unit"] == "GB/s" and rec["value"] > 0 and rec["cores"] >= 1


def test_refuses_to_run_without_gpu_or_with_wrong_world():
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "torch.distributed.run" in r.stderr
