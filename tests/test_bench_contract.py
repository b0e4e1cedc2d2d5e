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


def test_refuses_a_world_size_that_contradicts_the_launcher():
    """Under an external launcher (WORLD_SIZE set) --gpus must agree with it."""
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_plain_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself.  Without a GPU every rank
    stops at its own "needs a GPU" check -- in a child process: the parent reports which ranks failed and exits
    non-zero, it neither refuses the command line nor touches the GPU itself."""
    import pytest
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the plain command would run the real two-rank benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") == 2 and "ranks failed (rank, exit code): [(0, 1), (1, 1)]" in r.stderr
    assert r.stdout.strip() == ""


def test_self_launch_relays_rank0_line_and_exit_codes(tmp_path, monkeypatch):
    """The parent of a self-launched run: every child gets its rank and a common rendezvous in the environment;
    rank 0's standard output is passed through; a failing rank makes the whole command fail."""
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "fake_rank.py"
    fake.write_text(
        "import os, sys, json\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "print(json.dumps({'rank': r, 'world': w, 'argv': sys.argv[1:]}))\n"
        "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r)\n")
    monkeypatch.setattr(bench, "__file__", str(fake))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "2"])
    import io
    import contextlib
    import json
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.self_launch(3)
    assert rc == 0
    assert json.loads(buf.getvalue()) == {"rank": 0, "world": 3, "argv": ["--gpus", "3", "--steps", "2"]}
    monkeypatch.setenv("FAIL_RANK", "2")
    with contextlib.redirect_stdout(io.StringIO()):
        assert bench.self_launch(3) == 1
