"""End to end on the GPU: a GGUF file on disk -> mmap reader -> streamed decode -> device tensors,
checked against the oracle (the benchmark's first and third configurations in miniature: a TinyLlama-shaped layer set
under the Q4_K_M mixture, plus legacy and 16-bit tensors)."""
import numpy as np
import pytest

from util import same_bits, same_bits16

pytestmark = pytest.mark.gpu


def test_file_to_device_tensors(tmp_path, oracle):
    import torch
    from gguf_amd import gguf as G, synth
    from gguf_amd.quants import GGmlType as T
    assert torch.cuda.is_available()
    spec = [("token_embd.weight", T.Q4K, (2048, 96)), ("blk.0.attn_q.weight", T.Q4K, (2048, 64)),
            ("blk.0.attn_v.weight", T.Q6K, (2048, 8)), ("blk.0.ffn_down.weight", T.Q6K, (5632, 16)),
            ("blk.0.attn_norm.weight", T.F32, (2048,)), ("blk.1.ffn_up.weight", T.Q4_0, (2048, 33)),
            ("blk.1.ffn_gate.weight", T.Q8_0, (2048, 7)), ("blk.2.attn_k.weight", T.IQ4NL, (2048, 5)),
            ("blk.2.w16.weight", T.F16, (2048, 3)), ("blk.2.wb16.weight", T.BF16, (1000,)),
            ("blk.3.q2.weight", T.Q2K, (4096, 3)), ("blk.3.q3.weight", T.Q3K, (4096, 3)),
            ("blk.3.q5.weight", T.Q5K, (4096, 3)), ("output.weight", T.Q6K, (2048, 40))]
    tensors, packed = [], {}
    for i, (name, ty, shape) in enumerate(spec):
        n = int(np.prod(shape))
        if ty == T.F32:
            data = np.random.default_rng(i).standard_normal(n).astype(np.float32).view(np.uint8)
        else:
            data = synth.random_blocks(ty, n // oracle.block_elems(int(ty)), 0xC0FFEE03 + i)
        packed[name] = (int(ty), data)
        tensors.append((name, int(ty), shape, data))
    path = tmp_path / "tiny-q4_k_m.gguf"
    path.write_bytes(G.build_gguf({"general.architecture": "llama", "general.file_type": 15,
                                   "llama.block_count": 4}, tensors))
    with G.GGuf(path) as g:
        assert g.meta("general.file_type") == 15 and len(g.tensors) == len(spec)
        # one group: one kernel launch per block type present (8) plus one per F16 / BF16 tensor (2), for 13 tensors
        stats = {}
        G.dequantize_file(g, stats=stats)
        assert (stats["tensors"], stats["groups"], stats["launches"]) == (13, 1, 10)
        for out, okind, group_bytes in (("f32", None, 1 << 30), ("f16", "f16", 150_000), ("f32", None, 1)):
            stats = {}
            res = G.dequantize_file(g, out=out, staging_bytes=1 << 20, group_bytes=group_bytes, stats=stats)
            assert stats["groups"] == {1 << 30: 1, 1: 13}.get(group_bytes, stats["groups"]) and stats["groups"] >= 1
            if group_bytes == 150_000:
                assert 2 <= stats["groups"] < 13
            assert set(res) == {n for n, _, _ in spec}
            for name, (ty, data) in packed.items():
                got = res[name]
                if ty == 0:
                    assert np.array_equal(got.cpu().numpy().view(np.uint8), data)
                    continue
                if out == "f32":
                    want = oracle.dequantize(ty, data)
                    assert same_bits(got.cpu().numpy(), want), name
                else:
                    want = oracle.dequantize(ty, data, oracle.F16)
                    assert same_bits16(got.view(torch.int16).cpu().numpy().view(np.uint16), want, "f16"), name


def test_upload_async_pageable_and_pinned():
    """ggq_upload_async: bytes arrive unchanged, from pageable memory (staged, several chunks, odd sizes
    and offsets) and from page-locked memory (read in place)."""
    import torch
    import gguf_amd as g
    rng = np.random.default_rng(3)
    ctx = g.Context(0, staging_bytes=2 << 20)
    try:
        work = []
        for n, off in ((1, 0), (5_000_001, 3), (1 << 20, 1), ((2 << 20) + 17, 0), (12345, 7)):
            host = rng.integers(0, 256, n + off, dtype=np.uint8)[off:]
            dst = torch.zeros(n, dtype=torch.uint8, device="cuda")
            ctx.upload_async(dst, host)
            work.append((host.copy(), dst))
        pinned = torch.from_numpy(rng.integers(0, 256, 3_000_000, dtype=np.uint8)).pin_memory()
        dst = torch.zeros(3_000_000, dtype=torch.uint8, device="cuda")
        ctx.upload_async(dst, pinned)
        ctx.sync()
        assert np.array_equal(dst.cpu().numpy(), pinned.numpy())
        for host, dst in work:
            assert np.array_equal(dst.cpu().numpy(), host)
        with pytest.raises(g.LengthMismatch):
            ctx.upload_async(dst, np.zeros(5, dtype=np.uint8))
    finally:
        ctx.close()
