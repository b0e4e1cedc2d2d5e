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
        for out, okind in (("f32", None), ("f16", "f16")):
            res = G.dequantize_file(g, out=out, staging_bytes=1 << 20)
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
