"""The benchmark configurations at full size, through the batch path (ggq_batch_*): every element of every
tensor set against the CPU oracle (the oracle decodes 6.7 G elements in well under a second on the GPU box's
16 cores, so nothing needs to be sampled), and the reference's first configuration (F16 -> Q8_0 -> F32 over
the TinyLlama table) through the quantizer and back."""
import gc
import os

import numpy as np
import pytest

from util import same_bits, same_bits16, first_diff, plan_entries, with_adjacency

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


@pytest.mark.parametrize("name,out", [
    ("llama7b_q4_k_m", "f32"), ("llama7b_q4_k_m", "f16"),
    ("tinyllama_q4_k_m", "f32"), ("tinyllama_q4_k_m", "f16"), ("tinyllama_q4_k_m", "bf16"),
    ("llama7b_q2_k", "f32"), ("llama7b_q3_k", "f32"), ("llama7b_q5_k", "f32"), ("llama7b_q6_k", "f32"),
    ("llama7b_q4_0", "f32"), ("llama7b_q8_0", "f32"), ("llama7b_iq4_nl", "f32"), ("llama7b_iq4_xs", "f32"),
    ("llama70b_q4_k_m_iq4nl", "f16"),
])
def test_every_element_against_oracle(torch_cuda, oracle, name, out):
    """Whole tensor sets: all tensors through one launch plan, and every element (6.7 G per Llama-7B set) compared
    bit for bit with the CPU oracle's decode of the same packed bytes, one tensor at a time on the host.  The first
    entry is the configuration the benchmark's headline is quoted on, the last the streamed one (69 G elements:
    three layers and the two big matrices element by element, sampled blocks over the rest)."""
    import gguf_amd as g
    from gguf_amd import synth
    from gguf_amd.quants import GGmlType, block_elems
    t = torch_cuda
    items = synth.workload(name)
    ob = 4 if out == "f32" else 2
    pin, pout, elems = synth.workload_bytes(items, ob)
    gc.collect()
    t.cuda.empty_cache()             # blocks that earlier tests freed still sit in torch's allocator
    if pin + pout + (4 << 30) > t.cuda.mem_get_info()[0]:
        pytest.skip("not enough free HBM")
    out_dt = {"f32": t.float32, "f16": t.float16, "bf16": t.bfloat16}[out]
    dst_all = t.empty(elems, dtype=out_dt, device="cuda")
    if ob == 2:
        dst_all.view(t.int16).fill_(0x7FC1)
    else:
        dst_all.view(t.int32).fill_(0x7FC00001)
    tensors, at = [], 0
    for i, (_, ty, ne) in enumerate(items):
        src = synth.random_blocks_device(ty, ne // block_elems(ty), 0xC0FFEE00 + i, "cuda")
        tensors.append((ty, src, dst_all[at:at + ne]))
        at += ne
    b = g.Batch(tensors, {"f32": GGmlType.F32, "f16": GGmlType.F16, "bf16": GGmlType.BF16}[out])
    # one launch per block type through the tile table; where the single-tensor kernel of a (type, output) pair
    # has a group start, runs of adjacent tensors with 64 Mi elements and more get launches of their own
    # (util.plan_entries restates the rule; util.with_adjacency finds which of these allocations happen to be adjacent).
    ntypes = len({int(ty) for ty, _, _ in tensors})
    assert b.launches == plan_entries(with_adjacency(tensors), ob)
    if ob == 4 and name != "llama7b_q8_0":
        assert b.launches == ntypes
    b.run()
    t.cuda.synchronize()
    o_out = {"f32": oracle.F32, "f16": oracle.F16, "bf16": oracle.BF16}[out]
    checked = 0
    # The 70B set is 138 GB of output: every element of the first, a middle and the last layer, of the embedding and
    # of the output matrix (3.1 G elements), and 65,536 random blocks per block type over the rest.
    # GGQ_TEST_ALL_ELEMENTS=1 compares all 69 G elements of that set as well (a minute and a half; a one-off whose
    # record is kept under profiles/).
    everything = os.environ.get("GGQ_TEST_ALL_ELEMENTS") == "1"
    whole = lambda n: everything or (not name.startswith("llama70b")) or not n.startswith("blk.") or \
        n.split(".")[1] in ("0", "39", "79")
    rest = [x for k, x in enumerate(tensors) if not whole(items[k][0])]
    for ty in sorted({int(x[0]) for x in rest}):
        packed, got = synth.sample_blocks([x for x in rest if int(x[0]) == ty], 65536, 7 + ty)
        want = oracle.dequantize(ty, packed, o_out, threads=16)
        assert (same_bits16(got, want, out) if ob == 2 else same_bits(got, want)), (name, ty, first_diff(got, want))
    for k, (ty, src, dst) in enumerate(tensors):
        if not whole(items[k][0]):
            checked += dst.numel()
            continue
        want = oracle.dequantize(int(ty), src.cpu().numpy(), o_out, threads=16)
        if ob == 4:
            got = dst.cpu().numpy().view(np.uint32)
            want = want.view(np.uint32)
        else:
            got = dst.view(t.int16).cpu().numpy().view(np.uint16)
        assert np.array_equal(got, want), (name, out, items[k][0], first_diff(got, want))
        checked += got.size
    b.close()
    assert checked == elems
    if name.startswith("llama7b"):
        assert (len(tensors), elems) == (226, 6_738_149_376)
    if name.startswith("llama70b"):
        assert elems == 68_975_329_280


@pytest.mark.parametrize("rank", [0, 7])
def test_one_rank_of_the_eight_gpu_70b_split(torch_cuda, oracle, rank):
    """What ONE GPU of the 8-GPU run of the Llama-70B Q4_K_M + IQ4_NL set executes: its share of the index
    partition (about 8.6 G elements, 5 GB packed, 34 GB of f32 output), decoded twice -- resident, through one
    launch plan, every element against the oracle; then streamed from page-locked host memory through
    ggq_dequantize_async into a second buffer, which must equal the first bit for bit."""
    import gguf_amd as g
    from gguf_amd import shard, synth
    from gguf_amd.quants import GGmlType, block_bytes
    t = torch_cuda
    items = synth.workload("llama70b_q4_k_m_iq4nl")
    shares = shard.partition(items, 8, out_bytes=4)
    assert sum(p.elems for sh in shares for p in sh) == 68_975_329_280
    mine = shares[rank]
    elems = sum(p.elems for p in mine)
    packed_bytes = sum(p.packed_bytes for p in mine)
    assert 8.0e9 < elems < 9.3e9
    gc.collect()
    t.cuda.empty_cache()
    if 2 * elems * 4 + packed_bytes + (4 << 30) > t.cuda.mem_get_info()[0]:
        pytest.skip("not enough free HBM")
    dst_all = t.empty(elems, dtype=t.float32, device="cuda")
    dst_all.view(t.int32).fill_(0x7FC00001)
    tensors, at = [], 0
    for i, p in enumerate(mine):
        src = synth.random_blocks_device(p.ggml_type, p.blocks, 0xC0FFEE05 + 1000 * rank + i, "cuda")
        tensors.append((p.ggml_type, src, dst_all[at:at + p.elems]))
        at += p.elems
    b = g.Batch(tensors, GGmlType.F32)
    b.run()
    t.cuda.synchronize()
    for (ty, src, dst), p in zip(tensors, mine):
        want = oracle.dequantize(int(ty), src.cpu().numpy(), oracle.F32, threads=16).view(np.uint32)
        got = dst.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), (rank, p.name, p.block_begin, first_diff(got, want))
    b.close()
    # streamed: the same packed bytes from page-locked host memory, piece by piece, copy and decode overlapping
    host = t.empty(packed_bytes, dtype=t.uint8, pin_memory=True)
    again = t.empty(elems, dtype=t.float32, device="cuda")
    again.view(t.int32).fill_(0x7FC00002)
    work, at_b, at_e = [], 0, 0
    for (ty, src, _), p in zip(tensors, mine):
        n = p.blocks * block_bytes(ty)
        host[at_b:at_b + n].copy_(src)
        work.append((ty, again[at_e:at_e + p.elems], host[at_b:at_b + n]))
        at_b += n
        at_e += p.elems
    t.cuda.synchronize()
    ctx = g.Context(0, staging_bytes=128 << 20)
    for ty, dst, h in work:
        ctx.dequantize_async(ty, dst, h)
    ctx.sync()
    ctx.close()
    assert t.equal(again.view(t.int32), dst_all.view(t.int32)), (rank, "streamed decode differs from the resident one")
    del again, dst_all, host, tensors, work


def test_tinyllama_f16_q8_0_round_trip_whole_model(torch_cuda, oracle):
    """The first benchmark configuration at full size on the GPU: every 2-D F16 tensor of the reference's
    TinyLlama asset (shapes from tests/golden/tinyllama_table.json; the asset ships no payload, so it is
    seeded) goes F16 -> Q8_0 -> F32 on the device.  Checks: 2048 random blocks per tensor group against the
    oracle's quantizer (bytes) and decoder (bits); the whole set against the reference's own round-trip
    tolerance (q8_0.rs:50-53); the 1-D F32 norm tensors are not touched, as in the reference's cast."""
    import json
    import os
    import gguf_amd as g
    t = torch_cuda
    table = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tinyllama_table.json")))["tensors"]
    mats = [x for x in table if x["type"] == 1 and len(x["shape"]) == 2]
    assert len(mats) == 90 and sum(1 for x in table if x["type"] == 0) == 45
    gen = t.Generator(device="cuda")
    gen.manual_seed(0xC0FFEE01)
    total, worst = 0, 0.0
    rng = np.random.default_rng(1)
    for i, x in enumerate(mats):
        n = x["shape"][0] * x["shape"][1]
        assert x["shape"][0] % 32 == 0
        w16 = (t.randn(n, device="cuda", generator=gen) * 0.02).to(t.float16)
        q = t.empty(n // 32 * 34, dtype=t.uint8, device="cuda")
        g.Q8_0.quantize_slice(q, w16)
        y = t.empty(n, dtype=t.float32, device="cuda")
        g.Q8_0.dequantize_slice(y, q)
        worst = max(worst, float((y - w16.float()).abs().max()))
        total += n
        if i % 9 == 0 or n > 50_000_000:                    # sampled blocks of every ninth tensor and the two big ones
            idx = t.from_numpy(rng.integers(0, n // 32, 2048)).cuda()
            xin = w16.view(-1, 32)[idx].contiguous().view(t.int16).cpu().numpy().view(np.uint16).reshape(-1)
            want_q = oracle.quantize(8, xin, in_ty=oracle.F16)
            assert np.array_equal(q.view(-1, 34)[idx].cpu().numpy().reshape(-1), want_q), x["name"]
            want = oracle.dequantize(8, want_q)
            assert same_bits(y.view(-1, 32)[idx].cpu().numpy().reshape(-1), want), x["name"]
        del w16, q, y
    assert total == 1_099_956_224
    assert worst < 4.5e-3


def test_tinyllama_f16_to_q8_0_one_plan_every_byte(torch_cuda, oracle):
    """The first benchmark configuration's quantizing half at full size through ONE quantize plan
    (ggq_quant_batch_*): the 90 two-dimensional F16 tensors of the TinyLlama table (1.1 G elements, seeded payload)
    to Q8_0 with a single kernel launch, every packed byte against the oracle's quantizer."""
    import json
    import os
    import gguf_amd as g
    t = torch_cuda
    table = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tinyllama_table.json")))["tensors"]
    mats = [x for x in table if x["type"] == 1 and len(x["shape"]) == 2]
    assert len(mats) == 90
    gen = t.Generator(device="cuda")
    gen.manual_seed(0xC0FFEE01)
    tensors = []
    for x in mats:
        n = x["shape"][0] * x["shape"][1]
        w16 = (t.randn(n, device="cuda", generator=gen) * 0.02).to(t.float16)
        tensors.append((8, w16, t.empty(n // 32 * 34, dtype=t.uint8, device="cuda")))
    b = g.QuantBatch(tensors)
    # one launch through the tile table; separate allocations that happen to be adjacent in memory form a run, and a
    # run of 64 Mi elements gets a launch of its own (none of the 90 matrices is that large by itself)
    assert b.table_launches <= 1 and b.launches <= 1 + sum(w.numel() for _, w, _ in tensors) // (64 << 20)
    b.run()
    t.cuda.synchronize()
    from concurrent.futures import ThreadPoolExecutor

    def same(k):                # the oracle's quantizer is single-threaded; ctypes releases the interpreter lock
        _, w16, q = tensors[k]
        want = oracle.quantize(8, w16.view(t.int16).cpu().numpy().view(np.uint16), in_ty=oracle.F16)
        return np.array_equal(q.cpu().numpy(), want)

    with ThreadPoolExecutor(8) as pool:
        for x, ok in zip(mats, pool.map(same, range(len(tensors)))):
            assert ok, x["name"]
    total = sum(w16.numel() for _, w16, _ in tensors)
    b.close()
    assert total == 1_099_956_224
