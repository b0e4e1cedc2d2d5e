"""Synthetic packed-block inputs and the tensor sets of the benchmark configurations.

No model files exist offline, so every workload is a table of (name, shape, type)
with seeded random payload.  Payload rule for formats that have no quantizer in
the reference (K-quants, IQ4): uniformly random bytes, with every f16 header field
(d, dmin / min) overwritten by |N(0,1)| * 2^-6 so that all scales are finite and
normal ("benign" blocks).  "adversarial" blocks keep fully random bytes, so
headers include zeros, subnormals, negatives, infinities and NaNs.
"""
import numpy as np

from .quants import GGmlType, block_bytes, block_elems

# byte offsets of the f16 header fields inside one block
HEADER_F16 = {
    GGmlType.Q4_0: (0,), GGmlType.Q4_1: (0, 2), GGmlType.Q5_0: (0,), GGmlType.Q5_1: (0, 2),
    GGmlType.Q8_0: (0,), GGmlType.Q8_1: (0, 2), GGmlType.Q2K: (80, 82), GGmlType.Q3K: (108,),
    GGmlType.Q4K: (0, 2), GGmlType.Q5K: (0, 2), GGmlType.Q6K: (208,), GGmlType.Q8K: (0,),
    GGmlType.IQ4NL: (0,), GGmlType.IQ4XS: (0,),
}


def random_blocks(ty, blocks, seed, adversarial=False):
    """numpy uint8 [blocks * block_bytes] of packed blocks (host)."""
    ty = GGmlType(int(ty))
    rng = np.random.default_rng(seed)
    bb = block_bytes(ty)
    raw = rng.integers(0, 256, size=(blocks, bb), dtype=np.uint8)
    if ty in (GGmlType.F16, GGmlType.BF16):
        if not adversarial:
            vals = (rng.standard_normal(blocks) * 0.02).astype(np.float32)
            if ty == GGmlType.F16:
                raw = vals.astype(np.float16).view(np.uint8).reshape(blocks, 2)
            else:
                raw = (vals.view(np.uint32) >> 16).astype(np.uint16).view(np.uint8).reshape(blocks, 2)
        return np.ascontiguousarray(raw).reshape(-1)
    if not adversarial:
        for off in HEADER_F16[ty]:
            h = (np.abs(rng.standard_normal(blocks)) * 2.0 ** -6 + 2.0 ** -10).astype(np.float16)
            raw[:, off:off + 2] = h.view(np.uint8).reshape(blocks, 2)
    return raw.reshape(-1)


def random_blocks_device(ty, blocks, seed, device):
    """torch uint8 tensor of benign packed blocks generated on the device (bench-sized inputs)."""
    import torch
    ty = GGmlType(int(ty))
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    bb = block_bytes(ty)
    # randint over int64 is slow and memory hungry; draw int32 words and view as bytes
    nbytes = blocks * bb
    words = (nbytes + 3) // 4
    raw = torch.randint(-2 ** 31, 2 ** 31 - 1, (words,), dtype=torch.int32, device=device,
                        generator=gen).view(torch.uint8)[:nbytes]
    if ty in (GGmlType.F16, GGmlType.BF16):
        vals = torch.randn(blocks, device=device, generator=gen) * 0.02
        dt = torch.float16 if ty == GGmlType.F16 else torch.bfloat16
        return vals.to(dt).view(torch.uint8).reshape(-1)
    blk = raw.view(blocks, bb)
    for off in HEADER_F16[ty]:
        h = (torch.randn(blocks, device=device, generator=gen).abs() * 2.0 ** -6 + 2.0 ** -10)
        blk[:, off:off + 2] = h.to(torch.float16).view(torch.uint8).view(blocks, 2)
    return raw


# ---- tensor sets -------------------------------------------------------------

def q4_k_m_type(name, layer, n_layer):
    """llama.cpp's Q4_K_M mixture: Q4_K everywhere, Q6_K for output.weight and
    for attn_v / ffn_down on the first eighth, last eighth and every third layer between."""
    if name == "output.weight":
        return GGmlType.Q6K
    if name.endswith("attn_v.weight") or name.endswith("ffn_down.weight"):
        i, n = layer, n_layer
        if i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2:
            return GGmlType.Q6K
    return GGmlType.Q4K


def llama_tensor_table(n_embd, n_ff, n_layer, n_vocab, n_head=None, n_kv=None):
    """(name, shape, layer) of every 2-D weight; shape[0] is the contiguous dimension."""
    n_head = n_head or 1
    n_kv = n_kv or n_head
    kv_dim = n_embd // n_head * n_kv
    out = [("token_embd.weight", (n_embd, n_vocab), -1)]
    for i in range(n_layer):
        p = f"blk.{i}."
        out += [
            (p + "attn_q.weight", (n_embd, n_embd), i),
            (p + "attn_k.weight", (n_embd, kv_dim), i),
            (p + "attn_v.weight", (n_embd, kv_dim), i),
            (p + "attn_output.weight", (n_embd, n_embd), i),
            (p + "ffn_gate.weight", (n_embd, n_ff), i),
            (p + "ffn_up.weight", (n_embd, n_ff), i),
            (p + "ffn_down.weight", (n_ff, n_embd), i),
        ]
    out.append(("output.weight", (n_embd, n_vocab), -1))
    return out


def llama7b_table():
    """Llama-7B shapes: 6,738,149,376 quantized elements in 226 tensors."""
    return llama_tensor_table(4096, 11008, 32, 32000)


def llama70b_table():
    """Llama-70B shapes (GQA 64/8 heads): 68,975,329,280 elements."""
    return llama_tensor_table(8192, 28672, 80, 32000, n_head=64, n_kv=8)


def tinyllama_table():
    """TinyLlama-1.1B shapes (22 layers, GQA 32/4): about 1.10 G elements."""
    return llama_tensor_table(2048, 5632, 22, 32000, n_head=32, n_kv=4)


def workload(name):
    """[(tensor name, ggml type, elements)] for a named benchmark workload."""
    def numel(shape):
        n = 1
        for s in shape:
            n *= s
        return n

    if name == "q4_0_4096":
        return [("w", GGmlType.Q4_0, 4096 * 4096)]
    if name in ("llama7b_q4_k_m", "tinyllama_q4_k_m"):
        table, nl = (llama7b_table(), 32) if name.startswith("llama7b") else (tinyllama_table(), 22)
        return [(n, q4_k_m_type(n, l, nl), numel(s)) for n, s, l in table]
    if name.startswith("llama7b_"):
        ty = {"q2_k": GGmlType.Q2K, "q3_k": GGmlType.Q3K, "q4_k": GGmlType.Q4K,
              "q5_k": GGmlType.Q5K, "q6_k": GGmlType.Q6K, "q4_0": GGmlType.Q4_0,
              "q8_0": GGmlType.Q8_0, "iq4_nl": GGmlType.IQ4NL, "iq4_xs": GGmlType.IQ4XS}[name[len("llama7b_"):]]
        return [(n, ty, numel(s)) for n, s, l in llama7b_table()]
    if name == "llama70b_q4_k_m_iq4nl":
        out = []
        for n, s, l in llama70b_table():
            if n.endswith("attn_k.weight") or n.endswith("attn_v.weight"):
                ty = GGmlType.IQ4NL
            else:
                ty = q4_k_m_type(n, l, 80)
            out.append((n, ty, numel(s)))
        return out
    raise KeyError(name)


def workload_bytes(items, out_bytes=4):
    """(packed bytes, decoded bytes, elements) of a workload."""
    pin = sum(n // block_elems(ty) * block_bytes(ty) for _, ty, n in items)
    el = sum(n for _, _, n in items)
    return pin, el * out_bytes, el


def sample_blocks(tensors, count, seed):
    """Draws `count` random blocks (uniformly over all blocks) from device-resident tensors of ONE block
    type and returns (packed bytes, decoded values) of exactly those blocks as numpy arrays on the host:
    the material for checking a large decode against a CPU decoder without decoding everything there.

    tensors: [(ggml_type, src uint8 device tensor, dst device tensor)]; 16-bit outputs come back as
    uint16 bit patterns."""
    import torch
    ty = GGmlType(int(tensors[0][0]))
    bb, n = block_bytes(ty), block_elems(ty)
    counts = np.array([t[1].numel() // bb for t in tensors], dtype=np.int64)
    rng = np.random.default_rng(seed)
    picks = np.sort(rng.integers(0, int(counts.sum()), size=count))
    starts = np.concatenate([[0], np.cumsum(counts)])
    which = np.searchsorted(starts, picks, side="right") - 1
    packed, decoded = [], []
    for ti in np.unique(which):
        idx = torch.from_numpy(picks[which == ti] - starts[ti]).to(tensors[ti][1].device)
        _, src, dst = tensors[ti]
        packed.append(src.view(-1, bb).index_select(0, idx).cpu().numpy().reshape(-1))
        d = dst.view(-1, n).index_select(0, idx)
        if d.element_size() == 2:
            d = d.view(torch.int16)
        decoded.append(d.cpu().numpy().reshape(-1))
    dec = np.concatenate(decoded)
    if dec.dtype == np.int16:
        dec = dec.view(np.uint16)
    return np.concatenate(packed), dec
