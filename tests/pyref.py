"""Element-at-a-time pure-Python decoders, written straight from the block
specifications (reference .rs files for the implemented formats, the public ggml
block layout for the rest).  A second, deliberately naive restatement used only to
cross-check the C oracle on small cases: every float step is rounded to f32 with
numpy scalars, nothing is vectorised, nothing is shared with oracle/.
"""
import numpy as np

f32 = np.float32
KV = [-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113]


def h2f(lo, hi):
    return f32(np.array([lo | (hi << 8)], dtype=np.uint16).view(np.float16)[0])


def i8(b):
    return b - 256 if b >= 128 else b


def q4_0(b):
    d = h2f(b[0], b[1])
    y = [None] * 32
    for i in range(16):
        y[i] = f32((b[2 + i] & 15) - 8) * d
        y[i + 16] = f32((b[2 + i] >> 4) - 8) * d
    return y


def q4_1(b):
    d, m = h2f(b[0], b[1]), h2f(b[2], b[3])
    y = [None] * 32
    for i in range(16):
        y[i] = f32(f32(b[4 + i] & 15) * d) + m
        y[i + 16] = f32(f32(b[4 + i] >> 4) * d) + m
    return y


def _q5(b, qoff, hoff):
    qh = b[hoff] | (b[hoff + 1] << 8) | (b[hoff + 2] << 16) | (b[hoff + 3] << 24)
    v = [None] * 32
    for i in range(16):
        v[i] = (b[qoff + i] & 15) | (((qh >> i) & 1) << 4)
        v[i + 16] = (b[qoff + i] >> 4) | (((qh >> (i + 16)) & 1) << 4)
    return v


def q5_0(b):
    d = h2f(b[0], b[1])
    return [f32(v - 16) * d for v in _q5(b, 6, 2)]


def q5_1(b):
    d, m = h2f(b[0], b[1]), h2f(b[2], b[3])
    return [f32(f32(v) * d) + m for v in _q5(b, 8, 4)]


def q8_0(b):
    d = h2f(b[0], b[1])
    return [f32(i8(b[2 + i])) * d for i in range(32)]


def q8_1(b):
    d = h2f(b[0], b[1])
    return [f32(i8(b[4 + i])) * d for i in range(32)]


def q8_k(b):
    d = h2f(b[0], b[1])
    return [f32(i8(b[2 + i])) * d for i in range(256)]


def q2_k(b):
    d, dm = h2f(b[80], b[81]), h2f(b[82], b[83])
    y = []
    for e in range(256):
        half, r = divmod(e, 128)
        j, rr = divmod(r, 32)
        sc = b[e // 16]
        q = (b[16 + 32 * half + rr] >> (2 * j)) & 3
        dl = d * f32(sc & 15)
        ml = dm * f32(sc >> 4)
        y.append(f32(dl * f32(q)) - ml)
    return y


def q3_k(b):
    dall = h2f(b[108], b[109])
    s = b[96:108]
    sc = [0] * 16
    for i in range(4):
        sc[i] = (s[i] & 15) | ((s[8 + i] & 3) << 4)
        sc[4 + i] = (s[4 + i] & 15) | (((s[8 + i] >> 2) & 3) << 4)
        sc[8 + i] = (s[i] >> 4) | (((s[8 + i] >> 4) & 3) << 4)
        sc[12 + i] = (s[4 + i] >> 4) | (((s[8 + i] >> 6) & 3) << 4)
    y = []
    for e in range(256):
        half, r = divmod(e, 128)
        j, rr = divmod(r, 32)
        q = (b[32 + 32 * half + rr] >> (2 * j)) & 3
        hbit = (b[rr] >> (4 * half + j)) & 1
        v = q - (0 if hbit else 4)
        dl = dall * f32(sc[e // 16] - 32)
        y.append(dl * f32(v))
    return y


def _sm(s, j):
    if j < 4:
        return s[j] & 63, s[j + 4] & 63
    return (s[j + 4] & 15) | ((s[j - 4] >> 6) << 4), (s[j + 4] >> 4) | ((s[j] >> 6) << 4)


def q4_k(b):
    d, dm = h2f(b[0], b[1]), h2f(b[2], b[3])
    s = b[4:16]
    y = []
    for e in range(256):
        is_ = e // 32
        sc, mn = _sm(s, is_)
        byte = b[16 + 32 * (e // 64) + (e % 32)]
        q = (byte >> 4) if (is_ & 1) else (byte & 15)
        y.append(f32(f32(d * f32(sc)) * f32(q)) - f32(dm * f32(mn)))
    return y


def q5_k(b):
    d, dm = h2f(b[0], b[1]), h2f(b[2], b[3])
    s = b[4:16]
    y = []
    for e in range(256):
        is_ = e // 32
        sc, mn = _sm(s, is_)
        byte = b[48 + 32 * (e // 64) + (e % 32)]
        q = (byte >> 4) if (is_ & 1) else (byte & 15)
        q += 16 if (b[16 + (e % 32)] >> is_) & 1 else 0
        y.append(f32(f32(d * f32(sc)) * f32(q)) - f32(dm * f32(mn)))
    return y


def q6_k(b):
    d = h2f(b[208], b[209])
    y = []
    for e in range(256):
        half, r = divmod(e, 128)
        qd, l = divmod(r, 32)
        ql = b[64 * half + 32 * (qd & 1) + l]
        lo = (ql >> 4) if qd >= 2 else (ql & 15)
        hi = (b[128 + 32 * half + l] >> (2 * qd)) & 3
        q = (lo | (hi << 4)) - 32
        sc = i8(b[192 + 8 * half + l // 16 + 2 * qd])
        y.append(f32(d * f32(sc)) * f32(q))
    return y


def iq4_nl(b):
    d = h2f(b[0], b[1])
    y = [None] * 32
    for j in range(16):
        y[j] = d * f32(KV[b[2 + j] & 15])
        y[j + 16] = d * f32(KV[b[2 + j] >> 4])
    return y


def iq4_xs(b):
    d = h2f(b[0], b[1])
    sh = b[2] | (b[3] << 8)
    y = []
    for e in range(256):
        ib, r = divmod(e, 32)
        ls = ((b[4 + ib // 2] >> (4 * (ib % 2))) & 15) | (((sh >> (2 * ib)) & 3) << 4)
        dl = d * f32(ls - 32)
        byte = b[8 + 16 * ib + (r % 16)]
        q = (byte >> 4) if r >= 16 else (byte & 15)
        y.append(dl * f32(KV[q]))
    return y


DECODERS = {2: q4_0, 3: q4_1, 6: q5_0, 7: q5_1, 8: q8_0, 9: q8_1, 10: q2_k, 11: q3_k,
            12: q4_k, 13: q5_k, 14: q6_k, 15: q8_k, 20: iq4_nl, 23: iq4_xs}


def decode(ty, packed, block_bytes):
    packed = [int(x) for x in packed]
    out = []
    with np.errstate(all="ignore"):
        for off in range(0, len(packed), block_bytes):
            out += DECODERS[ty](packed[off:off + block_bytes])
    return np.array(out, dtype=np.float32)
