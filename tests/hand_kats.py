"""Known-answer blocks whose expected outputs were worked out by hand from the block formulas
and typed in as literals -- no decoder, neither the C oracle nor tests/pyref.py, produced them.

Each case: (ggml type id, packed block as uint8 array, expected float32 values).  The same cases
check the CPU oracle (test_oracle_golden.py) and the HIP kernels (test_gpu_parity.py).

Field layouts are the reference's struct field orders (ggml-quants/src/structs/q2_k.rs:6-13,
q3_k.rs:6-15, q4_k.rs:6-13, q5_k.rs:7-18, q6_k.rs:7-16, q5_0.rs:7-14, q5_1.rs:7-14, q8_1.rs:8-16,
q8_k.rs:8-15) and, for IQ4_NL / IQ4_XS, the 18 / 136-byte file layout.  f16 constants used:
0x3C00 = 1.0, 0x3800 = 0.5, 0x3400 = 0.25, 0x4000 = 2.0, 0xC000 = -2.0.
"""
import numpy as np

KV_IQ4NL = [-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113]


def _blk(n):
    return np.zeros(n, dtype=np.uint8)


def _f16(b, at, bits):
    b[at] = bits & 0xFF
    b[at + 1] = bits >> 8


# 6-bit scales / mins of the Q4_K and Q5_K cases, packed into the 12 scale bytes s[0..11]:
#   j < 4 : sc = s[j] & 63,                         mn = s[j+4] & 63
#   j >= 4: sc = (s[j+4] & 15) | ((s[j-4] >> 6) << 4),   mn = (s[j+4] >> 4) | ((s[j] >> 6) << 4)
#   s[0] = 0xC1: sc0 = 1,  top bits 3 -> sc4 gets 48      s[4] = 0x45: mn0 = 5, top bits 1 -> mn4 gets 16
#   s[1] = 0x82: sc1 = 2,  top bits 2 -> sc5 gets 32      s[5] = 0x86: mn1 = 6, top bits 2 -> mn5 gets 32
#   s[2] = 0x43: sc2 = 3,  top bits 1 -> sc6 gets 16      s[6] = 0xC7: mn2 = 7, top bits 3 -> mn6 gets 48
#   s[3] = 0x04: sc3 = 4,  top bits 0                     s[7] = 0x08: mn3 = 8, top bits 0
#   s[8] = 0x21: sc4 = 1|48 = 49, mn4 = 2|16 = 18         s[9]  = 0x43: sc5 = 3|32 = 35, mn5 = 4|32 = 36
#   s[10] = 0x65: sc6 = 5|16 = 21, mn6 = 6|48 = 54        s[11] = 0x87: sc7 = 7,         mn7 = 8
K4_SCALES = [0xC1, 0x82, 0x43, 0x04, 0x45, 0x86, 0xC7, 0x08, 0x21, 0x43, 0x65, 0x87]
# quants: 32-byte group g (0..3) holds low nibble g+1 (sub-block 2g) and high nibble g+5 (sub-block 2g+1)
K4_QS = [((g + 5) << 4) | (g + 1) for g in range(4) for _ in range(32)]


def q4_k():
    b = _blk(144)
    _f16(b, 0, 0x3C00)            # d = 1
    _f16(b, 2, 0x3800)            # dmin = 0.5
    b[4:16] = K4_SCALES
    b[16:144] = K4_QS
    # y = d*sc*q - dmin*mn per 32-element sub-block:
    #  0: 1*1 - 2.5 = -1.5     1: 2*5 - 3 = 7        2: 3*2 - 3.5 = 2.5     3: 4*6 - 4 = 20
    #  4: 49*3 - 9 = 138       5: 35*7 - 18 = 227    6: 21*4 - 27 = 57      7: 7*8 - 4 = 52
    per = [-1.5, 7.0, 2.5, 20.0, 138.0, 227.0, 57.0, 52.0]
    return 12, b, [v for v in per for _ in range(32)]


def q5_k():
    b = _blk(176)
    _f16(b, 0, 0x3C00)
    _f16(b, 2, 0x3800)
    b[4:16] = K4_SCALES
    # qh[l] = 0xA5 (bits 0,2,5,7) for even l, 0x5A (bits 1,3,4,6) for odd l; sub-block k adds 16 where bit k is set
    b[16:48] = [0xA5 if l % 2 == 0 else 0x5A for l in range(32)]
    b[48:176] = K4_QS
    # (even l, odd l) per sub-block, y = d*sc*(q + 16*bit) - dmin*mn:
    #  0: bit0 even -> 1*17-2.5 = 14.5 | 1*1-2.5 = -1.5        1: bit1 odd  -> 2*5-3 = 7      | 2*21-3 = 39
    #  2: bit2 even -> 3*18-3.5 = 50.5 | 3*2-3.5 = 2.5         3: bit3 odd  -> 4*6-4 = 20     | 4*22-4 = 84
    #  4: bit4 odd  -> 49*3-9 = 138    | 49*19-9 = 922         5: bit5 even -> 35*23-18 = 787 | 35*7-18 = 227
    #  6: bit6 odd  -> 21*4-27 = 57    | 21*20-27 = 393        7: bit7 even -> 7*24-4 = 164   | 7*8-4 = 52
    per = [(14.5, -1.5), (7.0, 39.0), (50.5, 2.5), (20.0, 84.0), (138.0, 922.0), (787.0, 227.0), (57.0, 393.0),
           (164.0, 52.0)]
    return 13, b, [pair[l % 2] for pair in per for l in range(32)]


# 2-bit codes of the Q2_K and Q3_K cases.  Byte 0xE4 = 11 10 01 00b reads 0,1,2,3 at shifts 0,2,4,6;
# byte 0x1B = 00 01 10 11b reads 3,2,1,0.  First half (bytes 0..31): 16 x 0xE4 then 16 x 0x1B; second
# half (bytes 32..63): 16 x 0x1B then 16 x 0xE4.  Sub-block i (16 elements): half i//8, shift 2*((i%8)//2),
# first or second 16 bytes of the half by i%2.  Codes per sub-block:
Q23_QS = [0xE4] * 16 + [0x1B] * 16 + [0x1B] * 16 + [0xE4] * 16
Q23_CODES = [0, 3, 1, 2, 2, 1, 3, 0, 3, 0, 2, 1, 1, 2, 0, 3]


def q2_k():
    b = _blk(84)
    # scales[i] = (mn << 4) | sc with sc = i%5 + 1, mn = 2*(i%3)
    sc = [1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1]
    mn = [0, 2, 4, 0, 2, 4, 0, 2, 4, 0, 2, 4, 0, 2, 4, 0]
    b[0:16] = [(m << 4) | s for s, m in zip(sc, mn)]
    b[16:80] = Q23_QS
    _f16(b, 80, 0x3C00)           # d = 1
    _f16(b, 82, 0x3800)           # dmin = 0.5
    # y = d*sc*code - dmin*mn:
    #  0: 1*0-0 = 0    1: 2*3-1 = 5    2: 3*1-2 = 1    3: 4*2-0 = 8    4: 5*2-1 = 9    5: 1*1-2 = -1
    #  6: 2*3-0 = 6    7: 3*0-1 = -1   8: 4*3-2 = 10   9: 5*0-0 = 0   10: 1*2-1 = 1   11: 2*1-2 = 0
    # 12: 3*1-0 = 3   13: 4*2-1 = 7   14: 5*0-2 = -2  15: 1*3-0 = 3
    per = [0.0, 5.0, 1.0, 8.0, 9.0, -1.0, 6.0, -1.0, 10.0, 0.0, 1.0, 0.0, 3.0, 7.0, -2.0, 3.0]
    return 10, b, [v for v in per for _ in range(16)]


def q3_k():
    b = _blk(110)
    # hmask bit i//2 of byte l (l = 0..15 for even i, 16..31 for odd i) set -> subtract 0, clear -> subtract 4.
    # bytes 0..15: 0xAA (bits 1,3,5,7) at even l, 0x55 (bits 0,2,4,6) at odd l; bytes 16..31: 0x0F (bits 0..3).
    b[0:16] = [0xAA if l % 2 == 0 else 0x55 for l in range(16)]
    b[16:32] = 0x0F
    b[32:96] = Q23_QS
    # 6-bit scales v[i], used as v - 32:
    #   v = 33 34 31 30 48 16 63  0 35 29 40 24 36 28 47 17   ->   1 2 -1 -2 16 -16 31 -32 3 -3 8 -8 4 -4 15 -15
    # packing (k = 0..3): s[k] = lo4(v[k]) | lo4(v[8+k]) << 4;  s[4+k] = lo4(v[4+k]) | lo4(v[12+k]) << 4;
    #                     s[8+k] = hi2(v[k]) | hi2(v[4+k]) << 2 | hi2(v[8+k]) << 4 | hi2(v[12+k]) << 6
    #   lo4: 1 2 15 14 | 0 0 15 0 | 3 13 8 8 | 4 12 15 1        hi2: 2 2 1 1 | 3 1 3 0 | 2 1 2 1 | 2 1 2 1
    b[96:108] = [0x31, 0xD2, 0x8F, 0x8E, 0x40, 0xC0, 0xFF, 0x10, 0xAE, 0x56, 0xAD, 0x51]
    _f16(b, 108, 0x3C00)          # d = 1
    # y = d*(v-32) * (code - (hmask bit ? 0 : 4)); (even l, odd l) for even i, one value for odd i:
    #  0: bit0: AA clear, 55 set  -> 1*(0-4) = -4 | 1*0 = 0          1: 0F bit0 set   -> 2*3 = 6
    #  2: bit1: AA set, 55 clear  -> -1*1 = -1 | -1*(1-4) = 3        3: 0F bit1 set   -> -2*2 = -4
    #  4: bit2: AA clear, 55 set  -> 16*(2-4) = -32 | 16*2 = 32      5: 0F bit2 set   -> -16*1 = -16
    #  6: bit3: AA set, 55 clear  -> 31*3 = 93 | 31*(3-4) = -31      7: 0F bit3 set   -> -32*0 = -0.0
    #  8: bit4: AA clear, 55 set  -> 3*(3-4) = -3 | 3*3 = 9          9: 0F bit4 clear -> -3*(0-4) = 12
    # 10: bit5: AA set, 55 clear  -> 8*2 = 16 | 8*(2-4) = -16       11: 0F bit5 clear -> -8*(1-4) = 24
    # 12: bit6: AA clear, 55 set  -> 4*(1-4) = -12 | 4*1 = 4        13: 0F bit6 clear -> -4*(2-4) = 8
    # 14: bit7: AA set, 55 clear  -> 15*0 = 0 | 15*(0-4) = -60      15: 0F bit7 clear -> -15*(3-4) = 15
    per = [(-4.0, 0.0), 6.0, (-1.0, 3.0), -4.0, (-32.0, 32.0), -16.0, (93.0, -31.0), -0.0,
           (-3.0, 9.0), 12.0, (16.0, -16.0), 24.0, (-12.0, 4.0), 8.0, (0.0, -60.0), 15.0]
    want = []
    for v in per:
        want += [v[l % 2] if isinstance(v, tuple) else v for l in range(16)]
    return 11, b, want


def q6_k():
    b = _blk(210)
    # first half: ql = 0, qh = 0 -> q = -32; scales 1; second half below.  d = 0.5
    b[192:200] = 1
    # second half: ql[64+l] = 0x2B (low 11, high 2), ql[96+l] = 0x9C (low 12, high 9),
    # qh[32+l] = 0x78 = 01 11 10 00b -> top bits 0, 2, 3, 1 for q1..q4:
    #   q1 = (11 | 0<<4) - 32 = -21   q2 = (12 | 2<<4) - 32 = 12   q3 = (2 | 3<<4) - 32 = 18   q4 = (9 | 1<<4) - 32 = -7
    b[64:96] = 0x2B
    b[96:128] = 0x9C
    b[160:192] = 0x78
    # int8 scales of the second half: -3 5 | 7 -128 | 127 -1 | 2 -64  (q1 l<16, l>=16 | q2 .. | q3 .. | q4 ..)
    b[200:208] = [0xFD, 0x05, 0x07, 0x80, 0x7F, 0xFF, 0x02, 0xC0]
    _f16(b, 208, 0x3800)
    # y = (d*sc)*q: first half 0.5*1*-32 = -16; second half per 16 elements:
    #   -1.5*-21 = 31.5   2.5*-21 = -52.5   3.5*12 = 42   -64*12 = -768   63.5*18 = 1143   -0.5*18 = -9   1*-7 = -7   -32*-7 = 224
    per = [31.5, -52.5, 42.0, -768.0, 1143.0, -9.0, -7.0, 224.0]
    return 14, b, [-16.0] * 128 + [v for v in per for _ in range(16)]


def iq4_nl():
    b = _blk(18)
    _f16(b, 0, 0xC000)            # d = -2
    b[2:18] = [((15 - j) << 4) | j for j in range(16)]     # low nibble j, high nibble 15 - j: all 16 codes
    want = [254.0, 208.0, 166.0, 130.0, 98.0, 70.0, 44.0, 20.0, -2.0, -26.0, -50.0, -76.0, -106.0, -138.0, -178.0,
            -226.0]
    return 20, b, want + want[::-1]


def iq4_xs():
    b = _blk(136)
    _f16(b, 0, 0x3C00)            # d = 1
    # 6-bit sub-block scales ls = 33 31 0 63 16 48 34 30, used as ls - 32 = 1 -1 -32 31 -16 16 2 -2
    #   low nibbles 1 15 0 15 0 0 2 14 -> scales_l = F1 F0 00 E2
    #   top bits    2 1 0 3 1 3 2 1    -> scales_h = 2 | 1<<2 | 0<<4 | 3<<6 | 1<<8 | 3<<10 | 2<<12 | 1<<14 = 0x6DC6
    b[2:4] = [0xC6, 0x6D]
    b[4:8] = [0xF1, 0xF0, 0x00, 0xE2]
    for ib in range(8):
        b[8 + 16 * ib: 8 + 16 * ib + 16] = [((15 - j) << 4) | j for j in range(16)]
    mult = [1, -1, -32, 31, -16, 16, 2, -2]
    want = []
    for m in mult:
        want += [float(m * k) for k in KV_IQ4NL] + [float(m * k) for k in KV_IQ4NL[::-1]]
    # spot values typed out: sub-block 2 (x -32): -127 -> 4064; sub-block 3 (x 31): 113 -> 3503, -127 -> -3937
    assert want[64] == 4064.0 and want[96 + 15] == 3503.0 and want[96] == -3937.0
    return 23, b, want


def q5_0():
    b = _blk(22)
    _f16(b, 0, 0x4000)            # d = 2
    b[2:6] = [0x55, 0x55, 0xAA, 0xAA]   # qh = 0xAAAA5555: bit i set for even i < 16, bit 16+i set for odd i
    b[6:22] = [((15 - i) << 4) | i for i in range(16)]
    lo = [2.0 * (i + (16 if i % 2 == 0 else 0) - 16) for i in range(16)]        # 0, -30, 4, -26, ...
    hi = [2.0 * ((15 - i) + (16 if i % 2 == 1 else 0) - 16) for i in range(16)]
    assert lo[:4] == [0.0, -30.0, 4.0, -26.0] and hi[:4] == [-2.0, 28.0, -6.0, 24.0]
    return 6, b, lo + hi


def q5_1():
    b = _blk(24)
    _f16(b, 0, 0x3C00)            # d = 1
    _f16(b, 2, 0x3800)            # min = 0.5
    b[4:8] = [0x55, 0x55, 0xAA, 0xAA]
    b[8:24] = [((15 - i) << 4) | i for i in range(16)]
    lo = [i + (16 if i % 2 == 0 else 0) + 0.5 for i in range(16)]               # 16.5, 1.5, 18.5, 3.5, ...
    hi = [(15 - i) + (16 if i % 2 == 1 else 0) + 0.5 for i in range(16)]
    assert lo[:4] == [16.5, 1.5, 18.5, 3.5] and hi[:4] == [15.5, 30.5, 13.5, 28.5]
    return 7, b, lo + hi


def q8_1():
    b = _blk(36)
    _f16(b, 0, 0x3400)            # d = 0.25
    _f16(b, 2, 0x7E00)            # sum = NaN: not used by the decode
    b[4:36] = np.arange(-16, 16, dtype=np.int8).view(np.uint8)
    return 9, b, [0.25 * q for q in range(-16, 16)]


def q8_k():
    b = _blk(290)
    _f16(b, 0, 0x3800)            # delta = 0.5 (an f16 in the reference's layout)
    b[2:258] = np.arange(-128, 128, dtype=np.int16).astype(np.int8).view(np.uint8)
    b[258:290] = 0xEE             # sums: not used by the decode
    return 15, b, [0.5 * q for q in range(-128, 128)]


def cases():
    out = []
    for fn in (q4_k, q5_k, q2_k, q3_k, q6_k, iq4_nl, iq4_xs, q5_0, q5_1, q8_1, q8_k):
        ty, blk, want = fn()
        out.append((ty, blk, np.array(want, dtype=np.float32)))
    return out


# ---- quantizers: input blocks whose packed bytes were worked out by hand -----------------------------------
# (ggml-quants/src/structs/q4_0.rs:23-44, q4_1.rs:23-47, q5_0.rs:26-58, q5_1.rs:26-62, q8_0.rs:23-41.)
# Each case: (ggml type id, 32 float32 inputs, expected packed bytes).  None of the scaled values lies within 0.01
# of a rounding boundary, so the last bit of 1 / delta cannot move a code.
#   Q4_0, x = -16 .. 15: max_by_abs = -16 (the first element of largest magnitude), delta = -16 / -8 = 2 (f16 0x4000),
#       code = trunc(min(x / 2 + 8.5, 15)) = (i + 1) // 2 capped at 15; byte j = code[j] | code[j + 16] << 4.
#   Q5_0, x = -16 .. 15: delta = -16 / -16 = 1 (0x3C00), code = trunc(x + 16.5) = i; fifth bits of elements 16 .. 31
#       -> qh = 0xFFFF0000; byte j = (j & 15) | (j & 15) << 4.
#   Q4_1, x = 0 .. 31: min 0, delta = 31 / 15 = 2.0667 (f16: 2 * (1 + 34 / 1024) -> 0x4022), code =
#       trunc(i * 15 / 31 + 0.5) = 0,0,1,1, ... ,15,15.
#   Q5_1, x = 0 .. 31: min 0, delta = 1 (0x3C00), code = i, qh = 0xFFFF0000.
#   Q8_0, x = 0 .. 31: delta = 31 / 127 = 0.244094 (f16: 2^-3 * (1 + 976 / 1024) -> 0x33D0), code = round(i * 127 / 31).
Q8_0_CODES = [0, 4, 8, 12, 16, 20, 25, 29, 33, 37, 41, 45, 49, 53, 57, 61, 66, 70, 74, 78, 82, 86, 90, 94, 98, 102, 107,
              111, 115, 119, 123, 127]


def quantizer_cases():
    ramp = np.arange(32, dtype=np.float32)
    out = []
    q4 = [min(15, (i + 1) // 2) for i in range(32)]
    out.append((2, ramp - 16, np.array([0x00, 0x40] + [q4[j] | q4[j + 16] << 4 for j in range(16)], dtype=np.uint8)))
    out.append((6, ramp - 16, np.array([0x00, 0x3C, 0x00, 0x00, 0xFF, 0xFF] + [0x11 * j for j in range(16)], dtype=np.uint8)))
    q41 = [i // 2 for i in range(32)]
    out.append((3, ramp, np.array([0x22, 0x40, 0x00, 0x00] + [q41[j] | q41[j + 16] << 4 for j in range(16)], dtype=np.uint8)))
    out.append((7, ramp, np.array([0x00, 0x3C, 0x00, 0x00, 0x00, 0x00, 0xFF, 0xFF] + [0x11 * j for j in range(16)],
                                  dtype=np.uint8)))
    out.append((8, ramp, np.array([0xD0, 0x33] + Q8_0_CODES, dtype=np.uint8)))
    return out
