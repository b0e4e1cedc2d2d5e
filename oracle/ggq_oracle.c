/*
 * ggq_oracle.c -- CPU restatement of the reference's per-block GGUF
 * quantize / dequantize arithmetic.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP kernels, never the thing shipped or
 * measured: only the test-suite, the smoke check and the benchmark's
 * cpu_baseline leg may load it.  The product library (libggq.so) does not
 * link, include or call anything in this directory.
 *
 * What it restates (paths relative to the reference checkout):
 *   slice API + error order ........ ggml-quants/src/lib.rs:116-148
 *   f16/bf16 element blocks ........ ggml-quants/src/structs/half.rs:8-38
 *   f16/bf16 output adapters ....... ggml-quants/src/lib.rs:62-90
 *   DeltaMin, fold helpers ......... ggml-quants/src/structs.rs:54-107
 *   Q4_0 ........................... ggml-quants/src/structs/q4_0.rs:23-57
 *   Q4_1 ........................... ggml-quants/src/structs/q4_1.rs:23-60
 *   Q5_0 ........................... ggml-quants/src/structs/q5_0.rs:26-73
 *   Q5_1 ........................... ggml-quants/src/structs/q5_1.rs:26-77
 *   Q8_0 ........................... ggml-quants/src/structs/q8_0.rs:23-47
 *   Q8_1 ........................... ggml-quants/src/structs/q8_1.rs:28-61
 *   Q8K (reference layout, 290 B) .. ggml-quants/src/structs/q8_k.rs:27-60
 *   block sizing ................... ggus/src/tensor.rs:83-96, 102-144
 *   to-llama's tensor scaling ...... xtask/src/utils/operator/to_llama.rs:99-125
 *
 * Pinning status.
 *   The reference is Rust and cannot be built in this environment, and it
 *   holds no golden vectors: every numeric test it has is an unseeded
 *   tolerance round trip (lib.rs:173-194) plus the slice error cases
 *   (lib.rs:293-338).  This restatement is checked against exactly those
 *   (tests/test_oracle_reference_tests.py), against numpy's IEEE binary16
 *   for the conversions that the reference delegates to the `half` 2.6.0
 *   crate, and against hand-derived known-answer blocks.
 *   Q2_K..Q6_K, IQ4_NL and IQ4_XS are `todo!()` in the reference (only the
 *   struct layouts exist: q2_k.rs:6-13, q3_k.rs:6-15, q4_k.rs:6-13,
 *   q5_k.rs:7-18, q6_k.rs:7-16, iq4nl.rs:6-11, iq4xs.rs:6-15); their
 *   arithmetic here follows the public ggml block specification with every
 *   float operation a separate IEEE f32 operation (what Rust would emit).
 *   For those seven formats PARITY IS UNPINNED by the reference.
 *
 * Third-party arithmetic: f16/bf16 <-> f32 lives in the crate `half`
 * (Cargo.lock pins 2.6.0), which is not vendored.  Its published behaviour:
 * widening is exact and quiets signalling NaNs; narrowing is IEEE
 * round-to-nearest-even, NaN -> sign | exp all ones | quiet bit | top payload.
 * From f64 (used only for the scaling factor): f16::from_f64 picks its routine
 * at run time -- on an x86 host with F16C it narrows `f as f32` with the
 * hardware instruction, i.e. two round-to-nearest-even steps, which is what is
 * restated here; its portable routine and bf16::from_f64 work on the upper 32
 * bits of the f64 only (the lower 32 are dropped before rounding).  These
 * routines are restated from the crate's published source as remembered, with
 * no copy of it at hand: for factors within one rounding step of a tie the
 * factor conversion is PARITY UNPINNED.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see Makefile).
 * -ffp-contract=off matters: Rust never fuses a*b+c.
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- status codes (same numbering as the product ABI, kept independent) -- */
enum { GGO_OK = 0, GGO_INDIVISIBLE = 1, GGO_LENGTH_MISMATCH = 2, GGO_UNSUPPORTED = 3 };

/* ---- GGmlType discriminants, ggus/src/tensor.rs:14-51 -------------------- */
enum {
    T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7,
    T_Q8_0 = 8, T_Q8_1 = 9, T_Q2K = 10, T_Q3K = 11, T_Q4K = 12, T_Q5K = 13,
    T_Q6K = 14, T_Q8K = 15, T_IQ4NL = 20, T_IQ4XS = 23, T_BF16 = 30
};

static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ======================================================================== *
 *  half-precision conversions (crate `half` 2.6.0 semantics)               *
 * ======================================================================== */

/* binary16 -> binary32, exact; NaN keeps sign and payload, quiet bit set. */
float ggo_f16_to_f32(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    if (exp == 0x1F) {
        if (man == 0) return bits_f32(sign | 0x7F800000u);
        return bits_f32(sign | 0x7FC00000u | (man << 13));
    }
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        /* subnormal half: value = man * 2^-24, renormalise */
        int shift = 0;
        while (!(man & 0x400u)) { man <<= 1; ++shift; }
        man &= 0x3FFu;
        return bits_f32(sign | ((uint32_t)(113 - shift) << 23) | (man << 13));
    }
    return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* binary32 -> binary16, round to nearest even. */
uint16_t ggo_f32_to_f16(float f)
{
    uint32_t x = f32_bits(f);
    uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFF) {
        if (man == 0) return sign | 0x7C00u;
        return (uint16_t)(sign | 0x7E00u | (man >> 13));
    }
    int e = (int)exp - 127 + 15;             /* biased half exponent */
    if (e >= 0x1F) return sign | 0x7C00u;    /* overflow -> inf       */
    if (e <= 0) {
        /* result is subnormal (or rounds up into the smallest normal) */
        if (e < -10) return sign;            /* below half the smallest subnormal */
        man |= 0x800000u;
        unsigned shift = (unsigned)(14 - e); /* 14..24 */
        uint32_t q = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) ++q;
        return (uint16_t)(sign | q);
    }
    uint32_t q = ((uint32_t)e << 10) | (man >> 13);
    uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ++q; /* may carry into exp/inf */
    return (uint16_t)(sign | q);
}

float ggo_bf16_to_f32(uint16_t b) { return bits_f32((uint32_t)b << 16); }

/* binary32 -> bfloat16, round to nearest even; NaN keeps top payload + quiet. */
uint16_t ggo_f32_to_bf16(float f)
{
    uint32_t x = f32_bits(f);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x0040u);
    uint32_t q = x >> 16;
    uint32_t rem = x & 0xFFFFu;
    if (rem > 0x8000u || (rem == 0x8000u && (q & 1u))) ++q;
    return (uint16_t)q;
}

/* ======================================================================== *
 *  type table                                                              *
 * ======================================================================== */

/* Elements per block.  0 = not a block type this path decodes. */
size_t ggo_block_elems(uint32_t ty)
{
    switch (ty) {
    case T_F16: case T_BF16: return 1;
    case T_Q4_0: case T_Q4_1: case T_Q5_0: case T_Q5_1:
    case T_Q8_0: case T_Q8_1: case T_IQ4NL: return 32;
    case T_Q2K: case T_Q3K: case T_Q4K: case T_Q5K: case T_Q6K:
    case T_Q8K: case T_IQ4XS: return 256;
    default: return 0;
    }
}

/* Bytes per block: size_of the reference #[repr(C)] struct, except IQ4_NL /
 * IQ4_XS where the reference struct declares u16 quants (iq4nl.rs:10,
 * iq4xs.rs:14) and the on-disk ggml block is used instead (18 / 136 B). */
size_t ggo_block_bytes(uint32_t ty)
{
    switch (ty) {
    case T_F16: case T_BF16: return 2;
    case T_Q4_0: return 18;  case T_Q4_1: return 20;
    case T_Q5_0: return 22;  case T_Q5_1: return 24;
    case T_Q8_0: return 34;  case T_Q8_1: return 36;
    case T_Q2K: return 84;   case T_Q3K: return 110;
    case T_Q4K: return 144;  case T_Q5K: return 176;
    case T_Q6K: return 210;  case T_Q8K: return 290;
    case T_IQ4NL: return 18; case T_IQ4XS: return 136;
    default: return 0;
    }
}

static inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline void wr16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

/* ======================================================================== *
 *  per-block dequantize -> f32                                             *
 * ======================================================================== */

/* q4_0.rs:46-57 */
static void deq_q4_0(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b));
    const uint8_t *q = b + 2;
    for (int i = 0; i < 16; ++i) {
        y[i]      = (float)((int)(q[i] & 0xF) - 8) * d;
        y[i + 16] = (float)((int)(q[i] >> 4) - 8) * d;
    }
}

/* q4_1.rs:49-60 : x * delta + min, two roundings */
static void deq_q4_1(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b)), m = ggo_f16_to_f32(rd16(b + 2));
    const uint8_t *q = b + 4;
    for (int i = 0; i < 16; ++i) {
        float lo = (float)(q[i] & 0xF) * d;
        float hi = (float)(q[i] >> 4) * d;
        y[i] = lo + m;
        y[i + 16] = hi + m;
    }
}

/* q5_0.rs:60-73 */
static void deq_q5_0(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b));
    uint32_t qh = rd32(b + 2);
    const uint8_t *q = b + 6;
    for (int i = 0; i < 16; ++i) {
        uint8_t lo = (uint8_t)((q[i] & 0xF) | (((qh >> i) & 1u) << 4));
        uint8_t hi = (uint8_t)((q[i] >> 4) | (((qh >> (i + 16)) & 1u) << 4));
        y[i]      = (float)((int)(int8_t)lo - 16) * d;
        y[i + 16] = (float)((int)(int8_t)hi - 16) * d;
    }
}

/* q5_1.rs:64-77 */
static void deq_q5_1(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b)), m = ggo_f16_to_f32(rd16(b + 2));
    uint32_t qh = rd32(b + 4);
    const uint8_t *q = b + 8;
    for (int i = 0; i < 16; ++i) {
        uint8_t lo = (uint8_t)((q[i] & 0xF) | (((qh >> i) & 1u) << 4));
        uint8_t hi = (uint8_t)((q[i] >> 4) | (((qh >> (i + 16)) & 1u) << 4));
        float a = (float)lo * d;
        float c = (float)hi * d;
        y[i] = a + m;
        y[i + 16] = c + m;
    }
}

/* q8_0.rs:44-47 */
static void deq_q8_0(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b));
    for (int i = 0; i < 32; ++i) y[i] = (float)(int8_t)b[2 + i] * d;
}

/* q8_1.rs:58-61 : `sum` (bytes 2..3) is not used on decode */
static void deq_q8_1(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b));
    for (int i = 0; i < 32; ++i) y[i] = (float)(int8_t)b[4 + i] * d;
}

/* q8_k.rs:57-60 : reference layout, delta is f16, quants at byte 2 */
static void deq_q8_k(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b));
    for (int i = 0; i < 256; ++i) y[i] = (float)(int8_t)b[2 + i] * d;
}

/* ---- K-quants: layouts from the reference structs, arithmetic from the
 *      public ggml block specification (reference bodies are todo!()). ---- */

/* 6-bit (scale, min) pair j of 8, packed in 12 bytes (Q4_K / Q5_K). */
static inline void scale_min_k4(int j, const uint8_t *s, uint8_t *sc, uint8_t *mn)
{
    if (j < 4) {
        *sc = s[j] & 63;
        *mn = s[j + 4] & 63;
    } else {
        *sc = (uint8_t)((s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4));
        *mn = (uint8_t)((s[j + 4] >> 4) | ((s[j] >> 6) << 4));
    }
}

/* q2_k.rs:6-13 : scales[16] | qs[64] | d | dmin  (84 B) */
static void deq_q2_k(const uint8_t *b, float *y)
{
    const uint8_t *scales = b, *q = b + 16;
    float d = ggo_f16_to_f32(rd16(b + 80)), dm = ggo_f16_to_f32(rd16(b + 82));
    int is = 0, k = 0;
    for (int n = 0; n < 256; n += 128) {
        int shift = 0;
        for (int j = 0; j < 4; ++j) {
            uint8_t sc = scales[is++];
            float dl = d * (float)(sc & 0xF), ml = dm * (float)(sc >> 4);
            for (int l = 0; l < 16; ++l) {
                float t = dl * (float)((q[l] >> shift) & 3);
                y[k++] = t - ml;
            }
            sc = scales[is++];
            dl = d * (float)(sc & 0xF); ml = dm * (float)(sc >> 4);
            for (int l = 0; l < 16; ++l) {
                float t = dl * (float)((q[l + 16] >> shift) & 3);
                y[k++] = t - ml;
            }
            shift += 2;
        }
        q += 32;
    }
}

/* q3_k.rs:6-15 : hmask[32] | qs[64] | scales[12] | d  (110 B) */
static void deq_q3_k(const uint8_t *b, float *y)
{
    const uint8_t *hm = b, *q = b + 32;
    float dall = ggo_f16_to_f32(rd16(b + 108));
    const uint32_t k1 = 0x03030303u, k2 = 0x0f0f0f0fu;
    uint32_t a0 = rd32(b + 96), a1 = rd32(b + 100), t = rd32(b + 104);
    uint32_t aux[4];
    aux[2] = ((a0 >> 4) & k2) | (((t >> 4) & k1) << 4);
    aux[3] = ((a1 >> 4) & k2) | (((t >> 6) & k1) << 4);
    aux[0] = (a0 & k2) | (((t >> 0) & k1) << 4);
    aux[1] = (a1 & k2) | (((t >> 2) & k1) << 4);
    int8_t sc[16];
    for (int i = 0; i < 16; ++i) sc[i] = (int8_t)((aux[i >> 2] >> (8 * (i & 3))) & 0xFF);

    int is = 0, k = 0;
    uint8_t m = 1;
    for (int n = 0; n < 256; n += 128) {
        int shift = 0;
        for (int j = 0; j < 4; ++j) {
            float dl = dall * (float)((int)sc[is++] - 32);
            for (int l = 0; l < 16; ++l)
                y[k++] = dl * (float)((int)((q[l] >> shift) & 3) - ((hm[l] & m) ? 0 : 4));
            dl = dall * (float)((int)sc[is++] - 32);
            for (int l = 0; l < 16; ++l)
                y[k++] = dl * (float)((int)((q[l + 16] >> shift) & 3) - ((hm[l + 16] & m) ? 0 : 4));
            shift += 2;
            m = (uint8_t)(m << 1);
        }
        q += 32;
    }
}

/* q4_k.rs:6-13 : d | dmin | scales[12] | qs[128]  (144 B) */
static void deq_q4_k(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b)), dm = ggo_f16_to_f32(rd16(b + 2));
    const uint8_t *s = b + 4, *q = b + 16;
    int is = 0, k = 0;
    for (int j = 0; j < 256; j += 64) {
        uint8_t sc, mn;
        scale_min_k4(is, s, &sc, &mn);
        float d1 = d * (float)sc, m1 = dm * (float)mn;
        scale_min_k4(is + 1, s, &sc, &mn);
        float d2 = d * (float)sc, m2 = dm * (float)mn;
        for (int l = 0; l < 32; ++l) { float t = d1 * (float)(q[l] & 0xF); y[k++] = t - m1; }
        for (int l = 0; l < 32; ++l) { float t = d2 * (float)(q[l] >> 4);  y[k++] = t - m2; }
        q += 32;
        is += 2;
    }
}

/* q5_k.rs:7-18 : d | dmin | scales[12] | qh[32] | qs[128]  (176 B) */
static void deq_q5_k(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b)), dm = ggo_f16_to_f32(rd16(b + 2));
    const uint8_t *s = b + 4, *qh = b + 16, *ql = b + 48;
    int is = 0, k = 0;
    uint8_t u1 = 1, u2 = 2;
    for (int j = 0; j < 256; j += 64) {
        uint8_t sc, mn;
        scale_min_k4(is, s, &sc, &mn);
        float d1 = d * (float)sc, m1 = dm * (float)mn;
        scale_min_k4(is + 1, s, &sc, &mn);
        float d2 = d * (float)sc, m2 = dm * (float)mn;
        for (int l = 0; l < 32; ++l) {
            float t = d1 * (float)((ql[l] & 0xF) + ((qh[l] & u1) ? 16 : 0));
            y[k++] = t - m1;
        }
        for (int l = 0; l < 32; ++l) {
            float t = d2 * (float)((ql[l] >> 4) + ((qh[l] & u2) ? 16 : 0));
            y[k++] = t - m2;
        }
        ql += 32;
        is += 2;
        u1 = (uint8_t)(u1 << 2);
        u2 = (uint8_t)(u2 << 2);
    }
}

/* q6_k.rs:7-16 : ql[128] | qh[64] | scales[16] (signed) | d  (210 B) */
static void deq_q6_k(const uint8_t *b, float *y)
{
    const uint8_t *ql = b, *qh = b + 128;
    const int8_t *sc = (const int8_t *)(b + 192);
    float d = ggo_f16_to_f32(rd16(b + 208));
    for (int n = 0; n < 256; n += 128) {
        for (int l = 0; l < 32; ++l) {
            int is = l / 16;
            int q1 = (int)(int8_t)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
            int q2 = (int)(int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
            int q3 = (int)(int8_t)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
            int q4 = (int)(int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            float s1 = d * (float)sc[is + 0], s2 = d * (float)sc[is + 2];
            float s3 = d * (float)sc[is + 4], s4 = d * (float)sc[is + 6];
            y[l]      = s1 * (float)q1;
            y[l + 32] = s2 * (float)q2;
            y[l + 64] = s3 * (float)q3;
            y[l + 96] = s4 * (float)q4;
        }
        y += 128; ql += 64; qh += 32; sc += 8;
    }
}

static const int8_t KV_IQ4NL[16] = { -127, -104, -83, -65, -49, -35, -22, -10,
                                     1, 13, 25, 38, 53, 69, 89, 113 };

/* on-disk ggml IQ4_NL: d:f16 | qs[16]  (18 B); non-linear 16-entry codebook */
static void deq_iq4_nl(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b));
    const uint8_t *q = b + 2;
    for (int j = 0; j < 16; ++j) {
        y[j]      = d * (float)KV_IQ4NL[q[j] & 0xF];
        y[j + 16] = d * (float)KV_IQ4NL[q[j] >> 4];
    }
}

/* on-disk ggml IQ4_XS: d:f16 | scales_h:u16 | scales_l[4] | qs[128]  (136 B) */
static void deq_iq4_xs(const uint8_t *b, float *y)
{
    float d = ggo_f16_to_f32(rd16(b));
    uint32_t sh = rd16(b + 2);
    const uint8_t *sl = b + 4, *q = b + 8;
    for (int ib = 0; ib < 8; ++ib) {
        int ls = (int)(((sl[ib / 2] >> (4 * (ib % 2))) & 0xF) | (((sh >> (2 * ib)) & 3u) << 4));
        float dl = d * (float)(ls - 32);
        for (int j = 0; j < 16; ++j) {
            y[j]      = dl * (float)KV_IQ4NL[q[j] & 0xF];
            y[j + 16] = dl * (float)KV_IQ4NL[q[j] >> 4];
        }
        y += 32;
        q += 16;
    }
}

typedef void (*deq_fn)(const uint8_t *, float *);

static deq_fn deq_for(uint32_t ty)
{
    switch (ty) {
    case T_Q4_0: return deq_q4_0;  case T_Q4_1: return deq_q4_1;
    case T_Q5_0: return deq_q5_0;  case T_Q5_1: return deq_q5_1;
    case T_Q8_0: return deq_q8_0;  case T_Q8_1: return deq_q8_1;
    case T_Q2K: return deq_q2_k;   case T_Q3K: return deq_q3_k;
    case T_Q4K: return deq_q4_k;   case T_Q5K: return deq_q5_k;
    case T_Q6K: return deq_q6_k;   case T_Q8K: return deq_q8_k;
    case T_IQ4NL: return deq_iq4_nl; case T_IQ4XS: return deq_iq4_xs;
    default: return NULL;
    }
}

static size_t out_elem_bytes(uint32_t out_ty)
{
    return out_ty == T_F32 ? 4 : (out_ty == T_F16 || out_ty == T_BF16) ? 2 : 0;
}

/* Decode blocks [b0, b1) of `src` into `dst` (element type out_ty).
 * f16 / bf16 outputs: f32 result narrowed per element (lib.rs:70-73, 86-89). */
static void decode_range(uint32_t ty, uint32_t out_ty, void *dst, const uint8_t *src,
                         size_t b0, size_t b1)
{
    size_t n = ggo_block_elems(ty), bb = ggo_block_bytes(ty);
    if (ty == T_F16 || ty == T_BF16) {
        for (size_t i = b0; i < b1; ++i) {
            uint16_t h = rd16(src + 2 * i);
            float f = ty == T_F16 ? ggo_f16_to_f32(h) : ggo_bf16_to_f32(h);
            if (out_ty == T_F32) ((float *)dst)[i] = f;
            else if (out_ty == T_F16) ((uint16_t *)dst)[i] = ggo_f32_to_f16(f);
            else ((uint16_t *)dst)[i] = ggo_f32_to_bf16(f);
        }
        return;
    }
    deq_fn fn = deq_for(ty);
    float tmp[256];
    for (size_t i = b0; i < b1; ++i) {
        if (out_ty == T_F32) {
            fn(src + i * bb, (float *)dst + i * n);
        } else {
            fn(src + i * bb, tmp);
            uint16_t *o = (uint16_t *)dst + i * n;
            if (out_ty == T_F16) for (size_t k = 0; k < n; ++k) o[k] = ggo_f32_to_f16(tmp[k]);
            else                 for (size_t k = 0; k < n; ++k) o[k] = ggo_f32_to_bf16(tmp[k]);
        }
    }
}

struct job { uint32_t ty, out_ty; void *dst; const uint8_t *src; size_t b0, b1; };
static void *job_main(void *p)
{
    struct job *j = (struct job *)p;
    decode_range(j->ty, j->out_ty, j->dst, j->src, j->b0, j->b1);
    return NULL;
}

/* QuantExt::dequantize_slice, lib.rs:135-147.  Indivisible is tested before
 * LengthMismatch.  The reference fans blocks out over rayon's pool; here
 * `threads` workers each take one contiguous range of blocks. */
int ggo_dequantize_slice_mt(uint32_t ty, uint32_t out_ty, void *dst, size_t dst_elems,
                            const void *src, size_t src_blocks, int threads)
{
    size_t n = ggo_block_elems(ty);
    if (n == 0 || out_elem_bytes(out_ty) == 0) return GGO_UNSUPPORTED;
    if (dst_elems % n != 0) return GGO_INDIVISIBLE;
    if (src_blocks != dst_elems / n) return GGO_LENGTH_MISMATCH;
    if (threads < 1) threads = 1;
    if ((size_t)threads > src_blocks) threads = src_blocks ? (int)src_blocks : 1;
    if (threads == 1) {
        decode_range(ty, out_ty, dst, (const uint8_t *)src, 0, src_blocks);
        return GGO_OK;
    }
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    struct job *jobs = (struct job *)malloc(sizeof(struct job) * (size_t)threads);
    size_t per = src_blocks / (size_t)threads, extra = src_blocks % (size_t)threads, at = 0;
    for (int t = 0; t < threads; ++t) {
        size_t cnt = per + ((size_t)t < extra ? 1 : 0);
        jobs[t] = (struct job){ ty, out_ty, dst, (const uint8_t *)src, at, at + cnt };
        at += cnt;
        pthread_create(&tid[t], NULL, job_main, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    free(tid);
    free(jobs);
    return GGO_OK;
}

int ggo_dequantize_slice(uint32_t ty, uint32_t out_ty, void *dst, size_t dst_elems,
                         const void *src, size_t src_blocks)
{
    return ggo_dequantize_slice_mt(ty, out_ty, dst, dst_elems, src, src_blocks, 1);
}

/* ======================================================================== *
 *  per-block quantize (the formats the reference implements)               *
 * ======================================================================== */

/* Rust `as u8` / `as i8` on f32: saturating, NaN -> 0. */
static inline uint8_t sat_u8(float x)
{
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}
static inline int8_t sat_i8(float x)
{
    if (!(x == x)) return 0;
    if (x <= -128.0f) return -128;
    if (x >= 127.0f) return 127;
    return (int8_t)x;
}
static inline uint8_t min_u8(uint8_t a, uint8_t b) { return a < b ? a : b; }

/* structs.rs:91-94 : acc.max(|x|) ignores a NaN element, quiet or signalling (compiled Rust compares and
 * selects; C's fmaxf answers a signalling NaN with a NaN, which the next step would then ignore together with
 * everything folded so far -- hence the explicit comparison, acc itself is never a NaN) */
static float max_abs(const float *x, int n)
{
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) { float v = fabsf(x[i]); if (v > acc) acc = v; }
    return acc;
}
/* structs.rs:96-100 : keeps the sign; a tie keeps the earlier element */
static float max_by_abs(const float *x, int n)
{
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) if (fabsf(x[i]) > fabsf(acc)) acc = x[i];
    return acc;
}
/* f32::min / f32::max: a NaN operand is ignored.  For (+0, -0) Rust leaves the
 * result unspecified (llvm.minnum / maxnum); it is fixed here as -0 < +0, which makes
 * the fold independent of the order of the elements. */
static inline float min_f32(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a < b) return a;
    if (b < a) return b;
    return (f32_bits(a) & 0x80000000u) ? a : b;
}
static inline float max_f32(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a > b) return a;
    if (b > a) return b;
    return (f32_bits(a) & 0x80000000u) ? b : a;
}
/* structs.rs:102-107 */
static void min_max(const float *x, int n, float *mn, float *mx)
{
    float lo = 3.40282347e+38f, hi = -3.40282347e+38f;
    for (int i = 0; i < n; ++i) { lo = min_f32(lo, x[i]); hi = max_f32(hi, x[i]); }
    *mn = lo; *mx = hi;
}

/* q4_0.rs:23-44 */
static void q_q4_0(const float *x, uint8_t *b)
{
    memset(b, 0, 18);
    float max = max_by_abs(x, 32);
    if (max == 0.0f) return;
    float delta = max / -8.0f, recip = 1.0f / delta;
    wr16(b, ggo_f32_to_f16(delta));
    for (int i = 0; i < 16; ++i) {
        float tl = x[i] * recip;       tl = tl + 8.5f;
        float th = x[i + 16] * recip;  th = th + 8.5f;
        uint8_t l = sat_u8(fminf(tl, 15.0f)), h = sat_u8(fminf(th, 15.0f));
        b[2 + i] = (uint8_t)((h << 4) | l);
    }
}

/* q4_1.rs:23-47 */
static void q_q4_1(const float *x, uint8_t *b)
{
    memset(b, 0, 20);
    float mn, mx;
    min_max(x, 32, &mn, &mx);
    if (mn == mx) { wr16(b + 2, ggo_f32_to_f16(mn)); return; }
    float delta = (mx - mn) / 15.0f, recip = 1.0f / delta;
    wr16(b, ggo_f32_to_f16(delta));
    wr16(b + 2, ggo_f32_to_f16(mn));
    for (int i = 0; i < 16; ++i) {
        float tl = (x[i] - mn) * recip;       tl = tl + 0.5f;
        float th = (x[i + 16] - mn) * recip;  th = th + 0.5f;
        uint8_t l = min_u8(sat_u8(tl), 15), h = min_u8(sat_u8(th), 15);
        b[4 + i] = (uint8_t)((h << 4) | l);
    }
}

/* q5_0.rs:26-58 */
static void q_q5_0(const float *x, uint8_t *b)
{
    memset(b, 0, 22);
    float max = max_by_abs(x, 32);
    if (max == 0.0f) return;
    float delta = max / -16.0f, recip = 1.0f / delta;
    uint32_t qh = 0;
    for (int i = 0; i < 16; ++i) {
        float tl = x[i] * recip;       tl = tl + 16.5f;
        float th = x[i + 16] * recip;  th = th + 16.5f;
        uint8_t l = min_u8(sat_u8(tl), 31), h = min_u8(sat_u8(th), 31);
        qh |= (((uint32_t)l >> 4) & 1u) << i;
        qh |= (((uint32_t)h >> 4) & 1u) << (i + 16);
        b[6 + i] = (uint8_t)(((h & 0xF) << 4) | (l & 0xF));
    }
    wr16(b, ggo_f32_to_f16(delta));
    b[2] = (uint8_t)qh; b[3] = (uint8_t)(qh >> 8); b[4] = (uint8_t)(qh >> 16); b[5] = (uint8_t)(qh >> 24);
}

/* q5_1.rs:26-62 */
static void q_q5_1(const float *x, uint8_t *b)
{
    memset(b, 0, 24);
    float mn, mx;
    min_max(x, 32, &mn, &mx);
    if (mn == mx) { wr16(b + 2, ggo_f32_to_f16(mn)); return; }
    float delta = (mx - mn) / 31.0f, recip = 1.0f / delta;
    uint32_t qh = 0;
    for (int i = 0; i < 16; ++i) {
        float tl = (x[i] - mn) * recip;       tl = tl + 0.5f;
        float th = (x[i + 16] - mn) * recip;  th = th + 0.5f;
        uint8_t l = min_u8(sat_u8(tl), 31), h = min_u8(sat_u8(th), 31);
        qh |= (((uint32_t)l >> 4) & 1u) << i;
        qh |= (((uint32_t)h >> 4) & 1u) << (i + 16);
        b[8 + i] = (uint8_t)(((h & 0xF) << 4) | (l & 0xF));
    }
    wr16(b, ggo_f32_to_f16(delta));
    wr16(b + 2, ggo_f32_to_f16(mn));
    b[4] = (uint8_t)qh; b[5] = (uint8_t)(qh >> 8); b[6] = (uint8_t)(qh >> 16); b[7] = (uint8_t)(qh >> 24);
}

/* q8_0.rs:23-41 ; f32::round = half away from zero = C roundf */
static void q_q8_0(const float *x, uint8_t *b)
{
    memset(b, 0, 34);
    float amax = max_abs(x, 32);
    if (amax == 0.0f) return;
    float delta = amax / 127.0f, recip = 1.0f / delta;
    wr16(b, ggo_f32_to_f16(delta));
    for (int i = 0; i < 32; ++i) b[2 + i] = (uint8_t)sat_i8(roundf(x[i] * recip));
}

/* q8_1.rs:28-55 ; sum = f16(sum_i16 as f32 * delta) with the UNROUNDED delta */
static void q_q8_1(const float *x, uint8_t *b)
{
    memset(b, 0, 36);
    float amax = max_abs(x, 32);
    if (amax == 0.0f) return;
    float delta = amax / 127.0f, recip = 1.0f / delta;
    int16_t sum = 0;
    for (int i = 0; i < 32; ++i) {
        int8_t q = sat_i8(roundf(x[i] * recip));
        b[4 + i] = (uint8_t)q;
        sum = (int16_t)(sum + q);
    }
    wr16(b, ggo_f32_to_f16(delta));
    wr16(b + 2, ggo_f32_to_f16((float)sum * delta));
}

/* q8_k.rs:27-54 ; reference layout: delta:f16 | quants[256] | sums[16]:i16 */
static void q_q8_k(const float *x, uint8_t *b)
{
    memset(b, 0, 290);
    float max = max_by_abs(x, 256);
    if (max == 0.0f) return;
    float delta = max / -127.0f, recip = 1.0f / delta;
    int16_t sums[16] = { 0 };
    for (int i = 0; i < 256; ++i) {
        int8_t q = sat_i8(fminf(roundf(x[i] * recip), 127.0f));
        b[2 + i] = (uint8_t)q;
        sums[i / 16] = (int16_t)(sums[i / 16] + q);
    }
    wr16(b, ggo_f32_to_f16(delta));
    for (int i = 0; i < 16; ++i) wr16(b + 258 + 2 * i, (uint16_t)sums[i]);
}

typedef void (*q_fn)(const float *, uint8_t *);
static q_fn q_for(uint32_t ty)
{
    switch (ty) {
    case T_Q4_0: return q_q4_0; case T_Q4_1: return q_q4_1;
    case T_Q5_0: return q_q5_0; case T_Q5_1: return q_q5_1;
    case T_Q8_0: return q_q8_0; case T_Q8_1: return q_q8_1;
    case T_Q8K: return q_q8_k;
    default: return NULL;
    }
}

static inline float load_in(const void *src, uint32_t in_ty, size_t i)
{
    if (in_ty == T_F32) return ((const float *)src)[i];
    uint16_t h = ((const uint16_t *)src)[i];
    return in_ty == T_F16 ? ggo_f16_to_f32(h) : ggo_bf16_to_f32(h);
}

/* QuantExt::quantize_slice, lib.rs:121-133; f16/bf16 inputs are widened first
 * (lib.rs:67-69, 82-84). */
int ggo_quantize_slice(uint32_t ty, uint32_t in_ty, void *dst, size_t dst_blocks,
                       const void *src, size_t src_elems)
{
    size_t n = ggo_block_elems(ty), bb = ggo_block_bytes(ty);
    if (n == 0 || out_elem_bytes(in_ty) == 0) return GGO_UNSUPPORTED;
    if (ty != T_F16 && ty != T_BF16 && !q_for(ty)) return GGO_UNSUPPORTED;
    if (src_elems % n != 0) return GGO_INDIVISIBLE;
    if (dst_blocks != src_elems / n) return GGO_LENGTH_MISMATCH;
    uint8_t *out = (uint8_t *)dst;
    if (ty == T_F16 || ty == T_BF16) {
        for (size_t i = 0; i < src_elems; ++i) {
            float f = load_in(src, in_ty, i);
            wr16(out + 2 * i, ty == T_F16 ? ggo_f32_to_f16(f) : ggo_f32_to_bf16(f));
        }
        return GGO_OK;
    }
    q_fn fn = q_for(ty);
    float tmp[256];
    for (size_t b = 0; b < dst_blocks; ++b) {
        for (size_t k = 0; k < n; ++k) tmp[k] = load_in(src, in_ty, b * n + k);
        fn(tmp, out + b * bb);
    }
    return GGO_OK;
}

/* ======================================================================== *
 *  to-llama: scale a float tensor (to_llama.rs:99-125)                     *
 * ======================================================================== */

/* f16::from_f64 on an x86 host with F16C: through f32 (see the header). */
uint16_t ggo_f64_to_f16(double v) { return ggo_f32_to_f16((float)v); }

/* bf16::from_f64 of half 2.6.0: the crate's software routine. */
uint16_t ggo_f64_to_bf16(double value)
{
    uint64_t val;
    memcpy(&val, &value, 8);
    uint32_t x = (uint32_t)(val >> 32);      /* the last 32 bits of mantissa are truncated */
    uint32_t sign = x & 0x80000000u;
    uint32_t exp = x & 0x7FF00000u;
    uint32_t man = x & 0x000FFFFFu;
    if (exp == 0x7FF00000u) {
        uint32_t nan_bit = (man == 0 && (uint32_t)val == 0) ? 0 : 0x0040u;
        return (uint16_t)((sign >> 16) | 0x7F80u | nan_bit | (man >> 13));
    }
    uint32_t half_sign = sign >> 16;
    int64_t unbiased_exp = (int64_t)(exp >> 20) - 1023;
    int64_t half_exp = unbiased_exp + 127;
    if (half_exp >= 0xFF) return (uint16_t)(half_sign | 0x7F80u);
    if (half_exp <= 0) {
        if (7 - half_exp > 21) return (uint16_t)half_sign;
        man |= 0x00100000u;
        uint32_t half_man = man >> (14 - half_exp);
        uint32_t round_bit = 1u << (13 - half_exp);
        if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) half_man += 1;
        return (uint16_t)(half_sign | half_man);
    }
    uint32_t he = (uint32_t)half_exp << 7;
    uint32_t half_man = man >> 13;
    uint32_t round_bit = 0x00001000u;
    if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0)
        return (uint16_t)((half_sign | he | half_man) + 1);
    return (uint16_t)(half_sign | he | half_man);
}

/* scale_tensor + scale_data: `*x *= scale` in the tensor's type.  For f16 / bf16 the crate's
 * Mul is from_f32(a.to_f32() * b.to_f32()). */
int ggo_scale(uint32_t ty, void *dst, const void *src, size_t n, double scale)
{
    enum { T_F64 = 28 };
    if (ty == T_F64) {
        const double *s = (const double *)src; double *d = (double *)dst;
        for (size_t i = 0; i < n; ++i) d[i] = s[i] * scale;
    } else if (ty == T_F32) {
        const float f = (float)scale;
        const float *s = (const float *)src; float *d = (float *)dst;
        for (size_t i = 0; i < n; ++i) d[i] = s[i] * f;
    } else if (ty == T_F16) {
        const float f = ggo_f16_to_f32(ggo_f64_to_f16(scale));
        const uint16_t *s = (const uint16_t *)src; uint16_t *d = (uint16_t *)dst;
        for (size_t i = 0; i < n; ++i) d[i] = ggo_f32_to_f16(ggo_f16_to_f32(s[i]) * f);
    } else if (ty == T_BF16) {
        const float f = ggo_bf16_to_f32(ggo_f64_to_bf16(scale));
        const uint16_t *s = (const uint16_t *)src; uint16_t *d = (uint16_t *)dst;
        for (size_t i = 0; i < n; ++i) d[i] = ggo_f32_to_bf16(ggo_bf16_to_f32(s[i]) * f);
    } else {
        return GGO_UNSUPPORTED;      /* todo!() in the reference */
    }
    return GGO_OK;
}
