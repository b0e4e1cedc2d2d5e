// formats.h -- per-format block decoders for gfx950, used by every kernel.
//
// Each format F provides
//     F::N      elements per block        F::BYTES  bytes per block
//     F::dec4(r, b, e, y)   decode elements [e, e+4) (e % 4 == 0) of the block
//                           that starts at byte offset b of byte-reader r.
// A group of 4 (or 8, when aligned to 8) consecutive elements never straddles a
// sub-block in any ggml format (sub-blocks are 16 or 32 elements), so one group
// needs one scale.
//
// Arithmetic follows the reference's Rust exactly where it exists, and the
// public ggml block specification where the reference body is `todo!()`.
// Every float operation is a separately rounded IEEE f32 operation
// (__fmul_rn / __fadd_rn / __fsub_rn are never contracted into an FMA), which
// is what the Rust source compiles to.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace ggq {

// ---- byte reader over a dword-aligned LDS image -----------------------------
// `w` points at dwords; byte offsets may have any alignment.  u32() reads the
// two aligned dwords that contain the 4 bytes and funnel-shifts; the image must
// therefore own one dword of slack after the last byte that is read.
struct LdsBytes {
    const uint32_t *w;
    __device__ __forceinline__ uint32_t u32(uint32_t off) const {
        uint32_t i = off >> 2;
        uint32_t lo = w[i], hi = w[i + 1];
        return __builtin_amdgcn_alignbit(hi, lo, (off & 3u) * 8u);
    }
    // G consecutive little-endian words starting at byte `off` (G + 1 aligned reads, G funnel shifts)
    template <int G>
    __device__ __forceinline__ void words(uint32_t off, uint32_t (&out)[G]) const {
        const uint32_t i = off >> 2, sh = (off & 3u) * 8u;
        uint32_t lo = w[i];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t hi = w[i + 1 + g];
            out[g] = __builtin_amdgcn_alignbit(hi, lo, sh);
            lo = hi;
        }
    }
    __device__ __forceinline__ uint32_t u16(uint32_t off) const { return u32(off) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t u8(uint32_t off) const {
        return (w[off >> 2] >> ((off & 3u) * 8u)) & 0xFFu;
    }
};

// ---- small helpers ----------------------------------------------------------
__device__ __forceinline__ float h2f(uint32_t bits16) {
    return __half2float(__ushort_as_half((unsigned short)bits16));
}
// byte k of w as float (v_cvt_f32_ubyteK)
template <int K> __device__ __forceinline__ float ub(uint32_t w) {
    return (float)((w >> (8 * K)) & 0xFFu);
}
// Hides how w was computed.  Without it the compiler folds the nibble mask into every byte
// extraction (one shift-and-mask plus one conversion per element); with it the four conversions
// read the bytes of the masked word directly.
__device__ __forceinline__ uint32_t opaque(uint32_t w) {
    asm("" : "+v"(w));
    return w;
}
// signed byte k of w as float
template <int K> __device__ __forceinline__ float sb(uint32_t w) {
    return (float)(int)(int8_t)((w >> (8 * K)) & 0xFFu);
}
// bits 0..3 of x -> bit 0 of bytes 0..3
__device__ __forceinline__ uint32_t spread4(uint32_t x) {
    return ((x & 0xFu) * 0x00204081u) & 0x01010101u;
}

// y[4g .. 4g+3] = EXPR for bytes K = 0..3 of w
#define GGQ_FOR4(EXPR)                                                         \
    do {                                                                       \
        w = opaque(w);                                                         \
        { constexpr int K = 0; y[4 * g + 0] = (EXPR); }                        \
        { constexpr int K = 1; y[4 * g + 1] = (EXPR); }                        \
        { constexpr int K = 2; y[4 * g + 2] = (EXPR); }                        \
        { constexpr int K = 3; y[4 * g + 3] = (EXPR); }                        \
    } while (0)

// A format's decode is split in two so that a thread that handles 8 consecutive
// elements (16-bit outputs) extracts the sub-block scale once and fetches its two
// quant words with one funnel read:
//     F::scale(r, b, e)           the multipliers of the sub-block that holds element e
//     F::vals<G>(r, b, e, s, y)   elements [e, e + 4G) given those, G = 1 or 2, e % 4G == 0
//     F::dec4(r, b, e, y)         = vals<1>(..., scale(...), ...)
// In every format the quant bytes of elements e+4..e+7 directly follow those of e..e+3
// and use the same shifts, which is what lets G = 2 share everything but the word.
struct Scale {
    float a, c;   // per format: (delta, min) or (sub-block delta, sub-block min)
};
#define GGQ_DEC4                                                                                       \
    template <class R>                                                                                 \
    static __device__ __forceinline__ void dec4(const R &r, uint32_t b, uint32_t e, float (&y)[4]) {   \
        vals<1>(r, b, e, scale(r, b, e), y);                                                           \
    }

// ============================ 32-element formats ============================

// ggml-quants/src/structs/q4_0.rs:46-57   delta:f16 | quants[16]
struct Q4_0 {
    static constexpr uint32_t N = 32, BYTES = 18;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        return {h2f(r.u16(b)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t s = e >> 2, q[G];
        r.template words<G>(b + 2 + 4 * (s & 3), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (4 * (s >> 2))) & 0x0F0F0F0Fu;
            GGQ_FOR4(__fmul_rn(__fsub_rn(ub<K>(w), 8.0f), sc.a));
        }
    }
    GGQ_DEC4
};

// ggml-quants/src/structs/q4_1.rs:49-60   delta:f16 | min:f16 | quants[16]
struct Q4_1 {
    static constexpr uint32_t N = 32, BYTES = 20;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        uint32_t dm = r.u32(b);
        return {h2f(dm & 0xFFFFu), h2f(dm >> 16)};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t s = e >> 2, q[G];
        r.template words<G>(b + 4 + 4 * (s & 3), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (4 * (s >> 2))) & 0x0F0F0F0Fu;
            GGQ_FOR4(__fadd_rn(__fmul_rn(ub<K>(w), sc.a), sc.c));
        }
    }
    GGQ_DEC4
};

// ggml-quants/src/structs/q5_0.rs:60-73   delta:f16 | qh[4] | ql[16]
// element i uses bit i of qh as its 5th bit, for the low half (i < 16, low
// nibbles) and the high half (i >= 16, high nibbles) alike.
struct Q5_0 {
    static constexpr uint32_t N = 32, BYTES = 22;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        return {h2f(r.u16(b)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t qh = r.u32(b + 2) >> e;
        uint32_t s = e >> 2, q[G];
        r.template words<G>(b + 6 + 4 * (s & 3), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (4 * (s >> 2))) & 0x0F0F0F0Fu;
            w |= spread4(qh >> (4 * g)) << 4;
            GGQ_FOR4(__fmul_rn(__fsub_rn(ub<K>(w), 16.0f), sc.a));
        }
    }
    GGQ_DEC4
};

// ggml-quants/src/structs/q5_1.rs:64-77   delta:f16 | min:f16 | qh[4] | ql[16]
struct Q5_1 {
    static constexpr uint32_t N = 32, BYTES = 24;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        uint32_t dm = r.u32(b);
        return {h2f(dm & 0xFFFFu), h2f(dm >> 16)};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t qh = r.u32(b + 4) >> e;
        uint32_t s = e >> 2, q[G];
        r.template words<G>(b + 8 + 4 * (s & 3), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (4 * (s >> 2))) & 0x0F0F0F0Fu;
            w |= spread4(qh >> (4 * g)) << 4;
            GGQ_FOR4(__fadd_rn(__fmul_rn(ub<K>(w), sc.a), sc.c));
        }
    }
    GGQ_DEC4
};

// ggml-quants/src/structs/q8_0.rs:44-47   delta:f16 | quants[32]:i8
struct Q8_0 {
    static constexpr uint32_t N = 32, BYTES = 34;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        return {h2f(r.u16(b)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t q[G];
        r.template words<G>(b + 2 + e, q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = q[g];
            GGQ_FOR4(__fmul_rn(sb<K>(w), sc.a));
        }
    }
    GGQ_DEC4
};

// ggml-quants/src/structs/q8_1.rs:58-61   delta:f16 | sum:f16 (unused) | quants[32]:i8
struct Q8_1 {
    static constexpr uint32_t N = 32, BYTES = 36;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        return {h2f(r.u16(b)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t q[G];
        r.template words<G>(b + 4 + e, q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = q[g];
            GGQ_FOR4(__fmul_rn(sb<K>(w), sc.a));
        }
    }
    GGQ_DEC4
};

// 16-entry non-linear codebook of IQ4_NL / IQ4_XS, stored biased by +128 so a
// byte permute can fetch four entries at once; the bias comes off exactly in
// float.  kv = {-127,-104,-83,-65,-49,-35,-22,-10,1,13,25,38,53,69,89,113}.
__device__ __forceinline__ uint32_t iq4_lookup4(uint32_t nibs) {
    const uint32_t T0 = 0x3F2D1801u, T1 = 0x766A5D4Fu, T2 = 0xA6998D81u, T3 = 0xF1D9C5B5u;
    uint32_t sel = nibs & 0x07070707u;
    uint32_t lo = __builtin_amdgcn_perm(T1, T0, sel);
    uint32_t hi = __builtin_amdgcn_perm(T3, T2, sel);
    uint32_t mask = ((nibs >> 3) & 0x01010101u) * 0xFFu;
    return (hi & mask) | (lo & ~mask);
}

// on-disk ggml IQ4_NL   d:f16 | qs[16]   (reference struct iq4nl.rs:6-11 is `todo!()`)
struct IQ4NL {
    static constexpr uint32_t N = 32, BYTES = 18;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        return {h2f(r.u16(b)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t s = e >> 2, q[G];
        r.template words<G>(b + 2 + 4 * (s & 3), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = iq4_lookup4((q[g] >> (4 * (s >> 2))) & 0x0F0F0F0Fu);
            GGQ_FOR4(__fmul_rn(sc.a, __fsub_rn(ub<K>(w), 128.0f)));
        }
    }
    GGQ_DEC4
};

// ============================ 256-element formats ===========================

// ggml-quants/src/structs/q8_k.rs:57-60   delta:f16 | quants[256]:i8 | sums[16]:i16
// (the reference's own layout: 290 bytes, f16 scale)
struct Q8K {
    static constexpr uint32_t N = 256, BYTES = 290;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t) {
        return {h2f(r.u16(b)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t q[G];
        r.template words<G>(b + 2 + e, q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = q[g];
            GGQ_FOR4(__fmul_rn(sb<K>(w), sc.a));
        }
    }
    GGQ_DEC4
};

// layout ggml-quants/src/structs/q2_k.rs:6-13   scales[16] | qs[64] | d:f16 | dmin:f16
// y = d*(sc&15)*q - dmin*(sc>>4), q = 2-bit field, sc = scales[e/16]
struct Q2K {
    static constexpr uint32_t N = 256, BYTES = 84;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t e) {
        uint32_t dm = r.u32(b + 80);
        float d = h2f(dm & 0xFFFFu), mn = h2f(dm >> 16);
        uint32_t sc = r.u8(b + (e >> 4));
        return {__fmul_rn(d, (float)(sc & 0xFu)), __fmul_rn(mn, (float)(sc >> 4))};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t q[G];
        r.template words<G>(b + 16 + 32 * (e >> 7) + (e & 31), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (2 * ((e >> 5) & 3))) & 0x03030303u;
            GGQ_FOR4(__fsub_rn(__fmul_rn(sc.a, ub<K>(w)), sc.c));
        }
    }
    GGQ_DEC4
};

// layout ggml-quants/src/structs/q3_k.rs:6-15   hmask[32] | qs[64] | scales[12] | d:f16
// y = d*(sc-32)*(q - (hbit ? 0 : 4)), 6-bit sc packed 4+2 bits over 12 bytes
struct Q3K {
    static constexpr uint32_t N = 256, BYTES = 110;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t e) {
        float dall = h2f(r.u16(b + 108));
        uint32_t is = e >> 4, i = is & 3, g = is >> 2;
        uint32_t lo4 = (r.u8(b + 96 + 4 * (g & 1) + i) >> (4 * (g >> 1))) & 0xFu;
        uint32_t hi2 = (r.u8(b + 104 + i) >> (2 * g)) & 3u;
        return {__fmul_rn(dall, (float)((int)(lo4 | (hi2 << 4)) - 32)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t hbit = 4 * (e >> 7) + ((e >> 5) & 3);
        uint32_t h[G], q[G];
        r.template words<G>(b + (e & 31), h);
        r.template words<G>(b + 32 + 32 * (e >> 7) + (e & 31), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (2 * ((e >> 5) & 3))) & 0x03030303u;
            w |= ((h[g] >> hbit) & 0x01010101u) << 2;   // q + 4*hbit, so the value is (w - 4)
            GGQ_FOR4(__fmul_rn(sc.a, __fsub_rn(ub<K>(w), 4.0f)));
        }
    }
    GGQ_DEC4
};

// 6-bit (scale, min) pair `is` of 8 from the 12 packed bytes at s (Q4_K, Q5_K)
// Branch-free: with j = is & 3, the three bytes A = s[j], B = s[j+4], C = s[j+8]
// hold everything either case needs (is < 4: sc = A & 63, mn = B & 63;
// is >= 4: sc = (C & 15) | (A >> 6) << 4, mn = (C >> 4) | (B >> 6) << 4), so lanes
// of one wave that sit in different sub-blocks never diverge.
template <class R>
__device__ __forceinline__ void scale_min_k4(const R &r, uint32_t s, uint32_t is, uint32_t &sc, uint32_t &mn) {
    uint32_t j = is & 3u;
    uint32_t A = r.u8(s + j), B = r.u8(s + j + 4), C = r.u8(s + j + 8);
    bool low = is < 4;
    sc = low ? (A & 63u) : ((C & 0xFu) | ((A >> 6) << 4));
    mn = low ? (B & 63u) : ((C >> 4) | ((B >> 6) << 4));
}
template <class R>
__device__ __forceinline__ Scale scale_k4(const R &r, uint32_t b, uint32_t e) {
    uint32_t dm = r.u32(b);
    float d = h2f(dm & 0xFFFFu), dmin = h2f(dm >> 16);
    uint32_t sc, mn;
    scale_min_k4(r, b + 4, e >> 5, sc, mn);
    return {__fmul_rn(d, (float)sc), __fmul_rn(dmin, (float)mn)};
}

// layout ggml-quants/src/structs/q4_k.rs:6-13   d:f16 | dmin:f16 | scales[12] | qs[128]
// y = d*sc*q - dmin*mn, sub-block is = e/32, nibble plane alternates every 32
struct Q4K {
    static constexpr uint32_t N = 256, BYTES = 144;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t e) { return scale_k4(r, b, e); }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t q[G];
        r.template words<G>(b + 16 + 32 * (e >> 6) + (e & 31), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (4 * ((e >> 5) & 1))) & 0x0F0F0F0Fu;
            GGQ_FOR4(__fsub_rn(__fmul_rn(sc.a, ub<K>(w)), sc.c));
        }
    }
    GGQ_DEC4
};

// layout ggml-quants/src/structs/q5_k.rs:7-18   d | dmin | scales[12] | qh[32] | qs[128]
// as Q4_K with bit `is` of qh[l] as the 5th bit
struct Q5K {
    static constexpr uint32_t N = 256, BYTES = 176;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t e) { return scale_k4(r, b, e); }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t is = e >> 5, h[G], q[G];
        r.template words<G>(b + 16 + (e & 31), h);
        r.template words<G>(b + 48 + 32 * (e >> 6) + (e & 31), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = (q[g] >> (4 * (is & 1))) & 0x0F0F0F0Fu;
            w |= ((h[g] >> is) & 0x01010101u) << 4;
            GGQ_FOR4(__fsub_rn(__fmul_rn(sc.a, ub<K>(w)), sc.c));
        }
    }
    GGQ_DEC4
};

// layout ggml-quants/src/structs/q6_k.rs:7-16   ql[128] | qh[64] | scales[16]:i8 | d:f16
// y = (d*sc) * (q - 32); per 128-element half, quadrant qd = (e%128)/32 picks
// the nibble plane of ql and the 2-bit field of qh
struct Q6K {
    static constexpr uint32_t N = 256, BYTES = 210;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t e) {
        float d = h2f(r.u16(b + 208));
        uint32_t n = e >> 7, qd = (e >> 5) & 3, l0 = e & 31;
        int sc = (int)(int8_t)r.u8(b + 192 + 8 * n + (l0 >> 4) + 2 * qd);
        return {__fmul_rn(d, (float)sc), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t n = e >> 7, qd = (e >> 5) & 3, l0 = e & 31;
        uint32_t ql[G], qh[G];
        r.template words<G>(b + 64 * n + 32 * (qd & 1) + l0, ql);
        r.template words<G>(b + 128 + 32 * n + l0, qh);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = ((ql[g] >> (4 * (qd >> 1))) & 0x0F0F0F0Fu) | (((qh[g] >> (2 * qd)) & 0x03030303u) << 4);
            GGQ_FOR4(__fmul_rn(sc.a, __fsub_rn(ub<K>(w), 32.0f)));
        }
    }
    GGQ_DEC4
};

// on-disk ggml IQ4_XS   d:f16 | scales_h:u16 | scales_l[4] | qs[128]
// (reference struct iq4xs.rs:6-15 is `todo!()`)
struct IQ4XS {
    static constexpr uint32_t N = 256, BYTES = 136;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &r, uint32_t b, uint32_t e) {
        uint32_t hdr = r.u32(b);
        float d = h2f(hdr & 0xFFFFu);
        uint32_t ib = e >> 5;
        uint32_t ls = ((r.u8(b + 4 + (ib >> 1)) >> (4 * (ib & 1))) & 0xFu) | ((((hdr >> 16) >> (2 * ib)) & 3u) << 4);
        return {__fmul_rn(d, (float)((int)ls - 32)), 0.0f};
    }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale sc, float (&y)[4 * G]) {
        uint32_t q[G];
        r.template words<G>(b + 8 + 16 * (e >> 5) + (e & 15), q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w = iq4_lookup4((q[g] >> (4 * ((e >> 4) & 1))) & 0x0F0F0F0Fu);
            GGQ_FOR4(__fmul_rn(sc.a, __fsub_rn(ub<K>(w), 128.0f)));
        }
    }
    GGQ_DEC4
};

#undef GGQ_DEC4
#undef GGQ_FOR4

}  // namespace ggq
