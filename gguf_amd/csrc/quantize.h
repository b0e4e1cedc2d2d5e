// quantize.h -- block quantizers for gfx950: the formats the reference implements
// (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1, Q8K) and the f16 / bf16 narrowing, bit-exact
// with the reference's Rust (QuantExt::quantize_slice, ggml-quants/src/lib.rs:121-133).
//
// Shape: the mirror image of the decode kernels.  One-shot grid, one tile per
// 512-thread workgroup, two non-temporal 16-byte loads per thread (4096 f32 or 8192
// f16/bf16 elements); measured against one load (-6 %), four (-12 %), plain loads
// (-2 %) and plain stores (-6 %) per thread.  A block's elements sit in G = N / EPL consecutive lanes of one wave;
// the per-block reductions (max by magnitude, min/max, sums, the 5th-bit mask) run over
// those lanes with xor shuffles; packed blocks are assembled in LDS and leave as
// contiguous 16-byte stores.  Reads dominate here (4 bytes in, 0.56-1.13 out per element).
//
// Order sensitivity.  The reference folds a block's elements first to last:
//   max_by_abs (structs.rs:96-100) keeps the FIRST element of maximal magnitude and its
//   sign -- the shuffle reduction carries the lane order and resolves ties to the
//   earlier lane; NaN elements never win (x.abs() > acc.abs() is false for NaN);
//   min_max (structs.rs:102-107) ignores NaN; for (+0, -0) Rust leaves min/max
//   unspecified, fixed here as -0 < +0.
#pragma once
#include <type_traits>

#include "kernels.h"

namespace ggq {

// ---- input element types ------------------------------------------------------
// The input is read exactly once: non-temporal loads (measured 7.0-7.2 TB/s for a pure read
// stream against 6.1-6.9 with plain loads).
template <bool NT>
__device__ __forceinline__ uint4v load16(const void *p) {
    return NT ? __builtin_nontemporal_load((const uint4v *)p) : *(const uint4v *)p;
}
struct InF32 {
    typedef float T;
    static constexpr uint32_t EPL = 4;
    template <bool NT = false>
    static __device__ __forceinline__ void load(const T *p, float (&x)[4]) {
        uint4v v = load16<NT>(p);
        x[0] = __uint_as_float(v.x); x[1] = __uint_as_float(v.y); x[2] = __uint_as_float(v.z); x[3] = __uint_as_float(v.w);
    }
};
// (8 consecutive f32 per lane, two adjacent 16-byte loads, so that a block spans 4 lanes and the
// per-block divisions are shared by more elements, measured 2-5 % SLOWER: the 32-byte lane stride
// costs more at the memory than the saved instructions give back.)
// f16 / bf16 inputs are widened first (lib.rs:67-69, 82-84)
struct InF16 {
    typedef uint16_t T;
    static constexpr uint32_t EPL = 8;
    template <bool NT = false>
    static __device__ __forceinline__ void load(const T *p, float (&x)[8]) {
        uint4v v = load16<NT>(p);
        x[0] = h2f(v.x & 0xFFFFu); x[1] = h2f(v.x >> 16); x[2] = h2f(v.y & 0xFFFFu); x[3] = h2f(v.y >> 16);
        x[4] = h2f(v.z & 0xFFFFu); x[5] = h2f(v.z >> 16); x[6] = h2f(v.w & 0xFFFFu); x[7] = h2f(v.w >> 16);
    }
};
struct InBF16 {
    typedef uint16_t T;
    static constexpr uint32_t EPL = 8;
    template <bool NT = false>
    static __device__ __forceinline__ void load(const T *p, float (&x)[8]) {
        uint4v v = load16<NT>(p);
        x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xFFFF0000u);
        x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xFFFF0000u);
        x[4] = __uint_as_float(v.z << 16); x[5] = __uint_as_float(v.z & 0xFFFF0000u);
        x[6] = __uint_as_float(v.w << 16); x[7] = __uint_as_float(v.w & 0xFFFF0000u);
    }
};

// ---- scalar helpers with the reference's semantics -------------------------------
__device__ __forceinline__ bool is_nan(float x) { return x != x; }
// Rust `as u8` / `as i8` on f32: saturating, NaN -> 0.  The hardware conversions behave exactly
// so (out-of-range saturates, NaN gives 0); they are issued as asm because the same conversion
// written in C is undefined for those inputs.
__device__ __forceinline__ int cvt_i32_sat(float x) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t cvt_u32_sat(float x) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int clamp_i8(int v) { return max(-128, min(v, 127)); }
__device__ __forceinline__ int sat_i8(float x) { return clamp_i8(cvt_i32_sat(x)); }
// f32::round (half away from zero) is trunc(x + copysign(pred(0.5), x)) for every float (checked exhaustively for
// |x| < 512 on the host; beyond that both sides are far outside the i8 range that follows); a NaN stays a NaN.
// `f32::round(x) as i32` in one go: the conversion truncates toward zero by itself, so the truncation of round_away
// is not issued separately (one instruction less per element; the quantizers from 16-bit input are VALU-bound).
__device__ __forceinline__ int round_away_i32(float x) { return cvt_i32_sat(__fadd_rn(x, copysignf(0.49999997f, x))); }
// (f32::min(x, c) for a non-NaN constant c gives c for a NaN x: that is fminf / v_min_f32.)
__device__ __forceinline__ uint32_t f2h_bits(float x) {
    asm volatile("" : "+v"(x));   // keep the producer from being folded into the conversion
    return (uint32_t)__half_as_ushort(__float2half_rn(x));
}
// ---- reductions over the G lanes of a block (G a power of two, lanes consecutive) --
// All of them are butterflies in which, before the step at distance M, the 2^k lanes of each
// half already agree; the step only needs ANY lane of the other half.  For M = 1, 2, 4, 8 that is
// a DPP lane pattern inside a row of 16 (quad permutes, then the mirrors), which costs a VALU
// modifier instead of a trip through the LDS crossbar; wider steps use the generic shuffle.
template <uint32_t CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <uint32_t M>
__device__ __forceinline__ uint32_t other_half(uint32_t v) {
    if constexpr (M == 1) return dpp_u32<0xB1>(v);          // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_u32<0x4E>(v);     // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return dpp_u32<0x141>(v);    // row_half_mirror
    else if constexpr (M == 8) return dpp_u32<0x140>(v);    // row_mirror
    else return (uint32_t)__shfl_xor((int)v, (int)M);
}
template <uint32_t M>
__device__ __forceinline__ float other_half(float v) { return __uint_as_float(other_half<M>(__float_as_uint(v))); }
template <uint32_t M>
__device__ __forceinline__ int other_half(int v) { return (int)other_half<M>((uint32_t)v); }
// f(M) for M = 1, 2, 4, ... < G
template <uint32_t G, uint32_t M = 1, class Fn>
__device__ __forceinline__ void butterfly(Fn &&f) {
    if constexpr (M < G) {
        f(std::integral_constant<uint32_t, M>());
        butterfly<G, M * 2>(f);
    }
}
template <uint32_t G>
__device__ __forceinline__ float group_max_by_abs(const float *x, uint32_t n) {
    // local fold, first to last; NaN never selected (the fold starts from +0.0)
    float best = 0.0f;
    for (uint32_t k = 0; k < n; ++k)
        if (fabsf(x[k]) > fabsf(best)) best = x[k];
    const uint32_t lane = threadIdx.x & 63u;
    butterfly<G>([&](auto m) {
        float other = other_half<m.value>(best);
        // the half with the lower lane numbers holds earlier elements
        float early = (lane & m.value) ? other : best, late = (lane & m.value) ? best : other;
        best = fabsf(late) > fabsf(early) ? late : early;
    });
    return best;
}
template <uint32_t G>
__device__ __forceinline__ float group_max_abs(const float *x, uint32_t n) {
    // acc.max(|x|): f32::max ignores a NaN operand, which is what v_max_f32 does (the compiler quiets a possible
    // signalling NaN first where the value comes straight from memory); one instruction per element
    float m = 0.0f;
    for (uint32_t k = 0; k < n; ++k) m = fmaxf(m, fabsf(x[k]));
    butterfly<G>([&](auto s) { m = fmaxf(m, other_half<s.value>(m)); });
    return m;
}
// min and max with NaN ignored, folded from (f32::MAX, f32::MIN) (structs.rs:101-107): f32::min / f32::max return
// the other operand when one is a NaN, which is v_min_f32 / v_max_f32 (minNum / maxNum; the compiler quiets a
// possible signalling NaN first).  For (+0, -0) Rust leaves the result open; the hardware orders -0 below +0, and so
// do the parity tests.  Three instructions per element where integer sort keys took eight.
template <uint32_t G>
__device__ __forceinline__ void group_min_max(const float *x, uint32_t n, float &lo, float &hi) {
    lo = 3.40282347e+38f;
    hi = -3.40282347e+38f;
    for (uint32_t k = 0; k < n; ++k) {
        lo = fminf(lo, x[k]);
        hi = fmaxf(hi, x[k]);
    }
    butterfly<G>([&](auto s) {
        lo = fminf(lo, other_half<s.value>(lo));
        hi = fmaxf(hi, other_half<s.value>(hi));
    });
}
template <uint32_t G>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
    butterfly<G>([&](auto s) { v |= other_half<s.value>(v); });
    return v;
}
template <uint32_t G>
__device__ __forceinline__ int group_add(int v) {
    butterfly<G>([&](auto s) { v += other_half<s.value>(v); });
    return v;
}

// LDS stores at 2-byte alignment
__device__ __forceinline__ void st16(uint8_t *p, uint32_t v) { *(uint16_t *)p = (uint16_t)v; }
__device__ __forceinline__ void st32_a2(uint8_t *p, uint32_t v) { st16(p, v); st16(p + 2, v >> 16); }

// nibble formats: EPL codes q[k] of elements e0..e0+EPL-1 (e0 = gl * EPL).  Byte i of the
// block's 16 payload bytes = code(i) | code(i + 16) << 4: the low-half lanes fetch the codes
// of the lane G/2 above and write EPL bytes each.
template <uint32_t EPL, uint32_t G>
__device__ __forceinline__ void pack_nibbles(const uint32_t *q, uint32_t gl, uint8_t *payload, bool write) {
    constexpr uint32_t W = EPL / 4;
    uint32_t w[W];
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) {
        // low bytes of four registers into one word (three byte permutes), then the low nibbles of all four at once
        const uint32_t lo = __builtin_amdgcn_perm(q[4 * j + 1], q[4 * j], 0x0c0c0400u);
        const uint32_t hi = __builtin_amdgcn_perm(q[4 * j + 3], q[4 * j + 2], 0x0c0c0400u);
        w[j] = __builtin_amdgcn_perm(hi, lo, 0x05040100u) & 0x0F0F0F0Fu;
    }
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) {
        // the word of the lane G/2 above (only the lower half of the group uses it)
        uint32_t hi;
        if constexpr (G == 4) hi = dpp_u32<0x4E>(w[j]);          // quad_perm [2,3,0,1]
        else if constexpr (G == 8) hi = dpp_u32<0x104>(w[j]);    // row_shl:4
        else hi = (uint32_t)__shfl_xor((int)w[j], (int)(G / 2));
        if (write && gl < G / 2) st32_a2(payload + gl * EPL + 4 * j, w[j] | (hi << 4));
    }
}

// ================================ the formats =====================================
// Each encode() is called by all lanes of a wave; `blk` = this lane's block in LDS,
// gl = lane within the block's group, x = this lane's EPL elements, write = block exists.

// ggml-quants/src/structs/q4_0.rs:23-44
struct EncQ4_0 {
    static constexpr uint32_t N = 32, BYTES = 18;
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float max = group_max_by_abs<G>(x, EPL);
        uint32_t q[EPL];
        uint32_t hdr = 0;
        if (max == 0.0f) {
            for (uint32_t k = 0; k < EPL; ++k) q[k] = 0;
        } else {
            float delta = __fdiv_rn(max, -8.0f), recip = __fdiv_rn(1.0f, delta);
            hdr = f2h_bits(delta);
            // `(y + 8.5).min(15.) as u8`: the float minimum (a NaN gives 15) leaves nothing for the cast to clamp from above
            for (uint32_t k = 0; k < EPL; ++k) q[k] = cvt_u32_sat(fminf(__fadd_rn(__fmul_rn(x[k], recip), 8.5f), 15.0f));
        }
        if (write && gl == 0) st16(blk, hdr);
        pack_nibbles<EPL, G>(q, gl, blk + 2, write);
    }
};

// ggml-quants/src/structs/q4_1.rs:23-47
struct EncQ4_1 {
    static constexpr uint32_t N = 32, BYTES = 20;
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float mn, mx;
        group_min_max<G>(x, EPL, mn, mx);
        uint32_t q[EPL];
        // No branch for the reference's `min == max` case (header (0, min), all codes 0): with delta = 0 there,
        // recip = inf, every element gives (x - min) * inf = 0 * inf = NaN (or NaN from a NaN or infinite element),
        // a NaN converts to 0, and f16(0) | f16(min) << 16 is the header -- the general path produces that block.
        // (delta is selected, not computed, in that case: for min = max = inf the difference would be NaN.)
        const float delta = mn == mx ? 0.0f : __fdiv_rn(__fsub_rn(mx, mn), 15.0f), recip = __fdiv_rn(1.0f, delta);
        const uint32_t hdr = f2h_bits(delta) | (f2h_bits(mn) << 16);
        // `as u8` saturates at 255, `.min(15)` follows: one integer minimum does both
        for (uint32_t k = 0; k < EPL; ++k)
            q[k] = min(cvt_u32_sat(__fadd_rn(__fmul_rn(__fsub_rn(x[k], mn), recip), 0.5f)), 15u);
        if (write && gl == 0) *(uint32_t *)blk = hdr;
        pack_nibbles<EPL, G>(q, gl, blk + 4, write);
    }
};

// ggml-quants/src/structs/q5_0.rs:26-58
struct EncQ5_0 {
    static constexpr uint32_t N = 32, BYTES = 22;
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float max = group_max_by_abs<G>(x, EPL);
        uint32_t q[EPL];
        uint32_t bits = 0;
        // No branch for `max == 0` (an all-zero block): there delta = -0 and recip = -inf, every element is +-0 or
        // NaN, so x * recip is NaN, which converts to code 0; only the header needs the case (f16(-0) is not 0).
        const float delta = __fdiv_rn(max, -16.0f), recip = __fdiv_rn(1.0f, delta);
        const uint32_t hdr = max == 0.0f ? 0u : f2h_bits(delta);
        for (uint32_t k = 0; k < EPL; ++k) {
            q[k] = min(cvt_u32_sat(__fadd_rn(__fmul_rn(x[k], recip), 16.5f)), 31u);
            bits |= (q[k] >> 4) << k;                       // this lane's fifth bits, at 0 .. EPL-1 for now
        }
        uint32_t qh = group_or<G>(bits << (gl * EPL));       // element e owns bit e of qh
        if (write && gl == 0) { st16(blk, hdr); st32_a2(blk + 2, qh); }
        pack_nibbles<EPL, G>(q, gl, blk + 6, write);
    }
};

// ggml-quants/src/structs/q5_1.rs:26-62
struct EncQ5_1 {
    static constexpr uint32_t N = 32, BYTES = 24;
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float mn, mx;
        group_min_max<G>(x, EPL, mn, mx);
        uint32_t q[EPL];
        uint32_t bits = 0;
        // no branch for `min == max`, for the reason given at EncQ4_1
        const float delta = mn == mx ? 0.0f : __fdiv_rn(__fsub_rn(mx, mn), 31.0f), recip = __fdiv_rn(1.0f, delta);
        const uint32_t hdr = f2h_bits(delta) | (f2h_bits(mn) << 16);
        for (uint32_t k = 0; k < EPL; ++k) {
            q[k] = min(cvt_u32_sat(__fadd_rn(__fmul_rn(__fsub_rn(x[k], mn), recip), 0.5f)), 31u);
            bits |= (q[k] >> 4) << k;
        }
        uint32_t qh = group_or<G>(bits << (gl * EPL));
        if (write && gl == 0) { *(uint32_t *)blk = hdr; *(uint32_t *)(blk + 4) = qh; }
        pack_nibbles<EPL, G>(q, gl, blk + 8, write);
    }
};

template <uint32_t EPL>
__device__ __forceinline__ void store_i8(const int *q, uint8_t *p) {   // p 2-byte aligned
#pragma unroll
    for (uint32_t j = 0; j < EPL / 4; ++j) {
        // the low bytes of four registers into one word: three byte permutes (selector bytes 0-3 pick from the
        // second operand, 4-7 from the first)
        const uint32_t lo = __builtin_amdgcn_perm((uint32_t)q[4 * j + 1], (uint32_t)q[4 * j], 0x0c0c0400u);
        const uint32_t hi = __builtin_amdgcn_perm((uint32_t)q[4 * j + 3], (uint32_t)q[4 * j + 2], 0x0c0c0400u);
        st32_a2(p + 4 * j, __builtin_amdgcn_perm(hi, lo, 0x05040100u));
    }
}

// ggml-quants/src/structs/q8_0.rs:23-41
struct EncQ8_0 {
    static constexpr uint32_t N = 32, BYTES = 34;
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float amax = group_max_abs<G>(x, EPL);
        int q[EPL];
        // No branch for `amax == 0`: there delta = +0 (header 0) and recip = +inf, every element is +-0 or NaN, so
        // x * recip is NaN, which converts to 0 -- the all-zero block the reference returns.
        const float delta = __fdiv_rn(amax, 127.0f), recip = __fdiv_rn(1.0f, delta);
        const uint32_t hdr = f2h_bits(delta);
        for (uint32_t k = 0; k < EPL; ++k) q[k] = clamp_i8(round_away_i32(__fmul_rn(x[k], recip)));
        if (write) {
            if (gl == 0) st16(blk, hdr);
            store_i8<EPL>(q, blk + 2 + gl * EPL);
        }
    }
};

// ggml-quants/src/structs/q8_1.rs:28-55 ; sum = f16(sum_i16 as f32 * delta), delta unrounded
struct EncQ8_1 {
    static constexpr uint32_t N = 32, BYTES = 36;
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float amax = group_max_abs<G>(x, EPL);
        int q[EPL];
        int part = 0;
        // no branch for `amax == 0`, as in EncQ8_0: codes 0, sum 0, delta +0, so the header word is 0 as well
        const float delta = __fdiv_rn(amax, 127.0f), recip = __fdiv_rn(1.0f, delta);
        for (uint32_t k = 0; k < EPL; ++k) { q[k] = clamp_i8(round_away_i32(__fmul_rn(x[k], recip))); part += q[k]; }
        int sum = group_add<G>(part);
        if (write) {
            if (gl == 0) *(uint32_t *)blk = f2h_bits(delta) | (f2h_bits(__fmul_rn((float)sum, delta)) << 16);
            store_i8<EPL>(q, blk + 4 + gl * EPL);
        }
    }
};

// ggml-quants/src/structs/q8_k.rs:27-54 ; reference layout delta:f16 | quants[256] | sums[16]:i16
struct EncQ8K {
    static constexpr uint32_t N = 256, BYTES = 290;
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float max = group_max_by_abs<G>(x, EPL);
        int q[EPL];
        int part = 0;
        uint32_t hdr = 0;
        if (max == 0.0f) {
            for (uint32_t k = 0; k < EPL; ++k) q[k] = 0;
        } else {
            float delta = __fdiv_rn(max, -127.0f), recip = __fdiv_rn(1.0f, delta);
            hdr = f2h_bits(delta);
            for (uint32_t k = 0; k < EPL; ++k) {
                const float y = __fmul_rn(x[k], recip);
                // truncation and the minimum with an integer bound commute, so the bound is applied before the
                // (truncating) conversion; a NaN becomes 127 either way, as f32::min(NaN, 127.) does
                q[k] = clamp_i8(cvt_i32_sat(fminf(__fadd_rn(y, copysignf(0.49999997f, y)), 127.0f)));
                part += q[k];
            }
        }
        int sum16 = group_add<16 / EPL>(part);        // 16 consecutive elements
        if (write) {
            if (gl == 0) st16(blk, hdr);
            store_i8<EPL>(q, blk + 2 + gl * EPL);
            if ((gl & (16 / EPL - 1)) == 0) st16(blk + 258 + 2 * (gl * EPL / 16), (uint32_t)sum16 & 0xFFFFu);
        }
    }
};

// ---- the kernel ------------------------------------------------------------------
// Workgroup shape per (encoder, input type): THR threads, each with LPT 16-byte loads in flight.
// 256 threads beat 512 for every encoder (2-9 %, 1 Gi elements, `ubench ... "qsw "`); one load per thread
// wins where the packed block is small against its input (Q4_0 and Q5_0 from f32: 6450 / 6360 GB/s against
// 6020 / 6190 with 512 threads x 2 loads; Q4_1, Q5_0 and Q5_1 from 16-bit input), two loads elsewhere.
template <class Q, class In> struct QuantShape { static constexpr uint32_t THR = 256, LPT = 2; };
template <> struct QuantShape<EncQ4_0, InF32> { static constexpr uint32_t THR = 256, LPT = 1; };
template <> struct QuantShape<EncQ5_0, InF32> { static constexpr uint32_t THR = 256, LPT = 1; };
template <> struct QuantShape<EncQ5_0, InF16> { static constexpr uint32_t THR = 256, LPT = 1; };
template <> struct QuantShape<EncQ5_0, InBF16> { static constexpr uint32_t THR = 256, LPT = 1; };
template <> struct QuantShape<EncQ4_1, InF16> { static constexpr uint32_t THR = 256, LPT = 1; };
template <> struct QuantShape<EncQ4_1, InBF16> { static constexpr uint32_t THR = 256, LPT = 1; };
template <> struct QuantShape<EncQ5_1, InF16> { static constexpr uint32_t THR = 256, LPT = 1; };
template <> struct QuantShape<EncQ5_1, InBF16> { static constexpr uint32_t THR = 256, LPT = 1; };

template <class Q, class In>
struct QuantGeom {
    static constexpr uint32_t THR = QuantShape<Q, In>::THR, LPT = QuantShape<Q, In>::LPT;
    static constexpr uint32_t ELEMS = THR * In::EPL * LPT;
    static constexpr uint32_t TB = ELEMS / Q::N;
    static constexpr uint32_t TBYTES = TB * Q::BYTES;
    static_assert(ELEMS % Q::N == 0, "a tile is a whole number of blocks");
    static_assert(TBYTES % 16 == 0, "tiles leave as whole 16-byte stores");
};

// One tile: `in` = the tile's first element (16-byte aligned), `out` = its first packed block (16-byte aligned),
// nblk blocks present (<= TB).
template <class Q, class In, bool NTL, bool NTS>
__device__ __forceinline__ void quant_tile(uint8_t *lds, const typename In::T *in, uint8_t *out, uint32_t nblk) {
    typedef QuantGeom<Q, In> G;
    constexpr uint32_t THR = G::THR, LPT = G::LPT;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t LANES = Q::N / In::EPL;          // lanes per block
    float x[LPT][In::EPL];
#pragma unroll
    for (uint32_t i = 0; i < LPT; ++i) {
        uint32_t el = (i * THR + tid) * In::EPL;
        if (el / Q::N < nblk) In::template load<NTL>(in + el, x[i]);
        else
            for (uint32_t k = 0; k < In::EPL; ++k) x[i][k] = 0.0f;
    }
#pragma unroll
    for (uint32_t i = 0; i < LPT; ++i) {
        uint32_t el = (i * THR + tid) * In::EPL;
        uint32_t bt = el / Q::N;
        Q::template encode<In::EPL>(x[i], tid % LANES, lds + bt * Q::BYTES, bt < nblk);
    }
    __syncthreads();
    // packed tile -> global, 16 bytes per thread per pass; a short last tile ends in a byte tail
    const uint32_t nbytes = nblk * Q::BYTES;
    for (uint32_t o = tid * 16; o + 16 <= nbytes; o += THR * 16) {
        const uint4v v = *(const uint4v *)(lds + o);
        if (NTS) __builtin_nontemporal_store(v, (uint4v *)(out + o));
        else *(uint4v *)(out + o) = v;
    }
    const uint32_t tail0 = nbytes & ~15u;
    if (tail0 + tid < nbytes) out[tail0 + tid] = lds[tail0 + tid];
}

// One tensor per launch.  src: elements (16-byte aligned), dst: packed blocks (16-byte aligned), workgroup b -> tile b
template <class Q, class In, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__((QuantShape<Q, In>::THR)) void quant_one(const typename In::T *__restrict__ src,
                                                                    uint8_t *__restrict__ dst, uint64_t nblocks) {
    typedef QuantGeom<Q, In> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::TBYTES];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    quant_tile<Q, In, NTL, NTS>(lds, src + t * G::ELEMS, dst + t * G::TBYTES, nblk);
}

// Many tensors per launch (the quantizing counterpart of deq_batch): workgroup b encodes tile tile0 + b of the
// group; its record (same TileRef as the decode plans, in tile order) holds the address of the tile's first
// ELEMENT with the tile's block count in the top 16 bits, and the address of its first packed block.  Without it a
// model's tensors cost one launch each, and below some 8 M elements a launch is mostly ramp and drain.
template <class Q, class In>
__global__ __launch_bounds__((QuantShape<Q, In>::THR)) void quant_batch(const TileRef *__restrict__ table, uint64_t tile0) {
    typedef QuantGeom<Q, In> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::TBYTES];
    const TileRef me = table[tile0 + blockIdx.x];
    quant_tile<Q, In, true, true>(lds, (const typename In::T *)(me.src_nblk & 0xFFFFFFFFFFFFull), (uint8_t *)me.dst,
                                  (uint32_t)(me.src_nblk >> 48));
}

// ---- f32 / f16 / bf16 -> f16 / bf16 (the 1-element "blocks", half.rs:10-17, 26-33) ----
// 256 threads, two steps per thread: 6510 GB/s for f32 -> f16 on 1 Gi elements whether a launch runs behind
// another kernel or behind thirty of its own; one step per thread measured 6550-6640 in the first setting but
// 6230 in the second, 512 threads x 2 steps 6300 in both.
constexpr uint32_t NARROW_STEPS = 2;
template <class In, class O>
__global__ __launch_bounds__(STREAM_THREADS) void narrow_vec(const typename In::T *__restrict__ src,
                                                             typename O::T *__restrict__ dst, uint64_t n) {
    // n = elements, a multiple of In::EPL handled here; each thread converts EPL elements per step
    const uint64_t base = ((uint64_t)blockIdx.x * STREAM_THREADS * NARROW_STEPS + threadIdx.x) * In::EPL;
    float x[NARROW_STEPS][In::EPL];
#pragma unroll
    for (uint32_t i = 0; i < NARROW_STEPS; ++i) {
        const uint64_t el = base + (uint64_t)i * STREAM_THREADS * In::EPL;
        if (el + In::EPL <= n) In::template load<true>(src + el, x[i]);
    }
#pragma unroll
    for (uint32_t i = 0; i < NARROW_STEPS; ++i) {
        const uint64_t el = base + (uint64_t)i * STREAM_THREADS * In::EPL;
        if (el + In::EPL > n) continue;
        if constexpr (In::EPL == 4) {
            uint2v o = {O::pk(x[i][0], x[i][1]), O::pk(x[i][2], x[i][3])};
            __builtin_nontemporal_store(o, (uint2v *)(dst + el));
        } else {
            uint4v o = {O::pk(x[i][0], x[i][1]), O::pk(x[i][2], x[i][3]), O::pk(x[i][4], x[i][5]), O::pk(x[i][6], x[i][7])};
            __builtin_nontemporal_store(o, (uint4v *)(dst + el));
        }
    }
}

template <class In, class O>
__global__ __launch_bounds__(WG_THREADS) void narrow_scalar(const typename In::T *__restrict__ src,
                                                            typename O::T *__restrict__ dst,
                                                            uint64_t first, uint64_t n) {
    uint64_t i = first + (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * WG_THREADS;
    for (; i < n; i += stride) {
        float f;
        if constexpr (sizeof(typename In::T) == 4) f = src[i];
        else if constexpr (std::is_same<In, InF16>::value) f = h2f(src[i]);
        else f = __uint_as_float((uint32_t)src[i] << 16);
        dst[i] = (uint16_t)(O::pk(f, 0.0f) & 0xFFFFu);
    }
}

}  // namespace ggq
