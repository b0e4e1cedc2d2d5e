// quantize.h -- block quantizers for gfx950: the formats the reference implements
// (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1, Q8K) and the f16 / bf16 narrowing, bit-exact
// with the reference's Rust (QuantExt::quantize_slice, ggml-quants/src/lib.rs:121-133).
//
// Shape: the mirror image of the decode kernels.  One-shot grid, one tile per
// workgroup of 128-512 threads, one or two non-temporal 16-byte loads per thread (QuantShape below).
// A block's elements sit in G = N / EPL consecutive lanes of one wave;
// the per-block reductions (max by magnitude, min/max, sums, the 5th-bit mask) run over
// those lanes as DPP butterflies; packed blocks are assembled in LDS and leave as
// contiguous 16-byte stores.  Reads dominate here (4 bytes in, 0.56-1.13 out per element).
//
// The kernels are bound by vector instructions, not by memory (64 lane operations per CU and clock without
// packed arithmetic: 15-29 instructions per element were more than the load stream leaves room for), so the
// code below counts instructions: reductions whose lane exchange rides on the arithmetic instruction itself,
// and, for waves whose blocks are all ordinary (see fast_range), divisions without the scaling steps of the
// general sequence and quantization without the clamps that only degenerate blocks need.
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
// QUIET: the caller folds the elements with v_min_f32 / v_max_f32, which ignore a quiet NaN operand, as f32::min /
// f32::max of the reference do, but answer a SIGNALLING NaN operand with a NaN -- and the next step of the fold
// then ignores that NaN, i.e. drops everything folded so far.  Such callers get their elements quieted on the way
// in: one instruction per f32 element; for bf16 the hardware conversion quiets where the shift would not; the f16
// conversion always does.
struct InF32 {
    typedef float T;
    static constexpr uint32_t EPL = 4;
    template <bool QUIET = false>
    static __device__ __forceinline__ void widen(uint4v v, float (&x)[4]) {
        x[0] = __uint_as_float(v.x); x[1] = __uint_as_float(v.y); x[2] = __uint_as_float(v.z); x[3] = __uint_as_float(v.w);
        if constexpr (QUIET)
            for (uint32_t k = 0; k < 4; ++k) x[k] = __builtin_canonicalizef(x[k]);
    }
};
// (8 consecutive f32 per lane, two adjacent 16-byte loads, so that a block spans 4 lanes and the
// per-block divisions are shared by more elements, measured 2-5 % SLOWER: the 32-byte lane stride
// costs more at the memory than the saved instructions give back.)
// f16 / bf16 inputs are widened first (lib.rs:67-69, 82-84)
struct InF16 {
    typedef uint16_t T;
    static constexpr uint32_t EPL = 8;
    template <bool QUIET = false>
    static __device__ __forceinline__ void widen(uint4v v, float (&x)[8]) {
        x[0] = h2f(v.x & 0xFFFFu); x[1] = h2f(v.x >> 16); x[2] = h2f(v.y & 0xFFFFu); x[3] = h2f(v.y >> 16);
        x[4] = h2f(v.z & 0xFFFFu); x[5] = h2f(v.z >> 16); x[6] = h2f(v.w & 0xFFFFu); x[7] = h2f(v.w >> 16);
    }
};
__device__ __forceinline__ void bf16_pair_quiet(uint32_t w, float &lo, float &hi) {
    asm("v_cvt_f32_bf16 %0, %1" : "=v"(lo) : "v"(w));
    asm("v_cvt_f32_bf16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(hi) : "v"(w));
}
struct InBF16 {
    typedef uint16_t T;
    static constexpr uint32_t EPL = 8;
    template <bool QUIET = false>
    static __device__ __forceinline__ void widen(uint4v v, float (&x)[8]) {
        if constexpr (QUIET) {
            bf16_pair_quiet(v.x, x[0], x[1]); bf16_pair_quiet(v.y, x[2], x[3]);
            bf16_pair_quiet(v.z, x[4], x[5]); bf16_pair_quiet(v.w, x[6], x[7]);
        } else {
            x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xFFFF0000u);
            x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xFFFF0000u);
            x[4] = __uint_as_float(v.z << 16); x[5] = __uint_as_float(v.z & 0xFFFF0000u);
            x[6] = __uint_as_float(v.w << 16); x[7] = __uint_as_float(v.w & 0xFFFF0000u);
        }
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
// ---- the ordinary case: divisions without the general sequence's scaling steps ------------------
// hipcc's IEEE division is eleven instructions (v_div_scale x 2, v_rcp, five fused steps, v_div_fmas, v_div_fixup),
// most of them there for operands near the ends of the exponent range.  Each block costs two divisions per lane
// (delta = extreme / c, then 1 / delta), a quarter to a third of a quantizer's instructions.  For a magnitude a
// with 2^-100 <= a <= 2^100 (fast_range) every intermediate below is a normal number and
//   * a / c for a constant c:  q = RN(a * RN(1/c));  r = a - q * c (exact in one fused step);  RN(q + r * RN(1/c))
//   * 1 / d:  r = v_rcp_f32(d) (1 ulp);  twice:  e = 1 - d * r (exact in one fused step);  r = RN(r + e * r)
// give the correctly rounded quotient, i.e. the bits of the general sequence and of the reference's `/`.  That is
// a theorem for the first form only up to properties of c, and depends on the rcp unit's rounding for the second:
// ggq_selftest_fast_division runs both against the general sequence for every float in range on the GPU at hand
// (tests/test_gpu_fast_division.py), and tests/test_fast_division_cpu.py does the first form on the host.
// A wave takes this path only when ALL its blocks qualify (one scalar branch); a wave with a zero, flat,
// infinite, NaN-only or extremely scaled block runs the general code for all of its blocks.
// The fused steps are asm so that they stay exactly these instructions; the `ggq:fastdiv` tag lets the check on
// the generated code tell them from contractions of the reference's separate multiplies and adds.
__device__ __forceinline__ bool fast_range(float a) { return fabsf(a) >= 0x1p-100f && fabsf(a) <= 0x1p+100f; }
// true when every lane of the wave holds an ordinary magnitude
__device__ __forceinline__ bool wave_all_fast(float a) {
    return __builtin_expect(__builtin_amdgcn_ballot_w64(!fast_range(a)) == 0, 1);
}
template <int C>
__device__ __forceinline__ float div_const_fast(float a) {
    constexpr float c = (float)C, rc = 1.0f / (float)C;       // RN(1/c)
    float q = __fmul_rn(a, rc), r;
    asm("v_fma_f32 %0, -%1, %2, %3 ; ggq:fastdiv" : "=v"(r) : "v"(q), "s"(c), "v"(a));
    asm("v_fma_f32 %0, %1, %2, %3 ; ggq:fastdiv" : "=v"(q) : "v"(r), "s"(rc), "v"(q));
    return q;
}
// One asm block, reciprocal included: the result of a transcendental instruction may not be read by the next
// vector instruction (one wait state on this hardware), and the compiler does not insert the wait for a reader
// it cannot see into.  (With v_rcp_f32 left to the compiler and the first fused step directly behind it, half of
// all reciprocals came out wrong -- found by the self-test.)
__device__ __forceinline__ float recip_fast(float d) {
    float r, e;
    asm("v_rcp_f32 %0, %2\n\t"
        "s_nop 0\n\t"
        "v_fma_f32 %1, -%2, %0, 1.0 ; ggq:fastdiv\n\t"
        "v_fma_f32 %0, %1, %0, %0 ; ggq:fastdiv\n\t"
        "v_fma_f32 %1, -%2, %0, 1.0 ; ggq:fastdiv\n\t"
        "v_fma_f32 %0, %1, %0, %0 ; ggq:fastdiv"
        : "=&v"(r), "=&v"(e)
        : "v"(d));
    return r;
}

// ---- reductions over the G lanes of a block (G a power of two, lanes consecutive) --
// All of them are butterflies in which, before the step at distance M, the 2^k lanes of each
// half already agree; the step only needs ANY lane of the other half.  For M = 1, 2, 4, 8 that is
// a DPP lane pattern inside a row of 16 (quad permutes, then the mirrors) carried by the arithmetic
// instruction itself -- one instruction per step; written as asm because the compiler emits a copy, a DPP
// move, a canonicalizing max and the operation (four) for the same step.  Wider steps (Q8K only, whose blocks span
// 32 or 64 lanes) use the generic shuffle, except in max_by_abs, which swaps rows and halves with v_permlane*_swap.
// Every lane of the wave is active wherever these run (quant_tile calls the encoders unconditionally), so each
// DPP source lane is valid.  `s_nop 1`: a DPP operand written by the preceding vector instruction needs two
// wait states, and the compiler does not look into asm.
template <uint32_t CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
#define GGQ_DPP_OP2(name, M, r, v)                                                                       \
    do {                                                                                                 \
        if constexpr ((M) == 1)                                                                          \
            asm("s_nop 1\n\t" name " %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
        else if constexpr ((M) == 2)                                                                     \
            asm("s_nop 1\n\t" name " %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
        else if constexpr ((M) == 4)                                                                     \
            asm("s_nop 1\n\t" name " %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));     \
        else                                                                                             \
            asm("s_nop 1\n\t" name " %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));          \
    } while (0)
// v of a lane in the other half (lane ^ M for the quad permutes, the mirrored lane for M = 4, 8)
template <uint32_t M>
__device__ __forceinline__ uint32_t other_half(uint32_t v) {
    if constexpr (M >= 16) return (uint32_t)__shfl_xor((int)v, (int)M);
    else {
        uint32_t r;
        if constexpr (M == 1) asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
        else if constexpr (M == 2) asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
        else if constexpr (M == 4) asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
        else asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
        return r;
    }
}
template <uint32_t M>
__device__ __forceinline__ float other_half(float v) { return __uint_as_float(other_half<M>(__float_as_uint(v))); }
template <uint32_t M>
__device__ __forceinline__ int other_half(int v) { return (int)other_half<M>((uint32_t)v); }
// op(v, v of the other half) in one instruction.  The float forms are only applied to results of a previous
// v_max / v_min, never to a signalling NaN (which v_max_f32 / v_min_f32 would turn into a quiet NaN result
// instead of ignoring it).
template <uint32_t M>
__device__ __forceinline__ float max_other_half(float v) {
    if constexpr (M >= 16) return fmaxf(v, other_half<M>(v));
    else { float r; GGQ_DPP_OP2("v_max_f32_dpp", M, r, v); return r; }
}
template <uint32_t M>
__device__ __forceinline__ float min_other_half(float v) {
    if constexpr (M >= 16) return fminf(v, other_half<M>(v));
    else { float r; GGQ_DPP_OP2("v_min_f32_dpp", M, r, v); return r; }
}
template <uint32_t M>
__device__ __forceinline__ uint32_t or_other_half(uint32_t v) {
    if constexpr (M >= 16) return v | other_half<M>(v);
    else { uint32_t r; GGQ_DPP_OP2("v_or_b32_dpp", M, r, v); return r; }
}
template <uint32_t M>
__device__ __forceinline__ int add_other_half(int v) {
    if constexpr (M >= 16) return v + other_half<M>(v);
    else { int r; GGQ_DPP_OP2("v_add_u32_dpp", M, r, v); return r; }
}
// f(M) for M = 1, 2, 4, ... < G
template <uint32_t G, uint32_t M = 1, class Fn>
__device__ __forceinline__ void butterfly(Fn &&f) {
    if constexpr (M < G) {
        f(std::integral_constant<uint32_t, M>());
        butterfly<G, M * 2>(f);
    }
}
// lanes whose bit M is set: in the step at distance M they hold the LATER elements of the pair of halves
template <uint32_t M>
constexpr uint64_t later_lanes() {
    uint64_t m = 0;
    for (uint32_t l = 0; l < 64; ++l)
        if (l & M) m |= 1ull << l;
    return m;
}
template <uint32_t G>
__device__ __forceinline__ float group_max_by_abs(const float *x, uint32_t n) {
    // local fold, first to last; NaN never selected (the fold starts from +0.0)
    float best = 0.0f;
    for (uint32_t k = 0; k < n; ++k)
        if (fabsf(x[k]) > fabsf(best)) best = x[k];
    // across lanes: take the other half's value when it is larger in magnitude, or equal and EARLIER (this lane
    // is in the later half).  The two comparisons land in scalar registers, the tie rule is scalar mask
    // arithmetic, and one select applies it: four vector instructions per step.  No value here is a NaN.
    butterfly<G>([&](auto m) {
        if constexpr (m.value >= 16) {
            // across rows of 16 lanes and halves of 32 (Q8K): one swap instruction leaves the earlier half's value
            // in r[0] and the later half's in r[1] on both sides -- no trip through the LDS crossbar, and the tie
            // rule needs no lane mask
            const uint32_t bits = __float_as_uint(best);
            float early, late;
            if constexpr (m.value == 16) {
                const auto r = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
                early = __uint_as_float(r[0]); late = __uint_as_float(r[1]);
            } else {
                const auto r = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
                early = __uint_as_float(r[0]); late = __uint_as_float(r[1]);
            }
            best = fabsf(late) > fabsf(early) ? late : early;
        } else {
            const float other = other_half<m.value>(best);
            const uint64_t larger = __builtin_amdgcn_ballot_w64(fabsf(other) > fabsf(best));
            const uint64_t equal = __builtin_amdgcn_ballot_w64(fabsf(other) == fabsf(best));
            const uint64_t take = larger | (equal & later_lanes<m.value>());
            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(best) : "v"(best), "v"(other), "s"(take));
        }
    });
    return best;
}
template <uint32_t G>
__device__ __forceinline__ float group_max_abs(const float *x, uint32_t n) {
    // acc.max(|x|): f32::max ignores a NaN operand, which is what v_max_f32 does for a quiet NaN (the elements
    // arrive quieted, see InF32); one instruction per element
    float m = 0.0f;
    for (uint32_t k = 0; k < n; ++k) m = fmaxf(m, fabsf(x[k]));
    butterfly<G>([&](auto s) { m = max_other_half<s.value>(m); });
    return m;
}
// min and max with NaN ignored, folded from (f32::MAX, f32::MIN) (structs.rs:101-107): f32::min / f32::max return
// the other operand when one is a NaN, which is v_min_f32 / v_max_f32 for a quiet NaN (minNum / maxNum; the
// elements arrive quieted, see InF32).  For (+0, -0) Rust leaves the result open; the hardware orders -0 below +0, and so
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
        lo = min_other_half<s.value>(lo);
        hi = max_other_half<s.value>(hi);
    });
}
template <uint32_t G>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
    butterfly<G>([&](auto s) { v = or_other_half<s.value>(v); });
    return v;
}
template <uint32_t G>
__device__ __forceinline__ int group_add(int v) {
    butterfly<G>([&](auto s) { v = add_other_half<s.value>(v); });
    return v;
}

// LDS stores at 2-byte alignment
__device__ __forceinline__ void st16(uint8_t *p, uint32_t v) { *(uint16_t *)p = (uint16_t)v; }
__device__ __forceinline__ void st32_a2(uint8_t *p, uint32_t v) { st16(p, v); st16(p + 2, v >> 16); }

// the low bytes of EPL registers as EPL / 4 words, element order = byte order: three byte permutes per word
// (selector bytes 0-3 pick from the second operand, 4-7 from the first)
template <uint32_t EPL>
__device__ __forceinline__ void pack_codes(const uint32_t *q, uint32_t (&w)[EPL / 4]) {
#pragma unroll
    for (uint32_t j = 0; j < EPL / 4; ++j) {
        const uint32_t lo = __builtin_amdgcn_perm(q[4 * j + 1], q[4 * j], 0x0c0c0400u);
        const uint32_t hi = __builtin_amdgcn_perm(q[4 * j + 3], q[4 * j + 2], 0x0c0c0400u);
        w[j] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
    }
}
// 5-bit codes (each byte of w <= 31): bit 4 of element k of this lane at bit k of the result.  A byte dot product
// with the weights 1, 2, 4, 8 (16, 32, 64, 128 for the second word) gathers four bits per instruction: the masked
// bytes are 0 or 16, so the sum is 16 x the wanted bits.
template <uint32_t EPL>
__device__ __forceinline__ uint32_t fifth_bits(const uint32_t (&w)[EPL / 4]) {
    static_assert(EPL == 4 || EPL == 8, "one or two words of codes per lane");
    uint32_t acc = __builtin_amdgcn_udot4(w[0] & 0x10101010u, 0x08040201u, 0u, false);
    if constexpr (EPL == 8) acc = __builtin_amdgcn_udot4(w[1] & 0x10101010u, 0x80402010u, acc, false);
    return acc >> 4;
}
// nibble formats: the words w hold the codes of elements e0..e0+EPL-1 (e0 = gl * EPL), one per byte.  Byte i of
// the block's 16 payload bytes = code(i) & 15 | code(i + 16) << 4: the low-half lanes fetch the words
// of the lane G/2 above and write EPL bytes each.
template <uint32_t EPL, uint32_t G>
__device__ __forceinline__ void pack_nibbles(const uint32_t (&w)[EPL / 4], uint32_t gl, uint8_t *payload, bool write) {
#pragma unroll
    for (uint32_t j = 0; j < EPL / 4; ++j) {
        const uint32_t lo = w[j] & 0x0F0F0F0Fu;
        // the word of the lane G/2 above (only the lower half of the group uses it)
        uint32_t hi;
        if constexpr (G == 4) hi = dpp_u32<0x4E>(lo);          // quad_perm [2,3,0,1]
        else if constexpr (G == 8) hi = dpp_u32<0x104>(lo);    // row_shl:4
        else hi = (uint32_t)__shfl_xor((int)lo, (int)(G / 2));
        if (write && gl < G / 2) st32_a2(payload + gl * EPL + 4 * j, lo | (hi << 4));
    }
}

// ================================ the formats =====================================
// Each encode() is called by all lanes of a wave; `blk` = this lane's block in LDS,
// gl = lane within the block's group, x = this lane's EPL elements, write = block exists.

// ggml-quants/src/structs/q4_0.rs:23-44
struct EncQ4_0 {
    static constexpr uint32_t N = 32, BYTES = 18;
    static constexpr bool QUIET = false;     // folds by comparison: any NaN compares false
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float max = group_max_by_abs<G>(x, EPL);
        uint32_t q[EPL];
        uint32_t hdr = 0;
        if (wave_all_fast(max)) {
            const float delta = __fmul_rn(max, -0.125f), recip = recip_fast(delta);
            hdr = f2h_bits(delta);
            for (uint32_t k = 0; k < EPL; ++k) q[k] = cvt_u32_sat(fminf(__fadd_rn(__fmul_rn(x[k], recip), 8.5f), 15.0f));
        } else if (max == 0.0f) {
            for (uint32_t k = 0; k < EPL; ++k) q[k] = 0;
        } else {
            float delta = __fdiv_rn(max, -8.0f), recip = __fdiv_rn(1.0f, delta);
            hdr = f2h_bits(delta);
            // `(y + 8.5).min(15.) as u8`: the float minimum (a NaN gives 15) leaves nothing for the cast to clamp from above
            for (uint32_t k = 0; k < EPL; ++k) q[k] = cvt_u32_sat(fminf(__fadd_rn(__fmul_rn(x[k], recip), 8.5f), 15.0f));
        }
        if (write && gl == 0) st16(blk, hdr);
        uint32_t w[EPL / 4];
        pack_codes<EPL>(q, w);
        pack_nibbles<EPL, G>(w, gl, blk + 2, write);
    }
};

// ggml-quants/src/structs/q4_1.rs:23-47
struct EncQ4_1 {
    static constexpr uint32_t N = 32, BYTES = 20;
    static constexpr bool QUIET = true;      // folds with v_min_f32 / v_max_f32 (see InF32)
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float mn, mx;
        group_min_max<G>(x, EPL, mn, mx);
        uint32_t q[EPL];
        float delta;
        const float span = __fsub_rn(mx, mn);
        if (wave_all_fast(span)) {
            // x - min <= max - min (rounding is monotone) and delta, recip are accurate to an ulp each, so the
            // scaled value stays below 15.5: the `.min(15)` of the reference cannot bind here
            delta = div_const_fast<15>(span);
            const float recip = recip_fast(delta);
            for (uint32_t k = 0; k < EPL; ++k) q[k] = cvt_u32_sat(__fadd_rn(__fmul_rn(__fsub_rn(x[k], mn), recip), 0.5f));
        } else {
            // No branch for the reference's `min == max` case (header (0, min), all codes 0): with delta = 0 there,
            // recip = inf, every element gives (x - min) * inf = 0 * inf = NaN (or NaN from a NaN or infinite element),
            // a NaN converts to 0, and f16(0) | f16(min) << 16 is the header -- the general path produces that block.
            // (delta is selected, not computed, in that case: for min = max = inf the difference would be NaN.)
            delta = mn == mx ? 0.0f : __fdiv_rn(span, 15.0f);
            const float recip = __fdiv_rn(1.0f, delta);
            // `as u8` saturates at 255, `.min(15)` follows: one integer minimum does both
            for (uint32_t k = 0; k < EPL; ++k)
                q[k] = min(cvt_u32_sat(__fadd_rn(__fmul_rn(__fsub_rn(x[k], mn), recip), 0.5f)), 15u);
        }
        const uint32_t hdr = f2h_bits(delta) | (f2h_bits(mn) << 16);
        if (write && gl == 0) *(uint32_t *)blk = hdr;
        uint32_t w[EPL / 4];
        pack_codes<EPL>(q, w);
        pack_nibbles<EPL, G>(w, gl, blk + 4, write);
    }
};

// ggml-quants/src/structs/q5_0.rs:26-58
struct EncQ5_0 {
    static constexpr uint32_t N = 32, BYTES = 22;
    static constexpr bool QUIET = false;     // folds by comparison: any NaN compares false
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float max = group_max_by_abs<G>(x, EPL);
        uint32_t q[EPL];
        float delta, recip;
        uint32_t hdr;
        if (wave_all_fast(max)) {
            delta = __fmul_rn(max, -0.0625f);
            recip = recip_fast(delta);
            hdr = f2h_bits(delta);
        } else {
            // No branch for `max == 0` (an all-zero block): there delta = -0 and recip = -inf, every element is +-0 or
            // NaN, so x * recip is NaN, which converts to code 0; only the header needs the case (f16(-0) is not 0).
            delta = __fdiv_rn(max, -16.0f);
            recip = __fdiv_rn(1.0f, delta);
            hdr = max == 0.0f ? 0u : f2h_bits(delta);
        }
        // (the element of largest magnitude with the sign opposite to max's lands on 32.5: `.min(31)` binds)
        for (uint32_t k = 0; k < EPL; ++k) q[k] = min(cvt_u32_sat(__fadd_rn(__fmul_rn(x[k], recip), 16.5f)), 31u);
        uint32_t w[EPL / 4];
        pack_codes<EPL>(q, w);
        uint32_t qh = group_or<G>(fifth_bits<EPL>(w) << (gl * EPL));       // element e owns bit e of qh
        if (write && gl == 0) { st16(blk, hdr); st32_a2(blk + 2, qh); }
        pack_nibbles<EPL, G>(w, gl, blk + 6, write);
    }
};

// ggml-quants/src/structs/q5_1.rs:26-62
struct EncQ5_1 {
    static constexpr uint32_t N = 32, BYTES = 24;
    static constexpr bool QUIET = true;      // folds with v_min_f32 / v_max_f32 (see InF32)
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float mn, mx;
        group_min_max<G>(x, EPL, mn, mx);
        uint32_t q[EPL];
        float delta;
        const float span = __fsub_rn(mx, mn);
        if (wave_all_fast(span)) {
            // as in EncQ4_1: the scaled value stays below 31.5, `.min(31)` cannot bind
            delta = div_const_fast<31>(span);
            const float recip = recip_fast(delta);
            for (uint32_t k = 0; k < EPL; ++k) q[k] = cvt_u32_sat(__fadd_rn(__fmul_rn(__fsub_rn(x[k], mn), recip), 0.5f));
        } else {
            // no branch for `min == max`, for the reason given at EncQ4_1
            delta = mn == mx ? 0.0f : __fdiv_rn(span, 31.0f);
            const float recip = __fdiv_rn(1.0f, delta);
            for (uint32_t k = 0; k < EPL; ++k)
                q[k] = min(cvt_u32_sat(__fadd_rn(__fmul_rn(__fsub_rn(x[k], mn), recip), 0.5f)), 31u);
        }
        const uint32_t hdr = f2h_bits(delta) | (f2h_bits(mn) << 16);
        uint32_t w[EPL / 4];
        pack_codes<EPL>(q, w);
        uint32_t qh = group_or<G>(fifth_bits<EPL>(w) << (gl * EPL));
        if (write && gl == 0) { *(uint32_t *)blk = hdr; *(uint32_t *)(blk + 4) = qh; }
        pack_nibbles<EPL, G>(w, gl, blk + 8, write);
    }
};

template <uint32_t EPL>
__device__ __forceinline__ void store_i8(const int *q, uint8_t *p) {   // p 2-byte aligned
    uint32_t w[EPL / 4];
    pack_codes<EPL>((const uint32_t *)q, w);
#pragma unroll
    for (uint32_t j = 0; j < EPL / 4; ++j) st32_a2(p + 4 * j, w[j]);
}

// ggml-quants/src/structs/q8_0.rs:23-41
struct EncQ8_0 {
    static constexpr uint32_t N = 32, BYTES = 34;
    static constexpr bool QUIET = true;      // folds with v_min_f32 / v_max_f32 (see InF32)
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float amax = group_max_abs<G>(x, EPL);
        int q[EPL];
        float delta;
        if (wave_all_fast(amax)) {
            // |x| <= amax and delta, recip are accurate to an ulp each: |x * recip| < 127.0001, the rounded code
            // is within -127 ..= 127 and the saturation of `as i8` cannot bind (a NaN element converts to 0)
            delta = div_const_fast<127>(amax);
            const float recip = recip_fast(delta);
            for (uint32_t k = 0; k < EPL; ++k) q[k] = round_away_i32(__fmul_rn(x[k], recip));
        } else {
            // No branch for `amax == 0`: there delta = +0 (header 0) and recip = +inf, every element is +-0 or NaN, so
            // x * recip is NaN, which converts to 0 -- the all-zero block the reference returns.
            delta = __fdiv_rn(amax, 127.0f);
            const float recip = __fdiv_rn(1.0f, delta);
            for (uint32_t k = 0; k < EPL; ++k) q[k] = clamp_i8(round_away_i32(__fmul_rn(x[k], recip)));
        }
        const uint32_t hdr = f2h_bits(delta);
        if (write) {
            if (gl == 0) st16(blk, hdr);
            store_i8<EPL>(q, blk + 2 + gl * EPL);
        }
    }
};

// ggml-quants/src/structs/q8_1.rs:28-55 ; sum = f16(sum_i16 as f32 * delta), delta unrounded
struct EncQ8_1 {
    static constexpr uint32_t N = 32, BYTES = 36;
    static constexpr bool QUIET = true;      // folds with v_min_f32 / v_max_f32 (see InF32)
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float amax = group_max_abs<G>(x, EPL);
        int q[EPL];
        int part = 0;
        float delta;
        if (wave_all_fast(amax)) {                       // as in EncQ8_0
            delta = div_const_fast<127>(amax);
            const float recip = recip_fast(delta);
            for (uint32_t k = 0; k < EPL; ++k) { q[k] = round_away_i32(__fmul_rn(x[k], recip)); part += q[k]; }
        } else {
            // no branch for `amax == 0`, as in EncQ8_0: codes 0, sum 0, delta +0, so the header word is 0 as well
            delta = __fdiv_rn(amax, 127.0f);
            const float recip = __fdiv_rn(1.0f, delta);
            for (uint32_t k = 0; k < EPL; ++k) { q[k] = clamp_i8(round_away_i32(__fmul_rn(x[k], recip))); part += q[k]; }
        }
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
    static constexpr bool QUIET = false;     // folds by comparison: any NaN compares false
    template <uint32_t EPL>
    static __device__ __forceinline__ void encode(const float (&x)[EPL], uint32_t gl, uint8_t *blk, bool write) {
        constexpr uint32_t G = N / EPL;
        float max = group_max_by_abs<G>(x, EPL);
        int q[EPL];
        int part = 0;
        uint32_t hdr = 0;
        if (wave_all_fast(max)) {
            // |x| <= |max|: the rounded code is within -127 ..= 127, `as i8` cannot bind; the float minimum stays
            // for the NaN element, which the reference turns into 127
            const float delta = div_const_fast<-127>(max), recip = recip_fast(delta);
            hdr = f2h_bits(delta);
            for (uint32_t k = 0; k < EPL; ++k) {
                const float y = __fmul_rn(x[k], recip);
                q[k] = cvt_i32_sat(fminf(__fadd_rn(y, copysignf(0.49999997f, y)), 127.0f));
                part += q[k];
            }
        } else if (max == 0.0f) {
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
// Workgroup shape per (encoder, input type): THR threads, each with LPT 16-byte loads in flight, and the group
// start of quant_one (QUANT_GROUP_LOG2).  From sweeps of 64 ... 512 threads x 1, 2, 4 loads and groups of 2 ... 128
// tiles on 1 Gi elements, taken after the instruction counts had come down (tools/ubench/sweep_quant_shapes.sh and
// the "qgw" variants of tools/ubench).  Without a group start small tiles with ONE load per thread win for f32 input
// (82-85 % of 8 TB/s) and 128 x 2 for 16-bit input (79-82 %); with it two loads per thread win everywhere
// (84-87 % for groups of 16-32 tiles; one load per thread then loses 12 points; Q8K 80-82 % for groups of 4-16 once
// its wide reduction steps stopped going through LDS).  Through a tile table (quant_batch, no group start) the same
// shapes reach 80-85 %.
#if defined(GGQ_QSHAPE_THR) && defined(GGQ_QSHAPE_LPT)
// measurement builds (tools/ubench): one shape for every (encoder, input type) whose tiles it fits, the default elsewhere
template <class Q, class In> struct QuantShape {
    static constexpr uint32_t E = GGQ_QSHAPE_THR * In::EPL * GGQ_QSHAPE_LPT;
    static constexpr bool FITS = E % Q::N == 0 && (E / Q::N * Q::BYTES) % 16 == 0;
    static constexpr uint32_t THR = FITS ? GGQ_QSHAPE_THR : 256, LPT = FITS ? GGQ_QSHAPE_LPT : 2;
};
#else
template <class Q, class In> struct QuantShape { static constexpr uint32_t THR = 256, LPT = 2; };
template <class In> struct QuantShape<EncQ4_1, In> { static constexpr uint32_t THR = 128, LPT = 2; };
template <class In> struct QuantShape<EncQ5_1, In> { static constexpr uint32_t THR = 128, LPT = 2; };
template <class In> struct QuantShape<EncQ8_1, In> { static constexpr uint32_t THR = 128, LPT = 2; };
template <> struct QuantShape<EncQ8_0, InF16> { static constexpr uint32_t THR = 128, LPT = 2; };
template <> struct QuantShape<EncQ8_0, InBF16> { static constexpr uint32_t THR = 128, LPT = 2; };
#endif

// groups of 2^5 = 32 tiles start together (quant_one's `wait`); Q8K, whose tiles hold 8 blocks: 8 tiles
template <class Q> struct QuantGroup { static constexpr uint32_t LOG2 = 5; };
template <> struct QuantGroup<EncQ8K> { static constexpr uint32_t LOG2 = 3; };

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
    // all loads first, conversion (and quieting) afterwards: a conversion next to its load would make the second
    // load wait for the first
    uint4v raw[LPT];
#pragma unroll
    for (uint32_t i = 0; i < LPT; ++i) {
        uint32_t el = (i * THR + tid) * In::EPL;
        raw[i] = el / Q::N < nblk ? load16<NTL>(in + el) : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (uint32_t i = 0; i < LPT; ++i) {
        uint32_t el = (i * THR + tid) * In::EPL;
        uint32_t bt = el / Q::N;
        float x[In::EPL];
        In::template widen<Q::QUIET>(raw[i], x);
        Q::template encode<In::EPL>(x, tid % LANES, lds + bt * Q::BYTES, bt < nblk);
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
// `wait` = k > 0: the workgroups of 2^k consecutive tiles start together, behind one load of the (cold) line where
// the group's packed output begins -- the group start of the decode kernels (group_wait, kernels.h); 0: off.
template <class Q, class In, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__((QuantShape<Q, In>::THR)) void quant_one(const typename In::T *__restrict__ src,
                                                                    uint8_t *__restrict__ dst, uint64_t nblocks,
                                                                    uint32_t wait) {
    typedef QuantGeom<Q, In> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::TBYTES];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint32_t z = wait ? group_wait(dst + ((t >> wait) << wait) * G::TBYTES) : 0u;
    quant_tile<Q, In, NTL, NTS>(lds, src + t * G::ELEMS + z, dst + t * G::TBYTES, nblk);
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
        if (el + In::EPL <= n) In::widen(load16<true>(src + el), x[i]);
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

// ---- self-test of the short division sequences ----------------------------------------------------
// Thread grid over all 2^32 bit patterns: for every float the quantizers may hand to div_const_fast / recip_fast
// (and a margin around it) the short sequence must give the bits of the general IEEE division.  bad[0..3]: a / 15,
// a / 31, a / 127, a / -127 for fast_range(a); bad[4]: 1 / d for 2^-110 <= |d| <= 2^110 (delta = a / c with
// |c| >= 8 lies well inside).  bad[5] counts the patterns tested, so that a launch that did nothing shows.
__global__ __launch_bounds__(256) void fast_division_selftest(unsigned long long *bad) {
    unsigned long long n[6] = {0, 0, 0, 0, 0, 0};
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nthreads = gridDim.x * 256u;
    for (uint64_t i = tid; i < (1ull << 32); i += nthreads) {
        const float a = __uint_as_float((uint32_t)i);
        if (fast_range(a)) {
            n[0] += __float_as_uint(div_const_fast<15>(a)) != __float_as_uint(__fdiv_rn(a, 15.0f));
            n[1] += __float_as_uint(div_const_fast<31>(a)) != __float_as_uint(__fdiv_rn(a, 31.0f));
            n[2] += __float_as_uint(div_const_fast<127>(a)) != __float_as_uint(__fdiv_rn(a, 127.0f));
            n[3] += __float_as_uint(div_const_fast<-127>(a)) != __float_as_uint(__fdiv_rn(a, -127.0f));
            n[5] += 1;
        }
        if (fabsf(a) >= 0x1p-110f && fabsf(a) <= 0x1p+110f)
            n[4] += __float_as_uint(recip_fast(a)) != __float_as_uint(__fdiv_rn(1.0f, a));
    }
    for (int k = 0; k < 6; ++k)
        if (n[k]) atomicAdd(bad + k, n[k]);
}

}  // namespace ggq
