// scale.h -- every element of a float tensor times one factor, in the tensor's own type.
//
// The reference's to-llama operator rescales token_embd, attn_output and ffn_down of a MiniCPM
// model (xtask/src/utils/operator/to_llama.rs:66-77, 99-125): `*x *= scale` over the tensor as
// f64, f32, f16 or bf16, with the factor converted to that type first.  For the 16-bit types the
// product of the two widened values is exact in f32 (11 x 11 and 8 x 8 significand bits), so one
// f32 multiply and one round-to-nearest-even narrowing give the correctly rounded product.
// A pure stream, half reads, half writes: one-shot grid, 256 threads, one 16-byte piece per thread.
#pragma once
#include "kernels.h"

namespace ggq {

enum ScaleKind : uint32_t { SC_F32 = 0, SC_F16 = 1, SC_BF16 = 2, SC_F64 = 3 };

// `factor`: the factor's bit pattern in the tensor's type (low 16 / 32 / all 64 bits)
template <uint32_t K>
struct ScaleOp {
    static constexpr uint32_t BYTES = K == SC_F64 ? 8 : K == SC_F32 ? 4 : 2;
    float s;
    double sd;
    __device__ __forceinline__ explicit ScaleOp(uint64_t factor) {
        s = 0.0f;
        sd = 0.0;
        if constexpr (K == SC_F32) s = __builtin_bit_cast(float, (uint32_t)factor);
        else if constexpr (K == SC_F16) s = h2f((uint32_t)factor & 0xFFFFu);
        else if constexpr (K == SC_BF16) s = __builtin_bit_cast(float, (uint32_t)factor << 16);
        else sd = __builtin_bit_cast(double, factor);
    }
    __device__ __forceinline__ uint32_t f32(uint32_t w) const {
        return __builtin_bit_cast(uint32_t, __fmul_rn(__builtin_bit_cast(float, w), s));
    }
    // two 16-bit elements in one word
    __device__ __forceinline__ uint32_t pair(uint32_t w) const {
        if constexpr (K == SC_F16) return OutF16::pk(__fmul_rn(h2f(w & 0xFFFFu), s), __fmul_rn(h2f(w >> 16), s));
        else return OutBF16::pk(__fmul_rn(__builtin_bit_cast(float, w << 16), s),
                                __fmul_rn(__builtin_bit_cast(float, w & 0xFFFF0000u), s));
    }
    __device__ __forceinline__ uint64_t f64(uint64_t w) const {
        return __builtin_bit_cast(uint64_t, __dmul_rn(__builtin_bit_cast(double, w), sd));
    }
    __device__ __forceinline__ uint4v piece(uint4v v) const {
        if constexpr (K == SC_F32) return {f32(v.x), f32(v.y), f32(v.z), f32(v.w)};
        else if constexpr (K == SC_F64) {
            const uint64_t a = f64(((uint64_t)v.y << 32) | v.x), b = f64(((uint64_t)v.w << 32) | v.z);
            return {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
        } else return {pair(v.x), pair(v.y), pair(v.z), pair(v.w)};
    }
};

// dst[i] = src[i] * factor over 16-byte pieces; dst may be src (hence no __restrict__ on either).  `first`: piece
// this launch starts at.
template <uint32_t K>
__global__ __launch_bounds__(STREAM_THREADS) void scale_vec(uint4v *dst, const uint4v *src,
                                                            uint64_t pieces, uint64_t first, uint64_t factor) {
    const uint64_t p = first + (uint64_t)blockIdx.x * STREAM_THREADS + threadIdx.x;
    if (p >= pieces) return;
    const ScaleOp<K> op(factor);
    const uint4v v = __builtin_nontemporal_load(src + p);
    __builtin_nontemporal_store(op.piece(v), dst + p);
}

// one element per thread: tails, and tensors whose addresses are not 16-byte aligned
template <uint32_t K>
__global__ __launch_bounds__(STREAM_THREADS) void scale_scalar(uint8_t *dst, const uint8_t *src,
                                                               uint64_t elems, uint64_t first, uint64_t factor) {
    const uint64_t i = first + (uint64_t)blockIdx.x * STREAM_THREADS + threadIdx.x;
    if (i >= elems) return;
    const ScaleOp<K> op(factor);
    if constexpr (K == SC_F32) {
        ((uint32_t *)dst)[i] = op.f32(((const uint32_t *)src)[i]);
    } else if constexpr (K == SC_F64) {
        ((uint64_t *)dst)[i] = op.f64(((const uint64_t *)src)[i]);
    } else {
        ((uint16_t *)dst)[i] = (uint16_t)op.pair(((const uint16_t *)src)[i]);
    }
}

}  // namespace ggq
