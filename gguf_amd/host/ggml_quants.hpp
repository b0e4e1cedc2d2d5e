// ggml_quants.hpp -- C++ host-side mirror of the reference crate's block
// interface, over the C ABI of include/ggq.h.
//
// The reference is a Rust crate (ggml-quants); Rust is not available where this
// library is built, so the host side above the C ABI is C++ with the same names,
// argument meaning and error behaviour:
//
//   trait DataBlock { ID, COUNT, ZEROS }          ggml-quants/src/lib.rs:11-21
//   trait QuantExt<T, N>::dequantize_slice        ggml-quants/src/lib.rs:98-104, 135-147
//   enum  QuantizeError                           ggml-quants/src/lib.rs:107-113
//   block structs                                 ggml-quants/src/structs/*.rs
//
//   Rust:  Q8_0::dequantize_slice(&mut out, &blocks)?;
//   C++ :  auto r = QuantExt<Q8_0, float>::dequantize_slice({out, n}, {blocks, m});
//
// Block structs are laid out exactly as the reference's #[repr(C)] structs, so a
// pointer into an mmap'ed GGUF file can be reinterpreted as `const Blk *`.
#pragma once
#include <cstddef>
#include <cstdint>
#include <optional>

#include "../../include/ggq.h"

namespace ggml_quants {

/// `half::f16` / `half::bf16`: transparent 16-bit patterns.
struct f16 { uint16_t bits; };
struct bf16 { uint16_t bits; };

/// structs.rs:54-61
struct alignas(4) DeltaMin { f16 delta; f16 min; };

/// lib.rs:107-113, plus the cases the reference handles by panicking.
enum class QuantizeError {
    Indivisible = GGQ_INDIVISIBLE,
    LengthMismatch = GGQ_LENGTH_MISMATCH,
    Unsupported = GGQ_UNSUPPORTED,
    Device = GGQ_DEVICE,
    Misaligned = GGQ_MISALIGNED,
    Invalid = GGQ_INVALID,
};
/// `Result<(), QuantizeError>`: empty = Ok(()).
using Result = std::optional<QuantizeError>;

template <class T> struct Slice {
    T *ptr;
    size_t len;
};

// ---- block types (field order and sizes: the cited struct definitions) ----------
struct Q4_0 { f16 delta; uint8_t quants[16]; };                                    // q4_0.rs:7-12
struct Q4_1 { DeltaMin delta_min; uint8_t quants[16]; };                           // q4_1.rs:7-12
struct Q5_0 { f16 delta; uint8_t qh[4]; uint8_t ql[16]; };                         // q5_0.rs:7-14
struct Q5_1 { DeltaMin delta_min; uint8_t qh[4]; uint8_t ql[16]; };                // q5_1.rs:7-14
struct Q8_0 { f16 delta; int8_t quants[32]; };                                     // q8_0.rs:7-12
struct alignas(4) Q8_1 { f16 delta; f16 sum; int8_t quants[32]; };                 // q8_1.rs:8-16
struct Q2K { uint8_t scales[16]; uint8_t qs[64]; DeltaMin delta_min; };            // q2_k.rs:6-13
struct Q3K { uint8_t hmask[32]; uint8_t qs[64]; uint8_t scales[12]; f16 delta; };  // q3_k.rs:6-15
struct Q4K { DeltaMin delta_min; uint8_t scales[12]; uint8_t qs[128]; };           // q4_k.rs:6-13
struct Q5K { f16 delta; f16 min; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; };  // q5_k.rs:7-18
struct Q6K { uint8_t ql[128]; uint8_t qh[64]; uint8_t scales[16]; f16 delta; };    // q6_k.rs:7-16
struct Q8K { f16 delta; int8_t quants[256]; int16_t sums[16]; };                   // q8_k.rs:8-15
// GGUF on-disk layouts (the reference structs iq4nl.rs:6-11 / iq4xs.rs:6-15 declare
// u16 quants and do not match files):
struct IQ4NL { f16 delta; uint8_t qs[16]; };
struct IQ4XS { f16 delta; uint16_t scales_h; uint8_t scales_l[4]; uint8_t qs[128]; };

/// DataBlock (lib.rs:11-21): ID is the GGmlType discriminant, NAME the layout name.
template <class Blk> struct DataBlock;
#define GGQ_DATA_BLOCK(TY, IDV, CNT, NAMEV)                       \
    template <> struct DataBlock<TY> {                            \
        static constexpr uint32_t ID = IDV;                       \
        static constexpr size_t COUNT = CNT;                      \
        static constexpr const char *NAME = NAMEV;                \
    };
GGQ_DATA_BLOCK(f16, GGQ_TYPE_F16, 1, "f16")
GGQ_DATA_BLOCK(bf16, GGQ_TYPE_BF16, 1, "bf16")
GGQ_DATA_BLOCK(Q4_0, GGQ_TYPE_Q4_0, 32, "q40")
GGQ_DATA_BLOCK(Q4_1, GGQ_TYPE_Q4_1, 32, "q41")
GGQ_DATA_BLOCK(Q5_0, GGQ_TYPE_Q5_0, 32, "q50")
GGQ_DATA_BLOCK(Q5_1, GGQ_TYPE_Q5_1, 32, "q51")
GGQ_DATA_BLOCK(Q8_0, GGQ_TYPE_Q8_0, 32, "q80")
GGQ_DATA_BLOCK(Q8_1, GGQ_TYPE_Q8_1, 32, "q81")
GGQ_DATA_BLOCK(Q2K, GGQ_TYPE_Q2K, 256, "q2k")
GGQ_DATA_BLOCK(Q3K, GGQ_TYPE_Q3K, 256, "q3k")
GGQ_DATA_BLOCK(Q4K, GGQ_TYPE_Q4K, 256, "q4k")
GGQ_DATA_BLOCK(Q5K, GGQ_TYPE_Q5K, 256, "q5k")
GGQ_DATA_BLOCK(Q6K, GGQ_TYPE_Q6K, 256, "q6k")
GGQ_DATA_BLOCK(Q8K, GGQ_TYPE_Q8K, 256, "q8k")
GGQ_DATA_BLOCK(IQ4NL, GGQ_TYPE_IQ4NL, 32, "iq4nl")
GGQ_DATA_BLOCK(IQ4XS, GGQ_TYPE_IQ4XS, 256, "iq4xs")
#undef GGQ_DATA_BLOCK

static_assert(sizeof(Q4_0) == 18 && sizeof(Q4_1) == 20 && sizeof(Q5_0) == 22 && sizeof(Q5_1) == 24, "");
static_assert(sizeof(Q8_0) == 34 && sizeof(Q8_1) == 36 && sizeof(Q2K) == 84 && sizeof(Q3K) == 110, "");
static_assert(sizeof(Q4K) == 144 && sizeof(Q5K) == 176 && sizeof(Q6K) == 210 && sizeof(Q8K) == 290, "");
static_assert(sizeof(IQ4NL) == 18 && sizeof(IQ4XS) == 136, "");

template <class T> struct OutType;
template <> struct OutType<float> { static constexpr uint32_t ID = GGQ_TYPE_F32; };
template <> struct OutType<f16> { static constexpr uint32_t ID = GGQ_TYPE_F16; };
template <> struct OutType<bf16> { static constexpr uint32_t ID = GGQ_TYPE_BF16; };

inline Result to_result(ggq_status st) {
    if (st == GGQ_OK) return std::nullopt;
    return static_cast<QuantizeError>(st);
}

/// QuantExt<T, N> (lib.rs:98-104); N is DataBlock<Blk>::COUNT.
template <class Blk, class T> struct QuantExt {
    /// Host slices, like the reference: `dst` and `src` are ordinary memory.
    static Result dequantize_slice(Slice<T> dst, Slice<const Blk> src) {
        return to_result(ggq_dequantize_slice(DataBlock<Blk>::ID, OutType<T>::ID, dst.ptr, dst.len,
                                              src.ptr, src.len));
    }
    /// lib.rs:121-133.  Implemented for the blocks the reference implements
    /// (f16, bf16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1, Q8K); others give Unsupported.
    static Result quantize_slice(Slice<Blk> dst, Slice<const T> src) {
        return to_result(ggq_quantize_slice(DataBlock<Blk>::ID, OutType<T>::ID, dst.ptr, dst.len,
                                            src.ptr, src.len));
    }
    /// Device slices on the current HIP device; enqueued on `hip_stream`, not synchronized.
    static Result dequantize_slice_device(Slice<T> dst, Slice<const Blk> src, void *hip_stream = nullptr) {
        return to_result(ggq_dequantize_device(DataBlock<Blk>::ID, OutType<T>::ID, dst.ptr, dst.len,
                                               src.ptr, src.len, hip_stream));
    }
};

}  // namespace ggml_quants
