// rearrange.h -- block-granular row rearrangement on the device.
//
// The reference's merge-linear / split-linear / permute-qk operators
// (xtask/src/utils/operator/merge.rs:224-358, permute_qk.rs:46-69) move whole rows of
// packed blocks around with the third-party crate mem-rearrange; no block is decoded.
// A tensor [c, r] (c = contiguous dimension) is r rows of row_bytes =
// elements_to_bytes([c]) bytes, so:
//   concat / split along rows  = copies of contiguous byte ranges (done with hipMemcpyAsync);
//   permute_qk                 = a row gather: within each head of dh = r / nh rows,
//                                dst row 2b + a  <-  src row a * dh/2 + b   (a in {0,1}, b < dh/2)
//     (tile_le(1, [dh/2, 2, nh]).transpose([2, 1]) of permute_qk.rs:57-59).
// A pure copy: half the bytes are reads, half writes; one-shot grid, 256 threads, one piece per thread
// (measured 6590-6690 GB/s on 2304- and 8192-byte rows against 6070-6120 with 512 threads x 2 pieces);
// with the index arithmetic reduced to multiplications it runs as fast as a plain copy of the same shape
// (6480-6550 GB/s against 6530 in long runs of launches, 6260-6320 before).
#pragma once
#include "kernels.h"

namespace ggq {

// dst row i <- src row perm(i); VEC = bytes per piece: the widest of 16, 4, 2, 1 that divides both addresses and the row size
template <uint32_t VEC>
__device__ __forceinline__ void move_piece(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t to, uint64_t from) {
    if constexpr (VEC == 16) {
        // read once, written once: non-temporal both ways
        const uint4v v = __builtin_nontemporal_load((const uint4v *)(src + from));
        __builtin_nontemporal_store(v, (uint4v *)(dst + to));
    } else if constexpr (VEC == 4) {
        *(uint32_t *)(dst + to) = *(const uint32_t *)(src + from);
    } else if constexpr (VEC == 2) {
        *(uint16_t *)(dst + to) = *(const uint16_t *)(src + from);
    } else {
        dst[to] = src[from];
    }
}

// Tensors of fewer than 2^32 - 256 pieces (64 GiB at 16 bytes a piece; the margin keeps the last workgroup's
// indices from wrapping): 32-bit indices, divisions by multiplication.
// `ppr`: pieces per row; `dhd`: rows per head; `first`: piece this launch starts at (a tensor beyond the grid
// limit takes several launches).
template <uint32_t VEC>
__global__ __launch_bounds__(STREAM_THREADS) void permute_qk_rows(uint8_t *__restrict__ dst,
                                                              const uint8_t *__restrict__ src,
                                                              uint64_t row_bytes, uint32_t total, FastDiv32 ppr,
                                                              FastDiv32 dhd, uint32_t first) {
    const uint32_t p = first + blockIdx.x * STREAM_THREADS + threadIdx.x;
    if (p >= total) return;
    const uint32_t row = ppr.div(p), col = p - row * ppr.d;
    const uint32_t head = dhd.div(row), in_head = row - head * dhd.d;
    const uint32_t srow = head * dhd.d + (in_head & 1u) * (dhd.d >> 1) + (in_head >> 1);
    move_piece<VEC>(dst, src, row * row_bytes + (uint64_t)col * VEC, srow * row_bytes + (uint64_t)col * VEC);
}

// Anything larger: 64-bit indices.
template <uint32_t VEC>
__global__ __launch_bounds__(STREAM_THREADS) void permute_qk_rows_wide(uint8_t *__restrict__ dst,
                                                                   const uint8_t *__restrict__ src,
                                                                   uint64_t row_bytes, uint64_t rows, uint64_t dh,
                                                                   uint64_t first) {
    const uint64_t pieces_per_row = row_bytes / VEC;
    const uint64_t total = pieces_per_row * rows;
    const uint64_t p = first + (uint64_t)blockIdx.x * STREAM_THREADS + threadIdx.x;
    if (p >= total) return;
    const uint64_t row = p / pieces_per_row, col = p % pieces_per_row;
    const uint64_t head = row / dh, in_head = row % dh;
    const uint64_t srow = head * dh + (in_head & 1) * (dh / 2) + (in_head >> 1);
    move_piece<VEC>(dst, src, row * row_bytes + col * VEC, srow * row_bytes + col * VEC);
}

}  // namespace ggq
