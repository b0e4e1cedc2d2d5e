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
// (measured 6590-6690 GB/s on 2304- and 8192-byte rows against 6070-6120 with 512 threads x 2 pieces).
#pragma once
#include "kernels.h"

namespace ggq {

// dst row i <- src row perm(i); VEC = bytes per piece: the widest of 16, 4, 2, 1 that divides both addresses and the row size
template <uint32_t VEC>
__global__ __launch_bounds__(STREAM_THREADS) void permute_qk_rows(uint8_t *__restrict__ dst,
                                                              const uint8_t *__restrict__ src,
                                                              uint64_t row_bytes, uint64_t rows, uint64_t dh,
                                                              uint64_t first) {
    // `first`: piece index this launch starts at (a tensor beyond the grid limit takes several launches)
    const uint64_t pieces_per_row = row_bytes / VEC;
    const uint64_t total = pieces_per_row * rows;
    const uint64_t p = first + (uint64_t)blockIdx.x * STREAM_THREADS + threadIdx.x;
    if (p >= total) return;
    const uint64_t row = p / pieces_per_row, col = p % pieces_per_row;
    const uint64_t head = row / dh, in_head = row % dh;
    const uint64_t a = in_head & 1, b = in_head >> 1;
    const uint64_t srow = head * dh + a * (dh / 2) + b;
    if constexpr (VEC == 16) {
        // read once, written once: non-temporal both ways
        const uint4v v = __builtin_nontemporal_load((const uint4v *)(src + srow * row_bytes + col * 16));
        __builtin_nontemporal_store(v, (uint4v *)(dst + row * row_bytes + col * 16));
    } else if constexpr (VEC == 4) {
        *(uint32_t *)(dst + row * row_bytes + col * 4) = *(const uint32_t *)(src + srow * row_bytes + col * 4);
    } else if constexpr (VEC == 2) {
        *(uint16_t *)(dst + row * row_bytes + col * 2) = *(const uint16_t *)(src + srow * row_bytes + col * 2);
    } else {
        dst[row * row_bytes + col] = src[srow * row_bytes + col];
    }
}

}  // namespace ggq
