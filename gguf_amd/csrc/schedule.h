// schedule.h -- burst prefetch schedule of the decode kernels, shared by the library and ubench.
#pragma once
#include "../../include/ggq.h"
#include "kernels.h"

namespace ggq {

// Burst prefetch schedule {period, lead}, in tiles, per (block type, output width, kernel):
// the best region of period x lead sweeps on an MI355X, 1 Gi elements per launch
// (`ubench 1024 5 "sw "` / `sw16` for the single-tensor kernel, `swb <fmt>` / `swb16` for the
// batch kernel, whose duty workgroups first look the far tiles up in the tile table and so like
// a longer period and lead).  f32 output gains 13-30 % over no prefetch (the 8-bit formats
// 5-6 %).  With 2-byte output the reads are a larger share of the traffic; bunching them gains
// 2-20 % for the 2- to 5-bit formats and costs 3-7 % for the 8-bit ones, where it stays off.
// period 1 = every workgroup touches its own tile `lead` ahead (no burst).
// A third field k > 0 switches the group start on (groups of 2^k workgroups).
inline Prefetch prefetch_for(uint32_t ty, uint32_t out_ty, bool batch, uint64_t elems = ~0ull) {
    const bool wide = out_ty == GGQ_TYPE_F32;
    if (batch) {
        if (!wide) {
            switch (ty) {
            case GGQ_TYPE_Q2K: return {1280, 2048};
            case GGQ_TYPE_Q3K: return {1024, 1536};
            case GGQ_TYPE_Q4K: return {768, 1536};
            case GGQ_TYPE_Q5K: return {1, 1280};
            case GGQ_TYPE_IQ4NL: case GGQ_TYPE_IQ4XS: return {384, 1536};
            case GGQ_TYPE_Q4_0: return {1024, 1280};
            default: return {0, 0};
            }
        }
        switch (ty) {
        case GGQ_TYPE_Q4K: case GGQ_TYPE_IQ4XS: return {1536, 2048};
        case GGQ_TYPE_Q5K: return {1280, 2048};
        case GGQ_TYPE_Q3K: return {1024, 2048};
        case GGQ_TYPE_Q6K: return {1024, 1536};
        case GGQ_TYPE_Q8K: case GGQ_TYPE_Q8_0: case GGQ_TYPE_Q8_1: return {1024, 1280};
        default: return {1280, 1536};
        }
    }
    // Single-tensor kernel.  Where the packed input is a fifth of the traffic or more -- every 2-byte output except
    // Q2_K and Q3_K, and the 8-bit formats with 4-byte output -- the group start (third field: groups of 2^k
    // workgroups, see group_wait in kernels.h) replaces the burst; {1, lead} then lets every workgroup touch its
    // own tile `lead` ahead.  Measured against the burst schedules they replace (`ubench 1024 7 gw <fmt>`, 3 and 30
    // launches per sample, % of 8 TB/s): Q8_0 -> f32 83.9 / 84.0 against 81.1 / 81.2, Q8_1 84.6 against 81.1, Q8K
    // 84.5 against 80.7; -> f16: Q4_K 83.0 / 82.7 against 80.5, Q6_K 85.9 / 85.1 against 77.9 / 78.3, Q5_K 84.5 /
    // 83.4 against 78.5, Q8K 83.5 / 84.0 against 77.0, IQ4_XS 81.7 / 80.8 against 80.7 / 79.4, IQ4_NL 83.3 against
    // 80.8, Q4_0 83.4 / 83.6 against 80.6 / 80.9, Q4_1 84.3 against 79.3, Q5_0 85.1 against 78.3, Q5_1 85.3 against
    // 79.0, Q8_0 83.6 against 77.2, Q8_1 83.9 against 76.6.  With a smaller share of reads the burst wins by 4-17 %.
    // The batch kernel does not take it: with the table's record loads in the stream the same wait LOSES 3-18 %.
    // Those figures are long runs over 1 Gi elements.  A tensor decoded once, its input cold
    // (tools/probe_cold.py), gains +2...+11 % at 512 Mi elements, +-3 % at
    // 128 Mi, and below that no schedule at all is as good or better (the old bursts lose up to 5 % there): tensors
    // under 64 Mi elements run these kernels without burst and without group start, except that the 8-bit formats
    // with 4-byte output keep their burst.
    const bool large = elems >= (64ull << 20);
    if (!wide) {
        switch (ty) {
        case GGQ_TYPE_Q2K: return {640, 1280};
        case GGQ_TYPE_Q3K: return {1024, 1024};
        default: break;
        }
        if (!large) return {0, 0};
        switch (ty) {
        // groups of 64 up to 16 Gi elements, of 16 beyond: measured on single runs of Q4_K -> f16 (tools/
        // probe_size_scaling.py, % of 8 TB/s at 2 / 4 / 8 / 16 / 32 / 48 Gi elements): groups of 64 83.1 / 83.7 / 83.7 /
        // 83.3-83.8 / 80.1-81.0 / 80.6, groups of 16 82.8 / 82.7 / 82.9 / 83.0-83.5 / 83.6-83.8 / 83.8
        case GGQ_TYPE_Q4K: case GGQ_TYPE_IQ4XS: case GGQ_TYPE_IQ4NL: case GGQ_TYPE_Q4_0:
            return elems > (16ull << 30) ? Prefetch{1, 2048, 4} : Prefetch{1, 2048, 6};
        case GGQ_TYPE_Q5K: return {1, 1280, 6};
        case GGQ_TYPE_Q4_1: case GGQ_TYPE_Q5_0: return {1, 1280, 5};
        case GGQ_TYPE_Q6K: case GGQ_TYPE_Q5_1: return {1, 1280, 4};
        default: return {0, 0, 3};                               // Q8_0, Q8_1, Q8K
        }
    }
    switch (ty) {
    case GGQ_TYPE_Q2K: case GGQ_TYPE_Q4K: case GGQ_TYPE_IQ4XS: return {768, 1280};
    case GGQ_TYPE_Q3K: return {768, 1536};
    case GGQ_TYPE_Q5K: return {640, 1280};
    case GGQ_TYPE_Q6K: return {1024, 1024};
    case GGQ_TYPE_Q4_1: return {512, 1280};
    case GGQ_TYPE_Q8_0: case GGQ_TYPE_Q8_1: case GGQ_TYPE_Q8K: return large ? Prefetch{0, 0, 5} : Prefetch{640, 1024};
    default: return {640, 1024};
    }
}

}  // namespace ggq
