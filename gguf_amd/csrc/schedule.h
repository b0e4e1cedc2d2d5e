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
inline Prefetch prefetch_for(uint32_t ty, uint32_t out_ty, bool batch) {
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
    if (!wide) {
        switch (ty) {
        case GGQ_TYPE_Q2K: return {640, 1280};
        case GGQ_TYPE_Q3K: case GGQ_TYPE_Q5K: case GGQ_TYPE_IQ4XS: case GGQ_TYPE_Q5_0: return {1024, 1024};
        case GGQ_TYPE_Q4K: return {768, 1280};
        case GGQ_TYPE_Q6K: return {512, 768};
        case GGQ_TYPE_Q4_0: case GGQ_TYPE_IQ4NL: return {768, 1024};
        case GGQ_TYPE_Q4_1: return {384, 1024};
        default: return {0, 0};
        }
    }
    switch (ty) {
    case GGQ_TYPE_Q2K: case GGQ_TYPE_Q4K: case GGQ_TYPE_IQ4XS: return {768, 1280};
    case GGQ_TYPE_Q3K: return {768, 1536};
    case GGQ_TYPE_Q5K: return {640, 1280};
    case GGQ_TYPE_Q6K: return {1024, 1024};
    case GGQ_TYPE_Q4_1: return {512, 1280};
    default: return {640, 1024};
    }
}

}  // namespace ggq
