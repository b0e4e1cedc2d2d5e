// kernels.h -- dequantize kernels for gfx950 (MI355X).
//
// Work decomposition ("wave tiles").  The unit of work is a tile of
// TILE_ELEMS = 4096 output elements (16 K-quant blocks or 128 legacy blocks),
// owned by ONE 64-lane wave:
//   1. the wave copies the tile's packed bytes HBM -> registers with aligned
//      16-byte loads (consecutive lanes, consecutive 16-byte slots; the source
//      pointer may have any alignment, the slots are those of the enclosing
//      16-byte grid), then registers -> its private LDS slice;
//   2. the loads for the wave's NEXT tile are issued before decoding starts,
//      so their HBM latency hides under the decode and the stores;
//   3. each lane decodes EPL consecutive elements per step straight from LDS
//      (bit unpack, per-sub-block scale) and stores 16 bytes; the 64 lanes of a
//      step write one contiguous 1 KiB run, so every store instruction is a
//      fully coalesced burst of 8 whole 128-byte lines.
// No workgroup barrier is needed: a wave only reads LDS bytes it wrote itself,
// and the LDS executes one wave's instructions in order.
//
// The kernel is bound by HBM traffic (about 88 % of it the output stores), not
// by arithmetic: there is no matrix work here.
#pragma once
#include "formats.h"

namespace ggq {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

constexpr uint32_t TILE_ELEMS = 4096;
constexpr uint32_t WAVES_PER_WG = 4;
constexpr uint32_t WG_THREADS = 64 * WAVES_PER_WG;

// ---- output element types ---------------------------------------------------
struct OutF32 {
    typedef float T;
    static constexpr uint32_t EPL = 4;   // elements per lane per step (16 bytes)
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, const float (&a)[4]) {
        float4v v = {a[0], a[1], a[2], a[3]};
        if (NT) __builtin_nontemporal_store(v, (float4v *)p);
        else *(float4v *)p = v;
    }
};
// f32 result narrowed per element, round to nearest even
// (ggml-quants/src/lib.rs:70-73)
struct OutF16 {
    typedef uint16_t T;
    static constexpr uint32_t EPL = 8;
    static __device__ __forceinline__ uint32_t pk(float a, float b) {
        return (uint32_t)__half_as_ushort(__float2half_rn(a)) |
               ((uint32_t)__half_as_ushort(__float2half_rn(b)) << 16);
    }
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, const float (&a)[4], const float (&b)[4]) {
        uint4v v = {pk(a[0], a[1]), pk(a[2], a[3]), pk(b[0], b[1]), pk(b[2], b[3])};
        if (NT) __builtin_nontemporal_store(v, (uint4v *)p);
        else *(uint4v *)p = v;
    }
};
// (ggml-quants/src/lib.rs:86-89)
struct OutBF16 {
    typedef uint16_t T;
    static constexpr uint32_t EPL = 8;
    static __device__ __forceinline__ uint32_t pk(float a, float b) {
        __bf16 x = (__bf16)a, y = (__bf16)b;
        return (uint32_t)__builtin_bit_cast(unsigned short, x) |
               ((uint32_t)__builtin_bit_cast(unsigned short, y) << 16);
    }
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, const float (&a)[4], const float (&b)[4]) {
        uint4v v = {pk(a[0], a[1]), pk(a[2], a[3]), pk(b[0], b[1]), pk(b[2], b[3])};
        if (NT) __builtin_nontemporal_store(v, (uint4v *)p);
        else *(uint4v *)p = v;
    }
};

// One tensor (or contiguous block range) of a batch.  tile_begin is the index
// of its first tile in the batch-wide tile numbering.
struct Segment {
    const uint8_t *src;   // packed blocks, any alignment
    void *dst;            // 16-byte aligned
    uint64_t blocks;
    uint64_t tile_begin;
};

template <class F>
struct TileGeom {
    static constexpr uint32_t TB = TILE_ELEMS / F::N;         // blocks per tile
    static constexpr uint32_t TBYTES = TB * F::BYTES;         // packed bytes per tile
    static constexpr uint32_t SLOTS = TBYTES / 16 + 1;        // 16-byte slots a tile can touch
    static constexpr uint32_t LDS_SLOTS = SLOTS + 1;          // + slack for the funnel read
    static constexpr uint32_t LPL = (SLOTS + 63) / 64;        // loads per lane
    static_assert(TBYTES % 16 == 0, "tile bytes must be whole slots");
};

// Issue the aligned 16-byte loads of one tile into registers.
//   base16 : source pointer rounded down to 16 bytes
//   slot0  : first slot of the tile, nslots : slots that hold tile bytes
template <class F>
__device__ __forceinline__ void tile_fetch(uint4v (&reg)[TileGeom<F>::LPL], const uint4v *base16,
                                           uint64_t slot0, uint32_t nslots, uint32_t lane) {
#pragma unroll
    for (uint32_t k = 0; k < TileGeom<F>::LPL; ++k) {
        uint32_t s = lane + 64 * k;
        if (s < nslots) reg[k] = __builtin_nontemporal_load(base16 + slot0 + s);
    }
}

template <class F>
__device__ __forceinline__ void tile_stage(uint4v *lds, const uint4v (&reg)[TileGeom<F>::LPL],
                                           uint32_t nslots, uint32_t lane) {
#pragma unroll
    for (uint32_t k = 0; k < TileGeom<F>::LPL; ++k) {
        uint32_t s = lane + 64 * k;
        if (s < nslots) lds[s] = reg[k];
    }
}

// Decode one staged tile and store it.  `nblk` = blocks of the tile that exist
// (TB except for a tensor's last tile), off0 = byte offset of the tile's first
// block inside the LDS image (0..15).
template <class F, class O, bool NT>
__device__ __forceinline__ void tile_decode(const uint4v *lds, uint32_t off0, uint32_t nblk,
                                            typename O::T *out, uint32_t lane) {
    LdsBytes r{(const uint32_t *)lds};
    constexpr uint32_t STEPS = TILE_ELEMS / (64 * O::EPL);
#pragma unroll 4
    for (uint32_t i = 0; i < STEPS; ++i) {
        uint32_t el = (i * 64 + lane) * O::EPL;
        uint32_t bt = el / F::N, e = el % F::N;
        if (bt < nblk) {
            uint32_t b = off0 + bt * F::BYTES;
            float y0[4];
            F::dec4(r, b, e, y0);
            if constexpr (O::EPL == 4) {
                O::template store<NT>(out + el, y0);
            } else {
                float y1[4];
                F::dec4(r, b, e + 4, y1);
                O::template store<NT>(out + el, y0, y1);
            }
        }
    }
}

// ---- one tensor per launch --------------------------------------------------
// Each wave walks tiles wave_id, wave_id + nwaves, ... of the tensor.
template <class F, class O, bool NT>
__global__ __launch_bounds__(WG_THREADS) void deq_tiles(const uint8_t *__restrict__ src,
                                                        typename O::T *__restrict__ dst,
                                                        uint64_t nblocks) {
    typedef TileGeom<F> G;
    __shared__ uint4v lds_all[WAVES_PER_WG][G::LDS_SLOTS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4v *lds = lds_all[wv];

    const uint64_t head = (uint64_t)src & 15u;
    const uint4v *base16 = (const uint4v *)(src - head);
    const uint64_t ntiles = (nblocks + G::TB - 1) / G::TB;
    const uint64_t nwaves = (uint64_t)gridDim.x * WAVES_PER_WG;
    uint64_t t = (uint64_t)blockIdx.x * WAVES_PER_WG + wv;
    if (t >= ntiles) return;

    uint4v reg[G::LPL];
    // geometry of tile t
    uint64_t byte0 = head + t * G::TBYTES;
    uint64_t rem = nblocks - t * G::TB;
    uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    uint32_t nslots = (uint32_t)(((byte0 & 15u) + (uint64_t)nblk * F::BYTES + 15u) >> 4);
    tile_fetch<F>(reg, base16, byte0 >> 4, nslots, lane);

    for (;;) {
        tile_stage<F>(lds, reg, nslots, lane);
        const uint32_t cur_nblk = nblk;
        const uint32_t off0 = (uint32_t)(byte0 & 15u);
        const uint64_t cur_t = t;
        t += nwaves;
        const bool more = t < ntiles;
        if (more) {
            byte0 = head + t * G::TBYTES;
            rem = nblocks - t * G::TB;
            nblk = rem < G::TB ? (uint32_t)rem : G::TB;
            nslots = (uint32_t)(((byte0 & 15u) + (uint64_t)nblk * F::BYTES + 15u) >> 4);
            tile_fetch<F>(reg, base16, byte0 >> 4, nslots, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        tile_decode<F, O, NT>(lds, off0, cur_nblk, dst + cur_t * TILE_ELEMS, lane);
        if (!more) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- many tensors per launch -------------------------------------------------
// Tiles of all segments are numbered consecutively; a wave finds the segment of
// its tile by binary search over tile_begin (wave-uniform, scalar loads).
template <class F, class O, bool NT>
__global__ __launch_bounds__(WG_THREADS) void deq_tiles_batch(const Segment *__restrict__ seg,
                                                              uint32_t nseg, uint64_t ntiles) {
    typedef TileGeom<F> G;
    __shared__ uint4v lds_all[WAVES_PER_WG][G::LDS_SLOTS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4v *lds = lds_all[wv];
    const uint64_t nwaves = (uint64_t)gridDim.x * WAVES_PER_WG;
    uint64_t t = (uint64_t)blockIdx.x * WAVES_PER_WG + wv;
    if (t >= ntiles) return;

    // locate(): fill the geometry of tile `tt`
    uint32_t si = 0;
    const uint4v *base16;
    uint64_t byte0;
    uint32_t nblk, nslots;
    typename O::T *out;
    auto locate = [&](uint64_t tt) {
        // segments are walked forwards only, so resume the search at si
        uint32_t lo = si, hi = nseg;   // invariant: tile_begin[lo] <= tt
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (seg[mid].tile_begin <= tt) lo = mid; else hi = mid;
        }
        si = lo;
        const uint8_t *s = seg[lo].src;
        uint64_t lt = tt - seg[lo].tile_begin;
        uint64_t head = (uint64_t)s & 15u;
        base16 = (const uint4v *)(s - head);
        byte0 = head + lt * G::TBYTES;
        uint64_t rem = seg[lo].blocks - lt * G::TB;
        nblk = rem < G::TB ? (uint32_t)rem : G::TB;
        nslots = (uint32_t)(((byte0 & 15u) + (uint64_t)nblk * F::BYTES + 15u) >> 4);
        out = (typename O::T *)seg[lo].dst + lt * TILE_ELEMS;
    };

    uint4v reg[G::LPL];
    locate(t);
    tile_fetch<F>(reg, base16, byte0 >> 4, nslots, lane);
    for (;;) {
        tile_stage<F>(lds, reg, nslots, lane);
        const uint32_t cur_nblk = nblk;
        const uint32_t off0 = (uint32_t)(byte0 & 15u);
        typename O::T *cur_out = out;
        t += nwaves;
        const bool more = t < ntiles;
        if (more) {
            locate(t);
            tile_fetch<F>(reg, base16, byte0 >> 4, nslots, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        tile_decode<F, O, NT>(lds, off0, cur_nblk, cur_out, lane);
        if (!more) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- F16 / BF16 sources: plain widening, no LDS -------------------------------
// (ggml-quants/src/structs/half.rs:19-21, 35-37).  Lane l of a step converts
// elements [4l, 4l+4) of a 256-element run from one 8-byte load and writes one
// 16-byte store (f32 out) so loads and stores are both contiguous per wave.
// Requires src 8-byte aligned and dst 16-byte aligned; n4 = elements / 4.
template <bool BF16SRC, class O, bool NT>
__global__ __launch_bounds__(WG_THREADS) void widen_vec(const uint2 *__restrict__ src,
                                                        typename O::T *__restrict__ dst, uint64_t n4) {
    uint64_t i = (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * WG_THREADS;
    for (; i < n4; i += stride) {
        uint2 v = src[i];
        float y[4];
        if (BF16SRC) {
            y[0] = __uint_as_float(v.x << 16); y[1] = __uint_as_float(v.x & 0xFFFF0000u);
            y[2] = __uint_as_float(v.y << 16); y[3] = __uint_as_float(v.y & 0xFFFF0000u);
        } else {
            y[0] = h2f(v.x & 0xFFFFu); y[1] = h2f(v.x >> 16);
            y[2] = h2f(v.y & 0xFFFFu); y[3] = h2f(v.y >> 16);
        }
        if constexpr (O::EPL == 4) {
            O::template store<NT>(dst + 4 * i, y);
        } else {
            // 2-byte outputs: 8 bytes per lane
            uint2 o = {O::pk(y[0], y[1]), O::pk(y[2], y[3])};
            *(uint2 *)(dst + 4 * i) = o;
        }
    }
}

// Element-wise tail / unaligned fallback for F16 / BF16 sources.
template <bool BF16SRC, class O>
__global__ __launch_bounds__(WG_THREADS) void widen_scalar(const uint16_t *__restrict__ src,
                                                           typename O::T *__restrict__ dst,
                                                           uint64_t first, uint64_t n) {
    uint64_t i = first + (uint64_t)blockIdx.x * WG_THREADS + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * WG_THREADS;
    for (; i < n; i += stride) {
        uint32_t h = src[i];
        float f = BF16SRC ? __uint_as_float(h << 16) : h2f(h);
        if constexpr (O::EPL == 4) dst[i] = f;
        else dst[i] = (uint16_t)(O::pk(f, 0.0f) & 0xFFFFu);
    }
}

}  // namespace ggq
