// shapes.h -- measurement-only kernels of the A/B harness (tools/ubench): the shipped tile decode with the
// workgroup shape as a template parameter, and the plain memory streams it is compared with.  Not part of the
// library.  (The many other experimental kernels of earlier versions -- table-free batch forms, gated starts, the
// LDS-free Q4_K decoder, chunk-strided stores ... -- were measured, recorded in profiles/EXPERIMENTS.md and removed
// from the tree; git history keeps them.)
#pragma once
#include "kernels.h"

namespace ggq {

// ---- the shipped tile kernel with the workgroup shape as a parameter (THR threads, S stores per thread) ----
template <class F, class O, int THR, int S>
struct SGeom {
    static constexpr uint32_t ELEMS = THR * O::EPL * S;
    static constexpr uint32_t TB = ELEMS / F::N;
    static constexpr uint32_t TBYTES = TB * F::BYTES;
    static constexpr uint32_t SLOTS = (TBYTES + 15) / 16 + 1;
    static constexpr uint32_t LDS_SLOTS = SLOTS + 1;
    static constexpr uint32_t LPT = (SLOTS + THR - 1) / THR;
    static constexpr uint32_t LINES = (TBYTES + 127) / 128 + 1;
    static constexpr uint32_t TPW = THR / LINES;
    static_assert(ELEMS % F::N == 0, "a tile is a whole number of blocks");
    static_assert(TPW >= 1, "tile too large to prefetch with one line per thread");
};

template <class F, class O, int THR, int S>
__global__ __launch_bounds__(THR) void deq_one_s(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                 uint64_t nblocks, Prefetch pf) {
    typedef SGeom<F, O, THR, S> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint32_t tid = threadIdx.x;
    const uint8_t *line = nullptr;
    if (pf.period) {
        const uint32_t ph = (uint32_t)(t % pf.period);
        const uint32_t nb = (pf.period + G::TPW - 1) / G::TPW;
        if (ph < nb) {
            const uint32_t j = tid / G::LINES, l = tid % G::LINES;
            const uint32_t k = ph * G::TPW + j;
            if (j < G::TPW && k < pf.period) {
                const uint64_t tt = t - ph + pf.lead + k;
                if ((tt + 1) * G::TB <= nblocks) {
                    const uint8_t *base = src + tt * G::TBYTES;
                    const uint8_t *ln = (const uint8_t *)((uintptr_t)base & ~(uintptr_t)127) + 128u * l;
                    if (ln < base + G::TBYTES) line = ln;
                }
            }
        }
    }
    const uint32_t z = pf.wait ? group_wait(dst + ((t >> pf.wait) << pf.wait) * G::ELEMS) : 0u;
    const uint8_t *ts = src + t * G::TBYTES + z;
    typename O::T *out = dst + t * G::ELEMS;
    const uint32_t off0 = (uint32_t)((uintptr_t)ts & 15u);
    const uint4v *base16 = (const uint4v *)(ts - off0);
    const uint32_t nslots = (off0 + nblk * F::BYTES + 15u) >> 4;
#pragma unroll
    for (uint32_t k = 0; k < G::LPT; ++k) {
        uint32_t sl = tid + THR * k;
        if (sl < nslots) lds[sl] = base16[sl];
    }
    uint32_t touch = 0;
    if (line != nullptr) touch = *(const uint32_t *)line;
    __syncthreads();
    LdsBytes r{(const uint32_t *)lds};
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)S; ++i) {
        uint32_t el = (i * THR + tid) * O::EPL;
        uint32_t bt = el / F::N, e = el % F::N;
        if (bt < nblk) {
            uint32_t b = off0 + bt * F::BYTES;
            const Scale sc = F::scale(r, b, e);
            float y[O::EPL];
            F::template vals<O::EPL / 4>(r, b, e, sc, y);
            O::template store<true>(out + el, y);
        }
    }
    asm volatile("" ::"v"(touch));
}

// ---- F16 / BF16 sources as 64-byte "blocks" of 32 elements through the tile kernel (measurement: the library widens
// them with widen_vec, no LDS).  The decode is the conversion itself; the scale is unused.
template <bool BF16SRC>
struct HalfBlocks {
    static constexpr uint32_t N = 32, BYTES = 64;
    template <class R>
    static __device__ __forceinline__ Scale scale(const R &, uint32_t, uint32_t) { return {0.0f, 0.0f}; }
    template <int G, class R>
    static __device__ __forceinline__ void vals(const R &r, uint32_t b, uint32_t e, Scale, float (&y)[4 * G]) {
        uint32_t q[2 * G];
        r.template words<2 * G>(b + 2 * e, q);
#pragma unroll
        for (int g = 0; g < 2 * G; ++g) {
            if (BF16SRC) {
                y[2 * g] = __uint_as_float(q[g] << 16);
                y[2 * g + 1] = __uint_as_float(q[g] & 0xFFFF0000u);
            } else {
                y[2 * g] = h2f(q[g] & 0xFFFFu);
                y[2 * g + 1] = h2f(q[g] >> 16);
            }
        }
    }
    template <class R>
    static __device__ __forceinline__ void dec4(const R &r, uint32_t b, uint32_t e, float (&y)[4]) {
        vals<1>(r, b, e, scale(r, b, e), y);
    }
};

// ---- widen_vec (f16 -> f32) with the steps per thread and a group start as parameters ----------------------
template <int S>
__global__ __launch_bounds__(STREAM_THREADS) void widen_s(const uint2v *__restrict__ src, float *__restrict__ dst,
                                                          uint64_t n4, uint32_t wait) {
    if (wait) {
        const uint64_t first = (((uint64_t)blockIdx.x >> wait) << wait) * STREAM_THREADS * S * 4;
        src += group_wait(dst + first);
    }
    const uint64_t base = (uint64_t)blockIdx.x * STREAM_THREADS * S + threadIdx.x;
    uint2v v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = base + k * STREAM_THREADS;
        if (i < n4) v[k] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = base + k * STREAM_THREADS;
        if (i >= n4) continue;
        float y[4] = {h2f(v[k].x & 0xFFFFu), h2f(v[k].x >> 16), h2f(v[k].y & 0xFFFFu), h2f(v[k].y >> 16)};
        OutF32::store<true>(dst + 4 * i, y);
    }
}

// ---- plain streams: one-shot grids, THR threads, one 16-byte piece per thread ------------------------
template <int THR>
__global__ __launch_bounds__(THR) void k_fill(uint4v *dst, uint64_t n16) {
    const uint64_t i = (uint64_t)blockIdx.x * THR + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(uint4v{1u, 2u, 3u, (uint32_t)threadIdx.x}, dst + i);
}
template <int THR>
__global__ __launch_bounds__(THR) void k_copy(uint4v *dst, const uint4v *src, uint64_t n16) {
    const uint64_t i = (uint64_t)blockIdx.x * THR + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
// pure read: sums its pieces; the store never happens (the pattern cannot occur in the test data)
template <int THR, int S, bool NT = true>
__global__ __launch_bounds__(THR) void k_read(uint4v *sink, const uint4v *src, uint64_t n16) {
    const uint64_t base = (uint64_t)blockIdx.x * (THR * S) + threadIdx.x;
    uint4v v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const uint64_t i = base + (uint64_t)k * THR;
        v[k] = i < n16 ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : uint4v{0, 0, 0, 0};
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < S; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    if (acc == 0x7FC12345u && v[0].x == 0x12345678u) sink[0] = v[0];
}

}  // namespace ggq
