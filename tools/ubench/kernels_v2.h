// kernels_v2.h -- experimental kernel variants measured by ubench against the
// shipped ones in kernels.h.  A variant that wins moves into kernels.h.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "kernels.h"
#include "rearrange.h"

namespace ggq {

// ---- the first design, kept for A/B: persistent waves walking 4096-element tiles
// ---- with a grid-stride loop and register prefetch of the next tile.  It loses to
// ---- the one-shot grid (kernels.h) by 15-20 %: looping workgroups drift apart and
// ---- their stores stop forming a compact advancing window.
constexpr uint32_t TILE_ELEMS = 4096;
constexpr uint32_t WAVES_PER_WG = 4;
constexpr uint32_t OLD_WG_THREADS = 64 * WAVES_PER_WG;

template <class F, uint32_t TE = TILE_ELEMS>
struct OldGeom {
    static constexpr uint32_t ELEMS = TE;
    static constexpr uint32_t TB = TE / F::N;                 // blocks per tile
    static constexpr uint32_t TBYTES = TB * F::BYTES;         // packed bytes per tile
    static constexpr uint32_t SLOTS = (TBYTES + 15) / 16 + 1; // 16-byte slots a tile can touch
    static constexpr uint32_t LDS_SLOTS = SLOTS + 1;          // + slack for the funnel read
    static constexpr uint32_t LPL = (SLOTS + 63) / 64;        // loads per lane
    static_assert(TE % F::N == 0 && TE % 512 == 0, "a tile is whole blocks and whole wave steps");
};

// Issue the aligned 16-byte loads of one tile into registers.
//   base16 : source pointer rounded down to 16 bytes
//   slot0  : first slot of the tile, nslots : slots that hold tile bytes
template <class F>
__device__ __forceinline__ void tile_fetch(uint4v (&reg)[OldGeom<F>::LPL], const uint4v *base16,
                                           uint64_t slot0, uint32_t nslots, uint32_t lane) {
#pragma unroll
    for (uint32_t k = 0; k < OldGeom<F>::LPL; ++k) {
        uint32_t s = lane + 64 * k;
        if (s < nslots) reg[k] = __builtin_nontemporal_load(base16 + slot0 + s);
    }
}

template <class F>
__device__ __forceinline__ void tile_stage(uint4v *lds, const uint4v (&reg)[OldGeom<F>::LPL],
                                           uint32_t nslots, uint32_t lane) {
#pragma unroll
    for (uint32_t k = 0; k < OldGeom<F>::LPL; ++k) {
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
__global__ __launch_bounds__(OLD_WG_THREADS) void deq_tiles(const uint8_t *__restrict__ src,
                                                        typename O::T *__restrict__ dst,
                                                        uint64_t nblocks) {
    typedef OldGeom<F> G;
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

// Q4_K -> f32 without LDS: a wave takes U consecutive blocks per step.  The 16-byte
// block header (d, dmin, 12 scale bytes) sits at a wave-uniform address, so it is
// fetched through the scalar cache and the 6-bit scale unpack runs on the scalar
// unit; each lane then picks its sub-block's (scale, min) and decodes four values
// from one dword of qs.  Needs src 4-byte aligned (scalar loads).
template <bool NT, int U>
__global__ __launch_bounds__(OLD_WG_THREADS) void q4k_direct(const uint8_t *__restrict__ src,
                                                         float *__restrict__ dst, uint64_t nblocks) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * WAVES_PER_WG;
    const uint64_t w = (uint64_t)blockIdx.x * WAVES_PER_WG + wv;
    const uint32_t is = lane >> 3;
    const bool low = is < 4;
    const uint32_t sh8 = 8 * (is & 3), nsh = 4 * (is & 1);
    const uint32_t qoff = 16 + 32 * (lane >> 4) + 4 * (lane & 7);

    for (uint64_t b0 = w * U; b0 < nblocks; b0 += nwaves * U) {
        uint4v hdr[U];
        uint32_t q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (b0 + u < nblocks) {
                const uint8_t *p = src + (b0 + u) * 144;
                hdr[u] = *(const uint4v *)p;
                q[u] = __builtin_nontemporal_load((const uint32_t *)(p + qoff));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (b0 + u < nblocks) {
                const uint32_t a0 = hdr[u].y, a1 = hdr[u].z, a2 = hdr[u].w;
                const float d = h2f(hdr[u].x & 0xFFFFu), dmin = h2f(hdr[u].x >> 16);
                const uint32_t sc_lo = a0 & 0x3F3F3F3Fu, mn_lo = a1 & 0x3F3F3F3Fu;
                const uint32_t sc_hi = (a2 & 0x0F0F0F0Fu) | ((a0 >> 2) & 0x30303030u);
                const uint32_t mn_hi = ((a2 >> 4) & 0x0F0F0F0Fu) | ((a1 >> 2) & 0x30303030u);
                const uint32_t scw = (low ? sc_lo : sc_hi) >> sh8, mnw = (low ? mn_lo : mn_hi) >> sh8;
                const float d1 = __fmul_rn(d, ub<0>(scw)), m1 = __fmul_rn(dmin, ub<0>(mnw));
                const uint32_t wq = (q[u] >> nsh) & 0x0F0F0F0Fu;
                float y[4];
                y[0] = __fsub_rn(__fmul_rn(d1, ub<0>(wq)), m1);
                y[1] = __fsub_rn(__fmul_rn(d1, ub<1>(wq)), m1);
                y[2] = __fsub_rn(__fmul_rn(d1, ub<2>(wq)), m1);
                y[3] = __fsub_rn(__fmul_rn(d1, ub<3>(wq)), m1);
                OutF32::store<NT>(dst + (b0 + u) * 256 + 4 * lane, y);
            }
        }
    }
}

// One tile of TE elements per wave, no loop: the grid covers every tile, so the
// hardware's in-order workgroup dispatch keeps all in-flight stores inside one
// compact, steadily advancing window of the output.
template <class F, class O, bool NT, uint32_t TE>
__global__ __launch_bounds__(OLD_WG_THREADS) void deq_oneshot(const uint8_t *__restrict__ src,
                                                          typename O::T *__restrict__ dst,
                                                          uint64_t nblocks) {
    typedef OldGeom<F, TE> G;
    __shared__ uint4v lds_all[WAVES_PER_WG][G::LDS_SLOTS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4v *lds = lds_all[wv];
    const uint64_t head = (uint64_t)src & 15u;
    const uint4v *base16 = (const uint4v *)(src - head);
    const uint64_t ntiles = (nblocks + G::TB - 1) / G::TB;
    const uint64_t t = (uint64_t)blockIdx.x * WAVES_PER_WG + wv;
    if (t >= ntiles) return;
    const uint64_t byte0 = head + t * G::TBYTES;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint32_t nslots = (uint32_t)(((byte0 & 15u) + (uint64_t)nblk * F::BYTES + 15u) >> 4);
#pragma unroll
    for (uint32_t k = 0; k < G::LPL; ++k) {
        uint32_t s = lane + 64 * k;
        if (s < nslots) lds[s] = __builtin_nontemporal_load(base16 + (byte0 >> 4) + s);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    LdsBytes r{(const uint32_t *)lds};
    const uint32_t off0 = (uint32_t)(byte0 & 15u);
    typename O::T *out = dst + t * TE;
    constexpr uint32_t STEPS = TE / (64 * O::EPL);
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

// One tile of TE elements per WORKGROUP, no loop: all 256 threads stage the tile,
// one barrier, then every step the workgroup writes one contiguous 4 KiB run.
// MODE: 0 normal, 1 no global loads, 2 no stores, 4 plain (not nt) loads.
// PF > 0: also touch the packed bytes of the tile PF workgroups ahead (a load whose
// value is never used), so that workgroup finds them in L2.
template <class F, class O, bool NT, uint32_t TE, uint32_t THR = 256, int MODE = 0, int MINW = 1, int PF = 0>
__global__ __launch_bounds__(THR, MINW) void deq_wgtile(const uint8_t *__restrict__ src,
                                                         typename O::T *__restrict__ dst,
                                                         uint64_t nblocks) {
    constexpr uint32_t TB = TE / F::N, TBYTES = TB * F::BYTES;
    constexpr uint32_t SLOTS = (TBYTES + 15) / 16 + 1;
    constexpr uint32_t LPT = (SLOTS + THR - 1) / THR;
    __shared__ uint4v lds[SLOTS + 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t head = (uint64_t)src & 15u;
    const uint4v *base16 = (const uint4v *)(src - head);
    const uint64_t t = blockIdx.x;
    const uint64_t byte0 = head + t * TBYTES;
    const uint64_t rem = nblocks - t * TB;
    const uint32_t nblk = rem < TB ? (uint32_t)rem : TB;
    const uint32_t nslots = (uint32_t)(((byte0 & 15u) + (uint64_t)nblk * F::BYTES + 15u) >> 4);
#pragma unroll
    for (uint32_t k = 0; k < LPT; ++k) {
        uint32_t s = tid + THR * k;
        if (MODE == 4) {
            if (s < nslots) lds[s] = base16[(byte0 >> 4) + s];
        } else if (MODE != 1) {
            if (s < nslots) lds[s] = __builtin_nontemporal_load(base16 + (byte0 >> 4) + s);
        } else {
            if (s < nslots) lds[s] = uint4v{0x3c003c00u + s, s * 77u, s * 1234567u, ~s};
        }
    }
    uint32_t pf = 0;
    if constexpr (PF > 0) {
        // one dword per 128-byte line of the future tile
        const uint64_t pb = ((head + (t + PF) * TBYTES) & ~(uint64_t)127) + 128ull * tid;
        if (128u * tid < TBYTES + 128 && (t + PF + 1) * TB <= nblocks)
            pf = *(const uint32_t *)((const uint8_t *)base16 + pb);
    }
    __syncthreads();
    LdsBytes r{(const uint32_t *)lds};
    const uint32_t off0 = (uint32_t)(byte0 & 15u);
    typename O::T *out = dst + t * TE;
    constexpr uint32_t STEPS = TE / (THR * O::EPL);
    static_assert(STEPS >= 1, "tile too small for the workgroup");
#pragma unroll 4
    for (uint32_t i = 0; i < STEPS; ++i) {
        uint32_t el = (i * THR + tid) * O::EPL;
        uint32_t bt = el / F::N, e = el % F::N;
        if (bt < nblk) {
            uint32_t b = off0 + bt * F::BYTES;
            float y0[4];
            F::dec4(r, b, e, y0);
            if constexpr (MODE == 2) {
                asm volatile("" ::"v"(y0[0]), "v"(y0[1]), "v"(y0[2]), "v"(y0[3]));
            } else if constexpr (O::EPL == 4) {
                O::template store<NT>(out + el, y0);
            } else {
                float y1[4];
                F::dec4(r, b, e + 4, y1);
                O::template store<NT>(out + el, y0, y1);
            }
        }
    }
    if constexpr (PF > 0) asm volatile("" ::"v"(pf));
}

// Burst prefetch experiment.  Reads trickling into HBM between the writes cost far
// more than their bytes (read/write turnarounds).  Here the reads are bunched: the first
// BURST workgroups of every PERIOD touch, together, ALL packed bytes of the tiles
// of period +K (one dword per 128-byte line, 512 lines per workgroup), which pulls them
// into the memory-side Infinity Cache; NEAR > 0 adds the per-workgroup L2 prefetch.
template <class F, class O, bool NT, uint32_t TE, uint32_t THR, uint32_t PERIOD, uint32_t K, int NEAR>
__global__ __launch_bounds__(THR) void deq_burst(const uint8_t *__restrict__ src,
                                                 typename O::T *__restrict__ dst, uint64_t nblocks) {
    constexpr uint32_t TB = TE / F::N, TBYTES = TB * F::BYTES;
    constexpr uint32_t SLOTS = (TBYTES + 15) / 16 + 1;
    constexpr uint32_t LPT = (SLOTS + THR - 1) / THR;
    constexpr uint64_t PBYTES = (uint64_t)PERIOD * TBYTES;
    constexpr uint32_t BURST = (uint32_t)((PBYTES + THR * 128 - 1) / (THR * 128));
    __shared__ uint4v lds[SLOTS + 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t head = (uint64_t)src & 15u;
    const uint4v *base16 = (const uint4v *)(src - head);
    const uint64_t t = blockIdx.x;
    const uint64_t byte0 = head + t * TBYTES;
    const uint64_t rem = nblocks - t * TB;
    const uint32_t nblk = rem < TB ? (uint32_t)rem : TB;
    const uint32_t nslots = (uint32_t)(((byte0 & 15u) + (uint64_t)nblk * F::BYTES + 15u) >> 4);
    const uint64_t total = nblocks * F::BYTES;
#pragma unroll
    for (uint32_t k = 0; k < LPT; ++k) {
        uint32_t s = tid + THR * k;
        if (s < nslots) lds[s] = __builtin_nontemporal_load(base16 + (byte0 >> 4) + s);
    }
    uint32_t pf = 0, pf2 = 0;
    const uint32_t ph = (uint32_t)(t % PERIOD);
    if (ph < BURST) {
        const uint64_t off = ((t / PERIOD + K) * PBYTES & ~(uint64_t)127) + ((uint64_t)ph * THR + tid) * 128;
        if (off < (t / PERIOD + K + 1) * PBYTES && off + 4 <= total) pf = *(const uint32_t *)(src + off);
    }
    if constexpr (NEAR > 0) {
        const uint64_t pb = ((t + NEAR) * TBYTES & ~(uint64_t)127) + 128ull * tid;
        if (128u * tid < TBYTES && pb + 4 <= total) pf2 = *(const uint32_t *)(src + pb);
    }
    __syncthreads();
    LdsBytes r{(const uint32_t *)lds};
    const uint32_t off0 = (uint32_t)(byte0 & 15u);
    typename O::T *out = dst + t * TE;
    constexpr uint32_t STEPS = TE / (THR * O::EPL);
#pragma unroll
    for (uint32_t i = 0; i < STEPS; ++i) {
        uint32_t el = (i * THR + tid) * O::EPL;
        uint32_t bt = el / F::N, e = el % F::N;
        if (bt < nblk) {
            uint32_t b = off0 + bt * F::BYTES;
            float y0[4];
            F::dec4(r, b, e, y0);
            O::template store<NT>(out + el, y0);
        }
    }
    asm volatile("" ::"v"(pf), "v"(pf2));
}

// Burst prefetch in BYTE space: the packed input is cut into bursts of CH aligned
// 64 KiB chunks.  When the decode front enters burst g, the first CH workgroups of that
// burst each touch one chunk of burst g + K.
template <class F, class O, bool NT, uint32_t TE, uint32_t THR, uint32_t CH, uint32_t K>
__global__ __launch_bounds__(THR) void deq_burst2(const uint8_t *__restrict__ src,
                                                  typename O::T *__restrict__ dst, uint64_t nblocks) {
    constexpr uint32_t TB = TE / F::N, TBYTES = TB * F::BYTES;
    constexpr uint32_t SLOTS = (TBYTES + 15) / 16 + 1;
    constexpr uint32_t LPT = (SLOTS + THR - 1) / THR;
    constexpr uint64_t CHUNK = 65536, B = (uint64_t)CH * CHUNK;
    static_assert(THR * 128 == CHUNK, "one line per thread");
    __shared__ uint4v lds[SLOTS + 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t head = (uint64_t)src & 15u;
    const uint4v *base16 = (const uint4v *)(src - head);
    const uint64_t t = blockIdx.x;
    const uint64_t byte0 = head + t * TBYTES;
    const uint64_t rem = nblocks - t * TB;
    const uint32_t nblk = rem < TB ? (uint32_t)rem : TB;
    const uint32_t nslots = (uint32_t)(((byte0 & 15u) + (uint64_t)nblk * F::BYTES + 15u) >> 4);
    const uint64_t total = nblocks * F::BYTES;
#pragma unroll
    for (uint32_t k = 0; k < LPT; ++k) {
        uint32_t s = tid + THR * k;
        if (s < nslots) lds[s] = __builtin_nontemporal_load(base16 + (byte0 >> 4) + s);
    }
    uint32_t pf = 0;
    {
        const uint64_t pos = t * TBYTES;                    // packed offset of this tile
        const uint64_t g = pos / B;                          // burst the front is in
        const uint64_t t0 = (g * B + TBYTES - 1) / TBYTES;   // first tile that starts inside burst g
        const uint64_t ph = t - t0;                          // t >= t0 except for the tile straddling in: then ph wraps huge
        if (ph < CH) {
            const uint64_t off = (g + K) * B + ph * CHUNK + (uint64_t)tid * 128;
            if (off + 4 <= total) pf = *(const uint32_t *)(src + off);
        }
    }
    __syncthreads();
    LdsBytes r{(const uint32_t *)lds};
    const uint32_t off0 = (uint32_t)(byte0 & 15u);
    typename O::T *out = dst + t * TE;
    constexpr uint32_t STEPS = TE / (THR * O::EPL);
#pragma unroll
    for (uint32_t i = 0; i < STEPS; ++i) {
        uint32_t el = (i * THR + tid) * O::EPL;
        uint32_t bt = el / F::N, e = el % F::N;
        if (bt < nblk) {
            uint32_t b = off0 + bt * F::BYTES;
            float y0[4];
            F::dec4(r, b, e, y0);
            O::template store<NT>(out + el, y0);
        }
    }
    asm volatile("" ::"v"(pf));
}

// one-shot mix probe: each thread loads one 16-byte value per 8 stored (the
// read:write ratio of a 4-bit format), S stores per thread
template <bool NT, int S>
__global__ __launch_bounds__(256) void k_mix_shot(float4v *dst, const float4v *src, uint64_t n16) {
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * 256 * S;
    float4v v = {1.0f, 2.0f, 3.0f, (float)tid};
    if (tid < 32 * S) v = __builtin_nontemporal_load(src + (base >> 3) + tid);
    v.x += __shfl(v.y, (tid * 7) & 63);
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = base + (uint64_t)k * 256 + tid;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(v, dst + i);
            else dst[i] = v;
        }
    }
}

// one-shot copy probe, S 16-byte loads then S stores per thread
template <bool NTL, bool NTS, int S>
__global__ __launch_bounds__(256) void k_copy_shot(float4v *dst, const float4v *src, uint64_t n16) {
    const uint64_t base = (uint64_t)blockIdx.x * 256 * S + threadIdx.x;
    float4v v[S];
#pragma unroll
    for (int k = 0; k < S; ++k)
        if (base + k * 256 < n16) v[k] = NTL ? __builtin_nontemporal_load(src + base + k * 256) : src[base + k * 256];
#pragma unroll
    for (int k = 0; k < S; ++k)
        if (base + k * 256 < n16) {
            if (NTS) __builtin_nontemporal_store(v[k], dst + base + k * 256);
            else dst[base + k * 256] = v[k];
        }
}

// mix probe with the loads spread over all threads: every thread loads 2 bytes... as one
// dword per 2 threads is awkward, each thread loads one dword (4 B) per 2 stores of 16 B
template <bool NTL, bool NTS, int S>
__global__ __launch_bounds__(256) void k_mix_shot2(float4v *dst, const uint32_t *src, uint64_t n16) {
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * 256 * S;
    uint32_t w[S / 2];
#pragma unroll
    for (int k = 0; k < S / 2; ++k)
        w[k] = NTL ? __builtin_nontemporal_load(src + (base >> 1) + k * 256 + tid) : src[(base >> 1) + k * 256 + tid];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = base + (uint64_t)k * 256 + tid;
        float4v v = {(float)(w[k / 2] & 0xFF), (float)((w[k / 2] >> 8) & 0xFF), (float)((w[k / 2] >> 16) & 0xFF), (float)k};
        if (i < n16) {
            if (NTS) __builtin_nontemporal_store(v, dst + i);
            else dst[i] = v;
        }
    }
}

// write-only probes: one-shot grids, S stores per thread
//   WGI = true : store k of the workgroup covers one contiguous 4 KiB run
//   WGI = false: each wave owns a contiguous run of S KiB
template <bool NT, int S, bool WGI>
__global__ __launch_bounds__(256) void k_fill_shot(float4v *dst, uint64_t n16) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float4v v = {1.0f, 2.0f, 3.0f, (float)tid};
    const uint64_t base = (uint64_t)blockIdx.x * 256 * S;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = WGI ? base + (uint64_t)k * 256 + tid : base + (uint64_t)wv * 64 * S + (uint64_t)k * 64 + lane;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(v, dst + i);
            else dst[i] = v;
        }
    }
}

// generic write-only probe: THR threads, S stores per thread, layout L:
//   0 wg-interleaved, 1 wave-contiguous, 2 thread-contiguous (each lane S*16 consecutive bytes)
template <bool NT, int S, int L, int THR>
__global__ __launch_bounds__(THR) void k_fill_lay(float4v *dst, uint64_t n16) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float4v v = {1.0f, 2.0f, 3.0f, (float)tid};
    const uint64_t base = (uint64_t)blockIdx.x * THR * S;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = L == 0 ? base + (uint64_t)k * THR + tid
                   : L == 1 ? base + (uint64_t)wv * 64 * S + (uint64_t)k * 64 + lane
                            : base + (uint64_t)tid * S + k;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(v, dst + i);
            else dst[i] = v;
        }
    }
}

// store-flavour probe: one-shot fill, 512 threads, S stores per thread, wg-interleaved.
// FL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0, 6 sc0 nt, 7 sc1 nt
template <int FL>
__device__ __forceinline__ void store_flavour(float4v *p, float4v v) {
    if constexpr (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int FL, int S, int THR>
__global__ __launch_bounds__(THR) void k_fill_fl(float4v *dst, uint64_t n16) {
    const uint32_t tid = threadIdx.x;
    float4v v = {1.0f, 2.0f, 3.0f, (float)tid};
    const uint64_t base = (uint64_t)blockIdx.x * THR * S;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = base + (uint64_t)k * THR + tid;
        if (i < n16) store_flavour<FL>(dst + i, v);
    }
}

struct UbVariant {
    std::string name;
    double bytes;
    std::function<void(hipStream_t)> run;
    std::vector<float> ms;
};


// XCD <-> memory locality probe.  Workgroups go to the 8 XCDs round robin (workgroup w -> XCD w % 8).
// Each workgroup of 256 threads writes S chunks of 4 KiB; chunk index = (g * S + j) * 8 + (x + shift) % 8
// with g = w / 8, x = w % 8: XCD x only ever writes chunks whose index is (x + shift) mod 8.  With S = 1 and
// shift = 0 this is the plain one-store-per-thread fill.
template <int S>
__global__ __launch_bounds__(256) void k_fill_numa(float4v *dst, uint64_t n16, uint32_t shift, uint32_t gran16) {
    const uint32_t w = blockIdx.x, g = w >> 3, x = w & 7;
    float4v v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    // gran16: chunk size in 16-byte units (256 = 4 KiB); a workgroup's 256 threads cover 4 KiB per store,
    // so for larger chunks consecutive j stay inside one chunk first
    const uint32_t per_chunk = gran16 / 256;          // stores (of 4 KiB) per chunk
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const uint32_t cj = j / per_chunk, within = j % per_chunk;
        const uint64_t chunk = ((uint64_t)g * (S / per_chunk) + cj) * 8 + ((x + shift) & 7);
        const uint64_t i = chunk * gran16 + within * 256 + threadIdx.x;
        if (i < n16) __builtin_nontemporal_store(v, dst + i);
    }
}

// The same with THR threads: the workgroup's THR/256 quarter-K groups of 256 threads each write their own 4 KiB chunk
// per store; all chunks of a workgroup are 8 chunks (32 KiB) apart.  GR = chunk size in units of 4 KiB is fixed at 1.
template <int S, int THR>
__global__ __launch_bounds__(THR) void k_fill_numa_t(float4v *dst, uint64_t n16) {
    constexpr uint32_t H = THR / 256;                 // 4 KiB chunks per store instruction of the workgroup
    const uint32_t w = blockIdx.x, g = w >> 3, x = w & 7;
    const uint32_t h = threadIdx.x >> 8, t = threadIdx.x & 255;
    float4v v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const uint64_t chunk = (((uint64_t)g * S + j) * H + h) * 8 + x;
        const uint64_t i = chunk * 256 + t;
        if (i < n16) __builtin_nontemporal_store(v, dst + i);
    }
}

// chunk granularity probe: 256 threads, S stores, chunks of GR16 16-byte units (128 = 2 KiB, 256 = 4 KiB), the
// workgroup's 4 KiB store covers 4096 / (16 * GR16) chunks, all of the workgroup's chunks 8 chunks apart
template <int S, int GR16>
__global__ __launch_bounds__(256) void k_fill_numa_g(float4v *dst, uint64_t n16) {
    constexpr uint32_t CPS = 256 / GR16;               // chunks per store
    const uint32_t w = blockIdx.x, g = w >> 3, x = w & 7;
    const uint32_t c = threadIdx.x / GR16, t = threadIdx.x % GR16;
    float4v v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const uint64_t chunk = (((uint64_t)g * S + j) * CPS + c) * 8 + x;
        const uint64_t i = chunk * GR16 + t;
        if (i < n16) __builtin_nontemporal_store(v, dst + i);
    }
}

// ---- chunk-strided decode prototype -------------------------------------------------------------
// The fills above say: a workgroup should write 4 KiB chunks that lie 8 chunks (32 KiB) apart, so
// that the 8 workgroups of a group (one per XCD) together write one contiguous run, chunk by chunk
// in round robin.  Here workgroup w = 8 g + x decodes chunks (g * C + k) * 8 + x, k = 0 .. C-1, of
// the tensor; a chunk is 4 KiB of output = CHB whole blocks.  Every chunk's packed bytes are staged
// into their own LDS region.  C = (THR / 256) * S: each group of 256 threads stores one chunk per step.
template <class F, class O, int THR, int S>
struct XGeom {
    static constexpr uint32_t CH_ELEMS = 4096 / sizeof(typename O::T);
    static constexpr uint32_t CHB = CH_ELEMS / F::N;
    static constexpr uint32_t CHBYTES = CHB * F::BYTES;
    static constexpr uint32_t H = THR / 256;
    static constexpr uint32_t C = H * S;
    static constexpr uint32_t RS = (CHBYTES + 15) / 16 + 2;      // 16-byte slots per region
    static constexpr uint32_t TPC = THR / C;                     // loader threads per chunk
    static constexpr uint32_t LPT = (RS + TPC - 1) / TPC;
    static constexpr uint32_t TB = C * CHB;                      // blocks per workgroup
    static constexpr uint32_t TBYTES = C * CHBYTES;
    static constexpr uint32_t ELEMS = C * CH_ELEMS;
    static constexpr uint32_t LINES = (TBYTES + 127) / 128 + 1;
    static constexpr uint32_t TPW = THR / LINES;
    static_assert(CH_ELEMS / O::EPL == 256, "256 threads store one chunk");
    static_assert(CH_ELEMS % F::N == 0, "a chunk is a whole number of blocks");
};

template <class F, class O, int THR, int S, bool NT>
__global__ __launch_bounds__(THR) void deq_one_x(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                 uint64_t nblocks, Prefetch pf) {
    typedef XGeom<F, O, THR, S> G;
    __shared__ uint4v lds[G::C * G::RS + 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t w = blockIdx.x, g = w >> 3;
    const uint32_t x = (uint32_t)(w & 7);
    {   // stage: TPC threads per chunk
        const uint32_t k = tid / G::TPC, lt = tid % G::TPC;
        const uint64_t ck = (g * G::C + k) * 8 + x;
        const uint64_t first = ck * G::CHB;
        const uint32_t nb = first >= nblocks ? 0u : (uint32_t)(nblocks - first < G::CHB ? nblocks - first : G::CHB);
        const uint8_t *sk = src + ck * G::CHBYTES;
        const uint32_t off0 = (uint32_t)((uintptr_t)sk & 15u);
        const uint4v *base16 = (const uint4v *)(sk - off0);
        const uint32_t nslots = nb ? (off0 + nb * F::BYTES + 15u) >> 4 : 0u;
#pragma unroll
        for (uint32_t q = 0; q < G::LPT; ++q) {
            const uint32_t sl = lt + G::TPC * q;
            if (sl < nslots) lds[k * G::RS + sl] = base16[sl];
        }
    }
    // burst prefetch over the linear packed bytes, unchanged: slice tt = bytes [tt * TBYTES, (tt + 1) * TBYTES)
    uint32_t touch = 0;
    if (pf.period) {
        const uint32_t ph = (uint32_t)(w % pf.period);
        const uint32_t nbw = (pf.period + G::TPW - 1) / G::TPW;
        if (ph < nbw) {
            const uint32_t j = tid / G::LINES, l = tid % G::LINES;
            const uint32_t kk = ph * G::TPW + j;
            if (j < G::TPW && kk < pf.period) {
                const uint64_t tt = w - ph + pf.lead + kk;
                if ((tt + 1) * G::TB <= nblocks) {
                    const uint8_t *base = src + tt * G::TBYTES;
                    const uint8_t *line = (const uint8_t *)((uintptr_t)base & ~(uintptr_t)127) + 128u * l;
                    if (line < base + G::TBYTES) touch = *(const uint32_t *)line;
                }
            }
        }
    }
    __syncthreads();
    LdsBytes r{(const uint32_t *)lds};
    const uint32_t h = tid >> 8, tt = tid & 255u;
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)S; ++i) {
        const uint32_t k = i * G::H + h;
        const uint64_t ck = (g * G::C + k) * 8 + x;
        const uint64_t first = ck * G::CHB;
        const uint32_t nb = first >= nblocks ? 0u : (uint32_t)(nblocks - first < G::CHB ? nblocks - first : G::CHB);
        const uint32_t el = tt * O::EPL, bt = el / F::N, e = el % F::N;
        if (bt < nb) {
            const uint32_t off0 = (uint32_t)((uintptr_t)(src + ck * G::CHBYTES) & 15u);
            const uint32_t b = k * G::RS * 16u + off0 + bt * F::BYTES;
            const Scale sc = F::scale(r, b, e);
            float y[O::EPL];
            F::template vals<O::EPL / 4>(r, b, e, sc, y);
            O::template store<NT>(dst + ck * G::CH_ELEMS + el, y);
        }
    }
    asm volatile("" ::"v"(touch));
}

// ---- scale-table prototype ------------------------------------------------------------------------
// The shipped kernel evaluates F::scale (header f16s, 6-bit scale unpack, one or two multiplies: 20-30
// VALU instructions) in every thread for every 4 (8) elements.  Here the distinct scales of the tile --
// TB * N / SUB of them -- are computed once by the first threads after the staging barrier and kept in LDS;
// the decode reads its pair with one ds_read_b64.
template <class F, class O, bool NT, uint32_t SUB>
__global__ __launch_bounds__(WG_THREADS) void deq_one_tab(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                          uint64_t nblocks, Prefetch pf) {
    typedef TileGeom<F, O> G;
    constexpr uint32_t SPB = F::N / SUB, ENT = G::TB * SPB;
    __shared__ uint4v lds[G::LDS_SLOTS];
    __shared__ float2v tab[ENT];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint8_t *line = prefetch_line<G, false>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    const uint8_t *ts = src + t * G::TBYTES;
    typename O::T *out = dst + t * G::ELEMS;
    const uint32_t tid = threadIdx.x;
    const uint32_t off0 = (uint32_t)((uintptr_t)ts & 15u);
    const uint4v *base16 = (const uint4v *)(ts - off0);
    const uint32_t nslots = (off0 + nblk * F::BYTES + 15u) >> 4;
#pragma unroll
    for (uint32_t k = 0; k < G::LPT; ++k) {
        uint32_t sl = tid + WG_THREADS * k;
        if (sl < nslots) lds[sl] = base16[sl];
    }
    uint32_t touch = 0;
    if (line != nullptr) touch = *(const uint32_t *)line;
    __syncthreads();
    LdsBytes r{(const uint32_t *)lds};
#pragma unroll
    for (uint32_t q0 = 0; q0 < ENT; q0 += WG_THREADS) {
        const uint32_t q = q0 + tid;
        if (q < nblk * SPB) {
            const Scale sc = F::scale(r, off0 + (q / SPB) * F::BYTES, (q % SPB) * SUB);
            tab[q] = float2v{sc.a, sc.c};
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < STORES_PER_THREAD; ++i) {
        uint32_t el = (i * WG_THREADS + tid) * O::EPL;
        uint32_t bt = el / F::N, e = el % F::N;
        if (bt < nblk) {
            uint32_t b = off0 + bt * F::BYTES;
            const float2v tv = tab[bt * SPB + e / SUB];
            const Scale sc{tv.x, tv.y};
            float y[O::EPL];
            F::template vals<O::EPL / 4>(r, b, e, sc, y);
            O::template store<NT>(out + el, y);
        }
    }
    asm volatile("" ::"v"(touch));
}

// ---- chunk-strided probes for the streaming kernels without LDS --------------------------------
// widen f16 -> f32: 256 threads, S steps; step j of workgroup 8g+x converts output chunk (g*S+j)*8+x
// (4 KiB of f32 = 1024 elements, from 2 KiB of f16): 8-byte load, 16-byte store per thread.
template <int S, bool STRIDED, bool NTL>
__global__ __launch_bounds__(256) void k_widen_x(const uint2v *__restrict__ src, float *__restrict__ dst, uint64_t n4) {
    const uint64_t w = blockIdx.x, g = w >> 3;
    const uint32_t x = (uint32_t)(w & 7);
    uint2v v[S];
    uint64_t idx[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const uint64_t chunk = STRIDED ? (g * S + j) * 8 + x : w * S + j;
        idx[j] = chunk * 256 + threadIdx.x;
        if (idx[j] < n4) v[j] = NTL ? __builtin_nontemporal_load(src + idx[j]) : src[idx[j]];
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
        if (idx[j] >= n4) continue;
        float4v y = {h2f(v[j].x & 0xFFFFu), h2f(v[j].x >> 16), h2f(v[j].y & 0xFFFFu), h2f(v[j].y >> 16)};
        __builtin_nontemporal_store(y, (float4v *)(dst + 4 * idx[j]));
    }
}
// copy: 16-byte load and store per thread and step, chunks of 4 KiB
template <int S, bool STRIDED, bool NTL>
__global__ __launch_bounds__(256) void k_copy_x(const float4v *__restrict__ src, float4v *__restrict__ dst, uint64_t n16) {
    const uint64_t w = blockIdx.x, g = w >> 3;
    const uint32_t x = (uint32_t)(w & 7);
    float4v v[S];
    uint64_t idx[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const uint64_t chunk = STRIDED ? (g * S + j) * 8 + x : w * S + j;
        idx[j] = chunk * 256 + threadIdx.x;
        if (idx[j] < n16) v[j] = NTL ? __builtin_nontemporal_load(src + idx[j]) : src[idx[j]];
    }
#pragma unroll
    for (int j = 0; j < S; ++j)
        if (idx[j] < n16) __builtin_nontemporal_store(v[j], dst + idx[j]);
}

// ---- workgroup-shape probes for narrow / quantize / permute (THR threads, S steps per thread) ----
template <class In, class O, int THR, int S>
__global__ __launch_bounds__(THR) void narrow_x(const typename In::T *__restrict__ src, typename O::T *__restrict__ dst,
                                                uint64_t n) {
    const uint64_t base = ((uint64_t)blockIdx.x * THR * S + threadIdx.x) * In::EPL;
    float x[S][In::EPL];
    uint64_t el[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        el[i] = base + (uint64_t)i * THR * In::EPL;
        if (el[i] + In::EPL <= n) In::template load<true>(src + el[i], x[i]);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
        if (el[i] + In::EPL > n) continue;
        if constexpr (In::EPL == 4) {
            uint2v o = {O::pk(x[i][0], x[i][1]), O::pk(x[i][2], x[i][3])};
            __builtin_nontemporal_store(o, (uint2v *)(dst + el[i]));
        } else {
            uint4v o = {O::pk(x[i][0], x[i][1]), O::pk(x[i][2], x[i][3]), O::pk(x[i][4], x[i][5]), O::pk(x[i][6], x[i][7])};
            __builtin_nontemporal_store(o, (uint4v *)(dst + el[i]));
        }
    }
}

template <class Q, class In, int THR, int LPT>
__global__ __launch_bounds__(THR) void quant_x(const typename In::T *__restrict__ src, uint8_t *__restrict__ dst,
                                               uint64_t nblocks) {
    constexpr uint32_t ELEMS = THR * In::EPL * LPT, TB = ELEMS / Q::N, TBYTES = TB * Q::BYTES;
    static_assert(TBYTES % 16 == 0, "tiles leave as whole 16-byte stores");
    __shared__ __attribute__((aligned(16))) uint8_t lds[TBYTES];
    const uint32_t tid = threadIdx.x;
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * TB;
    const uint32_t nblk = rem < TB ? (uint32_t)rem : TB;
    constexpr uint32_t LANES = Q::N / In::EPL;
    const typename In::T *in = src + t * ELEMS;
    float x[LPT][In::EPL];
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)LPT; ++i) {
        uint32_t el = (i * THR + tid) * In::EPL;
        if (el / Q::N < nblk) In::template load<true>(in + el, x[i]);
        else
            for (uint32_t k = 0; k < In::EPL; ++k) x[i][k] = 0.0f;
    }
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)LPT; ++i) {
        uint32_t el = (i * THR + tid) * In::EPL;
        uint32_t bt = el / Q::N;
        Q::template encode<In::EPL>(x[i], tid % LANES, lds + bt * Q::BYTES, bt < nblk);
    }
    __syncthreads();
    const uint32_t nbytes = nblk * Q::BYTES;
    uint8_t *out = dst + t * TBYTES;
    for (uint32_t o = tid * 16; o + 16 <= nbytes; o += THR * 16) {
        const uint4v v = *(const uint4v *)(lds + o);
        __builtin_nontemporal_store(v, (uint4v *)(out + o));
    }
    const uint32_t tail0 = nbytes & ~15u;
    if (tail0 + tid < nbytes) out[tail0 + tid] = lds[tail0 + tid];
}

template <int THR, int S>
__global__ __launch_bounds__(THR) void permute_x(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                 uint64_t row_bytes, uint64_t rows, uint64_t dh) {
    const uint64_t pieces_per_row = row_bytes / 16;
    const uint64_t total = pieces_per_row * rows;
    const uint64_t base = (uint64_t)blockIdx.x * THR * S + threadIdx.x;
    uint4v v[S];
    uint64_t off[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const uint64_t p = base + (uint64_t)k * THR;
        off[k] = ~0ull;
        if (p >= total) continue;
        const uint64_t row = p / pieces_per_row, col = p % pieces_per_row;
        const uint64_t head = row / dh, in_head = row % dh;
        const uint64_t srow = head * dh + (in_head & 1) * (dh / 2) + (in_head >> 1);
        v[k] = __builtin_nontemporal_load((const uint4v *)(src + srow * row_bytes + col * 16));
        off[k] = row * row_bytes + col * 16;
    }
#pragma unroll
    for (int k = 0; k < S; ++k)
        if (off[k] != ~0ull) __builtin_nontemporal_store(v[k], (uint4v *)(dst + off[k]));
}

template <bool BF16SRC, class O, int THR, int S>
__global__ __launch_bounds__(THR) void widen_x(const uint2v *__restrict__ src, typename O::T *__restrict__ dst, uint64_t n4) {
    const uint64_t base = (uint64_t)blockIdx.x * THR * S + threadIdx.x;
    uint2v v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = base + (uint64_t)k * THR;
        if (i < n4) v[k] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        uint64_t i = base + (uint64_t)k * THR;
        if (i >= n4) continue;
        float y[4];
        if (BF16SRC) {
            y[0] = __uint_as_float(v[k].x << 16); y[1] = __uint_as_float(v[k].x & 0xFFFF0000u);
            y[2] = __uint_as_float(v[k].y << 16); y[3] = __uint_as_float(v[k].y & 0xFFFF0000u);
        } else {
            y[0] = h2f(v[k].x & 0xFFFFu); y[1] = h2f(v[k].x >> 16);
            y[2] = h2f(v[k].y & 0xFFFFu); y[3] = h2f(v[k].y >> 16);
        }
        if constexpr (O::EPL == 4) {
            O::template store<true>(dst + 4 * i, y);
        } else {
            uint2v o = {O::pk(y[0], y[1]), O::pk(y[2], y[3])};
            __builtin_nontemporal_store(o, (uint2v *)(dst + 4 * i));
        }
    }
}

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
    const uint8_t *ts = src + t * G::TBYTES;
    // group wait (pf.wait), holding back the stores only
    const uint32_t z = pf.wait ? group_wait(dst + ((t >> pf.wait) << pf.wait) * G::ELEMS) : 0u;
    typename O::T *out = dst + t * G::ELEMS + z;
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

// ---- Q4_K without LDS ------------------------------------------------------------------------------
// Every lane fetches what it needs straight from global memory (the lines come from L1 / L2 / the
// memory-side cache after the burst prefetch): the block's 16-byte header with one dwordx4 load (all
// lanes of a block read the same 16 bytes) and its quants with one dword (f32 out: 4 elements) or
// dwordx2 (16-bit out: 8 elements) load.  No staging, no barrier; waves are independent.
// One-shot grid, THR threads, S stores per thread; thread -> elements exactly as in the tile kernel.
// Requires the source 16-byte aligned (Q4_K blocks are 144 = 9 x 16 bytes).
template <class O, int THR, int S>
__global__ __launch_bounds__(THR) void deq_q4k_direct(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                      uint64_t nblocks, Prefetch pf) {
    constexpr uint32_t ELEMS = THR * O::EPL * S, TB = ELEMS / 256, TBYTES = TB * 144;
    constexpr uint32_t LINES = (TBYTES + 127) / 128 + 1, TPW = THR / LINES;
    const uint64_t t = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint64_t rem = nblocks - t * TB;
    const uint32_t nblk = rem < TB ? (uint32_t)rem : TB;
    uint32_t touch = 0;
    if (pf.period) {
        const uint32_t ph = (uint32_t)(t % pf.period);
        const uint32_t nb = (pf.period + TPW - 1) / TPW;
        if (ph < nb) {
            const uint32_t j = tid / LINES, l = tid % LINES;
            const uint32_t k = ph * TPW + j;
            if (j < TPW && k < pf.period) {
                const uint64_t tt = t - ph + pf.lead + k;
                if ((tt + 1) * TB <= nblocks) {
                    const uint8_t *base = src + tt * TBYTES;
                    const uint8_t *ln = (const uint8_t *)((uintptr_t)base & ~(uintptr_t)127) + 128u * l;
                    if (ln < base + TBYTES) touch = *(const uint32_t *)ln;
                }
            }
        }
    }
    const uint8_t *ts = src + t * TBYTES;
    typename O::T *out = dst + t * ELEMS;
    uint4v hdr[S];
    uint32_t q[S][O::EPL / 4];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const uint32_t el = (i * THR + tid) * O::EPL, bt = el / 256, e = el % 256;
        if (bt < nblk) {
            const uint8_t *b = ts + bt * 144;
            hdr[i] = *(const uint4v *)b;
            const uint8_t *qp = b + 16 + 32 * (e >> 6) + (e & 31);
            if constexpr (O::EPL == 4) q[i][0] = *(const uint32_t *)qp;
            else { const uint2v v = *(const uint2v *)qp; q[i][0] = v.x; q[i][1] = v.y; }
        }
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const uint32_t el = (i * THR + tid) * O::EPL, bt = el / 256, e = el % 256;
        if (bt >= nblk) continue;
        const float d = h2f(hdr[i].x & 0xFFFFu), dmin = h2f(hdr[i].x >> 16);
        const uint32_t is = e >> 5, j8 = (is & 3u) * 8u;
        const uint32_t A = (hdr[i].y >> j8) & 0xFFu, B = (hdr[i].z >> j8) & 0xFFu, C = (hdr[i].w >> j8) & 0xFFu;
        const bool low = is < 4;
        const uint32_t sc = low ? (A & 63u) : ((C & 0xFu) | ((A >> 6) << 4));
        const uint32_t mn = low ? (B & 63u) : ((C >> 4) | ((B >> 6) << 4));
        const float a = __fmul_rn(d, (float)sc), c = __fmul_rn(dmin, (float)mn);
        float y[O::EPL];
#pragma unroll
        for (uint32_t g = 0; g < O::EPL / 4; ++g) {
            uint32_t w = (q[i][g] >> (4 * (is & 1))) & 0x0F0F0F0Fu;
            asm("" : "+v"(w));
            y[4 * g + 0] = __fsub_rn(__fmul_rn(a, (float)(w & 0xFFu)), c);
            y[4 * g + 1] = __fsub_rn(__fmul_rn(a, (float)((w >> 8) & 0xFFu)), c);
            y[4 * g + 2] = __fsub_rn(__fmul_rn(a, (float)((w >> 16) & 0xFFu)), c);
            y[4 * g + 3] = __fsub_rn(__fmul_rn(a, (float)(w >> 24)), c);
        }
        O::template store<true>(out + el, y);
    }
    asm volatile("" ::"v"(touch));
}

// ---- batch kernel with an XCD-aware record layout -------------------------------------------------------
// Workgroups go to the 8 XCDs in round robin, so the records of tiles t, t+1, ..., t+7 are each read by a
// different L2 and every workgroup misses on its record's line.  Here the record of tile t sits at slot
// (t & ~63) | ((t & 7) << 3) | ((t >> 3) & 7): the 8 records one XCD reads out of every 64 tiles share a
// 128-byte line.  PERM = false is the shipped layout (control).
__device__ __forceinline__ uint64_t xcd_slot(uint64_t t) { return (t & ~63ull) | ((t & 7ull) << 3) | ((t >> 3) & 7ull); }

template <class F, class O, bool PERM, int AHEAD = 0>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_px(const TileRef *__restrict__ table, uint64_t ntiles, Prefetch pf) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const TileRef me = table[PERM ? xcd_slot(t) : t];
    // AHEAD > 0: the first of the 8 workgroups that share a record line also touches the line that this XCD's
    // workgroups will read AHEAD runs of 64 tiles later
    uint32_t touch2 = 0;
    if (AHEAD > 0 && ((t >> 3) & 7) == 0 && threadIdx.x == 0 && t + 64ull * AHEAD < ntiles)
        touch2 = *(const uint32_t *)&table[xcd_slot(t + 64ull * AHEAD)];
    const uint8_t *line = prefetch_line<G, false>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        const uint64_t far = table[PERM ? xcd_slot(tt) : tt].src_nblk;
        return (far >> 48) == G::TB ? (const uint8_t *)(far & 0xFFFFFFFFFFFFull) : nullptr;
    });
    deq_tile<F, O, true>(lds, (const uint8_t *)(me.src_nblk & 0xFFFFFFFFFFFFull), (uint32_t)(me.src_nblk >> 48),
                         (typename O::T *)me.dst, line);
    asm volatile("" ::"v"(touch2));
}

// ---- single-tensor kernel behind a shared wait ------------------------------------------------------------
// G consecutive workgroups first load one dword of the same line (the first line of the first of their G tiles)
// and wait for it before they issue their own tile loads: the line arrives at all of them together, so their reads
// reach the memory bunched.  (The batch kernel does this by accident: the 8 records of a line in tile order.)
template <class F, class O, uint32_t G>
__global__ __launch_bounds__(WG_THREADS) void deq_one_gate(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                           uint64_t nblocks, Prefetch pf) {
    typedef TileGeom<F, O> TG;
    __shared__ uint4v lds[TG::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * TG::TB;
    const uint32_t nblk = rem < TG::TB ? (uint32_t)rem : TG::TB;
    const uint8_t *gate = (const uint8_t *)((uintptr_t)(src + (t / G * G) * TG::TBYTES) & ~(uintptr_t)3);
    uint32_t gv = *(const uint32_t *)gate, zero;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(gv));
    const uint32_t z = __builtin_amdgcn_readfirstlane(zero);       // 0, but only known once the gate line is here
    const uint8_t *line = prefetch_line<TG, false>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * TG::TB <= nblocks ? src + tt * TG::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + t * TG::TBYTES + z, nblk, dst + t * TG::ELEMS, line);
}

// the same shape as a plain copy: what the gather pattern itself costs
__global__ __launch_bounds__(STREAM_THREADS) void copy16_flat(uint4v *__restrict__ dst, const uint4v *__restrict__ src, uint64_t total) {
    const uint64_t p = (uint64_t)blockIdx.x * STREAM_THREADS + threadIdx.x;
    if (p >= total) return;
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + p), dst + p);
}

// ---- decode kernel whose prefetch phase is computed without a division ---------------------------------------
template <class G, class SrcOf>
__device__ __forceinline__ const uint8_t *prefetch_line_fd(uint32_t t, Prefetch pf, FastDiv32 per, SrcOf tile_src) {
    if (pf.period == 0) return nullptr;
    const uint32_t ph = t - per.div(t) * pf.period;
    const uint32_t nb = (pf.period + G::TPW - 1) / G::TPW;
    if (ph >= nb) return nullptr;
    const uint32_t j = threadIdx.x / G::LINES, l = threadIdx.x % G::LINES;
    const uint32_t k = ph * G::TPW + j;
    if (j >= G::TPW || k >= pf.period) return nullptr;
    const uint8_t *base = tile_src((uint64_t)t - ph + pf.lead + k);
    if (base == nullptr) return nullptr;
    const uint8_t *line = (const uint8_t *)((uintptr_t)base & ~(uintptr_t)127) + 128u * l;
    return line < base + G::TBYTES ? line : nullptr;
}

template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_one_fd(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                         uint64_t nblocks, Prefetch pf, FastDiv32 per) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;                 // addresses as in the shipped kernel; only the phase changes
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint8_t *line = prefetch_line_fd<G>((uint32_t)t, pf, per, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + t * G::TBYTES, nblk, dst + t * G::ELEMS, line);
}

// ---- single-tensor kernel that first waits for a cold record, as the batch kernel does ---------------------------
// MODE 0: a 16-byte record per tile in tile order (what deq_batch reads), value ignored.
// MODE 1: one dword of the destination line where this group of 8 tiles' output starts (cold, about to be overwritten).
template <class F, class O, int MODE>
__global__ __launch_bounds__(WG_THREADS) void deq_one_rec(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                          uint64_t nblocks, Prefetch pf, const TileRef *__restrict__ table) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    uint32_t probe;
    if (MODE == 0) probe = (uint32_t)table[t].dst;
    else probe = *(const uint32_t *)(dst + (t / 8 * 8) * G::ELEMS);
    uint32_t zero;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(probe));
    const uint32_t z = __builtin_amdgcn_readfirstlane(zero);
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + t * G::TBYTES + z, nblk, dst + t * G::ELEMS, line);
}

// ---- single-tensor kernel that first waits for one dword of its group's (cold) destination -------------------------
// `gshift`: log2 of the group size in tiles; the probed line is where the group's output starts.
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_one_cdw(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                          uint64_t nblocks, Prefetch pf, uint32_t gshift) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint32_t probe = *(const uint32_t *)(dst + ((t >> gshift) << gshift) * G::ELEMS);
    uint32_t zero;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(probe));
    const uint32_t z = __builtin_amdgcn_readfirstlane(zero);
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + t * G::TBYTES + z, nblk, dst + t * G::ELEMS, line);
}

// ---- group wait that holds back the STORES only (tile loads go out at once) -----------------------------------------
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_one_late(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                           uint64_t nblocks, Prefetch pf) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint32_t z = group_wait(dst + ((t >> pf.wait) << pf.wait) * G::ELEMS);
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + t * G::TBYTES, nblk, dst + t * G::ELEMS + z, line);
}

template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_late(const TileRef *__restrict__ table, uint64_t ntiles, Prefetch pf, uint32_t tile_out_bytes) {
    typedef TileGeom<F, O> G;
    constexpr bool PX = RecordOrder<O>::PER_XCD;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const TileRef me = table[record_slot<PX>(t)];
    // bits 48..55 of dst: tiles since the tensor's first tile, clamped to 255 (the host writes it)
    const uint64_t dstp = me.dst & 0xFFFFFFFFFFFFull;
    const uint32_t since = (uint32_t)(me.dst >> 48) & 0xFFu;
    const uint32_t back = min((uint32_t)(t - ((t >> pf.wait) << pf.wait)), since);
    const uint32_t z = group_wait((const void *)(dstp - (uint64_t)back * tile_out_bytes));
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        const uint64_t far = table[record_slot<PX>(tt)].src_nblk;
        return (far >> 48) == G::TB ? (const uint8_t *)(far & 0xFFFFFFFFFFFFull) : nullptr;
    });
    deq_tile<F, O, true>(lds, (const uint8_t *)(me.src_nblk & 0xFFFFFFFFFFFFull), (uint32_t)(me.src_nblk >> 48),
                         (typename O::T *)(dstp + z), line);
}
// the same with the tile loads held back as well (the wait before the loads, dst from the own record)
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_early(const TileRef *__restrict__ table, uint64_t ntiles, Prefetch pf, uint32_t tile_out_bytes) {
    typedef TileGeom<F, O> G;
    constexpr bool PX = RecordOrder<O>::PER_XCD;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const TileRef me = table[record_slot<PX>(t)];
    const uint64_t dstp = me.dst & 0xFFFFFFFFFFFFull;
    const uint32_t since = (uint32_t)(me.dst >> 48) & 0xFFu;
    const uint32_t back = min((uint32_t)(t - ((t >> pf.wait) << pf.wait)), since);
    const uint32_t z = group_wait((const void *)(dstp - (uint64_t)back * tile_out_bytes));
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        const uint64_t far = table[record_slot<PX>(tt)].src_nblk;
        return (far >> 48) == G::TB ? (const uint8_t *)(far & 0xFFFFFFFFFFFFull) : nullptr;
    });
    deq_tile<F, O, true>(lds, (const uint8_t *)(me.src_nblk & 0xFFFFFFFFFFFFull) + z, (uint32_t)(me.src_nblk >> 48),
                         (typename O::T *)dstp, line);
}

// ---- batch kernel, stores held back by a probe whose address comes from a side array (one entry per group) -----------
// The probe starts together with the record load, so the workgroup does not live longer than today.
template <class F, class O, bool EARLY>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_side(const TileRef *__restrict__ table, uint64_t ntiles, Prefetch pf,
                                                             const uint64_t *__restrict__ group_dst) {
    typedef TileGeom<F, O> G;
    constexpr bool PX = RecordOrder<O>::PER_XCD;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint64_t gd = group_dst[t >> pf.wait];
    const TileRef me = table[record_slot<PX>(t)];
    const uint32_t z = group_wait((const void *)gd);
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        const uint64_t far = table[record_slot<PX>(tt)].src_nblk;
        return (far >> 48) == G::TB ? (const uint8_t *)(far & 0xFFFFFFFFFFFFull) : nullptr;
    });
    deq_tile<F, O, true>(lds, (const uint8_t *)(me.src_nblk & 0xFFFFFFFFFFFFull) + (EARLY ? z : 0u), (uint32_t)(me.src_nblk >> 48),
                         (typename O::T *)(me.dst + (EARLY ? 0u : z)), line);
}

// ---- what breaks the group wait in the batch kernel?  The single-tensor kernel with a record load added. -----------
// MODE 0: the record is loaded and its value folded into the source address (the tile loads wait for it, as in the
// batch kernel).  MODE 1: the record is loaded and only kept alive (nothing waits for it until the end).
template <class F, class O, int MODE>
__global__ __launch_bounds__(WG_THREADS) void deq_one_wait_rec(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                               uint64_t nblocks, Prefetch pf, const TileRef *__restrict__ table) {
    typedef TileGeom<F, O> G;
    constexpr bool PX = RecordOrder<O>::PER_XCD;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint32_t rec = (uint32_t)table[record_slot<PX>(t)].dst;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    uint32_t zp = 0;
    if (MODE == 2) {       // the probe's own address waits for the record (a scalar load that hits L2 or the scalar cache)
        uint32_t zero;
        asm volatile("s_and_b32 %0, 0, %1" : "=s"(zero) : "s"(rec));
        zp = zero;
    }
    const uint32_t z = pf.wait ? group_wait(dst + ((t >> pf.wait) << pf.wait) * G::ELEMS + zp) : 0u;
    uint32_t zr = 0;
    if (MODE == 0) {
        uint32_t zero;
        asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(rec));
        zr = __builtin_amdgcn_readfirstlane(zero);
    }
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + t * G::TBYTES + z + zr, nblk, dst + t * G::ELEMS, line);
    if (MODE == 1) asm volatile("" ::"s"(rec));
}

// ---- batch kernel without a tile table: the tile's tensor is found by a scalar binary search over the descriptors ----
// The descriptor array is a few KB and stays in the scalar cache; no per-tile record is read from memory.
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_bs(const TensorDesc *__restrict__ desc, uint32_t ntensors,
                                                           uint64_t ntiles, Prefetch pf) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    uint32_t lo = 0, hi = ntensors - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (desc[mid].first_tile <= t) lo = mid;
        else hi = mid - 1;
    }
    const TensorDesc d = desc[lo];
    const uint64_t k = t - d.first_tile;
    const uint64_t left = d.blocks - k * G::TB;
    const uint32_t nblk = left < G::TB ? (uint32_t)left : G::TB;
    const uint8_t *src = (const uint8_t *)d.src + k * G::TBYTES;
    typename O::T *dst = (typename O::T *)d.dst + k * G::ELEMS;
    uint32_t z = 0;
    if (pf.wait) {
        const uint64_t g = (t >> pf.wait) << pf.wait;            // the group's first tile, or this tensor's first
        const uint64_t kg = g > d.first_tile ? g - d.first_tile : 0;
        z = group_wait((const typename O::T *)d.dst + kg * G::ELEMS);
    }
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        // the far tile is rarely more than a few tensors on: walk from this one
        uint32_t dd = lo;
        while (dd + 1 < ntensors && desc[dd + 1].first_tile <= tt) ++dd;
        const uint64_t kk = tt - desc[dd].first_tile;
        return (kk + 1) * G::TB <= desc[dd].blocks ? (const uint8_t *)desc[dd].src + kk * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + z, nblk, dst, line);
}

// ---- batch kernel with one record per 64 tiles ("run"): tensors start on run boundaries -------------------------------
// The table is 64 times smaller (it stays in L2), a record line serves 512 consecutive workgroups, and the group
// start needs no second lookup: a group of up to 64 tiles lies inside one run.
struct RunRef {
    uint64_t src;      // first block of the run
    uint64_t dst;      // first output element of the run
    uint64_t blocks;   // blocks from the run's start to the end of its tensor (0: padding)
    uint64_t pad;
};
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_run(const RunRef *__restrict__ runs, uint64_t nruns, Prefetch pf) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const RunRef r = runs[t >> 6];
    const uint32_t k = (uint32_t)t & 63u;
    if ((uint64_t)k * G::TB >= r.blocks) return;                  // padding behind a tensor's last tile
    const uint64_t left = r.blocks - (uint64_t)k * G::TB;
    const uint32_t nblk = left < G::TB ? (uint32_t)left : G::TB;
    uint32_t z = 0;
    if (pf.wait) z = group_wait((const typename O::T *)r.dst + (uint64_t)((k >> pf.wait) << pf.wait) * G::ELEMS);
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if ((tt >> 6) >= nruns) return nullptr;
        const RunRef f = runs[tt >> 6];
        const uint64_t kk = tt & 63u;
        return (kk + 1) * G::TB <= f.blocks ? (const uint8_t *)f.src + kk * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, (const uint8_t *)r.src + (uint64_t)k * G::TBYTES + z, nblk,
                         (typename O::T *)r.dst + (uint64_t)k * G::ELEMS, line);
}

// ---- burst prefetch (HBM -> memory-side cache, far ahead) plus a near touch (memory-side cache -> this XCD's L2) --------
// Every workgroup also touches the lines of the tile `near` tiles ahead (a multiple of 8: the same XCD in round-robin
// dispatch), so that the workgroup which decodes it finds its input in L2 instead of the memory-side cache.
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_one_near(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                           uint64_t nblocks, Prefetch pf, uint32_t near) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    uint32_t touch2 = 0;
    if (near && threadIdx.x < G::LINES && (t + near + 1) * G::TB <= nblocks) {
        const uint8_t *base = src + (t + near) * G::TBYTES;
        const uint8_t *ln = (const uint8_t *)((uintptr_t)base & ~(uintptr_t)127) + 128u * threadIdx.x;
        if (ln < base + G::TBYTES) touch2 = *(const uint32_t *)ln;
    }
    deq_tile<F, O, true>(lds, src + t * G::TBYTES, nblk, dst + t * G::ELEMS, line);
    asm volatile("" ::"v"(touch2));
}

// ---- record and probe through the VECTOR memory path instead of the scalar cache ---------------------------------------
// The compiler turns loads from wave-uniform addresses into scalar loads; the scalar cache has a narrow miss path that
// two or three CUs share.  A zero the compiler cannot see through makes the address look divergent.
__device__ __forceinline__ uint32_t opaque_lane_zero() {
    uint32_t z = 0;
    asm volatile("" : "+v"(z));
    return z;
}
__device__ __forceinline__ uint32_t group_wait_v(const void *group_dst) {
    const uint32_t seen = *(const uint32_t *)((const uint8_t *)group_dst + opaque_lane_zero());
    uint32_t zero;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(seen));
    return __builtin_amdgcn_readfirstlane(zero);
}
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    // the builtin returns int: without the casts the low half would be sign-extended over the high half
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// MODE bit 0: record by vector load; bit 1: probe by vector load (if pf.wait)
template <class F, class O, int MODE>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_v(const TileRef *__restrict__ table, uint64_t ntiles, Prefetch pf) {
    typedef TileGeom<F, O> G;
    constexpr bool PX = RecordOrder<O>::PER_XCD;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    TileRef me;
    if (MODE & 1) {
        const TileRef *p = table + record_slot<PX>(t) + opaque_lane_zero();
        const uint4v raw = *(const uint4v *)p;
        me.src_nblk = rfl64(((uint64_t)raw.y << 32) | raw.x);
        me.dst = rfl64(((uint64_t)raw.w << 32) | raw.z);
    } else {
        me = table[record_slot<PX>(t)];
    }
    uint32_t z = 0;
    if (pf.wait) {
        const uint64_t g = (t >> pf.wait) << pf.wait;
        uint64_t gd;
        if (MODE & 1) {
            const TileRef *p = table + record_slot<PX>(g) + opaque_lane_zero();
            const uint2v raw = *(const uint2v *)&p->dst;
            gd = rfl64(((uint64_t)raw.y << 32) | raw.x);
        } else {
            gd = table[record_slot<PX>(g)].dst;
        }
        z = (MODE & 2) ? group_wait_v((const void *)gd) : group_wait((const void *)gd);
    }
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        const uint64_t far = table[record_slot<PX>(tt)].src_nblk;
        return (far >> 48) == G::TB ? (const uint8_t *)(far & 0xFFFFFFFFFFFFull) : nullptr;
    });
    deq_tile<F, O, true>(lds, (const uint8_t *)(me.src_nblk & 0xFFFFFFFFFFFFull) + z, (uint32_t)(me.src_nblk >> 48),
                         (typename O::T *)me.dst, line);
}

// the single-tensor kernel with the probe through the vector path
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_one_vprobe(const uint8_t *__restrict__ src, typename O::T *__restrict__ dst,
                                                             uint64_t nblocks, Prefetch pf) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint32_t z = pf.wait ? group_wait_v(dst + ((t >> pf.wait) << pf.wait) * G::ELEMS) : 0u;
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + t * G::TBYTES + z, nblk, dst + t * G::ELEMS, line);
}

// ---- group start on a plain copy (what the LDS-free streaming kernels would gain) --------------------------------------
// Workgroup b copies pieces [b * 256 * S, (b + 1) * 256 * S); the group's probe is the first dword of the group's output.
template <int S>
__global__ __launch_bounds__(STREAM_THREADS) void copy16_group(uint4v *__restrict__ dst, const uint4v *__restrict__ src,
                                                               uint64_t total, uint32_t wait) {
    const uint64_t b = blockIdx.x;
    uint32_t z = 0;
    if (wait) z = group_wait(dst + ((b >> wait) << wait) * (STREAM_THREADS * S));
    const uint4v *s2 = src + z;
    uint4v v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const uint64_t p = (b * S + k) * STREAM_THREADS + threadIdx.x;
        if (p < total) v[k] = __builtin_nontemporal_load(s2 + p);
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const uint64_t p = (b * S + k) * STREAM_THREADS + threadIdx.x;
        if (p < total) __builtin_nontemporal_store(v[k], dst + p);
    }
}

// ---- batch kernel whose probe address needs no record: outputs lie in one arena, in tile order -------------------------
// probe = arena + group's first tile * tile_out_bytes + skew, clamped into the arena.  skew models the error that
// short last tiles of earlier tensors would cause in a real plan.
template <class F, class O>
__global__ __launch_bounds__(WG_THREADS) void deq_batch_arena(const TileRef *__restrict__ table, uint64_t ntiles, Prefetch pf,
                                                              const uint8_t *__restrict__ arena, uint64_t arena_bytes, uint64_t skew) {
    typedef TileGeom<F, O> G;
    constexpr bool PX = RecordOrder<O>::PER_XCD;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    uint32_t z = 0;
    if (pf.wait) {
        uint64_t off = ((t >> pf.wait) << pf.wait) * (uint64_t)(G::ELEMS * sizeof(typename O::T)) + skew;
        if (off + 128 > arena_bytes) off = arena_bytes - 128;
        z = group_wait(arena + off);
    }
    const TileRef me = table[record_slot<PX>(t)];
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        const uint64_t far = table[record_slot<PX>(tt)].src_nblk;
        return (far >> 48) == G::TB ? (const uint8_t *)(far & 0xFFFFFFFFFFFFull) : nullptr;
    });
    deq_tile<F, O, true>(lds, (const uint8_t *)(me.src_nblk & 0xFFFFFFFFFFFFull) + z, (uint32_t)(me.src_nblk >> 48),
                         (typename O::T *)me.dst, line);
}

// ---- several tensors per launch, their descriptors in the kernel arguments (no table in memory) -----------------------
// A launch covers up to K tensors; the workgroup finds its tensor by comparing its index with the K first-tile
// numbers (all in the kernel-argument segment, which every workgroup reads and the scalar cache keeps), then works
// exactly like the single-tensor kernel, group start included.
template <int K>
struct DescPack {
    uint32_t first[K];      // first tile of tensor j in this launch, ascending; unused entries 0xFFFFFFFF
    uint64_t src[K], dst[K], blocks[K];
};
template <class F, class O, int K>
__global__ __launch_bounds__(WG_THREADS) void deq_multi(const DescPack<K> pack, Prefetch pf) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint32_t t = blockIdx.x;
    uint32_t i = 0;
#pragma unroll
    for (int j = 1; j < K; ++j) i += t >= pack.first[j] ? 1u : 0u;
    const uint64_t k = t - pack.first[i];
    const uint64_t nblocks = pack.blocks[i];
    const uint8_t *src = (const uint8_t *)pack.src[i];
    typename O::T *dst = (typename O::T *)pack.dst[i];
    const uint64_t rem = nblocks - k * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint32_t z = pf.wait ? group_wait(dst + ((k >> pf.wait) << pf.wait) * G::ELEMS) : 0u;
    // the burst stays inside the tensor (a tensor's last period goes without, as in the single-tensor kernel)
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(k, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;
    });
    deq_tile<F, O, true>(lds, src + k * G::TBYTES + z, nblk, dst + k * G::ELEMS, line);
}

}  // namespace ggq
