// ubench.hip -- in-process A/B of memory-stream kernels on one MI355X.
//
// Measures, with HIP events on one stream and interleaved rounds:
//   fill     : 16-byte stores only (the write ceiling the decode kernels chase)
//   copy     : 16-byte loads + stores
//   deq:<fmt>: the shipped decode kernel for a block format, f32 / f16 out
// Buffers are sized well past the 256 MiB Infinity Cache.  Prints one line per
// variant: median and best time, algorithmic GB/s.
//
//   ./ubench [elements_in_millions] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"
#include "kernels_v2.h"
#include "quantize.h"
#include "schedule.h"
#include "rearrange.h"

using namespace ggq;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } \
    } while (0)

__global__ __launch_bounds__(256) void k_init_f32(float *dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint64_t z = i * 0x9E3779B97F4A7C15ull;
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        dst[i] = (float)(int32_t)(uint32_t)z * (1.0f / 2147483648.0f);
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_fill(float4v *dst, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    float4v v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (; i < n16; i += stride) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(float4v *dst, const float4v *src, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (; i < n16; i += stride) {
        float4v v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

// pure read: sums 16-byte pieces; the store never happens (the sum of finite test data is not NaN-patterned)
template <bool NT, int S, int WG>
__global__ __launch_bounds__(WG) void k_read_shot(float4v *sink, const float4v *src, uint64_t n16) {
    const uint64_t base = (uint64_t)blockIdx.x * (WG * S) + threadIdx.x;
    float4v v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const uint64_t i = base + (uint64_t)k * WG;
        v[k] = i < n16 ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : float4v{0, 0, 0, 0};
    }
    float acc = 0;
#pragma unroll
    for (int k = 0; k < S; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    if (__float_as_uint(acc) == 0x7FC12345u) sink[base] = v[0];
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read_loop(float4v *sink, const float4v *src, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    float acc = 0;
    for (; i < n16; i += stride) {
        float4v v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (__float_as_uint(acc) == 0x7FC12345u) sink[threadIdx.x] = float4v{acc, 0, 0, 0};
}

// each workgroup owns a contiguous span (instead of a grid-stride interleave)
template <bool NT>
__global__ __launch_bounds__(256) void k_fill_span(float4v *dst, uint64_t n16) {
    const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x;
    uint64_t i = (uint64_t)blockIdx.x * per + threadIdx.x;
    const uint64_t end = std::min<uint64_t>(n16, (uint64_t)(blockIdx.x + 1) * per);
    float4v v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (; i < end; i += 256) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

enum { Q4K_ID = 12, Q6K_ID = 14, Q2K_ID = 10, Q3K_ID = 11, Q5K_ID = 13, Q8K_ID = 15, IQ4XS_ID = 23, IQ4NL_ID = 20,
       Q4_0_ID = 2, Q4_1_ID = 3, Q5_0_ID = 6, Q5_1_ID = 7, Q8_0_ID = 8, Q8_1_ID = 9 };
typedef UbVariant Variant;   // name, algorithmic bytes per launch, launcher, samples

static const char *g_sweep_tag = "sw";   // labels start with the mode name so the name filter keeps them
// period x lead sweep of the shipped kernel for one format / output type
template <class F, class O>
void add_sweep(std::vector<UbVariant> &vs, const char *name, const uint8_t *d_in, void *d_out, uint64_t elems) {
    typedef TileGeom<F, O> G;
    const uint64_t blocks = elems / F::N;
    const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);
    const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    static const uint32_t periods[] = {0, 256, 384, 512, 640, 768, 1024, 1280, 1536};
    static const uint32_t leads[] = {512, 768, 1024, 1280, 1536, 2048, 3072};
    for (uint32_t P : periods)
        for (uint32_t L : leads) {
            if (P == 0 && L != leads[0]) continue;
            if (P && L < P) continue;
            char label[96];
            std::snprintf(label, sizeof label, "%s %s P%u L%u", g_sweep_tag, name, P, P ? L : 0);
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, d_in,
                                                 (typename O::T *)d_out, blocks, Prefetch{P, L}); }, {}});
        }
}

// the same sweep through the batch kernel (per-tile table), one contiguous tensor
template <class F, class O>
void add_sweep_batch(std::vector<UbVariant> &vs, const char *name, const uint8_t *d_in, void *d_out, uint64_t elems) {
    typedef TileGeom<F, O> G;
    const uint64_t blocks = elems / F::N;
    const uint64_t ntiles = (blocks + G::TB - 1) / G::TB;
    const uint64_t slots = (ntiles + 63) / 64 * 64;
    std::vector<TileRef> tab(slots);
    for (uint64_t t = 0; t < ntiles; ++t) {
        uint64_t nblk = std::min<uint64_t>(G::TB, blocks - t * G::TB);
        tab[record_slot<RecordOrder<O>::PER_XCD>(t)] = {((uint64_t)(uintptr_t)d_in + t * G::TBYTES) | (nblk << 48),
                                                        (uint64_t)(uintptr_t)d_out + t * G::ELEMS * sizeof(typename O::T)};
    }
    TileRef *d_tab;
    CK(hipMalloc(&d_tab, slots * sizeof(TileRef)));
    CK(hipMemcpy(d_tab, tab.data(), slots * sizeof(TileRef), hipMemcpyHostToDevice));
    const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    static const uint32_t periods[] = {0, 1, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 3072, 4096};
    static const uint32_t leads[] = {512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 6144};
    for (uint32_t P : periods)
        for (uint32_t L : leads) {
            if (P == 0 && L != leads[0]) continue;
            if (P > 1 && L < P) continue;
            char label[96];
            std::snprintf(label, sizeof label, "%s %s P%u L%u", g_sweep_tag[2] ? g_sweep_tag : "swb", name, P, P ? L : 0);
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch<F, O, true>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s,
                                                 d_tab, (uint64_t)0, ntiles, Prefetch{P, L}); }, {}});
        }
}

// batch kernel with the records laid out per XCD (deq_batch_px) against the shipped layout, at the shipped schedule
template <class F, class O>
void add_batch_px(std::vector<UbVariant> &vs, const char *name, const uint8_t *d_in, void *d_out, uint64_t elems, Prefetch pf, Prefetch ps) {
    typedef TileGeom<F, O> G;
    const uint64_t blocks = elems / F::N;
    const uint64_t ntiles = (blocks + G::TB - 1) / G::TB;
    const uint64_t slots = (ntiles + 63) / 64 * 64;
    std::vector<TileRef> tab(slots), perm(slots);
    for (uint64_t t = 0; t < ntiles; ++t) {
        uint64_t nblk = std::min<uint64_t>(G::TB, blocks - t * G::TB);
        tab[t] = {((uint64_t)(uintptr_t)d_in + t * G::TBYTES) | (nblk << 48),
                  (uint64_t)(uintptr_t)d_out + t * G::ELEMS * sizeof(typename O::T)};
        perm[(t & ~63ull) | ((t & 7ull) << 3) | ((t >> 3) & 7ull)] = tab[t];
    }
    TileRef *d_tab, *d_perm;
    CK(hipMalloc(&d_tab, slots * sizeof(TileRef)));
    CK(hipMalloc(&d_perm, slots * sizeof(TileRef)));
    CK(hipMemcpy(d_tab, tab.data(), slots * sizeof(TileRef), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_perm, perm.data(), slots * sizeof(TileRef), hipMemcpyHostToDevice));
    const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    char label[96];
    std::snprintf(label, sizeof label, "bpx %s records in tile order", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_batch_px<F, O, false>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s, d_tab, ntiles, pf); }, {}});
    std::snprintf(label, sizeof label, "bpx %s records per XCD", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_batch_px<F, O, true>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s, d_perm, ntiles, pf); }, {}});
    std::snprintf(label, sizeof label, "bpx %s records per XCD, next line touched 1 ahead", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_batch_px<F, O, true, 1>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s, d_perm, ntiles, pf); }, {}});
    std::snprintf(label, sizeof label, "bpx %s records per XCD, next line touched 4 ahead", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_batch_px<F, O, true, 4>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s, d_perm, ntiles, pf); }, {}});
    const unsigned grid = (unsigned)ntiles;
    std::snprintf(label, sizeof label, "bpx %s single-tensor kernel + cold record wait, batch schedule", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_one_rec<F, O, 0>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf, d_tab); }, {}});
    std::snprintf(label, sizeof label, "bpx %s single-tensor kernel + cold dst line wait, batch schedule", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_one_rec<F, O, 1>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf, d_tab); }, {}});
    std::snprintf(label, sizeof label, "bpx %s single-tensor kernel, batch schedule", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf); }, {}});
    std::snprintf(label, sizeof label, "bpx %s single-tensor kernel", name);
    vs.push_back({label, bytes, [=](hipStream_t s) {
                      hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, ps); }, {}});
}

// group wait (Prefetch::wait) sweep: single-tensor and batch kernel, wait groups of 4..64 with no burst, against the shipped schedule
template <class F, class O>
void add_group_wait(std::vector<UbVariant> &vs, const char *name, uint32_t ty, uint32_t out_ty, const uint8_t *d_in, void *d_out, uint64_t elems) {
    typedef TileGeom<F, O> G;
    const uint64_t blocks = elems / F::N;
    const uint64_t ntiles = (blocks + G::TB - 1) / G::TB;
    const uint64_t slots = (ntiles + 63) / 64 * 64;
    std::vector<TileRef> tab(slots);
    for (uint64_t t = 0; t < ntiles; ++t) {
        uint64_t nblk = std::min<uint64_t>(G::TB, blocks - t * G::TB);
        tab[record_slot<RecordOrder<O>::PER_XCD>(t)] = {((uint64_t)(uintptr_t)d_in + t * G::TBYTES) | (nblk << 48),
                                                        (uint64_t)(uintptr_t)d_out + t * G::ELEMS * sizeof(typename O::T)};
    }
    TileRef *d_tab;
    CK(hipMalloc(&d_tab, slots * sizeof(TileRef)));
    CK(hipMemcpy(d_tab, tab.data(), slots * sizeof(TileRef), hipMemcpyHostToDevice));
    std::vector<TileRef> tab2(tab);
    for (uint64_t t = 0; t < ntiles; ++t)
        tab2[record_slot<RecordOrder<O>::PER_XCD>(t)].dst |= (uint64_t)std::min<uint64_t>(t, 255) << 48;
    TileRef *d_tab2;
    CK(hipMalloc(&d_tab2, slots * sizeof(TileRef)));
    CK(hipMemcpy(d_tab2, tab2.data(), slots * sizeof(TileRef), hipMemcpyHostToDevice));
    const uint32_t tob = (uint32_t)(G::ELEMS * sizeof(typename O::T));
    const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    const unsigned grid = (unsigned)ntiles;
    auto add_late = [&](const char *kind, const char *what, Prefetch pf, int mode) {
        char label[112];
        std::snprintf(label, sizeof label, "gw %s %s %s", name, kind, what);
        if (mode == 0)
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_one_late<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf); }, {}});
        else if (mode == 1)
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch_late<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab2, ntiles, pf, tob); }, {}});
        else
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch_early<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab2, ntiles, pf, tob); }, {}});
    };
    {   // the tensor cut into 226 pieces of uneven size (whole tiles), as a model's tensor table would be
        const uint32_t NT = 226;
        std::vector<TensorDesc> desc;
        uint64_t tile = 0;
        for (uint32_t i = 0; i < NT && tile < ntiles; ++i) {
            uint64_t want = ntiles / NT * (1 + (i * 7) % 5) / 3 + 1;
            if (i + 1 == NT || tile + want > ntiles) want = ntiles - tile;
            const uint64_t blk0 = tile * G::TB;
            const uint64_t nb = std::min<uint64_t>(want * G::TB, blocks - blk0);
            desc.push_back({(uint64_t)(uintptr_t)d_in + tile * G::TBYTES, (uint64_t)(uintptr_t)d_out + tile * tob, nb, tile});
            tile += want;
        }
        if (tile < ntiles) {   // whatever is left goes to one last piece
            const uint64_t blk0 = tile * G::TB;
            desc.push_back({(uint64_t)(uintptr_t)d_in + tile * G::TBYTES, (uint64_t)(uintptr_t)d_out + tile * tob, blocks - blk0, tile});
        }
        TensorDesc *d_desc;
        CK(hipMalloc(&d_desc, desc.size() * sizeof(TensorDesc)));
        CK(hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(TensorDesc), hipMemcpyHostToDevice));
        const uint32_t nd = (uint32_t)desc.size();
        auto add_bs = [&](const char *what, Prefetch pf) {
            char label[112];
            std::snprintf(label, sizeof label, "gw %s batch-search %s", name, what);
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch_bs<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, d_desc, nd, ntiles, pf); }, {}});
        };
        add_bs("batch schedule", prefetch_for(ty, out_ty, true));
        add_bs("single schedule", prefetch_for(ty, out_ty, false));
        for (uint32_t w : {3u, 4u, 5u, 6u})
            for (uint32_t l : {0u, 1280u, 2048u}) {
                char what[48];
                std::snprintf(what, sizeof what, "W%u %s%u", 1u << w, l ? "P1 L" : "off ", l);
                add_bs(what, l ? Prefetch{1, l, w} : Prefetch{0, 0, w});
            }
    }
    {   // the tensor cut into 226 uneven pieces, K per launch, descriptors in the kernel arguments
        constexpr int K = 32;
        const uint32_t NT = 226;
        struct Piece { uint64_t tile0, tiles, blocks; };
        std::vector<Piece> pieces;
        uint64_t tile = 0;
        for (uint32_t i = 0; i < NT && tile < ntiles; ++i) {
            uint64_t want = ntiles / NT * (1 + (i * 7) % 5) / 3 + 1;
            if (i + 1 == NT || tile + want > ntiles) want = ntiles - tile;
            pieces.push_back({tile, want, std::min<uint64_t>(want * G::TB, blocks - tile * G::TB)});
            tile += want;
        }
        if (tile < ntiles) pieces.push_back({tile, ntiles - tile, blocks - tile * G::TB});
        struct Launch { DescPack<K> pack; unsigned grid; };
        auto launches = std::make_shared<std::vector<Launch>>();
        for (size_t p0 = 0; p0 < pieces.size(); p0 += K) {
            Launch L;
            uint64_t acc = 0;
            for (int j = 0; j < K; ++j) {
                if (p0 + j < pieces.size()) {
                    const Piece &pc = pieces[p0 + j];
                    L.pack.first[j] = (uint32_t)acc;
                    L.pack.src[j] = (uint64_t)(uintptr_t)d_in + pc.tile0 * G::TBYTES;
                    L.pack.dst[j] = (uint64_t)(uintptr_t)d_out + pc.tile0 * tob;
                    L.pack.blocks[j] = pc.blocks;
                    acc += pc.tiles;
                } else {
                    L.pack.first[j] = 0xFFFFFFFFu;
                    L.pack.src[j] = L.pack.dst[j] = L.pack.blocks[j] = 0;
                }
            }
            L.grid = (unsigned)acc;
            launches->push_back(L);
        }
        auto add_multi = [&](const char *what, Prefetch pf) {
            char label[112];
            std::snprintf(label, sizeof label, "gw %s multi%d %s", name, K, what);
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              for (const auto &L : *launches)
                                  hipLaunchKernelGGL((deq_multi<F, O, K>), dim3(L.grid), dim3(WG_THREADS), 0, s, L.pack, pf); }, {}});
        };
        add_multi("single schedule", prefetch_for(ty, out_ty, false));
        add_multi("batch schedule", prefetch_for(ty, out_ty, true));
    }
    {   // probe address by arithmetic over an output arena
        const uint64_t arena_bytes = ntiles * (uint64_t)tob;
        for (uint64_t skew : {0ull, 4ull << 20})
            for (uint32_t w : {3u, 4u, 5u, 6u})
                for (uint32_t l : {0u, 1280u}) {
                    char label[112];
                    std::snprintf(label, sizeof label, "gw %s batch-arena W%u %s%u skew %lluM", name, 1u << w, l ? "P1 L" : "off ", l,
                                  (unsigned long long)(skew >> 20));
                    const Prefetch pf = l ? Prefetch{1, l, w} : Prefetch{0, 0, w};
                    vs.push_back({label, bytes, [=](hipStream_t s) {
                                      hipLaunchKernelGGL((deq_batch_arena<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab, ntiles, pf,
                                                         (const uint8_t *)d_out, arena_bytes, skew); }, {}});
                }
    }
    {   // record and probe through the vector memory path
        auto add_v = [&](const char *what, Prefetch pf, int mode) {
            char label[112];
            std::snprintf(label, sizeof label, "gw %s batch-vec%d %s", name, mode, what);
            if (mode == 1)
                vs.push_back({label, bytes, [=](hipStream_t s) {
                                  hipLaunchKernelGGL((deq_batch_v<F, O, 1>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab, ntiles, pf); }, {}});
            else if (mode == 3)
                vs.push_back({label, bytes, [=](hipStream_t s) {
                                  hipLaunchKernelGGL((deq_batch_v<F, O, 3>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab, ntiles, pf); }, {}});
            else
                vs.push_back({label, bytes, [=](hipStream_t s) {
                                  hipLaunchKernelGGL((deq_batch_v<F, O, 2>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab, ntiles, pf); }, {}});
        };
        add_v("batch schedule", prefetch_for(ty, out_ty, true), 1);
        add_v("single schedule", prefetch_for(ty, out_ty, false), 1);
        add_v("single schedule", prefetch_for(ty, out_ty, false), 3);
        add_v("single schedule", prefetch_for(ty, out_ty, false), 2);
        for (uint32_t w : {3u, 5u})
            for (uint32_t l : {0u, 1280u}) {
                char what[48];
                std::snprintf(what, sizeof what, "W%u %s%u", 1u << w, l ? "P1 L" : "off ", l);
                add_v(what, l ? Prefetch{1, l, w} : Prefetch{0, 0, w}, 3);
                add_v(what, l ? Prefetch{1, l, w} : Prefetch{0, 0, w}, 1);
            }
        char label[112];
        std::snprintf(label, sizeof label, "gw %s single-vprobe single schedule", name);
        const Prefetch ps = prefetch_for(ty, out_ty, false);
        vs.push_back({label, bytes, [=](hipStream_t s) {
                          hipLaunchKernelGGL((deq_one_vprobe<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, ps); }, {}});
    }
    {   // one record per 64 tiles
        const uint64_t nruns = (ntiles + 63) / 64;
        std::vector<RunRef> runs(nruns);
        for (uint64_t r = 0; r < nruns; ++r)
            runs[r] = {(uint64_t)(uintptr_t)d_in + r * 64 * G::TBYTES, (uint64_t)(uintptr_t)d_out + r * 64 * tob, blocks - r * 64 * G::TB, 0};
        RunRef *d_runs;
        CK(hipMalloc(&d_runs, nruns * sizeof(RunRef)));
        CK(hipMemcpy(d_runs, runs.data(), nruns * sizeof(RunRef), hipMemcpyHostToDevice));
        const unsigned grid_r = (unsigned)(nruns * 64);
        auto add_run = [&](const char *what, Prefetch pf) {
            char label[112];
            std::snprintf(label, sizeof label, "gw %s batch-runs %s", name, what);
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch_run<F, O>), dim3(grid_r), dim3(WG_THREADS), 0, s, d_runs, nruns, pf); }, {}});
        };
        add_run("batch schedule", prefetch_for(ty, out_ty, true));
        add_run("single schedule", prefetch_for(ty, out_ty, false));
        for (uint32_t w : {3u, 4u, 5u, 6u})
            for (uint32_t l : {0u, 1280u, 2048u}) {
                char what[48];
                std::snprintf(what, sizeof what, "W%u %s%u", 1u << w, l ? "P1 L" : "off ", l);
                add_run(what, l ? Prefetch{1, l, w} : Prefetch{0, 0, w});
            }
    }
    for (uint32_t w : {0u, 3u, 5u})
        for (int mode = 0; mode < 2; ++mode) {
            char label[112];
            const Prefetch pf{0, 0, w};
            std::snprintf(label, sizeof label, "gw %s single+record-%s W%u off", name, mode ? "unused" : "waited", w ? 1u << w : 0u);
            if (mode == 0) {
                vs.push_back({label, bytes, [=](hipStream_t s) {
                                  hipLaunchKernelGGL((deq_one_wait_rec<F, O, 0>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf, d_tab); }, {}});
                std::snprintf(label, sizeof label, "gw %s single+record-before-probe W%u off", name, w ? 1u << w : 0u);
                vs.push_back({label, bytes, [=](hipStream_t s) {
                                  hipLaunchKernelGGL((deq_one_wait_rec<F, O, 2>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf, d_tab); }, {}});
            }
            else
                vs.push_back({label, bytes, [=](hipStream_t s) {
                                  hipLaunchKernelGGL((deq_one_wait_rec<F, O, 1>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf, d_tab); }, {}});
        }
    for (uint32_t w : {3u, 4u, 5u, 6u}) {
        const uint64_t ng = (ntiles >> w) + 1;
        std::vector<uint64_t> gd(ng);
        for (uint64_t g = 0; g < ng; ++g) gd[g] = (uint64_t)(uintptr_t)d_out + std::min<uint64_t>(g << w, ntiles - 1) * tob;
        uint64_t *d_gd;
        CK(hipMalloc(&d_gd, ng * 8));
        CK(hipMemcpy(d_gd, gd.data(), ng * 8, hipMemcpyHostToDevice));
        for (uint32_t l : {0u, 1280u}) {
            char label[112];
            const Prefetch pf = l ? Prefetch{1, l, w} : Prefetch{0, 0, w};
            std::snprintf(label, sizeof label, "gw %s batch-side-late W%u %s%u", name, 1u << w, l ? "P1 L" : "off ", l);
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch_side<F, O, false>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab, ntiles, pf, d_gd); }, {}});
            std::snprintf(label, sizeof label, "gw %s batch-side-early W%u %s%u", name, 1u << w, l ? "P1 L" : "off ", l);
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch_side<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab, ntiles, pf, d_gd); }, {}});
        }
    }
    for (uint32_t w : {3u, 4u, 5u, 6u})
        for (uint32_t l : {0u, 1280u}) {
            char what[48];
            std::snprintf(what, sizeof what, "W%u %s%u", 1u << w, l ? "P1 L" : "off ", l);
            const Prefetch pf = l ? Prefetch{1, l, w} : Prefetch{0, 0, w};
            add_late("single-late", what, pf, 0);
            add_late("batch-late", what, pf, 1);
            add_late("batch-early", what, pf, 2);
        }
    auto add = [&](const char *kind, const char *what, Prefetch pf, bool batch) {
        char label[112];
        std::snprintf(label, sizeof label, "gw %s %s %s", name, kind, what);
        if (batch)
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_batch<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, d_tab, (uint64_t)0, ntiles, pf); }, {}});
        else
            vs.push_back({label, bytes, [=](hipStream_t s) {
                              hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, d_in, (typename O::T *)d_out, blocks, pf); }, {}});
    };
    for (int batch = 0; batch < 2; ++batch) {
        const char *kind = batch ? "batch" : "single";
        add(kind, "shipped", prefetch_for(ty, out_ty, batch != 0), batch != 0);
        for (uint32_t w : {2u, 3u, 4u, 5u, 6u})
            for (uint32_t l : {0u, 512u, 1280u, 2048u}) {
                char what[48];
                std::snprintf(what, sizeof what, "W%u %s%u", 1u << w, l ? "P1 L" : "off ", l);
                add(kind, what, l ? Prefetch{1, l, w} : Prefetch{0, 0, w}, batch != 0);
            }
    }
}

int main(int argc, char **argv) {
    uint64_t melems = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1024;
    int rounds = argc > 2 ? std::atoi(argv[2]) : 15;
    const uint64_t elems = melems << 20;          // decoded elements per launch
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    std::printf("device CUs %d, %llu Mi elements per launch, %d rounds\n", cus, (unsigned long long)melems, rounds);

    float *d_out;
    uint8_t *d_in;
    CK(hipMalloc(&d_out, elems * 4));
    CK(hipMalloc(&d_in, elems * 2 + 4096));        // large enough to be the (half-size) copy source too
    // pseudo-random payload with benign f16 headers is not needed for timing, but the
    // bytes should not be constant: fill with a cheap pattern
    {
        std::vector<uint32_t> h(1 << 22);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (auto &w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16) & 0x3BFF3BFFu; }
        for (uint64_t off = 0; off < elems * 2; off += h.size() * 4)
            CK(hipMemcpy(d_in + off, h.data(), std::min<uint64_t>(h.size() * 4, elems * 2 - off), hipMemcpyHostToDevice));
    }
    // quantizer probes: packed output goes to a buffer of its own; their f32 source (d_out) starts out as
    // uniform values in (-1, 1), their f16 source is d_in (benign f16 bit patterns)
    uint8_t *d_q;
    CK(hipMalloc(&d_q, elems * 2 + 4096));         // packed blocks (<= 1.13 B / element) or 16-bit elements
    hipLaunchKernelGGL(k_init_f32, dim3(4096), dim3(256), 0, 0, d_out, elems);
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<Variant> vs;
    const uint64_t n16 = elems / 4;
    auto grid_mem = [&](int per_cu) { return dim3((unsigned)(cus * per_cu)); };

#define RSHOT(NT, S, WG, LABEL)                                                                             \
    vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                            \
                      hipLaunchKernelGGL((k_read_shot<NT, S, WG>), dim3((unsigned)((n16 + WG * S - 1) / (WG * S))), \
                                         dim3(WG), 0, s, (float4v *)d_in, (const float4v *)d_out, n16); }, {}});
    RSHOT(false, 1, 256, "read shot plain S1 wg256")
    RSHOT(false, 2, 256, "read shot plain S2 wg256")
    RSHOT(false, 4, 256, "read shot plain S4 wg256")
    RSHOT(false, 2, 512, "read shot plain S2 wg512")
    RSHOT(false, 4, 512, "read shot plain S4 wg512")
    RSHOT(false, 8, 512, "read shot plain S8 wg512")
    RSHOT(true, 2, 512, "read shot nt S2 wg512")
    RSHOT(true, 4, 512, "read shot nt S4 wg512")
    RSHOT(true, 1, 512, "read shot nt S1 wg512")
    RSHOT(false, 1, 512, "read shot plain S1 wg512")
    vs.push_back({"read loop plain 8/CU", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_read_loop<false>, grid_mem(8), dim3(256), 0, s, (float4v *)d_in, (const float4v *)d_out, n16); }, {}});
    vs.push_back({"read loop nt 8/CU", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_read_loop<true>, grid_mem(8), dim3(256), 0, s, (float4v *)d_in, (const float4v *)d_out, n16); }, {}});
    vs.push_back({"read loop nt 16/CU", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_read_loop<true>, grid_mem(16), dim3(256), 0, s, (float4v *)d_in, (const float4v *)d_out, n16); }, {}});
    vs.push_back({"read loop plain 32/CU", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_read_loop<false>, grid_mem(32), dim3(256), 0, s, (float4v *)d_in, (const float4v *)d_out, n16); }, {}});
    vs.push_back({"fill nt grid-stride 8/CU", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_fill<true>, grid_mem(8), dim3(256), 0, s, (float4v *)d_out, n16); }, {}});
    vs.push_back({"fill plain grid-stride 8/CU", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_fill<false>, grid_mem(8), dim3(256), 0, s, (float4v *)d_out, n16); }, {}});
    vs.push_back({"fill nt span 8/CU", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_fill_span<true>, grid_mem(8), dim3(256), 0, s, (float4v *)d_out, n16); }, {}});
    vs.push_back({"fill nt one-shot grid", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_fill<true>, dim3((unsigned)(n16 / 256)), dim3(256), 0, s, (float4v *)d_out, n16); }, {}});
    vs.push_back({"copy nt (half size)", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_copy<true>, grid_mem(8), dim3(256), 0, s, (float4v *)d_out, (const float4v *)d_in, n16 / 2); }, {}});
    vs.push_back({"copy plain (half size)", (double)elems * 4, [=](hipStream_t s) {
                      hipLaunchKernelGGL(k_copy<false>, grid_mem(8), dim3(256), 0, s, (float4v *)d_out, (const float4v *)d_in, n16 / 2); }, {}});

#define DEQ(F, O, NTF, PERCU, LABEL)                                                                        \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        uint64_t ntiles = (blocks + OldGeom<F>::TB - 1) / OldGeom<F>::TB;                                 \
        unsigned grid = (unsigned)std::min<uint64_t>((ntiles + 3) / 4, (uint64_t)cus * PERCU);              \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_tiles<F, O, NTF>), dim3(grid), dim3(OLD_WG_THREADS), 0, s,    \
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks); }, {}}); \
    }
    DEQ(Q4K, OutF32, true, 8, "deq Q4K f32 lds nt 8/CU")
    DEQ(Q4K, OutF32, false, 8, "deq Q4K f32 lds plain 8/CU")
    DEQ(Q4K, OutF32, true, 1000000, "deq Q4K f32 lds nt uncapped")
    DEQ(Q6K, OutF32, true, 8, "deq Q6K f32 lds nt 8/CU")
    DEQ(Q4_0, OutF32, true, 8, "deq Q4_0 f32 lds nt 8/CU")
    DEQ(Q8_0, OutF32, true, 8, "deq Q8_0 f32 lds nt 8/CU")
    DEQ(Q4K, OutF16, true, 8, "deq Q4K f16 lds nt 8/CU")
#define SHOT(F, O, NTF, TE, LABEL)                                                                          \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        uint64_t ntiles = (blocks + OldGeom<F, TE>::TB - 1) / OldGeom<F, TE>::TB;                         \
        unsigned grid = (unsigned)((ntiles + 3) / 4);                                                       \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_oneshot<F, O, NTF, TE>), dim3(grid), dim3(OLD_WG_THREADS), 0, s, \
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks); }, {}}); \
    }
#define WGT(F, O, NTF, TE, LABEL)                                                                           \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + TE / F::N - 1) / (TE / F::N));                                 \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_wgtile<F, O, NTF, TE>), dim3(grid), dim3(OLD_WG_THREADS), 0, s, \
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks); }, {}}); \
    }
#define WGT2(F, O, NTF, TE, THR, LABEL)                                                                     \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + TE / F::N - 1) / (TE / F::N));                                 \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_wgtile<F, O, NTF, TE, THR>), dim3(grid), dim3(THR), 0, s, \
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks); }, {}}); \
    }
#define FSHOT(NTF, S, WGI, LABEL)                                                                           \
    vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                            \
                      hipLaunchKernelGGL((k_fill_shot<NTF, S, WGI>), dim3((unsigned)(n16 / (256 * S))), dim3(256), 0, s, \
                                         (float4v *)d_out, n16); }, {}});
#define FLAY(NTF, S, L, THR, LABEL)                                                                         \
    vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                            \
                      hipLaunchKernelGGL((k_fill_lay<NTF, S, L, THR>), dim3((unsigned)(n16 / (THR * S))), dim3(THR), 0, s, \
                                         (float4v *)d_out, n16); }, {}});
#define FFL(FL, S, THR, LABEL)                                                                              \
    vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                            \
                      hipLaunchKernelGGL((k_fill_fl<FL, S, THR>), dim3((unsigned)(n16 / (THR * S))), dim3(THR), 0, s, \
                                         (float4v *)d_out, n16); }, {}});
    FFL(0, 2, 512, "fl S2 thr512 plain")
    FFL(1, 2, 512, "fl S2 thr512 nt")
    FFL(2, 2, 512, "fl S2 thr512 sc1")
    FFL(3, 2, 512, "fl S2 thr512 sc0 sc1")
    FFL(4, 2, 512, "fl S2 thr512 sc0 sc1 nt")
    FFL(5, 2, 512, "fl S2 thr512 sc0")
    FFL(6, 2, 512, "fl S2 thr512 sc0 nt")
    FFL(7, 2, 512, "fl S2 thr512 sc1 nt")
    FFL(0, 1, 256, "fl S1 thr256 plain")
    FFL(1, 1, 256, "fl S1 thr256 nt")
    FFL(3, 1, 256, "fl S1 thr256 sc0 sc1")
    FFL(0, 4, 256, "fl S4 thr256 plain")
    FFL(3, 4, 256, "fl S4 thr256 sc0 sc1")
    FFL(7, 4, 256, "fl S4 thr256 sc1 nt")
    FLAY(true, 1, 0, 256, "lay S1 thr256 nt")
    FLAY(true, 1, 0, 64, "lay S1 thr64 nt")
    FLAY(true, 1, 0, 1024, "lay S1 thr1024 nt")
    FLAY(true, 2, 0, 256, "lay S2 wgint thr256 nt")
    FLAY(true, 2, 1, 256, "lay S2 wavecontig thr256 nt")
    FLAY(true, 2, 2, 256, "lay S2 threadcontig thr256 nt")
    FLAY(true, 2, 0, 512, "lay S2 wgint thr512 nt")
    FLAY(true, 2, 1, 512, "lay S2 wavecontig thr512 nt")
    FLAY(true, 4, 0, 256, "lay S4 wgint thr256 nt")
    FLAY(true, 4, 1, 256, "lay S4 wavecontig thr256 nt")
    FLAY(true, 4, 2, 256, "lay S4 threadcontig thr256 nt")
    FLAY(true, 8, 1, 256, "lay S8 wavecontig thr256 nt")
    FLAY(false, 2, 1, 256, "lay S2 wavecontig thr256 plain")
    FLAY(false, 4, 1, 256, "lay S4 wavecontig thr256 plain")
    FSHOT(true, 4, true, "fill shot nt S4 wg-interleaved")
    FSHOT(true, 16, true, "fill shot nt S16 wg-interleaved")
    FSHOT(true, 16, false, "fill shot nt S16 wave-contig")
    FSHOT(false, 16, true, "fill shot plain S16 wg-interleaved")
    FSHOT(false, 1, true, "fill shot plain S1")
    SHOT(Q4K, OutF32, true, 4096, "Q4K f32 oneshot wave T4096 nt")
    SHOT(Q4K, OutF32, false, 4096, "Q4K f32 oneshot wave T4096 plain")
    SHOT(Q4K, OutF32, true, 2048, "Q4K f32 oneshot wave T2048 nt")
    SHOT(Q4K, OutF32, true, 1024, "Q4K f32 oneshot wave T1024 nt")
    SHOT(Q4K, OutF32, true, 8192, "Q4K f32 oneshot wave T8192 nt")
    WGT2(Q4K, OutF32, true, 1024, 256, "Q4K f32 wg256 T1024 nt")
    WGT2(Q4K, OutF32, false, 1024, 256, "Q4K f32 wg256 T1024 plain")
    WGT2(Q4K, OutF32, true, 2048, 256, "Q4K f32 wg256 T2048 nt")
    WGT2(Q4K, OutF32, false, 2048, 256, "Q4K f32 wg256 T2048 plain")
    WGT2(Q4K, OutF32, false, 4096, 256, "Q4K f32 wg256 T4096 plain")
    WGT2(Q4K, OutF32, true, 2048, 512, "Q4K f32 wg512 T2048 nt")
    WGT2(Q4K, OutF32, true, 4096, 512, "Q4K f32 wg512 T4096 nt")
    WGT2(Q4K, OutF32, true, 4096, 1024, "Q4K f32 wg1024 T4096 nt")
    WGT2(Q4K, OutF32, false, 4096, 1024, "Q4K f32 wg1024 T4096 plain")
    WGT2(Q4K, OutF32, true, 8192, 1024, "Q4K f32 wg1024 T8192 nt")
    WGT2(Q4K, OutF32, true, 1024, 128, "Q4K f32 wg128 T1024 nt")
    WGT2(Q4K, OutF32, true, 512, 128, "Q4K f32 wg128 T512 nt")
    WGT2(Q4K, OutF32, true, 256, 64, "Q4K f32 wg64 T256 nt")
    WGT2(Q4K, OutF32, true, 512, 64, "Q4K f32 wg64 T512 nt")
    WGT(Q4K, OutF32, true, 4096, "Q4K f32 oneshot wg T4096 nt")
    WGT(Q4K, OutF32, true, 8192, "Q4K f32 oneshot wg T8192 nt")
    WGT(Q4K, OutF32, true, 16384, "Q4K f32 oneshot wg T16384 nt")
    WGT(Q4K, OutF32, false, 16384, "Q4K f32 oneshot wg T16384 plain")
    WGT(Q4K, OutF32, true, 32768, "Q4K f32 oneshot wg T32768 nt")
    WGT(Q6K, OutF32, true, 16384, "Q6K f32 oneshot wg T16384 nt")
    SHOT(Q6K, OutF32, true, 4096, "Q6K f32 oneshot wave T4096 nt")
#define Q4KD(NTF, U, PERCU, LABEL)                                                                         \
    {                                                                                                       \
        uint64_t blocks = elems / 256;                                                                      \
        unsigned grid = (unsigned)std::min<uint64_t>((blocks + 4 * U - 1) / (4 * U), (uint64_t)cus * PERCU); \
        vs.push_back({LABEL, (double)elems * 4.5625, [=](hipStream_t s) {                                   \
                          hipLaunchKernelGGL((q4k_direct<NTF, U>), dim3(grid), dim3(OLD_WG_THREADS), 0, s,      \
                                             (const uint8_t *)d_in, d_out, blocks); }, {}});                \
    }
    Q4KD(true, 4, 8, "deq Q4K f32 direct U4 nt 8/CU")
    Q4KD(true, 4, 1000000, "deq Q4K f32 direct U4 nt oneshot")
    Q4KD(true, 1, 1000000, "deq Q4K f32 direct U1 nt oneshot")

#define WGT3(F, O, NTF, TE, THR, MODE, MINW, LABEL)                                                         \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + TE / F::N - 1) / (TE / F::N));                                 \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_wgtile<F, O, NTF, TE, THR, MODE, MINW>), dim3(grid), dim3(THR), 0, s, \
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks); }, {}}); \
    }
#define WGT4(F, O, NTF, TE, THR, MODE, MINW, PF, LABEL)                                                     \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + TE / F::N - 1) / (TE / F::N));                                 \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_wgtile<F, O, NTF, TE, THR, MODE, MINW, PF>), dim3(grid), dim3(THR), 0, s, \
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks); }, {}}); \
    }
#define SHIP(F, O, LABEL)                                                                                   \
    {                                                                                                       \
        typedef TileGeom<F, O> G;                                                                           \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);                                           \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s,     \
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks,         \
                                             prefetch_for(F##_ID, sizeof(typename O::T) == 4 ? 0u : 1u, false)); }, {}}); \
    }
#define BURST(PERIOD, K, NEAR, LABEL)                                                                       \
    {                                                                                                       \
        uint64_t blocks = elems / 256;                                                                      \
        unsigned grid = (unsigned)((blocks + 15) / 16);                                                     \
        vs.push_back({LABEL, (double)elems * 4.5625, [=](hipStream_t s) {                                   \
                          hipLaunchKernelGGL((deq_burst<Q4K, OutF32, true, 4096, 512, PERIOD, K, NEAR>), dim3(grid), dim3(512), 0, s, \
                                             (const uint8_t *)d_in, d_out, blocks); }, {}});                \
    }
#define BURSTG(F, TE, THR, PERIOD, K, LABEL)                                                                \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + TE / F::N - 1) / (TE / F::N));                                 \
        double bytes = (double)elems * ((double)F::BYTES / F::N + 4);                                       \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_burst<F, OutF32, true, TE, THR, PERIOD, K, 0>), dim3(grid), dim3(THR), 0, s, \
                                             (const uint8_t *)d_in, d_out, blocks); }, {}});                \
    }
#define BURST2(F, CH, K, LABEL)                                                                             \
    {                                                                                                       \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + 4096 / F::N - 1) / (4096 / F::N));                             \
        double bytes = (double)elems * ((double)F::BYTES / F::N + 4);                                       \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_burst2<F, OutF32, true, 4096, 512, CH, K>), dim3(grid), dim3(512), 0, s, \
                                             (const uint8_t *)d_in, d_out, blocks); }, {}});                \
    }
    BURST2(Q4K, 18, 3, "b5 Q4K CH18 K3")
    BURST2(Q4K, 9, 6, "b5 Q4K CH9 K6")
    BURST2(Q4K, 36, 2, "b5 Q4K CH36 K2")
    BURST2(Q4K, 16, 3, "b5 Q4K CH16 K3")
    BURST2(Q4K, 20, 3, "b5 Q4K CH20 K3")
    BURST2(Q6K, 18, 3, "b5 Q6K CH18 K3")
    BURST2(Q6K, 18, 4, "b5 Q6K CH18 K4")
    BURST2(Q6K, 26, 3, "b5 Q6K CH26 K3")
    BURST2(Q6K, 26, 2, "b5 Q6K CH26 K2")
    BURST2(Q6K, 13, 6, "b5 Q6K CH13 K6")
    BURST2(Q2K, 18, 3, "b5 Q2K CH18 K3")
    BURST2(Q2K, 10, 3, "b5 Q2K CH10 K3")
    BURST2(Q2K, 11, 3, "b5 Q2K CH11 K3")
    BURST2(Q8_0, 18, 3, "b5 Q8_0 CH18 K3")
    BURST2(Q8_0, 34, 3, "b5 Q8_0 CH34 K3")
    SHIP(Q6K, OutF32, "b5 ref: ship Q6K f32")
    SHIP(Q2K, OutF32, "b5 ref: ship Q2K f32")
    SHIP(Q8_0, OutF32, "b5 ref: ship Q8_0 f32")
    BURSTG(Q4K, 4096, 512, 512, 3, "b4 Q4K 512x4096 P512 K3")
    BURSTG(Q4K, 2048, 256, 1024, 3, "b4 Q4K 256x2048 P1024 K3")
    BURSTG(Q4K, 2048, 256, 1024, 2, "b4 Q4K 256x2048 P1024 K2")
    BURSTG(Q4K, 4096, 256, 512, 3, "b4 Q4K 256x4096 P512 K3")
    BURSTG(Q4K, 8192, 512, 256, 3, "b4 Q4K 512x8192 P256 K3")
    BURSTG(Q4K, 8192, 1024, 256, 3, "b4 Q4K 1024x8192 P256 K3")
    BURSTG(Q6K, 4096, 512, 512, 3, "b4 Q6K 512x4096 P512 K3")
    BURSTG(Q6K, 4096, 512, 512, 2, "b4 Q6K 512x4096 P512 K2")
    BURSTG(Q6K, 4096, 512, 512, 4, "b4 Q6K 512x4096 P512 K4")
    BURSTG(Q6K, 4096, 512, 384, 4, "b4 Q6K 512x4096 P384 K4")
    BURSTG(Q6K, 4096, 512, 256, 6, "b4 Q6K 512x4096 P256 K6")
    BURSTG(Q6K, 4096, 512, 1024, 2, "b4 Q6K 512x4096 P1024 K2")
    BURSTG(Q4_0, 4096, 512, 512, 3, "b4 Q4_0 512x4096 P512 K3")
    BURSTG(Q8_0, 4096, 512, 512, 3, "b4 Q8_0 512x4096 P512 K3")
    BURSTG(Q2K, 4096, 512, 512, 3, "b4 Q2K 512x4096 P512 K3")
    SHIP(Q6K, OutF32, "b4 ref: ship Q6K f32")
    SHIP(Q4K, OutF32, "b4 ref: ship Q4K f32")
    BURST(512, 3, 0, "b3 P512 K3")
    BURST(512, 4, 0, "b3 P512 K4")
    BURST(512, 5, 0, "b3 P512 K5")
    BURST(512, 6, 0, "b3 P512 K6")
    BURST(512, 8, 0, "b3 P512 K8")
    BURST(384, 4, 0, "b3 P384 K4")
    BURST(384, 5, 0, "b3 P384 K5")
    BURST(768, 2, 0, "b3 P768 K2")
    BURST(768, 3, 0, "b3 P768 K3")
    BURST(640, 3, 0, "b3 P640 K3")
    BURST(448, 4, 0, "b3 P448 K4")
    BURST(1024, 2, 0, "b3 P1024 K2")
    SHIP(Q4K, OutF32, "b3 ref: ship Q4K f32")
    BURST(1024, 1, 0, "b2 P1024 K1")
    BURST(512, 1, 0, "b2 P512 K1")
    BURST(512, 2, 0, "b2 P512 K2")
    BURST(512, 3, 0, "b2 P512 K3")
    BURST(256, 2, 0, "b2 P256 K2")
    BURST(256, 4, 0, "b2 P256 K4")
    BURST(128, 8, 0, "b2 P128 K8")
    BURST(1536, 1, 0, "b2 P1536 K1")
    BURST(1024, 1, 512, "b2 P1024 K1 near512")
    BURST(1024, 1, 256, "b2 P1024 K1 near256")
    BURST(512, 2, 256, "b2 P512 K2 near256")
    SHIP(Q4K, OutF32, "b2 ref: ship Q4K f32")
    BURST(1024, 1, 0, "burst P1024 K1")
    BURST(1024, 2, 0, "burst P1024 K2")
    BURST(2048, 1, 0, "burst P2048 K1")
    BURST(2048, 2, 0, "burst P2048 K2")
    BURST(4096, 1, 0, "burst P4096 K1")
    BURST(8192, 1, 0, "burst P8192 K1")
    BURST(16384, 1, 0, "burst P16384 K1")
    BURST(2048, 2, 1024, "burst P2048 K2 near1024")
    BURST(4096, 2, 1024, "burst P4096 K2 near1024")
    BURST(8192, 1, 1024, "burst P8192 K1 near1024")
    BURST(256, 8, 0, "burst P256 K8")
    SHIP(Q4K, OutF32, "burst ref: ship Q4K f32")
#define SHIPO(F, O, OFF, P, L, LABEL)                                                                       \
    {                                                                                                       \
        typedef TileGeom<F, O> G;                                                                           \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);                                           \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s,     \
                                             (const uint8_t *)d_in + OFF, (typename O::T *)d_out, blocks, Prefetch{P, L}); }, {}}); \
    }
#define SHIPL(F, O, OFF, P, L, LABEL)                                                                       \
    {                                                                                                       \
        typedef TileGeom<F, O> G;                                                                           \
        uint64_t blocks = elems / F::N;                                                                     \
        unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);                                           \
        double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));                   \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((deq_one<F, O, true, false>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                             (const uint8_t *)d_in + OFF, (typename O::T *)d_out, blocks, Prefetch{P, L}); }, {}}); \
    }
    SHIPL(Q4K, OutF32, 0, 768, 1280, "mis plainld Q4K off0 P768 L1280")
    SHIPL(Q4K, OutF32, 16, 768, 1280, "mis plainld Q4K off16 P768 L1280")
    SHIPL(Q4K, OutF32, 2, 768, 1280, "mis plainld Q4K off2 P768 L1280")
    SHIPL(Q6K, OutF32, 0, 512, 768, "mis plainld Q6K off0 P512 L768")
    SHIPL(Q6K, OutF32, 0, 768, 1280, "mis plainld Q6K off0 P768 L1280")
    SHIPL(Q2K, OutF32, 0, 1280, 1280, "mis plainld Q2K off0 P1280 L1280")
    SHIPL(Q3K, OutF32, 0, 1024, 1280, "mis plainld Q3K off0 P1024 L1280")
    SHIPO(Q3K, OutF32, 0, 1024, 1280, "mis Q3K off0 P1024 L1280")
    SHIPO(Q6K, OutF32, 0, 768, 1280, "mis Q6K off0 P768 L1280")
    SHIPO(Q4K, OutF32, 0, 768, 1280, "mis Q4K off0 P768 L1280")
    SHIPO(Q4K, OutF32, 16, 768, 1280, "mis Q4K off16 P768 L1280")
    SHIPO(Q4K, OutF32, 64, 768, 1280, "mis Q4K off64 P768 L1280")
    SHIPO(Q4K, OutF32, 2, 768, 1280, "mis Q4K off2 P768 L1280")
    SHIPO(Q4K, OutF32, 0, 0, 0, "mis Q4K off0 nopf")
    SHIPO(Q4K, OutF32, 16, 0, 0, "mis Q4K off16 nopf")
    SHIPO(Q4K, OutF32, 2, 0, 0, "mis Q4K off2 nopf")
    SHIPO(Q6K, OutF32, 0, 512, 768, "mis Q6K off0 P512 L768")
    SHIPO(Q2K, OutF32, 0, 1280, 1280, "mis Q2K off0 P1280 L1280")
#define QNT(Q, In, LABEL)                                                                                   \
    {                                                                                                       \
        typedef QuantGeom<Q, In> G;                                                                         \
        uint64_t blocks = elems / Q::N;                                                                     \
        unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);                                           \
        double bytes = (double)elems * ((double)Q::BYTES / Q::N + sizeof(typename In::T));                  \
        vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                    \
                          hipLaunchKernelGGL((quant_one<Q, In>), dim3(grid), dim3(G::THR), 0, s,            \
                                             sizeof(typename In::T) == 4 ? (const typename In::T *)d_out : (const typename In::T *)d_in, \
                                             d_q, blocks); }, {}}); \
    }
    {   // the 1-element "blocks": f16 / bf16 widening and narrowing
        const uint64_t n4 = elems / 4, per_wg = (uint64_t)STREAM_THREADS * WidenShape<OutF32>::STEPS;
        const unsigned grid = (unsigned)((n4 + per_wg - 1) / per_wg);
        vs.push_back({"widen f16 -> f32", (double)elems * 6, [=](hipStream_t s) {
                          hipLaunchKernelGGL((widen_vec<false, OutF32, true>), dim3(grid), dim3(STREAM_THREADS), 0, s,
                                             (const uint2v *)d_in, (float *)d_out, n4); }, {}});
        vs.push_back({"widen bf16 -> f32", (double)elems * 6, [=](hipStream_t s) {
                          hipLaunchKernelGGL((widen_vec<true, OutF32, true>), dim3(grid), dim3(STREAM_THREADS), 0, s,
                                             (const uint2v *)d_in, (float *)d_out, n4); }, {}});
        const uint64_t per_wg16 = (uint64_t)STREAM_THREADS * WidenShape<OutBF16>::STEPS;
        const unsigned grid16 = (unsigned)((n4 + per_wg16 - 1) / per_wg16);
        vs.push_back({"widen f16 -> bf16", (double)elems * 4, [=](hipStream_t s) {
                          hipLaunchKernelGGL((widen_vec<false, OutBF16, true>), dim3(grid16), dim3(STREAM_THREADS), 0, s,
                                             (const uint2v *)d_in, (uint16_t *)d_out, n4); }, {}});
        const uint64_t per_wg_n = (uint64_t)STREAM_THREADS * NARROW_STEPS * InF32::EPL;
        const unsigned gridn = (unsigned)((elems + per_wg_n - 1) / per_wg_n);
        vs.push_back({"narrow f32 -> f16", (double)elems * 6, [=](hipStream_t s) {
                          hipLaunchKernelGGL((narrow_vec<InF32, OutF16>), dim3(gridn), dim3(STREAM_THREADS), 0, s,
                                             (const float *)d_out, (uint16_t *)d_q, elems); }, {}});
        vs.push_back({"narrow f32 -> bf16", (double)elems * 6, [=](hipStream_t s) {
                          hipLaunchKernelGGL((narrow_vec<InF32, OutBF16>), dim3(gridn), dim3(STREAM_THREADS), 0, s,
                                             (const float *)d_out, (uint16_t *)d_q, elems); }, {}});
    }
    {   // row gather of permute_qk over opaque packed rows: [4096 x rows] Q4_K (2304-byte rows) and F16 (8192-byte rows)
        for (uint64_t rb : {2304ull, 8192ull}) {
            const uint64_t rows = (elems * 2 / rb) / 256 * 256, dh = 128;
            const uint64_t pieces = rb / 16 * rows, per_wg = STREAM_THREADS;
            const unsigned grid = (unsigned)((pieces + per_wg - 1) / per_wg);
            char label[64];
            std::snprintf(label, sizeof label, "permute_qk rows of %llu bytes", (unsigned long long)rb);
            const FastDiv32 fp = FastDiv32::make((uint32_t)(rb / 16)), fd = FastDiv32::make((uint32_t)dh);
            vs.push_back({label, 2.0 * rb * rows, [=](hipStream_t s) {
                              hipLaunchKernelGGL((permute_qk_rows<16>), dim3(grid), dim3(STREAM_THREADS), 0, s,
                                                 (uint8_t *)d_out, (const uint8_t *)d_in, rb, (uint32_t)pieces, fp, fd, 0u); }, {}});
            std::snprintf(label, sizeof label, "pqk shipped rows of %llu bytes", (unsigned long long)rb);
            vs.push_back({label, 2.0 * rb * rows, [=](hipStream_t s) {
                              hipLaunchKernelGGL((permute_qk_rows<16>), dim3(grid), dim3(STREAM_THREADS), 0, s,
                                                 (uint8_t *)d_out, (const uint8_t *)d_in, rb, (uint32_t)pieces, fp, fd, 0u); }, {}});
            std::snprintf(label, sizeof label, "pqk 64-bit index rows of %llu bytes", (unsigned long long)rb);
            vs.push_back({label, 2.0 * rb * rows, [=](hipStream_t s) {
                              hipLaunchKernelGGL((permute_qk_rows_wide<16>), dim3(grid), dim3(STREAM_THREADS), 0, s,
                                                 (uint8_t *)d_out, (const uint8_t *)d_in, rb, rows, dh, (uint64_t)0); }, {}});
            std::snprintf(label, sizeof label, "pqk plain copy of %llu-byte rows", (unsigned long long)rb);
            vs.push_back({label, 2.0 * rb * rows, [=](hipStream_t s) {
                              hipLaunchKernelGGL(copy16_flat, dim3(grid), dim3(STREAM_THREADS), 0, s,
                                                 (uint4v *)d_out, (const uint4v *)d_in, pieces); }, {}});
        }
    }
    QNT(EncQ4_0, InF32, "quant Q4_0 from f32")
    QNT(EncQ4_1, InF32, "quant Q4_1 from f32")
    QNT(EncQ5_0, InF32, "quant Q5_0 from f32")
    QNT(EncQ5_1, InF32, "quant Q5_1 from f32")
    QNT(EncQ8_0, InF32, "quant Q8_0 from f32")
    QNT(EncQ8_1, InF32, "quant Q8_1 from f32")
    QNT(EncQ8K, InF32, "quant Q8K from f32")
    QNT(EncQ8_0, InF16, "quant Q8_0 from f16")
    QNT(EncQ4_0, InBF16, "quant Q4_0 from bf16")
    SHIP(Q4K, OutF32, "ship Q4K f32")
    SHIP(Q6K, OutF32, "ship Q6K f32")
    SHIP(Q2K, OutF32, "ship Q2K f32")
    SHIP(Q3K, OutF32, "ship Q3K f32")
    SHIP(Q5K, OutF32, "ship Q5K f32")
    SHIP(Q8K, OutF32, "ship Q8K f32")
    SHIP(IQ4XS, OutF32, "ship IQ4XS f32")
    SHIP(IQ4NL, OutF32, "ship IQ4NL f32")
    SHIP(Q4_0, OutF32, "ship Q4_0 f32")
    SHIP(Q4_1, OutF32, "ship Q4_1 f32")
    SHIP(Q5_0, OutF32, "ship Q5_0 f32")
    SHIP(Q5_1, OutF32, "ship Q5_1 f32")
    SHIP(Q8_0, OutF32, "ship Q8_0 f32")
    SHIP(Q8_1, OutF32, "ship Q8_1 f32")
    SHIP(Q4K, OutF16, "ship Q4K f16")
    SHIP(Q6K, OutF16, "ship Q6K f16")
    SHIP(Q4K, OutBF16, "ship Q4K bf16")
    SHIP(Q8_0, OutF16, "ship Q8_0 f16")
    SHIP(Q2K, OutF16, "ship Q2K f16")
    SHIP(Q3K, OutF16, "ship Q3K f16")
    SHIP(Q5K, OutF16, "ship Q5K f16")
    SHIP(Q4_0, OutF16, "ship Q4_0 f16")
    SHIP(IQ4XS, OutBF16, "ship IQ4XS bf16")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 0, "s3 T4096 wg512 nt PF0")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 1024, "s3 T4096 wg512 nt PF1024")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 2048, "s3 T4096 wg512 nt PF2048")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 4096, "s3 T4096 wg512 nt PF4096")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 8192, "s3 T4096 wg512 nt PF8192")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 16384, "s3 T4096 wg512 nt PF16384")
    WGT4(Q4K, OutF32, true, 4096, 512, 4, 1, 4096, "s3 T4096 wg512 nt-st plain-ld PF4096")
    WGT4(Q4K, OutF32, false, 4096, 512, 4, 1, 4096, "s3 T4096 wg512 plain plain PF4096")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 4096, "s3 T2048 wg256 nt PF4096")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 8192, "s3 T2048 wg256 nt PF8192")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 16384, "s3 T2048 wg256 nt PF16384")
    WGT4(Q4K, OutF32, true, 4096, 256, 0, 1, 4096, "s3 T4096 wg256 nt PF4096")
    WGT4(Q4K, OutF32, true, 4096, 256, 0, 1, 8192, "s3 T4096 wg256 nt PF8192")
    WGT4(Q4K, OutF32, true, 8192, 512, 0, 1, 2048, "s3 T8192 wg512 nt PF2048")
    WGT4(Q4K, OutF32, true, 8192, 512, 0, 1, 4096, "s3 T8192 wg512 nt PF4096")
    WGT4(Q4K, OutF32, true, 2048, 512, 0, 1, 8192, "s3 T2048 wg512 nt PF8192")
    WGT4(Q4K, OutF32, true, 1024, 256, 0, 1, 16384, "s3 T1024 wg256 nt PF16384")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 0, "s2 Q4K f32 T4096 wg512 nt")
    WGT4(Q4K, OutF32, false, 4096, 512, 0, 1, 0, "s2 Q4K f32 T4096 wg512 plain")
    WGT4(Q4K, OutF32, true, 8192, 512, 0, 1, 0, "s2 Q4K f32 T8192 wg512 nt")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 0, "s2 Q4K f32 T2048 wg256 nt")
    WGT4(Q4K, OutF32, true, 4096, 256, 0, 1, 0, "s2 Q4K f32 T4096 wg256 nt")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 2, 0, "s2 Q4K f32 T4096 wg512 nt minw2")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 4096, "s2 Q4K f32 T4096 wg512 nt PF4096")
    WGT4(Q4K, OutF32, true, 8192, 1024, 0, 1, 0, "s2 Q4K f32 T8192 wg1024 nt")
    WGT4(Q6K, OutF32, true, 4096, 512, 0, 1, 0, "s2 Q6K f32 T4096 wg512 nt")
    WGT4(Q6K, OutF32, true, 2048, 256, 0, 1, 0, "s2 Q6K f32 T2048 wg256 nt")
    WGT4(Q6K, OutF32, true, 4096, 256, 0, 1, 0, "s2 Q6K f32 T4096 wg256 nt")
    WGT4(Q4_0, OutF32, true, 4096, 512, 0, 1, 0, "s2 Q4_0 f32 T4096 wg512 nt")
    WGT4(Q4_0, OutF32, true, 2048, 256, 0, 1, 0, "s2 Q4_0 f32 T2048 wg256 nt")
    WGT4(Q8_0, OutF32, true, 4096, 512, 0, 1, 0, "s2 Q8_0 f32 T4096 wg512 nt")
    WGT4(Q8_0, OutF32, true, 2048, 256, 0, 1, 0, "s2 Q8_0 f32 T2048 wg256 nt")
    WGT4(Q4K, OutF16, true, 8192, 512, 0, 1, 0, "s2 Q4K f16 T8192 wg512 nt")
    WGT4(Q4K, OutF16, true, 4096, 256, 0, 1, 0, "s2 Q4K f16 T4096 wg256 nt")
    WGT4(Q4K, OutF16, true, 8192, 256, 0, 1, 0, "s2 Q4K f16 T8192 wg256 nt")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 0, "sweep T2048 wg256 nt")
    WGT4(Q4K, OutF32, false, 2048, 256, 0, 1, 0, "sweep T2048 wg256 plain-st")
    WGT4(Q4K, OutF32, true, 2048, 256, 4, 1, 0, "sweep T2048 wg256 nt-st plain-ld")
    WGT4(Q4K, OutF32, false, 2048, 256, 4, 1, 0, "sweep T2048 wg256 plain-st plain-ld")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 2048, "sweep T2048 wg256 nt PF2048")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 4096, "sweep T2048 wg256 nt PF4096")
    WGT4(Q4K, OutF32, true, 2048, 256, 0, 1, 8192, "sweep T2048 wg256 nt PF8192")
    WGT4(Q4K, OutF32, true, 2048, 256, 4, 1, 4096, "sweep T2048 wg256 nt-st plain-ld PF4096")
    WGT4(Q4K, OutF32, false, 2048, 256, 4, 1, 4096, "sweep T2048 wg256 plain plain PF4096")
    WGT4(Q4K, OutF32, true, 4096, 256, 0, 1, 2048, "sweep T4096 wg256 nt PF2048")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 1024, "sweep T4096 wg512 nt PF1024")
    WGT4(Q4K, OutF32, true, 4096, 512, 0, 1, 0, "sweep T4096 wg512 nt")
    WGT3(Q4K, OutF32, true, 2048, 256, 1, 1, "probe Q4K wg256 T2048 nt NOLOAD")
    WGT3(Q4K, OutF32, true, 2048, 256, 2, 1, "probe Q4K wg256 T2048 nt NOSTORE")
    WGT3(Q4K, OutF32, true, 4096, 256, 1, 1, "probe Q4K wg256 T4096 nt NOLOAD")
    WGT3(Q4K, OutF32, true, 4096, 256, 2, 1, "probe Q4K wg256 T4096 nt NOSTORE")
    WGT3(Q4K, OutF32, true, 2048, 256, 0, 2, "probe Q4K wg256 T2048 nt minw2")
    WGT3(Q4K, OutF32, true, 4096, 256, 0, 2, "probe Q4K wg256 T4096 nt minw2")
    vs.push_back({"probe mix shot nt S2", (double)elems * 4 * 1.125, [=](hipStream_t s) {
                      hipLaunchKernelGGL((k_mix_shot<true, 2>), dim3((unsigned)(n16 / 512)), dim3(256), 0, s,
                                         (float4v *)d_out, (const float4v *)d_in, n16); }, {}});
    vs.push_back({"probe mix shot nt S4", (double)elems * 4 * 1.125, [=](hipStream_t s) {
                      hipLaunchKernelGGL((k_mix_shot<true, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, s,
                                         (float4v *)d_out, (const float4v *)d_in, n16); }, {}});
    vs.push_back({"probe mix shot plain S4", (double)elems * 4 * 1.125, [=](hipStream_t s) {
                      hipLaunchKernelGGL((k_mix_shot<false, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, s,
                                         (float4v *)d_out, (const float4v *)d_in, n16); }, {}});
#define CSHOT(NTL, NTS, S, LABEL)                                                                           \
    vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                            \
                      hipLaunchKernelGGL((k_copy_shot<NTL, NTS, S>), dim3((unsigned)(n16 / 2 / (256 * S))), dim3(256), 0, s, \
                                         (float4v *)d_out, (const float4v *)d_in, n16 / 2); }, {}});
    CSHOT(true, true, 1, "cal copy shot S1 ntl nts (half)")
    CSHOT(false, false, 1, "cal copy shot S1 plain (half)")
    CSHOT(true, true, 2, "cal copy shot S2 ntl nts (half)")
    CSHOT(false, false, 2, "cal copy shot S2 plain (half)")
    CSHOT(false, true, 2, "cal copy shot S2 pl-load nt-store")
    CSHOT(false, false, 4, "cal copy shot S4 plain (half)")
#define MSHOT2(NTL, NTS, S, LABEL)                                                                          \
    vs.push_back({LABEL, (double)elems * 4 * 1.125, [=](hipStream_t s) {                                    \
                      hipLaunchKernelGGL((k_mix_shot2<NTL, NTS, S>), dim3((unsigned)(n16 / (256 * S))), dim3(256), 0, s, \
                                         (float4v *)d_out, (const uint32_t *)d_in, n16); }, {}});
    MSHOT2(true, true, 2, "cal mix2 S2 ntl nts")
    MSHOT2(false, false, 2, "cal mix2 S2 plain")
    MSHOT2(false, true, 2, "cal mix2 S2 pl-load nt-store")
    MSHOT2(false, false, 4, "cal mix2 S4 plain")
    FSHOT(false, 1, true, "cal fill shot plain S1")
    FSHOT(true, 2, true, "probe fill shot nt S2")
    FSHOT(false, 2, true, "probe fill shot plain S2")
    // ---- XCD <-> memory locality probes
    {
        char label[96];
        for (uint32_t shift = 0; shift < 8; ++shift) {
            std::snprintf(label, sizeof label, "numa S1 g4K shift%u", shift);
            vs.push_back({label, (double)elems * 4, [=](hipStream_t s) {
                              hipLaunchKernelGGL((k_fill_numa<1>), dim3((unsigned)(n16 / 256)), dim3(256), 0, s,
                                                 (float4v *)d_out, n16, shift, 256u); }, {}});
        }
        for (uint32_t shift : {0u, 1u, 3u}) {
            std::snprintf(label, sizeof label, "numa S4 g4K shift%u", shift);
            vs.push_back({label, (double)elems * 4, [=](hipStream_t s) {
                              hipLaunchKernelGGL((k_fill_numa<4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, s,
                                                 (float4v *)d_out, n16, shift, 256u); }, {}});
            std::snprintf(label, sizeof label, "numa S2 g4K shift%u", shift);
            vs.push_back({label, (double)elems * 4, [=](hipStream_t s) {
                              hipLaunchKernelGGL((k_fill_numa<2>), dim3((unsigned)(n16 / 512)), dim3(256), 0, s,
                                                 (float4v *)d_out, n16, shift, 256u); }, {}});
            std::snprintf(label, sizeof label, "numa S4 g8K shift%u", shift);
            vs.push_back({label, (double)elems * 4, [=](hipStream_t s) {
                              hipLaunchKernelGGL((k_fill_numa<4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, s,
                                                 (float4v *)d_out, n16, shift, 512u); }, {}});
            std::snprintf(label, sizeof label, "numa S4 g16K shift%u", shift);
            vs.push_back({label, (double)elems * 4, [=](hipStream_t s) {
                              hipLaunchKernelGGL((k_fill_numa<4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, s,
                                                 (float4v *)d_out, n16, shift, 1024u); }, {}});
        }
    }
    {
#define NUMAT(S, THR, LABEL)                                                                                \
        vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                        \
                          hipLaunchKernelGGL((k_fill_numa_t<S, THR>), dim3((unsigned)(n16 / (THR * S))), dim3(THR), 0, s, \
                                             (float4v *)d_out, n16); }, {}});
        NUMAT(1, 256, "numat S1 thr256")
        NUMAT(2, 256, "numat S2 thr256")
        NUMAT(3, 256, "numat S3 thr256")
        NUMAT(4, 256, "numat S4 thr256")
        NUMAT(8, 256, "numat S8 thr256")
        NUMAT(1, 512, "numat S1 thr512")
        NUMAT(2, 512, "numat S2 thr512")
        NUMAT(4, 512, "numat S4 thr512")
        NUMAT(1, 1024, "numat S1 thr1024")
        NUMAT(2, 1024, "numat S2 thr1024")
#define NUMAG(S, GR, LABEL)                                                                                 \
        vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                        \
                          hipLaunchKernelGGL((k_fill_numa_g<S, GR>), dim3((unsigned)(n16 / (256 * S))), dim3(256), 0, s, \
                                             (float4v *)d_out, n16); }, {}});
        NUMAG(2, 256, "numag S2 g4K")
        NUMAG(2, 128, "numag S2 g2K")
        NUMAG(2, 64, "numag S2 g1K")
        NUMAG(4, 128, "numag S4 g2K")
    }
    // ---- chunk-strided decode prototype against the shipped kernel
    {
#define XDEQ(F, O, THR, S, P, L, LABEL)                                                                      \
        {                                                                                                   \
            typedef XGeom<F, O, THR, S> G;                                                                  \
            const uint64_t blocks = elems / F::N;                                                           \
            const uint64_t groups = (blocks + 8ull * G::TB - 1) / (8ull * G::TB);                           \
            const unsigned grid = (unsigned)(groups * 8);                                                   \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));         \
            vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                \
                              hipLaunchKernelGGL((deq_one_x<F, O, THR, S, true>), dim3(grid), dim3(THR), 0, s, \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, Prefetch{P, L}); }, {}}); \
        }
        XDEQ(Q4K, OutF32, 512, 2, 768, 1280, "xdeq Q4K f32 thr512 S2 P768 L1280")
        XDEQ(Q4K, OutF32, 512, 2, 0, 0, "xdeq Q4K f32 thr512 S2 nopf")
        XDEQ(Q4K, OutF32, 256, 2, 1536, 2560, "xdeq Q4K f32 thr256 S2 P1536 L2560")
        XDEQ(Q4K, OutF32, 256, 2, 0, 0, "xdeq Q4K f32 thr256 S2 nopf")
        XDEQ(Q4K, OutF32, 512, 1, 1536, 2560, "xdeq Q4K f32 thr512 S1 P1536 L2560")
        XDEQ(Q4K, OutF32, 256, 4, 768, 1280, "xdeq Q4K f32 thr256 S4 P768 L1280")
        XDEQ(Q4K, OutF32, 512, 4, 384, 640, "xdeq Q4K f32 thr512 S4 P384 L640")
        XDEQ(Q4K, OutF32, 256, 1, 3072, 5120, "xdeq Q4K f32 thr256 S1 P3072 L5120")
        XDEQ(Q4K, OutF16, 512, 2, 768, 1280, "xdeq Q4K f16 thr512 S2 P768 L1280")
        XDEQ(Q4K, OutF16, 256, 2, 1536, 2560, "xdeq Q4K f16 thr256 S2 P1536 L2560")
        XDEQ(Q4K, OutF16, 512, 2, 0, 0, "xdeq Q4K f16 thr512 S2 nopf")
        XDEQ(Q6K, OutF32, 512, 2, 1024, 1024, "xdeq Q6K f32 thr512 S2 P1024 L1024")
        XDEQ(Q8_0, OutF32, 512, 2, 640, 1024, "xdeq Q8_0 f32 thr512 S2 P640 L1024")
        SHIP(Q4K, OutF32, "xdeq ref ship Q4K f32")
        SHIP(Q4K, OutF16, "xdeq ref ship Q4K f16")
        SHIP(Q6K, OutF32, "xdeq ref ship Q6K f32")
        SHIP(Q8_0, OutF32, "xdeq ref ship Q8_0 f32")
    }
    // ---- scale-table prototype against the shipped kernel
    {
#define TDEQ(F, O, SUB, LABEL)                                                                              \
        {                                                                                                   \
            typedef TileGeom<F, O> G;                                                                       \
            const uint64_t blocks = elems / F::N;                                                           \
            const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);                                 \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));         \
            vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                \
                              hipLaunchKernelGGL((deq_one_tab<F, O, true, SUB>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks,     \
                                                 prefetch_for(F##_ID, sizeof(typename O::T) == 4 ? 0u : 1u, false)); }, {}}); \
        }
        TDEQ(Q4K, OutF32, 32, "tab Q4K f32")
        TDEQ(Q4K, OutF16, 32, "tab Q4K f16")
        TDEQ(Q5K, OutF32, 32, "tab Q5K f32")
        TDEQ(Q5K, OutF16, 32, "tab Q5K f16")
        TDEQ(Q6K, OutF32, 16, "tab Q6K f32")
        TDEQ(Q6K, OutF16, 16, "tab Q6K f16")
        TDEQ(Q3K, OutF32, 16, "tab Q3K f32")
        TDEQ(Q3K, OutF16, 16, "tab Q3K f16")
        TDEQ(Q2K, OutF32, 16, "tab Q2K f32")
        TDEQ(IQ4XS, OutF32, 32, "tab IQ4XS f32")
        TDEQ(Q4_0, OutF32, 32, "tab Q4_0 f32")
        TDEQ(Q8_0, OutF32, 32, "tab Q8_0 f32")
        TDEQ(Q8_0, OutF16, 32, "tab Q8_0 f16")
        SHIP(Q4K, OutF32, "tab ref ship Q4K f32")
        SHIP(Q4K, OutF16, "tab ref ship Q4K f16")
        SHIP(Q5K, OutF32, "tab ref ship Q5K f32")
        SHIP(Q5K, OutF16, "tab ref ship Q5K f16")
        SHIP(Q6K, OutF32, "tab ref ship Q6K f32")
        SHIP(Q6K, OutF16, "tab ref ship Q6K f16")
        SHIP(Q3K, OutF32, "tab ref ship Q3K f32")
        SHIP(Q3K, OutF16, "tab ref ship Q3K f16")
        SHIP(Q2K, OutF32, "tab ref ship Q2K f32")
        SHIP(IQ4XS, OutF32, "tab ref ship IQ4XS f32")
        SHIP(Q4_0, OutF32, "tab ref ship Q4_0 f32")
        SHIP(Q8_0, OutF32, "tab ref ship Q8_0 f32")
        SHIP(Q8_0, OutF16, "tab ref ship Q8_0 f16")
    }
    // ---- chunk-strided widen / copy probes
    {
        const uint64_t n4 = elems / 4;
#define WX(S, ST, NTL, LABEL)                                                                               \
        vs.push_back({LABEL, (double)elems * 6, [=](hipStream_t s) {                                        \
                          hipLaunchKernelGGL((k_widen_x<S, ST, NTL>), dim3((unsigned)((n4 / 256 / S + 7) / 8 * 8)), dim3(256), 0, s, \
                                             (const uint2v *)d_in, (float *)d_out, n4); }, {}});
        WX(1, false, true, "wx widen S1 contig ntl")
        WX(2, false, true, "wx widen S2 contig ntl")
        WX(2, true, true, "wx widen S2 strided ntl")
        WX(4, true, true, "wx widen S4 strided ntl")
        WX(2, true, false, "wx widen S2 strided plain-ld")
        WX(4, false, true, "wx widen S4 contig ntl")
        {
            const uint64_t per = (uint64_t)STREAM_THREADS * WidenShape<OutF32>::STEPS;
            const unsigned grid = (unsigned)((n4 + per - 1) / per);
            vs.push_back({"wx ref ship widen f16 -> f32", (double)elems * 6, [=](hipStream_t s) {
                              hipLaunchKernelGGL((widen_vec<false, OutF32, true>), dim3(grid), dim3(STREAM_THREADS), 0, s,
                                                 (const uint2v *)d_in, (float *)d_out, n4); }, {}});
        }
        const uint64_t h16 = n16 / 2;        // copy half the buffer onto the other half
#define CX(S, ST, NTL, LABEL)                                                                               \
        vs.push_back({LABEL, (double)elems * 4, [=](hipStream_t s) {                                        \
                          hipLaunchKernelGGL((k_copy_x<S, ST, NTL>), dim3((unsigned)((h16 / 256 / S + 7) / 8 * 8)), dim3(256), 0, s, \
                                             (const float4v *)d_out, (float4v *)d_out + h16, h16); }, {}});
        CX(1, false, true, "wx copy S1 contig ntl")
        CX(2, false, true, "wx copy S2 contig ntl")
        CX(2, true, true, "wx copy S2 strided ntl")
        CX(4, true, true, "wx copy S4 strided ntl")
        CX(2, true, false, "wx copy S2 strided plain-ld")
    }
    // ---- workgroup-shape probes: narrow, quantize, permute
    {
#define NX(THR, S, LABEL)                                                                                   \
        vs.push_back({LABEL, (double)elems * 6, [=](hipStream_t s) {                                        \
                          hipLaunchKernelGGL((narrow_x<InF32, OutF16, THR, S>), dim3((unsigned)(elems / 4 / (THR * S))), dim3(THR), 0, s, \
                                             (const float *)d_out, (uint16_t *)d_in, elems); }, {}});
        NX(256, 1, "shape narrow f32->f16 thr256 S1")
        NX(256, 2, "shape narrow f32->f16 thr256 S2")
        NX(512, 1, "shape narrow f32->f16 thr512 S1")
        NX(512, 2, "shape narrow f32->f16 thr512 S2 (=ship)")
        NX(128, 1, "shape narrow f32->f16 thr128 S1")
#define QX(Q, THR, LPT, LABEL)                                                                              \
        {                                                                                                   \
            const uint64_t blocks = elems / Q::N;                                                           \
            const uint64_t tb = (uint64_t)THR * 4 * LPT / Q::N;                                             \
            vs.push_back({LABEL, (double)elems * (4.0 + (double)Q::BYTES / Q::N), [=](hipStream_t s) {      \
                              hipLaunchKernelGGL((quant_x<Q, InF32, THR, LPT>), dim3((unsigned)((blocks + tb - 1) / tb)), dim3(THR), 0, s, \
                                                 (const float *)d_out, (uint8_t *)d_in, blocks); }, {}});   \
        }
        QX(EncQ8_0, 256, 1, "shape quant Q8_0 thr256 L1")
        QX(EncQ8_0, 256, 2, "shape quant Q8_0 thr256 L2")
        QX(EncQ8_0, 512, 1, "shape quant Q8_0 thr512 L1")
        QX(EncQ8_0, 512, 2, "shape quant Q8_0 thr512 L2 (=ship)")
        QX(EncQ8_0, 256, 4, "shape quant Q8_0 thr256 L4")
        QX(EncQ4_0, 256, 1, "shape quant Q4_0 thr256 L1")
        QX(EncQ4_0, 256, 2, "shape quant Q4_0 thr256 L2")
        QX(EncQ4_0, 512, 2, "shape quant Q4_0 thr512 L2 (=ship)")
        QX(EncQ4_1, 256, 2, "shape quant Q4_1 thr256 L2")
        QX(EncQ4_1, 512, 2, "shape quant Q4_1 thr512 L2 (=ship)")
        for (uint64_t rb : {2304ull, 8192ull}) {
            const uint64_t rows = (elems * 2 / rb) / 256 * 256, dh = 128;
            const uint64_t pieces = rb / 16 * rows;
            char label[96];
#define PX(THR, S)                                                                                          \
            std::snprintf(label, sizeof label, "shape permute rows %llu thr%d S%d", (unsigned long long)rb, THR, S); \
            vs.push_back({label, 2.0 * rb * rows, [=](hipStream_t s) {                                      \
                              hipLaunchKernelGGL((permute_x<THR, S>), dim3((unsigned)((pieces + THR * S - 1) / (THR * S))), dim3(THR), 0, s, \
                                                 (uint8_t *)d_out, (const uint8_t *)d_in, rb, rows, dh); }, {}});
            PX(256, 1) PX(256, 2) PX(512, 1) PX(512, 2)
        }
    }
    // ---- quantizer shape sweep: every encoder, f32 and f16 input
    {
#define QXI(Q, IN, INB, THR, LPT, LABEL)                                                                    \
        {                                                                                                   \
            const uint64_t blocks = elems / Q::N;                                                           \
            const uint64_t tb = (uint64_t)THR * IN::EPL * LPT / Q::N;                                       \
            vs.push_back({LABEL, (double)elems * ((double)INB + (double)Q::BYTES / Q::N), [=](hipStream_t s) { \
                              hipLaunchKernelGGL((quant_x<Q, IN, THR, LPT>), dim3((unsigned)((blocks + tb - 1) / tb)), dim3(THR), 0, s, \
                                                 sizeof(typename IN::T) == 4 ? (const typename IN::T *)d_out : (const typename IN::T *)d_in, \
                                                 d_q, blocks); }, {}}); \
        }
#define QSW(Q, NAME)                                                                                        \
        QXI(Q, InF32, 4, 256, 1, "qsw " NAME " f32 thr256 L1")                                              \
        QXI(Q, InF32, 4, 256, 2, "qsw " NAME " f32 thr256 L2")                                              \
        QXI(Q, InF32, 4, 512, 1, "qsw " NAME " f32 thr512 L1")                                              \
        QXI(Q, InF32, 4, 512, 2, "qsw " NAME " f32 thr512 L2 ship")                                         \
        QXI(Q, InF16, 2, 256, 1, "qsw " NAME " f16 thr256 L1")                                              \
        QXI(Q, InF16, 2, 256, 2, "qsw " NAME " f16 thr256 L2")                                              \
        QXI(Q, InF16, 2, 512, 1, "qsw " NAME " f16 thr512 L1")                                              \
        QXI(Q, InF16, 2, 512, 2, "qsw " NAME " f16 thr512 L2 ship")
        QSW(EncQ4_0, "Q4_0") QSW(EncQ4_1, "Q4_1") QSW(EncQ5_0, "Q5_0") QSW(EncQ5_1, "Q5_1") QSW(EncQ8_0, "Q8_0") QSW(EncQ8_1, "Q8_1")
        QXI(EncQ8K, InF32, 4, 256, 2, "qsw Q8K f32 thr256 L2")
        QXI(EncQ8K, InF32, 4, 512, 1, "qsw Q8K f32 thr512 L1")
        QXI(EncQ8K, InF32, 4, 512, 2, "qsw Q8K f32 thr512 L2 ship")
        QXI(EncQ8K, InF16, 2, 256, 1, "qsw Q8K f16 thr256 L1")
        QXI(EncQ8K, InF16, 2, 256, 2, "qsw Q8K f16 thr256 L2")
        QXI(EncQ8K, InF16, 2, 512, 2, "qsw Q8K f16 thr512 L2 ship")
    }
    // ---- widen shape sweep
    {
        const uint64_t n4 = elems / 4;
#define WS(THR, S, O, OB, LABEL)                                                                            \
        vs.push_back({LABEL, (double)elems * (2 + OB), [=](hipStream_t s) {                                 \
                          hipLaunchKernelGGL((widen_x<false, O, THR, S>), dim3((unsigned)((n4 + THR * S - 1) / (THR * S))), dim3(THR), 0, s, \
                                             (const uint2v *)d_in, (typename O::T *)d_out, n4); }, {}});
        WS(128, 1, OutF32, 4, "wsw widen f16->f32 thr128 S1")
        WS(256, 1, OutF32, 4, "wsw widen f16->f32 thr256 S1")
        WS(512, 1, OutF32, 4, "wsw widen f16->f32 thr512 S1")
        WS(512, 2, OutF32, 4, "wsw widen f16->f32 thr512 S2 ship")
        WS(128, 1, OutBF16, 2, "wsw widen f16->bf16 thr128 S1")
        WS(256, 1, OutBF16, 2, "wsw widen f16->bf16 thr256 S1")
        WS(512, 1, OutBF16, 2, "wsw widen f16->bf16 thr512 S1")
        WS(512, 2, OutBF16, 2, "wsw widen f16->bf16 thr512 S2 ship")
        WS(256, 2, OutBF16, 2, "wsw widen f16->bf16 thr256 S2")
    }
    // ---- workgroup shape of the decode kernel, 16-bit and f32 output, burst prefetch kept at the same BYTES
    {
#define SDEQ(F, O, THR, S, PB, LB, LABEL)                                                                   \
        {                                                                                                   \
            typedef SGeom<F, O, THR, S> G;                                                                  \
            const uint64_t blocks = elems / F::N;                                                           \
            const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);                                 \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));         \
            const uint32_t P = (uint32_t)((PB) / G::TBYTES), L = (uint32_t)((LB) / G::TBYTES);              \
            vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                \
                              hipLaunchKernelGGL((deq_one_s<F, O, THR, S>), dim3(grid), dim3(THR), 0, s,    \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, Prefetch{P, L}); }, {}}); \
        }
        // shipped single-tensor schedule for Q4K f16: {768, 1280} tiles of 4608 bytes; f32: {768, 1280} tiles of 2304 bytes
        const uint64_t PB16 = 768ull * 4608, LB16 = 1280ull * 4608, PB32 = 768ull * 2304, LB32 = 1280ull * 2304;
        SDEQ(Q4K, OutF16, 512, 2, PB16, LB16, "sdeq Q4K f16 thr512 S2 (=ship)")
        SDEQ(Q4K, OutF16, 256, 2, PB16, LB16, "sdeq Q4K f16 thr256 S2")
        SDEQ(Q4K, OutF16, 512, 1, PB16, LB16, "sdeq Q4K f16 thr512 S1")
        SDEQ(Q4K, OutF16, 256, 1, PB16, LB16, "sdeq Q4K f16 thr256 S1")
        SDEQ(Q4K, OutF16, 256, 4, PB16, LB16, "sdeq Q4K f16 thr256 S4")
        SDEQ(Q4K, OutF16, 128, 2, PB16, LB16, "sdeq Q4K f16 thr128 S2")
        SDEQ(Q4K, OutF16, 1024, 1, PB16, LB16, "sdeq Q4K f16 thr1024 S1")
        SDEQ(Q4K, OutF32, 512, 2, PB32, LB32, "sdeq Q4K f32 thr512 S2 (=ship)")
        SDEQ(Q4K, OutF32, 256, 2, PB32, LB32, "sdeq Q4K f32 thr256 S2")
        SDEQ(Q4K, OutF32, 512, 1, PB32, LB32, "sdeq Q4K f32 thr512 S1")
        SDEQ(Q4K, OutF32, 256, 1, PB32, LB32, "sdeq Q4K f32 thr256 S1")
        SDEQ(Q4K, OutF32, 1024, 1, PB32, LB32, "sdeq Q4K f32 thr1024 S1")
        SDEQ(Q4K, OutF32, 128, 2, PB32, LB32, "sdeq Q4K f32 thr128 S2")
        SDEQ(Q6K, OutF16, 512, 2, 0, 0, "sdeq Q6K f16 thr512 S2 nopf (=ship)")
        SDEQ(Q6K, OutF16, 256, 2, 0, 0, "sdeq Q6K f16 thr256 S2 nopf")
        SDEQ(Q6K, OutF16, 512, 1, 0, 0, "sdeq Q6K f16 thr512 S1 nopf")
        SDEQ(Q8_0, OutF16, 512, 2, 0, 0, "sdeq Q8_0 f16 thr512 S2 nopf (=ship)")
        SDEQ(Q8_0, OutF16, 256, 2, 0, 0, "sdeq Q8_0 f16 thr256 S2 nopf")
        SDEQ(Q8_0, OutF16, 512, 1, 0, 0, "sdeq Q8_0 f16 thr512 S1 nopf")
    }
    // ---- Q4_K without LDS against the shipped kernel
    {
#define DDEQ(O, THR, S, PB, LB, LABEL)                                                                      \
        {                                                                                                   \
            constexpr uint32_t ELEMS_ = THR * O::EPL * S, TB_ = ELEMS_ / 256, TBYTES_ = TB_ * 144;          \
            const uint64_t blocks = elems / 256;                                                            \
            const unsigned grid = (unsigned)((blocks + TB_ - 1) / TB_);                                     \
            const double bytes = (double)elems * (0.5625 + sizeof(typename O::T));                          \
            const uint32_t P = (uint32_t)((PB) / TBYTES_), L = (uint32_t)((LB) / TBYTES_);                  \
            vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                \
                              hipLaunchKernelGGL((deq_q4k_direct<O, THR, S>), dim3(grid), dim3(THR), 0, s,  \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, Prefetch{P, L}); }, {}}); \
        }
        const uint64_t PB16 = 768ull * 4608, LB16 = 1280ull * 4608, PB32 = 768ull * 2304, LB32 = 1280ull * 2304;
        DDEQ(OutF16, 256, 1, PB16, LB16, "ddeq Q4K f16 thr256 S1")
        DDEQ(OutF16, 256, 2, PB16, LB16, "ddeq Q4K f16 thr256 S2")
        DDEQ(OutF16, 512, 2, PB16, LB16, "ddeq Q4K f16 thr512 S2")
        DDEQ(OutF16, 256, 4, PB16, LB16, "ddeq Q4K f16 thr256 S4")
        DDEQ(OutF16, 128, 2, PB16, LB16, "ddeq Q4K f16 thr128 S2")
        DDEQ(OutF16, 64, 2, PB16, LB16, "ddeq Q4K f16 thr64 S2")
        DDEQ(OutF32, 256, 1, PB32, LB32, "ddeq Q4K f32 thr256 S1")
        DDEQ(OutF32, 256, 2, PB32, LB32, "ddeq Q4K f32 thr256 S2")
        DDEQ(OutF32, 512, 2, PB32, LB32, "ddeq Q4K f32 thr512 S2")
        DDEQ(OutF32, 256, 4, PB32, LB32, "ddeq Q4K f32 thr256 S4")
        DDEQ(OutF32, 128, 2, PB32, LB32, "ddeq Q4K f32 thr128 S2")
        SHIP(Q4K, OutF32, "ddeq ref ship Q4K f32")
        SHIP(Q4K, OutF16, "ddeq ref ship Q4K f16")
    }
    // ---- tile records laid out per XCD
    {
#define BPX(F, O, OT, NAME) add_batch_px<F, O>(vs, NAME, d_in, d_out, elems, prefetch_for(F##_ID, OT, true), prefetch_for(F##_ID, OT, false));
        BPX(Q4K, OutF32, GGQ_TYPE_F32, "Q4K f32") BPX(Q6K, OutF32, GGQ_TYPE_F32, "Q6K f32") BPX(Q8_0, OutF32, GGQ_TYPE_F32, "Q8_0 f32")
        BPX(Q4K, OutF16, GGQ_TYPE_F16, "Q4K f16") BPX(Q6K, OutF16, GGQ_TYPE_F16, "Q6K f16") BPX(Q2K, OutF16, GGQ_TYPE_F16, "Q2K f16")
        BPX(Q4_0, OutF16, GGQ_TYPE_F16, "Q4_0 f16") BPX(Q5K, OutF16, GGQ_TYPE_F16, "Q5K f16")
    }
    // ---- single-tensor kernel behind a shared wait of G workgroups
    {
#define GATE1(F, O, G, PF, LABEL)                                                                              \
        {                                                                                                      \
            typedef TileGeom<F, O> TG_;                                                                        \
            const uint64_t blocks = elems / F::N;                                                              \
            const unsigned grid = (unsigned)((blocks + TG_::TB - 1) / TG_::TB);                                \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));            \
            const Prefetch pf_ = PF;                                                                           \
            vs.push_back({LABEL, bytes, [=](hipStream_t s) {                                                   \
                              hipLaunchKernelGGL((deq_one_gate<F, O, G>), dim3(grid), dim3(WG_THREADS), 0, s,  \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, pf_); }, {}}); \
        }
#define GATES(F, O, OT, NAME)                                                                                  \
        {                                                                                                      \
            const Prefetch ps = prefetch_for(F##_ID, OT, false);                                               \
            GATE1(F, O, 1, ps, "gate " NAME " G1 (own line, shipped schedule)")                               \
            GATE1(F, O, 8, ps, "gate " NAME " G8 shipped schedule")                                           \
            GATE1(F, O, 16, ps, "gate " NAME " G16 shipped schedule")                                         \
            GATE1(F, O, 8, (Prefetch{0, 0}), "gate " NAME " G8 no prefetch")                                  \
            GATE1(F, O, 16, (Prefetch{0, 0}), "gate " NAME " G16 no prefetch")                                \
            GATE1(F, O, 32, (Prefetch{0, 0}), "gate " NAME " G32 no prefetch")                                \
            GATE1(F, O, 64, (Prefetch{0, 0}), "gate " NAME " G64 no prefetch")                                \
            const uint64_t blocks = elems / F::N;                                                              \
            const unsigned grid = (unsigned)((blocks + TileGeom<F, O>::TB - 1) / TileGeom<F, O>::TB);          \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));            \
            vs.push_back({"gate " NAME " shipped kernel", bytes, [=](hipStream_t s) {                          \
                              hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s,    \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ps); }, {}}); \
            vs.push_back({"gate " NAME " shipped kernel, no prefetch", bytes, [=](hipStream_t s) {             \
                              hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s,    \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, Prefetch{0, 0}); }, {}}); \
        }
        GATES(Q6K, OutF16, 1u, "Q6K f16") GATES(Q8_0, OutF16, 1u, "Q8_0 f16") GATES(Q5K, OutF16, 1u, "Q5K f16")
        GATES(Q4K, OutF16, 1u, "Q4K f16") GATES(Q8_0, OutF32, 0u, "Q8_0 f32") GATES(Q4K, OutF32, 0u, "Q4K f32")
    }
    // ---- prefetch phase by multiplication instead of a 64-bit remainder
    {
#define PFD(F, O, OT, NAME)                                                                                    \
        {                                                                                                      \
            const Prefetch ps = prefetch_for(F##_ID, OT, false);                                               \
            const FastDiv32 per = FastDiv32::make(ps.period ? ps.period : 1u);                                 \
            const uint64_t blocks = elems / F::N;                                                              \
            const unsigned grid = (unsigned)((blocks + TileGeom<F, O>::TB - 1) / TileGeom<F, O>::TB);          \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));            \
            vs.push_back({"pfd " NAME " phase by multiplication", bytes, [=](hipStream_t s) {                  \
                              hipLaunchKernelGGL((deq_one_fd<F, O>), dim3(grid), dim3(WG_THREADS), 0, s,       \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ps, per); }, {}}); \
            vs.push_back({"pfd " NAME " shipped kernel", bytes, [=](hipStream_t s) {                           \
                              hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s,    \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ps); }, {}}); \
        }
        PFD(Q4K, OutF32, 0u, "Q4K f32") PFD(Q4K, OutF16, 1u, "Q4K f16") PFD(Q3K, OutF16, 1u, "Q3K f16") PFD(Q6K, OutF16, 1u, "Q6K f16")
        PFD(Q5K, OutF16, 1u, "Q5K f16") PFD(IQ4XS, OutF16, 1u, "IQ4XS f16") PFD(Q6K, OutF32, 0u, "Q6K f32")
        PFD(IQ4NL, OutF16, 1u, "IQ4NL f16") PFD(Q2K, OutF16, 1u, "Q2K f16") PFD(Q4_0, OutF16, 1u, "Q4_0 f16") PFD(Q3K, OutF32, 0u, "Q3K f32")
        PFD(IQ4XS, OutF32, 0u, "IQ4XS f32")
    }
    // ---- non-temporal tile loads for formats whose tiles are whole 128-byte lines (aligned source)
    {
#define NTLV(F, O, OT, NAME)                                                                                   \
        {                                                                                                      \
            const Prefetch ps = prefetch_for(F##_ID, OT, false);                                               \
            const uint64_t blocks = elems / F::N;                                                              \
            const unsigned grid = (unsigned)((blocks + TileGeom<F, O>::TB - 1) / TileGeom<F, O>::TB);          \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));            \
            vs.push_back({"ntl " NAME " plain loads (shipped)", bytes, [=](hipStream_t s) {                    \
                              hipLaunchKernelGGL((deq_one<F, O, true, false>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ps); }, {}}); \
            vs.push_back({"ntl " NAME " non-temporal loads", bytes, [=](hipStream_t s) {                       \
                              hipLaunchKernelGGL((deq_one<F, O, true, true>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ps); }, {}}); \
        }
        NTLV(Q8_0, OutF32, 0u, "Q8_0 f32") NTLV(Q4K, OutF32, 0u, "Q4K f32") NTLV(Q8_0, OutF16, 1u, "Q8_0 f16") NTLV(Q4K, OutF16, 1u, "Q4K f16")
        NTLV(Q4_0, OutF32, 0u, "Q4_0 f32") NTLV(Q5K, OutF32, 0u, "Q5K f32")
    }
    // ---- wait for a cold destination line first (formats without a burst schedule)
    {
#define CDW(F, O, OT, NAME)                                                                                    \
        {                                                                                                      \
            const uint64_t blocks = elems / F::N;                                                              \
            const unsigned grid = (unsigned)((blocks + TileGeom<F, O>::TB - 1) / TileGeom<F, O>::TB);          \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));            \
            const Prefetch ship = prefetch_for(F##_ID, OT, false);                                             \
            struct { const char *n; Prefetch p; } pfs[] = {{"off", Prefetch{0, 0}}, {"P1 L1280", Prefetch{1, 1280}}, {"shipped", ship}}; \
            for (auto &pp : pfs) {                                                                             \
                const Prefetch pf_ = pp.p;                                                                     \
                char label[96];                                                                                \
                std::snprintf(label, sizeof label, "cdw %s plain kernel, prefetch %s", NAME, pp.n);            \
                vs.push_back({label, bytes, [=](hipStream_t s) {                                               \
                                  hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                                     (const uint8_t *)d_in, (typename O::T *)d_out, blocks, pf_); }, {}}); \
                for (uint32_t gs : {0u, 2u, 3u, 4u, 6u}) {                                                     \
                    std::snprintf(label, sizeof label, "cdw %s group %u, prefetch %s", NAME, 1u << gs, pp.n);  \
                    vs.push_back({label, bytes, [=](hipStream_t s) {                                           \
                                      hipLaunchKernelGGL((deq_one_cdw<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                                         (const uint8_t *)d_in, (typename O::T *)d_out, blocks, pf_, gs); }, {}}); \
                }                                                                                              \
            }                                                                                                  \
        }
        CDW(Q6K, OutF16, 1u, "Q6K f16") CDW(Q5K, OutF16, 1u, "Q5K f16") CDW(Q8_0, OutF16, 1u, "Q8_0 f16") CDW(Q5_1, OutF16, 1u, "Q5_1 f16")
        CDW(Q8_0, OutF32, 0u, "Q8_0 f32") CDW(Q6K, OutF32, 0u, "Q6K f32") CDW(Q4K, OutF16, 1u, "Q4K f16")
    }
    // ---- the same over every format: shipped kernel against the wait with groups of 8 / 16 / 64 and no burst
    {
#define CDA(F, O, OT, NAME)                                                                                    \
        {                                                                                                      \
            const uint64_t blocks = elems / F::N;                                                              \
            const unsigned grid = (unsigned)((blocks + TileGeom<F, O>::TB - 1) / TileGeom<F, O>::TB);          \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));            \
            const Prefetch ship = prefetch_for(F##_ID, OT, false);                                             \
            vs.push_back({"cda " NAME " shipped", bytes, [=](hipStream_t s) {                                  \
                              hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s,    \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ship); }, {}}); \
            struct { const char *n; Prefetch p; } pfs[] = {{"off", Prefetch{0, 0}}, {"P1 L1280", Prefetch{1, 1280}}}; \
            for (auto &pp : pfs)                                                                               \
                for (uint32_t gs : {3u, 4u, 6u}) {                                                             \
                    const Prefetch pf_ = pp.p;                                                                 \
                    char label[96];                                                                            \
                    std::snprintf(label, sizeof label, "cda %s wait group %u, %s", NAME, 1u << gs, pp.n);      \
                    vs.push_back({label, bytes, [=](hipStream_t s) {                                           \
                                      hipLaunchKernelGGL((deq_one_cdw<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                                         (const uint8_t *)d_in, (typename O::T *)d_out, blocks, pf_, gs); }, {}}); \
                }                                                                                              \
        }
#define CDA2(F, NAME) CDA(F, OutF32, 0u, NAME " f32") CDA(F, OutF16, 1u, NAME " f16")
        CDA2(Q4K, "Q4K") CDA2(Q6K, "Q6K") CDA2(Q2K, "Q2K") CDA2(Q3K, "Q3K") CDA2(Q5K, "Q5K") CDA2(Q8K, "Q8K") CDA2(IQ4XS, "IQ4XS")
        CDA2(IQ4NL, "IQ4NL") CDA2(Q4_0, "Q4_0") CDA2(Q4_1, "Q4_1") CDA2(Q5_0, "Q5_0") CDA2(Q5_1, "Q5_1") CDA2(Q8_0, "Q8_0") CDA2(Q8_1, "Q8_1")
    }
    // ---- group wait with more bytes in flight per workgroup (more stores per thread)
    {
#define SWT(F, O, THR, S, NAME)                                                                                 \
        {                                                                                                       \
            typedef SGeom<F, O, THR, S> G;                                                                      \
            const uint64_t blocks = elems / F::N;                                                               \
            const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);                                     \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));             \
            for (uint32_t w : {0u, 2u, 3u, 4u, 5u})                                                             \
                for (uint32_t l : {0u, 1280u}) {                                                                \
                    const uint32_t L = l * 2 / S;   /* the same distance in bytes whatever the tile size */    \
                    const Prefetch pf = l ? Prefetch{1, L, w} : Prefetch{0, 0, w};                              \
                    char label[112];                                                                            \
                    std::snprintf(label, sizeof label, "mbf %s thr%d S%d W%u %s", NAME, THR, S, w ? 1u << w : 0u, l ? "P1" : "off"); \
                    vs.push_back({label, bytes, [=](hipStream_t s) {                                            \
                                      hipLaunchKernelGGL((deq_one_s<F, O, THR, S>), dim3(grid), dim3(THR), 0, s, \
                                                         (const uint8_t *)d_in, (typename O::T *)d_out, blocks, pf); }, {}}); \
                }                                                                                               \
        }
#define SWTS(F, O, NAME) SWT(F, O, 512, 2, NAME) SWT(F, O, 512, 4, NAME) SWT(F, O, 256, 4, NAME) SWT(F, O, 256, 8, NAME) SWT(F, O, 512, 3, NAME)
        SWTS(Q6K, OutF16, "Q6K f16") SWTS(Q8_0, OutF32, "Q8_0 f32") SWTS(Q4K, OutF16, "Q4K f16") SWTS(Q4K, OutF32, "Q4K f32")
    }
    // ---- burst prefetch plus a near touch into the own L2
    {
#define NEAR(F, O, OT, NAME)                                                                                   \
        {                                                                                                      \
            const uint64_t blocks = elems / F::N;                                                              \
            const unsigned grid = (unsigned)((blocks + TileGeom<F, O>::TB - 1) / TileGeom<F, O>::TB);          \
            const double bytes = (double)elems * ((double)F::BYTES / F::N + sizeof(typename O::T));            \
            const Prefetch ship = prefetch_for(F##_ID, OT, false);                                             \
            for (uint32_t near : {0u, 8u, 16u, 32u, 64u, 128u, 256u, 512u, 12u, 100u}) {                       \
                char label[96];                                                                                \
                std::snprintf(label, sizeof label, "near %s +%u", NAME, near);                                 \
                vs.push_back({label, bytes, [=](hipStream_t s) {                                               \
                                  hipLaunchKernelGGL((deq_one_near<F, O>), dim3(grid), dim3(WG_THREADS), 0, s, \
                                                     (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ship, near); }, {}}); \
            }                                                                                                  \
            vs.push_back({"near " NAME " shipped", bytes, [=](hipStream_t s) {                                 \
                              hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s,    \
                                                 (const uint8_t *)d_in, (typename O::T *)d_out, blocks, ship); }, {}}); \
        }
        NEAR(Q4K, OutF32, 0u, "Q4K f32") NEAR(Q6K, OutF32, 0u, "Q6K f32") NEAR(Q2K, OutF16, 1u, "Q2K f16")
    }
    // ---- group start on a plain copy
    {
        const uint64_t total = elems * 2 / 16;          // 2 GiB copied: 2 B per element of `elems`
#define CPG(S)                                                                                                  \
        for (uint32_t w : {0u, 2u, 3u, 4u, 5u, 6u, 8u}) {                                                       \
            char label[96];                                                                                     \
            std::snprintf(label, sizeof label, "cpg copy 256 x %d, group %u", S, w ? 1u << w : 0u);             \
            const unsigned grid = (unsigned)((total + 256 * S - 1) / (256 * S));                                \
            vs.push_back({label, 2.0 * 16 * total, [=](hipStream_t s) {                                         \
                              hipLaunchKernelGGL((copy16_group<S>), dim3(grid), dim3(STREAM_THREADS), 0, s,     \
                                                 (uint4v *)d_out, (const uint4v *)d_in, total, w); }, {}});     \
        }
        CPG(1) CPG(2) CPG(4)
    }
    // ---- occupancy probes: dynamic LDS that the kernel never touches limits the workgroups per CU
    // (160 KiB of LDS per CU), i.e. the bytes of stores a CU has in flight
    {
        static const uint32_t pads[] = {0, 36u << 10, 56u << 10, 100u << 10};
        for (uint32_t X : pads) {
            char label[96];
            auto k1 = k_fill_lay<true, 2, 0, 512>;
            auto k2 = k_fill_lay<true, 1, 0, 256>;
            auto k3 = deq_one<Q4K, OutF32, true>;
            auto k4 = deq_one<Q4K, OutF16, true>;
            CK(hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10));
            CK(hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10));
            CK(hipFuncSetAttribute((const void *)k3, hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10));
            CK(hipFuncSetAttribute((const void *)k4, hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10));
            std::snprintf(label, sizeof label, "occ fill S2 thr512 lds%uK", X >> 10);
            vs.push_back({label, (double)elems * 4, [=](hipStream_t s) {
                              hipLaunchKernelGGL(k1, dim3((unsigned)(n16 / 1024)), dim3(512), X, s, (float4v *)d_out, n16); }, {}});
            std::snprintf(label, sizeof label, "occ fill S1 thr256 lds%uK", X >> 10);
            vs.push_back({label, (double)elems * 4, [=](hipStream_t s) {
                              hipLaunchKernelGGL(k2, dim3((unsigned)(n16 / 256)), dim3(256), X, s, (float4v *)d_out, n16); }, {}});
            {
                typedef TileGeom<Q4K, OutF32> G;
                const uint64_t blocks = elems / 256;
                const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);
                std::snprintf(label, sizeof label, "occ deq Q4K f32 lds%uK", X >> 10);
                vs.push_back({label, (double)elems * 4.5625, [=](hipStream_t s) {
                                  hipLaunchKernelGGL(k3, dim3(grid), dim3(WG_THREADS), X, s, (const uint8_t *)d_in, (float *)d_out,
                                                     blocks, prefetch_for(Q4K_ID, 0u, false)); }, {}});
            }
            {
                typedef TileGeom<Q4K, OutF16> G;
                const uint64_t blocks = elems / 256;
                const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);
                std::snprintf(label, sizeof label, "occ deq Q4K f16 lds%uK", X >> 10);
                vs.push_back({label, (double)elems * 2.5625, [=](hipStream_t s) {
                                  hipLaunchKernelGGL(k4, dim3(grid), dim3(WG_THREADS), X, s, (const uint8_t *)d_in, (uint16_t *)d_out,
                                                     blocks, prefetch_for(Q4K_ID, 1u, false)); }, {}});
            }
        }
    }
    if (argc > 3 && std::strncmp(argv[3], "sw", 2) == 0) {
        vs.clear();
        add_sweep<Q4K, OutF32>(vs, "Q4K f32", d_in, d_out, elems);
        add_sweep<Q6K, OutF32>(vs, "Q6K f32", d_in, d_out, elems);
        add_sweep<Q2K, OutF32>(vs, "Q2K f32", d_in, d_out, elems);
        add_sweep<Q3K, OutF32>(vs, "Q3K f32", d_in, d_out, elems);
        add_sweep<Q5K, OutF32>(vs, "Q5K f32", d_in, d_out, elems);
        add_sweep<Q8K, OutF32>(vs, "Q8K f32", d_in, d_out, elems);
        add_sweep<IQ4XS, OutF32>(vs, "IQ4XS f32", d_in, d_out, elems);
        add_sweep<Q4_0, OutF32>(vs, "Q4_0 f32", d_in, d_out, elems);
        add_sweep<Q4_1, OutF32>(vs, "Q4_1 f32", d_in, d_out, elems);
        add_sweep<Q5_0, OutF32>(vs, "Q5_0 f32", d_in, d_out, elems);
        add_sweep<Q5_1, OutF32>(vs, "Q5_1 f32", d_in, d_out, elems);
        add_sweep<Q8_0, OutF32>(vs, "Q8_0 f32", d_in, d_out, elems);
        add_sweep<Q8_1, OutF32>(vs, "Q8_1 f32", d_in, d_out, elems);
        add_sweep<Q4K, OutF16>(vs, "Q4K f16", d_in, d_out, elems);
        add_sweep<Q6K, OutF16>(vs, "Q6K f16", d_in, d_out, elems);
        add_sweep<Q8_0, OutF16>(vs, "Q8_0 f16", d_in, d_out, elems);
        add_sweep<Q4_0, OutF16>(vs, "Q4_0 f16", d_in, d_out, elems);
        add_sweep<Q2K, OutF16>(vs, "Q2K f16", d_in, d_out, elems);
    }
    if (argc > 3 && std::strcmp(argv[3], "sw16") == 0) {   // single-tensor kernel, every format, f16 output
        vs.clear();
        g_sweep_tag = "sw16";
#define SW16(F) add_sweep<F, OutF16>(vs, #F " f16", d_in, d_out, elems);
        SW16(Q4K) SW16(Q6K) SW16(Q2K) SW16(Q3K) SW16(Q5K) SW16(Q8K) SW16(IQ4XS) SW16(IQ4NL) SW16(Q4_0) SW16(Q4_1)
        SW16(Q5_0) SW16(Q5_1) SW16(Q8_0) SW16(Q8_1)
    }
    if (argc > 3 && std::strcmp(argv[3], "swb16") == 0) {  // batch kernel, every format, f16 output
        vs.clear();
        g_sweep_tag = "swb16";
#define SWB16(F) add_sweep_batch<F, OutF16>(vs, #F " f16", d_in, d_out, elems);
        SWB16(Q4K) SWB16(Q6K) SWB16(Q2K) SWB16(Q3K) SWB16(Q5K) SWB16(Q8K) SWB16(IQ4XS) SWB16(IQ4NL) SWB16(Q4_0) SWB16(Q4_1)
        SWB16(Q5_0) SWB16(Q5_1) SWB16(Q8_0) SWB16(Q8_1)
    }
    if (argc > 4 && std::strncmp(argv[3], "swb", 3) == 0) {
        vs.clear();
        std::string f = argv[4];
#define SWB(F, O, NAME) if (f == NAME) add_sweep_batch<F, O>(vs, NAME, d_in, d_out, elems);
        SWB(Q4K, OutF32, "Q4K_f32") SWB(Q6K, OutF32, "Q6K_f32") SWB(Q2K, OutF32, "Q2K_f32") SWB(Q3K, OutF32, "Q3K_f32")
        SWB(Q5K, OutF32, "Q5K_f32") SWB(Q8K, OutF32, "Q8K_f32") SWB(IQ4XS, OutF32, "IQ4XS_f32") SWB(IQ4NL, OutF32, "IQ4NL_f32")
        SWB(Q4_0, OutF32, "Q4_0_f32") SWB(Q4_1, OutF32, "Q4_1_f32") SWB(Q5_0, OutF32, "Q5_0_f32") SWB(Q5_1, OutF32, "Q5_1_f32")
        SWB(Q8_0, OutF32, "Q8_0_f32") SWB(Q8_1, OutF32, "Q8_1_f32")
        SWB(Q4K, OutF16, "Q4K_f16") SWB(Q6K, OutF16, "Q6K_f16") SWB(Q8_0, OutF16, "Q8_0_f16") SWB(Q4_0, OutF16, "Q4_0_f16")
        SWB(Q2K, OutF16, "Q2K_f16") SWB(Q5K, OutF16, "Q5K_f16") SWB(Q3K, OutF16, "Q3K_f16")
        SWB(Q5_1, OutF16, "Q5_1_f16") SWB(Q4_1, OutF16, "Q4_1_f16") SWB(IQ4XS, OutF16, "IQ4XS_f16") SWB(Q5_0, OutF16, "Q5_0_f16")
    }
    if (argc > 4 && std::strcmp(argv[3], "gw") == 0) {     // ubench <Mi> <rounds> gw <FMT>_<f32|f16>
        vs.clear();
        std::string f = argv[4];
#define GW(F, NAME) if (f == NAME "_f32") add_group_wait<F, OutF32>(vs, NAME "_f32", F##_ID, 0u, d_in, d_out, elems); \
                    if (f == NAME "_f16") add_group_wait<F, OutF16>(vs, NAME "_f16", F##_ID, 1u, d_in, d_out, elems);
        GW(Q4K, "Q4K") GW(Q6K, "Q6K") GW(Q2K, "Q2K") GW(Q3K, "Q3K") GW(Q5K, "Q5K") GW(Q8K, "Q8K") GW(IQ4XS, "IQ4XS") GW(IQ4NL, "IQ4NL")
        GW(Q4_0, "Q4_0") GW(Q4_1, "Q4_1") GW(Q5_0, "Q5_0") GW(Q5_1, "Q5_1") GW(Q8_0, "Q8_0") GW(Q8_1, "Q8_1")
    }
    if (argc > 3) {
        std::vector<Variant> keep;
        // argv[3]: one or more substrings separated by commas; a variant is kept if its name holds any of them
        std::vector<std::string> pats;
        {
            std::string a = argv[3];
            size_t at = 0;
            while (at <= a.size()) {
                size_t c = a.find(',', at);
                if (c == std::string::npos) c = a.size();
                if (c > at) pats.push_back(a.substr(at, c - at));
                at = c + 1;
            }
        }
        for (auto &v : vs)
            for (auto &p : pats)
                if (v.name.find(p) != std::string::npos) { keep.push_back(v); break; }
        vs.swap(keep);
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto &v : vs) { v.run(st); }                // warm-up
    CK(hipStreamSynchronize(st));
    // every round visits the variants in a fresh random order (so no variant always
    // runs behind the same predecessor); one sample = REP back-to-back launches
    // UBENCH_REP (environment): launches per sample; a long run of one kernel behind itself can behave
    // differently from three launches behind some other variant
    const int REP = std::getenv("UBENCH_REP") ? std::max(1, std::atoi(std::getenv("UBENCH_REP"))) : 3;
    std::vector<size_t> order(vs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    uint64_t rng = 0x2545F4914F6CDD1Dull;
    for (int r = 0; r < rounds; ++r) {
        for (size_t i = order.size(); i > 1; --i) {
            rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
            std::swap(order[i - 1], order[rng % i]);
        }
        for (size_t oi : order) {
            Variant &v = vs[oi];
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < REP; ++k) v.run(st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / REP);
        }
    }
    CK(hipGetLastError());
    std::printf("%-36s %10s %10s %12s %8s\n", "variant", "median ms", "best ms", "GB/s(median)", "of 8TB/s");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], best = v.ms[0];
        double gbps = v.bytes / (med * 1e-3) / 1e9;
        std::printf("%-36s %10.4f %10.4f %12.1f %7.1f%%\n", v.name.c_str(), med, best, gbps, gbps / 80.0);
    }
    return 0;
}
