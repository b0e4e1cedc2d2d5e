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
#include <string>
#include <vector>

#include "kernels.h"
#include "kernels_v2.h"

using namespace ggq;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } \
    } while (0)

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

typedef UbVariant Variant;   // name, algorithmic bytes per launch, launcher, samples

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
    CK(hipMalloc(&d_in, elems * 4 + 4096));        // large enough to be the copy source too
    // pseudo-random payload with benign f16 headers is not needed for timing, but the
    // bytes should not be constant: fill with a cheap pattern
    {
        std::vector<uint32_t> h(1 << 22);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (auto &w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16) & 0x3BFF3BFFu; }
        for (uint64_t off = 0; off < elems * 4; off += h.size() * 4)
            CK(hipMemcpy(d_in + off, h.data(), std::min<uint64_t>(h.size() * 4, elems * 4 - off), hipMemcpyHostToDevice));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<Variant> vs;
    const uint64_t n16 = elems / 4;
    auto grid_mem = [&](int per_cu) { return dim3((unsigned)(cus * per_cu)); };

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
                                             (const uint8_t *)d_in, (typename O::T *)d_out, blocks); }, {}}); \
    }
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
    if (argc > 3) {
        std::vector<Variant> keep;
        for (auto &v : vs) if (v.name.find(argv[3]) != std::string::npos) keep.push_back(v);
        vs.swap(keep);
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto &v : vs) { v.run(st); }                // warm-up
    CK(hipStreamSynchronize(st));
    // every round visits the variants in a fresh random order (so no variant always
    // runs behind the same predecessor); one sample = REP back-to-back launches
    const int REP = 3;
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
