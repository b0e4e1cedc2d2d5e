// ggq_api.hip -- host side of libggq: argument checks in the reference's order,
// kernel dispatch, the pinned-staging pipelines and the batch planner.
// The C ABI is declared in include/ggq.h, which cites the reference interface
// each entry point replaces.
#include "../../include/ggq.h"
#include "../../include/ggq_gguf.h"
#include "kernels.h"
#include "quantize.h"
#include "schedule.h"
#include "rearrange.h"
#include "scale.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>

using namespace ggq;

// Device staging buffer of ggq_dequantize_noalloc: one copy per GPU the library is loaded on,
// held until the process ends.  The name is only a host-side handle; the device address of each
// GPU's copy comes from hipGetSymbolAddress (noalloc_state below).  The 256 spare bytes cover the
// decode kernels' 16-byte loads, which may run up to 15 bytes past the last block.
static __device__ __align__(256) uint8_t scratch[GGQ_NOALLOC_MAX_BYTES + 256];

namespace {

// Non-temporal stores for the decoded output: it is written once and never
// re-read by this library, so it should not displace anything in L2 / MALL.
#ifndef GGQ_NT_STORES
#define GGQ_NT_STORES 1
#endif
constexpr bool kNT = GGQ_NT_STORES != 0;

thread_local std::string g_err;

// Measurement knobs, read once: GGQ_NO_PREFETCH=1 switches the burst prefetch and the group start off,
// GGQ_PREFETCH="period,lead[,wait]" replaces the schedule of every kernel.
Prefetch pick_prefetch(uint32_t ty, uint32_t out_ty, bool batch, uint64_t elems = ~0ull) {
    static const int mode = [] {
        const char *e = std::getenv("GGQ_NO_PREFETCH");
        if (e && *e && *e != '0') return 1;
        e = std::getenv("GGQ_PREFETCH");
        return e && *e ? 2 : 0;
    }();
    static const Prefetch forced = [] {
        unsigned period = 0, lead = 0, wait = 0;
        if (const char *e = std::getenv("GGQ_PREFETCH")) std::sscanf(e, "%u,%u,%u", &period, &lead, &wait);
        return Prefetch{period, lead, wait > 16 ? 0u : wait};     // built through the constructor: it derives the divisor
    }();
    if (mode == 1) return Prefetch{0, 0};
    if (mode == 2) return forced;
    return prefetch_for(ty, out_ty, batch, elems);
}

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return false;
}
#define HIP_TRY(call)                                  \
    do {                                               \
        if (!hip_ok((call), #call)) return GGQ_DEVICE; \
    } while (0)

struct TypeInfo {
    uint32_t id, elems, bytes;
    const char *name;
};
// (id, elements per block, bytes per block, layout name).  Bytes are size_of
// the reference's #[repr(C)] structs, except IQ4NL / IQ4XS: the reference
// structs declare u16 quants (34 / 264 bytes) where GGUF files hold the ggml
// blocks of 18 / 136 bytes; the file format wins.
const TypeInfo kTypes[] = {
    {GGQ_TYPE_F16, 1, 2, "f16"},      {GGQ_TYPE_BF16, 1, 2, "bf16"},
    {GGQ_TYPE_Q4_0, 32, 18, "q40"},   {GGQ_TYPE_Q4_1, 32, 20, "q41"},
    {GGQ_TYPE_Q5_0, 32, 22, "q50"},   {GGQ_TYPE_Q5_1, 32, 24, "q51"},
    {GGQ_TYPE_Q8_0, 32, 34, "q80"},   {GGQ_TYPE_Q8_1, 32, 36, "q81"},
    {GGQ_TYPE_Q2K, 256, 84, "q2k"},   {GGQ_TYPE_Q3K, 256, 110, "q3k"},
    {GGQ_TYPE_Q4K, 256, 144, "q4k"},  {GGQ_TYPE_Q5K, 256, 176, "q5k"},
    {GGQ_TYPE_Q6K, 256, 210, "q6k"},  {GGQ_TYPE_Q8K, 256, 290, "q8k"},
    {GGQ_TYPE_IQ4NL, 32, 18, "iq4nl"}, {GGQ_TYPE_IQ4XS, 256, 136, "iq4xs"},
};

const TypeInfo *type_info(uint32_t ty) {
    for (const TypeInfo &t : kTypes)
        if (t.id == ty) return &t;
    return nullptr;
}

size_t out_bytes(uint32_t out_ty) {
    return out_ty == GGQ_TYPE_F32 ? 4 : (out_ty == GGQ_TYPE_F16 || out_ty == GGQ_TYPE_BF16) ? 2 : 0;
}

// Checks shared by every dequantize entry point, in the order of
// ggml-quants/src/lib.rs:136-141.
ggq_status check_slice(uint32_t ty, uint32_t out_ty, size_t dst_elems, size_t src_blocks) {
    const TypeInfo *ti = type_info(ty);
    if (!ti || out_bytes(out_ty) == 0) return GGQ_UNSUPPORTED;
    if (dst_elems % ti->elems != 0) return GGQ_INDIVISIBLE;
    if (src_blocks != dst_elems / ti->elems) return GGQ_LENGTH_MISMATCH;
    return GGQ_OK;
}

// ---- device properties (per device, cached) -----------------------------------
struct DevInfo {
    int cus = 0;
};
std::mutex g_dev_mu;
std::vector<DevInfo> g_dev;

ggq_status dev_info(int dev, DevInfo &out) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if ((size_t)dev >= g_dev.size()) g_dev.resize(dev + 1);
    if (g_dev[dev].cus == 0) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        g_dev[dev].cus = cus > 0 ? cus : 256;
    }
    out = g_dev[dev];
    return GGQ_OK;
}

// Host-side tile geometry, the same numbers as TileGeom<F, O> in kernels.h.
struct HostGeom {
    uint64_t elems, blocks, bytes;   // per tile: output elements, blocks, packed bytes
};
HostGeom host_geom(const TypeInfo *ti, uint32_t out_ty) {
    uint64_t epl = out_ty == GGQ_TYPE_F32 ? 4 : 8;
    uint64_t elems = (uint64_t)WG_THREADS * epl * STORES_PER_THREAD;
    return {elems, elems / ti->elems, elems / ti->elems * ti->bytes};
}
// Workgroups per launch.  A dispatch's total work-item count (grid x 512) must fit 32 bits, so a
// launch carries at most 2^22 workgroups (4 Mi tiles = 16 Gi f32 elements); larger tensors or batch
// groups take several launches.  GGQ_MAX_GRID (environment, read once) lowers the limit so that tests
// can drive the multi-launch path with small inputs.
const uint64_t kMaxGrid = [] {
    uint64_t lim = ((1ull << 32) - 1) / WG_THREADS;
    lim = 1ull << 22 < lim ? 1ull << 22 : lim;
    if (const char *e = std::getenv("GGQ_MAX_GRID")) {
        const unsigned long long v = std::strtoull(e, nullptr, 10);
        if (v >= 1 && v < lim) lim = v;
    }
    return lim;
}();

// Launch plans (ggq_batch_*) are hybrids.  Tensors of one type that follow each other in the caller's table AND in
// memory -- the next one's blocks start where the previous one's end, and so do their outputs -- are one run: blocks
// are independent, so a run decodes as one long tensor.  A type's only run needs no table at all and is launched as
// the single tensor it is.  Among several runs of a type, one of at least this many elements whose single-tensor
// schedule has a group start (schedule.h) leaves the plan's tile table and gets a launch of its own: the table
// kernel cannot take the group start and runs 4-7 points of the roofline below the single-tensor kernel for those
// (type, output) pairs, while a launch costs 3-4 us of ramp and drain (measured on 235 Mi-element tensors: 202
// launches of their own bring the Llama-70B set's Q4_K -> f16 part from 23.28 to 23.07 ms, where 22.3 ms were to
// be had without the launch costs).  A loader that places the tensors of a type back to back therefore gets one
// launch per type at the single-tensor kernel's rate.
// GGQ_HYBRID_MIN_ELEMS (environment, read once) moves the limit: 0 switches the hybrid off, small values let tests
// reach both routes with small tensors.
const uint64_t kHybridMinElems = [] {
    uint64_t lim = 64ull << 20;
    if (const char *e = std::getenv("GGQ_HYBRID_MIN_ELEMS")) {
        const unsigned long long v = std::strtoull(e, nullptr, 10);
        lim = v == 0 ? ~0ull : v;
    }
    return lim;
}();

// GGQ_HYBRID_ANY=1 (environment, read once; measurement knob): runs above the limit get their own launch whatever
// their schedule.
const bool kHybridAny = [] {
    const char *e = std::getenv("GGQ_HYBRID_ANY");
    return e && e[0] == '1';
}();

// GGQ_QUANT_GROUP=k (environment, read once; measurement knob): group start of the single-tensor quantizer kernels
// over 2^k tiles (0 = off) in place of the per-encoder choice.
const int kQuantGroup = [] {
    const char *e = std::getenv("GGQ_QUANT_GROUP");
    if (!e || !*e) return -1;
    const long v = std::strtol(e, nullptr, 10);
    return v < 0 || v > 16 ? -1 : (int)v;
}();

// GGQ_WIDE_INDEX=1 (environment, read once): take the 64-bit index kernels also for small tensors, so that tests
// reach them.
const bool kWideIndex = [] {
    const char *e = std::getenv("GGQ_WIDE_INDEX");
    return e && e[0] == '1';
}();

#define GGQ_FORMATS(X)                                                                      \
    X(GGQ_TYPE_Q4_0, Q4_0) X(GGQ_TYPE_Q4_1, Q4_1) X(GGQ_TYPE_Q5_0, Q5_0) X(GGQ_TYPE_Q5_1, Q5_1) \
    X(GGQ_TYPE_Q8_0, Q8_0) X(GGQ_TYPE_Q8_1, Q8_1) X(GGQ_TYPE_Q2K, Q2K) X(GGQ_TYPE_Q3K, Q3K)     \
    X(GGQ_TYPE_Q4K, Q4K) X(GGQ_TYPE_Q5K, Q5K) X(GGQ_TYPE_Q6K, Q6K) X(GGQ_TYPE_Q8K, Q8K)         \
    X(GGQ_TYPE_IQ4NL, IQ4NL) X(GGQ_TYPE_IQ4XS, IQ4XS)

template <class O> constexpr uint32_t out_id() {
    return sizeof(typename O::T) == 4 ? GGQ_TYPE_F32 : GGQ_TYPE_F16;   // only the width matters here
}

// ---- single tensor --------------------------------------------------------------
template <class O>
ggq_status launch_blocks(uint32_t ty, void *d_dst, const void *d_src, uint64_t blocks, hipStream_t st) {
    switch (ty) {
#define X(ID, F)                                                                              \
    case ID: {                                                                                \
        typedef TileGeom<F, O> G;                                                             \
        const uint64_t ntiles = (blocks + G::TB - 1) / G::TB;                                 \
        for (uint64_t t0 = 0; t0 < ntiles; t0 += kMaxGrid) {                                  \
            uint64_t n = std::min(kMaxGrid, ntiles - t0);                                     \
            hipLaunchKernelGGL((deq_one<F, O, kNT>), dim3((uint32_t)n), dim3(WG_THREADS), 0, st, \
                               (const uint8_t *)d_src + t0 * G::TBYTES,                       \
                               (typename O::T *)d_dst + t0 * G::ELEMS, blocks - t0 * G::TB,   \
                               pick_prefetch(ID, out_id<O>(), false, blocks * F::N));         \
        }                                                                                     \
        break;                                                                                \
    }
        GGQ_FORMATS(X)
#undef X
    default:
        return GGQ_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

template <bool BF16SRC, class O>
ggq_status launch_widen(void *d_dst, const void *d_src, uint64_t n, int cus, hipStream_t st) {
    typedef typename O::T T;
    const bool fast = ((uintptr_t)d_src % 8 == 0);
    const uint64_t n4 = fast ? n / 4 : 0;
    const uint64_t per_wg = (uint64_t)STREAM_THREADS * WidenShape<O>::STEPS;
    for (uint64_t i0 = 0; i0 < n4; i0 += kMaxGrid * per_wg) {
        uint64_t cnt = std::min(kMaxGrid * per_wg, n4 - i0);
        hipLaunchKernelGGL((widen_vec<BF16SRC, O, kNT>), dim3((uint32_t)((cnt + per_wg - 1) / per_wg)),
                           dim3(STREAM_THREADS), 0, st, (const uint2v *)d_src + i0, (T *)d_dst + 4 * i0, cnt);
    }
    if (n4 * 4 < n) {
        uint64_t restn = n - n4 * 4;
        uint64_t wgs = (restn + WG_THREADS - 1) / WG_THREADS;
        uint32_t grid = (uint32_t)std::min<uint64_t>(wgs, (uint64_t)cus * 8);
        hipLaunchKernelGGL((widen_scalar<BF16SRC, O>), dim3(grid), dim3(WG_THREADS), 0, st,
                           (const uint16_t *)d_src, (T *)d_dst, n4 * 4, n);
    }
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

template <class O>
ggq_status launch_any(uint32_t ty, void *d_dst, const void *d_src, uint64_t blocks, int cus,
                      hipStream_t st) {
    if (blocks == 0) return GGQ_OK;
    if (ty == GGQ_TYPE_F16) return launch_widen<false, O>(d_dst, d_src, blocks, cus, st);
    if (ty == GGQ_TYPE_BF16) return launch_widen<true, O>(d_dst, d_src, blocks, cus, st);
    return launch_blocks<O>(ty, d_dst, d_src, blocks, st);
}

ggq_status launch_out(uint32_t ty, uint32_t out_ty, void *d_dst, const void *d_src, uint64_t blocks,
                      int cus, hipStream_t st) {
    switch (out_ty) {
    case GGQ_TYPE_F32: return launch_any<OutF32>(ty, d_dst, d_src, blocks, cus, st);
    case GGQ_TYPE_F16: return launch_any<OutF16>(ty, d_dst, d_src, blocks, cus, st);
    case GGQ_TYPE_BF16: return launch_any<OutBF16>(ty, d_dst, d_src, blocks, cus, st);
    default: return GGQ_UNSUPPORTED;
    }
}

// ---- quantize -----------------------------------------------------------------------
bool can_quantize(uint32_t ty) {
    switch (ty) {
    case GGQ_TYPE_F16: case GGQ_TYPE_BF16: case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q4_1: case GGQ_TYPE_Q5_0:
    case GGQ_TYPE_Q5_1: case GGQ_TYPE_Q8_0: case GGQ_TYPE_Q8_1: case GGQ_TYPE_Q8K: return true;
    default: return false;
    }
}

// Checks of ggml-quants/src/lib.rs:122-127, in that order.
ggq_status check_quantize(uint32_t ty, uint32_t in_ty, size_t dst_blocks, size_t src_elems) {
    const TypeInfo *ti = type_info(ty);
    if (!ti || !can_quantize(ty) || out_bytes(in_ty) == 0) return GGQ_UNSUPPORTED;
    if (src_elems % ti->elems != 0) return GGQ_INDIVISIBLE;
    if (dst_blocks != src_elems / ti->elems) return GGQ_LENGTH_MISMATCH;
    return GGQ_OK;
}

template <class In, class O>
ggq_status launch_narrow(void *d_dst, const void *d_src, uint64_t n, int cus, hipStream_t st) {
    typedef typename In::T TI;
    typedef typename O::T TO;
    const uint64_t nvec = n / In::EPL * In::EPL;
    const uint64_t per_wg = (uint64_t)STREAM_THREADS * NARROW_STEPS * In::EPL;
    for (uint64_t i0 = 0; i0 < nvec; i0 += kMaxGrid * per_wg) {
        uint64_t cnt = std::min(kMaxGrid * per_wg, nvec - i0);
        hipLaunchKernelGGL((narrow_vec<In, O>), dim3((uint32_t)((cnt + per_wg - 1) / per_wg)), dim3(STREAM_THREADS),
                           0, st, (const TI *)d_src + i0, (TO *)d_dst + i0, cnt);
    }
    if (nvec < n) {
        hipLaunchKernelGGL((narrow_scalar<In, O>), dim3(1), dim3(WG_THREADS), 0, st, (const TI *)d_src,
                           (TO *)d_dst, nvec, n);
    }
    (void)cus;
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

template <class In>
ggq_status launch_quant(uint32_t ty, void *d_dst, const void *d_src, uint64_t blocks, int cus, hipStream_t st) {
    if (blocks == 0) return GGQ_OK;
    if (ty == GGQ_TYPE_F16) return launch_narrow<In, OutF16>(d_dst, d_src, blocks, cus, st);
    if (ty == GGQ_TYPE_BF16) return launch_narrow<In, OutBF16>(d_dst, d_src, blocks, cus, st);
    switch (ty) {
#define XQ(ID, Q)                                                                                  \
    case ID: {                                                                                     \
        typedef QuantGeom<Q, In> G;                                                                \
        const uint64_t ntiles = (blocks + G::TB - 1) / G::TB;                                      \
        for (uint64_t t0 = 0; t0 < ntiles; t0 += kMaxGrid) {                                       \
            uint64_t n = std::min(kMaxGrid, ntiles - t0);                                          \
            hipLaunchKernelGGL((quant_one<Q, In>), dim3((uint32_t)n), dim3(G::THR), 0, st,         \
                               (const typename In::T *)d_src + t0 * G::ELEMS,                      \
                               (uint8_t *)d_dst + t0 * G::TBYTES, blocks - t0 * G::TB,             \
                               kQuantGroup < 0 ? QuantGroup<Q>::LOG2 : (uint32_t)kQuantGroup);     \
        }                                                                                          \
        break;                                                                                     \
    }
        XQ(GGQ_TYPE_Q4_0, EncQ4_0) XQ(GGQ_TYPE_Q4_1, EncQ4_1) XQ(GGQ_TYPE_Q5_0, EncQ5_0) XQ(GGQ_TYPE_Q5_1, EncQ5_1)
        XQ(GGQ_TYPE_Q8_0, EncQ8_0) XQ(GGQ_TYPE_Q8_1, EncQ8_1) XQ(GGQ_TYPE_Q8K, EncQ8K)
#undef XQ
    default:
        return GGQ_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

ggq_status launch_quant_in(uint32_t ty, uint32_t in_ty, void *d_dst, const void *d_src, uint64_t blocks, int cus,
                           hipStream_t st) {
    switch (in_ty) {
    case GGQ_TYPE_F32: return launch_quant<InF32>(ty, d_dst, d_src, blocks, cus, st);
    case GGQ_TYPE_F16: return launch_quant<InF16>(ty, d_dst, d_src, blocks, cus, st);
    case GGQ_TYPE_BF16: return launch_quant<InBF16>(ty, d_dst, d_src, blocks, cus, st);
    default: return GGQ_UNSUPPORTED;
    }
}

// Tile geometry of the quantizers on the host: the numbers of QuantGeom<Q, In>.
#define GGQ_ENCODERS(XQ)                                                                                    \
    XQ(GGQ_TYPE_Q4_0, EncQ4_0) XQ(GGQ_TYPE_Q4_1, EncQ4_1) XQ(GGQ_TYPE_Q5_0, EncQ5_0) XQ(GGQ_TYPE_Q5_1, EncQ5_1) \
    XQ(GGQ_TYPE_Q8_0, EncQ8_0) XQ(GGQ_TYPE_Q8_1, EncQ8_1) XQ(GGQ_TYPE_Q8K, EncQ8K)

template <class In>
bool quant_geom(uint32_t ty, HostGeom &g) {
    switch (ty) {
#define XQ(ID, Q) case ID: g = {QuantGeom<Q, In>::ELEMS, QuantGeom<Q, In>::TB, QuantGeom<Q, In>::TBYTES}; return true;
        GGQ_ENCODERS(XQ)
#undef XQ
    default: return false;
    }
}
bool quant_geom_in(uint32_t ty, uint32_t in_ty, HostGeom &g) {
    switch (in_ty) {
    case GGQ_TYPE_F32: return quant_geom<InF32>(ty, g);
    case GGQ_TYPE_F16: return quant_geom<InF16>(ty, g);
    case GGQ_TYPE_BF16: return quant_geom<InBF16>(ty, g);
    default: return false;
    }
}

template <class In>
ggq_status launch_quant_batch(uint32_t ty, const TileRef *d_table, uint64_t ntiles, hipStream_t st) {
    switch (ty) {
#define XQ(ID, Q)                                                                                        \
    case ID:                                                                                             \
        for (uint64_t t0 = 0; t0 < ntiles; t0 += kMaxGrid) {                                             \
            uint64_t n = std::min(kMaxGrid, ntiles - t0);                                                \
            hipLaunchKernelGGL((quant_batch<Q, In>), dim3((uint32_t)n), dim3(QuantGeom<Q, In>::THR), 0, st, d_table, t0); \
        }                                                                                                \
        break;
        GGQ_ENCODERS(XQ)
#undef XQ
    default:
        return GGQ_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

// ---- scaling ----------------------------------------------------------------------
// The factor in the tensor's type, as the reference converts it (to_llama.rs:102-105).  f32 is `scale as f32`.
// f16 is f16::from_f64 of the crate `half`, which on an x86 host with F16C narrows through f32 (two
// round-to-nearest-even steps); bf16 is bf16::from_f64, the crate's software routine, which drops the low 32 bits
// of the f64 before it rounds to nearest even.
static uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x > 0x7F800000u) return (uint16_t)(sign | 0x7E00u | ((x >> 13) & 0x3FFu));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);     // 65520 and above round to infinity
    if (x < 0x38800000u) {
        // below 2^-14: next to 0.5 one unit in the last place is 2^-24, so the addition rounds to half's subnormal grid
        float v;
        memcpy(&v, &x, 4);
        v += 0.5f;
        uint32_t y;
        memcpy(&y, &v, 4);
        return (uint16_t)(sign | (y - 0x3F000000u));
    }
    x += 0xC8000FFFu + ((x >> 13) & 1u);                         // rebias the exponent by -112, round half to even
    return (uint16_t)(sign | (x >> 13));
}

static uint16_t f64_to_bf16_bits(double v) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const uint32_t hi = (uint32_t)(bits >> 32);
    const uint32_t sign = (hi >> 16) & 0x8000u, e11 = (hi >> 20) & 0x7FFu;
    uint32_t frac = hi & 0xFFFFFu;
    if (e11 == 0x7FFu) {
        const bool nan = frac != 0 || (uint32_t)bits != 0;
        return (uint16_t)(sign | 0x7F80u | (nan ? 0x40u : 0u) | (frac >> 13));
    }
    const int e8 = (int)e11 - 1023 + 127;
    if (e8 >= 0xFF) return (uint16_t)(sign | 0x7F80u);
    uint32_t drop = 13, base = 0;
    if (e8 <= 0) {
        if (7 - e8 > 21) return (uint16_t)sign;
        frac |= 0x100000u;
        drop = (uint32_t)(14 - e8);
    } else {
        base = (uint32_t)e8 << 7;
    }
    uint32_t q = base | (frac >> drop);
    const uint32_t round_bit = 1u << (drop - 1);
    if ((frac & round_bit) && (frac & (3u * round_bit - 1u))) ++q;
    return (uint16_t)(sign | q);
}

template <uint32_t K>
static ggq_status launch_scale(void *d_dst, const void *d_src, uint64_t elems, uint64_t factor, hipStream_t q) {
    constexpr uint64_t EB = ScaleOp<K>::BYTES, PER = 16 / EB;
    uint64_t vec_elems = 0;
    if (((uintptr_t)d_dst | (uintptr_t)d_src) % 16 == 0) {
        const uint64_t pieces = elems / PER;
        vec_elems = pieces * PER;
        const uint64_t wgs = (pieces + STREAM_THREADS - 1) / STREAM_THREADS;
        for (uint64_t w0 = 0; w0 < wgs; w0 += kMaxGrid)
            hipLaunchKernelGGL((scale_vec<K>), dim3((uint32_t)std::min(kMaxGrid, wgs - w0)), dim3(STREAM_THREADS), 0, q,
                               (uint4v *)d_dst, (const uint4v *)d_src, pieces, w0 * STREAM_THREADS, factor);
    }
    const uint64_t rest = elems - vec_elems;
    const uint64_t wgs = (rest + STREAM_THREADS - 1) / STREAM_THREADS;
    for (uint64_t w0 = 0; w0 < wgs; w0 += kMaxGrid)
        hipLaunchKernelGGL((scale_scalar<K>), dim3((uint32_t)std::min(kMaxGrid, wgs - w0)), dim3(STREAM_THREADS), 0, q,
                           (uint8_t *)d_dst + vec_elems * EB, (const uint8_t *)d_src + vec_elems * EB, rest,
                           w0 * STREAM_THREADS, factor);
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

// ---- batch ------------------------------------------------------------------------
template <class O>
ggq_status launch_batch(uint32_t ty, const TileRef *d_table, uint64_t ntiles, hipStream_t st) {
    switch (ty) {
#define X(ID, F)                                                                                 \
    case ID:                                                                                     \
        for (uint64_t t0 = 0; t0 < ntiles; t0 += kMaxGrid) {                                     \
            uint64_t n = std::min(kMaxGrid, ntiles - t0);                                        \
            hipLaunchKernelGGL((deq_batch<F, O, kNT>), dim3((uint32_t)n), dim3(WG_THREADS), 0, st, \
                               d_table, t0, ntiles, pick_prefetch(ID, out_id<O>(), true));      \
        }                                                                                        \
        break;
        GGQ_FORMATS(X)
#undef X
    default:
        return GGQ_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

}  // namespace

struct ggq_batch {
    int device = 0;
    int cus = 0;
    uint32_t out_ty = 0;          // decode plans: the output element type; quantize plans: the INPUT element type
    bool quantize = false;        // a plan made by ggq_quant_batch_create: elements -> packed blocks
    struct Group {
        uint32_t ty;
        uint64_t first, ntiles;   // first record in the device tile table (a multiple of 64), tiles in the group
    };
    std::vector<Group> groups;
    // F16 / BF16 tensors are widened one launch each
    struct Plain {
        uint32_t ty;
        const void *src;
        void *dst;
        uint64_t n;
    };
    std::vector<Plain> plain;
    // runs of adjacent block-format tensors large enough for a launch of their own (kHybridMinElems)
    std::vector<Plain> single;
    TileRef *d_table = nullptr;
};

// ---- pinned staging pipeline (host-buffer entry point and ggq_ctx) ----------------
struct ggq_ctx {
    int device = 0;
    int cus = 0;
    static constexpr int SLOTS = 2;
    size_t in_cap = 0;      // bytes per slot, packed input
    size_t out_cap = 0;     // bytes per slot, decoded output (0 for a decode-only context)
    hipStream_t stream[SLOTS] = {};
    hipEvent_t done[SLOTS] = {};
    hipEvent_t fence[SLOTS] = {};   // ggq_ctx_fence: one per stream, re-recorded by every fence
    hipEvent_t gate = nullptr;      // ggq_ctx_wait_stream: recorded on the caller's stream, awaited by both streams
    uint8_t *h_in[SLOTS] = {};
    uint8_t *d_in[SLOTS] = {};
    uint8_t *h_out[SLOTS] = {};
    uint8_t *d_out[SLOTS] = {};
    bool busy[SLOTS] = {};
    int next = 0;           // slot the next chunk takes: slots alternate ACROSS calls too, so that a call's
                            // staging copy overlaps the DMA the call before it started
};

namespace {

// Copies between pageable host memory and the pinned staging buffers run on a few helper
// threads: one thread moves ~10-25 GB/s, PCIe Gen5 moves ~55 GB/s each way, and without help
// the calling thread's memcpy is what bounds the host entry points.  GGQ_COPY_THREADS sets the
// number of helpers (default 3; 0 = the caller copies alone).  The helpers are created on first
// use and never joined (the pool outlives every caller).
class CopyPool {
    struct Task {
        uint8_t *d;
        const uint8_t *s;
        size_t n;
        std::atomic<int> *left;
    };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Task> q;
    int helpers = 0;
    std::once_flag once;

    void work() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !q.empty(); });
                t = q.front();
                q.pop_front();
            }
            memcpy(t.d, t.s, t.n);
            t.left->fetch_sub(1, std::memory_order_release);
        }
    }
    void start() {
        const char *e = std::getenv("GGQ_COPY_THREADS");
        int n = e ? std::atoi(e) : 3;
        n = std::max(0, std::min(n, 15));
        for (int i = 0; i < n; ++i) {
            try {
                std::thread([this] { work(); }).detach();
                ++helpers;
            } catch (...) {
                break;
            }
        }
    }

public:
    void copy(void *dst, const void *src, size_t n) {
        constexpr size_t kMinSlice = 1u << 20;
        std::call_once(once, [this] { start(); });   // helpers is final (and visible) after this
        const size_t parts = std::min<size_t>((size_t)helpers + 1, n / kMinSlice);
        if (parts <= 1) {
            memcpy(dst, src, n);
            return;
        }
        const size_t slice = ((n / parts) + 4095) & ~(size_t)4095;
        std::atomic<int> left{0};
        uint8_t *d = (uint8_t *)dst;
        const uint8_t *s = (const uint8_t *)src;
        size_t off = slice;   // the caller keeps the first slice
        {
            std::lock_guard<std::mutex> lk(m);
            while (off < n) {
                const size_t len = std::min(slice, n - off);
                left.fetch_add(1, std::memory_order_relaxed);
                q.push_back({d + off, s + off, len, &left});
                off += len;
            }
        }
        cv.notify_all();
        memcpy(d, s, std::min(slice, n));
        // The caller takes queued slices too (its own or another caller's), so a copy finishes
        // even if no helper is alive, e.g. in a child process after fork.
        while (left.load(std::memory_order_acquire) != 0) {
            Task t;
            bool have = false;
            {
                std::lock_guard<std::mutex> lk(m);
                if (!q.empty()) {
                    t = q.front();
                    q.pop_front();
                    have = true;
                }
            }
            if (have) {
                memcpy(t.d, t.s, t.n);
                t.left->fetch_sub(1, std::memory_order_release);
            } else {
                std::this_thread::yield();
            }
        }
    }
};

CopyPool &copy_pool() {
    static CopyPool *p = new CopyPool();   // leaked on purpose: helpers may outlive static destructors
    return *p;
}

}  // namespace

namespace {

void ctx_free(ggq_ctx *c) {
    if (!c) return;
    for (int s = 0; s < ggq_ctx::SLOTS; ++s) {
        if (c->stream[s]) (void)hipStreamSynchronize(c->stream[s]);
        if (c->h_in[s]) (void)hipHostFree(c->h_in[s]);
        if (c->h_out[s]) (void)hipHostFree(c->h_out[s]);
        if (c->d_in[s]) (void)hipFree(c->d_in[s]);
        if (c->d_out[s]) (void)hipFree(c->d_out[s]);
        if (c->done[s]) (void)hipEventDestroy(c->done[s]);
        if (c->fence[s]) (void)hipEventDestroy(c->fence[s]);
        if (c->stream[s]) (void)hipStreamDestroy(c->stream[s]);
    }
    if (c->gate) (void)hipEventDestroy(c->gate);
    delete c;
}

// in_cap: packed bytes per slot; out_cap: decoded bytes per slot staged back to
// the host (0 = the decoded tensor stays on the device).
ggq_status ctx_make(int device, size_t in_cap, size_t out_cap, ggq_ctx **out) {
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    ggq_ctx *c = new (std::nothrow) ggq_ctx();
    if (!c) { (void)hipSetDevice(prev); return GGQ_INVALID; }
    c->device = device;
    DevInfo di;
    ggq_status st = dev_info(device, di);
    if (st != GGQ_OK) { delete c; (void)hipSetDevice(prev); return st; }
    c->cus = di.cus;
    c->in_cap = in_cap;
    c->out_cap = out_cap;
    bool ok = true;
    for (int s = 0; s < ggq_ctx::SLOTS && ok; ++s) {
        ok = ok && hip_ok(hipStreamCreateWithFlags(&c->stream[s], hipStreamNonBlocking), "hipStreamCreate");
        ok = ok && hip_ok(hipEventCreateWithFlags(&c->done[s], hipEventDisableTiming), "hipEventCreate");
        ok = ok && hip_ok(hipEventCreateWithFlags(&c->fence[s], hipEventDisableTiming), "hipEventCreate");
        ok = ok && hip_ok(hipHostMalloc((void **)&c->h_in[s], in_cap + 32, hipHostMallocDefault), "hipHostMalloc");
        ok = ok && hip_ok(hipMalloc((void **)&c->d_in[s], in_cap + 32), "hipMalloc");
        if (out_cap) {
            ok = ok && hip_ok(hipHostMalloc((void **)&c->h_out[s], out_cap, hipHostMallocDefault), "hipHostMalloc");
            ok = ok && hip_ok(hipMalloc((void **)&c->d_out[s], out_cap), "hipMalloc");
        }
    }
    ok = ok && hip_ok(hipEventCreateWithFlags(&c->gate, hipEventDisableTiming), "hipEventCreate");
    (void)hipSetDevice(prev);
    if (!ok) { ctx_free(c); return GGQ_DEVICE; }
    *out = c;
    return GGQ_OK;
}

// Pool of host-buffer pipelines, one list per device; a call borrows one and
// returns it.  Never torn down: freeing HIP objects from static destructors
// races the runtime's own shutdown.
std::mutex g_pool_mu;
std::vector<std::vector<ggq_ctx *>> *g_pool = nullptr;

constexpr size_t kHostChunkElems = 4u << 20;   // decoded elements per staged chunk

ggq_status pool_get(int device, ggq_ctx **out) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool) g_pool = new std::vector<std::vector<ggq_ctx *>>();
        if ((size_t)device >= g_pool->size()) g_pool->resize(device + 1);
        auto &v = (*g_pool)[device];
        if (!v.empty()) {
            *out = v.back();
            v.pop_back();
            return GGQ_OK;
        }
    }
    // worst case bytes per element of packed input is Q8K's 290/256 (< 2)
    return ctx_make(device, kHostChunkElems * 2, kHostChunkElems * 4, out);
}

void pool_put(ggq_ctx *c) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    (*g_pool)[c->device].push_back(c);
}

}  // namespace

namespace {
// Builds the device tile table of a plan from its runs: one TileRef per tile, the runs of one type consecutive
// (caller's order inside a type).  The host only writes one descriptor per run; a small kernel per group expands
// them.  geom(ty) gives the tile geometry; tile_in / tile_out = bytes a whole tile reads / writes.
struct PlanRun { uint32_t ty; uint64_t src, dst, blocks; };
template <class GeomOf>
ggq_status build_plan_table(ggq_batch *b, const std::vector<PlanRun> &runs, const std::vector<uint32_t> &types,
                            bool per_xcd, GeomOf geom) {
    std::vector<TensorDesc> descs;
    struct Span { size_t desc0, ndesc; };
    std::vector<Span> spans;
    uint64_t total_tiles = 0;
    for (uint32_t ty : types) {
        uint64_t tile_blocks, tile_in, tile_out;
        geom(ty, tile_blocks, tile_in, tile_out);
        ggq_batch::Group g{ty, total_tiles, 0};
        Span sp{descs.size(), 0};
        for (const PlanRun &r : runs) {
            if (r.ty != ty) continue;
            if (r.src >> 48) return GGQ_INVALID;
            descs.push_back({r.src, r.dst, r.blocks, g.ntiles});
            g.ntiles += (r.blocks + tile_blocks - 1) / tile_blocks;
            ++sp.ndesc;
        }
        total_tiles += (g.ntiles + 63) / 64 * 64;      // record_slot() permutes within runs of 64
        b->groups.push_back(g);
        spans.push_back(sp);
    }
    if (total_tiles == 0) return GGQ_OK;
    TensorDesc *d_desc = nullptr;
    bool ok = hip_ok(hipMalloc((void **)&b->d_table, total_tiles * sizeof(TileRef)), "hipMalloc") &&
              hip_ok(hipMalloc((void **)&d_desc, descs.size() * sizeof(TensorDesc)), "hipMalloc") &&
              hip_ok(hipMemcpy(d_desc, descs.data(), descs.size() * sizeof(TensorDesc), hipMemcpyHostToDevice), "hipMemcpy");
    for (size_t k = 0; ok && k < b->groups.size(); ++k) {
        const auto &g = b->groups[k];
        uint64_t tile_blocks, tile_in, tile_out;
        geom(g.ty, tile_blocks, tile_in, tile_out);
        for (uint64_t t0 = 0; t0 < g.ntiles; t0 += kMaxGrid * 256) {
            const uint64_t end = std::min(g.ntiles, t0 + kMaxGrid * 256);
            const dim3 grid((uint32_t)((end - t0 + 255) / 256)), wg(256);
            if (per_xcd)
                hipLaunchKernelGGL(build_tile_table<true>, grid, wg, 0, (hipStream_t)0, b->d_table + g.first,
                                   d_desc + spans[k].desc0, (uint32_t)spans[k].ndesc, t0, end, (uint32_t)tile_blocks,
                                   (uint32_t)tile_in, (uint32_t)tile_out);
            else
                hipLaunchKernelGGL(build_tile_table<false>, grid, wg, 0, (hipStream_t)0, b->d_table + g.first,
                                   d_desc + spans[k].desc0, (uint32_t)spans[k].ndesc, t0, end, (uint32_t)tile_blocks,
                                   (uint32_t)tile_in, (uint32_t)tile_out);
        }
        ok = hip_ok(hipGetLastError(), "build_tile_table");
    }
    ok = ok && hip_ok(hipStreamSynchronize((hipStream_t)0), "hipStreamSynchronize");
    if (d_desc) (void)hipFree(d_desc);
    return ok ? GGQ_OK : GGQ_DEVICE;
}
}  // namespace

// ---- per-device state of ggq_dequantize_noalloc ------------------------------------
namespace {
constexpr int kNoallocMaxDevices = 64;
struct NoallocDev {
    void *ptr = nullptr;   // device address of this GPU's copy of `scratch`
    int cus = 0;
};
std::mutex g_noalloc_mu;
NoallocDev g_noalloc[kNoallocMaxDevices];

// Resolves, once per device and with that device current, where its copy of `scratch` lives.
ggq_status noalloc_state(int dev, NoallocDev &out) {
    if (dev < 0 || dev >= kNoallocMaxDevices) { g_err = "device index out of range"; return GGQ_DEVICE; }
    std::lock_guard<std::mutex> lk(g_noalloc_mu);
    NoallocDev &d = g_noalloc[dev];
    if (d.ptr == nullptr) {
        void *p = nullptr;
        int cus = 0;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (p == nullptr || (uintptr_t)p % 256 != 0) { g_err = "scratch symbol did not resolve"; return GGQ_DEVICE; }
        d.cus = cus > 0 ? cus : 256;
        d.ptr = p;
    }
    out = d;
    return GGQ_OK;
}
}  // namespace


// =================================== C ABI =======================================
extern "C" {

size_t ggq_block_elems(uint32_t ty) {
    const TypeInfo *t = type_info(ty);
    return t ? t->elems : 0;
}

size_t ggq_block_bytes(uint32_t ty) {
    const TypeInfo *t = type_info(ty);
    return t ? t->bytes : 0;
}

const char *ggq_type_name(uint32_t ty) {
    const TypeInfo *t = type_info(ty);
    return t ? t->name : nullptr;
}

ggq_status ggq_elements_to_bytes(uint32_t ty, const uint64_t *shape, size_t ndim, uint64_t *bytes_out) {
    if (!bytes_out || (ndim && !shape)) return GGQ_INVALID;
    uint64_t blk, ele;
    if (ty == GGQ_TYPE_F32) { blk = 1; ele = 4; }
    else {
        const TypeInfo *t = type_info(ty);
        if (!t) return GGQ_UNSUPPORTED;
        blk = t->elems; ele = t->bytes;
    }
    if (ndim == 0) {
        if (blk != 1) return GGQ_INDIVISIBLE;
        *bytes_out = ele;
        return GGQ_OK;
    }
    if (shape[0] % blk != 0) return GGQ_INDIVISIBLE;
    uint64_t others = 1;
    for (size_t i = 1; i < ndim; ++i) others *= shape[i];
    *bytes_out = others * shape[0] / blk * ele;
    return GGQ_OK;
}

const char *ggq_status_str(ggq_status st) {
    switch (st) {
    case GGQ_OK: return "ok";
    case GGQ_INDIVISIBLE: return "indivisible: element count is not a multiple of the block size";
    case GGQ_LENGTH_MISMATCH: return "length mismatch: block count does not match element count";
    case GGQ_UNSUPPORTED: return "unsupported type";
    case GGQ_DEVICE: return "HIP device error";
    case GGQ_MISALIGNED: return "device pointer is not aligned as required";
    case GGQ_INVALID: return "invalid argument";
    case GGQ_TOO_LARGE: return "input exceeds the fixed capacity of this entry point";
    }
    return "unknown status";
}

const char *ggq_last_error(void) { return g_err.c_str(); }

int ggq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

ggq_status ggq_dequantize_device(uint32_t ty, uint32_t out_ty, void *d_dst, size_t dst_elems,
                                 const void *d_src, size_t src_blocks, void *hip_stream) {
    ggq_status st = check_slice(ty, out_ty, dst_elems, src_blocks);
    if (st != GGQ_OK) return st;
    if (src_blocks == 0) return GGQ_OK;
    if (!d_dst || !d_src) return GGQ_INVALID;
    if ((uintptr_t)d_dst % 16 != 0) return GGQ_MISALIGNED;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DevInfo di;
    st = dev_info(dev, di);
    if (st != GGQ_OK) return st;
    return launch_out(ty, out_ty, d_dst, d_src, src_blocks, di.cus, (hipStream_t)hip_stream);
}

ggq_status ggq_dequantize_slice(uint32_t ty, uint32_t out_ty, void *dst, size_t dst_elems,
                                const void *src, size_t src_blocks) {
    ggq_status st = check_slice(ty, out_ty, dst_elems, src_blocks);
    if (st != GGQ_OK) return st;
    if (src_blocks == 0) return GGQ_OK;
    if (!dst || !src) return GGQ_INVALID;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    ggq_ctx *c = nullptr;
    st = pool_get(dev, &c);
    if (st != GGQ_OK) return st;

    const TypeInfo *ti = type_info(ty);
    const size_t ob = out_bytes(out_ty);
    // 4 Mi elements per chunk: a multiple of every tile size, so each chunk's
    // output offset stays 16-byte aligned
    size_t chunk_blocks = kHostChunkElems / ti->elems;
    const uint8_t *s = (const uint8_t *)src;
    uint8_t *d = (uint8_t *)dst;

    struct Pending { size_t out_off, out_len; } pend[ggq_ctx::SLOTS] = {};
    ggq_status rc = GGQ_OK;
    size_t done_blocks = 0;
    int slot = 0;
    auto drain = [&](int k) -> bool {
        if (!c->busy[k]) return true;
        c->busy[k] = false;
        if (!hip_ok(hipEventSynchronize(c->done[k]), "hipEventSynchronize")) return false;
        copy_pool().copy(d + pend[k].out_off, c->h_out[k], pend[k].out_len);
        return true;
    };
    while (done_blocks < src_blocks && rc == GGQ_OK) {
        size_t nb = std::min(chunk_blocks, src_blocks - done_blocks);
        size_t in_len = nb * ti->bytes, out_len = nb * ti->elems * ob;
        if (!drain(slot)) { rc = GGQ_DEVICE; break; }
        copy_pool().copy(c->h_in[slot], s + done_blocks * ti->bytes, in_len);
        hipStream_t q = c->stream[slot];
        bool ok = hip_ok(hipMemcpyAsync(c->d_in[slot], c->h_in[slot], in_len, hipMemcpyHostToDevice, q), "H2D");
        if (ok) {
            rc = launch_out(ty, out_ty, c->d_out[slot], c->d_in[slot], nb, c->cus, q);
            ok = rc == GGQ_OK;
        }
        ok = ok && hip_ok(hipMemcpyAsync(c->h_out[slot], c->d_out[slot], out_len, hipMemcpyDeviceToHost, q), "D2H");
        ok = ok && hip_ok(hipEventRecord(c->done[slot], q), "hipEventRecord");
        if (!ok) { if (rc == GGQ_OK) rc = GGQ_DEVICE; break; }
        c->busy[slot] = true;
        pend[slot] = {done_blocks * ti->elems * ob, out_len};
        done_blocks += nb;
        slot ^= 1;
    }
    for (int k = 0; k < ggq_ctx::SLOTS; ++k) {
        int kk = (slot + k) % ggq_ctx::SLOTS;   // oldest first
        if (rc == GGQ_OK) { if (!drain(kk)) rc = GGQ_DEVICE; }
        else { (void)hipStreamSynchronize(c->stream[kk]); c->busy[kk] = false; }
    }
    pool_put(c);
    return rc;
}

// ---- quantize -----------------------------------------------------------------------
ggq_status ggq_quantize_device(uint32_t ty, uint32_t in_ty, void *d_dst, size_t dst_blocks, const void *d_src,
                               size_t src_elems, void *hip_stream) {
    ggq_status st = check_quantize(ty, in_ty, dst_blocks, src_elems);
    if (st != GGQ_OK) return st;
    if (dst_blocks == 0) return GGQ_OK;
    if (!d_dst || !d_src) return GGQ_INVALID;
    if ((uintptr_t)d_dst % 16 != 0 || (uintptr_t)d_src % 16 != 0) return GGQ_MISALIGNED;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DevInfo di;
    st = dev_info(dev, di);
    if (st != GGQ_OK) return st;
    return launch_quant_in(ty, in_ty, d_dst, d_src, dst_blocks, di.cus, (hipStream_t)hip_stream);
}

ggq_status ggq_quantize_slice(uint32_t ty, uint32_t in_ty, void *dst, size_t dst_blocks, const void *src,
                              size_t src_elems) {
    ggq_status st = check_quantize(ty, in_ty, dst_blocks, src_elems);
    if (st != GGQ_OK) return st;
    if (dst_blocks == 0) return GGQ_OK;
    if (!dst || !src) return GGQ_INVALID;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    ggq_ctx *c = nullptr;
    st = pool_get(dev, &c);
    if (st != GGQ_OK) return st;
    // The pool's pipelines are sized for decoding (small packed side, large element side);
    // quantizing runs them backwards: elements travel through the large buffers
    // (h_out / d_out), packed blocks come back through the small ones (h_in / d_in).
    const TypeInfo *ti = type_info(ty);
    const size_t ib = out_bytes(in_ty);
    const size_t chunk_blocks = kHostChunkElems / ti->elems;
    const uint8_t *s = (const uint8_t *)src;
    uint8_t *d = (uint8_t *)dst;
    struct Pending { size_t out_off, out_len; } pend[ggq_ctx::SLOTS] = {};
    ggq_status rc = GGQ_OK;
    size_t done_blocks = 0;
    int slot = 0;
    auto drain = [&](int k) -> bool {
        if (!c->busy[k]) return true;
        c->busy[k] = false;
        if (!hip_ok(hipEventSynchronize(c->done[k]), "hipEventSynchronize")) return false;
        copy_pool().copy(d + pend[k].out_off, c->h_in[k], pend[k].out_len);
        return true;
    };
    while (done_blocks < dst_blocks && rc == GGQ_OK) {
        size_t nb = std::min(chunk_blocks, dst_blocks - done_blocks);
        size_t in_len = nb * ti->elems * ib, out_len = nb * ti->bytes;
        if (!drain(slot)) { rc = GGQ_DEVICE; break; }
        copy_pool().copy(c->h_out[slot], s + done_blocks * ti->elems * ib, in_len);
        hipStream_t q = c->stream[slot];
        bool ok = hip_ok(hipMemcpyAsync(c->d_out[slot], c->h_out[slot], in_len, hipMemcpyHostToDevice, q), "H2D");
        if (ok) {
            rc = launch_quant_in(ty, in_ty, c->d_in[slot], c->d_out[slot], nb, c->cus, q);
            ok = rc == GGQ_OK;
        }
        ok = ok && hip_ok(hipMemcpyAsync(c->h_in[slot], c->d_in[slot], out_len, hipMemcpyDeviceToHost, q), "D2H");
        ok = ok && hip_ok(hipEventRecord(c->done[slot], q), "hipEventRecord");
        if (!ok) { if (rc == GGQ_OK) rc = GGQ_DEVICE; break; }
        c->busy[slot] = true;
        pend[slot] = {done_blocks * ti->bytes, out_len};
        done_blocks += nb;
        slot ^= 1;
    }
    for (int k = 0; k < ggq_ctx::SLOTS; ++k) {
        int kk = (slot + k) % ggq_ctx::SLOTS;
        if (rc == GGQ_OK) { if (!drain(kk)) rc = GGQ_DEVICE; }
        else { (void)hipStreamSynchronize(c->stream[kk]); c->busy[kk] = false; }
    }
    pool_put(c);
    return rc;
}

// ---- rearrangement -----------------------------------------------------------------
static ggq_status row_bytes_of(uint32_t ty, uint64_t row_elems, uint64_t *out) {
    const uint64_t be = ggq_gguf_type_elems(ty), bb = ggq_gguf_type_bytes(ty);
    if (be == 0) return GGQ_UNSUPPORTED;
    if (row_elems == 0 || row_elems % be != 0) return GGQ_INDIVISIBLE;
    *out = row_elems / be * bb;
    return GGQ_OK;
}

ggq_status ggq_concat_rows_device(uint32_t ty, uint64_t row_elems, void *d_dst, const void *const *d_parts,
                                  const uint64_t *part_rows, size_t n_parts, void *hip_stream) {
    uint64_t rb = 0;
    ggq_status st = row_bytes_of(ty, row_elems, &rb);
    if (st != GGQ_OK) return st;
    if (!d_dst || !d_parts || !part_rows) return GGQ_INVALID;
    uint8_t *out = (uint8_t *)d_dst;
    for (size_t i = 0; i < n_parts; ++i) {
        const uint64_t n = part_rows[i] * rb;
        if (n == 0) continue;
        if (!d_parts[i]) return GGQ_INVALID;
        HIP_TRY(hipMemcpyAsync(out, d_parts[i], n, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
        out += n;
    }
    return GGQ_OK;
}

ggq_status ggq_split_rows_device(uint32_t ty, uint64_t row_elems, const void *d_src, void *const *d_parts,
                                 const uint64_t *part_rows, size_t n_parts, void *hip_stream) {
    uint64_t rb = 0;
    ggq_status st = row_bytes_of(ty, row_elems, &rb);
    if (st != GGQ_OK) return st;
    if (!d_src || !d_parts || !part_rows) return GGQ_INVALID;
    const uint8_t *in = (const uint8_t *)d_src;
    for (size_t i = 0; i < n_parts; ++i) {
        const uint64_t n = part_rows[i] * rb;
        if (n == 0) continue;
        if (!d_parts[i]) return GGQ_INVALID;
        HIP_TRY(hipMemcpyAsync(d_parts[i], in, n, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
        in += n;
    }
    return GGQ_OK;
}

ggq_status ggq_permute_qk_device(uint32_t ty, uint64_t row_elems, uint64_t rows, uint64_t n_head, void *d_dst,
                                 const void *d_src, void *hip_stream) {
    uint64_t rb = 0;
    ggq_status st = row_bytes_of(ty, row_elems, &rb);
    if (st != GGQ_OK) return st;
    if (n_head == 0 || rows % (2 * n_head) != 0) return GGQ_INDIVISIBLE;
    if (rows == 0) return GGQ_OK;
    if (!d_dst || !d_src || d_dst == d_src) return GGQ_INVALID;
    const uint64_t dh = rows / n_head;
    const uintptr_t a = (uintptr_t)d_dst | (uintptr_t)d_src | (uintptr_t)rb;
    const uint32_t vec = a % 16 == 0 ? 16 : a % 4 == 0 ? 4 : a % 2 == 0 ? 2 : 1;
    const uint64_t pieces = rb / vec * rows;
    const uint64_t per_wg = STREAM_THREADS;
    const uint64_t wgs = (pieces + per_wg - 1) / per_wg;
    hipStream_t q = (hipStream_t)hip_stream;
    uint8_t *o = (uint8_t *)d_dst;
    const uint8_t *i = (const uint8_t *)d_src;
    // 32-bit indices when they cannot wrap (then rows, dh and pieces per row fit as well)
    const bool narrow = pieces <= 0xFFFFFF00ull - STREAM_THREADS && !kWideIndex;
    const FastDiv32 ppr = FastDiv32::make(narrow ? (uint32_t)(rb / vec) : 1u), dhd = FastDiv32::make(narrow ? (uint32_t)dh : 1u);
    for (uint64_t w0 = 0; w0 < wgs; w0 += kMaxGrid) {
        const dim3 grid((uint32_t)std::min(kMaxGrid, wgs - w0)), wg(STREAM_THREADS);
        const uint64_t first = w0 * per_wg;
        if (narrow) {
            const uint32_t total = (uint32_t)pieces, f32 = (uint32_t)first;
            if (vec == 16) hipLaunchKernelGGL((permute_qk_rows<16>), grid, wg, 0, q, o, i, rb, total, ppr, dhd, f32);
            else if (vec == 4) hipLaunchKernelGGL((permute_qk_rows<4>), grid, wg, 0, q, o, i, rb, total, ppr, dhd, f32);
            else if (vec == 2) hipLaunchKernelGGL((permute_qk_rows<2>), grid, wg, 0, q, o, i, rb, total, ppr, dhd, f32);
            else hipLaunchKernelGGL((permute_qk_rows<1>), grid, wg, 0, q, o, i, rb, total, ppr, dhd, f32);
        } else {
            if (vec == 16) hipLaunchKernelGGL((permute_qk_rows_wide<16>), grid, wg, 0, q, o, i, rb, rows, dh, first);
            else if (vec == 4) hipLaunchKernelGGL((permute_qk_rows_wide<4>), grid, wg, 0, q, o, i, rb, rows, dh, first);
            else if (vec == 2) hipLaunchKernelGGL((permute_qk_rows_wide<2>), grid, wg, 0, q, o, i, rb, rows, dh, first);
            else hipLaunchKernelGGL((permute_qk_rows_wide<1>), grid, wg, 0, q, o, i, rb, rows, dh, first);
        }
    }
    HIP_TRY(hipGetLastError());
    return GGQ_OK;
}

// ---- scaling ----------------------------------------------------------------------
ggq_status ggq_scale_device(uint32_t ty, void *d_dst, const void *d_src, size_t elems, double scale, void *hip_stream) {
    if (ty != 0 && ty != 1 && ty != 30 && ty != 28) return GGQ_UNSUPPORTED;       // F32, F16, BF16, F64
    if (elems == 0) return GGQ_OK;
    if (!d_dst || !d_src) return GGQ_INVALID;
    const uint64_t eb = ty == 28 ? 8 : ty == 0 ? 4 : 2;
    if (((uintptr_t)d_dst | (uintptr_t)d_src) % eb != 0) return GGQ_MISALIGNED;
    hipStream_t q = (hipStream_t)hip_stream;
    uint64_t factor = 0;
    if (ty == 0) {
        const float f = (float)scale;
        uint32_t b;
        memcpy(&b, &f, 4);
        factor = b;
        return launch_scale<SC_F32>(d_dst, d_src, elems, factor, q);
    }
    if (ty == 1) return launch_scale<SC_F16>(d_dst, d_src, elems, f32_to_f16_bits((float)scale), q);
    if (ty == 30) return launch_scale<SC_BF16>(d_dst, d_src, elems, f64_to_bf16_bits(scale), q);
    memcpy(&factor, &scale, 8);
    return launch_scale<SC_F64>(d_dst, d_src, elems, factor, q);
}

// ---- batch ------------------------------------------------------------------------
ggq_status ggq_batch_create(const ggq_tensor *tensors, size_t count, uint32_t out_ty, ggq_batch **out) {
    if (!out || (count && !tensors)) return GGQ_INVALID;
    if (out_bytes(out_ty) == 0) return GGQ_UNSUPPORTED;
    for (size_t i = 0; i < count; ++i) {
        if (!type_info(tensors[i].ggml_type)) return GGQ_UNSUPPORTED;
        if (tensors[i].blocks && (!tensors[i].d_src || !tensors[i].d_dst)) return GGQ_INVALID;
        if ((uintptr_t)tensors[i].d_dst % 16 != 0) return GGQ_MISALIGNED;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DevInfo di;
    ggq_status st = dev_info(dev, di);
    if (st != GGQ_OK) return st;

    ggq_batch *b = new (std::nothrow) ggq_batch();
    if (!b) return GGQ_INVALID;
    b->device = dev;
    b->cus = di.cus;
    b->out_ty = out_ty;

    // Route every tensor (the rule is stated at kHybridMinElems).  F16 / BF16 sources are widened one launch each.
    // Block types: adjacent tensors are merged into runs; a run goes to a launch of its own or, as ONE entry, into
    // the tile table of its type.
    std::vector<PlanRun> runs;             // table entries, in the caller's order
    std::vector<uint32_t> types;
    {
        const uint64_t ob = out_bytes(out_ty);
        std::vector<PlanRun> merged;
        for (size_t i = 0; i < count; ++i) {
            const uint32_t ty = tensors[i].ggml_type;
            if (tensors[i].blocks == 0) continue;
            if (ty == GGQ_TYPE_F16 || ty == GGQ_TYPE_BF16) {
                b->plain.push_back({ty, tensors[i].d_src, tensors[i].d_dst, tensors[i].blocks});
                continue;
            }
            const TypeInfo *ti = type_info(ty);
            const uint64_t src = (uint64_t)(uintptr_t)tensors[i].d_src, dst = (uint64_t)(uintptr_t)tensors[i].d_dst;
            // the run of this type that was extended last, if this tensor continues it
            PlanRun *prev = nullptr;
            for (size_t k = merged.size(); k-- > 0;)
                if (merged[k].ty == ty) { prev = &merged[k]; break; }
            if (prev && prev->src + prev->blocks * ti->bytes == src && prev->dst + prev->blocks * ti->elems * ob == dst)
                prev->blocks += tensors[i].blocks;
            else
                merged.push_back({ty, src, dst, tensors[i].blocks});
        }
        for (const PlanRun &r : merged) {
            const uint64_t elems = r.blocks * type_info(r.ty)->elems;
            // a launch of its own: long runs whose schedule has a group start, and a type's ONLY run (a table with
            // one entry would add a record load to every workgroup for nothing: one 16 Mi-element tensor decodes at
            // 0.70-0.72 of the roofline without it, 0.65 with it)
            size_t of_type = 0;
            for (const PlanRun &o : merged) of_type += o.ty == r.ty;
            if (kHybridMinElems != ~0ull &&
                (of_type == 1 || (elems >= kHybridMinElems && (kHybridAny || pick_prefetch(r.ty, out_ty, false, elems).wait != 0)))) {
                b->single.push_back({r.ty, (const void *)(uintptr_t)r.src, (void *)(uintptr_t)r.dst, r.blocks});
                continue;
            }
            runs.push_back(r);
            if (std::find(types.begin(), types.end(), r.ty) == types.end()) types.push_back(r.ty);
        }
    }
    const bool per_xcd = out_ty == GGQ_TYPE_F32 ? RecordOrder<OutF32>::PER_XCD : RecordOrder<OutF16>::PER_XCD;
    st = build_plan_table(b, runs, types, per_xcd, [&](uint32_t ty, uint64_t &tb, uint64_t &tin, uint64_t &tout) {
        const HostGeom hg = host_geom(type_info(ty), out_ty);
        tb = hg.blocks;
        tin = hg.bytes;
        tout = hg.elems * out_bytes(out_ty);
    });
    if (st != GGQ_OK) {
        ggq_batch_destroy(b);
        return st;
    }
    *out = b;
    return GGQ_OK;
}

ggq_status ggq_quant_batch_create(const ggq_tensor *tensors, size_t count, uint32_t in_ty, ggq_batch **out) {
    if (!out || (count && !tensors)) return GGQ_INVALID;
    const uint64_t ib = out_bytes(in_ty);
    if (ib == 0) return GGQ_UNSUPPORTED;
    for (size_t i = 0; i < count; ++i) {
        if (!type_info(tensors[i].ggml_type) || !can_quantize(tensors[i].ggml_type)) return GGQ_UNSUPPORTED;
        if (tensors[i].blocks && (!tensors[i].d_src || !tensors[i].d_dst)) return GGQ_INVALID;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DevInfo di;
    ggq_status st = dev_info(dev, di);
    if (st != GGQ_OK) return st;
    // Tensors of one target type that are adjacent in memory (elements and packed blocks) are one run; F16 / BF16
    // targets are narrowed one launch per run, block types share one launch per type.
    std::vector<PlanRun> merged, runs;
    std::vector<uint32_t> types;
    for (size_t i = 0; i < count; ++i) {
        const uint32_t ty = tensors[i].ggml_type;
        if (tensors[i].blocks == 0) continue;
        const TypeInfo *ti = type_info(ty);
        const uint64_t src = (uint64_t)(uintptr_t)tensors[i].d_src, dst = (uint64_t)(uintptr_t)tensors[i].d_dst;
        PlanRun *prev = nullptr;
        for (size_t k = merged.size(); k-- > 0;)
            if (merged[k].ty == ty) { prev = &merged[k]; break; }
        if (prev && prev->src + prev->blocks * ti->elems * ib == src && prev->dst + prev->blocks * ti->bytes == dst)
            prev->blocks += tensors[i].blocks;
        else
            merged.push_back({ty, src, dst, tensors[i].blocks});
    }
    // alignment is a property of a run's start: the tiles of a run are cut from there
    for (const PlanRun &r : merged)
        if (r.src % 16 != 0 || r.dst % 16 != 0) return GGQ_MISALIGNED;
    ggq_batch *b = new (std::nothrow) ggq_batch();
    if (!b) return GGQ_INVALID;
    b->device = dev;
    b->cus = di.cus;
    b->out_ty = in_ty;
    b->quantize = true;
    for (const PlanRun &r : merged) {
        // F16 / BF16 targets, and runs long enough for a launch of their own: through the table a workgroup starts
        // with a load of its record, which costs the short-lived workgroups of the quantizers 1-13 % (most for the
        // encoders with one load per thread: Q4_0 from f32 5666 against 6542 GB/s), a launch costs ~3 us
        // (and, as in the decode plans, a type's only run needs no table)
        size_t of_type = 0;
        for (const PlanRun &o : merged) of_type += o.ty == r.ty;
        if (r.ty == GGQ_TYPE_F16 || r.ty == GGQ_TYPE_BF16 ||
            (kHybridMinElems != ~0ull && (of_type == 1 || r.blocks * type_info(r.ty)->elems >= kHybridMinElems))) {
            b->plain.push_back({r.ty, (const void *)(uintptr_t)r.src, (void *)(uintptr_t)r.dst, r.blocks});
            continue;
        }
        runs.push_back(r);
        if (std::find(types.begin(), types.end(), r.ty) == types.end()) types.push_back(r.ty);
    }
    st = build_plan_table(b, runs, types, false, [&](uint32_t ty, uint64_t &tb, uint64_t &tin, uint64_t &tout) {
        HostGeom g{};
        (void)quant_geom_in(ty, in_ty, g);
        tb = g.blocks;
        tin = g.elems * ib;
        tout = g.bytes;
    });
    if (st != GGQ_OK) {
        ggq_batch_destroy(b);
        return st;
    }
    *out = b;
    return GGQ_OK;
}

ggq_status ggq_batch_run(ggq_batch *b, void *hip_stream) {
    if (!b) return GGQ_INVALID;
    // the plan's table and tensors live on the device that was current at ggq_batch_create
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != b->device) {
        g_err = "ggq_batch_run: the current device is not the one the batch was created on";
        return GGQ_INVALID;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    if (b->quantize) {
        for (const auto &g : b->groups) {
            ggq_status rc;
            const TileRef *tab = b->d_table + g.first;
            switch (b->out_ty) {
            case GGQ_TYPE_F32: rc = launch_quant_batch<InF32>(g.ty, tab, g.ntiles, st); break;
            case GGQ_TYPE_F16: rc = launch_quant_batch<InF16>(g.ty, tab, g.ntiles, st); break;
            default: rc = launch_quant_batch<InBF16>(g.ty, tab, g.ntiles, st); break;
            }
            if (rc != GGQ_OK) return rc;
        }
        for (const auto &p : b->plain) {      // F16 / BF16 targets: p.src = elements, p.dst = narrowed elements
            ggq_status rc = launch_quant_in(p.ty, b->out_ty, p.dst, p.src, p.n, b->cus, st);
            if (rc != GGQ_OK) return rc;
        }
        return GGQ_OK;
    }
    for (const auto &g : b->groups) {
        ggq_status rc;
        const TileRef *tab = b->d_table + g.first;
        switch (b->out_ty) {
        case GGQ_TYPE_F32: rc = launch_batch<OutF32>(g.ty, tab, g.ntiles, st); break;
        case GGQ_TYPE_F16: rc = launch_batch<OutF16>(g.ty, tab, g.ntiles, st); break;
        default: rc = launch_batch<OutBF16>(g.ty, tab, g.ntiles, st); break;
        }
        if (rc != GGQ_OK) return rc;
    }
    for (const auto &p : b->single) {
        ggq_status rc = launch_out(p.ty, b->out_ty, p.dst, p.src, p.n, b->cus, st);
        if (rc != GGQ_OK) return rc;
    }
    for (const auto &p : b->plain) {
        ggq_status rc = launch_out(p.ty, b->out_ty, p.dst, p.src, p.n, b->cus, st);
        if (rc != GGQ_OK) return rc;
    }
    return GGQ_OK;
}

size_t ggq_batch_table_launches(const ggq_batch *b) { return b ? b->groups.size() : 0; }

size_t ggq_batch_launches(const ggq_batch *b) {
    return b ? b->groups.size() + b->single.size() + b->plain.size() : 0;
}

void ggq_batch_destroy(ggq_batch *b) {
    if (!b) return;
    if (b->d_table) (void)hipFree(b->d_table);
    delete b;
}

// ---- streamed host -> device -------------------------------------------------------
ggq_status ggq_ctx_create(int device, size_t staging_bytes, ggq_ctx **out) {
    if (!out) return GGQ_INVALID;
    int n = ggq_device_count();
    if (device < 0 || device >= n) { g_err = "no such device"; return GGQ_DEVICE; }
    if (staging_bytes < (1u << 20)) staging_bytes = 1u << 20;
    size_t per_slot = (staging_bytes / ggq_ctx::SLOTS) & ~(size_t)4095;
    return ctx_make(device, per_slot, 0, out);
}

ggq_status ggq_dequantize_async(ggq_ctx *c, uint32_t ty, uint32_t out_ty, void *d_dst, size_t dst_elems,
                                const void *h_src, size_t src_blocks) {
    if (!c) return GGQ_INVALID;
    ggq_status st = check_slice(ty, out_ty, dst_elems, src_blocks);
    if (st != GGQ_OK) return st;
    if (src_blocks == 0) return GGQ_OK;
    if (!d_dst || !h_src) return GGQ_INVALID;
    if ((uintptr_t)d_dst % 16 != 0) return GGQ_MISALIGNED;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(c->device));

    const TypeInfo *ti = type_info(ty);
    const size_t ob = out_bytes(out_ty);
    // chunks are whole tiles of either output width (8192 elements), so every
    // chunk's output offset stays 16-byte aligned
    size_t tile_blk = 8192 / ti->elems;
    size_t chunk_blocks = c->in_cap / ti->bytes / tile_blk * tile_blk;
    if (chunk_blocks == 0) { (void)hipSetDevice(prev); return GGQ_INVALID; }
    const uint8_t *s = (const uint8_t *)h_src;
    uint8_t *d = (uint8_t *)d_dst;
    ggq_status rc = GGQ_OK;
    bool src_pinned = false;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h_src) == hipSuccess) src_pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();      // ordinary pageable memory is "not found", not an error
    }
    int slot = c->next;
    for (size_t done = 0; done < src_blocks && rc == GGQ_OK;) {
        size_t nb = std::min(chunk_blocks, src_blocks - done);
        if (c->busy[slot]) {
            if (!hip_ok(hipEventSynchronize(c->done[slot]), "hipEventSynchronize")) { rc = GGQ_DEVICE; break; }
            c->busy[slot] = false;
        }
        // page-locked sources go to the device straight from where they are; pageable ones
        // are first copied into the slot's pinned buffer
        const void *from = s + done * ti->bytes;
        if (!src_pinned) {
            copy_pool().copy(c->h_in[slot], from, nb * ti->bytes);
            from = c->h_in[slot];
        }
        hipStream_t q = c->stream[slot];
        if (!hip_ok(hipMemcpyAsync(c->d_in[slot], from, nb * ti->bytes, hipMemcpyHostToDevice, q), "H2D")) {
            rc = GGQ_DEVICE;
            break;
        }
        rc = launch_out(ty, out_ty, d + done * ti->elems * ob, c->d_in[slot], nb, c->cus, q);
        if (rc != GGQ_OK) break;
        if (!hip_ok(hipEventRecord(c->done[slot], q), "hipEventRecord")) { rc = GGQ_DEVICE; break; }
        c->busy[slot] = true;
        done += nb;
        slot ^= 1;
    }
    c->next = slot;
    (void)hipSetDevice(prev);
    return rc;
}

ggq_status ggq_upload_async(ggq_ctx *c, void *d_dst, const void *h_src, size_t bytes) {
    if (!c) return GGQ_INVALID;
    if (bytes == 0) return GGQ_OK;
    if (!d_dst || !h_src) return GGQ_INVALID;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(c->device));
    const uint8_t *s = (const uint8_t *)h_src;
    uint8_t *d = (uint8_t *)d_dst;
    bool src_pinned = false;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h_src) == hipSuccess) src_pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    ggq_status rc = GGQ_OK;
    int slot = c->next;
    for (size_t done = 0; done < bytes && rc == GGQ_OK;) {
        const size_t n = std::min(c->in_cap, bytes - done);
        hipStream_t q = c->stream[slot];
        const void *from = s + done;
        if (!src_pinned) {
            if (c->busy[slot]) {
                if (!hip_ok(hipEventSynchronize(c->done[slot]), "hipEventSynchronize")) { rc = GGQ_DEVICE; break; }
                c->busy[slot] = false;
            }
            copy_pool().copy(c->h_in[slot], from, n);
            from = c->h_in[slot];
        }
        if (!hip_ok(hipMemcpyAsync(d + done, from, n, hipMemcpyHostToDevice, q), "H2D")) { rc = GGQ_DEVICE; break; }
        if (!src_pinned) {
            if (!hip_ok(hipEventRecord(c->done[slot], q), "hipEventRecord")) { rc = GGQ_DEVICE; break; }
            c->busy[slot] = true;
        }
        done += n;
        slot ^= 1;
    }
    c->next = slot;
    (void)hipSetDevice(prev);
    return rc;
}

// ---- streamed host -> device, no allocation -----------------------------------------
ggq_status ggq_dequantize_noalloc(uint32_t ty, uint32_t out_ty, void *d_dst, size_t dst_elems, const void *h_src,
                                  size_t src_blocks, void *hip_stream) {
    ggq_status st = check_slice(ty, out_ty, dst_elems, src_blocks);
    if (st != GGQ_OK) return st;
    // the one-time set-up of the current device happens on every valid call, also an empty one, so that a
    // caller can get it out of the way before it starts capturing a graph
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    NoallocDev nd;
    st = noalloc_state(dev, nd);
    if (st != GGQ_OK) return st;
    if (src_blocks == 0) return GGQ_OK;
    if (!d_dst || !h_src) return GGQ_INVALID;
    if ((uintptr_t)d_dst % 16 != 0) return GGQ_MISALIGNED;
    const TypeInfo *ti = type_info(ty);
    if (src_blocks > (size_t)GGQ_NOALLOC_MAX_BYTES / ti->bytes) {
        g_err = "ggq_dequantize_noalloc: packed input exceeds GGQ_NOALLOC_MAX_BYTES";
        return GGQ_TOO_LARGE;
    }
    // Only page-locked memory goes to the device without the runtime staging (and possibly
    // allocating) on the way.
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, h_src) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        g_err = "ggq_dequantize_noalloc: h_src is not page-locked host memory";
        return GGQ_INVALID;
    }
    hipStream_t q = (hipStream_t)hip_stream;
    HIP_TRY(hipMemcpyAsync(nd.ptr, h_src, src_blocks * ti->bytes, hipMemcpyHostToDevice, q));
    return launch_out(ty, out_ty, d_dst, nd.ptr, src_blocks, nd.cus, q);
}

ggq_status ggq_selftest_fast_division(uint64_t mismatches[5], uint64_t *tested) {
    if (!mismatches) return GGQ_INVALID;
    unsigned long long *d = nullptr, h[6] = {0, 0, 0, 0, 0, 0};
    if (!hip_ok(hipMalloc((void **)&d, sizeof(h)), "hipMalloc")) return GGQ_DEVICE;
    bool ok = hip_ok(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice), "hipMemcpy");
    if (ok) {
        hipLaunchKernelGGL(fast_division_selftest, dim3(16384), dim3(256), 0, nullptr, d);
        ok = hip_ok(hipGetLastError(), "fast_division_selftest") &&
             hip_ok(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost), "hipMemcpy");
    }
    (void)hipFree(d);
    if (!ok) return GGQ_DEVICE;
    for (int k = 0; k < 5; ++k) mismatches[k] = h[k];
    if (tested) *tested = h[5];
    return GGQ_OK;
}

ggq_status ggq_ctx_fence(ggq_ctx *c, void *hip_stream) {
    if (!c) return GGQ_INVALID;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(c->device));
    ggq_status rc = GGQ_OK;
    for (int s = 0; s < ggq_ctx::SLOTS && rc == GGQ_OK; ++s) {
        if (!hip_ok(hipEventRecord(c->fence[s], c->stream[s]), "hipEventRecord") ||
            !hip_ok(hipStreamWaitEvent((hipStream_t)hip_stream, c->fence[s], 0), "hipStreamWaitEvent"))
            rc = GGQ_DEVICE;
    }
    (void)hipSetDevice(prev);
    return rc;
}

ggq_status ggq_ctx_wait_stream(ggq_ctx *c, void *hip_stream) {
    if (!c) return GGQ_INVALID;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(c->device));
    ggq_status rc = GGQ_OK;
    if (!hip_ok(hipEventRecord(c->gate, (hipStream_t)hip_stream), "hipEventRecord")) rc = GGQ_DEVICE;
    for (int s = 0; s < ggq_ctx::SLOTS && rc == GGQ_OK; ++s)
        if (!hip_ok(hipStreamWaitEvent(c->stream[s], c->gate, 0), "hipStreamWaitEvent")) rc = GGQ_DEVICE;
    (void)hipSetDevice(prev);
    return rc;
}

ggq_status ggq_sync(ggq_ctx *c) {
    if (!c) return GGQ_INVALID;
    for (int s = 0; s < ggq_ctx::SLOTS; ++s) {
        HIP_TRY(hipStreamSynchronize(c->stream[s]));
        c->busy[s] = false;
    }
    return GGQ_OK;
}

void ggq_ctx_destroy(ggq_ctx *c) {
    if (!c) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(c->device);
    ctx_free(c);
    (void)hipSetDevice(prev);
}

}  // extern "C"
