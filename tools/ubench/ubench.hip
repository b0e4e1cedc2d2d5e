// ubench.hip -- in-process A/B of the library's kernels on one MI355X (measurement code, not part of the product).
//
//   ./ubench [Mi elements per launch = 1024] [rounds = 15] [families and name filters, comma separated]
//
// Families (the first word of a variant's name; several may be given, and any other word filters by substring):
//   base    fill / copy / read streams (the ceilings the kernels are held against)
//   ship    every decode format x {f32, f16}: the single-tensor and the table kernel at their shipped schedules
//   sw, swb period x lead sweep of the single-tensor / table kernel   (needs a filter such as "sw Q4K_f16")
//   gw      group-start sweep of the single-tensor kernel             (needs a filter such as "gw Q6K_f16")
//   sdeq    workgroup shape of the decode kernel (threads x stores per thread), Q4_K and Q4_0
//   sgw     the same shapes with a group start of 4 ... 128 tiles in place of the shipped schedule (Q4_K, Q6_K)
//   half    F16 -> f32 as 64-byte blocks through the tile kernel under several schedules (the library uses widen_vec)
//   wsw     widen f16 -> f32 with 1, 2, 4 steps per thread and group starts of 4 ... 64 pieces
//   quant   every encoder x input type: one tensor per launch ("single", shipped group start), through a tile table
//           ("table"), and with group starts of 1 ... 128 tiles ("qgw")
//   stream  widen, narrow, the permute_qk row gather, scale
// Environment: UBENCH_REP = launches per sample (default 3; a long run of one kernel behind itself can behave
// differently from three launches behind some other variant); UBENCH_COLD=1 = a 2 GiB read with plain (cache-
// allocating) loads before every sample, so that every sample finds its input in HBM, not in a cache.
//
// Safety.  Every device buffer is allocated through ub_alloc(), which records its capacity.  Every variant
// declares its footprint -- for each pointer it hands to a kernel, the first byte and the number of bytes the kernel
// can touch from there, computed from the same geometry constants the kernel uses (tile sizes, the 16-byte grid of
// the tile loads, the 128-byte lines of the prefetch touches, table slots).  admit() checks every footprint against
// the recorded capacities before anything runs; a variant with no footprint, or with one that does not fit, is
// REFUSED and never launched.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "kernels.h"
#include "quantize.h"
#include "rearrange.h"
#include "scale.h"
#include "schedule.h"
#include "shapes.h"

using namespace ggq;

#define CK(x)                                                                                                         \
    do {                                                                                                              \
        hipError_t e_ = (x);                                                                                          \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } \
    } while (0)

// ---- buffers and footprints -----------------------------------------------------------------------------
struct Buf { const uint8_t *p; uint64_t cap; const char *name; };
static std::vector<Buf> g_bufs;
template <class T>
static T *ub_alloc(uint64_t bytes, const char *name) {
    void *p = nullptr;
    CK(hipMalloc(&p, bytes));
    g_bufs.push_back({(const uint8_t *)p, bytes, name});
    return (T *)p;
}
struct Foot { const void *p; uint64_t bytes; const char *what; };
struct Variant {
    std::string name;
    double bytes;                                   // algorithmic bytes per launch
    std::function<void(hipStream_t)> run;
    std::vector<Foot> foot;
    std::vector<float> ms;
};
static bool admit(const Variant &v) {
    if (v.foot.empty()) {
        std::printf("REFUSED %s: no footprint declared\n", v.name.c_str());
        return false;
    }
    for (const Foot &f : v.foot) {
        const uint8_t *p = (const uint8_t *)f.p;
        bool ok = false;
        for (const Buf &b : g_bufs)
            if (p >= b.p && p <= b.p + b.cap && f.bytes <= (uint64_t)(b.p + b.cap - p)) { ok = true; break; }
        if (!ok) {
            std::printf("REFUSED %s: %s touches %llu bytes from %p, which no buffer holds\n", v.name.c_str(), f.what,
                        (unsigned long long)f.bytes, f.p);
            return false;
        }
    }
    return true;
}
static bool fits(const Foot &f) {
    Variant v{"(harness)", 0, nullptr, {f}, {}};
    return admit(v);
}
static uint64_t up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// What a tile decode of `blocks` blocks of BYTES bytes at `src` can read: the tile loads use the enclosing 16-byte
// grid, the prefetch touches the enclosing 128-byte lines of whole tiles inside the tensor.
static Foot packed_foot(const uint8_t *src, uint64_t blocks, uint32_t block_bytes) {
    const uintptr_t first = (uintptr_t)src & ~(uintptr_t)127, last = up((uintptr_t)src + blocks * block_bytes, 16);
    return {(const void *)first, (uint64_t)(last - first), "packed input"};
}

enum { F32_ID = 0, F16_ID = 1 };
template <class O> constexpr uint32_t out_ty() { return sizeof(typename O::T) == 4 ? (uint32_t)F32_ID : (uint32_t)F16_ID; }
template <class O> constexpr const char *out_name() { return sizeof(typename O::T) == 4 ? "f32" : "f16"; }

#define UB_FORMATS(X)                                                                                   \
    X(Q4_0, GGQ_TYPE_Q4_0) X(Q4_1, GGQ_TYPE_Q4_1) X(Q5_0, GGQ_TYPE_Q5_0) X(Q5_1, GGQ_TYPE_Q5_1) X(Q8_0, GGQ_TYPE_Q8_0) \
    X(Q8_1, GGQ_TYPE_Q8_1) X(Q2K, GGQ_TYPE_Q2K) X(Q3K, GGQ_TYPE_Q3K) X(Q4K, GGQ_TYPE_Q4K) X(Q5K, GGQ_TYPE_Q5K)         \
    X(Q6K, GGQ_TYPE_Q6K) X(Q8K, GGQ_TYPE_Q8K) X(IQ4NL, GGQ_TYPE_IQ4NL) X(IQ4XS, GGQ_TYPE_IQ4XS)

struct Ctx {
    uint64_t elems;
    uint8_t *d_in;      // packed blocks / 16-bit elements
    uint8_t *d_out;     // decoded elements (4 bytes each at most) / f32 source of the quantizers
    uint8_t *d_q;       // packed output of the quantizers / 16-bit outputs
    std::vector<Variant> *vs;
};

// ---- decode: single-tensor kernel ------------------------------------------------------------------------
template <class F, class O>
static void add_one(const Ctx &c, const std::string &label, Prefetch pf) {
    typedef TileGeom<F, O> G;
    const uint64_t blocks = c.elems / F::N;
    const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);
    const double bytes = (double)c.elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    const uint8_t *src = c.d_in;
    typename O::T *dst = (typename O::T *)c.d_out;
    c.vs->push_back({label, bytes,
                     [=](hipStream_t s) { hipLaunchKernelGGL((deq_one<F, O, true>), dim3(grid), dim3(WG_THREADS), 0, s, src, dst, blocks, pf); },
                     {packed_foot(src, blocks, F::BYTES), {dst, blocks * F::N * sizeof(typename O::T), "decoded output"}}, {}});
}

// ---- decode: table kernel over one tensor -------------------------------------------------------------------
template <class F, class O>
struct OneTable {
    TileRef *d_tab = nullptr;
    uint64_t ntiles = 0, slots = 0;
    void build(const Ctx &c) {
        typedef TileGeom<F, O> G;
        if (d_tab) return;
        const uint64_t blocks = c.elems / F::N;
        ntiles = (blocks + G::TB - 1) / G::TB;
        slots = up(ntiles, 64);
        std::vector<TileRef> tab(slots, TileRef{0, 0});
        for (uint64_t t = 0; t < ntiles; ++t) {
            const uint64_t nblk = std::min<uint64_t>(G::TB, blocks - t * G::TB);
            tab[record_slot<RecordOrder<O>::PER_XCD>(t)] = {((uint64_t)(uintptr_t)c.d_in + t * G::TBYTES) | (nblk << 48),
                                                            (uint64_t)(uintptr_t)c.d_out + t * G::ELEMS * sizeof(typename O::T)};
        }
        d_tab = ub_alloc<TileRef>(slots * sizeof(TileRef), "tile table");
        CK(hipMemcpy(d_tab, tab.data(), slots * sizeof(TileRef), hipMemcpyHostToDevice));
    }
};
template <class F, class O>
static void add_batch(const Ctx &c, const std::string &label, Prefetch pf) {
    static OneTable<F, O> tab;
    tab.build(c);
    const uint64_t blocks = c.elems / F::N, ntiles = tab.ntiles;
    const double bytes = (double)c.elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    const TileRef *d_tab = tab.d_tab;
    c.vs->push_back({label, bytes,
                     [=](hipStream_t s) { hipLaunchKernelGGL((deq_batch<F, O, true>), dim3((unsigned)ntiles), dim3(WG_THREADS), 0, s, d_tab, (uint64_t)0, ntiles, pf); },
                     {packed_foot(c.d_in, blocks, F::BYTES), {c.d_out, blocks * F::N * sizeof(typename O::T), "decoded output"},
                      {d_tab, tab.slots * sizeof(TileRef), "tile table"}}, {}});
}

template <class F, class O>
static void add_ship(const Ctx &c, const char *fname, uint32_t ty) {
    const std::string tag = std::string(fname) + "_" + out_name<O>();
    add_one<F, O>(c, "ship " + tag + " single", prefetch_for(ty, out_ty<O>(), false, c.elems));
    add_batch<F, O>(c, "ship " + tag + " table", prefetch_for(ty, out_ty<O>(), true));
}
template <class F, class O>
static void add_sweeps(const Ctx &c, const char *fname, bool sw, bool swb, bool gw) {
    const std::string tag = std::string(fname) + "_" + out_name<O>();
    static const uint32_t periods[] = {0, 1, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048};
    static const uint32_t leads[] = {512, 768, 1024, 1280, 1536, 2048, 3072};
    char label[112];
    if (sw || swb)
        for (uint32_t P : periods)
            for (uint32_t L : leads) {
                if (P == 0 && L != leads[0]) continue;
                if (P > 1 && L < P) continue;
                std::snprintf(label, sizeof label, " %s P%u L%u", tag.c_str(), P, P ? L : 0);
                if (sw) add_one<F, O>(c, std::string("sw") + label, Prefetch{P, L});
                if (swb) add_batch<F, O>(c, std::string("swb") + label, Prefetch{P, L});
            }
    if (gw)
        for (uint32_t w : {0u, 2u, 3u, 4u, 5u, 6u})
            for (uint32_t l : {0u, 512u, 1280u, 2048u}) {
                std::snprintf(label, sizeof label, "gw %s W%u %s%u", tag.c_str(), w ? 1u << w : 0u, l ? "P1 L" : "off ", l);
                add_one<F, O>(c, label, l ? Prefetch{1, l, w} : Prefetch{0, 0, w});
            }
}

// ---- decode: workgroup shape -----------------------------------------------------------------------------
template <class F, class O, int THR, int S>
static void add_shape(const Ctx &c, const char *fname, uint32_t ty) {
    typedef SGeom<F, O, THR, S> G;
    typedef TileGeom<F, O> G0;
    const uint64_t blocks = c.elems / F::N;
    const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);
    const double bytes = (double)c.elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    // the shipped schedule, kept at the same BYTE distances whatever the tile size
    const Prefetch p0 = prefetch_for(ty, out_ty<O>(), false, c.elems);
    const Prefetch pf{p0.period > 1 ? (uint32_t)((uint64_t)p0.period * G0::TBYTES / G::TBYTES) : p0.period,
                      (uint32_t)((uint64_t)p0.lead * G0::TBYTES / G::TBYTES), p0.wait};
    char label[112];
    std::snprintf(label, sizeof label, "sdeq %s_%s thr%d S%d%s", fname, out_name<O>(), THR, S, THR == 512 && S == 2 ? " (=ship)" : "");
    const uint8_t *src = c.d_in;
    typename O::T *dst = (typename O::T *)c.d_out;
    c.vs->push_back({label, bytes,
                     [=](hipStream_t s) { hipLaunchKernelGGL((deq_one_s<F, O, THR, S>), dim3(grid), dim3(THR), 0, s, src, dst, blocks, pf); },
                     {packed_foot(src, blocks, F::BYTES), {dst, blocks * F::N * sizeof(typename O::T), "decoded output"}}, {}});
}
// the same shapes with a group start in place of the shipped schedule ("sgw"): 2^w tiles, no burst
template <class F, class O, int THR, int S>
static void add_shape_gw(const Ctx &c, const char *fname) {
    typedef SGeom<F, O, THR, S> G;
    const uint64_t blocks = c.elems / F::N;
    const unsigned grid = (unsigned)((blocks + G::TB - 1) / G::TB);
    const double bytes = (double)c.elems * ((double)F::BYTES / F::N + sizeof(typename O::T));
    const uint8_t *src = c.d_in;
    typename O::T *dst = (typename O::T *)c.d_out;
    for (uint32_t w : {2u, 3u, 4u, 5u, 6u, 7u}) {
        char label[112];
        std::snprintf(label, sizeof label, "sgw %s_%s thr%d S%d W%u", fname, out_name<O>(), THR, S, 1u << w);
        const Prefetch pf{0, 0, w};
        c.vs->push_back({label, bytes,
                         [=](hipStream_t s) { hipLaunchKernelGGL((deq_one_s<F, O, THR, S>), dim3(grid), dim3(THR), 0, s, src, dst, blocks, pf); },
                         {packed_foot(src, blocks, F::BYTES), {dst, blocks * F::N * sizeof(typename O::T), "decoded output"}}, {}});
    }
}
template <class F, class O>
static void add_shapes_gw(const Ctx &c, const char *fname) {
    add_shape_gw<F, O, 512, 2>(c, fname);
    add_shape_gw<F, O, 256, 2>(c, fname);
    add_shape_gw<F, O, 128, 2>(c, fname);
    add_shape_gw<F, O, 256, 1>(c, fname);
    add_shape_gw<F, O, 256, 4>(c, fname);
    add_shape_gw<F, O, 128, 4>(c, fname);
}
template <class F, class O>
static void add_shapes(const Ctx &c, const char *fname, uint32_t ty) {
    add_shape<F, O, 512, 2>(c, fname, ty);
    add_shape<F, O, 256, 2>(c, fname, ty);
    add_shape<F, O, 128, 2>(c, fname, ty);
    add_shape<F, O, 512, 1>(c, fname, ty);
    add_shape<F, O, 256, 1>(c, fname, ty);
    add_shape<F, O, 1024, 1>(c, fname, ty);
    add_shape<F, O, 256, 4>(c, fname, ty);
}

// ---- quantizers ----------------------------------------------------------------------------------------------
template <class Q, class In>
static void add_quant(const Ctx &c, const char *qname, const char *iname) {
    typedef QuantGeom<Q, In> G;
    typedef typename In::T TI;
    const uint64_t blocks = c.elems / Q::N;
    const uint64_t ntiles = (blocks + G::TB - 1) / G::TB;
    const double bytes = (double)c.elems * (sizeof(TI) + (double)Q::BYTES / Q::N);
    const TI *src = sizeof(TI) == 4 ? (const TI *)c.d_out : (const TI *)c.d_in;       // f32 values / f16 bit patterns
    uint8_t *dst = c.d_q;
    const std::vector<Foot> foot = {{src, blocks * Q::N * sizeof(TI), "elements"}, {dst, blocks * Q::BYTES, "packed output"}};
    c.vs->push_back({std::string("quant ") + qname + " from " + iname + " single", bytes,
                     [=](hipStream_t s) { hipLaunchKernelGGL((quant_one<Q, In>), dim3((unsigned)ntiles), dim3(G::THR), 0, s, src, dst, blocks, QuantGroup<Q>::LOG2); },
                     foot, {}});
    c.vs->push_back({std::string("quant ") + qname + " from " + iname + " qgw 1 (no group start)", bytes,
                     [=](hipStream_t s) { hipLaunchKernelGGL((quant_one<Q, In>), dim3((unsigned)ntiles), dim3(G::THR), 0, s, src, dst, blocks, 0u); },
                     foot, {}});
    for (uint32_t w = 1; w <= 7; ++w)       // group start over 2^w tiles (selected with the word "qgw")
        c.vs->push_back({std::string("quant ") + qname + " from " + iname + " qgw " + std::to_string(1u << w), bytes,
                         [=](hipStream_t s) { hipLaunchKernelGGL((quant_one<Q, In>), dim3((unsigned)ntiles), dim3(G::THR), 0, s, src, dst, blocks, w); },
                         foot, {}});
    std::vector<TileRef> tab(ntiles);
    for (uint64_t t = 0; t < ntiles; ++t) {
        const uint64_t nblk = std::min<uint64_t>(G::TB, blocks - t * G::TB);
        tab[t] = {((uint64_t)(uintptr_t)src + t * G::ELEMS * sizeof(TI)) | (nblk << 48), (uint64_t)(uintptr_t)dst + t * G::TBYTES};
    }
    TileRef *d_tab = ub_alloc<TileRef>(ntiles * sizeof(TileRef), "quantize tile table");
    CK(hipMemcpy(d_tab, tab.data(), ntiles * sizeof(TileRef), hipMemcpyHostToDevice));
    std::vector<Foot> foot2 = foot;
    foot2.push_back({d_tab, ntiles * sizeof(TileRef), "tile table"});
    c.vs->push_back({std::string("quant ") + qname + " from " + iname + " table", bytes,
                     [=](hipStream_t s) { hipLaunchKernelGGL((quant_batch<Q, In>), dim3((unsigned)ntiles), dim3(G::THR), 0, s, d_tab, (uint64_t)0); },
                     foot2, {}});
}

int main(int argc, char **argv) {
    const uint64_t melems = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1024;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 15;
    const std::string sel = argc > 3 ? argv[3] : "base,ship";
    const uint64_t elems = melems << 20;          // decoded elements per launch
    if (melems == 0 || melems > 8192) { std::printf("elements out of range\n"); return 2; }
    std::printf("%llu Mi elements per launch, %d rounds, selection \"%s\"\n", (unsigned long long)melems, rounds, sel.c_str());

    Ctx c{elems, nullptr, nullptr, nullptr, nullptr};
    // d_in: packed blocks (<= 290 / 256 bytes per element) or 16-bit elements, plus the slack the 16-byte tile loads
    // and 128-byte prefetch lines may reach; d_out: 4 bytes per element; d_q: packed blocks or 16-bit elements
    c.d_in = ub_alloc<uint8_t>(elems * 2 + 4096, "d_in");
    c.d_out = ub_alloc<uint8_t>(elems * 4, "d_out");
    c.d_q = ub_alloc<uint8_t>(elems * 2 + 4096, "d_q");
    {   // packed input: cheap non-constant bytes whose 16-bit fields are finite, non-negative f16 values
        std::vector<uint32_t> h(1 << 22);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (auto &w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16) & 0x3BFF3BFFu; }
        for (uint64_t off = 0; off < elems * 2; off += h.size() * 4)
            CK(hipMemcpy(c.d_in + off, h.data(), std::min<uint64_t>(h.size() * 4, elems * 2 - off), hipMemcpyHostToDevice));
        // f32 source of the quantizers: uniform values in (-1, 1)
        std::vector<float> f(1 << 22);
        for (auto &x : f) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (float)(int32_t)(uint32_t)(s >> 16) * (1.0f / 2147483648.0f); }
        for (uint64_t off = 0; off < elems * 4; off += f.size() * 4)
            CK(hipMemcpy(c.d_out + off, f.data(), std::min<uint64_t>(f.size() * 4, elems * 4 - off), hipMemcpyHostToDevice));
    }
    std::vector<Variant> vs;
    c.vs = &vs;
    auto has = [&](const char *w) {               // a selection word, whole (between commas / spaces)
        size_t at = 0;
        const std::string word = w;
        while ((at = sel.find(word, at)) != std::string::npos) {
            const bool l = at == 0 || sel[at - 1] == ',' || sel[at - 1] == ' ';
            const size_t e = at + word.size();
            const bool r = e == sel.size() || sel[e] == ',' || sel[e] == ' ';
            if (l && r) return true;
            at = e;
        }
        return false;
    };
    const uint64_t n16 = elems / 4;
    if (has("base")) {
        uint4v *o = (uint4v *)c.d_out;
        const uint4v *i = (const uint4v *)c.d_in;
        vs.push_back({"base fill 256 x 1", (double)elems * 4,
                      [=](hipStream_t s) { hipLaunchKernelGGL((k_fill<256>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, o, n16); },
                      {{o, n16 * 16, "fill"}}, {}});
        vs.push_back({"base copy 256 x 1 (half size)", (double)elems * 4,
                      [=](hipStream_t s) { hipLaunchKernelGGL((k_copy<256>), dim3((unsigned)((n16 / 2 + 255) / 256)), dim3(256), 0, s, o, i, n16 / 2); },
                      {{o, n16 / 2 * 16, "copy out"}, {i, n16 / 2 * 16, "copy in"}}, {}});
        vs.push_back({"base read 512 x 4", (double)elems * 4,
                      [=](hipStream_t s) { hipLaunchKernelGGL((k_read<512, 4>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(512), 0, s, (uint4v *)c.d_q, o, n16); },
                      {{o, n16 * 16, "read"}, {c.d_q, 16, "sink"}}, {}});
    }
    if (has("ship")) {
#define X(F, ID) add_ship<F, OutF32>(c, #F, ID); add_ship<F, OutF16>(c, #F, ID);
        UB_FORMATS(X)
#undef X
    }
    if (has("sw") || has("swb") || has("gw")) {
        // one format / output per run: the filter names it ("sw Q4K_f16" keeps the variants whose names hold that)
#define X(F, ID)                                                                                              \
    if (sel.find(#F "_f32") != std::string::npos) add_sweeps<F, OutF32>(c, #F, has("sw"), has("swb"), has("gw")); \
    if (sel.find(#F "_f16") != std::string::npos) add_sweeps<F, OutF16>(c, #F, has("sw"), has("swb"), has("gw"));
        UB_FORMATS(X)
#undef X
    }
    if (has("sdeq")) {
        add_shapes<Q4K, OutF32>(c, "Q4K", GGQ_TYPE_Q4K);
        add_shapes<Q4K, OutF16>(c, "Q4K", GGQ_TYPE_Q4K);
        add_shapes<Q4_0, OutF32>(c, "Q4_0", GGQ_TYPE_Q4_0);
        add_shapes<Q4_0, OutF16>(c, "Q4_0", GGQ_TYPE_Q4_0);
    }
    if (has("half")) {      // F16 -> f32 through the tile kernel: no schedule, bursts, group starts (against "stream widen")
        for (uint32_t w : {0u, 2u, 3u, 4u, 5u, 6u})
            for (uint32_t l : {0u, 512u, 2048u}) {
                char label[112];
                std::snprintf(label, sizeof label, "half F16_f32 tile W%u %s%u", w ? 1u << w : 0u, l ? "P1 L" : "off ", l);
                add_one<HalfBlocks<false>, OutF32>(c, label, l ? Prefetch{1, l, w} : Prefetch{0, 0, w});
            }
        add_one<HalfBlocks<false>, OutF32>(c, "half F16_f32 tile burst P512 L1024", Prefetch{512, 1024});
        add_one<HalfBlocks<false>, OutF32>(c, "half F16_f32 tile burst P1024 L2048", Prefetch{1024, 2048});
    }
    if (has("wsw")) {       // widen f16 -> f32: steps per thread x group start
        const uint2v *src = (const uint2v *)c.d_in;
        float *dst = (float *)c.d_out;
        const uint64_t n4 = elems / 4;
        auto add = [&](auto steps, uint32_t w) {
            constexpr int S = decltype(steps)::value;
            const unsigned grid = (unsigned)((n4 + STREAM_THREADS * S - 1) / (STREAM_THREADS * S));
            vs.push_back({"wsw widen f16 -> f32 steps " + std::to_string(S) + " group " + std::to_string(w ? 1u << w : 0u), (double)elems * 6,
                          [=](hipStream_t s) { hipLaunchKernelGGL((widen_s<S>), dim3(grid), dim3(STREAM_THREADS), 0, s, src, dst, n4, w); },
                          {{src, elems * 2, "f16 input"}, {dst, elems * 4, "f32 output"}}, {}});
        };
        for (uint32_t w : {0u, 2u, 3u, 4u, 5u, 6u}) {
            add(std::integral_constant<int, 1>(), w);
            add(std::integral_constant<int, 2>(), w);
            add(std::integral_constant<int, 4>(), w);
        }
    }
    if (has("sgw")) {
        add_shapes_gw<Q4K, OutF32>(c, "Q4K");
        add_shapes_gw<Q6K, OutF32>(c, "Q6K");
        add_shapes_gw<Q4K, OutF16>(c, "Q4K");
    }
    if (has("quant")) {
#define XQ(Q) add_quant<Q, InF32>(c, #Q, "f32"); add_quant<Q, InF16>(c, #Q, "f16"); add_quant<Q, InBF16>(c, #Q, "bf16");
        XQ(EncQ4_0) XQ(EncQ4_1) XQ(EncQ5_0) XQ(EncQ5_1) XQ(EncQ8_0) XQ(EncQ8_1) XQ(EncQ8K)
#undef XQ
    }
    if (has("stream")) {
        {   // widen f16 -> f32: d_in (2 bytes per element) -> d_out
            const uint64_t n4 = elems / 4;
            constexpr uint32_t per = STREAM_THREADS * WidenShape<OutF32>::STEPS;
            const uint2v *src = (const uint2v *)c.d_in;
            float *dst = (float *)c.d_out;
            vs.push_back({"stream widen f16 -> f32", (double)elems * 6,
                          [=](hipStream_t s) { hipLaunchKernelGGL((widen_vec<false, OutF32, true>), dim3((unsigned)((n4 + per - 1) / per)), dim3(STREAM_THREADS), 0, s, src, dst, n4); },
                          {{src, n4 * 8, "f16 input"}, {dst, n4 * 16, "f32 output"}}, {}});
        }
        {   // narrow f32 -> f16: d_out -> d_q
            constexpr uint64_t per = (uint64_t)STREAM_THREADS * NARROW_STEPS * InF32::EPL;
            const float *src = (const float *)c.d_out;
            uint16_t *dst = (uint16_t *)c.d_q;
            vs.push_back({"stream narrow f32 -> f16", (double)elems * 6,
                          [=](hipStream_t s) { hipLaunchKernelGGL((narrow_vec<InF32, OutF16>), dim3((unsigned)((elems + per - 1) / per)), dim3(STREAM_THREADS), 0, s, src, dst, elems); },
                          {{src, elems * 4, "f32 input"}, {dst, elems * 2, "f16 output"}}, {}});
        }
        {   // permute_qk over [4096, rows] Q4_K (2304-byte rows): d_in -> d_q
            const uint64_t rb = 2304, rows = elems / 4096 / 256 * 256, nh = rows / 128;
            if (rows >= 256) {
                const uint64_t pieces = rb / 16 * rows;
                const FastDiv32 ppr = FastDiv32::make((uint32_t)(rb / 16)), dhd = FastDiv32::make((uint32_t)(rows / nh));
                const uint8_t *src = c.d_in;
                uint8_t *dst = c.d_q;
                vs.push_back({"stream permute_qk Q4_K rows", (double)rows * rb * 2,
                              [=](hipStream_t s) { hipLaunchKernelGGL((permute_qk_rows<16>), dim3((unsigned)((pieces + STREAM_THREADS - 1) / STREAM_THREADS)), dim3(STREAM_THREADS), 0, s, dst, src, rb, (uint32_t)pieces, ppr, dhd, 0u); },
                              {{src, rows * rb, "rows in"}, {dst, rows * rb, "rows out"}}, {}});
            }
        }
        {   // scale f16 out of place: d_in -> d_q
            const uint64_t pieces = elems / 8;
            const uint4v *src = (const uint4v *)c.d_in;
            uint4v *dst = (uint4v *)c.d_q;
            vs.push_back({"stream scale f16", (double)elems * 4,
                          [=](hipStream_t s) { hipLaunchKernelGGL((scale_vec<SC_F16>), dim3((unsigned)((pieces + STREAM_THREADS - 1) / STREAM_THREADS)), dim3(STREAM_THREADS), 0, s, dst, src, pieces, (uint64_t)0, (uint64_t)0x3D9Au); },
                          {{src, pieces * 16, "elements in"}, {dst, pieces * 16, "elements out"}}, {}});
        }
    }
    {   // name filter: selection words that are not family names keep the variants whose names hold any of them
        static const char *fam[] = {"base", "ship", "sw", "swb", "gw", "sdeq", "sgw", "half", "wsw", "quant", "stream"};
        std::vector<std::string> pats;
        size_t at = 0;
        while (at <= sel.size()) {
            size_t e = sel.find(',', at);
            if (e == std::string::npos) e = sel.size();
            std::string w = sel.substr(at, e - at);
            bool is_fam = false;
            for (const char *f : fam) is_fam = is_fam || w == f;
            if (!w.empty() && !is_fam) pats.push_back(w);
            at = e + 1;
        }
        if (!pats.empty()) {
            std::vector<Variant> keep;
            for (auto &v : vs)
                for (auto &p : pats)
                    if (v.name.find(p) != std::string::npos) { keep.push_back(v); break; }
            vs.swap(keep);
        }
    }
    {   // nothing runs that has not declared a footprint that fits
        std::vector<Variant> ok;
        for (auto &v : vs)
            if (admit(v)) ok.push_back(v);
        vs.swap(ok);
    }
    if (vs.empty()) { std::printf("no variants selected\n"); return 1; }

    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const bool cold = std::getenv("UBENCH_COLD") && std::getenv("UBENCH_COLD")[0] == '1';
    uint4v *d_evict = nullptr;
    const uint64_t evict16 = (2ull << 30) / 16;
    if (cold) {
        d_evict = ub_alloc<uint4v>(evict16 * 16, "eviction buffer");
        if (!fits({d_evict, evict16 * 16, "eviction read"})) return 2;
    }
    for (auto &v : vs) { v.run(st); }                // warm-up
    CK(hipStreamSynchronize(st));
    // every round visits the variants in a fresh random order (so no variant always runs behind the same
    // predecessor); one sample = REP back-to-back launches
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
            if (cold)
                hipLaunchKernelGGL((k_read<512, 4, false>), dim3((unsigned)((evict16 + 2047) / 2048)), dim3(512), 0, st, d_evict, d_evict, evict16);
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
    std::printf("%-44s %10s %10s %12s %8s\n", "variant", "median ms", "best ms", "GB/s(median)", "of 8TB/s");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], best = v.ms[0];
        const double gbps = v.bytes / (med * 1e-3) / 1e9;
        std::printf("%-44s %10.4f %10.4f %12.1f %7.1f%%\n", v.name.c_str(), med, best, gbps, gbps / 80.0);
    }
    return 0;
}
