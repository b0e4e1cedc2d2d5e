// kernels.h -- dequantize kernels for gfx950 (MI355X).
//
// The work is a pure memory stream: 0.33-1.1 bytes read and 2-4 bytes written per
// element, a handful of integer and float operations in between, no reuse and no
// matrix work.  Everything here serves the one thing that decides the speed: the
// order and shape in which bytes reach HBM.  TileGeom / deq_tile below give the
// decomposition and the measurements behind each choice.
#pragma once
#include "formats.h"

namespace ggq {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

// ---- output element types ---------------------------------------------------
struct OutF32 {
    typedef float T;
    static constexpr uint32_t EPL = 4;   // elements per lane per store (16 bytes)
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
        // The empty asm keeps the f32 value opaque.  Without it hipcc folds the
        // producing multiply into the conversion as v_fma_mixlo_f16(x, y, +0), and
        // (-0) + (+0) = +0 loses the sign of a negative zero product.
        asm volatile("" : "+v"(a), "+v"(b));
        // one v_cvt_pk_f16_f32 (round to nearest even) for the pair
        const float2v v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2v));
    }
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, const float (&a)[8]) {
        uint4v v = {pk(a[0], a[1]), pk(a[2], a[3]), pk(a[4], a[5]), pk(a[6], a[7])};
        if (NT) __builtin_nontemporal_store(v, (uint4v *)p);
        else *(uint4v *)p = v;
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
        asm volatile("" : "+v"(a), "+v"(b));   // same reason as OutF16::pk
        const float2v v = {a, b};              // one v_cvt_pk_bf16_f32
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
    }
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, const float (&a)[8]) {
        uint4v v = {pk(a[0], a[1]), pk(a[2], a[3]), pk(a[4], a[5]), pk(a[6], a[7])};
        if (NT) __builtin_nontemporal_store(v, (uint4v *)p);
        else *(uint4v *)p = v;
    }
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, const float (&a)[4], const float (&b)[4]) {
        uint4v v = {pk(a[0], a[1]), pk(a[2], a[3]), pk(b[0], b[1]), pk(b[2], b[3])};
        if (NT) __builtin_nontemporal_store(v, (uint4v *)p);
        else *(uint4v *)p = v;
    }
};

// ---- tile shape ----------------------------------------------------------------
// A workgroup of 512 threads owns one tile and every thread issues exactly two
// 16-byte stores: 4096 elements (16 KiB) for f32 output, 8192 elements (16 KiB) for
// f16 / bf16 output.  Measured on MI355X: one store per thread is bound by workgroup
// launch rate, four or more stretch the window of addresses being written at any
// moment and cost DRAM efficiency; two is the optimum.
constexpr uint32_t WG_THREADS = 512;
constexpr uint32_t STORES_PER_THREAD = 2;

// Division of a 32-bit value by a launch constant as a multiply-high and shifts (round-up method: for d not a
// power of two, l = floor(log2 d) and m = floor(2^(32+l) / d) + 1 when that over-estimates by less than 2^l,
// else the 33-bit magic whose top bit is applied as an add).  Used where a kernel divides by a value the host
// knows at launch: the row gather divides twice per thread (the compiler's general 64-bit division cost it
// 2.5-4.5 % of its speed), the decode kernels with 2-byte output once per workgroup.
struct FastDiv32 {
    uint32_t magic, shift, add, d;
    __host__ static FastDiv32 make(uint32_t d) {
        FastDiv32 f{0, 0, 0, d};
        if ((d & (d - 1)) == 0) {                      // power of two (and d = 1): shift only
            while ((1u << f.shift) < d) ++f.shift;
            return f;
        }
        uint32_t l = 31;
        while (!((d >> l) & 1u)) --l;
        const uint64_t two = 1ull << (32 + l);
        uint64_t m = two / d;
        const uint64_t rem = two - m * d;
        f.shift = l;
        if (d - rem < (1ull << l)) {
            f.magic = (uint32_t)(m + 1);
        } else {
            m = 2 * m + (2 * rem >= d ? 1 : 0);
            f.magic = (uint32_t)(m + 1);
            f.add = 1;
        }
        return f;
    }
    __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
        if (magic == 0) return n >> shift;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t q = __umulhi(n, magic);
#else
        const uint32_t q = (uint32_t)(((uint64_t)n * magic) >> 32);
#endif
        return add ? (((n - q) >> 1) + q) >> shift : q >> shift;
    }
};

// Burst prefetch schedule, in tiles (= workgroups).  period 0 switches it off.
// See deq_tile for what it does and why.
struct Prefetch {
    uint32_t period;   // a burst of reads is issued once every `period` workgroups
    uint32_t lead;     // ... for the tiles `lead` workgroups ahead of the burst's first workgroup
    uint32_t wait;     // k > 0: groups of 2^k consecutive workgroups start together (group_wait below); 0: off
    FastDiv32 per;     // `period` as a divisor (made on the host)
    __host__ Prefetch(uint32_t p = 0, uint32_t l = 0, uint32_t w = 0) : period(p), lead(l), wait(w), per(FastDiv32::make(p)) {}
};

// Group start.  The workgroups of tiles [g, g + 2^k) all load one dword of the line where tile g's OUTPUT begins --
// memory that is about to be overwritten, so the line is cold and the load takes a full trip to HBM -- and do not
// issue their tile loads before it is back.  The line returns to all of them at about the same time, so the group
// reads its 30-100 KB of packed input together and then writes its 128 KB - 1 MB of contiguous output together,
// instead of eight XCDs drifting apart over the tensor.  The value is not used: it is reduced to a zero that the
// compiler cannot see through and added to the source address, which is what makes the tile loads wait.
// This is the alternative to the burst prefetch for streams with a larger share of reads: with the packed input
// at 21 % of the traffic and more (every 2-byte output except Q2_K and Q3_K; the 8-bit formats with 4-byte
// output) it beats the burst by 2-7 %, below that it loses by 4-17 % (measurements at schedule.h).
// The address is wave-uniform, so the compiler issues the load through the scalar cache, and that is part of the
// effect: the same probe forced onto the vector path (measured) turns the gain into a loss of 10-18 points.  The
// description above is therefore a model that fits the measurements, not an established mechanism.
// Cache assumption: the probe brings a line of the destination into the scalar cache, and this same kernel then
// overwrites that line with vector stores, which the scalar cache does not see.  Nothing in this library reads a
// decoded tensor through the scalar cache within the same kernel, and a LATER kernel that does (this kernel's own
// probe on the next launch into the same buffer, or a consumer with wave-uniform loads) is safe because every kernel
// dispatch on this runtime starts with an acquire that invalidates the scalar and vector L1 caches -- also between
// the nodes of a replayed graph and between back-to-back launches in one stream.  The probed value itself is never
// used, so a stale line could at worst be seen by a consumer, never by the decode.
__device__ __forceinline__ uint32_t group_wait(const void *group_dst) {
    // the address is named as global memory (address space 1) outright: a pointer of unknown address space -- one
    // assembled from integers, as in deq_seg -- would be loaded with a flat (vector) instruction
    typedef const __attribute__((address_space(1))) uint32_t *GlobalU32;
    const uint32_t seen = *(GlobalU32)(uintptr_t)group_dst;
    uint32_t zero;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(seen));
    return __builtin_amdgcn_readfirstlane(zero);
}

template <class F, class O>
struct TileGeom {
    static constexpr uint32_t ELEMS = WG_THREADS * O::EPL * STORES_PER_THREAD;
    static constexpr uint32_t TB = ELEMS / F::N;               // blocks per tile
    static constexpr uint32_t TBYTES = TB * F::BYTES;          // packed bytes per tile
    static constexpr uint32_t SLOTS = (TBYTES + 15) / 16 + 1;  // 16-byte slots a tile can touch
    static constexpr uint32_t LDS_SLOTS = SLOTS + 1;           // + slack for the funnel read
    static constexpr uint32_t LPT = (SLOTS + WG_THREADS - 1) / WG_THREADS;   // loads per thread
    static constexpr uint32_t LINES = (TBYTES + 127) / 128 + 1;  // 128-byte lines a tile can touch
    static constexpr uint32_t TPW = WG_THREADS / LINES;        // tiles one workgroup can prefetch (a line per thread)
    static_assert(ELEMS % F::N == 0, "a tile is a whole number of blocks");
    static_assert(TPW >= 1, "tile too large to prefetch with one line per thread");
};

// One tile of a batch: everything a workgroup needs, in one 16-byte record, so
// the lookup is a single scalar load with nothing depending on an earlier load.
struct TileRef {
    uint64_t src_nblk;   // bits 0..47 address of the tile's first block, bits 48..63 blocks in the tile
    uint64_t dst;        // address of the tile's first output element (16-byte aligned)
};

// One tensor of a batch group, as the host describes it: the tile table is expanded from these on the
// device (a 7B-parameter model is ~1.6 M tiles = 26 MB of records, which the host would spend longer
// writing and copying than the decode takes).
struct TensorDesc {
    uint64_t src, dst;      // first block, first output element
    uint64_t blocks;        // blocks in the tensor
    uint64_t first_tile;    // index of the tensor's first tile in the group (ascending over the array)
};

// Where tile t's record sits in its group's table.  Workgroups go to the 8 XCDs in round robin, so with the
// records in tile order the 8 records of one 128-byte line are read through 8 different L2s and every
// workgroup misses on its record.  For 4-byte outputs the record of tile t sits at
// (t & ~63) | ((t & 7) << 3) | ((t >> 3) & 7): the 8 records one XCD reads out of every 64 tiles share a
// line.  Measured on 1 Gi elements against tile order, 3 and 30 launches per sample: Q4_K -> f32 6694 /
// 6686 GB/s against 6653 / 6661, Q6_K 6612 / 6629 against 6603 / 6607, Q8_0 6475 / 6479 against 6454 /
// 6462 (half of the batch kernel's distance to the single-tensor kernel).  2-byte outputs keep tile order:
// Q4_K and Q2_K gain 0-0.7 % there, Q6_K loses 5 %.  A group's table is padded to a multiple of 64 records.
template <bool PER_XCD>
__host__ __device__ __forceinline__ uint64_t record_slot(uint64_t t) {
    return PER_XCD ? ((t & ~63ull) | ((t & 7ull) << 3) | ((t >> 3) & 7ull)) : t;
}
template <class O> struct RecordOrder { static constexpr bool PER_XCD = sizeof(typename O::T) == 4; };

// the records of tiles [tile0, tile_end) of the group; one thread per tile, the tensor found by binary search
template <bool PER_XCD>
__global__ __launch_bounds__(256) void build_tile_table(TileRef *__restrict__ table, const TensorDesc *__restrict__ desc,
                                                        uint32_t ntensors, uint64_t tile0, uint64_t tile_end,
                                                        uint32_t tile_blocks, uint32_t tile_bytes, uint32_t tile_out_bytes) {
    const uint64_t tile = tile0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tile >= tile_end) return;
    uint32_t lo = 0, hi = ntensors - 1;            // largest k with desc[k].first_tile <= tile
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (desc[mid].first_tile <= tile) lo = mid;
        else hi = mid - 1;
    }
    const TensorDesc d = desc[lo];
    const uint64_t t = tile - d.first_tile;
    const uint64_t left = d.blocks - t * tile_blocks;
    const uint64_t nblk = left < tile_blocks ? left : tile_blocks;
    table[record_slot<PER_XCD>(tile)] = TileRef{(d.src + t * tile_bytes) | (nblk << 48), d.dst + t * tile_out_bytes};
}

// Which 128-byte line, if any, this thread touches for the burst prefetch.
// Workgroups are numbered t = 0, 1, ... in dispatch order; the first
// NB = ceil(period / TPW) workgroups of every `period` are on prefetch duty: together
// they touch every line of the `period` tiles that start `lead` tiles after the first
// workgroup of their period, one line per thread (thread -> (tile j, line l)).
// `tile_src(tt)` returns the address of tile tt's packed bytes, or null if that tile
// does not exist or is not a whole tile.
// FAST: the phase t mod period by multiplication (t < 2^32) instead of the general 64-bit remainder.  With
// 2-byte output, where the decode is close to VALU-bound, that is worth 1-1.7 % for Q3_K and IQ4_XS, 0.3 % for
// Q2_K and nothing for the other formats; with 4-byte output the same cut measured 0.1-0.5 % SLOWER for Q4_K,
// Q3_K, IQ4_XS and Q6_K, in short and in long runs of launches, so those kernels keep the remainder.
template <class G, bool FAST, class SrcOf>
__device__ __forceinline__ const uint8_t *prefetch_line(uint64_t t, const Prefetch &pf, SrcOf tile_src) {
    if (pf.period == 0) return nullptr;
    const uint32_t ph = FAST ? (uint32_t)t - pf.per.div((uint32_t)t) * pf.period : (uint32_t)(t % pf.period);
    const uint32_t nb = (pf.period + G::TPW - 1) / G::TPW;
    if (ph >= nb) return nullptr;
    const uint32_t j = threadIdx.x / G::LINES, l = threadIdx.x % G::LINES;
    // Duty workgroup ph takes TPW CONSECUTIVE tiles, so its touches sweep one contiguous
    // run of packed bytes (~60 KB).  Handing each duty workgroup the tiles of its own XCD
    // instead (every 8th tile, so the lines land in the L2 that will read them) was
    // measured 18 % slower: what pays is the contiguous read burst at the memory, not
    // the cache level the bytes wait in.
    const uint32_t k = ph * G::TPW + j;             // tile within the prefetched period
    if (j >= G::TPW || k >= pf.period) return nullptr;
    const uint8_t *base = tile_src(t - ph + pf.lead + k);
    if (base == nullptr) return nullptr;
    const uint8_t *line = (const uint8_t *)((uintptr_t)base & ~(uintptr_t)127) + 128u * l;
    return line < base + G::TBYTES ? line : nullptr;   // only lines holding bytes of that tile
}

// Decode one tile.
//   src   first block of the tile (any alignment), nblk blocks present (<= TB)
//   out   first output element of the tile (16-byte aligned)
//   line  a 128-byte line of a far-ahead tile to touch, or null (prefetch_line)
// Steps: (1) aligned 16-byte loads of the packed bytes into LDS, consecutive
// threads -> consecutive 16-byte slots of the enclosing 16-byte grid.  These are
// PLAIN loads on purpose: a non-temporal load drops the line from the caches behind
// it, and whenever tiles do not start on 128-byte lines (any source offset, and every
// format whose tile is not a multiple of 128 bytes: Q2_K, Q3_K, Q6_K, Q8K) the next
// workgroup needs that same boundary line again and must fetch it from HBM --
// measured -13 % on Q4_K at a 16-byte source offset, -14 % on Q6_K; (2) burst
// prefetch: one dword of `line` is loaded and never used.  The packed input is only
// ~12 % of the traffic, but reads that trickle into HBM between the writes cost far
// more than their bytes (every few reads force the memory to turn its bus around).
// Bunching them -- a few workgroups per period pull a whole period's input, over 1 MB,
// into the memory-side cache at once -- lets every other workgroup find its input
// on chip: measured +15 % on Q4_K -> f32 against no prefetch, +8 % against a
// per-workgroup L2 prefetch of the same lead; (3) barrier; (4) two decode-and-store
// steps: thread t handles EPL consecutive elements at (step * 512 + t) * EPL, so each
// wave store instruction writes one contiguous 1 KiB run (8 whole 128-byte lines)
// and the workgroup writes two contiguous 8 KiB runs.
template <class F, class O, bool NT, bool NTL = false>
__device__ __forceinline__ void deq_tile(uint4v *lds, const uint8_t *src, uint32_t nblk,
                                         typename O::T *out, const uint8_t *line) {
    typedef TileGeom<F, O> G;
    const uint32_t tid = threadIdx.x;
    const uint32_t off0 = (uint32_t)((uintptr_t)src & 15u);
    const uint4v *base16 = (const uint4v *)(src - off0);
    const uint32_t nslots = (off0 + nblk * F::BYTES + 15u) >> 4;
#pragma unroll
    for (uint32_t k = 0; k < G::LPT; ++k) {
        uint32_t s = tid + WG_THREADS * k;
        if (s < nslots) lds[s] = NTL ? __builtin_nontemporal_load(base16 + s) : base16[s];
    }
    uint32_t touch = 0;
    if (line != nullptr) touch = *(const uint32_t *)line;
    __syncthreads();
    LdsBytes r{(const uint32_t *)lds};
#pragma unroll
    for (uint32_t i = 0; i < STORES_PER_THREAD; ++i) {
        uint32_t el = (i * WG_THREADS + tid) * O::EPL;
        uint32_t bt = el / F::N, e = el % F::N;
        if (bt < nblk) {
            uint32_t b = off0 + bt * F::BYTES;
            const Scale sc = F::scale(r, b, e);      // e % EPL == 0: one sub-block, one scale
            float y[O::EPL];
            F::template vals<O::EPL / 4>(r, b, e, sc, y);
            O::template store<NT>(out + el, y);
        }
    }
    // keeps the prefetch load alive; it is waited for only here, after the stores
    asm volatile("" ::"v"(touch));
}

// ---- one tensor per launch: workgroup b decodes tile b ---------------------------
// The grid covers every tile (no loop): the hardware dispatches workgroups in
// order, so at any moment the in-flight stores cover one compact, steadily advancing
// window of the output -- the access pattern HBM serves best.  (An XCD-aware order, in which each
// XCD takes C = 2..64 consecutive tiles so that the boundary lines two tiles share meet in one L2,
// measured equal for C <= 4 and 1-10 % slower beyond, also for the formats whose tiles do not end
// on 128-byte lines: the shared lines already come from the memory-side cache.)
template <class F, class O, bool NT, bool NTL = false>
__global__ __launch_bounds__(WG_THREADS) void deq_one(const uint8_t *__restrict__ src,
                                                      typename O::T *__restrict__ dst,
                                                      uint64_t nblocks, Prefetch pf) {
    typedef TileGeom<F, O> G;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = blockIdx.x;
    const uint64_t rem = nblocks - t * G::TB;
    const uint32_t nblk = rem < G::TB ? (uint32_t)rem : G::TB;
    const uint32_t z = pf.wait ? group_wait(dst + ((t >> pf.wait) << pf.wait) * G::ELEMS) : 0u;
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        return (tt + 1) * G::TB <= nblocks ? src + tt * G::TBYTES : nullptr;   // whole tiles inside the tensor
    });
    deq_tile<F, O, NT, NTL>(lds, src + t * G::TBYTES + z, nblk, dst + t * G::ELEMS, line);
}

// ---- many tensors per launch: workgroup b decodes tile tile0 + b of the group --------
// `table`: the group's records (record_slot); `ntiles`: tiles in the group; a group beyond the grid limit
// takes several launches, each starting at its `tile0`.
// (Touching the table record of a workgroup 64-2048 tiles ahead, to spare that workgroup the
// miss on its record, was measured 1-2 % slower: the record load is not what the kernel waits for.)
template <class F, class O, bool NT>
__global__ __launch_bounds__(WG_THREADS) void deq_batch(const TileRef *__restrict__ table, uint64_t tile0,
                                                        uint64_t ntiles, Prefetch pf) {
    typedef TileGeom<F, O> G;
    constexpr bool PX = RecordOrder<O>::PER_XCD;
    __shared__ uint4v lds[G::LDS_SLOTS];
    const uint64_t t = tile0 + blockIdx.x;
    const TileRef me = table[record_slot<PX>(t)];
    // the group's first tile may belong to another tensor than this one: its output line is as cold as any
    const uint32_t z = pf.wait ? group_wait((const void *)table[record_slot<PX>((t >> pf.wait) << pf.wait)].dst) : 0u;
    const uint8_t *line = prefetch_line<G, sizeof(typename O::T) == 2>(t, pf, [&](uint64_t tt) -> const uint8_t * {
        if (tt >= ntiles) return nullptr;
        const uint64_t far = table[record_slot<PX>(tt)].src_nblk;
        // whole tiles only (a short tile is a tensor's last one)
        return (far >> 48) == G::TB ? (const uint8_t *)(far & 0xFFFFFFFFFFFFull) : nullptr;
    });
    deq_tile<F, O, NT>(lds, (const uint8_t *)(me.src_nblk & 0xFFFFFFFFFFFFull) + z, (uint32_t)(me.src_nblk >> 48),
                       (typename O::T *)me.dst, line);
}

// ---- F16 / BF16 sources: plain widening, no LDS -------------------------------
// (ggml-quants/src/structs/half.rs:19-21, 35-37).  A thread converts 4 consecutive elements per step
// from one 8-byte load into one 16-byte store (f32 out; 8 bytes for 2-byte outputs).
// Requires src 8-byte aligned and dst 16-byte aligned; n4 = elements / 4.
// Shape of the streaming kernels that need no LDS (this one, narrow_vec, permute_qk_rows): 256 threads
// and one step whose wider side is 16 bytes per thread, so a wave stores 1 KiB and ends -- measured
// against the decode kernels' 512 x 2 shape on 1 Gi elements, three and thirty launches in a row:
// f16 -> f32 6610 GB/s against 6080 (both settings), the row gather 6590 / 6300 against 6070 / 6090.
// Where a thread's store is 8 bytes (f16 <-> bf16 here, narrow_vec) two steps do better (6530 against 5710).
// The decode kernels keep their shape:
// their workgroups stage a tile through LDS behind a barrier, and with one store per thread that
// fixed cost per wave dominates.
// (Treating f16 as 64-byte blocks and going through the tile kernel measured +2 % without the
// burst prefetch and -3 % with it: reads are a third of the traffic here, nothing to bunch.)
constexpr uint32_t STREAM_THREADS = 256;
template <class O> struct WidenShape { static constexpr uint32_t STEPS = O::EPL == 4 ? 1 : 2; };

template <bool BF16SRC, class O, bool NT>
__global__ __launch_bounds__(STREAM_THREADS) void widen_vec(const uint2v *__restrict__ src,
                                                            typename O::T *__restrict__ dst, uint64_t n4) {
    constexpr uint32_t S = WidenShape<O>::STEPS;
    const uint64_t base = (uint64_t)blockIdx.x * STREAM_THREADS * S + threadIdx.x;
    uint2v v[S];
#pragma unroll
    for (uint32_t k = 0; k < S; ++k) {
        uint64_t i = base + k * STREAM_THREADS;
        if (i < n4) v[k] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (uint32_t k = 0; k < S; ++k) {
        uint64_t i = base + k * STREAM_THREADS;
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
            O::template store<NT>(dst + 4 * i, y);
        } else {
            uint2v o = {O::pk(y[0], y[1]), O::pk(y[2], y[3])};
            if (NT) __builtin_nontemporal_store(o, (uint2v *)(dst + 4 * i));
            else *(uint2v *)(dst + 4 * i) = o;
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
