/*
 * ggq.h -- C ABI of libggq: GGUF block dequantization on AMD MI355X (gfx950).
 *
 * The library replaces the per-block decode path of the reference crate
 * `ggml-quants` (Rust, CPU) with hand-written HIP kernels.  Every entry point
 * below names the reference interface it stands in for (paths relative to the
 * reference checkout).  All signatures are plain C: pointers, sizes and u32
 * type ids; no C++ or framework types cross this boundary.
 *
 * Type ids are the reference's `GGmlType` discriminants verbatim
 * (ggus/src/tensor.rs:14-51): F32=0 F16=1 Q4_0=2 Q4_1=3 Q5_0=6 Q5_1=7 Q8_0=8
 * Q8_1=9 Q2K=10 Q3K=11 Q4K=12 Q5K=13 Q6K=14 Q8K=15 IQ4NL=20 IQ4XS=23 BF16=30.
 *
 * There is no CPU fallback: without a usable HIP device every compute entry
 * point returns GGQ_DEVICE.
 */
#ifndef GGQ_H
#define GGQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces `QuantizeError` (ggml-quants/src/lib.rs:107-113) plus the cases the
 * reference handles by panicking: `todo!()` bodies -> GGQ_UNSUPPORTED,
 * `reslice` alignment panic (xtask/src/utils/operator/cast.rs:164-169) ->
 * GGQ_MISALIGNED. */
typedef enum ggq_status {
    GGQ_OK = 0,
    GGQ_INDIVISIBLE = 1,     /* QuantizeError::Indivisible    */
    GGQ_LENGTH_MISMATCH = 2, /* QuantizeError::LengthMismatch */
    GGQ_UNSUPPORTED = 3,     /* type id has no decoder (IQ1/IQ2/IQ3 codebook types, Q4_0_x_x, integers) */
    GGQ_DEVICE = 4,          /* HIP runtime error or no device; see ggq_last_error() */
    GGQ_MISALIGNED = 5,      /* device pointer alignment requirement not met */
    GGQ_INVALID = 6,         /* null pointer / bad argument */
    GGQ_TOO_LARGE = 7        /* input beyond the fixed capacity of ggq_dequantize_noalloc */
} ggq_status;

enum {
    GGQ_TYPE_F32 = 0, GGQ_TYPE_F16 = 1, GGQ_TYPE_Q4_0 = 2, GGQ_TYPE_Q4_1 = 3,
    GGQ_TYPE_Q5_0 = 6, GGQ_TYPE_Q5_1 = 7, GGQ_TYPE_Q8_0 = 8, GGQ_TYPE_Q8_1 = 9,
    GGQ_TYPE_Q2K = 10, GGQ_TYPE_Q3K = 11, GGQ_TYPE_Q4K = 12, GGQ_TYPE_Q5K = 13,
    GGQ_TYPE_Q6K = 14, GGQ_TYPE_Q8K = 15, GGQ_TYPE_IQ4NL = 20, GGQ_TYPE_IQ4XS = 23,
    GGQ_TYPE_BF16 = 30
};

/* ---- type table ---------------------------------------------------------
 * Replaces `DataBlock::COUNT` (ggml-quants/src/lib.rs:11-21), `size_of` of the
 * block structs as used by `GGmlType::size` (ggus/src/tensor.rs:102-144) and
 * the layout names of ggml-quants/src/types.rs:10-33.
 * Both return 0 for a type this library does not decode. */
size_t ggq_block_elems(uint32_t ggml_type);
size_t ggq_block_bytes(uint32_t ggml_type);
/* Layout name as the reference prints it ("q40", "q4k", "iq4nl", ...,
 * types.rs:35-62); NULL if unsupported. */
const char *ggq_type_name(uint32_t ggml_type);
/* Replaces `GGmlTypeSize::elements_to_bytes` (ggus/src/tensor.rs:83-96):
 * shape[0] is the contiguous dimension and must be a multiple of the block
 * size.  Returns GGQ_INDIVISIBLE where the reference asserts. */
ggq_status ggq_elements_to_bytes(uint32_t ggml_type, const uint64_t *shape, size_t ndim,
                                 uint64_t *bytes_out);

const char *ggq_status_str(ggq_status st);
/* Text of the last HIP failure on the calling thread ("" if none). */
const char *ggq_last_error(void);
/* Number of visible HIP devices (0 if the runtime is unusable). */
int ggq_device_count(void);

/* ---- dequantize, host buffers -------------------------------------------
 * Replaces `<Blk as QuantExt<T, N>>::dequantize_slice(dst: &mut [T], src: &[Blk])`
 * (ggml-quants/src/lib.rs:135-147).  `ggml_type` selects Blk, `out_type`
 * selects T: F32 (0), F16 (1) or BF16 (30).  dst and src are HOST memory owned
 * by the caller (typically an mmap window and a fresh anonymous mapping,
 * xtask/src/utils/operator/cast.rs:150-161); any alignment is accepted.
 * Checks run in the reference's order: dst_elems % N -> GGQ_INDIVISIBLE, then
 * src_blocks != dst_elems / N -> GGQ_LENGTH_MISMATCH.
 * Synchronous; runs on the calling thread's current HIP device through a
 * per-thread pipeline of pinned staging buffers (H2D copy, decode kernel and
 * D2H copy of consecutive chunks overlap on separate streams).  Safe to call
 * from several threads at once, as the reference is from its writer threads
 * (xtask/src/utils/write.rs:64-68). */
ggq_status ggq_dequantize_slice(uint32_t ggml_type, uint32_t out_type,
                                void *dst, size_t dst_elems,
                                const void *src, size_t src_blocks);

/* ---- dequantize, device-resident ----------------------------------------
 * Same contract and checks as ggq_dequantize_slice, but d_dst and d_src are
 * device pointers on the current device, and the kernel is enqueued on
 * `hip_stream` (a hipStream_t, NULL = default stream) without synchronizing.
 * d_dst must be 16-byte aligned (GGQ_MISALIGNED otherwise); d_src may have any
 * alignment.  d_dst must be readable as well as writable: some kernels load a
 * word of the destination before they overwrite it (only for its timing; what
 * d_dst held before never reaches the result).  No counterpart in the reference
 * (it has no device memory); this is the form the hot path is benchmarked through. */
ggq_status ggq_dequantize_device(uint32_t ggml_type, uint32_t out_type,
                                 void *d_dst, size_t dst_elems,
                                 const void *d_src, size_t src_blocks,
                                 void *hip_stream);

/* ---- quantize -------------------------------------------------------------
 * Replaces `<Blk as QuantExt<T, N>>::quantize_slice(dst: &mut [Blk], src: &[T])`
 * (ggml-quants/src/lib.rs:121-133) for the block types the reference implements:
 * F16, BF16 (1-element blocks, structs/half.rs:8-38), Q4_0, Q4_1, Q5_0, Q5_1, Q8_0,
 * Q8_1 and Q8K.  `in_type` selects T: F32 (0), F16 (1) or BF16 (30); 16-bit inputs
 * are widened first, as lib.rs:67-69 / 82-84 do.  Every other target type returns
 * GGQ_UNSUPPORTED (`todo!()` in the reference).  Checks, in the reference's order:
 * src_elems % N -> GGQ_INDIVISIBLE, then dst_blocks != src_elems / N ->
 * GGQ_LENGTH_MISMATCH.  Output bytes equal the reference's for the same input; where
 * Rust leaves the result open (f32::min / max of +0 and -0) the rule is -0 < +0.
 * Host form: any alignment, synchronous, staged through the same per-device pipeline
 * pool as ggq_dequantize_slice.  Device form: d_src and d_dst 16-byte aligned
 * (GGQ_MISALIGNED otherwise), enqueued on `hip_stream`, not synchronized; the kernels
 * may load a word of d_dst before they overwrite it (for timing only, as the decode
 * kernels do), so d_dst must be readable device memory. */
ggq_status ggq_quantize_slice(uint32_t ggml_type, uint32_t in_type,
                              void *dst, size_t dst_blocks,
                              const void *src, size_t src_elems);
ggq_status ggq_quantize_device(uint32_t ggml_type, uint32_t in_type,
                               void *d_dst, size_t dst_blocks,
                               const void *d_src, size_t src_elems,
                               void *hip_stream);

/* ---- block-granular rearrangement (no decoding) ------------------------------
 * Device forms of the reference's merge-linear / split-linear / permute-qk operators
 * (xtask/src/utils/operator/merge.rs:224-358, permute_qk.rs:46-69).  A tensor of shape
 * [row_elems, rows] is `rows` rows of elements_to_bytes([row_elems]) bytes each; any
 * GGmlType with a known block size is accepted (blocks stay opaque), also the ones this
 * library cannot decode.  row_elems must be a whole number of blocks (GGQ_INDIVISIBLE).
 * All three enqueue on `hip_stream` and do not synchronize.
 *
 * ggq_concat_rows_device : `concat(1, parts)` (merge.rs:279-324) -- merge_qkv,
 *   merge_gate_up: the parts' rows one after another.  part_rows[i] rows from d_parts[i].
 * ggq_split_rows_device  : `split(1, tensor, sizes)` (merge.rs:326-358) -- split_qkv,
 *   split_gate_up: the inverse.
 * ggq_permute_qk_device  : `permute_qk(tensor, n_head)` (permute_qk.rs:46-69): within every
 *   head of dh = rows / n_head rows, dst row 2b + a <- src row a * dh/2 + b.
 *   rows must be a multiple of 2 * n_head (GGQ_INDIVISIBLE otherwise). */
ggq_status ggq_concat_rows_device(uint32_t ggml_type, uint64_t row_elems, void *d_dst,
                                  const void *const *d_parts, const uint64_t *part_rows, size_t n_parts,
                                  void *hip_stream);
ggq_status ggq_split_rows_device(uint32_t ggml_type, uint64_t row_elems, const void *d_src,
                                 void *const *d_parts, const uint64_t *part_rows, size_t n_parts,
                                 void *hip_stream);
ggq_status ggq_permute_qk_device(uint32_t ggml_type, uint64_t row_elems, uint64_t rows, uint64_t n_head,
                                 void *d_dst, const void *d_src, void *hip_stream);

/* ---- scaling a float tensor ---------------------------------------------------
 * Device form of `scale_tensor` / `scale_data` of the reference's to-llama operator
 * (xtask/src/utils/operator/to_llama.rs:99-125): d_dst[i] = d_src[i] * scale over `elems`
 * elements of type F32 (0), F16 (1), BF16 (30) or F64 (28), computed in that type: the factor
 * is converted to the type first, as the reference does (`scale as f32`, f16::from_f64,
 * bf16::from_f64), and every product is rounded once, to nearest even.  Other types:
 * GGQ_UNSUPPORTED (`todo!()` in the reference).  d_dst may equal d_src; both must be aligned
 * to the element size (GGQ_MISALIGNED), 16-byte alignment of both takes the fast path.
 * Enqueues on `hip_stream`, does not synchronize. */
ggq_status ggq_scale_device(uint32_t ggml_type, void *d_dst, const void *d_src, size_t elems,
                            double scale, void *hip_stream);

/* ---- dequantize, batch of device-resident tensors -------------------------
 * One call decodes many tensors (a whole model's tensor table) with one kernel
 * launch per block type present, instead of one per tensor.  Stands in for the
 * loop over tensors that the reference's cast operator performs
 * (xtask/src/utils/operator/cast.rs:73-90), each iteration of which ends in
 * dequantize_slice.  A plan is a hybrid: tensors of 64 Mi elements and more whose
 * (type, output) pair decodes faster on its own than through the shared launch
 * (2-byte outputs of most types, the 8-bit types with 4-byte output) get one launch
 * each inside the plan; like ggq_dequantize_device, those launches read one dword of
 * d_dst, so every d_dst must be readable as well as writable. */
typedef struct ggq_tensor {
    uint32_t ggml_type;   /* block type of src                      */
    uint32_t reserved;    /* must be 0                              */
    const void *d_src;    /* device pointer, packed blocks          */
    void *d_dst;          /* device pointer, 16-byte aligned        */
    uint64_t blocks;      /* number of blocks in d_src              */
} ggq_tensor;

typedef struct ggq_batch ggq_batch;

/* Validates the table, builds and uploads the launch plan (allocates a small
 * device table).  The plan can be run any number of times. */
ggq_status ggq_batch_create(const ggq_tensor *tensors, size_t count, uint32_t out_type,
                            ggq_batch **out);
/* The same for the other direction: a plan that QUANTIZES many device-resident tensors with one kernel launch per
 * target block type -- the reference's loop over tensors when its iterations end in quantize_slice
 * (xtask/src/utils/operator/cast.rs:140-148 -> ggml-quants/src/lib.rs:121-133).  In each entry d_src points at
 * `blocks * ggq_block_elems(ggml_type)` ELEMENTS of type `in_type` (F32, F16 or BF16, the same for the whole plan)
 * and d_dst receives `blocks` packed blocks; both 16-byte aligned (GGQ_MISALIGNED), except that a tensor whose
 * elements and blocks start exactly where those of the previous tensor of its type end continues that tensor's run
 * and needs no alignment of its own.  Target types: those
 * ggq_quantize_device encodes (others: GGQ_UNSUPPORTED); F16 / BF16 targets, a type's only run, and runs of
 * adjacent tensors with 64 Mi elements and more, get one launch per run.  The plan is run, counted and destroyed with ggq_batch_run / _launches / _destroy. */
ggq_status ggq_quant_batch_create(const ggq_tensor *tensors, size_t count, uint32_t in_type,
                                  ggq_batch **out);
/* Enqueues the plan's kernels on `hip_stream`; does not synchronize.  The device that was
 * current at ggq_batch_create must be current again (GGQ_INVALID otherwise). */
ggq_status ggq_batch_run(ggq_batch *batch, void *hip_stream);
/* Kernel launches one run issues (for measurement bookkeeping). */
size_t ggq_batch_launches(const ggq_batch *batch);
/* ... and how many of them go through a tile table (one per block type that has tensors in the table); the rest
 * are runs with a launch of their own and F16 / BF16 tensors. */
size_t ggq_batch_table_launches(const ggq_batch *batch);
void ggq_batch_destroy(ggq_batch *batch);

/* ---- streamed host -> device decode ---------------------------------------
 * A context owns pinned staging memory, device staging and two HIP streams on
 * one device.  ggq_dequantize_async copies packed blocks from host memory
 * (pageable or pinned) in chunks and decodes each chunk on the device as soon
 * as it lands, copy and decode overlapping; the decoded tensor stays in device
 * memory at d_dst.  The call returns once the last chunk has been staged;
 * ggq_sync waits for everything enqueued.  A page-locked source (hipHostMalloc /
 * hipHostRegister) is read in place by the DMA engine, without staging, and must
 * therefore stay valid until ggq_sync; a pageable source may be reused as soon as
 * the call returns.  d_dst is written on the context's own streams.  No counterpart
 * in the reference;
 * this is the whole-model conversion path (one context per GPU, tensors dealt
 * to GPUs by an index partition).
 * Threading: a context belongs to ONE caller at a time -- its slots and streams are not
 * locked; calls on the same context from several threads must be serialized by the caller.
 * Different contexts, also on the same device, are independent and may be used concurrently. */
typedef struct ggq_ctx ggq_ctx;
ggq_status ggq_ctx_create(int device, size_t staging_bytes, ggq_ctx **out);
ggq_status ggq_dequantize_async(ggq_ctx *ctx, uint32_t ggml_type, uint32_t out_type,
                                void *d_dst, size_t dst_elems,
                                const void *h_src, size_t src_blocks);
/* Copies `bytes` of host memory to d_dst through the same pipeline, without decoding: pageable
 * memory is staged through the context's pinned slots (the staging memcpy of one chunk overlaps the
 * DMA of the one before), page-locked memory is read in place.  For callers that bring many packed
 * tensors to the device and then decode them with ONE ggq_batch_* plan (one launch per block type)
 * instead of one launch per tensor.  Source lifetime and ggq_sync as for ggq_dequantize_async.
 * No counterpart in the reference (its tensors stay in the file mapping, xtask/src/utils/read.rs:49-55). */
ggq_status ggq_upload_async(ggq_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
/* Orders `hip_stream` (on the context's device) behind the context: work enqueued on hip_stream after this
 * call starts only when everything enqueued on the context so far -- copies and decodes -- has completed.
 * The host does not wait, so it can go on staging the next tensors while a plan for the ones just uploaded
 * (ggq_batch_run on hip_stream) waits its turn on the device.  No counterpart in the reference. */
ggq_status ggq_ctx_fence(ggq_ctx *ctx, void *hip_stream);
/* The other direction: copies and decodes enqueued on the context AFTER this call start only when everything
 * enqueued on `hip_stream` so far has completed.  For a caller whose destination memory comes from a
 * stream-ordered pool (torch's caching allocator): a block the pool hands out again may still be read by work
 * queued on the caller's stream, and the context's streams are not ordered behind that stream by themselves.
 * The host does not wait.  No counterpart in the reference. */
ggq_status ggq_ctx_wait_stream(ggq_ctx *ctx, void *hip_stream);
ggq_status ggq_sync(ggq_ctx *ctx);
void ggq_ctx_destroy(ggq_ctx *ctx);

/* ---- streamed host -> device decode that allocates nothing ------------------
 * No counterpart in the reference.  For callers that cannot call hipMalloc: code under
 * HIP graph capture, tight serving loops, harnesses that forbid allocation.  Does what one
 * ggq_dequantize_async call does -- packed blocks travel from host memory to a device staging
 * buffer, the decode kernel reads them there and writes d_dst -- but needs no context: the
 * staging buffer is a fixed device array of GGQ_NOALLOC_MAX_BYTES inside the library (every
 * GPU the library runs on holds one until the process ends), the copy and the kernel are
 * enqueued on `hip_stream` (NULL = default stream) and the call does not synchronize.  It calls
 * no hipMalloc / hipFree and creates no stream or event.
 * h_src must be page-locked host memory (hipHostMalloc / hipHostRegister; GGQ_INVALID
 * otherwise) and stay valid until the stream has passed the copy; src_blocks * block bytes
 * must not exceed GGQ_NOALLOC_MAX_BYTES (GGQ_TOO_LARGE otherwise -- split the tensor).
 * Otherwise the contract and the check order of ggq_dequantize_device apply (d_dst 16-byte aligned).
 * The first valid call on a device looks up where that device holds the staging array (once per
 * device); a call with src_blocks == 0 does only that, e.g. before a graph capture begins.
 * All calls on one GPU share that GPU's staging buffer: calls on the same stream are ordered
 * by the stream; calls that can overlap on one GPU (different streams, or several threads)
 * race on it and must be serialized by the caller. */
#define GGQ_NOALLOC_MAX_BYTES (32u << 20)
ggq_status ggq_dequantize_noalloc(uint32_t ggml_type, uint32_t out_type,
                                  void *d_dst, size_t dst_elems,
                                  const void *h_src, size_t src_blocks,
                                  void *hip_stream);

/* Self-test, no counterpart in the reference.  The quantizers divide twice per block (delta = extreme / c and
 * 1 / delta, ggml-quants/src/structs/q8_0.rs:31-32 and the other formats alike); for blocks of ordinary magnitude
 * the kernels use shorter instruction sequences than the compiler's general IEEE division.  This call runs those
 * sequences against the general division for every float in their range on the current device (about a second) and
 * returns the number of inputs with different bits: mismatches[0..3] for a / 15, a / 31, a / 127, a / -127,
 * mismatches[4] for 1 / d; *tested (optional) = the inputs checked per constant (0 would mean the kernel did not
 * run).  All five must be 0 on any GPU the library quantizes on. */
ggq_status ggq_selftest_fast_division(uint64_t mismatches[5], uint64_t *tested);

#ifdef __cplusplus
}
#endif
#endif /* GGQ_H */
