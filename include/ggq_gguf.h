/*
 * ggq_gguf.h -- C ABI of the GGUF container reader in libggq.
 *
 * Replaces the zero-copy v3 parser of the reference crate `ggus`:
 *   GGuf::new(&[u8])              ggus/src/file.rs:66-145
 *   GGufReader primitives          ggus/src/read.rs:23-91
 *   header                         ggus/src/header.rs:7-57
 *   meta key/value skip-scan       ggus/src/metadata/meta_kv.rs:12-54
 *   tensor meta / info             ggus/src/tensor.rs:191-290
 * and the mmap of input files (xtask/src/utils/mod.rs:42-46).  It feeds the
 * dequantize entry points of ggq.h with (type, shape, pointer) for each tensor.
 * Nothing is copied: every pointer handed out points into the caller's buffer
 * (ggq_gguf_parse) or into the file mapping (ggq_gguf_open) and is valid until
 * ggq_gguf_close.
 *
 * One deliberate difference: a file whose tensor data is shorter than its tensor
 * table says (the reference's own test asset is metadata-only) parses fine here;
 * tensors the file does not hold have data == NULL.  The reference panics on it
 * (slice out of range at file.rs:130-135).
 */
#ifndef GGQ_GGUF_H
#define GGQ_GGUF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GGufError (file.rs:25-40) with GGufReadError (read.rs:14-21) flattened in. */
typedef enum ggq_gguf_error {
    GGQ_GGUF_OK = 0,
    GGQ_GGUF_EOS = 1,              /* Reading(Eos): the data ends inside a field          */
    GGQ_GGUF_UTF8 = 2,             /* Reading(Utf8): a string is not valid UTF-8          */
    GGQ_GGUF_BOOL = 3,             /* Reading(Bool): a bool byte is neither 0 nor 1       */
    GGQ_GGUF_MAGIC = 4,            /* MagicMismatch                                       */
    GGQ_GGUF_ENDIAN = 5,           /* EndianNotSupport                                    */
    GGQ_GGUF_VERSION = 6,          /* VersionNotSupport (only version 3)                  */
    GGQ_GGUF_ALIGNMENT_TYPE = 7,   /* AlignmentTypeMismatch: general.alignment not u32/u64 */
    GGQ_GGUF_DUP_KEY = 8,          /* DuplicateMetaKey                                    */
    GGQ_GGUF_DUP_TENSOR = 9,       /* DuplicateTensorName                                 */
    GGQ_GGUF_IO = 10,              /* open / fstat / mmap failed (ggq_gguf_open only)     */
    GGQ_GGUF_VALUE_TYPE = 11,      /* unknown metadata value type id                      */
    GGQ_GGUF_NOT_FOUND = 12,       /* no such key / tensor / index                        */
    GGQ_GGUF_INVALID = 13,         /* null argument / call out of order                   */
    GGQ_GGUF_INDIVISIBLE = 14,     /* shape[0] is not a whole number of blocks (the reference asserts,
                                      ggus/src/tensor.rs:83-96), or the type has no known size */
    GGQ_GGUF_TENSOR_RANGE = 15     /* a tensor has more than 4 dimensions, or its shape product, byte size
                                      or offset + size wraps 64 bits (the reference's checked arithmetic
                                      panics, ggus/src/file.rs:104-106)                                  */
} ggq_gguf_error;

/* GGufMetaDataValueType (ggus/src/metadata/mod.rs) */
enum {
    GGQ_META_U8 = 0, GGQ_META_I8 = 1, GGQ_META_U16 = 2, GGQ_META_I16 = 3, GGQ_META_U32 = 4,
    GGQ_META_I32 = 5, GGQ_META_F32 = 6, GGQ_META_BOOL = 7, GGQ_META_STRING = 8, GGQ_META_ARRAY = 9,
    GGQ_META_U64 = 10, GGQ_META_I64 = 11, GGQ_META_F64 = 12
};

typedef struct ggq_gguf ggq_gguf;

typedef struct ggq_gguf_info {
    uint32_t version;
    uint64_t tensor_count;
    uint64_t kv_count;
    uint64_t alignment;     /* general.alignment, default 32 (ggus/src/metadata/mod.rs:13) */
    uint64_t data_offset;   /* offset of the tensor data region in the file / buffer       */
    uint64_t data_len;      /* bytes the tensor table needs: max(offset + nbytes)          */
    uint64_t data_avail;    /* bytes actually present after data_offset                    */
} ggq_gguf_info;

typedef struct ggq_gguf_tensor {
    const char *name;       /* not NUL-terminated */
    size_t name_len;
    uint32_t ggml_type;     /* GGmlType discriminant */
    uint32_t ndim;          /* at most 4 dimensions are reported in shape[] */
    uint64_t shape[4];      /* shape[0] is the contiguous dimension */
    uint64_t offset;        /* relative to the data region */
    uint64_t nbytes;        /* GGmlTypeSize::elements_to_bytes(shape); 0 if the type's size is unknown */
    const void *data;       /* NULL when the file does not hold the whole tensor */
} ggq_gguf_tensor;

typedef struct ggq_gguf_kv {
    const char *key;        /* not NUL-terminated */
    size_t key_len;
    uint32_t type;          /* GGQ_META_* */
    const void *value;      /* raw value bytes as stored (strings: u64 length + bytes; arrays: type, count, items) */
    size_t value_len;
} ggq_gguf_kv;

/* GGuf::new over caller memory (which must outlive the handle). */
ggq_gguf_error ggq_gguf_parse(const void *data, size_t len, ggq_gguf **out);
/* Maps the file read-only and parses it. */
ggq_gguf_error ggq_gguf_open(const char *path, ggq_gguf **out);
void ggq_gguf_close(ggq_gguf *g);
/* The offending key / tensor name of the last DUP_KEY / DUP_TENSOR on this thread. */
const char *ggq_gguf_last_name(void);
const char *ggq_gguf_error_str(ggq_gguf_error e);

ggq_gguf_error ggq_gguf_get_info(const ggq_gguf *g, ggq_gguf_info *out);
/* Tensors and key/values keep the file's order (the reference's IndexMap does too). */
ggq_gguf_error ggq_gguf_tensor_at(const ggq_gguf *g, size_t index, ggq_gguf_tensor *out);
ggq_gguf_error ggq_gguf_find_tensor(const ggq_gguf *g, const char *name, ggq_gguf_tensor *out);
ggq_gguf_error ggq_gguf_kv_at(const ggq_gguf *g, size_t index, ggq_gguf_kv *out);
ggq_gguf_error ggq_gguf_find_kv(const ggq_gguf *g, const char *key, ggq_gguf_kv *out);

/* Bytes per block / elements per block of EVERY GGmlType as stored in GGUF files (also the
 * types this library cannot decode); 0 for ids that are not in the format. */
size_t ggq_gguf_type_bytes(uint32_t ggml_type);
size_t ggq_gguf_type_elems(uint32_t ggml_type);

/* ---- writer, simulator, shard planner -----------------------------------------------------
 * Replaces ggus' write side:
 *   GGufWriter              ggus/src/write/writer.rs:10-106   (header, string, key/value, tensor
 *                                                              info, padding, data; byte counter)
 *   GGufFileWriter /
 *   GGufTensorWriter        ggus/src/write/file_writer.rs:39-137
 *   GGufFileSimulator /
 *   GGufTensorSimulator     ggus/src/write/simulator.rs:26-95  (the same calls into a sink that
 *                                                              only counts; used to plan shards)
 *   shard planning          xtask/src/utils/write.rs:21-51
 * One handle covers both phases: key/values first, then tensors, then finish.  A call out of
 * phase returns GGQ_GGUF_INVALID.  With path == NULL nothing is written and only bytes are
 * counted (the simulator). */
typedef struct ggq_gguf_writer ggq_gguf_writer;

/* DataFuture::get (file_writer.rs:26-29).  Called by ggq_gguf_writer_finish, once per tensor that
 * was written without a data pointer, in writing order; returns `nbytes` bytes that stay valid
 * until the next call of it or the end of finish.  This is where a caller hands over a tensor
 * that is decoded on the GPU only when the file reaches it.  NULL aborts with GGQ_GGUF_IO. */
typedef const void *(*ggq_gguf_data_fn)(void *user, uint64_t tensor_index, uint64_t nbytes);

/* GGufFileWriter::new (file_writer.rs:42-49): creates/truncates `path` and writes the header
 * (version 3, the two counts as given -- they are not checked against what follows, as in the
 * reference).  write_data = 0 writes header, key/values and tensor infos only (finish(false),
 * file_writer.rs:83-91, 104-106). */
ggq_gguf_error ggq_gguf_writer_create(const char *path, uint64_t n_tensors, uint64_t n_kv, int write_data,
                                      ggq_gguf_writer **out);
/* write_alignment (writer.rs:49-56): the general.alignment key/value as u32. */
ggq_gguf_error ggq_gguf_writer_alignment(ggq_gguf_writer *w, uint32_t alignment);
/* write_meta_kv (writer.rs:59-77): `value` = the raw value bytes as stored.  The key
 * general.alignment also sets the alignment and must carry exactly 4 bytes
 * (GGQ_GGUF_ALIGNMENT_TYPE otherwise; the reference panics). */
ggq_gguf_error ggq_gguf_writer_kv(ggq_gguf_writer *w, const char *key, size_t key_len, uint32_t type,
                                  const void *value, size_t value_len);
/* write_tensor (file_writer.rs:96-108): pads the running data offset, writes the tensor info
 * and remembers the data.  data == NULL: fn(user, index, nbytes) supplies it during finish
 * (both NULL is allowed when nothing will be written). */
ggq_gguf_error ggq_gguf_writer_tensor(ggq_gguf_writer *w, const char *name, size_t name_len, uint32_t ggml_type,
                                      const uint64_t *shape, uint32_t ndim, const void *data,
                                      ggq_gguf_data_fn fn, void *user);
/* GGufTensorSimulator::written_bytes (simulator.rs:87-94): bytes written so far plus, for every
 * tensor so far, padding to the alignment and its data.  0 for a NULL handle. */
uint64_t ggq_gguf_writer_written_bytes(const ggq_gguf_writer *w);
uint64_t ggq_gguf_writer_alignment_value(const ggq_gguf_writer *w);
/* GGufTensorWriter::finish (file_writer.rs:111-136): padding + data of every tensor, in order;
 * flushes and closes the file; *n_bytes = file size. */
ggq_gguf_error ggq_gguf_writer_finish(ggq_gguf_writer *w, uint64_t *n_bytes);
void ggq_gguf_writer_destroy(ggq_gguf_writer *w);

/* Shard planning of Content::write_files (xtask/src/utils/write.rs:21-51), decisions identical:
 * tensors go to the current shard while it holds fewer than max_tensor_count tensors and the
 * simulated file (first shard: header + alignment + all key/values; later shards: header +
 * alignment only) stays below max_file_size bytes, judged AFTER adding the tensor; otherwise the
 * tensor opens a new shard.  no_tensor_first keeps the first shard free of tensors.
 * Only name, type, ndim and shape of `tensors` and key, type, value of `kvs` are read (the structs
 * the reader fills can be passed as they are).  shard_of[i] = shard index of tensor i;
 * *n_shards >= 1. */
ggq_gguf_error ggq_gguf_plan_shards(uint32_t alignment, const ggq_gguf_kv *kvs, size_t n_kv,
                                    const ggq_gguf_tensor *tensors, size_t n_tensors, uint64_t max_tensor_count,
                                    uint64_t max_file_size, int no_tensor_first, uint32_t *shard_of,
                                    size_t *n_shards);

#ifdef __cplusplus
}
#endif
#endif /* GGQ_GGUF_H */
