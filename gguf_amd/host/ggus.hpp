// ggus.hpp -- C++ host mirror of the reference crate `ggus` (GGUF container) over the C ABI of
// include/ggq_gguf.h.  Names and error behaviour follow the reference:
//   GGuf                      ggus/src/file.rs:11-23, 66-145     (zero-copy view of a file or buffer)
//   GGufError                 ggus/src/file.rs:25-40
//   GGufFileWriter /
//   GGufTensorWriter          ggus/src/write/file_writer.rs:39-137
//   GGufFileSimulator /
//   GGufTensorSimulator       ggus/src/write/simulator.rs:26-95
//   plan_shards               xtask/src/utils/write.rs:21-51
// Errors are returned the way the reference returns them (a value, here std::optional /
// an `error` member), never thrown, except for misuse of a moved-from object.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <optional>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../../include/ggq_gguf.h"

namespace ggus {

constexpr uint64_t DEFAULT_ALIGNMENT = 32;   // ggus/src/metadata/mod.rs:13

enum class GGufMetaDataValueType : uint32_t {
    U8 = 0, I8, U16, I16, U32, I32, F32, Bool, String, Array, U64, I64, F64
};

// GGufError with the offending name where the reference carries one
struct GGufError {
    ggq_gguf_error code;
    std::string name;   // DuplicateMetaKey / DuplicateTensorName / AlignmentTypeMismatch
    std::string to_string() const {
        std::string s = ggq_gguf_error_str(code);
        return name.empty() ? s : s + ": " + name;
    }
};

struct GGufMetaKV {
    std::string_view key;
    GGufMetaDataValueType ty;
    std::string_view value_bytes;   // as stored
};

struct GGufTensorInfo {
    std::string_view name;
    uint32_t ty;                    // GGmlType discriminant
    std::vector<uint64_t> shape;
    uint64_t offset;
    uint64_t nbytes;
    const uint8_t *data;            // null when the file holds no data for the tensor
};

class GGuf {
    ggq_gguf *h_ = nullptr;
    ggq_gguf_info info_{};
    explicit GGuf(ggq_gguf *h) : h_(h) { ggq_gguf_get_info(h_, &info_); }
    static GGufError fail(ggq_gguf_error e) {
        const bool named = e == GGQ_GGUF_DUP_KEY || e == GGQ_GGUF_DUP_TENSOR || e == GGQ_GGUF_ALIGNMENT_TYPE;
        return {e, named ? ggq_gguf_last_name() : ""};
    }

public:
    struct Result;   // GGuf or GGufError
    // GGuf::new(data): the buffer must outlive the object
    static Result parse(const void *data, size_t len);
    static Result open(const char *path);
    GGuf(GGuf &&o) noexcept : h_(o.h_), info_(o.info_) { o.h_ = nullptr; }
    GGuf &operator=(GGuf &&o) noexcept {
        if (this != &o) {
            ggq_gguf_close(h_);
            h_ = o.h_;
            info_ = o.info_;
            o.h_ = nullptr;
        }
        return *this;
    }
    GGuf(const GGuf &) = delete;
    GGuf &operator=(const GGuf &) = delete;
    ~GGuf() { ggq_gguf_close(h_); }

    uint32_t version() const { return info_.version; }
    uint64_t tensor_count() const { return info_.tensor_count; }
    uint64_t metadata_kv_count() const { return info_.kv_count; }
    uint64_t alignment() const { return info_.alignment; }
    uint64_t data_offset() const { return info_.data_offset; }

    GGufTensorInfo tensor_at(size_t i) const {
        ggq_gguf_tensor t{};
        ggq_gguf_tensor_at(h_, i, &t);
        return convert(t);
    }
    std::optional<GGufTensorInfo> tensor(const char *name) const {
        ggq_gguf_tensor t{};
        if (ggq_gguf_find_tensor(h_, name, &t) != GGQ_GGUF_OK) return std::nullopt;
        return convert(t);
    }
    GGufMetaKV meta_kv_at(size_t i) const {
        ggq_gguf_kv kv{};
        ggq_gguf_kv_at(h_, i, &kv);
        return convert(kv);
    }
    // GGufMetaMap::get (file.rs:58-62)
    std::optional<GGufMetaKV> get(const char *key) const {
        ggq_gguf_kv kv{};
        if (ggq_gguf_find_kv(h_, key, &kv) != GGQ_GGUF_OK) return std::nullopt;
        return convert(kv);
    }

private:
    static GGufTensorInfo convert(const ggq_gguf_tensor &t) {
        GGufTensorInfo o{{t.name, t.name_len}, t.ggml_type, {}, t.offset, t.nbytes, (const uint8_t *)t.data};
        o.shape.assign(t.shape, t.shape + (t.ndim < 4 ? t.ndim : 4));
        return o;
    }
    static GGufMetaKV convert(const ggq_gguf_kv &kv) {
        return {{kv.key, kv.key_len}, (GGufMetaDataValueType)kv.type, {(const char *)kv.value, kv.value_len}};
    }
};

struct GGuf::Result {
    std::optional<GGuf> ok;
    std::optional<GGufError> err;
    explicit operator bool() const { return ok.has_value(); }
};

inline GGuf::Result GGuf::parse(const void *data, size_t len) {
    ggq_gguf *h = nullptr;
    ggq_gguf_error e = ggq_gguf_parse(data, len, &h);
    if (e != GGQ_GGUF_OK) return {std::nullopt, fail(e)};
    return {GGuf(h), std::nullopt};
}

inline GGuf::Result GGuf::open(const char *path) {
    ggq_gguf *h = nullptr;
    ggq_gguf_error e = ggq_gguf_open(path, &h);
    if (e != GGQ_GGUF_OK) return {std::nullopt, fail(e)};
    return {GGuf(h), std::nullopt};
}

// ---- writing ----------------------------------------------------------------------------------

// DataFuture (file_writer.rs:26-29): bytes that may be produced only when the file reaches them
using DataFuture = std::function<const void *()>;

class GGufTensorWriter {
    friend class GGufFileWriter;
    ggq_gguf_writer *w_;
    std::vector<DataFuture *> lazies_;
    uint64_t offset_ = 0, alignment_;
    explicit GGufTensorWriter(ggq_gguf_writer *w) : w_(w), alignment_(ggq_gguf_writer_alignment_value(w)) {}
    static const void *thunk(void *user, uint64_t, uint64_t) { return (*(DataFuture *)user)(); }

public:
    GGufTensorWriter(GGufTensorWriter &&o) noexcept
        : w_(o.w_), lazies_(std::move(o.lazies_)), offset_(o.offset_), alignment_(o.alignment_) { o.w_ = nullptr; }
    GGufTensorWriter(const GGufTensorWriter &) = delete;
    ~GGufTensorWriter() {
        ggq_gguf_writer_destroy(w_);
        for (DataFuture *f : lazies_) delete f;
    }
    uint64_t alignment() const { return alignment_; }
    uint64_t offset() const { return offset_; }   // end of the last tensor inside the data region

    // write_tensor (file_writer.rs:96-108); data stays owned by the caller until finish()
    ggq_gguf_error write_tensor(std::string_view name, uint32_t ty, const std::vector<uint64_t> &shape,
                                const void *data) {
        return track(ggq_gguf_writer_tensor(w_, name.data(), name.size(), ty, shape.data(), (uint32_t)shape.size(),
                                            data, nullptr, nullptr), ty, shape);
    }
    ggq_gguf_error write_tensor(std::string_view name, uint32_t ty, const std::vector<uint64_t> &shape,
                                DataFuture data) {
        lazies_.push_back(new DataFuture(std::move(data)));
        return track(ggq_gguf_writer_tensor(w_, name.data(), name.size(), ty, shape.data(), (uint32_t)shape.size(),
                                            nullptr, &thunk, lazies_.back()), ty, shape);
    }
    // GGufTensorSimulator::written_bytes (simulator.rs:87-94)
    uint64_t written_bytes() const { return ggq_gguf_writer_written_bytes(w_); }
    // finish (file_writer.rs:111-136): total bytes written, or the error
    std::pair<ggq_gguf_error, uint64_t> finish() {
        uint64_t n = 0;
        ggq_gguf_error e = ggq_gguf_writer_finish(w_, &n);
        return {e, n};
    }

private:
    ggq_gguf_error track(ggq_gguf_error e, uint32_t ty, const std::vector<uint64_t> &shape) {
        if (e != GGQ_GGUF_OK) return e;
        uint64_t n = shape.empty() ? 1 : shape[0] / ggq_gguf_type_elems(ty);
        for (size_t i = 1; i < shape.size(); ++i) n *= shape[i];
        offset_ += (alignment_ - offset_ % alignment_) % alignment_;
        offset_ += n * ggq_gguf_type_bytes(ty);
        return e;
    }
};

// GGufFileWriter when given a path, GGufFileSimulator when not (same calls, bytes only counted)
class GGufFileWriter {
    ggq_gguf_writer *w_ = nullptr;

public:
    ggq_gguf_error error = GGQ_GGUF_OK;   // set by the constructor when the file cannot be created
    // GGufFileWriter::new(File::create(path), GGufFileHeader::new(3, n_tensors, n_kv))
    GGufFileWriter(const char *path, uint64_t n_tensors, uint64_t n_kv, bool write_data = true) {
        error = ggq_gguf_writer_create(path, n_tensors, n_kv, write_data, &w_);
    }
    GGufFileWriter(const GGufFileWriter &) = delete;
    ~GGufFileWriter() { ggq_gguf_writer_destroy(w_); }
    uint64_t alignment() const { return ggq_gguf_writer_alignment_value(w_); }
    ggq_gguf_error write_alignment(uint32_t alignment) { return ggq_gguf_writer_alignment(w_, alignment); }
    ggq_gguf_error write_meta_kv(std::string_view key, GGufMetaDataValueType ty, std::string_view value) {
        return ggq_gguf_writer_kv(w_, key.data(), key.size(), (uint32_t)ty, value.data(), value.size());
    }
    uint64_t written_bytes() const { return ggq_gguf_writer_written_bytes(w_); }
    // finish(write_data) -> GGufTensorWriter (file_writer.rs:83-91); this object is spent afterwards
    GGufTensorWriter finish() {
        ggq_gguf_writer *w = w_;
        w_ = nullptr;
        return GGufTensorWriter(w);
    }
};

// GGufFileSimulator::new / with_alignment (simulator.rs:29-44)
inline GGufFileWriter simulator() { return GGufFileWriter(nullptr, 0, 0); }

struct ShardPlan {
    ggq_gguf_error error;
    std::vector<uint32_t> shard_of;   // per tensor
    size_t n_shards;
};

struct PlanTensor {
    std::string name;
    uint32_t ty;
    std::vector<uint64_t> shape;
};
struct PlanKV {
    std::string key;
    GGufMetaDataValueType ty;
    std::string value;
};

// the planning loop of Content::write_files (xtask/src/utils/write.rs:21-51)
inline ShardPlan plan_shards(uint32_t alignment, const std::vector<PlanKV> &kvs, const std::vector<PlanTensor> &tensors,
                             uint64_t max_tensor_count, uint64_t max_file_size, bool no_tensor_first) {
    std::vector<ggq_gguf_kv> k(kvs.size());
    for (size_t i = 0; i < kvs.size(); ++i)
        k[i] = {kvs[i].key.data(), kvs[i].key.size(), (uint32_t)kvs[i].ty, kvs[i].value.data(), kvs[i].value.size()};
    std::vector<ggq_gguf_tensor> t(tensors.size());
    for (size_t i = 0; i < tensors.size(); ++i) {
        t[i] = {};
        t[i].name = tensors[i].name.data();
        t[i].name_len = tensors[i].name.size();
        t[i].ggml_type = tensors[i].ty;
        t[i].ndim = (uint32_t)tensors[i].shape.size();
        for (size_t d = 0; d < tensors[i].shape.size() && d < 4; ++d) t[i].shape[d] = tensors[i].shape[d];
    }
    ShardPlan p{GGQ_GGUF_OK, std::vector<uint32_t>(tensors.size()), 0};
    p.error = ggq_gguf_plan_shards(alignment, k.data(), k.size(), t.data(), t.size(), max_tensor_count, max_file_size,
                                   no_tensor_first, p.shard_of.data(), &p.n_shards);
    return p;
}

}  // namespace ggus
