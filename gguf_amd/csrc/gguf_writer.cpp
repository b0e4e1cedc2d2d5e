// gguf_writer.cpp -- GGUF v3 writer, byte-counting simulator and shard planner behind the C ABI
// of include/ggq_gguf.h.  Host code only.
//
// Layout written (ggus/src/write/writer.rs:26-105): 24-byte header {"GGUF", u32 version, u64
// tensor count, u64 key/value count}; per key/value: u64 length + key bytes, u32 type, raw value;
// per tensor: u64 length + name, u32 ndim, ndim x u64 shape, u32 type, u64 offset; then for each
// tensor zero padding to the alignment followed by its bytes.  All little-endian (the host is).
#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/ggq_gguf.h"

namespace {

constexpr uint64_t kDefaultAlignment = 32;   // ggus/src/metadata/mod.rs:13
constexpr std::string_view kAlignmentKey = "general.alignment";

uint64_t pad_to(uint64_t pos, uint64_t align) { return align ? (align - pos % align) % align : 0; }

struct PendingData {
    const void *data;
    ggq_gguf_data_fn fn;
    void *user;
    uint64_t nbytes;
};

}  // namespace

struct ggq_gguf_writer {
    int fd = -1;                    // -1: count only
    bool write_data = true;
    bool tensors_started = false;
    bool finished = false;
    uint64_t written = 0;           // bytes handed to the sink so far
    uint64_t alignment = kDefaultAlignment;
    uint64_t offset = 0;            // running offset inside the data region
    std::vector<PendingData> data;
    std::vector<uint64_t> lens;     // every tensor's byte count, also when data is not kept
    std::vector<uint8_t> buf;       // metadata is small: buffered, flushed before the data
    bool io_error = false;

    void put(const void *p, size_t n) {
        written += n;
        if (fd < 0 || n == 0) return;
        const uint8_t *b = (const uint8_t *)p;
        buf.insert(buf.end(), b, b + n);
        if (buf.size() >= (1u << 20)) flush();
    }
    template <class T>
    void put_val(T v) { put(&v, sizeof v); }
    void put_str(const char *s, size_t n) {
        put_val<uint64_t>(n);
        put(s, n);
    }
    void raw(const uint8_t *p, size_t n) {
        while (n && !io_error) {
            ssize_t k = ::write(fd, p, n > (1u << 30) ? (1u << 30) : n);
            if (k < 0) {
                if (errno == EINTR) continue;
                io_error = true;
                return;
            }
            p += k;
            n -= (size_t)k;
        }
    }
    void flush() {
        if (fd >= 0 && !buf.empty()) raw(buf.data(), buf.size());
        buf.clear();
    }
    void zeros(uint64_t n) {
        static const uint8_t z[256] = {0};
        while (n) {
            const size_t k = n > sizeof z ? sizeof z : (size_t)n;
            put(z, k);
            n -= k;
        }
    }
};

namespace {

ggq_gguf_error add_kv(ggq_gguf_writer *w, const char *key, size_t key_len, uint32_t type, const void *value,
                      size_t value_len) {
    if (std::string_view(key, key_len) == kAlignmentKey && value_len != 4) return GGQ_GGUF_ALIGNMENT_TYPE;
    w->put_str(key, key_len);
    w->put_val<uint32_t>(type);
    w->put(value, value_len);
    if (std::string_view(key, key_len) == kAlignmentKey) {
        uint32_t a;
        std::memcpy(&a, value, 4);
        w->alignment = a;
    }
    return GGQ_GGUF_OK;
}

ggq_gguf_error tensor_bytes(uint32_t ty, const uint64_t *shape, uint32_t ndim, uint64_t *out) {
    // GGmlTypeSize::elements_to_bytes (ggus/src/tensor.rs:83-96); an empty shape is one element
    const uint64_t be = ggq_gguf_type_elems(ty), bb = ggq_gguf_type_bytes(ty);
    if (be == 0) return GGQ_GGUF_INDIVISIBLE;
    if (ndim == 0) {
        if (be != 1) return GGQ_GGUF_INDIVISIBLE;
        *out = bb;
        return GGQ_GGUF_OK;
    }
    if (shape[0] % be != 0) return GGQ_GGUF_INDIVISIBLE;
    uint64_t n = 0;
    bool wraps = __builtin_mul_overflow(shape[0] / be, bb, &n);
    for (uint32_t i = 1; i < ndim; ++i) wraps |= __builtin_mul_overflow(n, shape[i], &n);
    if (wraps) return GGQ_GGUF_TENSOR_RANGE;      // the size does not fit 64 bits
    *out = n;
    return GGQ_GGUF_OK;
}

ggq_gguf_error add_tensor(ggq_gguf_writer *w, const char *name, size_t name_len, uint32_t ty, const uint64_t *shape,
                          uint32_t ndim, const void *data, ggq_gguf_data_fn fn, void *user) {
    uint64_t len = 0;
    ggq_gguf_error e = tensor_bytes(ty, shape, ndim, &len);
    if (e != GGQ_GGUF_OK) return e;
    w->tensors_started = true;
    w->offset += pad_to(w->offset, w->alignment);
    w->put_str(name, name_len);
    w->put_val<uint32_t>(ndim);
    w->put(shape, sizeof(uint64_t) * ndim);
    w->put_val<uint32_t>(ty);
    w->put_val<uint64_t>(w->offset);
    w->offset += len;
    w->lens.push_back(len);
    if (w->write_data) w->data.push_back({data, fn, user, len});
    return GGQ_GGUF_OK;
}

uint64_t simulated_bytes(const ggq_gguf_writer *w) {
    uint64_t total = w->written;
    for (uint64_t len : w->lens) {
        total += pad_to(total, w->alignment);
        total += len;
    }
    return total;
}

void start_simulator(ggq_gguf_writer *w, uint32_t alignment) {
    // GGufFileSimulator::with_alignment (simulator.rs:29-44): a header, then the alignment key/value
    *w = ggq_gguf_writer();
    w->written = 24;
    const uint32_t a = alignment;
    add_kv(w, kAlignmentKey.data(), kAlignmentKey.size(), GGQ_META_U32, &a, 4);
}

}  // namespace

extern "C" {

ggq_gguf_error ggq_gguf_writer_create(const char *path, uint64_t n_tensors, uint64_t n_kv, int write_data,
                                      ggq_gguf_writer **out) {
    if (!out) return GGQ_GGUF_INVALID;
    *out = nullptr;
    int fd = -1;
    if (path) {
        fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
        if (fd < 0) return GGQ_GGUF_IO;
    }
    ggq_gguf_writer *w = new ggq_gguf_writer();
    w->fd = fd;
    w->write_data = write_data != 0;
    w->put("GGUF", 4);
    w->put_val<uint32_t>(3);
    w->put_val<uint64_t>(n_tensors);
    w->put_val<uint64_t>(n_kv);
    *out = w;
    return GGQ_GGUF_OK;
}

ggq_gguf_error ggq_gguf_writer_alignment(ggq_gguf_writer *w, uint32_t alignment) {
    if (!w || w->tensors_started || w->finished) return GGQ_GGUF_INVALID;
    return add_kv(w, kAlignmentKey.data(), kAlignmentKey.size(), GGQ_META_U32, &alignment, 4);
}

ggq_gguf_error ggq_gguf_writer_kv(ggq_gguf_writer *w, const char *key, size_t key_len, uint32_t type,
                                  const void *value, size_t value_len) {
    if (!w || !key || (!value && value_len) || w->tensors_started || w->finished) return GGQ_GGUF_INVALID;
    return add_kv(w, key, key_len, type, value, value_len);
}

ggq_gguf_error ggq_gguf_writer_tensor(ggq_gguf_writer *w, const char *name, size_t name_len, uint32_t ggml_type,
                                      const uint64_t *shape, uint32_t ndim, const void *data, ggq_gguf_data_fn fn,
                                      void *user) {
    if (!w || !name || (!shape && ndim) || w->finished) return GGQ_GGUF_INVALID;
    return add_tensor(w, name, name_len, ggml_type, shape, ndim, data, fn, user);
}

uint64_t ggq_gguf_writer_written_bytes(const ggq_gguf_writer *w) { return w ? simulated_bytes(w) : 0; }

uint64_t ggq_gguf_writer_alignment_value(const ggq_gguf_writer *w) { return w ? w->alignment : 0; }

ggq_gguf_error ggq_gguf_writer_finish(ggq_gguf_writer *w, uint64_t *n_bytes) {
    if (!w || w->finished) return GGQ_GGUF_INVALID;
    w->finished = true;
    ggq_gguf_error result = GGQ_GGUF_OK;
    uint64_t index = 0;
    for (const PendingData &d : w->data) {
        const void *p = d.data;
        if (!p && d.nbytes && w->fd >= 0) {
            p = d.fn ? d.fn(d.user, index, d.nbytes) : nullptr;
            if (!p) {
                result = GGQ_GGUF_IO;
                break;
            }
        }
        w->zeros(pad_to(w->written, w->alignment));
        if (w->fd >= 0) {
            w->flush();
            w->raw((const uint8_t *)p, d.nbytes);
        }
        w->written += d.nbytes;
        ++index;
    }
    w->flush();
    if (w->fd >= 0) {
        if (::close(w->fd) != 0) w->io_error = true;
        w->fd = -1;
    }
    w->data.clear();
    w->lens.clear();   // their bytes are part of `written` now
    if (n_bytes) *n_bytes = w->written;
    if (w->io_error) return GGQ_GGUF_IO;
    return result;
}

void ggq_gguf_writer_destroy(ggq_gguf_writer *w) {
    if (!w) return;
    if (w->fd >= 0) ::close(w->fd);
    delete w;
}

ggq_gguf_error ggq_gguf_plan_shards(uint32_t alignment, const ggq_gguf_kv *kvs, size_t n_kv,
                                    const ggq_gguf_tensor *tensors, size_t n_tensors, uint64_t max_tensor_count,
                                    uint64_t max_file_size, int no_tensor_first, uint32_t *shard_of,
                                    size_t *n_shards) {
    if ((!kvs && n_kv) || (!tensors && n_tensors) || (!shard_of && n_tensors) || !n_shards) return GGQ_GGUF_INVALID;
    ggq_gguf_writer sim;
    start_simulator(&sim, alignment);
    for (size_t i = 0; i < n_kv; ++i) {
        ggq_gguf_error e = add_kv(&sim, kvs[i].key, kvs[i].key_len, kvs[i].type, kvs[i].value, kvs[i].value_len);
        if (e != GGQ_GGUF_OK) return e;
    }
    sim.write_data = false;
    uint32_t shard = 0;
    uint64_t in_shard = 0;
    for (size_t i = 0; i < n_tensors; ++i) {
        const ggq_gguf_tensor &t = tensors[i];
        const uint32_t nd = t.ndim > 4 ? 4 : t.ndim;
        bool fresh = shard == 0 && no_tensor_first;
        if (!fresh) {
            ggq_gguf_error e = add_tensor(&sim, t.name, t.name_len, t.ggml_type, t.shape, nd, nullptr, nullptr, nullptr);
            if (e != GGQ_GGUF_OK) return e;
            fresh = !(in_shard < max_tensor_count && simulated_bytes(&sim) < max_file_size);
        }
        if (fresh) {
            start_simulator(&sim, alignment);
            sim.write_data = false;
            ggq_gguf_error e = add_tensor(&sim, t.name, t.name_len, t.ggml_type, t.shape, nd, nullptr, nullptr, nullptr);
            if (e != GGQ_GGUF_OK) return e;
            ++shard;
            in_shard = 0;
        }
        shard_of[i] = shard;
        ++in_shard;
    }
    *n_shards = (size_t)shard + 1;
    return GGQ_GGUF_OK;
}

}  // extern "C"
