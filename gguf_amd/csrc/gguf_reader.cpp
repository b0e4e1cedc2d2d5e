// gguf_reader.cpp -- zero-copy GGUF v3 parser and file mapping behind include/ggq_gguf.h.
// Host code only; linked into libggq.so.  The entry points and the reference code they
// replace are documented in the header.
#include "../../include/ggq_gguf.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_name;

// Cursor over the remaining bytes, like GGufReader (ggus/src/read.rs:23-91).
struct Reader {
    const uint8_t *p;
    size_t left;
    ggq_gguf_error err = GGQ_GGUF_OK;

    bool skip(uint64_t n) {
        if (err) return false;
        if (n > left) { err = GGQ_GGUF_EOS; return false; }
        p += n; left -= (size_t)n;
        return true;
    }
    template <class T> T read() {
        T v{};
        if (err) return v;
        if (sizeof(T) > left) { err = GGQ_GGUF_EOS; return v; }
        std::memcpy(&v, p, sizeof(T));       // unaligned read
        p += sizeof(T); left -= sizeof(T);
        return v;
    }
    // u64 length + bytes, validated as UTF-8 (read_str)
    std::string_view read_str();
    bool skip_value(uint32_t ty, uint64_t count, int depth);
};

// Same acceptance as Rust's core::str::from_utf8: shortest form, no surrogates, <= U+10FFFF.
bool valid_utf8(const uint8_t *s, size_t n) {
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        size_t need;
        uint32_t cp;
        if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
        else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; }
        else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; }
        else return false;
        if (n - i <= need) return false;
        for (size_t k = 1; k <= need; ++k) {
            uint8_t cc = s[i + k];
            if ((cc & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (need == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
        if (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
        i += need + 1;
    }
    return true;
}

std::string_view Reader::read_str() {
    uint64_t n = read<uint64_t>();
    if (err) return {};
    if (n > left) { err = GGQ_GGUF_EOS; return {}; }
    const uint8_t *s = p;
    if (!valid_utf8(s, (size_t)n)) { err = GGQ_GGUF_UTF8; return {}; }
    p += n; left -= (size_t)n;
    return std::string_view((const char *)s, (size_t)n);
}

size_t scalar_size(uint32_t ty) {
    switch (ty) {
    case GGQ_META_U8: case GGQ_META_I8: case GGQ_META_BOOL: return 1;
    case GGQ_META_U16: case GGQ_META_I16: return 2;
    case GGQ_META_U32: case GGQ_META_I32: case GGQ_META_F32: return 4;
    case GGQ_META_U64: case GGQ_META_I64: case GGQ_META_F64: return 8;
    default: return 0;
    }
}

// read_meta_value (meta_kv.rs:24-53): skip `count` values of type ty
bool Reader::skip_value(uint32_t ty, uint64_t count, int depth) {
    if (err) return false;
    switch (ty) {
    case GGQ_META_BOOL:
        for (uint64_t i = 0; i < count; ++i) {
            uint8_t b = read<uint8_t>();
            if (err) return false;
            if (b > 1) { err = GGQ_GGUF_BOOL; return false; }
        }
        return true;
    case GGQ_META_STRING:
        for (uint64_t i = 0; i < count; ++i) {
            read_str();
            if (err) return false;
        }
        return true;
    case GGQ_META_ARRAY: {
        if (depth > 64) { err = GGQ_GGUF_VALUE_TYPE; return false; }
        // the reference reads ONE array header whatever `count` is, then its items
        uint32_t ety = read<uint32_t>();
        uint64_t n = read<uint64_t>();
        if (err) return false;
        return skip_value(ety, n, depth + 1);
    }
    default: {
        size_t sz = scalar_size(ty);
        if (sz == 0) { err = GGQ_GGUF_VALUE_TYPE; return false; }
        if (count > left / sz) { err = GGQ_GGUF_EOS; return false; }
        return skip(count * sz);
    }
    }
}

struct TypeSize { uint32_t id, elems, bytes; };
// Block geometry of every GGmlType as GGUF files store it (ggus/src/tensor.rs:102-144 takes
// size_of the crate's structs; four of those disagree with the files -- IQ4NL, IQ4XS, IQ3XXS
// declare u16 quants, Q8K an f16 scale -- and the file sizes are used here, except Q8K, kept
// at the reference's 290 bytes because its decoder follows the reference struct and the type
// never occurs in files).
const TypeSize kFileTypes[] = {
    {0, 1, 4}, {1, 1, 2}, {2, 32, 18}, {3, 32, 20}, {6, 32, 22}, {7, 32, 24}, {8, 32, 34}, {9, 32, 36},
    {10, 256, 84}, {11, 256, 110}, {12, 256, 144}, {13, 256, 176}, {14, 256, 210}, {15, 256, 290},
    {16, 256, 66}, {17, 256, 74}, {18, 256, 98}, {19, 256, 50}, {20, 32, 18}, {21, 256, 110},
    {22, 256, 82}, {23, 256, 136}, {24, 1, 1}, {25, 1, 2}, {26, 1, 4}, {27, 1, 8}, {28, 1, 8},
    {29, 256, 56}, {30, 1, 2},
};
const TypeSize *file_type(uint32_t id) {
    for (const TypeSize &t : kFileTypes)
        if (t.id == id) return &t;
    return nullptr;
}

struct Tensor {
    std::string_view name;
    uint32_t ty, ndim;
    uint64_t shape[4];
    uint64_t offset, nbytes;
};
struct KV {
    std::string_view key;
    uint32_t ty;
    const uint8_t *value;
    size_t value_len;
};

}  // namespace

struct ggq_gguf {
    const uint8_t *base = nullptr;
    size_t len = 0;
    void *map = nullptr;      // non-null when this handle owns a file mapping
    size_t map_len = 0;
    ggq_gguf_info info{};
    std::vector<KV> kvs;
    std::vector<Tensor> tensors;
    std::unordered_map<std::string_view, size_t> kv_index, tensor_index;
};

namespace {

uint64_t pad_to(uint64_t pos, uint64_t align) {
    return align ? (align - pos % align) % align : 0;
}

ggq_gguf_error parse_into(ggq_gguf *g) {
    Reader r{g->base, g->len};
    // header (header.rs:7-57): magic, version, tensor_count, metadata_kv_count
    if (g->len < 24) return GGQ_GGUF_EOS;
    if (std::memcmp(g->base, "GGUF", 4) != 0) return GGQ_GGUF_MAGIC;
    r.skip(4);
    uint32_t version = r.read<uint32_t>();
    uint64_t n_tensors = r.read<uint64_t>();
    uint64_t n_kv = r.read<uint64_t>();
    // a file of the other endianness shows its version byte-swapped: a small version number
    // then has its low 16 bits clear (header.rs:50-57 compares against the swapped value)
    if (version != 0 && (version & 0xFFFFu) == 0) return GGQ_GGUF_ENDIAN;
    if (version != 3) return GGQ_GGUF_VERSION;

    uint64_t alignment = 32;
    g->kvs.reserve((size_t)std::min<uint64_t>(n_kv, 1u << 16));
    for (uint64_t i = 0; i < n_kv; ++i) {
        std::string_view key = r.read_str();
        uint32_t ty = r.read<uint32_t>();
        const uint8_t *v0 = r.p;
        if (r.err) return r.err;
        Reader vr = r;                      // a copy positioned at the value, for general.alignment
        if (!r.skip_value(ty, 1, 0)) return r.err;
        if (key == "general.alignment") {
            if (ty == GGQ_META_U32) alignment = vr.read<uint32_t>();
            else if (ty == GGQ_META_U64) alignment = vr.read<uint64_t>();
            else { g_name = "general.alignment"; return GGQ_GGUF_ALIGNMENT_TYPE; }
        }
        if (!g->kv_index.emplace(key, g->kvs.size()).second) { g_name = std::string(key); return GGQ_GGUF_DUP_KEY; }
        g->kvs.push_back({key, ty, v0, (size_t)(r.p - v0)});
    }

    uint64_t data_len = 0;
    g->tensors.reserve((size_t)std::min<uint64_t>(n_tensors, 1u << 20));
    for (uint64_t i = 0; i < n_tensors; ++i) {
        Tensor t{};
        t.name = r.read_str();
        t.ndim = r.read<uint32_t>();
        if (r.err) return r.err;
        if ((uint64_t)t.ndim > r.left / 8) return GGQ_GGUF_EOS;
        // ggml tensors have at most 4 dimensions, and the C view of a tensor holds 4
        if (t.ndim > 4) { g_name = std::string(t.name); return GGQ_GGUF_TENSOR_RANGE; }
        uint64_t others = 1, first = 1;
        bool wraps = false;
        for (uint32_t d = 0; d < t.ndim; ++d) {
            uint64_t s = r.read<uint64_t>();
            t.shape[d] = s;
            if (d == 0) first = s;
            else wraps |= __builtin_mul_overflow(others, s, &others);
        }
        t.ty = r.read<uint32_t>();
        t.offset = r.read<uint64_t>();
        if (r.err) return r.err;
        // GGmlTypeSize::elements_to_bytes (tensor.rs:83-96).  The numbers come from the file:
        // where the reference asserts (shape[0] % block) or its arithmetic would wrap, the
        // parse fails instead of handing out geometry that does not match the bytes.
        const TypeSize *ts = file_type(t.ty);
        if (!ts) { g_name = std::string(t.name); return GGQ_GGUF_INDIVISIBLE; }
        if (t.ndim == 0) {
            if (ts->elems != 1) { g_name = std::string(t.name); return GGQ_GGUF_INDIVISIBLE; }
            t.nbytes = ts->bytes;
        } else {
            if (first % ts->elems != 0) { g_name = std::string(t.name); return GGQ_GGUF_INDIVISIBLE; }
            uint64_t blocks = 0;
            wraps |= __builtin_mul_overflow(others, first / ts->elems, &blocks);
            wraps |= __builtin_mul_overflow(blocks, (uint64_t)ts->bytes, &t.nbytes);
        }
        uint64_t end = 0;
        wraps |= __builtin_add_overflow(t.offset, t.nbytes, &end);
        if (wraps) { g_name = std::string(t.name); return GGQ_GGUF_TENSOR_RANGE; }
        data_len = std::max(data_len, end);
        if (!g->tensor_index.emplace(t.name, g->tensors.size()).second) { g_name = std::string(t.name); return GGQ_GGUF_DUP_TENSOR; }
        g->tensors.push_back(t);
    }

    uint64_t cursor = g->len - r.left;
    uint64_t padding = g->tensors.empty() ? 0 : pad_to(cursor, alignment);
    // The reference skips the padding with a checked read; a metadata-only file ends before
    // it.  Tolerated here: the data region is then simply empty.
    uint64_t data_offset = std::min<uint64_t>(cursor + padding, g->len);
    g->info = {version, n_tensors, n_kv, alignment, data_offset, data_len, g->len - data_offset};
    return GGQ_GGUF_OK;
}

void fill_tensor(const ggq_gguf *g, const Tensor &t, ggq_gguf_tensor *out) {
    out->name = t.name.data();
    out->name_len = t.name.size();
    out->ggml_type = t.ty;
    out->ndim = t.ndim;
    std::memcpy(out->shape, t.shape, sizeof t.shape);
    out->offset = t.offset;
    out->nbytes = t.nbytes;
    // offset + nbytes cannot wrap: parse_into refused such tensors
    const bool held = t.nbytes > 0 && t.offset <= g->info.data_avail && t.nbytes <= g->info.data_avail - t.offset;
    out->data = held ? g->base + g->info.data_offset + t.offset : nullptr;
}

void fill_kv(const KV &kv, ggq_gguf_kv *out) {
    out->key = kv.key.data();
    out->key_len = kv.key.size();
    out->type = kv.ty;
    out->value = kv.value;
    out->value_len = kv.value_len;
}

}  // namespace

extern "C" {

ggq_gguf_error ggq_gguf_parse(const void *data, size_t len, ggq_gguf **out) {
    if (!out || (!data && len)) return GGQ_GGUF_INVALID;
    ggq_gguf *g = new (std::nothrow) ggq_gguf();
    if (!g) return GGQ_GGUF_IO;
    g->base = (const uint8_t *)data;
    g->len = len;
    ggq_gguf_error e = parse_into(g);
    if (e != GGQ_GGUF_OK) { delete g; return e; }
    *out = g;
    return GGQ_GGUF_OK;
}

ggq_gguf_error ggq_gguf_open(const char *path, ggq_gguf **out) {
    if (!out || !path) return GGQ_GGUF_INVALID;
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return GGQ_GGUF_IO;
    struct stat st{};
    if (::fstat(fd, &st) != 0) { ::close(fd); return GGQ_GGUF_IO; }
    if (st.st_size <= 0) { ::close(fd); return st.st_size == 0 ? GGQ_GGUF_EOS : GGQ_GGUF_IO; }
    void *m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return GGQ_GGUF_IO;
    ggq_gguf *g = new (std::nothrow) ggq_gguf();
    if (!g) { ::munmap(m, (size_t)st.st_size); return GGQ_GGUF_IO; }
    g->base = (const uint8_t *)m;
    g->len = (size_t)st.st_size;
    g->map = m;
    g->map_len = (size_t)st.st_size;
    ggq_gguf_error e = parse_into(g);
    if (e != GGQ_GGUF_OK) { ggq_gguf_close(g); return e; }
    *out = g;
    return GGQ_GGUF_OK;
}

void ggq_gguf_close(ggq_gguf *g) {
    if (!g) return;
    if (g->map) ::munmap(g->map, g->map_len);
    delete g;
}

const char *ggq_gguf_last_name(void) { return g_name.c_str(); }

const char *ggq_gguf_error_str(ggq_gguf_error e) {
    switch (e) {
    case GGQ_GGUF_OK: return "ok";
    case GGQ_GGUF_EOS: return "reading error: Eos";
    case GGQ_GGUF_UTF8: return "reading error: Utf8";
    case GGQ_GGUF_BOOL: return "reading error: Bool";
    case GGQ_GGUF_MAGIC: return "magic mismatch";
    case GGQ_GGUF_ENDIAN: return "endian not support";
    case GGQ_GGUF_VERSION: return "version not support";
    case GGQ_GGUF_ALIGNMENT_TYPE: return "alignment type mismatch";
    case GGQ_GGUF_DUP_KEY: return "duplicate meta key";
    case GGQ_GGUF_DUP_TENSOR: return "duplicate tensor name";
    case GGQ_GGUF_IO: return "cannot open or map the file";
    case GGQ_GGUF_VALUE_TYPE: return "unknown metadata value type";
    case GGQ_GGUF_NOT_FOUND: return "not found";
    case GGQ_GGUF_INVALID: return "invalid argument";
    case GGQ_GGUF_INDIVISIBLE: return "shape is not a whole number of blocks of the type";
    case GGQ_GGUF_TENSOR_RANGE: return "tensor geometry out of range (more than 4 dimensions, or sizes that overflow 64 bits)";
    }
    return "unknown error";
}

ggq_gguf_error ggq_gguf_get_info(const ggq_gguf *g, ggq_gguf_info *out) {
    if (!g || !out) return GGQ_GGUF_INVALID;
    *out = g->info;
    return GGQ_GGUF_OK;
}

ggq_gguf_error ggq_gguf_tensor_at(const ggq_gguf *g, size_t index, ggq_gguf_tensor *out) {
    if (!g || !out) return GGQ_GGUF_INVALID;
    if (index >= g->tensors.size()) return GGQ_GGUF_NOT_FOUND;
    fill_tensor(g, g->tensors[index], out);
    return GGQ_GGUF_OK;
}

ggq_gguf_error ggq_gguf_find_tensor(const ggq_gguf *g, const char *name, ggq_gguf_tensor *out) {
    if (!g || !out || !name) return GGQ_GGUF_INVALID;
    auto it = g->tensor_index.find(std::string_view(name));
    if (it == g->tensor_index.end()) return GGQ_GGUF_NOT_FOUND;
    fill_tensor(g, g->tensors[it->second], out);
    return GGQ_GGUF_OK;
}

ggq_gguf_error ggq_gguf_kv_at(const ggq_gguf *g, size_t index, ggq_gguf_kv *out) {
    if (!g || !out) return GGQ_GGUF_INVALID;
    if (index >= g->kvs.size()) return GGQ_GGUF_NOT_FOUND;
    fill_kv(g->kvs[index], out);
    return GGQ_GGUF_OK;
}

ggq_gguf_error ggq_gguf_find_kv(const ggq_gguf *g, const char *key, ggq_gguf_kv *out) {
    if (!g || !out || !key) return GGQ_GGUF_INVALID;
    auto it = g->kv_index.find(std::string_view(key));
    if (it == g->kv_index.end()) return GGQ_GGUF_NOT_FOUND;
    fill_kv(g->kvs[it->second], out);
    return GGQ_GGUF_OK;
}

size_t ggq_gguf_type_bytes(uint32_t ty) {
    const TypeSize *t = file_type(ty);
    return t ? t->bytes : 0;
}

size_t ggq_gguf_type_elems(uint32_t ty) {
    const TypeSize *t = file_type(ty);
    return t ? t->elems : 0;
}

}  // extern "C"
