// Mutation fuzzing of the GGUF parser, meant to be built with the host sanitizers
// (make -C gguf_amd/host asan): a valid file is written with the library's writer, then bytes
// are flipped, inserted, dropped and the file truncated; the parser must return a code and every
// pointer it hands out must be readable.  Host code only.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include "ggus.hpp"

using namespace ggus;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char **argv) {
    const long iters = argc > 1 ? std::atol(argv[1]) : 20000;
    char path[] = "/tmp/ggq_fuzz_XXXXXX";
    int fd = mkstemp(path);
    if (fd < 0) return 2;
    close(fd);
    std::vector<uint8_t> d1(64 * 8 * 2, 1), d2(4 * 144, 2), d3(256, 3);
    {
        GGufFileWriter w(path, 3, 4);
        auto s = [](const std::string &v) { uint64_t n = v.size(); return std::string((const char *)&n, 8) + v; };
        uint32_t one = 1, a64 = 64, ety = 8;
        uint64_t cnt = 2;
        w.write_meta_kv("general.architecture", GGufMetaDataValueType::String, s("llama"));
        w.write_meta_kv("llama.block_count", GGufMetaDataValueType::U32, std::string((const char *)&one, 4));
        w.write_meta_kv("general.alignment", GGufMetaDataValueType::U32, std::string((const char *)&a64, 4));
        w.write_meta_kv("tokens", GGufMetaDataValueType::Array,
                        std::string((const char *)&ety, 4) + std::string((const char *)&cnt, 8) + s("ab") + s("cde"));
        GGufTensorWriter t = w.finish();
        t.write_tensor("token_embd.weight", 1, {64, 8}, d1.data());
        t.write_tensor("blk.0.attn_q.weight", 12, {256, 4}, d2.data());
        t.write_tensor("blk.0.norm.weight", 0, {64}, d3.data());
        if (t.finish().first != GGQ_GGUF_OK) return 2;
    }
    std::vector<uint8_t> good;
    {
        FILE *f = std::fopen(path, "rb");
        uint8_t buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) good.insert(good.end(), buf, buf + n);
        std::fclose(f);
        unlink(path);
    }
    size_t meta_len = 0;
    {
        auto r = GGuf::parse(good.data(), good.size());
        if (!r) return 2;
        meta_len = r.ok->data_offset();
    }
    long ok = 0, bad = 0;
    uint64_t sink = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> b = good;
        const int edits = 1 + (int)(rnd() % 4);
        for (int e = 0; e < edits; ++e) {
            const size_t pos = rnd() % (b.size() < meta_len ? b.size() : meta_len);
            switch (rnd() % 5) {
            case 0: b[pos] = (uint8_t)rnd(); break;
            case 1: b[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 2: b.insert(b.begin() + pos, (uint8_t)rnd()); break;
            case 3: if (b.size() > 1) b.erase(b.begin() + pos); break;
            case 4: b.resize(rnd() % (b.size() + 1)); break;
            }
            if (b.empty()) break;
        }
        // exact-size heap copy so the sanitizer sees any read past the end
        uint8_t *exact = (uint8_t *)std::malloc(b.size() ? b.size() : 1);
        std::memcpy(exact, b.data(), b.size());
        auto r = GGuf::parse(exact, b.size());
        if (r) {
            ++ok;
            GGuf &g = *r.ok;
            for (uint64_t i = 0; i < g.tensor_count(); ++i) {
                GGufTensorInfo t = g.tensor_at(i);
                for (char c : t.name) sink += (uint8_t)c;
                if (t.data && t.nbytes) sink += t.data[0] + t.data[t.nbytes - 1];
            }
            for (uint64_t i = 0; i < g.metadata_kv_count(); ++i) {
                GGufMetaKV kv = g.meta_kv_at(i);
                for (char c : kv.key) sink += (uint8_t)c;
                if (!kv.value_bytes.empty()) sink += (uint8_t)kv.value_bytes.front() + (uint8_t)kv.value_bytes.back();
            }
        } else {
            ++bad;
        }
        std::free(exact);
    }
    std::printf("fuzz ok: %ld parsed, %ld rejected (%llu)\n", ok, bad, (unsigned long long)sink);
    return 0;
}
