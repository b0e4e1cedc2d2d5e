// The reference's container tests against the C++ mirror ggus.hpp:
//   simulator      ggus/src/write/simulator.rs:110-131
//   file writer    ggus/src/write/file_writer.rs:150-371
//   reader         ggus/src/file.rs:313-443 (files are built here with the writer under test)
// Host code only; no device is touched.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unistd.h>
#include <vector>

#include "ggus.hpp"

using namespace ggus;
using T = GGufMetaDataValueType;

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static std::string u32le(uint32_t v) { return std::string((const char *)&v, 4); }
static std::string gstr(const std::string &s) {
    uint64_t n = s.size();
    return std::string((const char *)&n, 8) + s;
}
static std::vector<uint8_t> slurp(const std::string &path) {
    std::vector<uint8_t> out;
    if (FILE *f = std::fopen(path.c_str(), "rb")) {
        uint8_t buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
        std::fclose(f);
    }
    return out;
}
constexpr uint32_t F32 = 0, F16 = 1;

int main() {
    char dir_template[] = "/tmp/ggus_api_XXXXXX";
    const char *dir = mkdtemp(dir_template);
    CHECK(dir != nullptr);
    const std::string base = std::string(dir ? dir : "/tmp") + "/";

    {   // test_simulator (simulator.rs:110-131)
        GGufFileWriter sim = simulator();
        CHECK(sim.alignment() == DEFAULT_ALIGNMENT);
        CHECK(sim.write_meta_kv("test_key", T::String, "test_value") == GGQ_GGUF_OK);
        CHECK(sim.write_alignment(16) == GGQ_GGUF_OK);
        GGufTensorWriter ts = sim.finish();
        CHECK(ts.write_tensor("test_tensor", F32, {2, 2}, (const void *)nullptr) == GGQ_GGUF_OK);
        CHECK(ts.written_bytes() == 160);
    }
    {   // test_file_writer_new / with_alignment / write_alignment (file_writer.rs:150-184)
        GGufFileWriter w((base + "a.gguf").c_str(), 0, 0);
        CHECK(w.error == GGQ_GGUF_OK);
        CHECK(w.alignment() == DEFAULT_ALIGNMENT);
        CHECK(w.write_alignment(128) == GGQ_GGUF_OK);
        CHECK(w.alignment() == 128);
    }
    {   // test_write_meta_kv (file_writer.rs:186-222)
        GGufFileWriter w(nullptr, 0, 0);
        CHECK(w.write_meta_kv("test.key", T::U32, u32le(1)) == GGQ_GGUF_OK);
        CHECK(w.write_meta_kv("general.alignment", T::U32, u32le(64)) == GGQ_GGUF_OK);
        CHECK(w.alignment() == 64);
        // a value that is not 4 bytes: the reference panics, here an error value
        CHECK(w.write_meta_kv("general.alignment", T::String, std::string("test\0", 5)) == GGQ_GGUF_ALIGNMENT_TYPE);
    }
    {   // test_tensor_writer_write_tensor / multiple_tensors (file_writer.rs:240-291)
        GGufFileWriter w(nullptr, 0, 0);
        GGufTensorWriter t = w.finish();
        CHECK(t.alignment() == DEFAULT_ALIGNMENT && t.offset() == 0);
        std::vector<uint8_t> d1(24), d2(32);
        CHECK(t.write_tensor("tensor1", F32, {2, 3}, d1.data()) == GGQ_GGUF_OK);
        CHECK(t.offset() == 24);
        CHECK(t.write_tensor("tensor2", F16, {4, 4}, d2.data()) == GGQ_GGUF_OK);
        CHECK(t.offset() == 24 + 8 + 32);   // 24 + pad(24, 32) + 32
    }
    std::string model = base + "model.gguf";
    std::vector<uint8_t> d1(48), d2(10), d3(100);
    for (size_t i = 0; i < d1.size(); ++i) d1[i] = (uint8_t)i;
    for (size_t i = 0; i < d2.size(); ++i) d2[i] = (uint8_t)(200 + i);
    for (size_t i = 0; i < d3.size(); ++i) d3[i] = (uint8_t)(3 * i);
    {   // test_end_to_end_write_process (file_writer.rs:316-371), then read back:
        // test_valid_gguf_parsing (file.rs:313-351): 3 key/values, 2 tensors + one lazy tensor
        GGufFileWriter w(model.c_str(), 3, 3);
        CHECK(w.error == GGQ_GGUF_OK);
        CHECK(w.write_meta_kv("general.architecture", T::String, gstr("llama")) == GGQ_GGUF_OK);
        CHECK(w.write_meta_kv("llm.context_length", T::U32, u32le(4096)) == GGQ_GGUF_OK);
        CHECK(w.write_meta_kv("general.name", T::String, gstr("test_model")) == GGQ_GGUF_OK);
        GGufTensorWriter t = w.finish();
        CHECK(t.write_tensor("tensor1", F32, {3, 4}, d1.data()) == GGQ_GGUF_OK);
        CHECK(t.write_tensor("tensor2", F16, {5}, d2.data()) == GGQ_GGUF_OK);
        int calls = 0;
        CHECK(t.write_tensor("embeddings", F32, {5, 5}, DataFuture([&] { ++calls; return (const void *)d3.data(); })) == GGQ_GGUF_OK);
        CHECK(calls == 0);                 // lazy: not evaluated before finish
        const uint64_t forecast = t.written_bytes();
        auto [e, n] = t.finish();
        CHECK(e == GGQ_GGUF_OK && n > 158 && n == forecast && calls == 1);
        CHECK(slurp(model).size() == n);
    }
    {
        auto r = GGuf::open(model.c_str());
        CHECK((bool)r);
        if (r) {
            GGuf &g = *r.ok;
            CHECK(g.version() == 3 && g.tensor_count() == 3 && g.metadata_kv_count() == 3 && g.alignment() == 32);
            auto arch = g.get("general.architecture");
            CHECK(arch && arch->ty == T::String && arch->value_bytes == gstr("llama"));
            auto ctx = g.get("llm.context_length");
            CHECK(ctx && ctx->ty == T::U32 && ctx->value_bytes == u32le(4096));
            CHECK(!g.get("no.such.key"));
            auto t1 = g.tensor("tensor1"), t2 = g.tensor("tensor2"), t3 = g.tensor("embeddings");
            CHECK(t1 && t1->ty == F32 && t1->shape == std::vector<uint64_t>({3, 4}) && t1->offset == 0);
            CHECK(t2 && t2->ty == F16 && t2->shape == std::vector<uint64_t>({5}) && t2->offset == 64);
            CHECK(t3 && t3->offset == 96 && t3->nbytes == 100);
            CHECK(t1 && t1->data && std::memcmp(t1->data, d1.data(), 48) == 0);
            CHECK(t2 && t2->data && std::memcmp(t2->data, d2.data(), 10) == 0);
            CHECK(t3 && t3->data && std::memcmp(t3->data, d3.data(), 100) == 0);
            CHECK(g.tensor_at(0).name == "tensor1" && g.tensor_at(2).name == "embeddings");   // file order kept
            CHECK(g.data_offset() % 32 == 0);
        }
    }
    std::vector<uint8_t> bytes = slurp(model);
    {   // test_invalid_magic / test_invalid_version (file.rs:354-365)
        std::vector<uint8_t> b = bytes;
        b[0] = 'X';
        auto r = GGuf::parse(b.data(), b.size());
        CHECK(!r && r.err->code == GGQ_GGUF_MAGIC);
        b = bytes;
        b[4] = 2;
        r = GGuf::parse(b.data(), b.size());
        CHECK(!r && r.err->code == GGQ_GGUF_VERSION);
    }
    {   // test_reading_truncated_data (file.rs:427-435): cut inside the key/values
        auto r = GGuf::parse(bytes.data(), 40);
        CHECK(!r && r.err->code == GGQ_GGUF_EOS);
    }
    {   // test_extra_data_handling (file.rs:438-443)
        std::vector<uint8_t> b = bytes;
        b.insert(b.end(), 100, 0xEE);
        auto r = GGuf::parse(b.data(), b.size());
        CHECK(r && r.ok->tensor_count() == 3);
    }
    {   // test_duplicate_meta_key / test_duplicate_tensor_name / test_invalid_alignment_type (file.rs:368-395)
        std::string p = base + "dup.gguf";
        {
            GGufFileWriter w(p.c_str(), 0, 2);
            w.write_meta_kv("duplicate.key", T::U32, u32le(1));
            w.write_meta_kv("duplicate.key", T::U32, u32le(2));
            w.finish().finish();
        }
        auto r = GGuf::open(p.c_str());
        CHECK(!r && r.err->code == GGQ_GGUF_DUP_KEY && r.err->name == "duplicate.key");
        {
            GGufFileWriter w(p.c_str(), 2, 0, false);
            GGufTensorWriter t = w.finish();
            t.write_tensor("duplicate_tensor", F32, {1}, (const void *)nullptr);
            t.write_tensor("duplicate_tensor", F32, {1}, (const void *)nullptr);
            t.finish();
        }
        r = GGuf::open(p.c_str());
        CHECK(!r && r.err->code == GGQ_GGUF_DUP_TENSOR && r.err->name == "duplicate_tensor");
        {   // a string where general.alignment should be a number: only a hand-built file can hold it
            std::string f = std::string("GGUF") + u32le(3) + std::string(8, '\0');
            uint64_t one = 1;
            f.replace(16, 0, std::string((const char *)&one, 8));
            f.resize(24);
            f += gstr("general.alignment") + u32le((uint32_t)T::String) + gstr("x");
            auto r2 = GGuf::parse(f.data(), f.size());
            CHECK(!r2 && r2.err->code == GGQ_GGUF_ALIGNMENT_TYPE);
        }
    }
    {   // shard planning (xtask/src/utils/write.rs:21-51)
        std::vector<PlanTensor> t = {{"a", F32, {8}}, {"b", F32, {8}}};
        auto p = plan_shards(32, {}, t, UINT64_MAX, UINT64_MAX, false);
        CHECK(p.error == GGQ_GGUF_OK && p.n_shards == 1 && p.shard_of == std::vector<uint32_t>({0, 0}));
        p = plan_shards(32, {}, t, UINT64_MAX, UINT64_MAX, true);
        CHECK(p.n_shards == 2 && p.shard_of == std::vector<uint32_t>({1, 1}));
        p = plan_shards(32, {}, t, 1, UINT64_MAX, false);
        CHECK(p.n_shards == 2 && p.shard_of == std::vector<uint32_t>({0, 1}));
        p = plan_shards(32, {{"k", T::String, gstr("v")}}, t, UINT64_MAX, 1, false);
        CHECK(p.n_shards == 3 && p.shard_of == std::vector<uint32_t>({1, 2}));
    }
    std::string cmd = "rm -rf " + std::string(dir ? dir : "/tmp/ggus_api_none");
    if (dir) (void)!std::system(cmd.c_str());
    if (failures) {
        std::printf("%d failure(s)\n", failures);
        return 1;
    }
    std::printf("ggus api ok\n");
    return 0;
}
