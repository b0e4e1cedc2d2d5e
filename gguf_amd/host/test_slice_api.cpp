// The reference's slice-API tests (ggml-quants/src/lib.rs:274-369, instantiated
// there for Q8_0) against the C++ mirror.  `--no-device` runs only the cases that
// are decided before any device work (the error cases); without the flag the
// decode cases run too and need a GPU.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ggml_quants.hpp"

using namespace ggml_quants;

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

constexpr size_t N = 32;

// f32 -> f16 bits, round to nearest even (finite inputs only; used to build a test block)
static uint16_t f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int e = (int)((x >> 23) & 0xFF) - 112;
    uint32_t man = x & 0x7FFFFFu;
    if (e <= 0) return (uint16_t)sign;
    uint32_t q = ((uint32_t)e << 10) | (man >> 13);
    uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1))) ++q;
    return (uint16_t)(sign | q);
}

int main(int argc, char **argv) {
    bool device = !(argc > 1 && std::strcmp(argv[1], "--no-device") == 0);

    static_assert(DataBlock<Q8_0>::COUNT == N, "test_block_zeros_and_count");

    {   // test_dequant_slice_indivisible (lib.rs:313-321)
        std::vector<float> out(N - 1);
        std::vector<Q8_0> src(1);
        auto r = QuantExt<Q8_0, float>::dequantize_slice({out.data(), out.size()}, {src.data(), src.size()});
        CHECK(r && *r == QuantizeError::Indivisible);
    }
    {   // test_dequant_slice_length_mismatch (lib.rs:323-331)
        std::vector<float> out(2 * N);
        std::vector<Q8_0> src(3);
        auto r = QuantExt<Q8_0, float>::dequantize_slice({out.data(), out.size()}, {src.data(), src.size()});
        CHECK(r && *r == QuantizeError::LengthMismatch);
    }
    {   // order of the two checks (lib.rs:136-141)
        std::vector<float> out(N - 1);
        std::vector<Q8_0> src(3);
        auto r = QuantExt<Q8_0, float>::dequantize_slice({out.data(), out.size()}, {src.data(), src.size()});
        CHECK(r && *r == QuantizeError::Indivisible);
    }
    {   // test_quant_slice_indivisible (lib.rs:293-301)
        std::vector<float> in(N - 1);
        std::vector<Q8_0> q(1);
        auto r = QuantExt<Q8_0, float>::quantize_slice({q.data(), q.size()}, {in.data(), in.size()});
        CHECK(r && *r == QuantizeError::Indivisible);
    }
    {   // test_quant_slice_length_mismatch (lib.rs:303-311)
        std::vector<float> in(2 * N);
        std::vector<Q8_0> q(3);
        auto r = QuantExt<Q8_0, float>::quantize_slice({q.data(), q.size()}, {in.data(), in.size()});
        CHECK(r && *r == QuantizeError::LengthMismatch);
    }
    if (device) {
        {   // quantize on the device, then decode: test_quant_dequant_slice_ok (lib.rs:276-291)
            std::vector<float> in(2 * N), out(2 * N);
            for (size_t i = 0; i < in.size(); ++i) in[i] = std::cos(0.41f * (float)i) * 0.95f;
            std::vector<Q8_0> q(2);
            auto r1 = QuantExt<Q8_0, float>::quantize_slice({q.data(), q.size()}, {in.data(), in.size()});
            auto r2 = QuantExt<Q8_0, float>::dequantize_slice({out.data(), out.size()}, {q.data(), q.size()});
            CHECK(!r1 && !r2);
            for (size_t i = 0; i < in.size(); ++i) CHECK(std::fabs(in[i] - out[i]) < 4.5e-3f);
        }
        {   // test_block_zeros_and_count (lib.rs:333-338)
            std::vector<Q8_0> src(2);
            std::memset(src.data(), 0, sizeof(Q8_0) * 2);
            std::vector<float> out(2 * N, 5.0f);
            auto r = QuantExt<Q8_0, float>::dequantize_slice({out.data(), out.size()}, {src.data(), src.size()});
            CHECK(!r);
            for (float v : out) CHECK(v == 0.0f && !std::signbit(v));
        }
        {   // test_quant_dequant_slice_ok (lib.rs:276-291), quantized here by the rule of q8_0.rs:23-41
            std::vector<float> in(2 * N);
            for (size_t i = 0; i < in.size(); ++i) in[i] = std::sin(0.37f * (float)i) * 0.9f;
            std::vector<Q8_0> q(2);
            for (size_t b = 0; b < 2; ++b) {
                float amax = 0;
                for (size_t i = 0; i < N; ++i) amax = std::fmax(amax, std::fabs(in[b * N + i]));
                float delta = amax / 127.0f, recip = 1.0f / delta;
                q[b].delta.bits = f16_bits(delta);
                for (size_t i = 0; i < N; ++i) q[b].quants[i] = (int8_t)std::round(in[b * N + i] * recip);
            }
            std::vector<float> out(2 * N);
            auto r = QuantExt<Q8_0, float>::dequantize_slice({out.data(), out.size()}, {q.data(), q.size()});
            CHECK(!r);
            for (size_t i = 0; i < in.size(); ++i) CHECK(std::fabs(in[i] - out[i]) < 4.5e-3f);
            // f16 output path (lib.rs:340-351): same values narrowed
            std::vector<f16> out16(2 * N);
            auto r2 = QuantExt<Q8_0, f16>::dequantize_slice({out16.data(), out16.size()}, {q.data(), q.size()});
            CHECK(!r2);
            for (size_t i = 0; i < in.size(); ++i) CHECK(out16[i].bits == f16_bits(out[i]));
        }
    }
    if (device) {
        {   // a block whose packed bytes were worked out by hand (q8_0.rs:23-41): x = 0 .. 31, delta = 31 / 127 ->
            // f16 0x33D0, code = round(i * 127 / 31)
            static const int8_t want[N] = {0, 4, 8, 12, 16, 20, 25, 29, 33, 37, 41, 45, 49, 53, 57, 61, 66, 70, 74, 78, 82,
                                           86, 90, 94, 98, 102, 107, 111, 115, 119, 123, 127};
            std::vector<float> in(N);
            for (size_t i = 0; i < N; ++i) in[i] = (float)i;
            std::vector<Q8_0> q(1);
            auto r = QuantExt<Q8_0, float>::quantize_slice({q.data(), q.size()}, {in.data(), in.size()});
            CHECK(!r);
            CHECK(q[0].delta.bits == 0x33D0);
            for (size_t i = 0; i < N; ++i) CHECK(q[0].quants[i] == want[i]);
        }
        {   // the quantizer kernels' short divisions against the general division, every float in range, on this GPU
            uint64_t bad[5] = {1, 1, 1, 1, 1}, tested = 0;
            CHECK(ggq_selftest_fast_division(bad, &tested) == GGQ_OK);
            CHECK(tested == 2 * (200ull * (1u << 23) + 1));
            for (uint64_t b : bad) CHECK(b == 0);
        }
    }
    if (failures) std::printf("%d FAILED\n", failures);
    else std::printf("slice api ok (%s)\n", device ? "with device" : "no device");
    return failures ? 1 : 0;
}
