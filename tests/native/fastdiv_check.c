/* Host-side check of the quantizers' short division by a constant (gguf_amd/csrc/quantize.h, div_const_fast):
 *   q = RN(a * RN(1/c));  r = fma(-q, c, a);  result = fma(r, RN(1/c), q)
 * against the correctly rounded a / c, for every float a of the given binary exponents (all 2^23 mantissas, both
 * signs).  Prints the number of inputs whose bits differ, per constant.  Test infrastructure only.
 *   usage: fastdiv_check <first exponent> <last exponent>          (unbiased, 2^e <= |a| < 2^(e+1)) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static uint64_t run(float c, int e0, int e1) {
    const float rc = 1.0f / c;
    uint64_t bad = 0;
    for (int e = e0; e <= e1; ++e) {
        const uint32_t base = (uint32_t)(e + 127) << 23;
        for (uint32_t sign = 0; sign < 2; ++sign) {
            uint64_t n = 0;
            for (uint32_t m = 0; m < (1u << 23); ++m) {
                const float a = from_bits(base | m | (sign << 31));
                const float q = a * rc;
                const float r = fmaf(-q, c, a);
                const float f = fmaf(r, rc, q);
                n += to_bits(f) != to_bits(a / c);
            }
            bad += n;
        }
    }
    return bad;
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    const int e0 = atoi(argv[1]), e1 = atoi(argv[2]);
    const float cs[4] = {15.0f, 31.0f, 127.0f, -127.0f};
    for (int k = 0; k < 4; ++k) printf("%g %llu\n", cs[k], (unsigned long long)run(cs[k], e0, e1));
    return 0;
}
