// Host build of lm_math.h: the shortcut Halton radical inverse (bases 2 and 3) against the reference-shaped digit loop, bit for bit.
// Test infrastructure only (tests/test_cpu_host.py); prints the number of mismatches.
#include <cstring>
#include "lm_math.h"
#include <cstdio>
#include <random>
int main(int argc, char** argv)
{
    const uint32_t dense = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000000u;
    std::mt19937 rng(7); unsigned long long bad = 0, n = 0;
    auto chk = [&](uint32_t i) {
        for (uint32_t b : {2u, 3u}) {
            const float a = lm_halton(i, b), c = lm_halton_loop(i, b);
            uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &c, 4); n++;
            if (x != y) { if (bad < 5) printf("mismatch index %u base %u: %08x %08x\n", i, b, x, y); bad++; }
        }
    };
    for (uint32_t i = 0; i < dense; i++) chk(i);                                   // every pixel index of a frame
    for (uint32_t k = 0; k < dense; k++) chk(rng());                               // all magnitudes
    for (uint32_t i = 0xffffff00u; i != 0; i++) chk(i);                            // up to the index that wraps to 0 in `++index`
    for (uint32_t i = (1u << 24) - 1000; i < (1u << 24) + 1000; i++) chk(i);       // where base 2 changes from the bit reversal to the loop
    for (uint32_t p = 3, k = 1; k < 21; k++, p *= 3) for (int d = -2; d <= 2; d++) chk(p + (uint32_t)d);     // digit-count boundaries of base 3
    printf("%llu checks, %llu mismatches\n", n, bad);
    return bad != 0;
}
