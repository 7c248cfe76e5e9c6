// check_libstdcxx.cpp — pins the oracle's restated libstdc++ distributions against the REAL
// libstdc++ of this image (the reference is built with the same family: std::uniform_int_distribution
// and std::generate_canonical at src/algorithms/path_sgd_layout.cpp:175-176,235; inside dirtyzipf).
// TEST INFRASTRUCTURE ONLY.  Prints "OK <n>" or the first mismatch; exit code 0/1.
#include <cstdint>
#include <cstdio>
#include <limits>
#include <random>
#include "pgsgd_oracle.h"

struct XoshiroAdaptor {             // a UniformRandomBitGenerator over the oracle's Xoshiro256+
    using result_type = uint64_t;
    uint64_t s[4];
    explicit XoshiroAdaptor(uint64_t seed) { orc_rng_seed(seed, s); }
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return std::numeric_limits<uint64_t>::max(); }
    result_type operator()() { return orc_rng_next(s); }
};

int main() {
    uint64_t checked = 0;
    const uint64_t ranges[] = {1, 2, 3, 7, 12, 100, 3100, 35059, 202806, 50000000ull,
                               (1ull << 32) + 12345, (1ull << 63) + 99, 0xFFFFFFFFFFFFFFFEull};
    for (uint64_t seed = 9399220; seed < 9399220 + 8; ++seed) {
        for (uint64_t range : ranges) {
            XoshiroAdaptor a(seed);
            uint64_t s[4];
            orc_rng_seed(seed, s);
            std::uniform_int_distribution<uint64_t> dis(0, range - 1);
            for (int i = 0; i < 20000; ++i) {
                const uint64_t want = dis(a), got = orc_uniform_u64(s, range);
                if (want != got) { std::printf("MISMATCH uniform seed=%llu range=%llu i=%d want=%llu got=%llu\n",
                    (unsigned long long)seed, (unsigned long long)range, i, (unsigned long long)want, (unsigned long long)got); return 1; }
                ++checked;
            }
        }
        {   // flip(0,1) interleaved with canonical draws, as the worker loop does
            XoshiroAdaptor a(seed);
            uint64_t s[4];
            orc_rng_seed(seed, s);
            std::uniform_int_distribution<uint64_t> flip(0, 1);
            for (int i = 0; i < 20000; ++i) {
                const uint64_t wf = flip(a), gf = orc_uniform_u64(s, 2);
                const double wc = std::generate_canonical<double, std::numeric_limits<double>::digits>(a);
                const double gc = orc_canonical(s);
                if (wf != gf || wc != gc) { std::printf("MISMATCH flip/canonical seed=%llu i=%d\n", (unsigned long long)seed, i); return 1; }
                checked += 2;
            }
        }
    }
    {   // canonical edge: a draw of 2^64-1 rounds to 1.0 and must come back as nextafter(1,0)
        struct Fixed { using result_type = uint64_t; static constexpr uint64_t min() { return 0; }
                       static constexpr uint64_t max() { return ~0ull; } uint64_t operator()() { return ~0ull; } } f;
        const double w = std::generate_canonical<double, 53>(f);
        if (!(w < 1.0) || w != 0x1.fffffffffffffp-1) { std::printf("MISMATCH canonical edge %a\n", w); return 1; }
        ++checked;
    }
    std::printf("OK %llu\n", (unsigned long long)checked);
    return 0;
}
