// pgsgd_math.hpp — sampler arithmetic shared by the host code and the HIP kernels.
//
// The per-stream generator and distributions are the ones a reference worker thread uses
// (src/algorithms/path_sgd_layout.cpp:168-176): XoshiroCpp::Xoshiro256Plus seeded through
// SplitMix64, libstdc++'s uniform_int_distribution / generate_canonical on top of it, and
// dirtyzipf's approximate-pow Zipf sampler.  All of it is integer or IEEE fp64 arithmetic with no
// fused multiply-adds (the translation units are built with -ffp-contract=off), so the device
// streams can be checked term by term against the CPU oracle.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PGSGD_HD __host__ __device__ __forceinline__
#else
#define PGSGD_HD inline
#endif

namespace pgsgd {

struct Xoshiro256Plus {
    uint64_t s0, s1, s2, s3;

    PGSGD_HD static uint64_t splitmix64(uint64_t& x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    PGSGD_HD void seed(uint64_t seed) {
        uint64_t x = seed;
        s0 = splitmix64(x);
        s1 = splitmix64(x);
        s2 = splitmix64(x);
        s3 = splitmix64(x);
    }
    PGSGD_HD uint64_t next() {
        const uint64_t result = s0 + s3;
        const uint64_t t = s1 << 17;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 45) | (s3 >> 19);
        return result;
    }
};

PGSGD_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// std::uniform_int_distribution<uint64_t>(0, range-1) of libstdc++ >= 11 on a full-range 64-bit
// generator: Lemire multiply-shift with rejection.  range must be >= 1.
PGSGD_HD uint64_t uniform_below(Xoshiro256Plus& g, uint64_t range) {
    uint64_t x = g.next();
    uint64_t low = x * range;
    if (low < range) {
        const uint64_t threshold = (0 - range) % range;
        while (low < threshold) {
            x = g.next();
            low = x * range;
        }
    }
    return mulhi64(x, range);
}

// flip(0,1): range 2 never rejects, the result is the top bit of one draw
PGSGD_HD uint32_t coin(Xoshiro256Plus& g) { return (uint32_t)(g.next() >> 63); }

// std::generate_canonical<double,53>: one draw / 2^64, >= 1 replaced by nextafter(1,0)
PGSGD_HD double canonical(Xoshiro256Plus& g) {
    double r = (double)g.next() * 0x1p-64;
    if (r >= 1.0) r = 0x1.fffffffffffffp-1;
    return r;
}

// dirtyzipf::fast_precise_pow: exact a^int(b) by squaring times a bit-level estimate of a^frac(b)
PGSGD_HD double fast_precise_pow(double a, double b) {
    int e = (int)b;
    const int64_t bits = __builtin_bit_cast(int64_t, a);
    const int32_t hi = (int32_t)(bits >> 32);
    const int32_t nhi = (int32_t)((b - (double)e) * (double)(hi - 1072632447) + 1072632447.0);
    const double frac = __builtin_bit_cast(double, (int64_t)((uint64_t)(uint32_t)nhi << 32));
    double r = 1.0;
    while (e) {
        if (e & 1) r *= a;
        a *= a;
        e >>= 1;
    }
    return r * frac;
}

// fast_precise_pow(a, b) with the exponent split once, on the host: e = (int)b, bfrac = b - (double)e.  The same
// operations in the same order, so the same bits; e is then a uniform (scalar) loop count instead of a per-lane one.
// a^E for a compile-time E: the multiplications of the square-and-multiply loop below, in its order, without the
// ones whose result the loop throws away (the squaring after the top bit) or that multiply by 1.0 (the first r *= a,
// exact) — the same bits.  E = 100 (theta = 0.99, the default of `odgi layout`): 6 squarings and 2 products instead
// of 7 + 7 products and 14 selects.
template <int E>
PGSGD_HD double pow_int_fixed(double a) {
    static_assert(E > 0, "positive exponent");
    double r = 1.0;
    bool first = true;
#pragma unroll
    for (int e = E; e; e >>= 1) {
        if (e & 1) {
            r = first ? a : r * a;
            first = false;
        }
        if (e > 1) a *= a;
    }
    return r;
}

PGSGD_HD double pow_split(double a, int e, double bfrac) {
    const int64_t bits = __builtin_bit_cast(int64_t, a);
    const int32_t hi = (int32_t)(bits >> 32);
    const int32_t nhi = (int32_t)(bfrac * (double)(hi - 1072632447) + 1072632447.0);
    const double frac = __builtin_bit_cast(double, (int64_t)((uint64_t)(uint32_t)nhi << 32));
    if (e == 100) return pow_int_fixed<100>(a) * frac;  // (e is the same for every lane: a scalar branch)
    double r = 1.0;
    while (e) {
        if (e & 1) r *= a;
        a *= a;
        e >>= 1;
    }
    return r * frac;
}

// constants of the Zipf sampler that depend on theta only
struct ZipfConst {
    double theta, alpha, one_minus_theta, zeta2, one_plus_half_pow;
    double alpha_frac, omt_frac;  // exponents split for pow_split
    int alpha_e, omt_e;
    PGSGD_HD void init(double th) {
        theta = th;
        alpha = 1.0 / (1.0 - th);
        one_minus_theta = 1.0 - th;
        const double half_pow = fast_precise_pow(0.5, th);
        zeta2 = fast_precise_pow(1.0, th) + half_pow;
        one_plus_half_pow = 1.0 + half_pow;
        alpha_e = (int)alpha;
        alpha_frac = alpha - (double)alpha_e;
        omt_e = (int)one_minus_theta;
        omt_frac = one_minus_theta - (double)omt_e;
    }
};

// dirtyzipf::dirty_zipfian_int_distribution<uint64_t>(1, n, theta, zeta_n): value in [1, n]
PGSGD_HD uint64_t zipf(Xoshiro256Plus& g, const ZipfConst& zc, uint64_t n, double zeta_n) {
    const double eta = (1.0 - fast_precise_pow(2.0 / (double)n, zc.one_minus_theta)) / (1.0 - zc.zeta2 / zeta_n);
    const double u = canonical(g);
    const double uz = u * zeta_n;
    if (uz < 1.0) return 1;
    if (uz < zc.one_plus_half_pow) return 2;
    const double v = 1.0 + (double)n * fast_precise_pow(eta * u - eta + 1.0, zc.alpha);
    uint64_t r = (v >= 1.0 && v < 1.8446744073709552e19) ? (uint64_t)v : 1;
    if (r < 1) r = 1;
    if (r > n) r = n;
    return r;
}

// The same draw with what depends on zeta_n alone taken from a table: denom = 1.0 - zc.zeta2 / zeta_n, computed once
// per table entry by the same two fp64 operations (zipf_denominator), and the exponents pre-split (pow_split).
// Bit-identical to zipf(): one fp64 division and two per-lane exponent loops less per draw.
PGSGD_HD double zipf_denominator(const ZipfConst& zc, double zeta_n) { return 1.0 - zc.zeta2 / zeta_n; }
// (the variate u = canonical(g) drawn by the caller: a pipelined kernel draws it a stage before it has zeta_n)
PGSGD_HD uint64_t zipf_tabled_u(double u, const ZipfConst& zc, uint64_t n, double zeta_n, double denom) {
    const double eta = (1.0 - pow_split(2.0 / (double)n, zc.omt_e, zc.omt_frac)) / denom;
    const double uz = u * zeta_n;
    if (uz < 1.0) return 1;
    if (uz < zc.one_plus_half_pow) return 2;
    const double v = 1.0 + (double)n * pow_split(eta * u - eta + 1.0, zc.alpha_e, zc.alpha_frac);
    uint64_t r = (v >= 1.0 && v < 1.8446744073709552e19) ? (uint64_t)v : 1;
    if (r < 1) r = 1;
    if (r > n) r = n;
    return r;
}
PGSGD_HD uint64_t zipf_tabled(Xoshiro256Plus& g, const ZipfConst& zc, uint64_t n, double zeta_n, double denom) {
    return zipf_tabled_u(canonical(g), zc, n, zeta_n, denom);  // (eta does not depend on the draw: the same operations, the same bits)
}

// index into the zeta cache for a jump of `jump` steps (path_sgd_layout.cpp:208-212)
PGSGD_HD uint64_t zeta_index(uint64_t jump, uint64_t space_max, uint64_t quant) {
    return jump > space_max ? space_max + (jump - space_max) / quant + 1 : jump;
}

}  // namespace pgsgd
