"""Third, pure-Python restatement of the third-party sampler arithmetic (small cases only).
Independent of both the C oracle and the product's pgsgd_math.hpp; used to cross-check the oracle."""
import math
import struct

M64 = (1 << 64) - 1


def splitmix64(state):
    state = (state + 0x9e3779b97f4a7c15) & M64
    z = state
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
    return state, z ^ (z >> 31)


class Xoshiro256Plus:
    def __init__(self, seed):
        s = []
        st = seed & M64
        for _ in range(4):
            st, z = splitmix64(st)
            s.append(z)
        self.s = s

    def next(self):
        s = self.s
        result = (s[0] + s[3]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & M64
        return result


def uniform_below(g, rng):
    x = g.next()
    prod = x * rng
    low = prod & M64
    if low < rng:
        thr = ((1 << 64) - rng) % rng
        while low < thr:
            x = g.next()
            prod = x * rng
            low = prod & M64
    return prod >> 64


def canonical(g):
    r = float(g.next()) * 2.0 ** -64   # int -> float is round-to-nearest-even, like the C cast
    return r if r < 1.0 else 1.0 - 2.0 ** -53


def fast_precise_pow(a, b):
    e = int(b)
    bits = struct.unpack("<q", struct.pack("<d", a))[0]
    hi = bits >> 32                      # arithmetic shift: signed high word
    nhi = int((b - e) * (hi - 1072632447) + 1072632447)
    frac = struct.unpack("<d", struct.pack("<q", (nhi & 0xFFFFFFFF) << 32 if nhi >= 0 else ((nhi & 0xFFFFFFFF) << 32) - (1 << 64)))[0]
    r = 1.0
    while e:
        if e & 1:
            r *= a
        a *= a
        e >>= 1
    return r * frac


def zipf(g, n, theta, zeta_n):
    alpha = 1.0 / (1.0 - theta)
    zeta2 = fast_precise_pow(1.0, theta) + fast_precise_pow(0.5, theta)
    num, den = 1.0 - fast_precise_pow(2.0 / n, 1.0 - theta), 1.0 - zeta2 / zeta_n
    eta = num / den if den != 0 else math.copysign(math.inf, num)
    u = canonical(g)
    uz = u * zeta_n
    if uz < 1.0:
        return 1
    if uz < 1.0 + fast_precise_pow(0.5, theta):
        return 2
    v = 1.0 + n * fast_precise_pow(eta * u - eta + 1.0, alpha)
    return max(1, min(n, int(v)))
