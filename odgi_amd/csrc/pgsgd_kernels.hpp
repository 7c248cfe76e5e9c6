// pgsgd_kernels.hpp — device code of the per-lane SGD kernel: data structures, the sampler, the update,
// and the streaming set-up / exchange kernels.  Included by pgsgd_session.hip only (one translation unit).
//
// The HIP side of the path-guided SGD 2D layout for MI355X (gfx950, wave64).
//
// One launch = one learning-rate step ("iteration") of the reference's worker loop
// (src/algorithms/path_sgd_layout.cpp:165-377): every GPU lane is one reference worker.  Lane g
// owns a persistent Xoshiro256+ stream seeded with seed+g (the reference seeds worker tid with
// 9399220+tid, :168-169) and draws, per term, exactly the reference's sequence of variates
// (:182,205-206,215/228,235-237,253,262).  Terms i, i+L, i+2L, ... of an iteration belong to lane i.
//
// HBM layout (built once per session from the caller's SoA view, see build_step_records):
//   recs   [S] x 16 B  {u32 handle = 2*rank+rev, u32 node length, u64 bp position of the step}
//                      one 16-byte gather returns everything the term needs about a step, so the
//                      node_len[] gather and the separate handle/position gathers disappear;
//   coords [2N] x 8 B  one word per node END, the two ends of a node adjacent: {u32 Xq, u32 Yq} fixed
//                      point by default (one 64-bit atomic moves both), {f32 x, f32 y} optionally;
//   path_first [P+1] u64 — staged into LDS, (path, rank) of a flat step index by binary search,
//                      which replaces the npi_iv/nr_iv gathers of the reference (xp.cpp:421-434);
//   zetas  [~space_max + space/quant] f64 — read-only, L2 resident;
//   rng    [4][L] u64 — SoA stream states, read and written once per launch, coalesced.
// Coordinates are read with agent-scope loads (they bypass the per-CU L1, which is never refreshed
// by other CUs' atomics) and updated with hardware atomic adds at agent scope.
//
// Built with -ffp-contract=off: the sampler is integer/fp64 work that the CPU oracle reproduces
// bit for bit, and the fp32 update arithmetic matches the oracle's mirrors for a one-stream run.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pgsgd_math.hpp"

namespace pgsgd {

// Coordinate formats.  Both keep one 8-byte word per node END (the two ends of a node share 16 B):
//   kFmtQ32 (default): {u32 Xq, u32 Yq} fixed point, x = x_off + Xq / scale.  One 64-bit integer
//            atomic add moves x and y together: for signed steps (qx, qy) the word changes by
//            qx + qy*2^32 (mod 2^64), exact as long as each field stays inside [0, 2^32).  The
//            atomic units of MI355X retire ~21-24 G scattered atomics/s whatever their width
//            (tools/microbench.hip), so this halves the cost of the update, the kernel's limiter.
//   kFmtF32: {f32 x, f32 y}, four fp32 atomic adds per term (the literal north-star form).
enum : int { kFmtQ32 = 0, kFmtF32 = 1 };

struct Xform {  // kFmtQ32 only
    double x_off, y_off;
    float scale, inv_scale;  // scale = 2^k quanta per bp
};

struct DevConst {
    const uint4* recs;
    const uint64_t* path_first;
    const double* zetas;
    const double2* zeta_denom;  // [same index] {zeta_n, 1 - zeta2/zeta_n}: one 16-byte load per Zipf draw (zipf_tabled)
    uint64_t* coords;  // [2N] words
    uint64_t* rng;
    unsigned int* delta_max_bits;
    unsigned int* frame_flag;  // set when a coordinate is seen in the outer quarter of the fixed-point frame
    // Concurrent terms on one node end are all computed from the same (stale) position and then added up; with mu = 1 each
    // is a full projection, so h of them overshoot h-fold.  With L lanes in flight an end of a node that carries s of the S
    // path steps sees h = L * s / S terms at once: terms that touch such a node are capped at mu = 1/h (together they
    // still amount to one projection).  node_steps = path steps per node, null when no node reaches h > 1 (every large
    // graph); hot_scale = L / S.
    const uint32_t* node_steps;
    float hot_scale;
    uint64_t n_steps;
    uint32_t n_paths;
    uint32_t n_streams;
    uint32_t n_nodes;
    uint64_t space, space_max, space_quant;
    uint64_t terms_per_anchor;
    uint64_t seed_base;  // params.seed + params.stream_offset
    ZipfConst zc;
    Xform xf;
};

// Fixed-point frame guard.  A field of a coordinate word leaves [0, 2^32) by wrapping (and carries one quantum into
// its neighbour), which would silently move a node end across the whole frame.  The frame is chosen 8x wider than the
// layout, so kernels only watch the outer quarter on either side: a coordinate seen there sets frame_flag, and the
// host doubles the frame (halving the resolution) before the next iteration (reframe_kernel).
__device__ __forceinline__ bool in_frame_guard(uint64_t w) {
    const uint32_t x = (uint32_t)w, y = (uint32_t)(w >> 32);
    return (uint32_t)(x - 0x40000000u) >= 0x80000000u || (uint32_t)(y - 0x40000000u) >= 0x80000000u;
}
__global__ void reframe_kernel(uint64_t* words, uint64_t n) {  // same centre, twice the span: q' = q/2 + 2^30 per field
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = words[i];
        words[i] = (uint64_t)(((uint32_t)w >> 1) + 0x40000000u) | ((uint64_t)(((uint32_t)(w >> 32) >> 1) + 0x40000000u) << 32);
    }
}

struct IterArgs {
    uint64_t n_terms;
    float eta;
    uint32_t cooling;
    uint64_t epoch;  // tile kernel: iterations started so far (part of every term's seed)
};

constexpr int kBlock = 256;
constexpr uint32_t kPathLdsCap = 4096;  // path_first entries staged in LDS (32 KiB)

// largest p with pf[p] <= k  (pf[0] = 0 <= k < pf[n_paths])
template <typename PF>
__device__ __forceinline__ uint32_t find_path(const PF pf, uint32_t n_paths, uint64_t k) {
    uint32_t lo = 0, hi = n_paths;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pf[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}

// Coordinates are read at agent scope (sc1): the load bypasses the per-CU L1, which other CUs'
// atomics never refresh.  COORD_LOAD 0 is the plain (L1-cached) load, kept for A/B profiling.
template <int COORD_LOAD>
__device__ __forceinline__ uint64_t load_word(const uint64_t* coords, uint32_t end_idx) {
    if (COORD_LOAD == 0) return coords[end_idx];
    return __hip_atomic_load(coords + end_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Anchor {
    uint64_t k, pstart, cnt, s_rank;
    uint4 rec;
};

struct Term {
    uint64_t kb;
    uint64_t pos_a, pos_b;
    uint32_t end_a, end_b;  // 2*rank + end offset
    uint32_t handle_b;      // the partner step's handle (2*rank + is_reverse): end_b ^ handle_b = 1 when the far end was chosen
    uint32_t dither;        // low 32 bits of the draw whose top bit chose end a (otherwise unused)
};

// The sampler: path_sgd_layout.cpp:182-270 on the lowered index, split at the point where the
// first step is fixed.  One anchor followed by one partner is exactly the reference's sequence of
// draws; `terms_per_anchor` > 1 draws further partners (:205-270) for the same first step.
template <typename PF>
__device__ __forceinline__ Anchor sample_anchor(const DevConst& c, const PF pf, Xoshiro256Plus& rng) {
    Anchor a;
    do {  // :182-192 — a single-step path makes the reference draw again without counting a term
        a.k = uniform_below(rng, c.n_steps);
        const uint32_t p = find_path(pf, c.n_paths, a.k);
        a.pstart = pf[p];
        a.cnt = pf[p + 1] - a.pstart;
    } while (a.cnt == 1);
    a.rec = c.recs[a.k];  // issued early; consumed after the partner is chosen
    a.s_rank = a.k - a.pstart;
    return a;
}

struct GlobalRecs {  // partner records straight from HBM
    const uint4* recs;
    __device__ __forceinline__ uint4 operator()(uint64_t k) const { return recs[k]; }
};

// What a term draws once its first step is fixed (:205-237, then the two end choices :253,262): none of it depends on
// the partner's record, so the tile kernel can issue the partner's gather and go on with other work.
struct PartnerDraw {
    uint64_t kb;
    uint32_t flip_a, flip_b;
    uint32_t dither;  // low 32 bits of the draw whose top bit chose end a (otherwise unused)
};

__device__ __forceinline__ PartnerDraw draw_partner(const DevConst& c, const Anchor& a, uint32_t cooling, Xoshiro256Plus& rng) {
    PartnerDraw d;
    uint64_t b_rank;
    if (cooling || coin(rng)) {                                               // :205
        const bool back = (a.s_rank > 0 && coin(rng)) || a.s_rank == a.cnt - 1;  // :206
        const uint64_t room = back ? a.s_rank : a.cnt - a.s_rank - 1;
        const uint64_t jump = c.space < room ? c.space : room;
        const double2 zd = c.zeta_denom[zeta_index(jump, c.space_max, c.space_quant)];
        const uint64_t z = zipf_tabled(rng, c.zc, jump, zd.x, zd.y);
        b_rank = back ? a.s_rank - z : a.s_rank + z;
    } else {
        b_rank = uniform_below(rng, a.cnt);                                   // :235-237
    }
    d.kb = a.pstart + b_rank;
    // :242-269 — choose an end of each node; flip(0,1) is the top bit of one draw (uniform_int_distribution never
    // rejects for range 2)
    const uint64_t draw_a = rng.next(), draw_b = rng.next();
    d.flip_a = (uint32_t)(draw_a >> 63);
    d.flip_b = (uint32_t)(draw_b >> 63);
    d.dither = (uint32_t)draw_a;
    return d;
}

// the path position moves to the chosen end of each node (:242-269)
__device__ __forceinline__ Term make_term(const Anchor& a, const PartnerDraw& d, const uint4& rb) {
    Term t;
    t.kb = d.kb;
    t.dither = d.dither;
    const uint32_t h_a = a.rec.x, h_b = rb.x;
    uint64_t pos_a = (uint64_t)a.rec.z | ((uint64_t)a.rec.w << 32);
    uint64_t pos_b = (uint64_t)rb.z | ((uint64_t)rb.w << 32);
    uint32_t off_a = h_a & 1u, off_b = h_b & 1u;
    if (d.flip_a) { pos_a += a.rec.y; off_a ^= 1u; }
    if (d.flip_b) { pos_b += rb.y; off_b ^= 1u; }
    t.pos_a = pos_a;
    t.pos_b = pos_b;
    t.end_a = (h_a & ~1u) | off_a;
    t.end_b = (h_b & ~1u) | off_b;
    t.handle_b = h_b;
    return t;
}

template <class RecFetch>
__device__ __forceinline__ Term sample_partner(const DevConst& c, const Anchor& a, uint32_t cooling, Xoshiro256Plus& rng, RecFetch&& fetch) {
    const PartnerDraw d = draw_partner(c, a, cooling, rng);
    const uint4 rb = fetch(d.kb);
    return make_term(a, d, rb);
}

// The displacement of one term in bp, fp32 (path_sgd_layout.cpp:280-352); dx,dy = p_a - p_b.
// BELOW_2_52: path positions are below 2^52 (checked when a tiled session is created), so the distance converts to
// fp64 exactly from its two halves and is rounded once on the way to fp32 — the bits of (float)(uint64_t), in a
// third of the instructions
template <bool BELOW_2_52 = false>
__device__ __forceinline__ void term_displacement(float eta, uint64_t pos_a, uint64_t pos_b, float dx, float dy,
                                                  float& r_x, float& r_y, float& abs_delta, float mu_cap = 1.0f) {
    const int64_t diff = (int64_t)pos_a - (int64_t)pos_b;
    const uint64_t ad = (uint64_t)(diff < 0 ? -diff : diff);
    float d = BELOW_2_52 ? (float)((double)(uint32_t)(ad >> 32) * 4294967296.0 + (double)(uint32_t)ad) : (float)ad;
    if (d == 0.0f) d = 1e-9f;
    const float w = 1.0f / d;
    float mu = eta * w;
    if (mu > mu_cap) mu = mu_cap;
    if (dx == 0.0f) dx = 1e-9f;
    const float dx2 = dx * dx;
    const float dy2 = dy * dy;
    const float mag = sqrtf(dx2 + dy2);
    const float Delta = (mu * (mag - d)) / 2.0f;
    abs_delta = fabsf(Delta);
    const float r = Delta / mag;
    r_x = r * dx;
    r_y = r * dy;
}

// learning-rate cap of a term by the expected number of concurrent terms on its busier node (DevConst::node_steps)
__device__ __forceinline__ float hot_mu_cap(const DevConst& c, uint32_t end_a, uint32_t end_b) {
    if (!c.node_steps) return 1.0f;
    const uint32_t sa = c.node_steps[end_a >> 1], sb = c.node_steps[end_b >> 1];
    const float h = c.hot_scale * (float)(sa > sb ? sa : sb);
    return h > 1.0f ? 1.0f / h : 1.0f;
}

__device__ __forceinline__ uint64_t pack_f32(float x, float y) {
    return (uint64_t)__float_as_uint(x) | ((uint64_t)__float_as_uint(y) << 32);
}
__device__ __forceinline__ uint64_t q32_shift(uint64_t w, int64_t qx, int64_t qy) {  // per-field add, no carry between fields
    return (uint64_t)(uint32_t)((int64_t)(uint32_t)w + qx) | ((uint64_t)(uint32_t)((int64_t)(w >> 32) + qy) << 32);
}

// UPD: how a term's displacement reaches memory.
//   kUpdAtomic: atomic adds — every concurrent displacement is applied (they accumulate).
//   kUpdStore : the reference CPU's Hogwild form (path_sgd_layout.cpp:360-363: load, subtract,
//               store): the new position of an end is computed from the loaded one and written back
//               with one 8-byte agent-scope store; a concurrent update of the same end in between is
//               overwritten, never summed.
enum : int { kUpdAtomic = 0, kUpdStore = 1 };

// ABL: profiling ablations (never used by the product path; PGSGD_FLAG_ABLATE selects them)
//   1 = no atomics, 3 = no coordinate loads, 4 = neither
//
// One anchor group = one first step `a` with `terms_per_anchor` partners.  The anchor's record and
// both of its node ends are fetched once per group; the anchor-side displacement of each term is
// applied to the lane's private copy at once (the next partner sees it) and reaches memory as ONE
// update per touched end when the group ends.  Partner-side updates go out term by term.
// terms_per_anchor = 1 is the reference's term stream.
// GROUPED = false is the terms_per_anchor == 1 instance: same results, no group bookkeeping, fewer
// registers (7 instead of 5 resident waves per SIMD).
template <bool PF_LDS, int COORD_LOAD, int FMT, int UPD, bool GROUPED, int ABL = 0>
__global__ __launch_bounds__(kBlock) void sgd_iteration_kernel(DevConst c, IterArgs a) {
    extern __shared__ uint64_t s_pf[];
    if (PF_LDS) {
        for (uint32_t i = threadIdx.x; i <= c.n_paths; i += blockDim.x) s_pf[i] = c.path_first[i];
        __syncthreads();
    }
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_streams) return;
    const uint64_t* pf = PF_LDS ? s_pf : c.path_first;
    Xoshiro256Plus rng;
    const size_t L = c.n_streams;
    rng.s0 = c.rng[g];
    rng.s1 = c.rng[L + g];
    rng.s2 = c.rng[2 * L + g];
    rng.s3 = c.rng[3 * L + g];
    float dmax = 0.0f;
    bool guard = false;
    if (!GROUPED) {
        for (uint64_t ti = g; ti < a.n_terms; ti += L) {
            const Anchor an = sample_anchor(c, pf, rng);
            const Term t = sample_partner(c, an, a.cooling, rng, GlobalRecs{c.recs});
            uint64_t wa, wb;
            if (ABL == 3 || ABL == 4) {
                wa = (uint64_t)t.end_a * 0x100000001ull;
                wb = (uint64_t)t.end_b * 0x100000003ull;
            } else {
                wa = load_word<COORD_LOAD>(c.coords, t.end_a);
                wb = load_word<COORD_LOAD>(c.coords, t.end_b);
            }
            float dx, dy;
            if (FMT == kFmtQ32) {  // integer differences are exact: no cancellation at large coordinates
                dx = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * c.xf.inv_scale;
                dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * c.xf.inv_scale;
                guard |= in_frame_guard(wa) || in_frame_guard(wb);
            } else {
                dx = __uint_as_float((uint32_t)wa) - __uint_as_float((uint32_t)wb);
                dy = __uint_as_float((uint32_t)(wa >> 32)) - __uint_as_float((uint32_t)(wb >> 32));
            }
            float r_x, r_y, abs_delta;
            term_displacement(a.eta, t.pos_a, t.pos_b, dx, dy, r_x, r_y, abs_delta, hot_mu_cap(c, t.end_a, t.end_b));
            dmax = fmaxf(dmax, abs_delta);
            if (ABL == 1 || ABL == 4) {
                dmax = fmaxf(dmax, fabsf(r_x) + fabsf(r_y));  // keep the arithmetic alive
                continue;
            }
            if (UPD == kUpdStore && t.end_a == t.end_b) continue;  // the reference's two load/store pairs cancel
            if (FMT == kFmtQ32) {
                // stochastic rounding to quanta with 16+16 spare random bits: steps smaller than a
                // quantum still act in expectation; a moves by -(qx,qy), b by +(qx,qy), so the sum
                // of all coordinates is conserved exactly.
                const float ux = (float)(t.dither & 0xffffu) * (1.0f / 65536.0f);
                const float uy = (float)(t.dither >> 16) * (1.0f / 65536.0f);
                float fx = r_x * c.xf.scale;
                float fy = r_y * c.xf.scale;
                fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
                fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
                const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
                if (UPD == kUpdAtomic) {
                    // a step that rounds to no quantum adds zero: nothing to send (most far-apart terms of the
                    // late iterations, where eta / d^2 is tiny)
                    if ((qx | qy) == 0) continue;
                    const uint64_t delta = (uint64_t)qx + ((uint64_t)qy << 32);
                    // partner end first, anchor end second: the order of the grouped path and of the oracle mirror
                    atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + t.end_b), (unsigned long long)delta);
                    atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + t.end_a), (unsigned long long)(0 - delta));
                } else {
                    __hip_atomic_store(c.coords + t.end_b, q32_shift(wb, qx, qy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(c.coords + t.end_a, q32_shift(wa, -qx, -qy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (UPD == kUpdAtomic) {
                float* ca = reinterpret_cast<float*>(c.coords + t.end_a);
                float* cb = reinterpret_cast<float*>(c.coords + t.end_b);
                unsafeAtomicAdd(cb, r_x);
                unsafeAtomicAdd(cb + 1, r_y);
                unsafeAtomicAdd(ca, -r_x);
                unsafeAtomicAdd(ca + 1, -r_y);
            } else {
                const float ax = __uint_as_float((uint32_t)wa) + (-r_x), ay = __uint_as_float((uint32_t)(wa >> 32)) + (-r_y);
                const float bx = __uint_as_float((uint32_t)wb) + r_x, by = __uint_as_float((uint32_t)(wb >> 32)) + r_y;
                __hip_atomic_store(c.coords + t.end_b, pack_f32(bx, by), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c.coords + t.end_a, pack_f32(ax, ay), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    const uint64_t m = c.terms_per_anchor;
    const uint64_t n_groups = GROUPED ? (a.n_terms + m - 1) / m : 0;
    for (uint64_t grp = g; grp < n_groups; grp += L) {
        const uint64_t first = grp * m;
        const uint32_t mt = (uint32_t)(a.n_terms - first < m ? a.n_terms - first : m);
        const Anchor an = sample_anchor(c, pf, rng);
        const uint32_t node_a = an.rec.x & ~1u;  // word index of the anchor node's start end
        // private copy of the anchor node's two ends (loaded on first use) and what this group
        // moved each of them by; scalars and bit masks, not arrays: a dynamically indexed array
        // would live in scratch memory
        uint64_t la0 = 0, la1 = 0, dq0 = 0, dq1 = 0;
        float dfx0 = 0.0f, dfy0 = 0.0f, dfx1 = 0.0f, dfy1 = 0.0f;
        uint32_t have = 0, touched = 0;
        for (uint32_t r = 0; r < mt; ++r) {
            const Term t = sample_partner(c, an, a.cooling, rng, GlobalRecs{c.recs});
            const uint32_t ea = t.end_a & 1u, bit = 1u << ea;
            if (!(have & bit)) {
                const uint64_t w = (ABL == 3 || ABL == 4) ? (uint64_t)t.end_a * 0x100000001ull : load_word<COORD_LOAD>(c.coords, t.end_a);
                if (ea) la1 = w; else la0 = w;
                have |= bit;
            }
            const uint64_t wa = ea ? la1 : la0;
            uint64_t wb;
            if (ABL == 3 || ABL == 4) wb = (uint64_t)t.end_b * 0x100000003ull;
            else wb = load_word<COORD_LOAD>(c.coords, t.end_b);
            float dx, dy;
            if (FMT == kFmtQ32) {  // integer differences are exact: no cancellation at large coordinates
                dx = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * c.xf.inv_scale;
                dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * c.xf.inv_scale;
                guard |= in_frame_guard(wa) || in_frame_guard(wb);
            } else {
                dx = __uint_as_float((uint32_t)wa) - __uint_as_float((uint32_t)wb);
                dy = __uint_as_float((uint32_t)(wa >> 32)) - __uint_as_float((uint32_t)(wb >> 32));
            }
            float r_x, r_y, abs_delta;
            term_displacement(a.eta, t.pos_a, t.pos_b, dx, dy, r_x, r_y, abs_delta, hot_mu_cap(c, t.end_a, t.end_b));
            dmax = fmaxf(dmax, abs_delta);
            if (ABL == 1 || ABL == 4) {
                dmax = fmaxf(dmax, fabsf(r_x) + fabsf(r_y));  // keep the arithmetic alive
                continue;
            }
            // a term on one and the same node end: the reference's two load/store pairs cancel
            if (UPD == kUpdStore && t.end_a == t.end_b) continue;
            touched |= bit;
            uint64_t na;  // the anchor end after this term
            if (FMT == kFmtQ32) {
                // stochastic rounding to quanta with 16+16 spare random bits: steps smaller than a
                // quantum still act in expectation; a moves by -(qx,qy), b by +(qx,qy), so the sum
                // of all coordinates is conserved exactly.
                const float ux = (float)(t.dither & 0xffffu) * (1.0f / 65536.0f);
                const float uy = (float)(t.dither >> 16) * (1.0f / 65536.0f);
                float fx = r_x * c.xf.scale;
                float fy = r_y * c.xf.scale;
                fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
                fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
                const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
                if (UPD == kUpdAtomic) {
                    const uint64_t delta = (uint64_t)qx + ((uint64_t)qy << 32);
                    na = wa - delta;   // what the word will hold once the group's add has landed
                    if (ea) dq1 += delta; else dq0 += delta;
                    atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + t.end_b), (unsigned long long)delta);
                } else {
                    na = q32_shift(wa, -qx, -qy);
                    __hip_atomic_store(c.coords + t.end_b, q32_shift(wb, qx, qy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                const float ax = __uint_as_float((uint32_t)wa) + (-r_x), ay = __uint_as_float((uint32_t)(wa >> 32)) + (-r_y);
                na = pack_f32(ax, ay);
                if (UPD == kUpdAtomic) {
                    if (ea) { dfx1 += -r_x; dfy1 += -r_y; } else { dfx0 += -r_x; dfy0 += -r_y; }
                    float* cb = reinterpret_cast<float*>(c.coords + t.end_b);
                    unsafeAtomicAdd(cb, r_x);
                    unsafeAtomicAdd(cb + 1, r_y);
                } else {
                    const float bx = __uint_as_float((uint32_t)wb) + r_x, by = __uint_as_float((uint32_t)(wb >> 32)) + r_y;
                    __hip_atomic_store(c.coords + t.end_b, pack_f32(bx, by), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (ea) la1 = na; else la0 = na;
        }
        // the anchor side reaches memory once per touched end
        if (ABL != 1 && ABL != 4) {
            if (touched & 1u) {
                if (UPD == kUpdStore) __hip_atomic_store(c.coords + node_a, la0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (FMT == kFmtQ32) atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + node_a), (unsigned long long)(0 - dq0));
                else {
                    float* ca = reinterpret_cast<float*>(c.coords + node_a);
                    unsafeAtomicAdd(ca, dfx0);
                    unsafeAtomicAdd(ca + 1, dfy0);
                }
            }
            if (touched & 2u) {
                if (UPD == kUpdStore) __hip_atomic_store(c.coords + node_a + 1, la1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (FMT == kFmtQ32) atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + node_a + 1), (unsigned long long)(0 - dq1));
                else {
                    float* ca = reinterpret_cast<float*>(c.coords + node_a + 1);
                    unsafeAtomicAdd(ca, dfx1);
                    unsafeAtomicAdd(ca + 1, dfy1);
                }
            }
        }
    }
    c.rng[g] = rng.s0;
    c.rng[L + g] = rng.s1;
    c.rng[2 * L + g] = rng.s2;
    c.rng[3 * L + g] = rng.s3;
    // per-wavefront reduction of the early-stop quantity, one atomic per wave (:341-347)
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(c.delta_max_bits, __float_as_uint(dmax));
    if (__ballot(guard) && (threadIdx.x & 63) == 0) atomicOr(c.frame_flag, 1u);
}

// ---------------------------------------------------------------------------------------------
// The same kernel for the default configuration (fixed-point coordinates, terms_per_anchor = 1, fewer than 2^32 path
// steps, no hot-node cap), software-pipelined: a lane has FOUR terms in different stages at any time.
//
// The loop above is one dependent chain per term — zeta entry (L2) -> partner record (HBM) -> the two coordinate words
// -> two atomics — and a lane waits out each of the three loads in turn.  With the stream count bounded by the busiest
// node (auto_streams: 128 ... 5632 lanes on the reference's fixture graphs, on a chip that holds half a million) the
// lanes' own latency is the whole run time.  Here every trip of the loop
//   S4  finishes term j-3: its two coordinate words have arrived — displacement, rounding, the two atomics;
//   S3  takes term j-2's partner record, which has arrived, and requests the two coordinate words;
//   S2  takes term j-1's zeta entry and first record, draws nothing, computes the Zipf partner, requests its record;
//   S1  draws ALL of term j's variates, in the reference's order (path_sgd_layout.cpp:182,205-206,215/228,235-237,
//       253,262 — so a lane's stream is consumed exactly as above and trace_kernel still describes it), and requests
//       the first step's record and the zeta entry;
// so whatever a trip requests is consumed a trip later, in one place (gfx9 counts loads and stores in one counter: one
// full wait per trip, for requests that are a trip old).  Oldest stage first: a stage's registers are read before the
// younger stage overwrites them.  The coordinate words of a term are requested after the atomics of the term before it
// were issued (S4 runs before S3), so a lane's own updates are seen by its next term exactly as in the loop above: a
// one-stream run gives the same bits (tests: one-stream mirrors), and the window between reading an end and moving it
// — what the stream-count rule is about — holds one term per lane, as before.
struct PipeS1 {            // a term whose variates are drawn; first record and zeta entry on their way
    uint4 ra;
    double2 zd;
    double u;              // the Zipf variate (generate_canonical), a Zipf term only
    uint32_t s_rank, pstart;
    uint32_t aux;          // Zipf term: the jump length; otherwise the partner's rank in the path
    uint32_t flags;        // bit 0 valid, 1 Zipf partner, 2 backwards, 3 far end of the first step's node, 4 of the partner's
    uint32_t dither;
};
struct PipeS2 {            // partner chosen; its record on its way
    uint4 rb;
    uint64_t pos_a;
    uint32_t end_a, flags, dither;
};
struct PipeS3 {            // both ends known; their coordinate words on their way
    uint64_t wa, wb;
    float d;               // path distance of the two ends, (float)|pos_a - pos_b|
    uint32_t end_a, end_b, flags, dither;
};

// Tried and dropped (profiles/r03/per_lane_plain_vs_piped*.jsonl): two terms per stage — the second term of a stage reads
// its ends before the first one's atomics go out, so the window between reading an end and moving it holds two terms
// per lane, twice the concurrency the stream-count rule allows: DRB1-3123, LPA and chr6.C4 diverge; and issuing every
// load of the loop on every path (an empty slot asking for entry 0), which spares the compiler its waits inside the
// stages' branches but measured 10-20 % slower on three of the four fixture graphs than the loop below; and that variant
// with RETURNING atomics (which come back in order with the loads, so the compiler can wait for exactly the request a
// stage needs instead of for everything): the empty slots' atomics all add zero to word 0 — 1.4-3x slower.
template <bool PF_LDS, int COORD_LOAD, int UPD>
__global__ __launch_bounds__(kBlock) void sgd_iteration_kernel_piped(DevConst c, IterArgs a) {
    extern __shared__ uint64_t s_pf[];
    if (PF_LDS) {
        for (uint32_t i = threadIdx.x; i <= c.n_paths; i += blockDim.x) s_pf[i] = c.path_first[i];
        __syncthreads();
    }
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_streams) return;
    const uint64_t* pf = PF_LDS ? s_pf : c.path_first;
    Xoshiro256Plus rng;
    const size_t L = c.n_streams;
    rng.s0 = c.rng[g];
    rng.s1 = c.rng[L + g];
    rng.s2 = c.rng[2 * L + g];
    rng.s3 = c.rng[3 * L + g];
    float dmax = 0.0f;
    bool guard = false;
    const uint64_t n_mine = a.n_terms > g ? (a.n_terms - g + L - 1) / L : 0;  // terms g, g + L, ... of the iteration
    PipeS1 p1;
    PipeS2 p2;
    PipeS3 p3;
    p1.flags = p2.flags = p3.flags = 0;
    // (the generator state has to be here before the loop: the compiler would otherwise wait for it at its first use
    // inside the loop — a counted wait that, executed every trip, also waits for the loads the trip has just issued)
    asm volatile("" ::"v"(rng.s0), "v"(rng.s1), "v"(rng.s2), "v"(rng.s3));
    p1.ra = make_uint4(0, 0, 0, 0);
    p1.zd = make_double2(0.0, 0.0);
    p2.rb = make_uint4(0, 0, 0, 0);
    p3.wa = p3.wb = 0;
    for (uint64_t j = 0; j < n_mine + 3; ++j) {
        // The trip's one wait.  Everything the last trip requested is used here, by every lane, before this trip's atomics
        // go out: with a use only inside the stages' branches the compiler has to wait again after the atomics (a branch
        // may have been skipped), and that wait — loads and atomics share a counter — would be for the atomics' round trip.
        asm volatile("" ::"v"(p3.wa), "v"(p3.wb), "v"(p2.rb.x), "v"(p2.rb.w), "v"(p1.ra.x), "v"(p1.ra.w), "v"(p1.zd.x), "v"(p1.zd.y));
        // ---- S4: term j - 3 (path_sgd_layout.cpp:280-363) ----
        if (p3.flags & 1u) {
            const uint64_t wa = p3.wa, wb = p3.wb;
            const float dx0 = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * c.xf.inv_scale;  // exact integer differences
            const float dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * c.xf.inv_scale;
            guard |= in_frame_guard(wa) || in_frame_guard(wb);
            // term_displacement() with the distance already converted: the same operations
            float d = p3.d;
            if (d == 0.0f) d = 1e-9f;
            const float w = 1.0f / d;
            float mu = a.eta * w;
            if (mu > 1.0f) mu = 1.0f;
            float dx = dx0;
            if (dx == 0.0f) dx = 1e-9f;
            const float dx2 = dx * dx;
            const float dy2 = dy * dy;
            const float mag = sqrtf(dx2 + dy2);
            const float Delta = (mu * (mag - d)) / 2.0f;
            dmax = fmaxf(dmax, fabsf(Delta));
            const float r = Delta / mag;
            const float r_x = r * dx, r_y = r * dy;
            if (!(UPD == kUpdStore && p3.end_a == p3.end_b)) {  // (Hogwild stores: the reference's two load/store pairs cancel)
                const float ux = (float)(p3.dither & 0xffffu) * (1.0f / 65536.0f);
                const float uy = (float)(p3.dither >> 16) * (1.0f / 65536.0f);
                float fx = r_x * c.xf.scale;
                float fy = r_y * c.xf.scale;
                fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
                fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
                const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
                if (UPD == kUpdAtomic) {
                    if ((qx | qy) != 0) {  // a step that rounds to no quantum adds zero: nothing to send
                        const uint64_t delta = (uint64_t)qx + ((uint64_t)qy << 32);
                        atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + p3.end_b), (unsigned long long)delta);
                        atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + p3.end_a), (unsigned long long)(0 - delta));
                    }
                } else {
                    __hip_atomic_store(c.coords + p3.end_b, q32_shift(wb, qx, qy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(c.coords + p3.end_a, q32_shift(wa, -qx, -qy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ---- S3: term j - 2: the partner's record is here (:242-269), the coordinate words go out ----
        p3.flags = p2.flags;
        if (p2.flags & 1u) {
            const uint4 rb = p2.rb;
            uint64_t pos_b = (uint64_t)rb.z | ((uint64_t)rb.w << 32);
            uint32_t off_b = rb.x & 1u;
            if (p2.flags & 16u) { pos_b += rb.y; off_b ^= 1u; }
            p3.end_a = p2.end_a;
            p3.end_b = (rb.x & ~1u) | off_b;
            p3.dither = p2.dither;
            const int64_t diff = (int64_t)p2.pos_a - (int64_t)pos_b;
            p3.d = (float)(uint64_t)(diff < 0 ? -diff : diff);
            p3.wa = load_word<COORD_LOAD>(c.coords, p3.end_a);
            p3.wb = load_word<COORD_LOAD>(c.coords, p3.end_b);
        }
        // ---- S2: term j - 1: zeta entry and first record are here; the partner (:207-237), its record goes out ----
        p2.flags = p1.flags;
        if (p1.flags & 1u) {
            uint32_t b_rank = p1.aux;
            if (p1.flags & 2u) {
                const uint32_t z = (uint32_t)zipf_tabled_u(p1.u, c.zc, p1.aux, p1.zd.x, p1.zd.y);  // (z <= the jump length < 2^32)
                b_rank = (p1.flags & 4u) ? p1.s_rank - z : p1.s_rank + z;
            }
            p2.rb = c.recs[(uint64_t)p1.pstart + b_rank];
            const uint4 ra = p1.ra;
            uint64_t pos_a = (uint64_t)ra.z | ((uint64_t)ra.w << 32);
            uint32_t off_a = ra.x & 1u;
            if (p1.flags & 8u) { pos_a += ra.y; off_a ^= 1u; }
            p2.pos_a = pos_a;
            p2.end_a = (ra.x & ~1u) | off_a;
            p2.dither = p1.dither;
        }
        // ---- S1: term j: every variate, in the reference's order; first record and zeta entry go out ----
        p1.flags = 0;
        if (j < n_mine) {
            uint32_t k, pstart, cnt;
            do {  // :182-192 — a single-step path makes the reference draw again without counting a term
                k = (uint32_t)uniform_below(rng, c.n_steps);
                const uint32_t p = find_path(pf, c.n_paths, (uint64_t)k);
                pstart = (uint32_t)pf[p];
                cnt = (uint32_t)(pf[p + 1] - pf[p]);
            } while (cnt == 1);
            p1.ra = c.recs[k];
            p1.s_rank = k - pstart;
            p1.pstart = pstart;
            uint32_t flags = 1u;
            if (a.cooling || coin(rng)) {                                                       // :205
                const bool back = (p1.s_rank > 0 && coin(rng)) || p1.s_rank == cnt - 1;         // :206
                const uint32_t room = back ? p1.s_rank : cnt - p1.s_rank - 1;
                const uint32_t jump = c.space < room ? (uint32_t)c.space : room;
                p1.zd = c.zeta_denom[zeta_index(jump, c.space_max, c.space_quant)];
                p1.u = canonical(rng);
                p1.aux = jump;
                flags |= 2u | (back ? 4u : 0u);
            } else {
                p1.aux = (uint32_t)uniform_below(rng, (uint64_t)cnt);                           // :235-237
            }
            const uint64_t draw_a = rng.next(), draw_b = rng.next();                            // :253, :262
            flags |= (uint32_t)(draw_a >> 63) << 3 | (uint32_t)(draw_b >> 63) << 4;
            p1.dither = (uint32_t)draw_a;
            p1.flags = flags;
        }
    }
    c.rng[g] = rng.s0;
    c.rng[L + g] = rng.s1;
    c.rng[2 * L + g] = rng.s2;
    c.rng[3 * L + g] = rng.s3;
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(c.delta_max_bits, __float_as_uint(dmax));
    if (__ballot(guard) && (threadIdx.x & 63) == 0) atomicOr(c.frame_flag, 1u);
}

// ---------------------------------------------------------------------------------------------
// Small lane-bound graphs: the iteration in two passes, sampling apart from moving, the layout in one CU's LDS.
//
// On a graph whose busiest node bounds the lanes to a few hundred or thousand (auto_streams: every fixture graph of
// the reference) a lane is alone on its SIMD, and what a trip of the loops above costs is not memory but the wave's
// own instruction stream: ~700 vector instructions per term, most of them the sampler (generator steps on 32-bit
// lanes, the path search, two fp64 divisions and the approximate pow of the Zipf draw) — 1.8 us per term and lane with
// 128 lanes whatever the memory system does (the single-pass kernel with the whole layout in LDS measured 92 ms
// against 98 on DRB1-3123_unsorted: profiles/r03/split_experiments.txt) — and then the round trip of two atomics to L2.
// But only MOVING node ends is bounded by the busiest node; the sampler reads nothing a term writes.  And such a graph
// is small: its 2N coordinate words (16 bytes per node) fit the 160 KB of LDS a gfx950 compute unit has.  So:
//   sample_terms_kernel          every stream the GPU holds (not the lanes the graph allows) draws its terms of the
//                                iteration — the per-lane kernel's streams, draws and arithmetic (sample_anchor /
//                                sample_partner, the reference's order: trace_kernel describes them) — and writes each
//                                as a 16-byte record {end a, end b, path distance, dither} at the term's index;
//   apply_terms_resident_kernel  ONE workgroup with the lanes the graph allows stages the coordinates in LDS; lane l
//                                takes terms l, l + L, ... in order: a coalesced 16-byte load requested four trips
//                                ahead (nothing else comes from global memory — no store, no atomic shares the
//                                counter — so the records are waited for one by one), two LDS loads, the
//                                displacement, two LDS atomics; a hundred-odd instructions and a round trip of a
//                                hundred-odd cycles per term; the coordinates go back with plain stores (the only
//                                writer).
// One stream and one lane are the sequential program: the same bits as the loops above (tests).  Moving the ends in
// global memory from the term records (the same two passes for graphs beyond the LDS) was built and dropped: a trip
// is then the round trip of the atomics plus that of the streamed records, slower than the pipelined loop above from
// 1 500 lanes on (profiles/r03/split_experiments.txt).
struct TermRec {  // 16 bytes
    uint32_t end_a, end_b;  // end_a = kNoEnd: no term
    float d;                // path distance of the two ends, (float)|pos_a - pos_b|
    uint32_t dither;
};
constexpr uint32_t kNoEnd = 0xffffffffu;

template <bool PF_LDS>
__global__ __launch_bounds__(kBlock) void sample_terms_kernel(DevConst c, uint32_t cooling, uint64_t n_terms, uint4* out) {
    extern __shared__ uint64_t s_pf[];
    if (PF_LDS) {
        for (uint32_t i = threadIdx.x; i <= c.n_paths; i += blockDim.x) s_pf[i] = c.path_first[i];
        __syncthreads();
    }
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_streams) return;
    const uint64_t* pf = PF_LDS ? s_pf : c.path_first;
    Xoshiro256Plus rng;
    const size_t L = c.n_streams;
    rng.s0 = c.rng[g];
    rng.s1 = c.rng[L + g];
    rng.s2 = c.rng[2 * L + g];
    rng.s3 = c.rng[3 * L + g];
    for (uint64_t q = g; q < n_terms; q += L) {  // terms g, g + L, ... of the call, as in sgd_iteration_kernel
        const Anchor an = sample_anchor(c, pf, rng);
        const Term t = sample_partner(c, an, cooling, rng, GlobalRecs{c.recs});
        const int64_t diff = (int64_t)t.pos_a - (int64_t)t.pos_b;
        const float d = (float)(uint64_t)(diff < 0 ? -diff : diff);
        out[q] = make_uint4(t.end_a, t.end_b, __float_as_uint(d), t.dither);
    }
    c.rng[g] = rng.s0;
    c.rng[L + g] = rng.s1;
    c.rng[2 * L + g] = rng.s2;
    c.rng[3 * L + g] = rng.s3;
}

// What a term adds to its partner's coordinate word (the first end gets the negative): term_displacement() and the
// stochastic rounding of sgd_iteration_kernel, the same operations in the same order.
__device__ __forceinline__ uint64_t term_step_q32(const DevConst& c, float eta, uint64_t wa, uint64_t wb, float d, uint32_t dither, float& dmax) {
    const float dx0 = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * c.xf.inv_scale;  // exact integer differences
    const float dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * c.xf.inv_scale;
    if (d == 0.0f) d = 1e-9f;
    const float w = 1.0f / d;
    float mu = eta * w;
    if (mu > 1.0f) mu = 1.0f;
    float dx = dx0;
    if (dx == 0.0f) dx = 1e-9f;
    const float dx2 = dx * dx;
    const float dy2 = dy * dy;
    const float mag = sqrtf(dx2 + dy2);
    const float Delta = (mu * (mag - d)) / 2.0f;
    dmax = fmaxf(dmax, fabsf(Delta));
    const float r = Delta / mag;
    const float r_x = r * dx, r_y = r * dy;
    const float ux = (float)(dither & 0xffffu) * (1.0f / 65536.0f);
    const float uy = (float)(dither >> 16) * (1.0f / 65536.0f);
    float fx = r_x * c.xf.scale;
    float fy = r_y * c.xf.scale;
    fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
    fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
    const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
    return (uint64_t)qx + ((uint64_t)qy << 32);  // 0: the step rounds to no quantum
}

// One workgroup, the 2N coordinate words in its LDS; term records requested kResidentAhead trips before they are used.
constexpr int kResidentBlock = 1024;
constexpr int kResidentAhead = 4;
__global__ __launch_bounds__(kResidentBlock) void apply_terms_resident_kernel(DevConst c, IterArgs a, const uint4* terms, uint32_t lanes) {
    extern __shared__ uint64_t s_win[];
    const uint32_t n_ends = 2 * c.n_nodes;
    for (uint32_t i = threadIdx.x; i < n_ends; i += blockDim.x) s_win[i] = c.coords[i];
    __syncthreads();
    const uint32_t l = threadIdx.x;
    const uint64_t n_mine = l < lanes && a.n_terms > l ? (a.n_terms - l + lanes - 1) / lanes : 0;
    float dmax = 0.0f;
    if (n_mine) {  // (lanes differ by at most one trip: nothing in the loop is wave-cooperative)
        uint4 t[kResidentAhead];
#pragma unroll
        for (int k = 0; k < kResidentAhead; ++k) t[k] = terms[l + (uint64_t)(k < (int64_t)n_mine ? k : 0) * lanes];
        for (uint64_t j = 0; j < n_mine; j += kResidentAhead) {
#pragma unroll
            for (int k = 0; k < kResidentAhead; ++k) {
                const uint4 r = t[k];
                const uint64_t nxt = j + k + kResidentAhead;
                t[k] = terms[l + (nxt < n_mine ? nxt : 0) * lanes];  // (every slot asks: the requests in flight are the same on every path)
                if (j + k < n_mine) {
                    const uint64_t wa = s_win[r.x], wb = s_win[r.y];
                    const uint64_t delta = term_step_q32(c, a.eta, wa, wb, __uint_as_float(r.z), r.w, dmax);
                    if (delta != 0) {
                        atomicAdd(reinterpret_cast<unsigned long long*>(s_win + r.y), (unsigned long long)delta);
                        atomicAdd(reinterpret_cast<unsigned long long*>(s_win + r.x), (unsigned long long)(0 - delta));
                    }
                }
            }
        }
    }
    __syncthreads();
    bool guard = false;
    for (uint32_t i = threadIdx.x; i < n_ends; i += blockDim.x) {  // the only writer since the words were staged: plain stores
        const uint64_t w = s_win[i];
        c.coords[i] = w;
        guard |= in_frame_guard(w);
    }
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(c.delta_max_bits, __float_as_uint(dmax));
    if (__ballot(guard) && (threadIdx.x & 63) == 0) atomicOr(c.frame_flag, 1u);
}

// ---------------------------------------------------------------------------------------------
// 1D path-guided SGD of `odgi sort -Y` (reference src/algorithms/path_sgd.cpp:12-500): the layout's
// sibling — same first-step/partner sampler, one coordinate per node, no end choice.  Differences:
// the Zipf draw uses adj_theta = 0.001 once cooling starts while the zeta cache keeps the user's theta
// (:127-137,195,246); a term of path distance 0 is dropped uncounted (:320-323); with target sorting a frozen
// node does not move, and a term between two frozen nodes is counted but does nothing (:289-301).
// Coordinates: one signed 64-bit fixed-point word per node, x = q * inv_scale, moved with one 64-bit
// integer atomic per node; arithmetic in fp64 like the reference.
struct SortArgs {
    long long* X;
    const uint8_t* frozen;  // [n_nodes] target nodes of `odgi sort -Y --target-paths`, or null
    double scale, inv_scale;
    ZipfConst zc_cool;  // theta = 0.001
    uint64_t n_terms;
    double eta;
    uint32_t cooling;
};

struct Term1D {
    uint64_t ka, kb, pos_a, pos_b;
    uint32_t node_a, node_b;
    bool move_a, move_b;
};

template <typename PF>
__device__ __forceinline__ Term1D sample_term_1d(const DevConst& c, const PF pf, const SortArgs& sa, Xoshiro256Plus& rng) {
    Term1D t;
    for (;;) {
        const Anchor a = sample_anchor(c, pf, rng);
        uint64_t b_rank;
        if (sa.cooling || coin(rng)) {                                            // path_sgd.cpp:245
            const bool back = (a.s_rank > 0 && coin(rng)) || a.s_rank == a.cnt - 1;  // :247
            const uint64_t room = back ? a.s_rank : a.cnt - a.s_rank - 1;
            const uint64_t jump = c.space < room ? c.space : room;
            const double zeta_n = c.zetas[zeta_index(jump, c.space_max, c.space_quant)];
            const uint64_t z = zipf(rng, sa.cooling ? sa.zc_cool : c.zc, jump, zeta_n);
            b_rank = back ? a.s_rank - z : a.s_rank + z;
        } else {
            b_rank = uniform_below(rng, a.cnt);                                   // :275-277
        }
        t.ka = a.k;
        t.kb = a.pstart + b_rank;
        const uint4 rb = c.recs[t.kb];
        t.pos_a = (uint64_t)a.rec.z | ((uint64_t)a.rec.w << 32);
        t.pos_b = (uint64_t)rb.z | ((uint64_t)rb.w << 32);
        t.node_a = a.rec.x >> 1;
        t.node_b = rb.x >> 1;
        t.move_a = !(sa.frozen && sa.frozen[t.node_a]);                           // :289-296
        t.move_b = !(sa.frozen && sa.frozen[t.node_b]);
        if (!t.move_a && !t.move_b) return t;                                     // counted, nothing to move (:297-301)
        if (t.pos_a != t.pos_b) return t;                                         // :320-323
    }
}

template <bool PF_LDS>
__global__ __launch_bounds__(kBlock) void sort_iteration_kernel(DevConst c, SortArgs sa) {
    extern __shared__ uint64_t s_pf[];
    if (PF_LDS) {
        for (uint32_t i = threadIdx.x; i <= c.n_paths; i += blockDim.x) s_pf[i] = c.path_first[i];
        __syncthreads();
    }
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_streams) return;
    const uint64_t* pf = PF_LDS ? s_pf : c.path_first;
    Xoshiro256Plus rng;
    const size_t L = c.n_streams;
    rng.s0 = c.rng[g];
    rng.s1 = c.rng[L + g];
    rng.s2 = c.rng[2 * L + g];
    rng.s3 = c.rng[3 * L + g];
    float dmax = 0.0f;
    for (uint64_t ti = g; ti < sa.n_terms; ti += L) {
        const Term1D t = sample_term_1d(c, pf, sa, rng);
        if (!t.move_a && !t.move_b) continue;
        const long long qa = __hip_atomic_load(sa.X + t.node_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long qb = __hip_atomic_load(sa.X + t.node_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t diff = (int64_t)t.pos_a - (int64_t)t.pos_b;
        const double term_dist = (double)(uint64_t)(diff < 0 ? -diff : diff);     // :316-318
        double mu = sa.eta * (1.0 / term_dist);                                   // :328-332
        if (mu > 1.0) mu = 1.0;
        double dx = (double)(qa - qb) * sa.inv_scale;
        if (dx == 0.0) dx = 1e-9;
        const double mag = fabs(dx);
        const double Delta = mu * (mag - term_dist) / 2.0;                        // :355
        const double r_x = (Delta / mag) * dx;
        const long long dq = __double2ll_rn(r_x * sa.scale);
        dmax = fmaxf(dmax, (float)fabs(Delta));
        if (dq == 0) continue;  // a step below half a quantum adds zero: nothing to send
        if (t.move_b) atomicAdd(reinterpret_cast<unsigned long long*>(sa.X + t.node_b), (unsigned long long)dq);
        if (t.move_a) atomicAdd(reinterpret_cast<unsigned long long*>(sa.X + t.node_a), (unsigned long long)(-dq));
    }
    c.rng[g] = rng.s0;
    c.rng[L + g] = rng.s1;
    c.rng[2 * L + g] = rng.s2;
    c.rng[3 * L + g] = rng.s3;
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(c.delta_max_bits, __float_as_uint(dmax));
}

// The two passes of small lane-bound graphs (sample_terms_kernel / apply_terms_resident_kernel above), 1D.  A term record is
// {node a, node b, path distance as fp64}; bit 31 of a node word says the node is frozen (target sorting).
constexpr uint32_t kFrozenBit = 0x80000000u;
template <bool PF_LDS>
__global__ __launch_bounds__(kBlock) void sort_sample_terms_kernel(DevConst c, SortArgs sa, uint4* out) {
    extern __shared__ uint64_t s_pf[];
    if (PF_LDS) {
        for (uint32_t i = threadIdx.x; i <= c.n_paths; i += blockDim.x) s_pf[i] = c.path_first[i];
        __syncthreads();
    }
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_streams) return;
    const uint64_t* pf = PF_LDS ? s_pf : c.path_first;
    Xoshiro256Plus rng;
    const size_t L = c.n_streams;
    rng.s0 = c.rng[g];
    rng.s1 = c.rng[L + g];
    rng.s2 = c.rng[2 * L + g];
    rng.s3 = c.rng[3 * L + g];
    for (uint64_t ti = g; ti < sa.n_terms; ti += L) {
        const Term1D t = sample_term_1d(c, pf, sa, rng);
        const int64_t diff = (int64_t)t.pos_a - (int64_t)t.pos_b;
        const uint64_t dist = (uint64_t)__double_as_longlong((double)(uint64_t)(diff < 0 ? -diff : diff));  // :316-318
        out[ti] = make_uint4(t.node_a | (t.move_a ? 0u : kFrozenBit), t.node_b | (t.move_b ? 0u : kFrozenBit), (uint32_t)dist, (uint32_t)(dist >> 32));
    }
    c.rng[g] = rng.s0;
    c.rng[L + g] = rng.s1;
    c.rng[2 * L + g] = rng.s2;
    c.rng[3 * L + g] = rng.s3;
}

// one workgroup, the N position words in its LDS; lane l moves the nodes of terms l, l + lanes, ... in order: the
// arithmetic of sort_iteration_kernel (path_sgd.cpp:316-363)
__global__ __launch_bounds__(kResidentBlock) void sort_apply_terms_resident_kernel(DevConst c, SortArgs sa, const uint4* terms, uint32_t lanes) {
    extern __shared__ uint64_t s_win[];
    long long* x = reinterpret_cast<long long*>(s_win);
    for (uint32_t i = threadIdx.x; i < c.n_nodes; i += blockDim.x) x[i] = sa.X[i];
    __syncthreads();
    const uint32_t l = threadIdx.x;
    const uint64_t n_mine = l < lanes && sa.n_terms > l ? (sa.n_terms - l + lanes - 1) / lanes : 0;
    float dmax = 0.0f;
    if (n_mine) {
        uint4 t[kResidentAhead];
#pragma unroll
        for (int k = 0; k < kResidentAhead; ++k) t[k] = terms[l + (uint64_t)(k < (int64_t)n_mine ? k : 0) * lanes];
        for (uint64_t j = 0; j < n_mine; j += kResidentAhead) {
#pragma unroll
            for (int k = 0; k < kResidentAhead; ++k) {
                const uint4 r = t[k];
                const uint64_t nxt = j + k + kResidentAhead;
                t[k] = terms[l + (nxt < n_mine ? nxt : 0) * lanes];  // (every slot asks: the requests in flight are the same on every path)
                const bool move_a = !(r.x & kFrozenBit), move_b = !(r.y & kFrozenBit);
                if (j + k < n_mine && (move_a || move_b)) {
                    const uint32_t node_a = r.x & ~kFrozenBit, node_b = r.y & ~kFrozenBit;
                    const long long qa = x[node_a], qb = x[node_b];
                    const double term_dist = __longlong_as_double((long long)((uint64_t)r.z | ((uint64_t)r.w << 32)));
                    double mu = sa.eta * (1.0 / term_dist);                                   // :328-332
                    if (mu > 1.0) mu = 1.0;
                    double dx = (double)(qa - qb) * sa.inv_scale;
                    if (dx == 0.0) dx = 1e-9;
                    const double mag = fabs(dx);
                    const double Delta = mu * (mag - term_dist) / 2.0;                        // :355
                    const double r_x = (Delta / mag) * dx;
                    const long long dq = __double2ll_rn(r_x * sa.scale);
                    dmax = fmaxf(dmax, (float)fabs(Delta));
                    if (dq != 0) {
                        if (move_b) atomicAdd(reinterpret_cast<unsigned long long*>(x + node_b), (unsigned long long)dq);
                        if (move_a) atomicAdd(reinterpret_cast<unsigned long long*>(x + node_a), (unsigned long long)(-dq));
                    }
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < c.n_nodes; i += blockDim.x) sa.X[i] = x[i];
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(c.delta_max_bits, __float_as_uint(dmax));
}

template <bool PF_LDS>
__global__ __launch_bounds__(kBlock) void sort_trace_kernel(DevConst c, SortArgs sa, uint64_t seed_base, uint64_t terms_per_stream, uint64_t* out) {
    extern __shared__ uint64_t s_pf[];
    if (PF_LDS) {
        for (uint32_t i = threadIdx.x; i <= c.n_paths; i += blockDim.x) s_pf[i] = c.path_first[i];
        __syncthreads();
    }
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_streams) return;
    const uint64_t* pf = PF_LDS ? s_pf : c.path_first;
    Xoshiro256Plus rng;
    rng.seed(seed_base + g);
    for (uint64_t j = 0; j < terms_per_stream; ++j) {
        const Term1D t = sample_term_1d(c, pf, sa, rng);
        uint64_t* o = out + (j * (uint64_t)c.n_streams + g) * 2;
        o[0] = t.ka;
        o[1] = t.kb;
    }
}

__global__ void sort_pack_kernel(const double* X, uint64_t n, double scale, long long* W) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) W[i] = __double2ll_rn(X[i] * scale);
}
__global__ void sort_unpack_kernel(const long long* W, uint64_t n, double inv_scale, double* X) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) X[i] = (double)W[i] * inv_scale;
}

// sampler-only launch for parity checks: fresh streams, nothing is modified
template <bool PF_LDS>
__global__ __launch_bounds__(kBlock) void trace_kernel(DevConst c, uint32_t cooling, uint64_t seed_base,
                                                       uint64_t terms_per_stream, uint64_t* out) {
    extern __shared__ uint64_t s_pf[];
    if (PF_LDS) {
        for (uint32_t i = threadIdx.x; i <= c.n_paths; i += blockDim.x) s_pf[i] = c.path_first[i];
        __syncthreads();
    }
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= c.n_streams) return;
    const uint64_t* pf = PF_LDS ? s_pf : c.path_first;
    Xoshiro256Plus rng;
    rng.seed(seed_base + g);
    Anchor an{};
    for (uint64_t j = 0; j < terms_per_stream; ++j) {
        if (j % c.terms_per_anchor == 0) an = sample_anchor(c, pf, rng);
        const Term t = sample_partner(c, an, cooling, rng, GlobalRecs{c.recs});
        uint64_t* o = out + (j * (uint64_t)c.n_streams + g) * 4;
        o[0] = an.k;
        o[1] = t.kb;
        o[2] = t.end_a & 1u;
        o[3] = t.end_b & 1u;
    }
}

__global__ void seed_streams_kernel(uint64_t* rng, uint32_t n_streams, uint64_t seed_base) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_streams) return;
    Xoshiro256Plus r;
    r.seed(seed_base + g);
    const size_t L = n_streams;
    rng[g] = r.s0;
    rng[L + g] = r.s1;
    rng[2 * L + g] = r.s2;
    rng[3 * L + g] = r.s3;
}

// SoA view -> 16-byte step records, on the device (the gather of node_len happens once, here)
// ---- index build on the device (the reference walks its paths under OpenMP, src/cuda/layout.cu:371-410) ----
// path steps on every node (the hot-node rule and the outbox pool shares need them) and the range check of the handles
__global__ void node_steps_kernel(const uint32_t* step_handle, uint64_t n_steps, uint32_t n_nodes, uint32_t* node_steps, unsigned int* bad_handle) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_steps; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = step_handle[k] >> 1;
        if (r >= n_nodes) atomicOr(bad_handle, 1u);
        else atomicAdd(node_steps + r, 1u);
    }
}

// Per tile of consecutive path steps: lowest and highest node rank and the most visits of one node.  One wavefront per
// tile; the tile's ranks sit in LDS and every lane counts the occurrences of its ranks (T^2 / 64 compares per lane:
// 900 at T = 224).  out[3 * tile + {0,1,2}] = {rmin, rmax, maxmult}.
__global__ __launch_bounds__(256) void tile_stats_kernel(const uint32_t* step_handle, const uint64_t* tile_t0, const uint32_t* tile_n, uint64_t n_tiles,
                                                         uint32_t tile_cap, uint32_t* out) {
    extern __shared__ uint32_t s_ranks[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* ranks = s_ranks + (size_t)wave * tile_cap;
    for (uint64_t t = (uint64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (uint64_t)gridDim.x * 4) {
        const uint64_t t0 = tile_t0[t];
        const uint32_t n = tile_n[t];
        uint32_t rmin = 0xffffffffu, rmax = 0;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t r = step_handle[t0 + i] >> 1;
            ranks[i] = r;
            rmin = r < rmin ? r : rmin;
            rmax = r > rmax ? r : rmax;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t mult = 1;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t r = ranks[i];
            uint32_t c = 0;
            for (uint32_t j = 0; j < n; ++j) c += ranks[j] == r ? 1u : 0u;
            mult = c > mult ? c : mult;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t a = __shfl_xor(rmin, off), b = __shfl_xor(rmax, off), m = __shfl_xor(mult, off);
            rmin = a < rmin ? a : rmin;
            rmax = b > rmax ? b : rmax;
            mult = m > mult ? m : mult;
        }
        if (lane == 0) {
            out[3 * t] = rmin;
            out[3 * t + 1] = rmax;
            out[3 * t + 2] = mult;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The tile kernel's gather records (recs2).  A step has a static 16-byte piece — the record {handle, length, position} — and a
// 16-byte snapshot piece, the coordinate words of its node's two ends (the end the step enters first in front).  They are laid
// out in groups of FOUR steps per 128-byte line: [s0 s1 s2 s3 | n0 n1 n2 n3] — static pieces in the first 64-byte unit,
// snapshot pieces in the second.  A partner outside the tile still costs ONE line (a read request moves a whole 128-byte line:
// profiles/r04/pmc_calibration.json), and whoever refreshes snapshots — a tile for its own steps, snapshot_kernel for all of
// them — writes whole 64-byte units of snapshot pieces only: half the bytes of the round-4 layout ({s, n} pairs of 32 bytes,
// where leaving the static halves out would have cost a read-for-ownership of every unit).
__host__ __device__ inline uint64_t recs2_pieces(uint64_t n_steps) { return ((n_steps + 3) / 4) * 8; }   // uint4 pieces in the array
__host__ __device__ inline uint64_t recs2_static_piece(uint64_t k) { return ((k >> 2) << 3) | (k & 3u); }
__host__ __device__ inline uint64_t recs2_snap_piece(uint64_t k) { return ((k >> 2) << 3) | 4u | (k & 3u); }
// recs2 (tile kernel only, may be null): the static pieces are the same records; the snapshot pieces are filled by
// snapshot_kernel before the first tile launch
// bad_handle: set when a step names a node rank outside the graph (the caller's arrays are not trusted: such a step
// would index node_len and, later, the coordinates out of bounds)
__global__ void build_step_records(const uint32_t* step_handle, const uint64_t* step_pos, const uint32_t* node_len, uint32_t n_nodes,
                                   uint64_t n_steps, uint4* recs, uint4* recs2, unsigned int* bad_handle) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_steps; k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = step_handle[k];
        if ((h >> 1) >= n_nodes) {
            atomicOr(bad_handle, 1u);
            h = 0;
        }
        const uint64_t pos = step_pos[k];
        const uint4 r = make_uint4(h, node_len[h >> 1], (uint32_t)pos, (uint32_t)(pos >> 32));
        recs[k] = r;
        if (recs2) {
            recs2[recs2_static_piece(k)] = r;
            recs2[recs2_snap_piece(k)] = make_uint4(0, 0, 0, 0);
        }
    }
}

// ---- step positions on the device (the reference's GPU route flattens paths in parallel, src/cuda/layout.cu:371-410) ----
// step_pos[k] = bp offset of step k inside its path (xp.cpp:607-617) = an exclusive prefix sum of node_len[step_handle >> 1] that
// starts again at every path.  A caller that hands over no step_pos (pgsgd_graph_view::step_pos == NULL) uploads 4 bytes per step
// instead of 12; the session builds the positions from the handles: (1) sums of tiles of kPosTile steps, (2) their exclusive scan,
// (3) the prefix of every step over ALL steps, (4) minus the prefix at its path's first step.
constexpr uint32_t kPosItems = 16, kPosBlock = 256, kPosTile = kPosItems * kPosBlock;
__global__ __launch_bounds__(kPosBlock) void step_len_tile_sums(const uint32_t* step_handle, const uint32_t* node_len, uint64_t n_steps, uint64_t n_tiles, uint64_t* tile_sum) {
    __shared__ uint64_t part[kPosBlock / 64];
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t k0 = t * kPosTile + (uint64_t)threadIdx.x * kPosItems;
        uint64_t sum = 0;
        for (uint32_t i = 0; i < kPosItems; ++i)
            if (k0 + i < n_steps) sum += node_len[step_handle[k0 + i] >> 1];
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t tot = 0;
            for (uint32_t w = 0; w < kPosBlock / 64; ++w) tot += part[w];
            tile_sum[t] = tot;
        }
        __syncthreads();
    }
}
// exclusive scan of the tile sums in place: one workgroup, chunks of 1024 (1.1e5 tiles at 4.7e8 steps)
__global__ __launch_bounds__(1024) void scan_tile_sums(uint64_t* tile_sum, uint64_t n_tiles) {
    __shared__ uint64_t buf[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_tiles; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n_tiles ? tile_sum[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
            const uint64_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_tiles) tile_sum[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(kPosBlock) void step_prefix_kernel(const uint32_t* step_handle, const uint32_t* node_len, uint64_t n_steps, uint64_t n_tiles, const uint64_t* tile_off,
                                                                uint64_t* prefix) {
    __shared__ uint64_t part[kPosBlock];
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t k0 = t * kPosTile + (uint64_t)threadIdx.x * kPosItems;
        uint32_t len[kPosItems];
        uint64_t sum = 0;
        for (uint32_t i = 0; i < kPosItems; ++i) {
            len[i] = k0 + i < n_steps ? node_len[step_handle[k0 + i] >> 1] : 0u;
            sum += len[i];
        }
        part[threadIdx.x] = sum;
        __syncthreads();
        for (uint32_t off = 1; off < kPosBlock; off <<= 1) {
            const uint64_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        uint64_t pos = tile_off[t] + part[threadIdx.x] - sum;
        for (uint32_t i = 0; i < kPosItems; ++i) {
            if (k0 + i < n_steps) prefix[k0 + i] = pos;
            pos += len[i];
        }
        __syncthreads();
    }
}
// the path of step k: the last p with path_first[p] <= k (empty paths share their first step with the next path)
__device__ __forceinline__ uint32_t path_of_step(const uint64_t* path_first, uint32_t n_paths, uint64_t k) {
    uint32_t lo = 0, hi = n_paths;   // path_first[lo] <= k < path_first[hi]
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (path_first[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}
__global__ void path_base_kernel(const uint64_t* prefix, const uint64_t* path_first, uint32_t n_paths, uint64_t n_steps, uint64_t* base) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_paths) base[p] = path_first[p] < n_steps ? prefix[path_first[p]] : 0;
}
__global__ void step_pos_rebase_kernel(uint64_t* prefix, uint64_t n_steps, const uint64_t* path_first, uint32_t n_paths, const uint64_t* base) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_steps; k += (uint64_t)gridDim.x * blockDim.x)
        prefix[k] -= base[path_of_step(path_first, n_paths, k)];
}
// out[i] = values[index[i]] (the few positions the host needs back: the last step of every path, the layout check's pairs)
__global__ void gather_u64_kernel(const uint64_t* values, const uint64_t* index, uint64_t n, uint64_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = values[index[i]];
}

__device__ __forceinline__ uint32_t quantize(float v, double off, float scale) {
    double q = rint(((double)v - off) * (double)scale);
    q = q < 0.0 ? 0.0 : (q > 4294967295.0 ? 4294967295.0 : q);
    return (uint32_t)q;
}
__device__ __forceinline__ float dequantize(uint32_t q, double off, float inv_scale) {
    return (float)(off + (double)q * (double)inv_scale);
}

// host X[2N],Y[2N] (staged on the device) <-> coordinate words
template <int FMT>
__global__ void pack_coords(const float* X, const float* Y, uint64_t n_ends, Xform xf, uint64_t* coords) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ends; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo, hi;
        if (FMT == kFmtQ32) {
            lo = quantize(X[i], xf.x_off, xf.scale);
            hi = quantize(Y[i], xf.y_off, xf.scale);
        } else {
            lo = __float_as_uint(X[i]);
            hi = __float_as_uint(Y[i]);
        }
        coords[i] = (uint64_t)lo | ((uint64_t)hi << 32);
    }
}
template <int FMT>
__global__ void unpack_coords(const uint64_t* coords, uint64_t n_ends, Xform xf, float* X, float* Y) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ends; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = coords[i];
        if (FMT == kFmtQ32) {
            X[i] = dequantize((uint32_t)w, xf.x_off, xf.inv_scale);
            Y[i] = dequantize((uint32_t)(w >> 32), xf.y_off, xf.inv_scale);
        } else {
            X[i] = __uint_as_float((uint32_t)w);
            Y[i] = __uint_as_float((uint32_t)(w >> 32));
        }
    }
}

// Multi-GPU exchange, step 1: buf[2e], buf[2e+1] = what this rank moved node end e by since the
// last exchange (bp), buf[4N + e] = squared length of that move.  One fused buffer, one all-reduce.
template <int FMT>
__global__ void exchange_prepare_kernel(const uint64_t* coords, const uint64_t* base, uint64_t n_ends, Xform xf, float* buf) {
    float2* S = reinterpret_cast<float2*>(buf);
    float* Q = buf + 2 * n_ends;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ends; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t c = coords[i], b = base[i];
        float dx, dy;
        if (FMT == kFmtQ32) {
            dx = (float)((int64_t)(uint32_t)c - (int64_t)(uint32_t)b) * xf.inv_scale;
            dy = (float)((int64_t)(c >> 32) - (int64_t)(b >> 32)) * xf.inv_scale;
        } else {
            dx = __uint_as_float((uint32_t)c) - __uint_as_float((uint32_t)b);
            dy = __uint_as_float((uint32_t)(c >> 32)) - __uint_as_float((uint32_t)(b >> 32));
        }
        S[i] = make_float2(dx, dy);
        Q[i] = dx * dx + dy * dy;
    }
}

// the tail of the fused buffer: a slot per rank for its max |Delta| and one for its frame-guard flag (zero in the other
// ranks' slots: the SUM all-reduce then hands every rank all of them)
__global__ void exchange_stats_kernel(const unsigned int* delta_max_bits, uint32_t rank, uint32_t world, float* tail) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * world) return;
    float v = 0.0f;
    if (i == rank) v = __uint_as_float(delta_max_bits[0]);
    if (i == world + rank) v = delta_max_bits[1] ? 1.0f : 0.0f;
    tail[i] = v;
}

// step 2, after the all-reduce (SUM) over G ranks: S = sum of the ranks' moves, Q = sum of their
// squared lengths.  Each node end moves by S * f with f = clamp(Q / |S|^2, 1/G, 1): ranks that
// pulled the end the same way (coherent moves, |S|^2 = G*Q: every rank already made the full
// correction) are averaged, f = 1/G; uncorrelated small steps (|S|^2 ~ Q) add up, f = 1.
template <int FMT>
__global__ void exchange_apply_kernel(uint64_t* coords, uint64_t* base, uint64_t n_ends, Xform xf, const float* buf, float inv_world) {
    const float2* S = reinterpret_cast<const float2*>(buf);
    const float* Q = buf + 2 * n_ends;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ends; i += (uint64_t)gridDim.x * blockDim.x) {
        const float2 d = S[i];
        const float s2 = d.x * d.x + d.y * d.y;
        const float f = s2 > 0.0f ? fminf(fmaxf(Q[i] / s2, inv_world), 1.0f) : 1.0f;
        const uint64_t b = base[i];
        uint64_t w;
        if (FMT == kFmtQ32) {
            const int64_t qx = (int64_t)rintf(d.x * f * xf.scale), qy = (int64_t)rintf(d.y * f * xf.scale);
            w = (uint64_t)(uint32_t)((int64_t)(uint32_t)b + qx) | ((uint64_t)(uint32_t)((int64_t)(b >> 32) + qy) << 32);
        } else {
            w = (uint64_t)__float_as_uint(__uint_as_float((uint32_t)b) + d.x * f) |
                ((uint64_t)__float_as_uint(__uint_as_float((uint32_t)(b >> 32)) + d.y * f) << 32);
        }
        coords[i] = w;
        base[i] = w;
    }
}

// The exact exchange of a region-sharded tiled session (pgsgd_session_exchange_exact_*): after the launch of one colour the
// windows a rank ran are its alone and its far pulls are drained, so what it changed is a 64-bit integer delta per node end
// that nobody else's delta collides with but by plain addition: buf[e] = coords[e] - base[e] (mod 2^64), an integer SUM
// all-reduce, coords[e] = base[e] + sum — bit for bit what one GPU computes, with no merge rule.  The tail carries, in a
// slot per rank, the launch's far-pull count (the learning-rate cap of far terms needs the count over all ranks), the
// bits of its max |Delta| and its frame-guard flag.
__global__ void exchange_exact_prepare_kernel(const uint64_t* coords, const uint64_t* base, uint64_t n_ends, uint64_t* buf) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ends; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = coords[i] - base[i];
}
__global__ void exchange_exact_tail_kernel(const unsigned int* delta_max_bits, const unsigned long long* far_count, uint32_t rank, uint32_t world, uint64_t* tail) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * world) return;
    uint64_t v = 0;
    if (i == rank) v = far_count ? *far_count : 0ull;
    if (i == world + rank) v = delta_max_bits[0];
    if (i == 2 * world + rank) v = delta_max_bits[1] ? 1ull : 0ull;
    tail[i] = v;
}
__global__ void exchange_exact_apply_kernel(uint64_t* coords, uint64_t* base, uint64_t n_ends, const uint64_t* buf, unsigned long long* far_count, uint32_t world) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ends; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = base[i] + buf[i];
        coords[i] = w;
        base[i] = w;
    }
    if (far_count && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (uint32_t r = 0; r < world; ++r) acc += buf[n_ends + r];
        *far_count = acc;
    }
}

}  // namespace pgsgd