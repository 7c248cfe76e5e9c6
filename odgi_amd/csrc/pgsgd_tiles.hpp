// pgsgd_tiles.hpp — device code of the region-exclusive tile kernel (the kernel large sorted graphs run).
// Included by pgsgd_session.hip after pgsgd_kernels.hpp.
#pragma once
#include "pgsgd_kernels.hpp"

namespace pgsgd {
// ---------------------------------------------------------------------------------------------
// Region-exclusive tiles — the kernel large sorted graphs run by default (PGSGD_FLAG_NO_TILES turns
// it off): the same terms, but grouped so that most memory requests never leave the CU.
//
// The default kernel pays ~6 scattered memory requests per term (2 record gathers, 2 coordinate
// loads, 2 atomics) and the memory system retires ~55 G of them per second; nothing else limits
// it.  Here the steps of every path are cut into tiles of T consecutive steps and node ranks into
// regions of R nodes.  A work item is one region r0 with every tile whose nodes fall inside the
// window [r0*R, (r0+2)*R) (the input of `odgi layout` is a sorted graph, so a run of path steps
// visits a run of node ranks).  A workgroup that takes a work item
//   * stages the window's 4R coordinate words in LDS (plus a copy of what it staged),
//   * runs each of its tiles: the tile's records go to LDS; every term draws its first step
//     uniformly inside the tile (each tile gets its exact share of the iteration's terms, so the
//     first step is uniform over all steps, as in the reference) and its partner by the
//     reference's rule; ends inside the window are read from LDS and moved with LDS atomics,
//     only ends outside it touch global memory (record gather, agent-scope load, atomic add),
//   * adds (staged now - staged then) back to global memory, one atomic per word that moved.
// Windows of regions of one parity are disjoint, so one launch per parity gives every window a
// single owner: no two private copies of a node end exist at the same time, which is what makes
// private copies safe (summing the moves of several stale copies of one end overshoots — the
// failure mode of a plain-sum multi-GPU merge, reproduced for tiles in tools/tile_sim.c).
// Tiles that do not fit a window (unsorted stretches) run with every end in global memory.
struct Tile {
    uint64_t t0;   // first flat step
    uint64_t cum;  // steps of all tiles before this one, in tile order (for the term partition)
    uint32_t n;    // steps in the tile
    uint32_t path;
    uint32_t lanes;  // lanes that may work on the tile at once: 2*n / (most visits of one node inside the tile),
    uint32_t pad;    // the per-lane kernel's hot-node rule applied to the tile (tandem repeats, tiny tail tiles)
};
struct WorkItem {
    uint32_t tile_begin, tile_end;
    uint32_t win0;   // first node rank of the window
    uint32_t local;  // 1 = window staged in LDS, 0 = every end in global memory
};
struct TileArgs {
    const Tile* tiles;
    const WorkItem* items;
    uint32_t* queue;      // work-item counter of this launch
    uint32_t n_items;
    uint32_t region;      // R
    uint32_t tile_steps;  // T
    uint64_t steps_total; // steps covered by tiles (paths of one step have none)
    uint32_t sub, n_sub;  // this launch runs the tiles with index = sub (mod n_sub), each with its whole share
    uint32_t shard_rank, shard_world;  // multi-GPU: this device owns work items rank, rank+world, ...
    float far_mu_cap;                  // learning-rate cap of terms whose partner is outside the window
    unsigned long long* far_count;     // partner ends updated outside the window, this launch
};

constexpr int kTileBlock = 256;

struct TileRecs {  // partner records from the tile staged in LDS when they are in it
    const uint4* lds;
    const uint4* recs;
    uint64_t t0;
    uint32_t n;
    __device__ __forceinline__ uint4 operator()(uint64_t k) const { return (k - t0 < (uint64_t)n) ? lds[k - t0] : recs[k]; }
};

__device__ __forceinline__ uint64_t tile_term_seed(uint64_t seed_base, uint64_t epoch, uint64_t q) {
    return seed_base + epoch * 0x9e3779b97f4a7c15ull + q;  // fed to SplitMix64 by Xoshiro256Plus::seed
}

__device__ __forceinline__ uint64_t mul_div(uint64_t a, uint64_t b, uint64_t c) {
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) / (unsigned __int128)c);
}

// FAR selects what a term whose partner lies outside the window does:
//   kFarTwoSided   the reference's update: both ends move by -/+ delta (the partner through a global atomic);
//   kFarExclusive  experiment (PGSGD_FLAG_ONE_SIDED_FAR, graphs without window-less tiles only): only the
//                  first end moves, by -2 delta.  The pair is drawn from either side with equal
//                  probability, so every end still receives the same expected displacement per
//                  iteration, but nobody except its owner writes a window during a launch: no copy of
//                  the staged state is kept and the window goes back with plain stores.  Measured
//                  (profiles/r01/one_sided_far_experiment.jsonl): +21 % terms/s, stress +1..9 %.
constexpr int kFarTwoSided = 0, kFarExclusive = 2;

template <int COORD_LOAD, int FAR>
__global__ __launch_bounds__(kTileBlock) void sgd_tile_kernel(DevConst c, TileArgs ta, IterArgs a) {
    extern __shared__ uint64_t lds[];
    uint64_t* win = lds;                                   // [4R] window words
    uint64_t* orig = lds + 4 * (size_t)ta.region;          // [4R] as staged (not with kFarExclusive)
    uint4* trec = reinterpret_cast<uint4*>(lds + (FAR == kFarExclusive ? 4 : 8) * (size_t)ta.region);  // [T] tile records
    __shared__ uint32_t s_item;
    float dmax = 0.0f;
    uint32_t n_far = 0;
    const uint64_t n_ends = 2 * (uint64_t)c.n_nodes;
    const uint32_t win_words = 4 * ta.region;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ta.queue, 1u);
        __syncthreads();
        const uint32_t item = s_item * ta.shard_world + ta.shard_rank;
        if (item >= ta.n_items) break;
        const WorkItem wi = ta.items[item];
        const uint64_t wbase = 2 * (uint64_t)wi.win0;  // first coordinate word of the window
        if (wi.local) {
            for (uint32_t i = threadIdx.x; i < win_words; i += blockDim.x) {
                const uint64_t w = wbase + i < n_ends ? load_word<COORD_LOAD>(c.coords, (uint32_t)(wbase + i)) : 0;
                win[i] = w;
                if (FAR != kFarExclusive) orig[i] = w;
            }
        }
        for (uint32_t ti = wi.tile_begin; ti < wi.tile_end; ++ti) {
            if (ti % ta.n_sub != ta.sub) continue;  // block-uniform
            const Tile t = ta.tiles[ti];
            __syncthreads();  // previous tile's terms are done with trec; window staging is complete
            for (uint32_t i = threadIdx.x; i < t.n; i += blockDim.x) trec[i] = c.recs[t.t0 + i];
            __syncthreads();
            const uint64_t term_begin = mul_div(t.cum, a.n_terms, ta.steps_total);
            const uint64_t term_end = mul_div(t.cum + t.n, a.n_terms, ta.steps_total);
            const uint64_t pstart = c.path_first[t.path];
            const uint64_t cnt = c.path_first[t.path + 1] - pstart;
            const uint32_t lanes = t.lanes < blockDim.x ? t.lanes : blockDim.x;
            for (uint64_t q = term_begin + threadIdx.x; threadIdx.x < lanes && q < term_end; q += lanes) {
                // Every term owns a generator seeded from (seed, iteration, term index): which workgroup
                // runs a tile, and when, changes nothing about the terms that are drawn, and the oracle can
                // reproduce any of them (tests/test_gpu_parity.py: tile terms bit-exact).
                Xoshiro256Plus rng;
                rng.seed(tile_term_seed(c.seed_base, a.epoch, q));
                // first step: uniform inside the tile; partner: the shared sampler (path_sgd_layout.cpp:205-270)
                Anchor an;
                an.k = t.t0 + uniform_below(rng, t.n);
                an.pstart = pstart;
                an.cnt = cnt;
                an.s_rank = an.k - pstart;
                an.rec = trec[an.k - t.t0];
                const Term tm = sample_partner(c, an, a.cooling, rng, TileRecs{trec, c.recs, t.t0, t.n});
                const uint32_t end_a = tm.end_a, end_b = tm.end_b, dither = tm.dither;
                const uint64_t pos_a = tm.pos_a, pos_b = tm.pos_b;
                // ends inside the staged window live in LDS (unsigned compare covers "below the window")
                const uint32_t la = end_a - (uint32_t)wbase, lb = end_b - (uint32_t)wbase;
                const bool in_a = wi.local && la < win_words, in_b = wi.local && lb < win_words;
                const uint64_t wa = in_a ? win[la] : load_word<COORD_LOAD>(c.coords, end_a);
                const uint64_t wb = in_b ? win[lb] : load_word<COORD_LOAD>(c.coords, end_b);
                const float dx = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * c.xf.inv_scale;
                const float dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * c.xf.inv_scale;
                // A partner outside the window is read as it was when its own window was staged, and what
                // this term adds to it reaches its owner only at that owner's next staging: all the far
                // pulls an end receives during one launch are computed against one stale position and land
                // together.  With mu = 1 each is a full projection and h of them overshoot h-fold (stress
                // 1e7 in the first iterations, profiles/r01/convergence_*.jsonl), so such terms are capped
                // at mu = 1/h, h = far pulls per node end per launch as counted in the previous launch:
                // together they still amount to one projection.  Inactive once eta/d < 1/h.
                float r_x, r_y, abs_delta;
                const bool one_sided = FAR == kFarExclusive && !in_b;
                term_displacement(a.eta, pos_a, pos_b, dx, dy, r_x, r_y, abs_delta, (in_b || one_sided) ? 1.0f : ta.far_mu_cap);
                if (one_sided) {
                    r_x *= 2.0f;
                    r_y *= 2.0f;
                }
                dmax = fmaxf(dmax, abs_delta);
                const float ux = (float)(dither & 0xffffu) * (1.0f / 65536.0f);
                const float uy = (float)(dither >> 16) * (1.0f / 65536.0f);
                float fx = r_x * c.xf.scale;
                float fy = r_y * c.xf.scale;
                fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
                fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
                const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
                // a step that rounds to no quantum adds zero: nothing to send, in particular no global atomic
                // for a far partner (most far terms of the late iterations, where eta / d^2 is tiny)
                if ((qx | qy) == 0) continue;
                n_far += (in_b || one_sided) ? 0u : 1u;
                const unsigned long long delta = (unsigned long long)((uint64_t)qx + ((uint64_t)qy << 32));
                if (in_b) atomicAdd(reinterpret_cast<unsigned long long*>(win + lb), delta);
                else if (!one_sided) atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + end_b), delta);
                if (in_a) atomicAdd(reinterpret_cast<unsigned long long*>(win + la), 0ull - delta);
                else atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + end_a), 0ull - delta);
            }
        }
        __syncthreads();
        if (wi.local) {
            for (uint32_t i = threadIdx.x; i < win_words; i += blockDim.x) {
                if (FAR == kFarExclusive) {  // sole writer of these words since they were staged
                    if (wbase + i < n_ends) __hip_atomic_store(c.coords + wbase + i, win[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {  // what this workgroup moved, added to whatever others added meanwhile
                    const uint64_t d = win[i] - orig[i];
                    if (d != 0 && wbase + i < n_ends) atomicAdd(reinterpret_cast<unsigned long long*>(c.coords + wbase + i), (unsigned long long)d);
                }
            }
        }
        __syncthreads();  // s_item and the window are reused
    }
    for (int off = 32; off > 0; off >>= 1) n_far += __shfl_xor(n_far, off);
    if ((threadIdx.x & 63) == 0 && n_far) atomicAdd(ta.far_count, (unsigned long long)n_far);
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(c.delta_max_bits, __float_as_uint(dmax));
}

// sampler-only replay of one tile's terms (parity hook): out[(q - first_term)*4 + {0..3}] = {ka, kb, off_a, off_b}
__global__ __launch_bounds__(kTileBlock) void tile_trace_kernel(DevConst c, Tile t, uint64_t steps_total, IterArgs a, uint64_t* out) {
    const uint64_t term_begin = mul_div(t.cum, a.n_terms, steps_total);
    const uint64_t term_end = mul_div(t.cum + t.n, a.n_terms, steps_total);
    const uint64_t pstart = c.path_first[t.path];
    const uint64_t cnt = c.path_first[t.path + 1] - pstart;
    for (uint64_t q = term_begin + blockIdx.x * blockDim.x + threadIdx.x; q < term_end; q += (uint64_t)gridDim.x * blockDim.x) {
        Xoshiro256Plus rng;
        rng.seed(tile_term_seed(c.seed_base, a.epoch, q));
        Anchor an;
        an.k = t.t0 + uniform_below(rng, t.n);
        an.pstart = pstart;
        an.cnt = cnt;
        an.s_rank = an.k - pstart;
        an.rec = c.recs[an.k];
        const Term tm = sample_partner(c, an, a.cooling, rng, GlobalRecs{c.recs});
        uint64_t* o = out + (q - term_begin) * 4;
        o[0] = an.k;
        o[1] = tm.kb;
        o[2] = tm.end_a & 1u;
        o[3] = tm.end_b & 1u;
    }
}

}  // namespace pgsgd
