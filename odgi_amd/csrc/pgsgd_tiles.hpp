// pgsgd_tiles.hpp — device code of the region-exclusive tile kernel (the kernel large sorted graphs run).
// Included by pgsgd_session.hip after pgsgd_kernels.hpp.
#pragma once
#include "pgsgd_kernels.hpp"

namespace pgsgd {
// ---------------------------------------------------------------------------------------------
// Region-exclusive tiles — the kernel large sorted graphs run by default (PGSGD_FLAG_NO_TILES turns
// it off): the same terms, but grouped so that most memory requests never leave the CU.
//
// The per-lane kernel pays ~6 scattered memory requests per term (2 record gathers, 2 coordinate
// loads, 2 atomics); MI355X retires ~56 G scattered 64-byte reads or writes per second and ~23 G of
// anything smaller than 64 bytes that has to be merged into memory (atomics, 16-byte stores:
// tools/microbench.hip, profiles/r02/microbench_r2*.jsonl).  Here the steps of every path are cut into
// tiles of T consecutive steps and node ranks into regions of R nodes.  A work item is one region r0
// with every tile whose nodes fall inside the window [r0*R, (r0+2)*R) (the input of `odgi layout` is a
// sorted graph, so a run of path steps visits a run of node ranks).  One launch per region parity
// ("colour"): windows of one parity are disjoint, so every window has a single owner and no two private
// copies of a node end exist at the same time (summing the moves of several stale copies of one end
// overshoots — reproduced for tiles in tools/tile_sim.c).  Around every launch:
//   snapshot_kernel   streams the coordinates of every step's node into the second half of the step's
//                     32-byte record, so that a partner outside the window costs ONE gather that brings
//                     its handle, position, node length and both end coordinates (as they were when the
//                     launch began);
//   sgd_tile_kernel   a workgroup takes a work item, stages the window's 4R coordinate words in LDS and
//                     runs the item's tiles: tile records in LDS, first step uniform inside the tile (each
//                     tile gets its exact share of the iteration's terms, so the first step is uniform
//                     over all steps, as in the reference), partner by the reference's rule; ends inside
//                     the window are read from LDS and moved with LDS atomics; a partner outside is read
//                     from its snapshot and what the term adds to it is not written to its coordinate
//                     word but appended, as a 16-byte message, to the OUTBOX bucket of the partner's node
//                     range — staged in LDS and written as whole 64-byte lines;  at the end the window
//                     goes back with plain stores (nobody else writes coordinates during the launch);
//   far_drain_kernel  one workgroup per bucket adds the bucket's messages up in LDS and moves the node
//                     ends: exact 64-bit integer adds, so the sum is the one direct atomics would give.
// Tiles that do not fit a window (unsorted stretches) run with every end read from global memory and
// every update sent through the outbox.
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

// The far-update outbox.  Messages {node end, 0, delta lo, delta hi} are grouped by bucket = end >> shift.  A
// workgroup stages kObLine messages per bucket in LDS and writes a full line with kObLine lanes of one store
// instruction: a write that covers a whole 64-byte unit needs no read-for-ownership (55 G such writes/s against
// 23 G 16-byte ones, profiles/r02/microbench_r2b.jsonl).  Lines go to chunks of kObChunk messages that the
// workgroup owns (one returning global atomic on the bucket's chunk counter per chunk); `fill` says how many
// messages a chunk holds (lines fill in order, only a workgroup's last line of a bucket can be partial).
constexpr uint32_t kObLine = 4;                      // messages per staged line (64 bytes)
constexpr uint32_t kObChunk = 64;                    // messages per chunk (1 KiB)
constexpr uint32_t kObLinesPerChunk = kObChunk / kObLine;
constexpr uint32_t kObNone = 0xffffffu;              // LDS line word (chunk << 8 | lines used): no chunk yet
constexpr uint32_t kObOverflow = 0xfffffeu;          // the bucket's share of the pool is used up
constexpr uint32_t kObNoLine = 0xffffffffu;
struct Outbox {
    uint4* pool;               // [total chunks][kObChunk] messages
    const uint32_t* chunk0;    // [B] first chunk of the bucket's share of the pool
    const uint32_t* cap;       // [B] chunks in that share
    uint32_t* next;            // [B] chunks handed out this launch
    uint32_t* fill;            // [total chunks] messages in the chunk (written when the chunk is closed)
    unsigned long long* spill; // [2N] per node end: what messages that found no room in the pool add up to
    unsigned long long* overflow;  // their number (0 in normal operation)
    uint32_t n_buckets;
    uint32_t shift;            // bucket = node end >> shift
};

struct TileArgs {
    const Tile* tiles;
    const WorkItem* items;
    const uint64_t* term0;  // [n_tiles + 1] first term of every tile for this call's term count (tile_terms_kernel)
    uint32_t* queue;      // work-item counter of this launch
    uint32_t n_items;
    uint32_t region;      // R
    uint32_t tile_steps;  // T
    uint32_t sub, n_sub;  // this launch runs the tiles with index = sub (mod n_sub), each with its whole share
    uint32_t shard_rank, shard_world;  // multi-GPU: this device owns work items rank, rank+world, ...
    // learning-rate cap of terms whose partner is outside the window: 1/h, h = far pulls per node end in the previous
    // launch of this colour (read on the device: no host round trip), or far_mu_cap_first in the first iteration
    float far_mu_cap_first;
    uint32_t far_from_prev;
    const unsigned long long* far_prev;
    unsigned long long* far_count;     // partner ends updated outside the window, this launch
    const uint4* recs2;                // [2S] step records with the coordinate snapshot: {handle,len,pos}, {w_first, w_second}
    Outbox ob;
};

constexpr int kTileBlock = 256;
constexpr int kTileWaves = kTileBlock / 64;

// Every (tile, lane) pair owns a generator per iteration: lane l of the `lanes` lanes that work on a tile draws the
// tile's terms l, l + lanes, l + 2 lanes, ... from one stream.  Which workgroup runs a tile, and when, changes nothing
// about the terms that are drawn, and the oracle reproduces them (tests/test_gpu_parity.py: tile terms bit-exact).
// Fed to SplitMix64 by Xoshiro256Plus::seed, which steps by 0x9e3779b97f4a7c15: the multipliers must not be (small
// multiples of) that constant, or neighbouring streams would share state words.
__device__ __forceinline__ uint64_t tile_stream_seed(uint64_t seed_base, uint64_t epoch, uint64_t tile, uint32_t lane) {
    return seed_base + epoch * 0xd1342543de82ef95ull + ((tile << 10) | lane);
}

__device__ __forceinline__ uint64_t mul_div(uint64_t a, uint64_t b, uint64_t c) {
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) / (unsigned __int128)c);
}

// first term of every tile for a call of n_terms terms: tile ti runs terms [term0[ti], term0[ti + 1]); the shares
// telescope to exactly n_terms (one 128-bit division per tile and call instead of two per tile, wave and launch)
__global__ void tile_terms_kernel(const Tile* tiles, uint64_t n_tiles, uint64_t steps_total, uint64_t n_terms, uint64_t* term0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tiles) term0[i] = mul_div(tiles[i].cum, n_terms, steps_total);
    if (i == n_tiles) term0[i] = n_terms;
}

// LDS of one workgroup's outbox
struct OutboxLds {
    uint4* stage;     // [B][kObLine]
    uint32_t* cnt;    // [B] slots claimed (low half) | slots written (high half)
    uint32_t* line;   // [B] (chunk << 8) | lines used in the chunk
    uint2* list;      // [waves][64] lines completed in one round of one wave: {bucket, global line index}
};

// the next line of bucket b in the workgroup's current chunk (a new chunk when that one is full).  Called by the one
// lane that completed the bucket's staged line: nobody else touches line[b] until that lane reopens the bucket.
__device__ __forceinline__ uint32_t outbox_next_line(const Outbox& ob, const OutboxLds& L, uint32_t b) {
    const uint32_t lp = L.line[b];
    uint32_t chunk = lp >> 8, used = lp & 0xffu;
    if (chunk == kObOverflow) return kObNoLine;
    if (chunk == kObNone || used == kObLinesPerChunk) {
        if (chunk != kObNone) ob.fill[ob.chunk0[b] + chunk] = kObChunk;  // close the full chunk
        const uint32_t nc = atomicAdd(ob.next + b, 1u);
        if (nc >= ob.cap[b]) {
            L.line[b] = kObOverflow << 8;
            return kObNoLine;
        }
        chunk = nc;
        used = 0;
    }
    L.line[b] = (chunk << 8) | (used + 1);
    return (ob.chunk0[b] + chunk) * kObLinesPerChunk + used;
}

// Append one message per lane that has one.  Called by all 64 lanes of a wave together (converged).
__device__ __forceinline__ void outbox_push(const Outbox& ob, const OutboxLds& L, bool has, uint32_t end, uint64_t delta) {
    const uint32_t b = end >> ob.shift;
    const uint32_t lane = threadIdx.x & 63u;
    uint2* list = L.list + (threadIdx.x >> 6) * 64;
    bool pending = has;
    while (__ballot(pending)) {
        bool completes = false;
        if (pending && (__hip_atomic_load(L.cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 0xffffu) < kObLine) {
            // (a bucket whose line is full is being written out by the wave that completed it: try again next round)
            const uint32_t slot = atomicAdd(L.cnt + b, 1u) & 0xffffu;
            if (slot < kObLine) {
                L.stage[b * kObLine + slot] = make_uint4(end, 0u, (uint32_t)delta, (uint32_t)(delta >> 32));
                pending = false;
                completes = (atomicAdd(L.cnt + b, 1u << 16) >> 16) + 1 == kObLine;  // the last of the line's writers
            }
        }
        const uint64_t mask = __ballot(completes);
        if (mask) {  // wave-uniform: write the lines completed in this round, kObLine lanes per line
            const uint32_t dst = completes ? outbox_next_line(ob, L, b) : 0u;
            if (completes) list[__popcll(mask & ((1ull << lane) - 1ull))] = make_uint2(b, dst);
            // the wave's LDS operations execute in order; keep the compiler from moving the reads below above the writes
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            const uint32_t n = (uint32_t)__popcll(mask);
            for (uint32_t base = 0; base < n; base += 64u / kObLine) {
                const uint32_t e = base + lane / kObLine, piece = lane % kObLine;
                if (e < n) {
                    const uint2 it = list[e];
                    const uint4 m = L.stage[it.x * kObLine + piece];
                    if (it.y != kObNoLine) {
                        ob.pool[(uint64_t)it.y * kObLine + piece] = m;
                    } else {  // no room left in the pool: the spill words, added to the coordinates by the drain
                        atomicAdd(ob.spill + m.x, (unsigned long long)m.z | ((unsigned long long)m.w << 32));
                        atomicAdd(ob.overflow, 1ull);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if (completes) __hip_atomic_store(L.cnt + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // reopen the bucket
        }
    }
}

// FAR selects what a term whose partner lies outside the window does:
//   kFarTwoSided   the reference's update: both ends move by -/+ delta (the partner through the outbox);
//   kFarExclusive  experiment (PGSGD_FLAG_ONE_SIDED_FAR, graphs without window-less tiles only): only the
//                  first end moves, by -2 delta.  The pair is drawn from either side with equal
//                  probability, so every end still receives the same expected displacement per
//                  iteration.  Measured (profiles/r01/one_sided_far_experiment.jsonl): stress +1..9 %; not the
//                  reference's rule, never the default.
constexpr int kFarTwoSided = 0, kFarExclusive = 2;

// a term whose first step and partner are drawn and whose partner record is on its way
struct PendingTerm {
    Anchor an;
    PartnerDraw d;
    uint4 rb;    // partner record {handle, len, pos}
    uint4 snap;  // its coordinate snapshot {w_first, w_second} when the record came from global memory
    bool valid, from_global;
};

template <int COORD_LOAD, int FAR>
__global__ __launch_bounds__(kTileBlock) void sgd_tile_kernel(DevConst c, TileArgs ta, IterArgs a) {
    extern __shared__ uint64_t lds[];
    uint64_t* win = lds;                                                         // [4R] window words
    uint4* trec = reinterpret_cast<uint4*>(lds + 4 * (size_t)ta.region);         // [T] tile records
    OutboxLds L;
    L.stage = trec + ta.tile_steps;                                              // [B][kObLine]
    L.cnt = reinterpret_cast<uint32_t*>(L.stage + (size_t)ta.ob.n_buckets * kObLine);
    L.line = L.cnt + ta.ob.n_buckets;
    L.list = reinterpret_cast<uint2*>(L.line + ta.ob.n_buckets + (ta.ob.n_buckets & 1u));  // 8-byte aligned
    __shared__ uint32_t s_item;
    for (uint32_t b = threadIdx.x; b < ta.ob.n_buckets; b += blockDim.x) {
        L.cnt[b] = 0;
        L.line[b] = kObNone << 8;
    }
    float dmax = 0.0f;
    uint32_t n_far = 0;
    const uint64_t n_ends = 2 * (uint64_t)c.n_nodes;
    const uint32_t win_words = 4 * ta.region;
    // A partner outside the window is read as it was when this launch began, and what the term adds to it reaches
    // its owner after the launch: all the far pulls an end receives during one launch are computed against one stale
    // position and land together.  With mu = 1 each is a full projection and h of them overshoot h-fold (stress
    // 1e7 in the first iterations, profiles/r01/convergence_*.jsonl), so such terms are capped at mu = 1/h, h = far
    // pulls per node end in the previous launch of this colour: together they still amount to one projection.
    // Inactive once eta/d < 1/h.
    float far_mu_cap = ta.far_mu_cap_first;
    if (ta.far_from_prev) {
        const double h = (double)*ta.far_prev / (double)n_ends;
        far_mu_cap = h > 1.0 ? (float)(1.0 / h) : 1.0f;
    }
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(ta.queue, 1u);
        __syncthreads();
        const uint32_t item = s_item * ta.shard_world + ta.shard_rank;
        if (item >= ta.n_items) break;
        const WorkItem wi = ta.items[item];
        const uint64_t wbase = 2 * (uint64_t)wi.win0;  // first coordinate word of the window
        if (wi.local) {
            for (uint32_t i = threadIdx.x; i < win_words; i += blockDim.x)
                win[i] = wbase + i < n_ends ? load_word<COORD_LOAD>(c.coords, (uint32_t)(wbase + i)) : 0;
        }
        for (uint32_t ti = wi.tile_begin; ti < wi.tile_end; ++ti) {
            if (ti % ta.n_sub != ta.sub) continue;  // block-uniform
            const Tile t = ta.tiles[ti];
            __syncthreads();  // previous tile's terms are done with trec; window staging is complete
            for (uint32_t i = threadIdx.x; i < t.n; i += blockDim.x) trec[i] = c.recs[t.t0 + i];
            __syncthreads();
            const uint64_t term_begin = ta.term0[ti], term_end = ta.term0[ti + 1];
            const uint64_t pstart = c.path_first[t.path];
            const uint64_t cnt = c.path_first[t.path + 1] - pstart;
            const uint32_t lanes = t.lanes < blockDim.x ? t.lanes : blockDim.x;
            const bool worker = threadIdx.x < lanes;
            Xoshiro256Plus rng;
            if (worker) rng.seed(tile_stream_seed(c.seed_base, a.epoch, ti, threadIdx.x));
            // The same trip count for every lane of the workgroup (the outbox is wave-cooperative), and one trip more
            // than the longest lane needs: every trip draws term j and finishes term j - 1, whose partner record was
            // requested one trip earlier — the gather's latency hides behind the next term's sampling arithmetic.
            const uint64_t trips = (term_end - term_begin + lanes - 1) / lanes;
            PendingTerm P;
            P.valid = false;
            for (uint64_t j = 0; j <= trips; ++j) {
                PendingTerm N;
                N.valid = false;
                const uint64_t q = term_begin + threadIdx.x + j * lanes;
                if (worker && j < trips && q < term_end) {
                    // first step: uniform inside the tile; partner: the shared sampler (path_sgd_layout.cpp:205-270)
                    N.valid = true;
                    N.an.k = t.t0 + uniform_below(rng, t.n);
                    N.an.pstart = pstart;
                    N.an.cnt = cnt;
                    N.an.s_rank = N.an.k - pstart;
                    N.an.rec = trec[N.an.k - t.t0];
                    N.d = draw_partner(c, N.an, a.cooling, rng);
                    // the partner's record: the tile's LDS copy when it is a step of the tile, otherwise ONE 32-byte gather
                    // that also brings the coordinates both ends of its node had when this launch began
                    N.from_global = !(N.d.kb - t.t0 < (uint64_t)t.n);
                    if (N.from_global) {
                        N.rb = ta.recs2[2 * N.d.kb];
                        N.snap = ta.recs2[2 * N.d.kb + 1];
                    } else {
                        N.rb = trec[N.d.kb - t.t0];
                    }
                }
                bool msg_a = false, msg_b = false;
                uint32_t end_a = 0, end_b = 0;
                uint64_t delta = 0;
                if (P.valid) {
                    const Term tm = make_term(P.an, P.d, P.rb);
                    end_a = tm.end_a;
                    end_b = tm.end_b;
                    // ends inside the staged window live in LDS (unsigned compare covers "below the window")
                    const uint32_t la = end_a - (uint32_t)wbase, lb = end_b - (uint32_t)wbase;
                    const bool in_a = wi.local && la < win_words, in_b = wi.local && lb < win_words;
                    const uint64_t wa = in_a ? win[la] : load_word<COORD_LOAD>(c.coords, end_a);
                    uint64_t wb;
                    if (in_b) wb = win[lb];
                    else if (P.from_global)  // the snapshot that came with the record; w_first belongs to end `handle`
                        wb = ((end_b ^ tm.handle_b) & 1u) ? ((uint64_t)P.snap.z | ((uint64_t)P.snap.w << 32))
                                                           : ((uint64_t)P.snap.x | ((uint64_t)P.snap.y << 32));
                    else wb = load_word<COORD_LOAD>(c.coords, end_b);  // window-less tile, partner inside the tile
                    const float dx = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * c.xf.inv_scale;
                    const float dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * c.xf.inv_scale;
                    float r_x, r_y, abs_delta;
                    const bool one_sided = FAR == kFarExclusive && !in_b;
                    term_displacement(a.eta, tm.pos_a, tm.pos_b, dx, dy, r_x, r_y, abs_delta, (in_b || one_sided) ? 1.0f : far_mu_cap);
                    if (one_sided) {
                        r_x *= 2.0f;
                        r_y *= 2.0f;
                    }
                    dmax = fmaxf(dmax, abs_delta);
                    const float ux = (float)(tm.dither & 0xffffu) * (1.0f / 65536.0f);
                    const float uy = (float)(tm.dither >> 16) * (1.0f / 65536.0f);
                    float fx = r_x * c.xf.scale;
                    float fy = r_y * c.xf.scale;
                    fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
                    fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
                    const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
                    // a step that rounds to no quantum adds zero: nothing to send, in particular no message
                    // for a far partner (most far terms of the late iterations, where eta / d^2 is tiny)
                    if ((qx | qy) != 0) {
                        n_far += (in_b || one_sided) ? 0u : 1u;
                        delta = (uint64_t)qx + ((uint64_t)qy << 32);
                        if (in_b) atomicAdd(reinterpret_cast<unsigned long long*>(win + lb), (unsigned long long)delta);
                        else msg_b = !one_sided;
                        if (in_a) atomicAdd(reinterpret_cast<unsigned long long*>(win + la), (unsigned long long)(0ull - delta));
                        else msg_a = true;
                    }
                }
                outbox_push(ta.ob, L, msg_b, end_b, delta);
                if (!wi.local) outbox_push(ta.ob, L, msg_a, end_a, 0ull - delta);  // block-uniform: window-less tiles only
                P = N;
            }
        }
        __syncthreads();
        if (wi.local) {  // the window's only writer since it was staged: plain, coalesced stores
            for (uint32_t i = threadIdx.x; i < win_words; i += blockDim.x)
                if (wbase + i < n_ends) c.coords[wbase + i] = win[i];
        }
        __syncthreads();  // s_item and the window are reused
    }
    // write out the partly filled lines (16-byte stores: few) and close the chunks this workgroup still has open
    for (uint32_t b = threadIdx.x; b < ta.ob.n_buckets; b += blockDim.x) {
        const uint32_t n = L.cnt[b] & 0xffffu;  // every claimed slot is written: the workgroup is past its last barrier
        if (n) {
            const uint32_t dst = outbox_next_line(ta.ob, L, b);
            for (uint32_t i = 0; i < n; ++i) {
                const uint4 m = L.stage[b * kObLine + i];
                if (dst != kObNoLine) {
                    ta.ob.pool[(uint64_t)dst * kObLine + i] = m;
                } else {
                    atomicAdd(ta.ob.spill + m.x, (unsigned long long)m.z | ((unsigned long long)m.w << 32));
                    atomicAdd(ta.ob.overflow, 1ull);
                }
            }
        }
        const uint32_t lp = L.line[b], chunk = lp >> 8, used = lp & 0xffu;
        if (chunk < kObOverflow) ta.ob.fill[ta.ob.chunk0[b] + chunk] = n ? (used - 1) * kObLine + n : used * kObLine;
    }
    for (int off = 32; off > 0; off >>= 1) n_far += __shfl_xor(n_far, off);
    if ((threadIdx.x & 63) == 0 && n_far) atomicAdd(ta.far_count, (unsigned long long)n_far);
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(c.delta_max_bits, __float_as_uint(dmax));
}

// Before every tile launch: every 32-byte step record is rewritten — the static half from the 16-byte records, the
// second half with the coordinates of the two ends of the step's node (the end the step enters first) — so that a
// partner outside the window costs one gather, not a record gather plus a dependent coordinate load.  Whole-line
// writes (writing only the second halves costs a read-for-ownership of every line: 0.70 against 0.47 ms at 4.7e7
// steps, profiles/r02/microbench_r2b.jsonl).  Also resets the launch's work queue and far-pull counter.
__global__ void snapshot_kernel(const uint4* recs, const uint64_t* coords, uint64_t n_steps, uint4* recs2,
                                uint32_t* queue, unsigned long long* far_count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *queue = 0;
        *far_count = 0;
    }
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_steps; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 r = recs[k];
        const uint4 pr = *reinterpret_cast<const uint4*>(coords + (r.x & ~1u));  // both ends of the node, 16-byte aligned
        recs2[2 * k] = r;
        recs2[2 * k + 1] = (r.x & 1u) ? make_uint4(pr.z, pr.w, pr.x, pr.y) : pr;
    }
}

// After every tile launch: one workgroup per (bucket, part) adds the messages that fall into its part of the bucket's
// node range up in LDS and moves the node ends.  Nothing else writes coordinates while it runs, so the
// read-modify-write of a word is plain.  part_shift = log2 of the node ends one workgroup accumulates; a bucket wider
// than that is read by several workgroups, each keeping its own part (large graphs only).
__global__ __launch_bounds__(1024) void far_drain_kernel(Outbox ob, uint64_t* coords, uint64_t n_ends, uint32_t part_shift) {
    extern __shared__ uint64_t acc[];
    const uint32_t parts = 1u << (ob.shift - part_shift);
    const uint32_t b = blockIdx.x / parts, part = blockIdx.x % parts, span = 1u << part_shift;
    for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const uint32_t handed = ob.next[b], cap = ob.cap[b];
    const uint64_t n_slots = (uint64_t)(handed < cap ? handed : cap) * kObChunk;
    const uint64_t first = (uint64_t)ob.chunk0[b] * kObChunk;
    const uint64_t base = ((uint64_t)b << ob.shift) + ((uint64_t)part << part_shift);
    for (uint64_t i = threadIdx.x; i < n_slots; i += blockDim.x) {
        if ((uint32_t)(i % kObChunk) >= ob.fill[ob.chunk0[b] + (uint32_t)(i / kObChunk)]) continue;
        const uint4 m = ob.pool[first + i];
        const uint64_t off = (uint64_t)m.x - base;
        if (off < span) atomicAdd(reinterpret_cast<unsigned long long*>(acc + off), (unsigned long long)m.z | ((unsigned long long)m.w << 32));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) {
        if (base + i >= n_ends) break;
        const uint64_t sp = ob.spill[base + i];
        if (sp) ob.spill[base + i] = 0;
        if (acc[i] + sp != 0) coords[base + i] += acc[i] + sp;
    }
}

// the drain is done with the chunk counters: ready for the next launch
__global__ void outbox_reset_kernel(uint32_t* next, uint32_t n_buckets) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_buckets) next[b] = 0;
}

// sampler-only replay of one tile's terms (parity hook): out[(q - first_term)*4 + {0..3}] = {ka, kb, off_a, off_b};
// one thread per lane of the tile, terms in the lane's stream order
__global__ __launch_bounds__(kTileBlock) void tile_trace_kernel(DevConst c, Tile t, uint64_t tile_index, uint32_t lanes, uint64_t term_begin,
                                                                uint64_t term_end, IterArgs a, uint64_t* out) {
    const uint64_t pstart = c.path_first[t.path];
    const uint64_t cnt = c.path_first[t.path + 1] - pstart;
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;
    Xoshiro256Plus rng;
    rng.seed(tile_stream_seed(c.seed_base, a.epoch, tile_index, lane));
    for (uint64_t q = term_begin + lane; q < term_end; q += lanes) {
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
