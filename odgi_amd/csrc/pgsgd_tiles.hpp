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
//   snapshot_kernel   streams the coordinates of every step's node into the snapshot piece of the step's gather
//                     record (recs2: four steps per 128-byte line, pgsgd_kernels.hpp), so that a partner outside
//                     the window costs ONE line that brings its handle, position, node length and both end
//                     coordinates as they were at the snapshot — when a session starts, and whenever the
//                     coordinates changed behind the tile kernel's back (a tile refreshes its own steps' pieces);
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
    uint32_t local;  // bit 0: 1 = window staged in LDS, 0 = every end in global memory; the other bits: a window's tiles cut into
                     // several items of one launch (kItemHasNext, kItemDepShift)
};
// A launch of a few rounds of work items ends with a tail: slots idle while the last items finish (7-10 % of the
// workgroup-time at config 4, two rounds of ~1.9-ms items: tools/gpu_tile_tail.py).  A session whose launches have few
// rounds therefore cuts every window's tiles into k consecutive parts, each an item of its own: part j of every window
// sits in the queue behind part j - 1 of every window and starts when its predecessor has written the window back — the
// same tiles in the same order on the same window, in items a k-th as long.  The predecessor was taken from the queue
// earlier, so it is running or done when its successor is taken: the wait cannot deadlock (and gives up with a trap
// after 5 s instead of hanging the device, should that reasoning ever fail).
constexpr uint32_t kItemLocal = 1u;
constexpr uint32_t kItemHasNext = 2u;   // another item of this launch waits for this one: write the window through and raise its flag
constexpr uint32_t kItemDepShift = 2;   // bits 31..2: 1 + index (in the launch's item list) of the item this one waits for; 0: none

// The far-update outbox.  Messages are grouped by bucket = end >> shift and packed into 8 bytes: the end's offset
// inside its bucket (shift bits) and the two coordinate steps as signed qbits-bit quanta (25 bits each with the
// default 8192-end buckets; a far step is capped at a fraction of a projection and is far below 2^24 quanta — one
// that is not goes to the spill words directly, outbox_pack).  A workgroup stages kObLine messages per bucket in LDS
// and writes a full line with kObLine lanes of one store instruction: a write that covers a whole 64-byte unit needs
// no read-for-ownership (55 G such writes/s against 23 G 16-byte ones, profiles/r02/microbench_r2b.jsonl).  Lines go
// to chunks of kObChunk messages that the workgroup owns (one returning global atomic on the bucket's chunk counter
// per group of chunks); `fill` says how many messages a chunk holds (lines fill in order, only a workgroup's last
// line of a bucket can be partial).
constexpr uint32_t kObLine = 8;                      // messages per staged line (64 bytes) = 1 << kObLineLog2
constexpr uint32_t kObChunk = 128;                   // messages per chunk (1 KiB)
constexpr uint32_t kObLinesPerChunk = kObChunk / kObLine;
constexpr uint32_t kObGroup = 4;                     // chunks a workgroup takes from a bucket's share at a time (one returning global
constexpr uint32_t kObLinesPerGroup = kObGroup * kObLinesPerChunk;  // atomic per 512 messages: the window's own bucket fills 64 in under two trips)
constexpr uint32_t kObUsedBits = 10;                 // LDS line word per bucket: (chunk << 10) | lines claimed in the chunk
constexpr uint32_t kObUsedMask = (1u << kObUsedBits) - 1;
constexpr uint32_t kObNone = (1u << (32 - kObUsedBits)) - 1;   // no chunk yet
constexpr uint32_t kObOverflow = kObNone - 1;                  // the bucket's share of the pool is used up
constexpr uint32_t kObNoLine = 0xffffffffu;
// Partners just outside the window are the bulk of the far ones (Zipf: a third of all messages go to the bucket the
// window lies in), so the kObRings buckets at the window get a ring of kObRingLines staged lines each instead of
// one: a wave claims slots for all its lanes with one LDS atomic and never waits for a line to be written out unless
// 64 claims are outstanding.  (With one line per bucket a wave whose lanes hold 19 messages for the same bucket needs
// five claim-write-flush rounds per trip: measured, the cooling iterations ran 45 % slower than with direct atomics.)
constexpr uint32_t kObRings = 2;        // the window's bucket and the next one (a window can straddle a bucket border)
constexpr uint32_t kObRingLinesLog2 = 4;
constexpr uint32_t kObRingLines = 1u << kObRingLinesLog2;   // 128 slots: the waves of a workgroup drift apart by a trip or two
constexpr uint32_t kObLineLog2 = 3;
struct Outbox {
    unsigned long long* pool;  // [total chunks][kObChunk] packed messages
    const uint32_t* chunk0;    // [B] first chunk of the bucket's share of the pool
    const uint32_t* cap;       // [B] chunks in that share
    uint32_t* next;            // [B] chunks handed out this launch
    uint32_t* fill;            // [total chunks] messages in the chunk (written when the chunk is closed)
    unsigned long long* spill; // [2N] per node end: what messages that found no room in the pool add up to
    unsigned long long* overflow;  // their number (0 in normal operation)
    uint32_t n_buckets;
    uint32_t shift;            // bucket = node end >> shift
    uint32_t qbits;            // bits of one packed coordinate step: min(25, (64 - shift) / 2)
};

__host__ __device__ inline uint32_t outbox_qbits(uint32_t shift) { return (64u - shift) / 2u < 25u ? (64u - shift) / 2u : 25u; }

// message = end offset in the bucket | x step << shift | y step << (shift + qbits), steps in two's complement.
// (qx, qy) are the signed 32-bit steps of the two fields; the word the per-lane kernel would add atomically is
// outbox_delta(qx, qy) = qx + (qy << 32) in 64-bit arithmetic.
__device__ __forceinline__ uint64_t outbox_delta(int32_t qx, int32_t qy) { return (uint64_t)(int64_t)qx + ((uint64_t)(int64_t)qy << 32); }
__device__ __forceinline__ bool outbox_pack(const Outbox& ob, uint32_t end, int32_t qx, int32_t qy, uint64_t& msg) {
    const uint32_t lim = 1u << (ob.qbits - 1);
    const uint32_t mask = (1u << ob.qbits) - 1u;  // qbits <= 25
    const uint32_t fx = (uint32_t)qx & mask, fy = (uint32_t)qy & mask, off = end & ((1u << ob.shift) - 1u);
    if (ob.shift + ob.qbits >= 32u) {  // (wave-uniform; always, but for the experiment knob that narrows the fields) the y field lies
        // in the high word: five 32-bit operations with scalar shift counts instead of two variable 64-bit shifts
        const uint32_t lo = off | (fx << ob.shift);
        const uint32_t hi = (fx >> (32u - ob.shift)) | (fy << (ob.shift + ob.qbits - 32u));
        msg = (uint64_t)lo | ((uint64_t)hi << 32);
    } else {
        msg = (uint64_t)off | ((uint64_t)fx << ob.shift) | ((uint64_t)fy << (ob.shift + ob.qbits));
    }
    return (((uint32_t)qx + lim) | ((uint32_t)qy + lim)) >> ob.qbits == 0;  // both in [-lim, lim)
}
__device__ __forceinline__ uint64_t outbox_unpack(const Outbox& ob, uint64_t msg, uint32_t& end_off) {
    end_off = (uint32_t)msg & ((1u << ob.shift) - 1u);
    const int64_t qx = (int64_t)(msg << (64u - ob.shift - ob.qbits)) >> (64u - ob.qbits);
    const int64_t qy = (int64_t)(msg << (64u - ob.shift - 2u * ob.qbits)) >> (64u - ob.qbits);
    return (uint64_t)qx + ((uint64_t)qy << 32);
}
#ifndef PGSGD_TILE_ABL_NO_LINE_STORES
#define PGSGD_TILE_ABL_NO_LINE_STORES 0   // experiment build: the rings' protocol runs, completed lines are not written out (results invalid)
#endif
constexpr uint32_t kItemQueues = 8;
constexpr uint32_t kNoItem = 0xffffffffu;
constexpr int kTileBlock = 256;
constexpr int kTileWaves = kTileBlock / 64;

struct TileArgs {
    const Tile* tiles;
    const uint4* tile_heads;  // [n_tiles] what the kernel needs of a tile before its first term, one 16-byte load: {first flat step, steps | lanes << 16,
                              // first flat step of its path, steps of its path} (a tiled session has fewer than 2^32 path steps)
    const WorkItem* items;
    const uint64_t* term0;  // [n_tiles + 1] first term of every tile for this call's term count (tile_terms_kernel)
    // Work items of a launch are pulled from up to kItemQueues runs.  By default there is one run (items by decreasing
    // size); the experiment PGSGD_TILE_ORDER=region keeps the items in node order and cuts them into one run per XCD: a
    // workgroup pulls from the run of the XCD it runs on (HW_REG_XCC_ID; observed: block b -> XCD b % 8) and, when that
    // is used up, from the following ones, so that the workgroups of one XCD work on neighbouring windows at the same
    // time and find the partner records just outside a tile in their common L2 (measured: 8 % fewer L2 misses, and a
    // slower kernel — pgsgd_session.hip: group_tiles).  Which workgroup runs an item changes nothing about its terms.
    uint32_t* queue;      // [kItemQueues] work-item counters of this launch
    uint32_t chunk[9];    // queue q serves items [chunk[q], chunk[q + 1]) (relative to `items`)
    uint32_t n_items;
    uint32_t region;      // R
    uint32_t tile_steps;  // T
    uint32_t sub, n_sub;  // this launch runs the tiles with index = sub (mod n_sub), each with its whole share
    uint32_t shard_rank, shard_world;  // multi-GPU: this device owns work items rank, rank+world, ...
    // learning-rate cap of terms whose partner is outside the window: 1/h, h = far pulls per node end in the previous
    // launch of this colour (read on the device: no host round trip), or far_mu_cap_first in the first iteration
    float far_mu_cap_first;
    float far_relax;      // tile_far_relax(iteration)
    uint32_t far_from_prev;
    const unsigned long long* far_prev;
    unsigned long long* far_count;     // partner ends updated outside the window, this launch
    const uint4* recs2;                // gather records with the coordinate snapshot, four steps per 128-byte group (recs2_static_piece / recs2_snap_piece)
    uint4* recs2_out;                  // the same array, written by a tile for its own steps when its terms are done; null: a
                                       // sharded session, whose records are all rewritten by snapshot_kernel before a launch
    uint64_t seed_base;                // of the tile streams (tile_stream_seed)
    unsigned long long* clock_probe;   // [6] {shader cycles, 100 MHz ticks} when workgroup 0 starts and ends: the launch's achieved shader clock; [4], [5]: terms
                                       // that went for their ends' locks, and that lost one (cumulative)
    uint32_t* item_done;               // [items of the launch] launch stamp of the last launch that finished the item (items cut into parts, WorkItem::local)
    uint32_t stamp;                    // this launch's
    unsigned long long* tail_probe;    // debug knob PGSGD_TILE_TAIL: [4] {sum of the workgroups' lifetimes, last finish, first start, workgroups} in 100 MHz ticks; null: off
    uint32_t pair_uniform;             // 1: the lanes of a wave share uniform partners in pairs (tile_pair_partner); 0: PGSGD_FLAG_NO_PARTNER_PAIRS
    float lock_mu;                     // conflict resolution inside a window: terms whose learning rate mu reaches this take both their ends' locks or do nothing (0: off)
    uint32_t snap_every;               // debug knob PGSGD_TILE_SNAP_EVERY: a tile rewrites its snapshot records every k-th iteration only
    uint32_t lane_coin;                // debug knob PGSGD_TILE_LANE_COIN: the Zipf/uniform coin per lane (bit 31 of its word), as in round 3 (A/B only)
    uint32_t wq_threshold;             // messages a wave's queue holds before it goes to the rings: 64 per message a lane hands over in one call (PUSH); debug knob PGSGD_TILE_WQ
    unsigned long long* term_count;    // terms this session's tile launches have executed (cumulative; every wave adds what its lanes finished): pgsgd_session_terms_executed
    uint32_t tile_rotate;              // debug knob PGSGD_TILE_ROTATE: an item's tiles start at another one every iteration (which path has the last word on a window)
    Outbox ob;
};

// The head of the tile kernel's argument segment as the ABI lays it out (arguments in order, each at its natural
// alignment; the sampler's tables and the iteration's numbers follow), and a read of one argument from it AT THE PLACE
// OF USE.  The kernel has ~45 pointer-sized arguments; the compiler loads them all on entry and keeps them in scalar
// registers across the term loop, where two thirds are never used — 70-90 scalar registers spilled into vector lanes
// (profiles/r03: .sgpr_spill_count) and read back with v_readlane inside the loop.
// Arguments that only cold code needs (work-item pick-up, the staging of a window, the outbox's line replacement and
// flush, the epilogue) are read through TILE_COLD where they are used: one s_load there, no register held in between.
// (The empty asm hides the segment's address from the optimiser, which would otherwise merge the load with the entry
// loads.)  sgd_tile_kernel checks the layout assumption once per launch and traps if it does not hold.
struct TileKernelArgs {
    DevConst c;
    TileArgs ta;
};
template <class T>
__device__ __forceinline__ T tile_kernarg(uint32_t offset) {
    uint64_t base = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    return *reinterpret_cast<const T __attribute__((address_space(4)))*>(base + offset);
}
#define TILE_COLD(field) tile_kernarg<std::remove_reference_t<decltype(TileKernelArgs::field)>>((uint32_t)__builtin_offsetof(TileKernelArgs, field))

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

// LDS of one workgroup's outbox.  Every bucket has a ring of staged lines: one line for an ordinary bucket, kObRingLines
// for the kObRings buckets around the workgroup's current window ("hot" rings; ring index kObHot + r).  Slots are claimed
// from a monotonic head counter: slot s lives in line (s / kObLine) % lines of the ring, round s / (kObLine * lines); a
// lane may write it once the line is back from its previous round, and the writer that completes a line writes it out
// and reopens it.  A line's state word is (rounds completed << kObStateShift) | slots written in the open round.
//
// In front of the rings every WAVE keeps a private queue of packed messages (tile_wq_cap(PUSH) entries: message word + bucket byte).
// A term appends its message there — a ballot, a prefix count and two LDS stores, the fill count lives in a scalar
// register — and the ring protocol below runs only when 64 messages are waiting, one per lane: its cost does not depend
// on how many lanes take part (about a hundred vector instructions per call), and in most iterations only a third of the
// lanes have a message per trip (a step that rounds to no quantum sends nothing).  Since round 6's third session a WARM launch
// waits for 128 and hands the rings TWO messages per lane and call (outbox_push<2>): the protocol's cost is its chain of dependent
// LDS round trips, every round trip then carries both messages, and a wave calls half as often — warm launches +2 ... 3.5 %, the
// driver's window 0.668-0.673 -> 0.680-0.681 on one box (profiles/r06/push2_ab.jsonl); a cooling launch, whose messages go to the two
// hot rings, keeps one (with two its waves wait for each other's lines there).  The queue grew by 64 entries per wave for it and
// the list of completed lines moved INTO the queue (the entries a call has just taken): 30 848 bytes of LDS per workgroup, five per CU.
constexpr uint32_t kObStateShift = 4;                 // (kObLine = 8 slots < 1 << 4)
constexpr uint32_t kWqPush = 2;                       // most messages a lane hands to the rings in one call of their protocol (outbox_push<K>: the warm
                                                      // instance 2, the cooling instance 1)
__host__ __device__ constexpr uint32_t tile_wq_cap(uint32_t push) { return 64u * (push + 1u); }   // fewer than 64 * push waiting + at most 64 appended per call
// The arrays follow one another behind the tile records; the struct keeps their common base as a byte offset into the
// workgroup's LDS and works the others out where they are used (a few scalar operations in the rings' protocol, which
// runs once per 64 messages of a wave) instead of holding eight pointers in scalar registers across the term loop.
extern __shared__ uint64_t tile_lds[];
struct OutboxLds {
    uint32_t base;      // byte offset of `stage` in the workgroup's LDS
    uint32_t n_buckets;
    uint32_t ring_b0;   // bucket of hot ring 0 (the window's), wave-uniform
    uint32_t wq_cap;    // entries of a wave's queue: tile_wq_cap(PUSH), a constant of the kernel instance
    __device__ __forceinline__ uint32_t lines() const { return n_buckets + kObRings * kObRingLines; }
    __device__ __forceinline__ uint32_t at(uint32_t bytes) const {  // (the empty asm keeps the sum at its use: not hoisted out of the term loop into a register of its own)
        uint32_t off = base + bytes;
        asm volatile("" : "+s"(off));
        return off;
    }
    template <class T> __device__ __forceinline__ T* ptr(uint32_t off) const { return reinterpret_cast<T*>(reinterpret_cast<char*>(tile_lds) + off); }
    // [B + kObRings * kObRingLines][kObLine] staged lines: bucket b's line is b, hot ring r's lines follow
    __device__ __forceinline__ uint64_t* stage() const { return ptr<uint64_t>(at(0)); }
    // [waves][wq_cap] the waves' private queues: packed messages ...
    __device__ __forceinline__ uint32_t wq_msg_off() const { return lines() * kObLine * 8u; }
    __device__ __forceinline__ uint64_t* wq_msg() const { return ptr<uint64_t>(at(wq_msg_off())); }
    // (the lines completed in one round of one wave — {staged line, global line index} — are listed in the part of the wave's queue
    // the round's messages were taken from: no array of their own)
    // [B + kObRings] slots claimed
    __device__ __forceinline__ uint32_t head_off() const { return wq_msg_off() + kTileWaves * wq_cap * 8u; }
    __device__ __forceinline__ uint32_t* head() const { return ptr<uint32_t>(at(head_off())); }
    // [B + kObRings * kObRingLines] per line: (rounds completed << kObStateShift) | slots written
    __device__ __forceinline__ uint32_t state_off() const { return head_off() + (n_buckets + kObRings) * 4u; }
    __device__ __forceinline__ uint32_t* state() const { return ptr<uint32_t>(at(state_off())); }
    // [B] (first chunk of the open group << 10) | lines claimed in the group
    __device__ __forceinline__ uint32_t line_off() const { return state_off() + lines() * 4u; }
    __device__ __forceinline__ uint32_t* line() const { return ptr<uint32_t>(at(line_off())); }
    // [B] copy of Outbox::chunk0 (read for every line that goes out: not from global memory)
    __device__ __forceinline__ uint32_t chunk0_off() const { return line_off() + n_buckets * 4u; }
    __device__ __forceinline__ uint32_t* chunk0() const { return ptr<uint32_t>(at(chunk0_off())); }
    // [waves][wq_cap] ... and the queued messages' buckets
    __device__ __forceinline__ uint32_t wq_b_off() const { return chunk0_off() + n_buckets * 4u; }
    __device__ __forceinline__ uint8_t* wq_b() const { return ptr<uint8_t>(at(wq_b_off())); }
};

__host__ __device__ inline uint32_t tile_lock_words(uint32_t region) { return (4u * region + 31u) / 32u; }
__host__ __device__ inline size_t outbox_lds_bytes(uint32_t n_buckets, uint32_t wq_cap) {
    const size_t lines = (size_t)n_buckets + kObRings * kObRingLines;
    return lines * kObLine * sizeof(uint64_t) + (size_t)kTileWaves * wq_cap * (sizeof(uint64_t) + 1) +
           ((size_t)n_buckets + kObRings + lines + 2 * (size_t)n_buckets) * sizeof(uint32_t);
}

// The next line of bucket b in the workgroup's current group of chunks (a new group when that one is full).  Several
// lanes may ask for lines of one bucket at once (two lines of a ring completed in the same round): lines are claimed
// with an LDS atomic; the lane that claims line kObLinesPerGroup replaces the group, the others wait for it without
// adding to the word (every lane strays at most once per replacement, so the 10-bit field cannot overflow).
// (`c0` = L.chunk0()[b] and `first` = the value a caller's own atomicAdd(L.line() + b, 1) returned: a caller with several lines to
// place issues all those LDS requests before it waits for any)
__device__ __forceinline__ uint32_t outbox_next_line(const Outbox& ob, const OutboxLds& L, uint32_t b, uint32_t c0, uint32_t first) {
    // (No spin loop of its own for a lane that finds the group being replaced: the replacing lane may be in the same
    // wave, and a wave runs the two sides of a branch one after the other — a lane spinning in an inner loop would
    // wait for a lane that is not running.  The poll sits at the top of the one loop instead: every lane still in it
    // reaches the loop's end before any starts the next pass, and the replacing lane finishes within its pass.)
    bool strayed = false, have = true;
    for (;;) {
        if (!have && strayed && (__hip_atomic_load(L.line() + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & kObUsedMask) > kObLinesPerGroup) {
            __builtin_amdgcn_s_sleep(1);  // still being replaced (by a lane of another wave)
            continue;
        }
        const uint32_t lp = have ? first : atomicAdd(L.line() + b, 1u);
        have = false;
        const uint32_t chunk = lp >> kObUsedBits, used = lp & kObUsedMask;  // first chunk of the group, lines claimed in the group
        if (used < kObLinesPerGroup && chunk < kObOverflow) return (c0 + chunk) * kObLinesPerChunk + used;  // the group's chunks are consecutive
        if (chunk == kObOverflow) {  // the bucket's share of the pool is used up (sticky; the count field is put back so that it cannot run over)
            atomicExch(L.line() + b, kObOverflow << kObUsedBits);
            return kObNoLine;
        }
        if (used == kObLinesPerGroup) {  // this lane replaces the full group and takes the new one's first line
            if (chunk != kObNone)
                for (uint32_t k = 0; k < kObGroup; ++k) TILE_COLD(ta.ob.fill)[c0 + chunk + k] = kObChunk;
            const uint32_t nc = atomicAdd(TILE_COLD(ta.ob.next) + b, kObGroup);
            if (nc + kObGroup > TILE_COLD(ta.ob.cap)[b]) {
                atomicExch(L.line() + b, kObOverflow << kObUsedBits);
                return kObNoLine;
            }
            atomicExch(L.line() + b, (nc << kObUsedBits) | 1u);
            return (c0 + nc) * kObLinesPerChunk;
        }
        strayed = true;  // the group is being replaced: every lane adds to the word at most once per replacement, so the 10-bit field holds
    }
}

__device__ __forceinline__ uint32_t outbox_next_line(const Outbox& ob, const OutboxLds& L, uint32_t b) {
    // (the read BEFORE the claim it does not depend on: the two LDS requests travel together)
    const uint32_t c0 = L.chunk0()[b];
    return outbox_next_line(ob, L, b, c0, atomicAdd(L.line() + b, 1u));
}

// Stage up to kWqPush packed messages per lane (message k of the lane: has[k], bucket b[k]), writing out every line this completes.
// Called by all 64 lanes of a wave together (converged).  The protocol's cost is its chain of dependent LDS round trips — claim, line
// state, slot write + count, the completed lines' places in the pool, their list, their words — not its instructions
// (profiles/r06/NOTES.md sections 9, 15): with two messages per lane every round trip carries both, and a wave calls half as often.
template <uint32_t K>
__device__ __forceinline__ void outbox_push(const Outbox& ob, const OutboxLds& L, uint2* list, const bool (&has_in)[K], const uint32_t (&b)[K], const uint64_t (&packed)[K]) {
    static_assert((K == 1 || K == 2) && K <= kWqPush && kObRings == 2, "the claims below are written for one or two messages per lane and two hot rings");
    constexpr uint32_t k1 = K - 1;   // the lane's last message (= its first when K == 1: the second's code then folds away)
    if (!__ballot(has_in[0] || has_in[k1])) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = (1ull << lane) - 1ull;
    auto rank_in = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };  // lanes below this one in the mask
    // claim a slot: the hot rings with one LDS instruction per wave (lane rr claims for ring rr what the wave's lanes
    // need of it), ordinary buckets with one atomic per message — all of them issued before any is waited for
    uint32_t slot[K] = {}, r[K];
    bool hot[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        r[k] = b[k] - L.ring_b0;  // hot ring of the bucket, if it has one
        hot[k] = has_in[k] && r[k] < kObRings;
        if (has_in[k] && !hot[k]) slot[k] = atomicAdd(L.head() + b[k], 1u);
    }
    const uint64_t m00 = __ballot(hot[0] && r[0] == 0), m01 = __ballot(hot[0] && r[0] == 1);
    const uint64_t m10 = K > 1 ? __ballot(hot[k1] && r[k1] == 0) : 0ull, m11 = K > 1 ? __ballot(hot[k1] && r[k1] == 1) : 0ull;
    if (m00 | m01 | m10 | m11) {  // wave-uniform
        const uint32_t n00 = (uint32_t)__popcll(m00), n01 = (uint32_t)__popcll(m01);
        const uint32_t want = lane == 0 ? n00 + (uint32_t)__popcll(m10) : n01 + (uint32_t)__popcll(m11);
        uint32_t base = 0;
        if (lane < kObRings && want) base = atomicAdd(L.head() + L.n_buckets + lane, want);
        const uint32_t base0 = (uint32_t)__builtin_amdgcn_readlane((int)base, 0), base1 = (uint32_t)__builtin_amdgcn_readlane((int)base, 1);
        if (hot[0]) slot[0] = r[0] == 0 ? base0 + rank_in(m00) : base1 + rank_in(m01);
        if (K > 1 && hot[k1]) slot[k1] = r[k1] == 0 ? base0 + n00 + rank_in(m10) : base1 + n01 + rank_in(m11);
    }
    uint32_t src[K], round[K];
    bool pending[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        const uint32_t lines_log2 = hot[k] ? kObRingLinesLog2 : 0u;  // (shifts and masks: a ring's line count is a power of two)
        src[k] = (hot[k] ? L.n_buckets + r[k] * kObRingLines : b[k]) + ((slot[k] >> kObLineLog2) & ((1u << lines_log2) - 1u));  // staged line of the slot
        round[k] = slot[k] >> (kObLineLog2 + lines_log2);
        pending[k] = has_in[k];
    }
    bool defer = false;   // the lane's two messages completed a line each in one pass: the second line goes out in the next pass
    do {
        // the line must be back from its previous round (it is, unless every slot of the ring is claimed and not yet out — or the lane's
        // other message holds the slot in front of it in the same line: that one goes out in this pass, this one in the next)
        uint32_t st[K] = {};
#pragma unroll
        for (uint32_t k = 0; k < K; ++k)
            if (pending[k]) st[k] = __hip_atomic_load(L.state() + src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool wrote[K] = {};
        uint32_t cnt[K] = {};
#pragma unroll
        for (uint32_t k = 0; k < K; ++k)
            if (pending[k] && (st[k] >> kObStateShift) == (round[k] & (0xffffffffu >> kObStateShift))) {
                L.stage()[src[k] * kObLine + slot[k] % kObLine] = packed[k];
                pending[k] = false;
                wrote[k] = true;
                cnt[k] = atomicAdd(L.state() + src[k], 1u);
            }
        const bool comp0 = wrote[0] && ((cnt[0] + 1u) & ((1u << kObStateShift) - 1u)) == kObLine;  // the last of a line's writers writes it out
        const bool comp1 = K > 1 && (defer || (wrote[k1] && ((cnt[k1] + 1u) & ((1u << kObStateShift) - 1u)) == kObLine));
        if ((pending[0] || pending[k1]) && !(wrote[0] || wrote[k1])) __builtin_amdgcn_s_sleep(2);  // waiting for another wave to write a line out: leave it the issue slots
        // one completed line per lane and pass (two at once are rare: one time in 64)
        const bool completes = comp0 || comp1;
        defer = comp0 && comp1;
        const uint32_t csrc = comp0 ? src[0] : src[k1], cb = comp0 ? b[0] : b[k1], cround = comp0 ? round[0] : round[k1];
        const uint64_t mask = __ballot(completes);
        if (mask) {  // wave-uniform: the lines completed in this pass, kObLine lanes per line
            const uint32_t dst = completes ? outbox_next_line(ob, L, cb) : 0u;
            if (completes) list[__popcll(mask & below)] = make_uint2(csrc, dst);
            // the wave's LDS operations execute in order; keep the compiler from moving the reads below above the writes
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            const uint32_t n = (uint32_t)__popcll(mask);
            for (uint32_t base = 0; base < n; base += 64u / kObLine) {
                const uint32_t e = base + lane / kObLine, piece = lane % kObLine;
                if (e < n) {
                    const uint2 it = list[e];
                    const uint64_t m = L.stage()[it.x * kObLine + piece];
                    if (it.y != kObNoLine) {
                        if (PGSGD_TILE_ABL_NO_LINE_STORES == 0) TILE_COLD(ta.ob.pool)[(uint64_t)it.y * kObLine + piece] = m;
                    } else {  // no room left in the pool: the spill words, added to the coordinates by the drain
                        const uint32_t mb = it.x < L.n_buckets ? it.x : L.ring_b0 + (it.x - L.n_buckets) / kObRingLines;  // the staged line's bucket
                        uint32_t off;
                        const uint64_t d = outbox_unpack(ob, m, off);
                        atomicAdd(TILE_COLD(ta.ob.spill) + (((uint64_t)mb << ob.shift) | off), (unsigned long long)d);
                        atomicAdd(TILE_COLD(ta.ob.overflow), 1ull);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if (completes)  // reopen the line for its next round
                __hip_atomic_store(L.state() + csrc, (cround + 1u) << kObStateShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } while (__ballot(pending[0] || pending[k1] || defer));
}

// A wave's private message queue in front of outbox_push (see OutboxLds).  `n` is the wave's fill count: wave-uniform,
// kept in a scalar register by its callers (it only ever changes by ballot counts).
struct WaveQueue {
    uint64_t* msg;
    uint8_t* b;
};
// append one message per lane that has one; a step too wide for the packed form (never seen with the far cap) goes
// straight to the spill words, which the drain adds with the messages
__device__ __forceinline__ void wq_append(const Outbox& ob, const WaveQueue& q, uint32_t& n, bool has, uint32_t end, int32_t qx, int32_t qy) {
    if (!__ballot(has)) return;
    uint64_t packed = 0;
    const bool fits = outbox_pack(ob, end, qx, qy, packed);
    if (has && !fits) atomicAdd(TILE_COLD(ta.ob.spill) + end, (unsigned long long)outbox_delta(qx, qy));
    has = has && fits;
    const uint64_t m = __ballot(has);  // (taken by the whole wave, outside the branch: the fill count must stay wave-uniform)
    const uint32_t idx = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (has) {
        q.msg[idx] = packed;
        q.b[idx] = (uint8_t)(end >> ob.shift);
    }
    n += (uint32_t)__popcll(m);
}
// hand the last min(n, 64 * K) queued messages to the rings, K per lane (the wave's LDS operations execute in order: what
// wq_append stored is what this reads).  The part of the queue they are taken from is free from then on: the rings' protocol lists the
// lines it completes there.
template <uint32_t K>
__device__ __forceinline__ void wq_push(const Outbox& ob, const OutboxLds& L, const WaveQueue& q, uint32_t& n) {
    const uint32_t take = n < 64u * K ? n : 64u * K, lane = threadIdx.x & 63u;
    const uint32_t first = n - take;
    bool has[K];
    uint64_t packed[K];
    uint32_t b[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        has[k] = 64u * k + lane < take;
        packed[k] = 0;
        b[k] = 0;
        if (has[k]) {
            packed[k] = q.msg[first + 64u * k + lane];
            b[k] = q.b[first + 64u * k + lane];
        }
    }
    n -= take;
    // (the messages are in registers before the list is written over them)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    outbox_push<K>(ob, L, reinterpret_cast<uint2*>(q.msg + first), has, b, packed);
}

// Write out the partly filled line of every ring in [first, first + count) (ring index: bucket, or n_buckets + r for a
// hot ring), padded to a whole line with messages that add zero (so every line in the pool is full and is written as
// one 64-byte unit), and put the rings back to their initial state.  Called by the whole workgroup, between barriers,
// when no push is in flight: kObLine consecutive lanes per ring.
__device__ __forceinline__ void outbox_flush_rings(const Outbox& ob, const OutboxLds& L, uint32_t first, uint32_t count) {
    const uint32_t piece = threadIdx.x % kObLine;
    for (uint32_t base = 0; base < count; base += blockDim.x / kObLine) {
        const uint32_t i = base + threadIdx.x / kObLine;
        uint32_t n = 0, b = 0, src = 0;
        if (i < count) {
            const uint32_t ring = first + i, head = L.head()[ring];
            const bool hot = ring >= L.n_buckets;
            b = hot ? L.ring_b0 + (ring - L.n_buckets) : ring;
            n = head % kObLine;  // every full line went out when its last writer finished
            src = hot ? L.n_buckets + (ring - L.n_buckets) * kObRingLines + (head / kObLine) % kObRingLines : ring;
        }
        uint32_t dst = 0;
        if (n && piece == 0) dst = outbox_next_line(ob, L, b);
        dst = __shfl(dst, (int)((threadIdx.x & 63u) & ~(kObLine - 1u)));
        if (n) {
            const uint64_t m = piece < n ? L.stage()[src * kObLine + piece] : 0ull;  // (0 = add nothing to the bucket's first end)
            if (dst != kObNoLine) {
                TILE_COLD(ta.ob.pool)[(uint64_t)dst * kObLine + piece] = m;
            } else if (piece < n) {
                uint32_t off;
                const uint64_t d = outbox_unpack(ob, m, off);
                atomicAdd(TILE_COLD(ta.ob.spill) + (((uint64_t)b << ob.shift) | off), (unsigned long long)d);
                atomicAdd(TILE_COLD(ta.ob.overflow), 1ull);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const uint32_t ring = first + i;
        L.head()[ring] = 0;
        if (ring < L.n_buckets) {
            L.state()[ring] = 0;
        } else {
            for (uint32_t l = 0; l < kObRingLines; ++l) L.state()[L.n_buckets + (ring - L.n_buckets) * kObRingLines + l] = 0;
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
// Relaxation of a launch's far pulls: together they amount to this fraction of a projection (see the kernel).
// ONE projection: an end that h far terms pull in a launch moves by the mean of their h half-errors, which is what one term of
// the reference does to it — except in the first iterations: there the layout is globally inconsistent (a `-N d` layout has to
// contract by the share of nodes a path skips), the dozen-odd long-range pulls an end averages over are then a noisy
// estimate of where it should go, and the noise — not the mean — is what neighbouring nodes see (at a constant 0.5 the first
// iteration leaves neighbours ten thousand bp apart: sampled stress 20-60x the reference's at that point).  Noise and
// mean both scale with the factor, the reference itself does not converge globally before its learning rate falls
// (iteration ~15 of 30), so the first iterations pull gently and the factor ramps up while the windows' local terms and
// the mean field bring the layout together (tools/cpu_transient.py, profiles/r03/far_relax_schedules.txt).
// Rounds 3-5 ramped 0.1 0.1 0.2 0.3 0.4 to HALF a projection.  Measured in round 6 with the evaluator that has no sampling
// error (pgsgd_path_stress_near; profiles/r06/NOTES.md section 1) at 1e7 nodes: on the default schedule the ramp and its ceiling
// change nothing (tile kernel +1.2 % over the per-lane kernel with either), on a SHORT schedule (`-x 15 -G 2`) the slow ramp
// to 0.5 costs +7.3 %, this one +0.4 % (constant 0.75 / 1.0: -1.4 / -0.6 %; ramp 0.1.. to 1.0: +5.2 %; 0.25: +18 %).
constexpr uint64_t kFarGentleIterations = 5;   // the iterations of the ramp — and those whose far pulls always arrive before the very next launch (pgsgd_session.hip: pulls_urgent)
__host__ __device__ inline float tile_far_relax(uint64_t iteration /* 0-based */) {
    return iteration < 2 ? 0.2f : iteration < kFarGentleIterations ? 0.2f * (float)iteration : 1.0f;   // 0.2 0.2 0.4 0.6 0.8 1.0 ...
}

// What the tile kernel's sampler needs, trimmed: step indices and jump lengths fit 32 bits here (a tiled session has
// fewer than 2^32 path steps), and everything about a Zipf draw that depends on the jump length n alone comes from a
// table: zipf_tab[n] = {zeta_n, eta_n}.  The steps of a tile have consecutive ranks, so their jump lengths are two
// runs of consecutive n: the tile's 7 KB of the table stay in L2 while it runs, and the draw loses two fp64
// divisions, one exponent loop and the 32-bit division of the zeta cache index (a third of its instructions; the
// kernel is bound by VALU issue).
struct TileSampler {
    const double2* zipf_tab;  // [min(space, longest path) + 1]
    uint32_t space;
    int alpha_e;
    double alpha_frac, one_plus_half_pow;
};

// zipf_tab[n] for every jump length: the zeta cache entry the reference would look up (path_sgd_layout.cpp:208-212)
// and eta of dirty_zipfian_int_distribution, by the operations of zipf_tabled() in the same order — the same bits
__global__ void zipf_tab_kernel(const double2* zeta_denom, uint32_t space_max, uint32_t space_quant, int omt_e, double omt_frac, uint32_t n_entries, double2* out) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_entries) return;
    const double2 zd = zeta_denom[n > space_max ? space_max + (n - space_max) / space_quant + 1 : n];
    const double eta = (1.0 - pow_split(2.0 / (double)n, omt_e, omt_frac)) / zd.y;
    out[n] = make_double2(zd.x, eta);
}

// The tile kernel's draws.  A lane's stream yields TWO 64-bit words per term (the per-lane kernel and the reference
// worker draw five or six: one per coin).  Same distributions, fewer generator steps — the kernel is bound by VALU
// issue, and a Xoshiro256+ step is 15 instructions on 32-bit lanes:
//   word 1  bits 63..32  the first step, uniform in the tile: Lemire's multiply-and-reject on 32 bits (a rejected word
//                        is redrawn whole)
//           bit 31       Zipf or uniform partner (path_sgd_layout.cpp:205; ignored in a cooling iteration)
//           bit 30       direction of the Zipf jump (:206)
//           bit 29, 28   which end of the first step's node, of the partner's (:253, :262)
//           bits 27..14, 13..0   rounding dither of the x and y steps, 14 bits each
//   word 2  Zipf partner: the variate of generate_canonical, as in the reference's distribution;
//           uniform partner (:235-237): Lemire on bits 63..32 over the path's step count (< 2^32 in a tiled session)
// All fields are disjoint bits of one equidistributed 64-bit output, hence independent fair coins / uniforms; the
// generator's weak lowest bits only reach the low end of the y dither.  The oracle draws the same words
// (orc_tile_pick_first / orc_tile_partner) and tests/test_oracle.py compares the term distribution with the
// reference-order sampler's.
constexpr uint32_t kDitherBits = 14;
__device__ __forceinline__ uint32_t below32_hi(Xoshiro256Plus& g, uint32_t range, uint32_t& low_half) {
    uint64_t w = g.next();
    uint64_t m = (uint64_t)(uint32_t)(w >> 32) * range;
    if ((uint32_t)m < range) {
        const uint32_t threshold = (0u - range) % range;
        while ((uint32_t)m < threshold) {
            w = g.next();
            m = (uint64_t)(uint32_t)(w >> 32) * range;
        }
    }
    low_half = (uint32_t)w;
    return (uint32_t)(m >> 32);
}

// dirty-Zipf draw of zipf_tabled() with eta from the table: the same operations in the same order
__device__ __forceinline__ uint32_t zipf_tile(Xoshiro256Plus& g, const TileSampler& ts, uint32_t n, double zeta_n, double eta) {
    const double u = canonical(g);
    const double uz = u * zeta_n;
    if (uz < 1.0) return 1;
    if (uz < ts.one_plus_half_pow) return 2;
    const double v = 1.0 + (double)n * pow_split(eta * u - eta + 1.0, ts.alpha_e, ts.alpha_frac);
    // (uint64_t)v clamped to [1, n] with n < 2^32: whatever does not fit 32 bits is clamped to n either way
    uint32_t r = v >= 1.0 ? (v < 4294967296.0 ? (uint32_t)v : 0xffffffffu) : 1u;  // (NaN -> 1, as in zipf_tabled)
    if (r < 1) r = 1;
    if (r > n) r = n;
    return r;
}

// The Zipf/uniform coin of a warm iteration (path_sgd_layout.cpp:205) is drawn once per WAVE and TRIP, not per lane:
// the 64 terms a wave draws in one trip all take a Zipf partner or all a uniform one, so the wave runs one of the two
// partner paths instead of both under divergence (the Zipf draw is ~55 vector instructions, the uniform one ~40, and the
// kernel is bound by vector issue).  Every term is still a Zipf term with probability 1/2, independently of every term
// outside its wave's trip and of its own other draws — the reference's marginal distribution of terms; bit 31 of the
// term's first word, the per-lane coin of rounds 2 and 3, is ignored.  The coins of wave w of a tile are a SplitMix64
// stream seeded like a lane's generator with lane id 1023 - w (lane ids stop at 255): trip j takes bit j % 64 of output
// number j / 64.  The oracle draws the same coins (orc_tile_wave_coin).
static_assert(kTileBlock <= 1020, "lane ids must stay clear of the wave coin streams");
__host__ __device__ inline uint64_t tile_coin_seed(uint64_t seed_base, uint64_t epoch, uint64_t tile, uint32_t wave) {
    return seed_base + epoch * 0xd1342543de82ef95ull + ((tile << 10) | (1023u - wave));
}

// Partner pairs (rounds 4-6; quads since: tile_quad_partner below).  A uniform partner (path_sgd_layout.cpp:235-237) is a random step of the path: one 128-byte line from HBM
// per term that nothing else shares, and HBM delivers 51-55 G random lines per second (profiles/r04/NOTES.md section 6) —
// the tile kernel's warm iterations ran at that ceiling.  A 64-byte half of a line
// holds the records of TWO consecutive steps, so the lanes of a wave pair up in a uniform trip: the even lane draws its
// partner as before, the odd lane draws its own (its stream advances the same way) and then takes the step that shares
// the even lane's unit, flat step ^ 1, when that is a step of the path (it is, but for the one step at either end of a
// path whose first or last step has no twin: the odd lane then keeps its own draw).  Every term's partner is still
// uniform over the path's steps (the twin of a uniform step is a uniform step; the two end steps are chosen by odd lanes
// 1/cnt less often than the others, one part in 1e6 on a 1e6-step path); what changes is that the two terms of a pair
// pull towards neighbouring steps instead of independent ones — half as many distinct far places per iteration.
// Measured: warm iterations 0.475 -> 0.535 of the roofline figure, same final stress (profiles/r04/bench_pairs.txt);
// PGSGD_FLAG_NO_PARTNER_PAIRS turns it off; the oracle draws the same pairs (orc_tile_partner).
__host__ __device__ inline uint32_t tile_pair_partner(uint32_t lead_flat_step, uint32_t pstart, uint32_t cnt, uint32_t own_rank) {
    const uint32_t twin = (lead_flat_step ^ 1u) - pstart;
    return twin < cnt ? twin : own_rank;
}
// Partner QUADS (what sessions run since round 6's second session; pairs: rounds 4-6, debug knob PGSGD_TILE_PAIRS).  A 128-byte line of
// gather records holds FOUR consecutive steps (recs2: [s0 s1 s2 s3 | n0 n1 n2 n3]) and a read request moves the whole line, so lane r of
// four consecutive lanes takes flat step lead ^ r of the line its group's first lane drew — one memory request for four terms, every
// partner still uniform over the path's steps (the map step -> step ^ r permutes the steps of every line that lies inside the path; the
// up to three steps at either end of a path whose line is cut are chosen a shade less often, one part in 1e6 on a 1e6-step path), and
// the lane keeps its own draw where the line is cut.  Once the waits in front of the term loops were gone the kernel ran at 90 % of the
// memory's random-line rate in warm launches: quads take 24 % of their read requests away — warm launches 0.624 -> 0.645 of the roofline
// figure, the driver's window 0.667 -> 0.678, final stress unchanged (exact figure 0.20574 / 0.20583; profiles/r06/quads_ab.jsonl).
// The oracle draws the same quads (orc_tile_partner).
__host__ __device__ inline uint32_t tile_quad_partner(uint32_t lead_flat_step, uint32_t r, uint32_t pstart, uint32_t cnt, uint32_t own_rank) {
    const uint32_t twin = (lead_flat_step ^ r) - pstart;
    return r && twin < cnt ? twin : own_rank;
}

// MATH: kMathFast is the instance sessions run — the term's geometry with the hardware's reciprocal and reciprocal
// square root (1 ulp each, tile_displacement) and path positions as 32-bit words (every path shorter than 2^32 bp);
// kMathExact keeps the IEEE divisions and the correctly rounded square root of the per-lane kernel and 64-bit
// positions: bit for bit the oracle's mirror (PGSGD_FLAG_EXACT_MATH; the one-lane mirror tests; paths of 2^32 bp or more).
constexpr int kMathExact = 0, kMathFast = 1;
// The displacement of one term of the tile kernel (path_sgd_layout.cpp:280-352) from its path distance d, already a
// float, and dx, dy = p_a - p_b in bp.  kMathExact: the operations of term_displacement(), bit for bit.  kMathFast:
//   mu = min(eta * rcp(d), cap),  r = Delta / mag = (mu / 2) * (1 - d * rsq(dx^2 + dy^2)),  |Delta| = |r| * mag
// — one v_rcp_f32 and one v_rsq_f32 (1 ulp each) where the exact form takes two IEEE divisions and a correctly rounded
// square root (38 instructions).  r differs from the exact form's by at most ~4e-7 * mu / 2 in absolute terms (the
// cancellation in 1 - d / mag is taken in fp32 in both forms), i.e. a displacement error below 2^-21 of the pair's
// layout distance — tests/test_gpu_parity.py measures it on the device against the exact instance.
template <int MATH>
__device__ __forceinline__ void tile_displacement(float eta, float d, float dx, float dy, float mu_cap, float& r_x, float& r_y, float& abs_delta) {
    if (d == 0.0f) d = 1e-9f;
    if (dx == 0.0f) dx = 1e-9f;
    const float dx2 = dx * dx;
    const float dy2 = dy * dy;
    if (MATH == kMathFast) {
        float mu = eta * __builtin_amdgcn_rcpf(d);
        if (mu > mu_cap) mu = mu_cap;
        const float s = dx2 + dy2;
        const float rs = __builtin_amdgcn_rsqf(s);
        const float r = (0.5f * mu) * __builtin_fmaf(-d, rs, 1.0f);
        abs_delta = fabsf(r) * (s * rs);
        r_x = r * dx;
        r_y = r * dy;
    } else {
        const float w = 1.0f / d;
        float mu = eta * w;
        if (mu > mu_cap) mu = mu_cap;
        const float mag = sqrtf(dx2 + dy2);
        const float Delta = (mu * (mag - d)) / 2.0f;
        abs_delta = fabsf(Delta);
        const float r = Delta / mag;
        r_x = r * dx;
        r_y = r * dy;
    }
}
// (float)(a - b) of two unsigned 32-bit fields: magnitude and sign apart — the difference is exact as an unsigned
// integer and is rounded once, to nearest even, which is symmetric in the sign: the bits of (float)((int64_t)a - (int64_t)b)
__device__ __forceinline__ float field_diff(uint32_t a, uint32_t b) {
    const float m = (float)(a > b ? a - b : b - a);
    return a >= b ? m : -m;
}

// The stage registers of the term loop.  Every VGPR a stage keeps across a trip is held twice (two alternating sets, see the
// loop), and the kernel's occupancy is set by its VGPRs (round 2: 110 -> 4 waves per SIMD), so a stage keeps only
// what cannot be had again for a few instructions: step offsets instead of step records (the records are re-read from
// the tile's LDS copy where they are used), the generator word with the term's coins instead of the decoded coins,
// no validity flags (kNoTerm in `ka`: divergent flags live in SGPR pairs, and the kernel was spilling SGPRs).
constexpr uint32_t kNoTerm = 0xffffffffu;

// a term whose first step is picked and whose Zipf table entry is on its way
struct PickedTerm {
    double2 zd;      // {zeta_n, eta_n} for the jump (a Zipf term only)
    uint32_t ka;     // first step, as an offset inside the tile; kNoTerm: no term in this slot
    uint32_t flags;  // low half of the term's first word: coins and dither
};

// a term whose first step and partner are drawn and whose partner record is on its way
struct PendingTerm {
    uint4 rb_g;                // partner's record {handle, len, pos} from global memory (a partner outside the tile) with
    unsigned long long snapw;  // the snapshot word of the chosen end of its node — the one of the record's two the term reads
    uint32_t ka;               // kNoTerm: no term in this slot
    uint32_t kb_off;           // partner step minus the tile's first step: a step of the tile when < tile steps
    uint32_t flags;
};

// COOLING: the launch's iteration is a cooling one (every partner by Zipf) — known per launch, so the coin and the
// uniform-partner branch are compiled out.  LOCAL: the work items have a window (the usual case); window-less items
// (tiles of unsorted stretches) run in their own launch with every end read from global memory and every update sent
// through the outbox.
// Waves per SIMD the register allocation aims at (a workgroup is one wave per SIMD, so this is also the workgroups per
// CU): the kernel hides its LDS round trips and its gathers behind other waves.
#ifndef PGSGD_TILE_WAVES
#define PGSGD_TILE_WAVES 5
#endif
// LOCK: the instance with conflict resolution on the window's node ends (PGSGD_FLAG_LOCK_WINDOW_ENDS; an option, see the term loop).
// The hand-off of a window between consecutive parts (kItemHasNext) has no release/acquire fence: the words go out with
// agent-scope atomic stores, `s_waitcnt vmcnt(0)` waits for their acknowledgement (on gfx9 the one counter covers stores — gfx10+
// count them in vscnt), a relaxed flag follows, and the next part stages the window with agent-scope loads.  That is the memory
// model of gfx950 (and gfx942: write-through L2 at agent scope, store acknowledged when visible to the agent) and of no other
// target: this file is built for those only, and the windowed instance must read coordinates with the agent-scope flavour.
// This file is built for gfx950 ONLY: besides the hand-off, far_drain_kernel accumulates a bucket in 128 KiB of a workgroup's LDS
// (160 KB per CU on gfx950; gfx942 has 64 KB) and the tile kernel's LDS budget assumes five workgroups in those 160 KB.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "pgsgd_tiles.hpp is written for gfx950 (MI355X): the fence-free window hand-off and the 128-KiB drain accumulators do not carry over"
#endif
template <int COORD_LOAD, int FAR, bool COOLING, bool LOCAL, int MATH = kMathFast, bool LOCK = false, int ABL = 0, uint32_t PUSH = 1>
__global__ __launch_bounds__(kTileBlock) __attribute__((amdgpu_waves_per_eu(PGSGD_TILE_WAVES, PGSGD_TILE_WAVES))) void sgd_tile_kernel(DevConst c, TileArgs ta, TileSampler ts, IterArgs a) {
    static_assert(COORD_LOAD == 1, "the tile kernel stages windows with agent-scope loads: a part of a window may have been written by another workgroup of this launch");
    // PUSH: messages a lane hands to the rings per call of their protocol.  Two in the warm launches of a session whose longer wave queues do not cost
    // it a workgroup per CU (pgsgd_session.hip: tile_push) — every uniform term sends a message, scattered over all buckets, and half as many calls buy
    // 2 ... 3.5 % of the launch — one in a cooling launch, whose messages go to the two hot rings at the window and wait for each other's lines there when
    // a call brings 128 (measured: -0.7 %; profiles/r06/NOTES.md section 15).
    static_assert(PUSH == 1 || (PUSH == 2 && !COOLING), "two messages per lane and call: warm instances only");
    constexpr uint32_t kPush = PUSH;
    extern __shared__ uint64_t lds[];
    uint64_t* win = lds;                                                         // [4R] window words
    uint4* trec = reinterpret_cast<uint4*>(lds + 4 * (size_t)ta.region);         // [T] tile records
    OutboxLds L;
    L.n_buckets = ta.ob.n_buckets;
    L.wq_cap = tile_wq_cap(PUSH);   // (a constant of the instance: the launch brings the LDS for it, pgsgd_session.hip: tile_lds_for)
    {
        L.base = (uint32_t)(4 * (size_t)ta.region * sizeof(uint64_t) + (size_t)ta.tile_steps * sizeof(uint4));  // behind the window and the tile records
        for (uint32_t i = threadIdx.x; i < L.n_buckets + kObRings + L.lines(); i += blockDim.x) L.head()[i] = 0;  // heads and line states
    }
    // one lock bit per window word (tile_lock_words(region) 32-bit words behind the queues' bucket bytes)
    uint32_t* lockw = reinterpret_cast<uint32_t*>(L.wq_b() + kTileWaves * L.wq_cap);
    for (uint32_t i = threadIdx.x; i < tile_lock_words(ta.region); i += blockDim.x) lockw[i] = 0;
    uint32_t n_locked = 0, n_lost = 0;
    // profiling instance 5 (make libpgsgd_x5.so): where a workgroup's time goes, summed by its thread 0 in 100 MHz ticks —
    // [0] taking an item .. its first tile, [1] a tile's start .. its term loop, [2] the term loop, [3] the loop's end .. the tile's end
    // (snapshot pieces), [4] the item's last tile .. the item's end (queue flush, window write-back), [5] everything
    uint64_t ph[6] = {0, 0, 0, 0, 0, 0};
    const uint64_t ph_start = ABL == 5 ? wall_clock64() : 0;
    if (TILE_COLD(ta.ob.n_buckets) != ta.ob.n_buckets || TILE_COLD(c.n_nodes) != c.n_nodes) __builtin_trap();  // the argument segment is not laid out as TileKernelArgs says
    if (blockIdx.x == 0 && threadIdx.x == 0 && TILE_COLD(ta.clock_probe)) {  // (workgroups are persistent: workgroup 0 lives as long as the launch has work)
        unsigned long long* probe = TILE_COLD(ta.clock_probe);
        probe[0] = __builtin_readcyclecounter();   // s_memtime: shader cycles
        probe[1] = wall_clock64();                 // s_memrealtime: constant 100 MHz
    }
    const uint64_t wg_start = TILE_COLD(ta.tail_probe) ? wall_clock64() : 0;  // (debug: how much of a launch its workgroups are alive, pgsgd_session_tile_tail)
    // this wave's message queue; its fill count is wave-uniform and lives in a scalar register
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveQueue wq;
    wq.msg = L.wq_msg() + wave * L.wq_cap;
    wq.b = L.wq_b() + wave * L.wq_cap;
    uint32_t wq_n = 0;
    L.ring_b0 = 0x7fffffffu;
    __shared__ uint32_t s_item;
    for (uint32_t b = threadIdx.x; b < L.n_buckets; b += blockDim.x) {
        L.line()[b] = (kObNone << kObUsedBits) | kObLinesPerGroup;
        L.chunk0()[b] = TILE_COLD(ta.ob.chunk0)[b];
    }
    float dmax = 0.0f;
    uint32_t n_far = 0;
    uint32_t n_done = 0;   // terms this wave's lanes have finished (wave-uniform: a ballot's population count per trip, scalar arithmetic)
    bool guard = false;  // a coordinate in the outer quarter of the fixed-point frame was seen (in_frame_guard)
    const uint64_t n_ends = 2 * (uint64_t)c.n_nodes;
    const uint32_t win_words = 4 * ta.region;
    // A partner outside the window is read from the last snapshot, and what the term adds to it reaches its owner
    // after the launch: all the far pulls an end receives during one launch are computed against one stale position
    // and land together — a Jacobi step.  With mu = 1 each is a full projection and h of them overshoot h-fold (stress
    // 1e7 in the first iterations, profiles/r01/convergence_*.jsonl), so such terms are capped at mu = far_relax / h,
    // h = far pulls per node end in the previous launch of this colour: together they amount to one projection (rounds 2-5:
    // half, the usual under-relaxation of a Jacobi step; measured again in round 6: tile_far_relax),
    // less in the first iterations (tile_far_relax).  Inactive once eta/d < far_relax / h.
    float far_mu_cap = TILE_COLD(ta.far_mu_cap_first);
    if (TILE_COLD(ta.far_from_prev)) {
        const double h = (double)*TILE_COLD(ta.far_prev) / (double)n_ends;
        far_mu_cap = h > 1.0 ? (float)(1.0 / h) : 1.0f;
    }
    far_mu_cap *= TILE_COLD(ta.far_relax);
    uint32_t my_queue = 0, queues_done = 0;  // (lane 0's copies are the ones used)
    if (gridDim.x >= kItemQueues && tile_kernarg<uint32_t>((uint32_t)__builtin_offsetof(TileKernelArgs, ta.chunk) + 4u) != tile_kernarg<uint32_t>((uint32_t)__builtin_offsetof(TileKernelArgs, ta.chunk) + 4u * kItemQueues)) {  // more than one run, and enough workgroups to have one per XCD
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        my_queue = xcc & (kItemQueues - 1u);
    }
    // Taking an item used to be a chain of six dependent round trips in front of its first tile — the queue's returning atomic, the
    // item's record (thread 0, for the window it waits for), the record again (everybody), the window's words, the tiles' numbers, the
    // first tile's step records: 7.5 us of an item's ~150 (profiles/r06/NOTES.md section 10).  Now thread 0 passes the record on through
    // LDS, the tiles' numbers travel with the window's words, and the first tile's step records follow as soon as those have arrived.
    // (Measured and NOT taken: thread 0 taking the next item while the current one starts, so that record, numbers and step records are
    // there when the item begins — 6 % SLOWER: an item held back by a busy workgroup is what the next part of its window waits for, and
    // with two items per workgroup in flight the queue's distance between the parts of a window — 1 953 items, 1.5 rounds — is used up.)
    __shared__ WorkItem s_wi;        // the current item's record
    auto claim = [&]() -> uint32_t {   // (thread 0) the next item of this workgroup's runs
        for (; queues_done < kItemQueues; ++queues_done) {  // a run that is used up stays used up: never asked again
            const uint32_t q = (my_queue + queues_done) & (kItemQueues - 1u);
            const uint32_t lo = tile_kernarg<uint32_t>((uint32_t)__builtin_offsetof(TileKernelArgs, ta.chunk) + 4u * q), hi = tile_kernarg<uint32_t>((uint32_t)__builtin_offsetof(TileKernelArgs, ta.chunk) + 4u * (q + 1u));
            const uint32_t cand = lo + atomicAdd(TILE_COLD(ta.queue) + q, 1u) * TILE_COLD(ta.shard_world) + TILE_COLD(ta.shard_rank);
            if (cand < hi) return cand;
        }
        return kNoItem;
    };
    uint32_t tb_first = kNoItem;   // the lanes hold the numbers of tiles tb_first .. tb_first + 63 of the current item (tb_* below)
    uint32_t tb_t0 = 0, tb_nl = 0, tb_terms = 0, tb_pstart = 0, tb_cnt = 0;   // this lane's tile of the batch
    uint4 r_pref = make_uint4(0, 0, 0, 0);   // step record threadIdx.x of tile pref_k of the current item, requested ahead
    uint32_t pref_k = kNoItem;
    for (;;) {
        uint64_t ph_t = ABL == 5 ? wall_clock64() : 0;
        if (threadIdx.x == 0) {
            const uint32_t it = claim();
            if (it != kNoItem) s_wi = TILE_COLD(ta.items)[it];
            if (LOCAL && it != kNoItem) {  // a later part of a window waits until the part before it has written the window back
                const uint32_t dep = s_wi.local >> kItemDepShift;
                if (dep) {
                    const uint32_t* flag = TILE_COLD(ta.item_done) + (dep - 1u);
                    const uint32_t stamp = TILE_COLD(ta.stamp);
                    const uint64_t t_wait = wall_clock64();
                    // (relaxed: an acquire at agent scope invalidates the XCD's L2 on every poll, and the words that follow are read with agent-scope loads anyway)
                    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != stamp) {
                        __builtin_amdgcn_s_sleep(32);
                        // a FATAL diagnostic, never seen: 5 s of 100 MHz ticks without the predecessor's flag aborts the launch (the HIP
                        // context is lost, every later call returns PGSGD_E_HIP) instead of hanging the device
                        if (wall_clock64() - t_wait > 500000000ull) __builtin_trap();
                    }
                }
            }
            s_item = it;
        }
        __syncthreads();
        const uint32_t item = s_item;
        if (item == kNoItem) break;
        const WorkItem wi = s_wi;
        const uint32_t wbase = 2 * wi.win0;  // first coordinate word of the window
        const uint32_t n_item_tiles = wi.tile_end - wi.tile_begin;
        const uint32_t rot = TILE_COLD(ta.tile_rotate) && n_item_tiles ? (uint32_t)(((uint32_t)a.epoch * 0x9e3779b1u + wi.win0) % n_item_tiles) : 0u;
        // What a tile needs before its first term — its record, its share of the terms, its path's extent, its step records — used to be
        // fetched when the tile started: three dependent round trips to memory in front of every term loop, 4.0 us (warm) / 2.4 us (cooling)
        // of a tile's 33 / 26 us with the workgroup's four waves idle (profiles/r06/NOTES.md section 10).  Now lane l of every wave
        // fetches the numbers of the item's l-th tile when the item starts (64 tiles per batch; one 16-byte head and two term counts per
        // tile, beside the window's words), a tile reads its own with v_readlane, and the step records of the NEXT tile are requested
        // when a tile starts.  The same tiles in the same order with the same numbers: nothing about a term changes.
        const uint32_t wlane = threadIdx.x & 63u;
        auto tile_of = [&](uint32_t k) -> uint32_t { return wi.tile_begin + (k + rot >= n_item_tiles ? k + rot - n_item_tiles : k + rot); };
        auto load_batch = [&](uint32_t first_k) {
            const uint32_t k = first_k + wlane;
            if (k < n_item_tiles) {
                const uint32_t ti = tile_of(k);
                const uint4 h = TILE_COLD(ta.tile_heads)[ti];
                const uint64_t* term0 = TILE_COLD(ta.term0);
                const uint32_t h_lanes = h.y >> 16;
                tb_t0 = h.x;
                tb_nl = (h.y & 0xffffu) | ((h_lanes < blockDim.x ? h_lanes : blockDim.x) << 16);
                tb_terms = (uint32_t)(term0[ti + 1] - term0[ti]);
                tb_pstart = h.z;
                tb_cnt = h.w;
            }
        };
        auto runs = [&](uint32_t k) -> bool { return tile_of(k) % TILE_COLD(ta.n_sub) == TILE_COLD(ta.sub); };  // (multi-GPU by tile, sub-steps: block-uniform)
        tb_first = kNoItem;
        if (n_item_tiles) {   // the first batch's numbers travel with the window's words
            tb_first = 0;
            load_batch(0);
        }
        if (LOCAL) {
            L.ring_b0 = wbase >> ta.ob.shift;  // the hot rings serve the window's bucket and the next one
            for (uint32_t i = threadIdx.x; i < win_words; i += blockDim.x)
                win[i] = (uint64_t)wbase + i < n_ends ? load_word<COORD_LOAD>(LOCAL ? TILE_COLD(c.coords) : c.coords, wbase + i) : 0;
        }
        pref_k = kNoItem;
        if (n_item_tiles && runs(0)) {   // ... and the first tile's step records follow them
            pref_k = 0;
            const uint32_t nt0 = (uint32_t)__builtin_amdgcn_readlane((int)tb_t0, 0);
            const uint32_t nn = (uint32_t)__builtin_amdgcn_readlane((int)tb_nl, 0) & 0xffffu;
            if (threadIdx.x < nn) r_pref = TILE_COLD(c.recs)[(uint64_t)nt0 + threadIdx.x];
        }
        for (uint32_t tk = 0; tk < n_item_tiles; ++tk) {
            const uint32_t ti = tile_of(tk);
            if (!runs(tk)) continue;
            if (tb_first != (tk & ~63u)) {   // the lanes hold another batch of the item's tiles
                tb_first = tk & ~63u;
                load_batch(tb_first);
            }
            if (ABL == 5) { const uint64_t now = wall_clock64(); ph[tk ? 3 : 0] += now - ph_t; ph_t = now; }
            const uint32_t bl = tk & 63u;
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)tb_t0, (int)bl), nl = (uint32_t)__builtin_amdgcn_readlane((int)tb_nl, (int)bl);
            const uint32_t n_tile_terms = (uint32_t)__builtin_amdgcn_readlane((int)tb_terms, (int)bl);
            const uint32_t pstart = (uint32_t)__builtin_amdgcn_readlane((int)tb_pstart, (int)bl);
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)tb_cnt, (int)bl);
            struct { uint32_t n; } t;   // (the tile's step count, under the name the stages below use)
            t.n = nl & 0xffffu;
            const uint32_t lanes = nl >> 16;
            __syncthreads();  // previous tile's terms are done with trec; window staging is complete
            if (pref_k == tk) {
                if (threadIdx.x < t.n) trec[threadIdx.x] = r_pref;
                for (uint32_t i = threadIdx.x + blockDim.x; i < t.n; i += blockDim.x) trec[i] = TILE_COLD(c.recs)[(uint64_t)t0 + i];
            } else {
                for (uint32_t i = threadIdx.x; i < t.n; i += blockDim.x) trec[i] = TILE_COLD(c.recs)[(uint64_t)t0 + i];
            }
            __syncthreads();
            pref_k = kNoItem;
            {   // the next tile's step records are on their way while this one runs its terms
                uint32_t nk = tk + 1;
                while (nk < n_item_tiles && (nk & 63u) != 0 && !runs(nk)) ++nk;
                if (nk < n_item_tiles && (nk & 63u) != 0) {   // (a tile that opens a batch has its numbers fetched first)
                    pref_k = nk;
                    const uint32_t nt0 = (uint32_t)__builtin_amdgcn_readlane((int)tb_t0, (int)(nk & 63u));
                    const uint32_t nn = (uint32_t)__builtin_amdgcn_readlane((int)tb_nl, (int)(nk & 63u)) & 0xffffu;
                    if (threadIdx.x < nn) r_pref = TILE_COLD(c.recs)[(uint64_t)nt0 + threadIdx.x];
                }
            }
            const bool worker = threadIdx.x < lanes;
            Xoshiro256Plus rng;
            const uint64_t seed_base = TILE_COLD(ta.seed_base);
            if (worker) rng.seed(tile_stream_seed(seed_base, a.epoch, ti, threadIdx.x));
            uint64_t coin_x = tile_coin_seed(seed_base, a.epoch, ti, wave);   // (scalar registers: the wave's coin stream)
            uint64_t coin_cur = COOLING ? 0ull : Xoshiro256Plus::splitmix64(coin_x);
            // The same trip count for every lane of the workgroup (the outbox is wave-cooperative), and two trips more
            // than the longest lane needs.  Every trip finishes term j - 2, draws the partner of term j - 1 (and requests
            // its record) and picks the first step of term j (and requests its Zipf table entry): whatever a trip loads
            // from global memory is consumed at the start of the next one.  (gfx9 counts loads and stores in one counter
            // and cannot wait for a particular load while stores are pending: with a single consumption point per trip the
            // one full wait falls where everything outstanding is a trip old.)  A lane's stream yields its terms' draws in
            // term order.
            const uint32_t trips = (n_tile_terms + lanes - 1) / lanes;
            // Two sets of stage registers, used alternately: a trip WRITES one set and READS the other, so nothing that
            // is still in flight has to be copied between registers at the end of a trip (a copy is a use: it would wait
            // for the load right where it was issued).
            PickedTerm K0, K1;
            PendingTerm Q0, Q1;
            K0.ka = K1.ka = kNoTerm;
            Q0.ka = Q1.ka = kNoTerm;
            // (stages take and return their registers BY VALUE: structs reached through references stay in memory, and a
            // select between two members becomes a load from a selected address — scratch traffic and flat loads)
            // what the first word says about the partner (path_sgd_layout.cpp:205-206), from the step offset and the coins
            auto jump_of = [&](uint32_t ka, uint32_t flags, uint32_t& s_rank, bool& zipf, bool& back) -> uint32_t {
                s_rank = t0 + ka - pstart;
                zipf = COOLING || (flags >> 31);
                back = (s_rank > 0 && ((flags >> 30) & 1u)) || s_rank == cnt - 1;
                const uint32_t room = back ? s_rank : cnt - s_rank - 1;
                return ts.space < room ? ts.space : room;
            };
            auto pick_stage = [&](uint32_t j) -> PickedTerm {
                PickedTerm Kw;
                Kw.ka = kNoTerm;
                Kw.flags = 0;
                if (worker && j < trips && threadIdx.x + j * lanes < n_tile_terms) {
                    // first step: uniform inside the tile; the reference's coins (path_sgd_layout.cpp:206,253,262) from the same word
                    Kw.ka = below32_hi(rng, t.n, Kw.flags);
                    // the Zipf/uniform coin (:205) is the wave's for this trip (tile_coin_seed); it rides in bit 31 of the flags
                    if (!COOLING && !ta.lane_coin) Kw.flags = (Kw.flags & 0x7fffffffu) | ((uint32_t)(coin_cur >> (j & 63u)) << 31);
                    uint32_t s_rank;
                    bool zipf, back;
                    const uint32_t jump = jump_of(Kw.ka, Kw.flags, s_rank, zipf, back);
                    if (zipf) Kw.zd = ts.zipf_tab[jump];
                }
                return Kw;
            };
            auto partner_stage = [&](const PickedTerm Kr) -> PendingTerm {
                PendingTerm Qw;
                Qw.ka = kNoTerm;
                if (Kr.ka != kNoTerm) {
                    // partner by the reference's rule (:207-237) from the term's second word
                    Qw.ka = Kr.ka;
                    Qw.flags = Kr.flags;
                    uint32_t s_rank, b_rank;
                    bool zipf, back;
                    const uint32_t jump = jump_of(Kr.ka, Kr.flags, s_rank, zipf, back);
                    if (zipf) {
                        const uint32_t z = zipf_tile(rng, ts, jump, Kr.zd.x, Kr.zd.y);
                        b_rank = back ? s_rank - z : s_rank + z;
                    } else {
                        uint32_t unused;
                        b_rank = below32_hi(rng, cnt, unused);
                        // partner pairs (tile_pair_partner): an odd lane takes the step that shares a 64-byte unit with its even
                        // neighbour's partner — one memory request for the two of them
                        if (ta.pair_uniform == 1u) {
                            const uint32_t lead = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(pstart + b_rank), 0xA0, 0xf, 0xf, false);  // quad_perm [0,0,2,2]: the even lane's step
                            if (threadIdx.x & 1u) b_rank = tile_pair_partner(lead, pstart, cnt, b_rank);
                        } else if (ta.pair_uniform == 2u) {  // partner quads (tile_quad_partner): four lanes share a 128-byte line of four records
                            const uint32_t lead = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(pstart + b_rank), 0x00, 0xf, 0xf, false);  // quad_perm [0,0,0,0]: the quad's first lane's step
                            b_rank = tile_quad_partner(lead, threadIdx.x & 3u, pstart, cnt, b_rank);
                        }
                    }
                    // the partner's record: the tile's LDS copy when it is a step of the tile (read where it is used), otherwise
                    // ONE 128-byte line of gather records: the static piece and, of the coordinates both ends of its node had at the last snapshot, the
                    // word of the end the term's coin chose (:262, bit 28)
                    const uint32_t kb = pstart + b_rank;
                    Qw.kb_off = kb - t0;
                    if (!(Qw.kb_off < t.n)) {
                        Qw.rb_g = ta.recs2[recs2_static_piece(kb)];
                        Qw.snapw = reinterpret_cast<const unsigned long long*>(ta.recs2)[2 * recs2_snap_piece(kb) + ((Kr.flags >> 28) & 1u)];
                    }
                }
                return Qw;
            };
            struct FarMessages { int32_t qx, qy; uint32_t end_a, end_b; bool to_a, to_b; };
            auto finish_stage = [&](const PendingTerm Qr) -> FarMessages {
                bool msg_a = false, msg_b = false;
                uint32_t end_a = 0, end_b = 0;
                int32_t mqx = 0, mqy = 0;
                n_done += (uint32_t)__popcll(__ballot(Qr.ka != kNoTerm));
                if (Qr.ka != kNoTerm) {
                    const uint32_t flip_a = (Qr.flags >> 29) & 1u, flip_b = (Qr.flags >> 28) & 1u;  // the two end choices (:253,262)
                    const bool from_global = !(Qr.kb_off < t.n);
                    const uint4 ra = trec[Qr.ka];
                    uint4 rb = Qr.rb_g;
                    if (!from_global) rb = trec[Qr.kb_off];
                    // the path position moves to the chosen end of each node (:242-269); d = |pos_a - pos_b| as a float
                    float d;
                    if (MATH == kMathFast) {  // every path is shorter than 2^32 bp: 32-bit positions, the same float
                        const uint32_t pa = ra.z + (flip_a ? ra.y : 0u), pb = rb.z + (flip_b ? rb.y : 0u);
                        d = (float)(pa > pb ? pa - pb : pb - pa);
                    } else {
                        uint64_t pos_a = (uint64_t)ra.z | ((uint64_t)ra.w << 32), pos_b = (uint64_t)rb.z | ((uint64_t)rb.w << 32);
                        if (flip_a) pos_a += ra.y;
                        if (flip_b) pos_b += rb.y;
                        // positions are below 2^52 (checked when the session is created), so the distance converts to fp64 exactly
                        // from its two halves and is rounded once on the way to fp32: the bits of (float)(uint64_t)
                        const int64_t diff = (int64_t)pos_a - (int64_t)pos_b;
                        const uint64_t ad = (uint64_t)(diff < 0 ? -diff : diff);
                        d = (float)((double)(uint32_t)(ad >> 32) * 4294967296.0 + (double)(uint32_t)ad);
                    }
                    end_a = ra.x ^ flip_a;
                    end_b = rb.x ^ flip_b;
                    // ends inside the staged window live in LDS (unsigned compare covers "below the window")
                    // A tile with a window has all its nodes inside it (that is what binds it to the window): the first end is
                    // always in LDS, and so is a partner that is a step of the tile.  Only window-less tiles read global words
                    // here (no dead global loads in the windowed instance: the compiler could not count what is in flight).
                    const uint32_t la = end_a - wbase, lb = end_b - wbase;
                    const bool in_a = LOCAL, in_b = LOCAL && lb < win_words;
                    // Conflict resolution on shared node coordinates.  256 lanes work on a window of ~1 000 node ends: an eighth of
                    // the ends a wave touches in a trip are touched by another lane at the same time, and two projections computed
                    // from the same stale position add up to twice the move.  While a term's learning rate is large (mu >= lock_mu:
                    // the projection regime) it therefore takes a lock bit on each of its window ends before it reads them — one LDS
                    // atomic OR per end; the lanes of a wave that go for the same end are served one after the other by the LDS, and
                    // exactly one of them finds the bit clear — and a term that finds an end taken does nothing at all (like a
                    // lost update of the reference's Hogwild loop, path_sgd_layout.cpp:360-363, but of the whole term: the sums of
                    // the coordinates stay conserved).  The owner clears its bits after its atomic adds.  A one-lane run never
                    // loses a term: the mirror tests are unaffected.
                    bool own_a = false, own_b = false, lost = false;
                    if (LOCK && LOCAL && ta.lock_mu > 0.0f) {
                        const float mu0 = a.eta * __builtin_amdgcn_rcpf(d > 0.0f ? d : 1e-9f);
                        if (mu0 >= ta.lock_mu) {
                            ++n_locked;
                            const uint32_t bit_a = 1u << (la & 31u);
                            own_a = !(atomicOr(lockw + (la >> 5), bit_a) & bit_a);
                            lost = !own_a;
                            if (in_b && lb != la) {
                                const uint32_t bit_b = 1u << (lb & 31u);
                                own_b = !(atomicOr(lockw + (lb >> 5), bit_b) & bit_b);
                                lost = lost || !own_b;
                            }
                        }
                    }
                    uint64_t wa, wb;
                    if (LOCAL) {
                        wa = win[la];
                        wb = in_b ? win[lb] : Qr.snapw;  // live word, or the snapshot that came with the record
                    } else {
                        wa = load_word<COORD_LOAD>(c.coords, end_a);
                        wb = from_global ? Qr.snapw : load_word<COORD_LOAD>(c.coords, end_b);
                    }
                    const float dx = field_diff((uint32_t)wa, (uint32_t)wb) * c.xf.inv_scale;
                    const float dy = field_diff((uint32_t)(wa >> 32), (uint32_t)(wb >> 32)) * c.xf.inv_scale;
                    float r_x, r_y, abs_delta;
                    const bool one_sided = FAR == kFarExclusive && !in_b;
                    tile_displacement<MATH>(a.eta, d, dx, dy, (in_b || one_sided) ? 1.0f : far_mu_cap, r_x, r_y, abs_delta);
                    if (one_sided) {
                        r_x *= 2.0f;
                        r_y *= 2.0f;
                    }
                    dmax = fmaxf(dmax, abs_delta);
                    const uint32_t dither = Qr.flags & ((1u << 2 * kDitherBits) - 1u);
                    const float ux = (float)(dither >> kDitherBits) * (1.0f / (float)(1u << kDitherBits));
                    const float uy = (float)(dither & ((1u << kDitherBits) - 1u)) * (1.0f / (float)(1u << kDitherBits));
                    float fx = r_x * c.xf.scale;
                    float fy = r_y * c.xf.scale;
                    fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
                    fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
                    const int32_t qx = (int32_t)floorf(fx), qy = (int32_t)floorf(fy);  // (clamped to the 32-bit range above)
                    // a step that rounds to no quantum adds zero: nothing to send, in particular no message
                    // for a far partner (most far terms of the late iterations, where eta / d^2 is tiny)
                    if (LOCK && lost) ++n_lost;
                    if ((qx | qy) != 0 && !(LOCK && lost)) {
                        n_far += (in_b || one_sided) ? 0u : 1u;
                        mqx = qx;
                        mqy = qy;
                        const uint64_t delta = outbox_delta(qx, qy);
                        if (in_b) atomicAdd(reinterpret_cast<unsigned long long*>(win + lb), (unsigned long long)delta);
                        else msg_b = !one_sided;
                        if (in_a) atomicAdd(reinterpret_cast<unsigned long long*>(win + la), (unsigned long long)(0ull - delta));
                        else msg_a = true;
                    }
                    if (LOCK && own_a) atomicAnd(lockw + (la >> 5), ~(1u << (la & 31u)));   // (LDS operations of a wave execute in order: after the adds)
                    if (LOCK && own_b) atomicAnd(lockw + (lb >> 5), ~(1u << (lb & 31u)));
                }
                if (ABL == 1) msg_a = msg_b = false;  // profiling instances (results invalid; experiment builds libpgsgd_x<N>.so only): 1 far updates are dropped,
                return FarMessages{mqx, mqy, end_a, end_b, msg_a, msg_b};
            };
            // messages wait in the wave's queue; the rings' protocol runs when 64 are there, one per lane
            auto send = [&](const FarMessages m) {
                wq_append(ta.ob, wq, wq_n, m.to_b, m.end_b, m.qx, m.qy);
                if (wq_n >= ta.wq_threshold) { if (ABL == 3) wq_n = 0; else wq_push<kPush>(ta.ob, L, wq, wq_n); }  // 3 messages are queued but never leave the wave's queue,
                if (!LOCAL) {
                    wq_append(ta.ob, wq, wq_n, m.to_a, m.end_a, -m.qx, -m.qy);
                    if (wq_n >= ta.wq_threshold) wq_push<kPush>(ta.ob, L, wq, wq_n);
                }
            };
            if (ABL == 5) { const uint64_t now = wall_clock64(); ph[1] += now - ph_t; ph_t = now; }
            for (uint32_t j = 0; j <= trips + 1; j += 2) {  // (a trip past the end finds nothing valid and does nothing)
                if (!COOLING && j && !(j & 63u)) coin_cur = Xoshiro256Plus::splitmix64(coin_x);  // the wave's next 64 coins
                // one trip: everything the previous trip requested is consumed FIRST (one wait, for loads that have been
                // in flight for a whole trip), then this trip's requests go out, then its messages
                FarMessages m = finish_stage(Q1);   // term j - 2
                Q0 = partner_stage(K1);             // term j - 1: Zipf entry arrived; requests the partner's record
                K0 = pick_stage(j);                 // term j: requests its Zipf entry
                send(m);
                m = finish_stage(Q0);
                Q1 = partner_stage(K0);
                K1 = pick_stage(j + 1);
                send(m);
            }
            // The tile's terms are done: rewrite the snapshot pieces of its OWN steps' records — the coordinates of the step's
            // node from the window (a tile with a window has all its nodes in it) — so that
            // a partner outside somebody's window is seen as its tile last left it, this iteration or the one before,
            // and no pass over all the records is needed between iterations (snapshot_kernel: 0.45 ms per iteration at
            // config 4, a twentieth of the iteration).  Four consecutive lanes write the four snapshot pieces of a 128-byte group:
            // one whole 64-byte unit (only the units at a tile's two ends are shared with the neighbouring tile), and the static
            // pieces are never written again (recs2_snap_piece).  Readers in other workgroups may see a record's old or new
            // words (each 8-byte word is written whole); the far pulls the drain delivers after the launch reach the
            // records when the tile runs again — the same staleness the per-iteration pass had (tools/cpu_transient.py).
            if (ABL == 5) { const uint64_t now = wall_clock64(); ph[2] += now - ph_t; ph_t = now; }
            uint4* const recs2_out = TILE_COLD(ta.recs2_out);
            if (ABL != 2 && recs2_out && (TILE_COLD(ta.snap_every) <= 1u || ((uint32_t)a.epoch + ti) % TILE_COLD(ta.snap_every) == 0u)) {  // (experiment knob PGSGD_TILE_SNAP_EVERY: every k-th iteration, a k-th of the tiles each)
                __syncthreads();
                for (uint32_t i = threadIdx.x; i < t.n; i += blockDim.x) {  // the two ends of the step's node, the one the step enters first in front
                    const uint32_t e0 = trec[i].x;
                    const uint64_t w0 = LOCAL ? win[e0 - wbase] : load_word<COORD_LOAD>(c.coords, e0);
                    const uint64_t w1 = LOCAL ? win[(e0 ^ 1u) - wbase] : load_word<COORD_LOAD>(c.coords, e0 ^ 1u);
                    // (plain stores; write-through (sc1) and non-temporal ones measured the same: profiles/r03/bench_variants_call4.txt)
                    recs2_out[recs2_snap_piece((uint64_t)t0 + i)] = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
                }
            }
        }
        // (what still waits in the wave's queue stays there: a message finds its bucket's ring — hot or not — whenever it is pushed, so
        // the queue carries over to the next item and a call of the rings' protocol per wave and item is saved; flushed when the
        // workgroup is out of items.  ABL 2: the tiles do not rewrite their snapshot pieces.)
        __syncthreads();
        if (LOCAL) {  // the window's only writer since it was staged: plain, coalesced stores
            const bool has_next = (wi.local & kItemHasNext) != 0;  // ... unless the window's next part may be staged by another workgroup in this launch
            for (uint32_t i = threadIdx.x; i < win_words; i += blockDim.x)
                if ((uint64_t)wbase + i < n_ends) {
                    if (has_next) __hip_atomic_store(TILE_COLD(c.coords) + wbase + i, win[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (staging reads with agent-scope loads: load_word)
                    else TILE_COLD(c.coords)[wbase + i] = win[i];
                    guard |= in_frame_guard(win[i]);
                }
            // the hot rings move on with the window: write out what they hold and unbind them (while the window's words travel)
            outbox_flush_rings(ta.ob, L, L.n_buckets, kObRings);
            // every thread's words are out before the flag goes up: the write-through stores have completed when the counter is
            // back at zero (a release fence at agent scope would also write back the XCD's whole L2: measured, 200 us per item)
            if (has_next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(TILE_COLD(ta.item_done) + item, TILE_COLD(ta.stamp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();  // s_item and the window are reused
        if (ABL == 5) ph[4] += wall_clock64() - ph_t;   // (from the last tile's term loop on: its snapshot pieces are counted here, not in [3])
    }
    // what still waits in the waves' queues, then the partly filled lines (the hot rings' too: the last pushes may have used them);
    // close the chunks this workgroup still has open
    while (wq_n) { if (ABL == 3) wq_n = 0; else wq_push<kPush>(ta.ob, L, wq, wq_n); }
    __syncthreads();
    outbox_flush_rings(ta.ob, L, 0, L.n_buckets + kObRings);
    for (uint32_t b = threadIdx.x; b < L.n_buckets; b += blockDim.x) {
        const uint32_t lp = L.line()[b], chunk = lp >> kObUsedBits, used = lp & kObUsedMask;  // lines written into the open group
        if (chunk < kObOverflow)
            for (uint32_t k = 0; k < kObGroup; ++k) {
                const uint32_t lines = used > k * kObLinesPerChunk ? (used - k * kObLinesPerChunk < kObLinesPerChunk ? used - k * kObLinesPerChunk : kObLinesPerChunk) : 0u;
                TILE_COLD(ta.ob.fill)[L.chunk0()[b] + chunk + k] = lines * kObLine;
            }
    }
    unsigned long long* const probe = TILE_COLD(ta.clock_probe);
    if (LOCK && probe && ta.lock_mu > 0.0f) {
        for (int off = 32; off > 0; off >>= 1) { n_locked += __shfl_xor(n_locked, off); n_lost += __shfl_xor(n_lost, off); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(probe + 4, (unsigned long long)n_locked); atomicAdd(probe + 5, (unsigned long long)n_lost); }
    }
    if (ABL == 5 && threadIdx.x == 0 && probe) {   // (the words of the conflict counters and, with PGSGD_TILE_TAIL=1, of the tail probe)
        ph[5] = wall_clock64() - ph_start;
        atomicAdd(probe + 4, (unsigned long long)ph[0]);
        atomicAdd(probe + 5, (unsigned long long)ph[1]);
        if (unsigned long long* tail = TILE_COLD(ta.tail_probe))
            for (int k = 0; k < 4; ++k) atomicAdd(tail + k, (unsigned long long)ph[2 + k]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && probe) {
        probe[2] = __builtin_readcyclecounter();
        probe[3] = wall_clock64();
    }
    if (ABL != 5 && threadIdx.x == 0 && TILE_COLD(ta.tail_probe)) {
        unsigned long long* tail = TILE_COLD(ta.tail_probe);
        const uint64_t now = wall_clock64();
        atomicAdd(tail, (unsigned long long)(now - wg_start));
        atomicMax(tail + 1, (unsigned long long)now);
        atomicMin(tail + 2, (unsigned long long)wg_start);
        atomicAdd(tail + 3, 1ull);
    }
    for (int off = 32; off > 0; off >>= 1) n_far += __shfl_xor(n_far, off);
    if ((threadIdx.x & 63) == 0 && n_far) atomicAdd(TILE_COLD(ta.far_count), (unsigned long long)n_far);
    if ((threadIdx.x & 63) == 0 && n_done && TILE_COLD(ta.term_count)) atomicAdd(TILE_COLD(ta.term_count), (unsigned long long)n_done);
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f) atomicMax(TILE_COLD(c.delta_max_bits), __float_as_uint(dmax));
    if (__ballot(guard) && (threadIdx.x & 63) == 0) atomicOr(TILE_COLD(c.frame_flag), 1u);
}

// The snapshot pieces of all step records rewritten in one pass — the coordinates of the two ends of the step's node (the
// end the step enters first in front) — so that a partner outside the window costs one gather, not a record gather plus a
// dependent coordinate load.  Run when the pieces do not follow from the tile kernel's own writes (sgd_tile_kernel rewrites a
// tile's when its terms are done): before a session's first tile launch, after iterations of the per-lane kernel, after the
// frame was widened or the coordinates were merged with other devices', and once per iteration of a sharded session (which
// runs only its share of the tiles).  Reads the 4-byte step handles and the (cache-resident) coordinates, writes the 64-byte
// units of snapshot pieces: 20 bytes of memory traffic per step where the round-4 layout's pass read the 16-byte records and
// wrote whole 32-byte records (48 bytes per step: 0.45 ms at config 4's 4.7e7 steps).
__global__ __launch_bounds__(256) void snapshot_kernel(const uint32_t* step_handle, const uint64_t* coords, uint64_t n_steps, uint4* recs2) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_steps; k += (uint64_t)gridDim.x * 256) {
        const uint32_t h = step_handle[k];
        const uint4 pr = *reinterpret_cast<const uint4*>(coords + (h & ~1u));  // both ends of the node, 16-byte aligned
        // (consecutive lanes, consecutive steps: every store instruction covers whole 64-byte units of snapshot pieces)
        recs2[recs2_snap_piece(k)] = (h & 1u) ? make_uint4(pr.z, pr.w, pr.x, pr.y) : pr;
    }
}

// After every tile launch: one workgroup per (bucket, part, slice) adds the messages of its SLICE of the bucket's chunks that
// fall into its PART of the bucket's node range up in LDS.  part_shift = log2 of the node ends one workgroup accumulates (at
// most 2^14 = 128 KiB of LDS); a bucket wider than that is read by several workgroups, each keeping its own part (large graphs
// only: every part unpacks every message).  slices: where buckets x parts would not fill the device (config 4: 123 buckets), a
// bucket's message stream is cut into `slices` runs of chunks, one workgroup each: every message is unpacked ONCE, by a
// workgroup that sees half (a third, a quarter) of them — round 4 ran two parts per bucket there instead, both unpacking
// everything: its instruction issue and LDS adds, not the memory, were what the drain waited for (profiles/r05/NOTES.md).
// With one slice the workgroup moves the node ends itself (nothing else writes coordinates while the drain runs: a plain
// read-modify-write); with several, each writes its sums to partial[slice][2N] — every word, zero where nothing arrived — and
// far_combine_kernel adds them to the coordinates: exact 64-bit integer adds, the sum direct atomics would give.  `partial`
// non-null with ONE slice: the drain runs on its own stream BESIDE the other colour's tile launch (which is writing coordinates:
// the drain must not) and its sums wait in partial until the same colour's next launch (pgsgd_session.hip: enqueue_drain).
__global__ __launch_bounds__(1024) void far_drain_kernel(Outbox ob, uint64_t* coords, uint64_t n_ends, uint32_t part_shift, unsigned int* frame_flag,
                                                         uint32_t slices, uint64_t* partial) {
    extern __shared__ uint64_t acc[];
    const uint32_t parts = 1u << (ob.shift - part_shift);
    // The parts of one bucket stream the SAME messages, each keeping its share.  Workgroups are dealt to the 8 XCDs round-robin
    // by index, so workgroup 8 m + x is XCD x's m-th: XCD x takes buckets x, x + 8, ... and runs a bucket's parts back to back
    // — at the same time, on CUs that share an L2: one of them fetches a line from memory, the others find it there (1e7
    // nodes, 8 parts: the drain's memory reads fall from 8x the messages towards 1x; with a bucket's parts on 8 different
    // XCDs every part fetched everything itself).  One part per bucket: the identity.  (A bucket's slices follow one another too.)
    const uint32_t xcd = blockIdx.x % kItemQueues, m = blockIdx.x / kItemQueues;
    const uint32_t slice = m % slices, mp = m / slices;
    const uint32_t b = (mp / parts) * kItemQueues + xcd, part = mp % parts, span = 1u << part_shift;
    if (b >= ob.n_buckets) return;
    for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const uint32_t handed = ob.next[b], cap = ob.cap[b];
    const uint64_t first = (uint64_t)ob.chunk0[b] * kObChunk;
    const uint64_t base = ((uint64_t)b << ob.shift) + ((uint64_t)part << part_shift);
    // One wave per chunk and pass: a chunk is 64 pairs of messages, one 16-byte load per lane, and how much of it is
    // filled is a single (wave-uniform) word.  Eight chunks per wave in flight before the LDS adds.
    const ulonglong2* pairs = reinterpret_cast<const ulonglong2*>(ob.pool) + first / 2;
    const uint32_t n_chunks = handed < cap ? handed : cap;
    const uint32_t c_lo = (uint32_t)((uint64_t)n_chunks * slice / slices), c_hi = (uint32_t)((uint64_t)n_chunks * (slice + 1) / slices);  // this slice's chunks
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint32_t part_off = part << part_shift;
    const uint32_t* fill = ob.fill + ob.chunk0[b];
    static_assert(kObChunk == 128, "one chunk = 64 lanes x 2 messages");
    // (the fill words of a pass are loaded during the pass before: a pass then waits for memory once, not twice.  Round 5
    // measured 16 and 24 chunks per wave and pass — the same, slower with spills — and a 32-bit unpack without the three
    // variable 64-bit shifts — the same.  With two slices per bucket the kernel streams its 1.33 GB in ~0.21 ms, the memory's
    // streaming rate; three or four slices of 128 KiB of LDS each no longer fit one round of workgroups and are slower.)
    constexpr int kU = 8;  // chunks per wave and pass
    uint32_t f[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
        const uint32_t c = c_lo + wave + (uint32_t)k * waves;
        f[k] = c < c_hi ? fill[c] : 0u;
    }
    for (uint32_t c0 = c_lo + wave; c0 < c_hi; c0 += kU * waves) {
        ulonglong2 m2[kU];
        bool ok[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            const uint32_t c = c0 + (uint32_t)k * waves;
            ok[k] = 2 * lane < f[k];  // (0 past the end; fill is a whole number of 8-message lines)
            if (ok[k]) m2[k] = pairs[(uint64_t)c * (kObChunk / 2) + lane];
        }
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            const uint32_t c = c0 + (uint32_t)(kU + k) * waves;
            f[k] = c < c_hi ? fill[c] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            if (!ok[k]) continue;
            uint32_t off;
            uint64_t d = outbox_unpack(ob, m2[k].x, off);
            if (d && off - part_off < span) atomicAdd(reinterpret_cast<unsigned long long*>(acc + (off - part_off)), (unsigned long long)d);
            d = outbox_unpack(ob, m2[k].y, off);
            if (d && off - part_off < span) atomicAdd(reinterpret_cast<unsigned long long*>(acc + (off - part_off)), (unsigned long long)d);
        }
    }
    __syncthreads();
    bool guard = false;
    for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) {
        if (base + i >= n_ends) break;
        uint64_t sp = 0;
        if (slice == 0) {  // (the spill words — steps too wide for a message, a pool that ran out: rare — go with the first slice)
            sp = ob.spill[base + i];
            if (sp) ob.spill[base + i] = 0;
        }
        if (partial) {
            partial[(uint64_t)slice * n_ends + base + i] = acc[i] + sp;
        } else if (acc[i] + sp != 0) {
            const uint64_t w = coords[base + i] + acc[i] + sp;
            coords[base + i] = w;
            guard |= in_frame_guard(w);
        }
    }
    if (__ballot(guard) && (threadIdx.x & 63) == 0) atomicOr(frame_flag, 1u);
}

// coords += the slices' partial sums (far_drain_kernel with several slices, or beside a launch); `partial` null: nothing to add.
// Also what a tile launch needs zeroed right before it (a session whose drains run beside the launches has no drain in front of
// the launch to do it): the chunk counters of the outbox whose sums these are, the work-item counters of the runs q_a, q_b, q_c
// and the far-pull counter the launch writes (any of them null: left alone).
__global__ __launch_bounds__(256) void far_combine_kernel(uint64_t* coords, const uint64_t* partial, uint64_t n_ends, uint32_t slices, unsigned int* frame_flag,
                                                          uint32_t* next, uint32_t n_buckets, uint32_t* q_a, uint32_t* q_b, uint32_t* q_c, unsigned long long* far_next) {
    if (blockIdx.x == 0) {   // (the drain whose sums these are is done with the chunk counters)
        if (next)
            for (uint32_t b = threadIdx.x; b < n_buckets; b += blockDim.x) next[b] = 0;
        if (threadIdx.x < kItemQueues) {
            if (q_a) q_a[threadIdx.x] = 0;
            if (q_b) q_b[threadIdx.x] = 0;
            if (q_c) q_c[threadIdx.x] = 0;
        }
        if (threadIdx.x == 0 && far_next) *far_next = 0;
    }
    if (!partial) return;
    bool guard = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ends; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t d = 0;
        for (uint32_t sl = 0; sl < slices; ++sl) d += partial[(uint64_t)sl * n_ends + i];
        if (d) {
            const uint64_t w = coords[i] + d;
            coords[i] = w;
            guard |= in_frame_guard(w);
        }
    }
    if (__ballot(guard) && (threadIdx.x & 63) == 0) atomicOr(frame_flag, 1u);
}

// the drain is done with the chunk counters; the next launch's work queues and far-pull counter start from zero
// (queues / far_next null: a drain beside a launch resets its own outbox's chunk counters only — far_combine_kernel does the rest
// right before the launch)
__global__ void outbox_reset_kernel(uint32_t* next, uint32_t n_buckets, uint32_t* queues, unsigned long long* far_next) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_buckets) next[b] = 0;
    if (queues && b < 3 * kItemQueues) queues[b] = 0;
    if (b == 0 && far_next) *far_next = 0;
}

// sampler-only replay of one tile's terms (parity hook): out[(q - first_term)*4 + {0..3}] = {ka, kb, off_a, off_b};
// one thread per lane of the tile, terms in the lane's stream order
__global__ __launch_bounds__(kTileBlock) void tile_trace_kernel(DevConst c, Tile t, uint64_t tile_index, uint32_t lanes, uint64_t term_begin,
                                                                uint64_t term_end, IterArgs a, uint64_t seed_base, uint32_t pair_uniform, uint64_t* out) {
    const uint64_t pstart = c.path_first[t.path];
    const uint64_t cnt = c.path_first[t.path + 1] - pstart;
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    Xoshiro256Plus rng;
    if (lane < lanes) rng.seed(tile_stream_seed(seed_base, a.epoch, tile_index, lane));
    uint64_t coin_x = tile_coin_seed(seed_base, a.epoch, tile_index, lane >> 6), coin_cur = 0;  // the coins of the lane's wave
    const uint64_t n_terms = term_end - term_begin;
    const uint64_t trips = (n_terms + lanes - 1) / lanes;   // the same trip count for every lane (the pairs talk through the wave)
    for (uint64_t j = 0; j < trips; ++j) {
        if (!(j & 63u)) coin_cur = Xoshiro256Plus::splitmix64(coin_x);
        const uint64_t q = term_begin + lane + j * lanes;
        const bool live = lane < lanes && q < term_end;
        const bool zipf_trip = a.cooling || ((coin_cur >> (j & 63u)) & 1u);   // wave-uniform
        uint32_t flags = 0;
        uint64_t k = 0, b_rank = 0;
        if (live) {
            k = t.t0 + below32_hi(rng, t.n, flags);
            const uint64_t s_rank = k - pstart;
            if (zipf_trip) {
                const bool back = (s_rank > 0 && ((flags >> 30) & 1u)) || s_rank == cnt - 1;
                const uint64_t room = back ? s_rank : cnt - s_rank - 1;
                const uint64_t jump = c.space < room ? c.space : room;
                const double2 zd = c.zeta_denom[zeta_index(jump, c.space_max, c.space_quant)];
                const uint64_t z = zipf_tabled(rng, c.zc, jump, zd.x, zd.y);
                b_rank = back ? s_rank - z : s_rank + z;
            } else {
                uint32_t unused;
                b_rank = below32_hi(rng, (uint32_t)cnt, unused);
            }
        }
        if (!zipf_trip && pair_uniform == 1u) {   // (every lane of the wave: an odd live lane's even neighbour is live too)
            const uint32_t lead = (uint32_t)__shfl((int)(uint32_t)(pstart + b_rank), (int)((threadIdx.x & 63u) & ~1u));
            if (live && (lane & 1u)) b_rank = tile_pair_partner(lead, (uint32_t)pstart, (uint32_t)cnt, (uint32_t)b_rank);
        } else if (!zipf_trip && pair_uniform == 2u) {   // (a live lane's quad leader is live too: the live lanes of a trip are a prefix)
            const uint32_t lead = (uint32_t)__shfl((int)(uint32_t)(pstart + b_rank), (int)((threadIdx.x & 63u) & ~3u));
            if (live) b_rank = tile_quad_partner(lead, lane & 3u, (uint32_t)pstart, (uint32_t)cnt, (uint32_t)b_rank);
        }
        if (live) {
            const uint64_t kb = pstart + b_rank;
            uint64_t* o = out + (q - term_begin) * 4;
            o[0] = k;
            o[1] = kb;
            o[2] = (c.recs[k].x ^ (flags >> 29)) & 1u;   // end offsets of the two node ends the term moves
            o[3] = (c.recs[kb].x ^ (flags >> 28)) & 1u;
        }
    }
}

}  // namespace pgsgd
