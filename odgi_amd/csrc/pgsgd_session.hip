// pgsgd_session.hip — host side of the GPU path: the session (device buffers, stream, kernel
// selection and launches, tile table, multi-GPU exchange, parity hooks) and pgsgd_layout_run.
// Device code: pgsgd_kernels.hpp (per-lane kernel, set-up and exchange kernels), pgsgd_tiles.hpp.
// Built with -ffp-contract=off (see pgsgd_kernels.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pgsgd_internal.hpp"
#include "pgsgd_kernels.hpp"
#include "pgsgd_tiles.hpp"

// ---------------------------------------------------------------------------------------------
// host side: the session
using pgsgd::set_error;

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PGSGD_E_HIP;                                                               \
        }                                                                                     \
    } while (0)

struct pgsgd_session {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    pgsgd_params params{};
    uint64_t n_nodes = 0, n_steps = 0, n_paths = 0;
    uint64_t max_path_bp = 0;
    uint64_t max_node_steps = 0;  // path steps on the most visited node
    uint32_t n_streams = 0;
    int fmt = pgsgd::kFmtQ32;
    int upd = pgsgd::kUpdAtomic;
    bool pf_lds = false;
    size_t lds_bytes = 0;
    // small lane-bound graphs (pgsgd_kernels.hpp: sample_terms_kernel / apply_terms_resident_kernel): n_streams streams
    // sample, apply_lanes lanes of one workgroup move the ends in LDS
    bool split = false;
    uint32_t apply_lanes = 0;             // the stream-count rule's number, at most one workgroup
    uint64_t split_chunk = 0;             // terms sampled and moved per pair of launches: whole rounds of the sampler streams
    size_t resident_lds = 0;
    uint4* d_terms = nullptr;             // [terms_cap] term records of one chunk of an iteration
    uint64_t terms_cap = 0;
    // device buffers
    uint4* d_recs = nullptr;
    uint64_t* d_path_first = nullptr;
    double* d_zetas = nullptr;
    double2* d_zeta_denom = nullptr;
    double2* d_zipf_tab = nullptr;  // tile kernel: {zeta_n, eta_n} per jump length (zipf_tab_kernel)
    uint64_t* d_coords = nullptr;         // [2N] coordinate words
    uint64_t* d_base = nullptr;           // coordinates at the last exchange (multi-GPU only)
    uint64_t* d_rng = nullptr;
    unsigned int* d_delta_max = nullptr;  // [2]: max |Delta| of the iteration (float bits), frame-guard flag
    unsigned int* h_delta_max = nullptr;  // pinned copy
    uint32_t frame_doublings = 0;         // times the fixed-point frame was widened (reframe)
    pgsgd::DevConst dc{};
    // region-exclusive tiles
    bool tiled = false;
    bool warm_per_lane = false;           // tiled session whose initial layout has no global structure: the
                                          // iterations before cooling run the per-lane kernel (set at upload)
    uint32_t tile_lanes = 0;              // lanes of a tile-kernel launch (n_streams stays the per-lane count)
    struct CheckPair { uint32_t end_a, end_b; float d; };
    std::vector<CheckPair> check_pairs;   // long-range step pairs for the initial-layout check
    uint32_t region = 256, tile_steps = 224, tile_block = pgsgd::kTileBlock, tile_substeps = 1;
    uint32_t shard_rank = 0, shard_world = 1;    // multi-GPU by node region: work items rank, rank+world, ...
    uint32_t tshard_rank = 0, tshard_world = 1;  // multi-GPU by tile: tiles rank, rank+world, ... of every work item
    bool exact_colours = false;           // region shard with the exact exchange: iteration_part(c, 2) runs colour c alone, the ranks
                                          // exchange integer deltas after it (pgsgd_session_exchange_exact_*)
    int last_colour = -1;                 // colour of the last tile launch
    uint64_t tile_epoch = 0;              // iterations started (tile kernel: part of every term's seed)
    uint64_t relax_iter = 0;              // iterations of the current layout, i.e. since the last upload (index of the far pulls' gentle start)
    uint64_t tile_seed_base = 0;          // seed + stream_offset; a sharded session: seed alone (pgsgd_session_set_shard)
    unsigned long long* d_clock = nullptr;  // [6] TileArgs::clock_probe of the last windowed tile launch, [4] its tail_probe
    bool tile_tail = false;                 // debug knob PGSGD_TILE_TAIL: the launches record their workgroups' lifetimes
    unsigned long long* d_far = nullptr;  // [2 colours][2]: far-partner updates of the last two launches of each colour
    uint32_t far_launches[2] = {0, 0};    // tile launches so far, per colour (parity selects the counter a launch writes)
    uint4* d_recs2 = nullptr;             // the tile kernel's gather records: static + snapshot pieces, four steps per 128-byte group (pgsgd_kernels.hpp)
    uint32_t* d_step_handle = nullptr;    // [S] step handles, kept by a tiled session for snapshot_kernel
    std::vector<uint64_t> ob_bucket_steps;  // path steps on the nodes of each outbox bucket (sizes the pool shares)
    std::vector<uint32_t> node_steps;       // path steps on every node (automatic stream count only)
    uint32_t* d_node_steps = nullptr;       // uploaded when the hot-node learning-rate cap is active
    pgsgd::Outbox ob{};                   // far-update outbox (device pointers)
    uint32_t* d_ob_chunk0 = nullptr;
    uint32_t* d_ob_cap = nullptr;
    uint64_t ob_budget_terms = 0;         // terms per call the pool was sized for
    double ob_msgs_per_term = 0.5;        // bound on the messages of one launch per term of the call (set from the tile table)
    unsigned long long* d_ob_overflow = nullptr;  // messages that found their bucket's pool share used up (sent as atomics)
    uint64_t ob_total_chunks = 0;
    double aux_ms[2] = {0, 0};            // snapshot_kernel, far_drain_kernel (HIP events)
    bool ob_pending = false;              // the last tile launch's far pulls wait in the outbox (drained before the next launch)
    bool snap_stale = true;               // the snapshot pieces of the gather records do not follow from the tile kernel's own writes
    bool tile_forced = false;             // PGSGD_TILE_FORCE (parity knob) was set when the session was created
    bool snapshot_pass = false;           // PGSGD_TILE_SNAPSHOT_PASS (experiment knob): a pass over all records per iteration, as a sharded session takes
    uint64_t* d_term0 = nullptr;          // [n_tiles + 1] first term of every tile for term0_terms terms per call
    uint64_t term0_terms = 0;
    uint32_t ob_part_shift = 13;          // log2 of the node ends one drain workgroup accumulates in LDS
    uint64_t tile_until = 0;              // experiment knob PGSGD_TILE_UNTIL
    uint32_t ob_slices = 1;               // workgroups that share a (bucket, part)'s message stream (far_drain_kernel); > 1: d_ob_partial
    uint64_t* d_ob_partial = nullptr;     // [ob_slices][2N] the slices' sums, added to the coordinates by far_combine_kernel
    unsigned long long* d_ob_spill = nullptr;
    // The drain BESIDE the next launch.  A session of a schedule as long as the reference's default (iter_max >= 30),
    // unsharded, keeps one outbox per region colour: colour c's launch writes outbox c; right after it far_drain_kernel sums
    // outbox c into d_pend[c] on the DRAIN stream — beside the next launch, which is the other colour's and writes the other
    // outbox — and far_combine_kernel adds d_pend[c] to the coordinates right before colour c's NEXT launch.  A launch's far pulls
    // then arrive one launch later than with the drain in front of the very next launch: measured free on the default schedule,
    // costly on short ones (profiles/r06/NOTES.md section 3), hence the gate; PGSGD_FLAG_SYNC_DRAIN turns it off.
    bool async_drain = false;             // the rule says so for this session's schedule (create; set_shard turns it off)
    pgsgd::Outbox ob1{};                  // colour 1's outbox (colour 0's is `ob`): own pool, fill, next, spill
    uint64_t* d_pend[2] = {nullptr, nullptr};   // [ob_slices][2N] the sums of colour c's last drain
    bool pend_waiting[2] = {false, false};      // ... not in the coordinates yet
    // While the layout's global structure is still forming — the five iterations of the far pulls' ramp (kFarGentleIterations) — a
    // launch's pulls reach the coordinates right before the very next launch, whatever its colour: the launch stream waits for their
    // drain, nothing runs beside it.  Measured at config 4 (exact figure of the final layout / stress after iterations 10, 12, 14 /
    // terms per s; profiles/r06/drain_beside_from_1e6.jsonl): never late 0.20548 / 91 86 90 / 7.09e10; late from iteration 15 (the
    // cooling launches) 0.20528 / 91 86 90 / 7.13e10; from 10: 0.20531 / 91 165 169 / 7.21e10; **from 5: 0.20545 / 358 185 171 / 7.32e10**;
    // from 1: 0.20583 / 3037 545 196; every launch: the transient's damage reaches the final layout of graphs with window-less tiles
    // (2.2 instead of 0.2).  The reference's own stress at those points: 12 740.
    bool pulls_urgent[2] = {false, false};
    uint64_t async_from = 0;                    // experiment knob PGSGD_ASYNC_FROM
    bool queues_dirty = false;                  // a tile launch has run since the work queues and far-pull counters were last zeroed
    int pend_order = 0;                         // the colour whose sums have waited longer (a flush delivers it first)
    hipStream_t drain_stream = nullptr;
    hipEvent_t ev_launch[2] = {nullptr, nullptr}, ev_drain[2] = {nullptr, nullptr};
    struct DrainEv { hipEvent_t a, b; };
    std::vector<DrainEv> free_drain_events, pending_drain_events;   // far_drain_kernel's duration on the drain stream, collected when complete
    double drain_beside_ms = 0;           // ... summed (HIP events on the drain stream): NOT on the launch stream's critical path
    std::vector<pgsgd::Tile> h_tiles;     // host copy of the tile table (parity hooks)
    std::vector<pgsgd::WorkItem> h_items; // host copy of the work items, colour 0 first (parity hooks)
    pgsgd::Tile* d_tiles = nullptr;
    uint4* d_tile_heads = nullptr;        // TileArgs::tile_heads
    pgsgd::WorkItem* d_items = nullptr;   // colour 0 items, then colour 1 items
    uint32_t n_items[2] = {0, 0};
    // the session's own windows with their tiles cut into tile_split consecutive parts (WorkItem::local: kItemHasNext,
    // kItemDepShift): what it launches when tile_split > 1 (build_launch_items)
    uint32_t tile_split = 1;
    uint32_t tile_split_knob = 0;         // debug knob PGSGD_TILE_SPLIT (0: the rule)
    bool items_one_run = true;            // the work items of a colour are one run by decreasing size (not PGSGD_TILE_ORDER=region)
    std::vector<pgsgd::WorkItem> h_items_split;
    pgsgd::WorkItem* d_items_split = nullptr;
    uint32_t n_items_split[2] = {0, 0};
    uint32_t* d_item_done = nullptr;      // [max items of a launch] TileArgs::item_done
    uint32_t launch_stamp = 0;            // TileArgs::stamp of the last tile launch
    uint32_t item_chunk[2][pgsgd::kItemQueues + 1] = {};  // per colour: runs of the windowed items, one per XCD
    uint32_t item_chunk_split[2][pgsgd::kItemQueues + 1] = {};  // the same for the list in parts (build_launch_items)
    uint32_t n_windowless = 0;            // the last n_windowless items of colour 0 have no window (their own launch)
    uint32_t* d_queue = nullptr;          // [3][kItemQueues] work-item counters: colour 0, colour 1, colour 0's window-less items
    uint64_t tile_steps_total = 0;
    uint64_t n_tiles = 0, n_nonlocal_tiles = 0;
    size_t tile_lds = 0;                  // LDS of a workgroup of the session's warm windowed instance (sgd_tile_kernel<.., PUSH = tile_push>)
    size_t tile_lds1 = 0;                 // ... of every other instance (PUSH = 1: cooling launches, window-less tiles, the lock instance)
    int tile_far = 0;  // pgsgd::kFarTwoSided / kFarExclusive
    uint32_t tile_pair_uniform = 1;   // TileArgs::pair_uniform (0: PGSGD_FLAG_NO_PARTNER_PAIRS)
    float tile_lock_mu = 0.0f;        // TileArgs::lock_mu (debug knob PGSGD_TILE_LOCK_MU)
    uint32_t tile_snap_every = 1;     // debug knob PGSGD_TILE_SNAP_EVERY
    uint32_t tile_rotate = 0;         // debug knob PGSGD_TILE_ROTATE
    uint32_t tile_lane_coin = 0;      // debug knob PGSGD_TILE_LANE_COIN
    float tile_far_relax_max = 0.0f, tile_far_relax_slope = 0.0f;   // debug knobs PGSGD_TILE_FAR_RELAX_MAX / _SLOPE: min(max, slope * iteration) after the two gentle iterations
    float tile_far_relax_override = 0.0f;  // debug knob PGSGD_TILE_FAR_RELAX: a constant under-relaxation of the far pulls instead of tile_far_relax()
    uint32_t tile_push = 1;           // messages a lane hands to the rings per call of their protocol in a WARM launch (sgd_tile_kernel<.., PUSH>): 2 where the longer queues cost no workgroup per CU
    uint32_t tile_wq_threshold = 64 * pgsgd::kWqPush;  // TileArgs::wq_threshold (debug knob PGSGD_TILE_WQ: 1 = every message goes to the rings at once; 64: one message per lane and call, as until round 6's third session)
    int tile_math = 1; // pgsgd::kMathFast / kMathExact (PGSGD_FLAG_EXACT_MATH, or a path of 2^32 bp or more)
    uint32_t tile_grid = 0;
    // kernel timing: e[0..1] bracket the update kernel; a tile launch also has e[2] (before the drain of the launch
    // before it) and e[3] (after that drain, before its snapshot kernel)
    struct EvSet { hipEvent_t e[4]; int n; };
    std::vector<EvSet> free_events, pending_events;
    double kernel_ms = 0;
    uint64_t launches = 0;
    uint64_t n_kernels = 0, n_copies = 0;  // everything the iteration calls put on the stream (pgsgd_session_launch_counts)
};

static int pick_device(int requested, int* out) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); the layout kernels run on MI355X only, there is no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return PGSGD_E_NODEVICE;
    }
    int dev = requested;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= count) { set_error("device %d requested but only %d present", dev, count); return PGSGD_E_NODEVICE; }
    *out = dev;
    return PGSGD_OK;
}

static int collect_drain_events(pgsgd_session* s);
static int collect_events(pgsgd_session* s) {
    for (auto& ev : s->pending_events) {
        float ms = 0;
        if (ev.n == 1) {  // a drain on its own (pgsgd_session_flush): e[2] .. drain .. e[3]
            HIP_TRY(hipEventElapsedTime(&ms, ev.e[2], ev.e[3]));
            s->aux_ms[1] += ms;
            s->free_events.push_back(ev);
            continue;
        }
        HIP_TRY(hipEventElapsedTime(&ms, ev.e[0], ev.e[1]));
        s->kernel_ms += ms;
        s->launches++;
        if (ev.n == 4) {  // e[2] .. drain of the launch before .. e[3] .. snapshot .. e[0] .. tile kernel .. e[1]
            HIP_TRY(hipEventElapsedTime(&ms, ev.e[3], ev.e[0]));
            s->aux_ms[0] += ms;
            HIP_TRY(hipEventElapsedTime(&ms, ev.e[2], ev.e[3]));
            s->aux_ms[1] += ms;
        }
        s->free_events.push_back(ev);
    }
    s->pending_events.clear();
    return collect_drain_events(s);
}

// drains beside a launch run on their own stream and may still be running when the launch stream is idle: take the finished ones
static int collect_drain_events(pgsgd_session* s) {
    size_t kept = 0;
    for (auto& de : s->pending_drain_events) {
        if (hipEventQuery(de.b) == hipSuccess) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, de.a, de.b));
            s->drain_beside_ms += ms;
            s->free_drain_events.push_back(de);
        } else {
            (void)hipGetLastError();  // (hipErrorNotReady is not an error)
            s->pending_drain_events[kept++] = de;
        }
    }
    s->pending_drain_events.resize(kept);
    return PGSGD_OK;
}

static int take_events(pgsgd_session* s, pgsgd_session::EvSet* out) {
    if (!s->free_events.empty()) {
        *out = s->free_events.back();
        s->free_events.pop_back();
        return PGSGD_OK;
    }
    for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreate(&out->e[i]));
    return PGSGD_OK;
}

// PGSGD_FLAG_HOT_NODE_CAP (experiment, off by default).  Can write conflicts on busy nodes be answered by a capped
// learning rate instead of by fewer lanes (auto_streams)?  Measured, three seeds each, against the lane rule and the
// CPU restatement (profiles/r02/hotcap_per_lane_*.jsonl):
//  * every lane the GPU holds, every term capped at 1/h (h = L * steps on the node / S concurrent terms per node end):
//    no — on the fixture graphs h is 30-80 for EVERY node, an iteration then amounts to a few projections per node end
//    instead of hundreds, and the layouts end at stress 3.6 ... 5e4 (against 0.3 ... 0.9);
//  * lanes bounded by the BULK of the nodes (L = 2 S / s*, nodes with more than s* steps carry a tenth of all steps),
//    only terms on busier nodes capped — this function: a hub no longer idles the GPU (DRB1-3123_unsorted, one node
//    with 268 of 21 882 steps: 3584 lanes instead of 128, 12x faster) but its layout is 30 % worse (0.44 vs 0.33), and
//    where the lane count stays the same the cap alone costs 5-25 % stress (DRB1-3123 0.84 vs 0.69, chr6.C4 0.57 vs 0.53).
// The reference's rule wants full projections on every pair; fewer lanes keep that, smaller steps do not.
static uint32_t capped_streams(const pgsgd_session* s, int cus, int blocks_per_cu, uint64_t terms_per_launch) {
    const uint64_t full = (uint64_t)cus * (uint64_t)blocks_per_cu * pgsgd::kBlock;
    uint64_t n = std::min<uint64_t>(full, std::max<uint64_t>(terms_per_launch / 8, 256));  // at least eight terms per lane and launch
    const uint64_t by_busiest = 2 * s->n_steps / std::max<uint64_t>(1, s->max_node_steps);
    if (by_busiest < n && !s->node_steps.empty()) {
        std::vector<uint32_t> v(s->node_steps);
        std::sort(v.begin(), v.end(), std::greater<uint32_t>());
        uint64_t acc = 0, s_star = v.front();
        for (uint32_t c : v) {  // busiest first: stop when a tenth of the steps is covered
            s_star = c;
            acc += c;
            if (10 * acc >= s->n_steps) break;
        }
        n = std::min<uint64_t>(n, std::max<uint64_t>(by_busiest, 2 * s->n_steps / std::max<uint64_t>(1, s_star)));
    }
    n = std::max<uint64_t>(64, (n / 64) * 64);
    if (n >= pgsgd::kBlock) n = (n / pgsgd::kBlock) * pgsgd::kBlock;
    return (uint32_t)n;
}

// two-pass iterations of small lane-bound graphs: taken, with an automatic stream count, when the coordinates fit one
// compute unit's LDS and the stream-count rule allows at most kSplitMaxLanes lanes
constexpr uint64_t kSplitMaxLanes = 2048;
constexpr uint32_t kSplitApplyLanes = 1024;  // lanes of the moving workgroup, at most
constexpr uint64_t kSplitChunkTerms = 8ull << 20;  // term records of at most this many terms (128 MB) are held at once

static uint32_t auto_streams(const pgsgd_session* s, int cus, int blocks_per_cu) {
    // Full residency of the update kernel, unless the graph cannot take that many concurrent terms.
    // Concurrent displacements of one node end are all computed from the same (stale) position and
    // then added together; with mu = 1 each of them is a full projection, so about four or more in
    // flight on one end overshoot and the layout oscillates apart.  What matters is the HOTTEST
    // node: a term touches a node in proportion to the path steps on it, so with L terms in flight
    // the busiest node sees L * max_node_steps / S of them.  Measured on MI355X
    // (profiles/r01/sweep_*.jsonl, tools/gpu_replicates.py): layouts keep oracle quality up to 4-16
    // in flight on the busiest node and diverge beyond; the cap is 2.  (Hogwild stores never
    // diverge but lose quality at about the same point, so they share the rule.)
    const uint64_t full = (uint64_t)cus * (uint64_t)blocks_per_cu * pgsgd::kBlock;
    uint64_t in_flight = 2;
    if (const char* e = pgsgd::debug_env("PGSGD_LANE_CAP")) in_flight = (uint64_t)std::min(64, std::max(1, atoi(e)));  // experiment knob (tools/gpu_lane_cap.py)
    const uint64_t cap = in_flight * s->n_steps / std::max<uint64_t>(1, s->max_node_steps);
    uint64_t n = std::min(full, std::max<uint64_t>(cap, 64));
    n = std::max<uint64_t>(64, (n / 64) * 64);
    if (n >= pgsgd::kBlock) n = (n / pgsgd::kBlock) * pgsgd::kBlock;
    return (uint32_t)n;
}

// every (PF_LDS, COORD_LOAD, FMT, UPD, GROUPED, ABL) instance the host can launch
typedef void (*iter_kernel_t)(pgsgd::DevConst, pgsgd::IterArgs);
template <int FMT, int UPD, bool GROUPED>
static iter_kernel_t select_kernel_fug(bool pf_lds, bool plain, uint32_t abl) {
    using namespace pgsgd;
    if (abl) {
        if (!pf_lds) return nullptr;
        return abl == 1 ? sgd_iteration_kernel<true, 1, FMT, UPD, GROUPED, 1> : abl == 3 ? sgd_iteration_kernel<true, 1, FMT, UPD, GROUPED, 3>
                                                                                         : sgd_iteration_kernel<true, 1, FMT, UPD, GROUPED, 4>;
    }
    if (pf_lds) return plain ? sgd_iteration_kernel<true, 0, FMT, UPD, GROUPED, 0> : sgd_iteration_kernel<true, 1, FMT, UPD, GROUPED, 0>;
    return plain ? sgd_iteration_kernel<false, 0, FMT, UPD, GROUPED, 0> : sgd_iteration_kernel<false, 1, FMT, UPD, GROUPED, 0>;
}
template <int FMT, int UPD>
static iter_kernel_t select_kernel_fu(bool pf_lds, bool plain, bool grouped, uint32_t abl) {
    return grouped ? select_kernel_fug<FMT, UPD, true>(pf_lds, plain, abl) : select_kernel_fug<FMT, UPD, false>(pf_lds, plain, abl);
}
static iter_kernel_t select_kernel(bool pf_lds, bool plain, int fmt, int upd, bool grouped, uint32_t abl) {
    using namespace pgsgd;
    if (fmt == kFmtQ32) return upd == kUpdStore ? select_kernel_fu<kFmtQ32, kUpdStore>(pf_lds, plain, grouped, abl) : select_kernel_fu<kFmtQ32, kUpdAtomic>(pf_lds, plain, grouped, abl);
    return upd == kUpdStore ? select_kernel_fu<kFmtF32, kUpdStore>(pf_lds, plain, grouped, abl) : select_kernel_fu<kFmtF32, kUpdAtomic>(pf_lds, plain, grouped, abl);
}

// the software-pipelined instance of the default configuration (pgsgd_kernels.hpp: sgd_iteration_kernel_piped)
static iter_kernel_t select_piped(bool pf_lds, bool plain, int upd) {
    using namespace pgsgd;
    if (upd == kUpdStore) {
        if (pf_lds) return plain ? sgd_iteration_kernel_piped<true, 0, kUpdStore> : sgd_iteration_kernel_piped<true, 1, kUpdStore>;
        return plain ? sgd_iteration_kernel_piped<false, 0, kUpdStore> : sgd_iteration_kernel_piped<false, 1, kUpdStore>;
    }
    if (pf_lds) return plain ? sgd_iteration_kernel_piped<true, 0, kUpdAtomic> : sgd_iteration_kernel_piped<true, 1, kUpdAtomic>;
    return plain ? sgd_iteration_kernel_piped<false, 0, kUpdAtomic> : sgd_iteration_kernel_piped<false, 1, kUpdAtomic>;
}
// which per-lane kernel a session launches: the pipelined one for fixed-point coordinates, one term per first step,
// fewer than 2^32 path steps and no hot-node cap; the general one otherwise
static iter_kernel_t session_kernel(const pgsgd_session* s, bool plain, uint32_t abl) {
    const bool grouped = s->params.terms_per_anchor > 1;
    if (s->fmt == pgsgd::kFmtQ32 && !grouped && !abl && s->n_steps < 0xffffffffull && !(s->params.flags & (PGSGD_FLAG_HOT_NODE_CAP | PGSGD_FLAG_NO_PIPELINE)))
        return select_piped(s->pf_lds, plain, s->upd);
    return select_kernel(s->pf_lds, plain, s->fmt, s->upd, grouped, abl);
}

// Host-side copies of two rules of the tile kernel's sampler, for the CPU suite (the kernel, the trace kernel and the
// oracle's mirror are compared on the GPU): the Zipf/uniform coin a wave's lanes share in a trip of a warm iteration, and
// the partner an odd lane takes from its even neighbour's in a uniform trip.
extern "C" int pgsgd_tile_wave_coin(uint64_t seed_base, uint64_t epoch, uint64_t tile, uint32_t wave, uint64_t trip) {
    uint64_t x = pgsgd::tile_coin_seed(seed_base, epoch, tile, wave), w = 0;
    for (uint64_t k = 0; k <= trip / 64; ++k) w = pgsgd::Xoshiro256Plus::splitmix64(x);
    return (int)((w >> (trip % 64)) & 1u);
}
extern "C" uint32_t pgsgd_tile_pair_partner(uint32_t lead_flat_step, uint32_t path_first_step, uint32_t path_steps, uint32_t own_rank) {
    return pgsgd::tile_pair_partner(lead_flat_step, path_first_step, path_steps, own_rank);
}
extern "C" uint32_t pgsgd_tile_quad_partner(uint32_t lead_flat_step, uint32_t lane_in_quad, uint32_t path_first_step, uint32_t path_steps, uint32_t own_rank) {
    return pgsgd::tile_quad_partner(lead_flat_step, lane_in_quad & 3u, path_first_step, path_steps, own_rank);
}
// Host side of the tiled kernel: cut paths into tiles, bind tiles to region windows, order the work.
struct HostTiles {
    std::vector<pgsgd::Tile> tiles;
    std::vector<pgsgd::WorkItem> items[2];
    uint32_t chunk[2][pgsgd::kItemQueues + 1] = {};  // per colour: the windowed items' runs, one per XCD (TileArgs::chunk)
    uint64_t steps_total = 0, n_nonlocal = 0;
};

typedef void (*tile_kernel_t)(pgsgd::DevConst, pgsgd::TileArgs, pgsgd::TileSampler, pgsgd::IterArgs);
template <int FAR, int MATH>
static tile_kernel_t tile_kernel_f(bool cooling, bool local, bool push2) {
    using namespace pgsgd;
#ifndef PGSGD_TILE_ABL
#define PGSGD_TILE_ABL 0   // experiment builds (make libpgsgd_x<N>.so): a profiling instance of the windowed tile kernel, results invalid
#endif
    if (local) return cooling ? sgd_tile_kernel<1, FAR, true, true, MATH, false, PGSGD_TILE_ABL> : push2 ? sgd_tile_kernel<1, FAR, false, true, MATH, false, PGSGD_TILE_ABL, 2> : sgd_tile_kernel<1, FAR, false, true, MATH, false, PGSGD_TILE_ABL>;
    return cooling ? sgd_tile_kernel<1, FAR, true, false, MATH> : sgd_tile_kernel<1, FAR, false, false, MATH>;
}
// math: pgsgd::kMathFast (what sessions run) or kMathExact (PGSGD_FLAG_EXACT_MATH, paths of 2^32 bp and more); lock: the
// instance with conflict resolution on the window ends (PGSGD_FLAG_LOCK_WINDOW_ENDS: two-sided far rule, fast math, windowed items)
// push2: the warm windowed instance that hands the rings two messages per lane and call (pgsgd_session::tile_push)
static tile_kernel_t tile_kernel(int far, int math, bool cooling = false, bool local = true, bool lock = false, bool push2 = false) {
    using namespace pgsgd;
    if (lock && local && far == kFarTwoSided && math == kMathFast)
        return cooling ? sgd_tile_kernel<1, kFarTwoSided, true, true, kMathFast, true> : sgd_tile_kernel<1, kFarTwoSided, false, true, kMathFast, true>;
    if (math == pgsgd::kMathExact)
        return far == pgsgd::kFarExclusive ? tile_kernel_f<pgsgd::kFarExclusive, pgsgd::kMathExact>(cooling, local, push2) : tile_kernel_f<pgsgd::kFarTwoSided, pgsgd::kMathExact>(cooling, local, push2);
    return far == pgsgd::kFarExclusive ? tile_kernel_f<pgsgd::kFarExclusive, pgsgd::kMathFast>(cooling, local, push2) : tile_kernel_f<pgsgd::kFarTwoSided, pgsgd::kMathFast>(cooling, local, push2);
}

struct RawTile { uint64_t t0; uint32_t n, path, rmin, rmax, maxmult; };

// paths cut into tiles of T consecutive steps (host: one entry per tile, no pass over the steps)
static std::vector<RawTile> cut_tiles(const pgsgd_graph_view* g, uint32_t T) {
    std::vector<RawTile> raw;
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        const uint64_t b = g->path_first[p], cnt = g->path_first[p + 1] - b;
        if (cnt <= 1) continue;  // single-step paths are never sampled (path_sgd_layout.cpp:189-192)
        for (uint64_t o = 0; o < cnt; o += T) {
            RawTile r;
            r.t0 = b + o;
            r.n = (uint32_t)std::min<uint64_t>(T, cnt - o);
            r.path = (uint32_t)p;
            r.rmin = UINT32_MAX;
            r.rmax = 0;
            r.maxmult = 1;
            raw.push_back(r);
        }
    }
    return raw;
}

// rank range and the most visits of one node, per tile: on the device, from the uploaded step handles
static int device_tile_stats(hipStream_t stream, const uint32_t* d_handle, uint32_t T, std::vector<RawTile>& raw) {
    const uint64_t n = raw.size();
    if (!n) return PGSGD_OK;
    std::vector<uint64_t> t0(n);
    std::vector<uint32_t> tn(n), out(3 * n);
    for (uint64_t i = 0; i < n; ++i) { t0[i] = raw[i].t0; tn[i] = raw[i].n; }
    uint64_t* d_t0 = nullptr;
    uint32_t *d_n = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(&d_t0, n * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&d_n, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_out, 3 * n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_t0, t0.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_n, tn.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32);
        hipLaunchKernelGGL(pgsgd::tile_stats_kernel, dim3(grid), dim3(256), 4 * (size_t)T * sizeof(uint32_t), stream, d_handle, d_t0, d_n, n, T, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out, 3 * n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_t0);
    (void)hipFree(d_n);
    (void)hipFree(d_out);
    if (e != hipSuccess) { set_error("tile statistics on the device: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP; }
    for (uint64_t i = 0; i < n; ++i) { raw[i].rmin = out[3 * i]; raw[i].rmax = out[3 * i + 1]; raw[i].maxmult = out[3 * i + 2]; }
    return PGSGD_OK;
}

// bind tiles to region windows and order the work (host: linear in the number of tiles)
static HostTiles group_tiles(const std::vector<RawTile>& raw, uint64_t n_nodes, uint32_t R, bool by_size) {
    // local tiles grouped by (colour, region of rmin); a tile that does not fit two regions is its own global item
    struct Group { uint32_t r0 = 0; uint64_t steps = 0; std::vector<uint32_t> members; };
    std::vector<Group> groups[2];
    std::vector<uint32_t> nonlocal;
    {   // one bucket per region, tiles in input order inside it (a stable counting sort by region)
        const uint64_t n_regions = (n_nodes + R - 1) / R;
        std::vector<Group> by_region(n_regions);
        for (uint32_t i = 0; i < raw.size(); ++i) {
            const uint32_t r0 = raw[i].rmin / R;
            if ((uint64_t)raw[i].rmax < ((uint64_t)r0 + 2) * R) {
                Group& gr = by_region[r0];
                gr.members.push_back(i);
                gr.steps += raw[i].n;
            } else {
                nonlocal.push_back(i);
            }
        }
        for (uint64_t r0 = 0; r0 < n_regions; ++r0)
            if (!by_region[r0].members.empty()) {
                by_region[r0].r0 = (uint32_t)r0;
                groups[r0 & 1u].push_back(std::move(by_region[r0]));
            }
    }
    HostTiles ht;
    ht.n_nonlocal = nonlocal.size();
    auto emit_tile = [&](uint32_t i) {
        pgsgd::Tile t;
        t.t0 = raw[i].t0;
        t.cum = ht.steps_total;
        t.n = raw[i].n;
        t.path = raw[i].path;
        t.lanes = std::max<uint32_t>(1, 2 * raw[i].n / raw[i].maxmult);
        t.pad = 0;
        ht.steps_total += raw[i].n;
        ht.tiles.push_back(t);
    };
    for (int colour = 0; colour < 2; ++colour) {
        // by_size (the default): big work items first, one run that every workgroup pulls from.
        // !by_size (experiment, PGSGD_TILE_ORDER=region): items in node order (as the regions were visited above), cut
        // into one run per XCD of about equal step count, so that the workgroups of an XCD work on neighbouring windows at
        // the same time and share the partner records just outside their tiles in their L2 (TileArgs::chunk).  Measured
        // at config 4 (profiles/r03/bench_variants_call2.txt, sq_tcc_xcd_vs_size.json): L2 misses of the tile kernel fall
        // 8 % (1.51e8 -> 1.39e8 per launch) and the kernel gets SLOWER, 9.09 / 7.48 ms against 8.75 / 7.37 per warm /
        // cooling iteration — neighbours in lock step also fill the same outbox buckets at the same time.
        if (by_size)
            std::sort(groups[colour].begin(), groups[colour].end(),
                      [](const Group& a, const Group& b) { return a.steps != b.steps ? a.steps > b.steps : a.r0 < b.r0; });  // a total order
        {
            uint64_t total = 0, acc = 0;
            for (const Group& gr : groups[colour]) total += gr.steps;
            uint32_t q = 0, i = 0;
            ht.chunk[colour][0] = 0;
            for (const Group& gr : groups[colour]) {
                // item i starts run q + 1 when the runs before it hold their share of the steps
                while (!by_size && q + 1 < pgsgd::kItemQueues && acc * pgsgd::kItemQueues >= (uint64_t)(q + 1) * total) ht.chunk[colour][++q] = i;
                acc += gr.steps;
                ++i;
            }
            while (q < pgsgd::kItemQueues) ht.chunk[colour][++q] = i;
        }
        for (const Group& gr : groups[colour]) {
            pgsgd::WorkItem wi;
            wi.tile_begin = (uint32_t)ht.tiles.size();
            for (uint32_t i : gr.members) emit_tile(i);
            wi.tile_end = (uint32_t)ht.tiles.size();
            wi.win0 = gr.r0 * R;
            wi.local = 1;
            ht.items[colour].push_back(wi);
        }
        if (colour == 0)
            for (uint32_t i : nonlocal) {
                pgsgd::WorkItem wi;
                wi.tile_begin = (uint32_t)ht.tiles.size();
                emit_tile(i);
                wi.tile_end = (uint32_t)ht.tiles.size();
                wi.win0 = 0;
                wi.local = 0;
                ht.items[0].push_back(wi);
            }
    }
    return ht;
}

// The rule of build_launch_items as host arithmetic (tests): parts a window's tiles are cut into when a launch has
// `windows` windows (the fewer of the two colours') of `tiles_per_window` tiles on average for `resident_workgroups` slots.
extern "C" uint32_t pgsgd_tile_parts_for(uint64_t windows, uint64_t tiles_per_window, uint64_t resident_workgroups) {
    if (!windows || !resident_workgroups || windows < resident_workgroups) return 1;  // the device is not filled once: nothing to balance
    const uint64_t for_rounds = (24 * resident_workgroups + windows - 1) / windows;   // 24 rounds of work items per launch ...
    const uint64_t by_length = tiles_per_window / 4;                                  // ... of no fewer than four tiles each ...
    return (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, std::min(for_rounds, by_length)));  // ... and at most 16 per window
}

// One colour's work items with every window's tiles cut into k consecutive parts (see WorkItem): part 0 of every window in the
// given order, then part 1 of every window, ...; a part waits for the part before it of the same window.  A window with
// fewer than k tiles has fewer parts.  The window-less items (the last n_windowless) follow unchanged.
static std::vector<pgsgd::WorkItem> split_items(const std::vector<pgsgd::WorkItem>& items, uint32_t n_windowless, uint32_t k) {
    const uint32_t n_local = (uint32_t)items.size() - n_windowless;
    std::vector<pgsgd::WorkItem> out;
    std::vector<uint32_t> prev(n_local, 0xffffffffu);  // index in `out` of the window's last part so far
    for (uint32_t j = 0; j < k; ++j)
        for (uint32_t w = 0; w < n_local; ++w) {
            const pgsgd::WorkItem& wi = items[w];
            const uint32_t len = wi.tile_end - wi.tile_begin, parts = std::min(k, std::max<uint32_t>(1, len));
            if (j >= parts) continue;
            pgsgd::WorkItem part = wi;
            part.tile_begin = wi.tile_begin + (uint32_t)((uint64_t)len * j / parts);
            part.tile_end = wi.tile_begin + (uint32_t)((uint64_t)len * (j + 1) / parts);
            part.local = pgsgd::kItemLocal;
            if (prev[w] != 0xffffffffu) {
                part.local |= (prev[w] + 1u) << pgsgd::kItemDepShift;
                out[prev[w]].local |= pgsgd::kItemHasNext;
            }
            prev[w] = (uint32_t)out.size();
            out.push_back(part);
        }
    out.insert(out.end(), items.begin() + n_local, items.end());
    return out;
}

// split_items on caller's arrays (tests without a device: the item list a session of these windows would launch).
// flags: WorkItem::local — bit 0 window, bit 1 another item waits for this one, bits 31..2: 1 + the item it waits for.
extern "C" int64_t pgsgd_tile_split_items(const uint32_t* tile_begin, const uint32_t* tile_end, const uint32_t* win0, uint64_t n_items, uint32_t n_windowless,
                                          uint32_t k, uint32_t* out_begin, uint32_t* out_end, uint32_t* out_win0, uint32_t* out_flags, uint64_t capacity) {
    if (!tile_begin || !tile_end || !win0 || n_windowless > n_items || k == 0) return PGSGD_E_INVALID;
    std::vector<pgsgd::WorkItem> items(n_items);
    for (uint64_t i = 0; i < n_items; ++i) {
        items[i].tile_begin = tile_begin[i];
        items[i].tile_end = tile_end[i];
        items[i].win0 = win0[i];
        items[i].local = i + n_windowless < n_items ? pgsgd::kItemLocal : 0u;
    }
    const std::vector<pgsgd::WorkItem> cut = split_items(items, n_windowless, k);
    for (uint64_t i = 0; i < cut.size() && i < capacity; ++i) {
        if (out_begin) out_begin[i] = cut[i].tile_begin;
        if (out_end) out_end[i] = cut[i].tile_end;
        if (out_win0) out_win0[i] = cut[i].win0;
        if (out_flags) out_flags[i] = cut[i].local;
    }
    return (int64_t)cut.size();
}

// The item lists a session launches.  A launch of few rounds of work items loses 7-10 % of its workgroup-time to its tail
// (WorkItem in pgsgd_tiles.hpp), so when the session's windows fill the device at least once but fewer than 24 times, every
// window's tiles are cut into k parts: k = what brings a launch to 24 rounds, parts no shorter than four tiles, at most 16
// (measured at config 4, 2 017 windows of 53 tiles on 1 024 workgroups: k = 1 / 2 / 4 / 8 / 12 / 16 / 24 / 32 / 64 ->
// 0.557 / 0.563 / 0.567 / 0.588 / 0.596 / 0.593 / 0.581 / 0.564 / 0.487 of the roofline, workgroups alive 0.91 -> 0.98 of
// the launch at k = 16; profiles/r04/NOTES.md).  A session that owns every shard_world-th window (region shard) cuts its
// own windows; one that runs every tshard_world-th tile of every window counts its share of a part's tiles.  Called when
// the session is created and when its shard changes.
static int build_launch_items(pgsgd_session* s) {
    if (s->d_items_split) { (void)hipFree(s->d_items_split); s->d_items_split = nullptr; }
    if (s->d_item_done) { (void)hipFree(s->d_item_done); s->d_item_done = nullptr; }
    s->h_items_split.clear();
    s->tile_split = 1;
    s->n_items_split[0] = s->n_items_split[1] = 0;
    if (s->h_items.empty()) return PGSGD_OK;
    // the windows this session runs, run by run (one run, or one per XCD: item_chunk): those a kernel with the session's
    // shard arguments would take — every shard_world-th of a run
    std::vector<pgsgd::WorkItem> own[2][pgsgd::kItemQueues];
    uint64_t n_local_min = ~0ull, tiles = 0, windows = 0;
    for (int colour = 0; colour < 2; ++colour) {
        const uint32_t base = colour ? s->n_items[0] : 0, windowless = colour == 0 ? s->n_windowless : 0, n_local = s->n_items[colour] - windowless;
        uint64_t mine = 0;
        for (uint32_t q = 0; q < pgsgd::kItemQueues; ++q) {
            const uint32_t lo = std::min(s->item_chunk[colour][q], n_local), hi = std::min(s->item_chunk[colour][q + 1], n_local);
            for (uint32_t w = lo + s->shard_rank; w < hi; w += s->shard_world) {
                own[colour][q].push_back(s->h_items[base + w]);
                tiles += s->h_items[base + w].tile_end - s->h_items[base + w].tile_begin;
                ++mine;
            }
        }
        if (mine) n_local_min = std::min<uint64_t>(n_local_min, mine);
        windows += mine;
    }
    uint32_t k = 1;
    if (s->tile_split_knob) k = s->tile_split_knob;
    else if (windows && n_local_min != ~0ull) k = pgsgd_tile_parts_for(n_local_min, tiles / windows / std::max<uint32_t>(1, s->tshard_world), s->tile_grid);
    if (k <= 1) return PGSGD_OK;
    std::vector<pgsgd::WorkItem> cut[2];
    for (int colour = 0; colour < 2; ++colour) {
        const uint32_t base = colour ? s->n_items[0] : 0, windowless = colour == 0 ? s->n_windowless : 0, n_local = s->n_items[colour] - windowless;
        for (uint32_t q = 0; q < pgsgd::kItemQueues; ++q) {
            s->item_chunk_split[colour][q] = (uint32_t)cut[colour].size();
            std::vector<pgsgd::WorkItem> parts = split_items(own[colour][q], 0, k);
            const uint32_t off = (uint32_t)cut[colour].size();
            for (pgsgd::WorkItem& wi : parts)
                if (wi.local >> pgsgd::kItemDepShift) wi.local += off << pgsgd::kItemDepShift;  // (a part names its predecessor by its index in the launch's list)
            cut[colour].insert(cut[colour].end(), parts.begin(), parts.end());
        }
        s->item_chunk_split[colour][pgsgd::kItemQueues] = (uint32_t)cut[colour].size();
        cut[colour].insert(cut[colour].end(), s->h_items.begin() + base + n_local, s->h_items.begin() + base + s->n_items[colour]);  // every window-less item
    }
    s->tile_split = k;
    s->n_items_split[0] = (uint32_t)cut[0].size();
    s->n_items_split[1] = (uint32_t)cut[1].size();
    s->h_items_split = cut[0];
    s->h_items_split.insert(s->h_items_split.end(), cut[1].begin(), cut[1].end());
    const size_t n_flags = std::max<size_t>(1, std::max(cut[0].size(), cut[1].size()));
    hipError_t e = hipSetDevice(s->device);
    if (e == hipSuccess) e = hipMalloc(&s->d_items_split, s->h_items_split.size() * sizeof(pgsgd::WorkItem));
    if (e == hipSuccess) e = hipMalloc(&s->d_item_done, n_flags * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(s->d_items_split, s->h_items_split.data(), s->h_items_split.size() * sizeof(pgsgd::WorkItem), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(s->d_item_done, 0, n_flags * sizeof(uint32_t));
    if (e != hipSuccess) { set_error("work-item lists: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP; }
    return PGSGD_OK;
}

// Long-range step pairs for the initial-layout check of a tiled session: first step uniform over all steps,
// partner uniform over the same path (the pairs the non-cooling phase draws half of the time), kept when they
// are more than eight windows apart.  The tile kernel
// moves a node end over long distances only twice per iteration (the capped far pulls of the two launches), which
// is enough to refine a layout whose global structure is there — `-N d` on a sorted graph — and too little to
// form it: from `-N g`, `-N r`, `-N h` or on a graph sorted only in blocks the tiled layouts end 6-40 % worse
// (profiles/r01/init_modes_tile_vs_per_lane.jsonl, shuffled_graphs.jsonl).  So upload measures the stress of
// the initial layout on these pairs, and when it is not small the iterations before cooling — where long
// moves happen (mu = 1 for every distance) — run the per-lane kernel.
template <class FetchPos>
static int sample_check_pairs(pgsgd_session* s, const pgsgd_graph_view* g, FetchPos&& fetch_pos) {
    pgsgd::Xoshiro256Plus rng;
    rng.seed(0x5eedc0de);
    s->check_pairs.clear();
    // "long range" = farther than eight windows: nearer pairs are moved at full strength inside the windows
    // whatever the initial layout (and the Gaussian Y of `-N d` dominates their distances)
    uint64_t total_bp = 0;
    for (uint64_t i = 0; i < g->n_nodes; ++i) total_bp += g->node_len[i];
    const double d_min = 8.0 * 2.0 * (double)s->region * ((double)total_bp / (double)g->n_nodes);
    // candidates first (the draws do not depend on the positions), their positions in one go (they may live on the device only)
    std::vector<uint64_t> idx, pos;
    for (int tries = 0; tries < 65536; ++tries) {
        const uint64_t ka = pgsgd::uniform_below(rng, g->n_steps);
        const uint64_t path = g->step_path ? g->step_path[ka] : (uint64_t)(std::upper_bound(g->path_first, g->path_first + g->n_paths + 1, ka) - g->path_first) - 1;
        const uint64_t b = g->path_first[path], cnt = g->path_first[path + 1] - b;
        if (cnt < 2) continue;
        const uint64_t kb = b + pgsgd::uniform_below(rng, cnt);
        idx.push_back(ka);
        idx.push_back(kb);
    }
    const int rc = fetch_pos(idx, pos);
    if (rc) return rc;
    for (size_t j = 0; j + 1 < idx.size() && s->check_pairs.size() < 16384; j += 2) {
        const uint64_t pa = pos[j], pb = pos[j + 1];
        if ((double)(pa > pb ? pa - pb : pb - pa) < d_min) continue;
        pgsgd_session::CheckPair cp;
        cp.end_a = g->step_handle[idx[j]];  // the end a step starts at: 2 * rank + is_reverse
        cp.end_b = g->step_handle[idx[j + 1]];
        cp.d = (float)(pa > pb ? pa - pb : pb - pa);
        s->check_pairs.push_back(cp);
    }
    return PGSGD_OK;
}

static double check_pairs_stress(const pgsgd_session* s, const float* X, const float* Y) {
    if (s->check_pairs.size() < 256) return 0.0;  // (almost) no long-range pairs: nothing global to form
    double sum = 0;
    for (const auto& cp : s->check_pairs) {
        const double dx = (double)X[cp.end_a] - (double)X[cp.end_b], dy = (double)Y[cp.end_a] - (double)Y[cp.end_b];
        const double e = (std::sqrt(dx * dx + dy * dy) - (double)cp.d) / (double)cp.d;
        sum += e * e;
    }
    return sum / (double)s->check_pairs.size();
}

static int ensure_outbox(pgsgd_session* s, uint64_t n_terms, uint32_t n_parts);
static int drain_outbox(pgsgd_session* s, unsigned long long* far_next);
static inline bool pulls_waiting(const pgsgd_session* s) { return s->ob_pending || s->pend_waiting[0] || s->pend_waiting[1]; }

extern "C" int pgsgd_session_create(const pgsgd_graph_view* g, const pgsgd_params* p, pgsgd_session** out) {
    pgsgd::clear_error();
    if (!out || !p) return PGSGD_E_INVALID;
    *out = nullptr;
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (g->n_steps == 0 || g->n_paths == 0) { set_error("graph has no path steps"); return PGSGD_E_INVALID; }
    {   // path_sgd_layout.cpp:64-74: nothing can be sampled unless some path has more than one step (the sampler would
        // draw first steps for ever, :182-192)
        bool multi = false;
        for (uint64_t i = 0; i < g->n_paths && !multi; ++i) multi = g->path_first[i + 1] - g->path_first[i] > 1;
        if (!multi) { set_error("no path has more than one step: there is no term to sample"); return PGSGD_E_INVALID; }
    }
    if (p->space == 0 || p->space_quantization_step == 0 || !(p->theta < 1.0) || p->iter_max == 0) {
        set_error("invalid SGD parameters (space, quantization step, theta < 1, iter_max)");
        return PGSGD_E_INVALID;
    }
    int dev = 0;
    rc = pick_device(p->device, &dev);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(dev));
    pgsgd::PhaseTimer timer;
    auto s = new pgsgd_session();
    s->device = dev;
    s->params = *p;
    s->n_nodes = g->n_nodes;
    s->n_steps = g->n_steps;
    s->n_paths = g->n_paths;
    s->fmt = (p->flags & PGSGD_FLAG_FP32_ATOMICS) ? pgsgd::kFmtF32 : pgsgd::kFmtQ32;
    s->upd = (p->flags & PGSGD_FLAG_HOGWILD_STORES) ? pgsgd::kUpdStore : pgsgd::kUpdAtomic;
    for (uint64_t i = 0; i < g->n_paths; ++i) {
        const uint64_t e = g->path_first[i + 1];
        if (e > g->path_first[i] && (g->step_handle[e - 1] >> 1) >= g->n_nodes) { set_error("a step names a node rank outside the graph"); delete s; return PGSGD_E_INVALID; }
    }
    std::vector<uint64_t> path_end_bp(g->n_paths, 0);   // bp length of every path (filled once the step positions are there)
    auto fail = [&](int code) {
        pgsgd_session_destroy(s);
        return code;
    };
#define S_TRY(expr)                                                                             \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return fail(_e == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP);               \
        }                                                                                       \
    } while (0)
    S_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->own_stream = true;
    // The index is built on the device from the uploaded arrays: the step handles go up first, the
    // per-node step counts (hot-node rule, outbox pool shares, range check of the handles) and the per-tile rank
    // ranges come from kernels over them; the host only cuts paths into tiles and groups tiles by region, both linear
    // in the number of TILES.
    uint32_t* d_handle = nullptr;
    S_TRY(hipMalloc(&d_handle, g->n_steps * sizeof(uint32_t)));
    struct HandleGuard { uint32_t*& p; ~HandleGuard() { if (p) (void)hipFree(p); } } handle_guard{d_handle};
    S_TRY(hipMemcpyAsync(d_handle, g->step_handle, g->n_steps * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    {
        uint32_t* d_counts = nullptr;
        unsigned int* d_bad = nullptr;
        unsigned int h_bad = 0;
        S_TRY(hipMalloc(&d_counts, g->n_nodes * sizeof(uint32_t)));
        S_TRY(hipMalloc(&d_bad, sizeof(unsigned int)));
        S_TRY(hipMemsetAsync(d_counts, 0, g->n_nodes * sizeof(uint32_t), s->stream));
        S_TRY(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), s->stream));
        const int grid = (int)std::min<uint64_t>((g->n_steps + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(pgsgd::node_steps_kernel, dim3(grid), dim3(256), 0, s->stream, d_handle, g->n_steps, (uint32_t)g->n_nodes, d_counts, d_bad);
        S_TRY(hipGetLastError());
        s->node_steps.resize(g->n_nodes);
        S_TRY(hipMemcpyAsync(s->node_steps.data(), d_counts, g->n_nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        S_TRY(hipMemcpyAsync(&h_bad, d_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, s->stream));
        S_TRY(hipStreamSynchronize(s->stream));
        (void)hipFree(d_counts);
        (void)hipFree(d_bad);
        if (h_bad) { set_error("a step names a node rank outside the graph"); return fail(PGSGD_E_INVALID); }
    }
    // Step positions (xp.cpp:607-617): the caller's array — or, when the view carries none (pgsgd_graph_view::step_pos == NULL: 4 bytes
    // per step over PCIe instead of 12, and no walk over the paths on the host), built here from the handles (pgsgd_kernels.hpp:
    // step_prefix_kernel).  The host needs a few of them back: every path's last step (its bp length) and the layout check's pairs.
    uint64_t* d_pos = nullptr;
    uint32_t* d_len = nullptr;
    struct PosGuard { uint64_t*& p; uint32_t*& l; ~PosGuard() { if (p) (void)hipFree(p); if (l) (void)hipFree(l); } } pos_guard{d_pos, d_len};
    S_TRY(hipMalloc(&d_len, g->n_nodes * sizeof(uint32_t)));
    S_TRY(hipMemcpyAsync(d_len, g->node_len, g->n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    S_TRY(hipMalloc(&d_pos, std::max<uint64_t>(1, g->n_steps) * sizeof(uint64_t)));
    if (g->step_pos) {
        S_TRY(hipMemcpyAsync(d_pos, g->step_pos, g->n_steps * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
    } else if (g->n_steps) {
        const uint64_t n_tiles = (g->n_steps + pgsgd::kPosTile - 1) / pgsgd::kPosTile;
        uint64_t *d_tile = nullptr, *d_first = nullptr, *d_pbase = nullptr;
        S_TRY(hipMalloc(&d_tile, n_tiles * sizeof(uint64_t)));
        S_TRY(hipMalloc(&d_first, (g->n_paths + 1) * sizeof(uint64_t)));
        S_TRY(hipMalloc(&d_pbase, std::max<uint64_t>(1, g->n_paths) * sizeof(uint64_t)));
        S_TRY(hipMemcpyAsync(d_first, g->path_first, (g->n_paths + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
        const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, 256 * 16);
        hipLaunchKernelGGL(pgsgd::step_len_tile_sums, dim3(grid), dim3(pgsgd::kPosBlock), 0, s->stream, d_handle, d_len, g->n_steps, n_tiles, d_tile);
        hipLaunchKernelGGL(pgsgd::scan_tile_sums, dim3(1), dim3(1024), 0, s->stream, d_tile, n_tiles);
        hipLaunchKernelGGL(pgsgd::step_prefix_kernel, dim3(grid), dim3(pgsgd::kPosBlock), 0, s->stream, d_handle, d_len, g->n_steps, n_tiles, d_tile, d_pos);
        hipLaunchKernelGGL(pgsgd::path_base_kernel, dim3((unsigned)((g->n_paths + 255) / 256)), dim3(256), 0, s->stream, d_pos, d_first, (uint32_t)g->n_paths, g->n_steps, d_pbase);
        hipLaunchKernelGGL(pgsgd::step_pos_rebase_kernel, dim3((unsigned)std::min<uint64_t>((g->n_steps + 255) / 256, 256 * 16)), dim3(256), 0, s->stream, d_pos, g->n_steps, d_first,
                           (uint32_t)g->n_paths, d_pbase);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        (void)hipFree(d_tile); (void)hipFree(d_first); (void)hipFree(d_pbase);
        if (e != hipSuccess) { set_error("step positions on the device: %s", hipGetErrorString(e)); return fail(e == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP); }
        timer.lap("step positions (device)");
    }
    // positions of chosen steps, on the host: out[i] = step_pos[idx[i]]
    auto fetch_pos = [&](const std::vector<uint64_t>& idx, std::vector<uint64_t>& out) -> int {
        out.resize(idx.size());
        if (idx.empty()) return PGSGD_OK;
        if (g->step_pos) { for (size_t i = 0; i < idx.size(); ++i) out[i] = g->step_pos[idx[i]]; return PGSGD_OK; }
        uint64_t *d_idx = nullptr, *d_out = nullptr;
        hipError_t e = hipMalloc(&d_idx, idx.size() * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(&d_out, idx.size() * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pgsgd::gather_u64_kernel, dim3((unsigned)((idx.size() + 255) / 256)), dim3(256), 0, s->stream, d_pos, d_idx, (uint64_t)idx.size(), d_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out, idx.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (d_idx) (void)hipFree(d_idx);
        if (d_out) (void)hipFree(d_out);
        if (e != hipSuccess) { set_error("reading step positions back: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP; }
        return PGSGD_OK;
    };
    {
        std::vector<uint64_t> last_idx, last_pos;
        std::vector<uint64_t> of_path;
        for (uint64_t i = 0; i < g->n_paths; ++i)
            if (g->path_first[i + 1] > g->path_first[i]) { last_idx.push_back(g->path_first[i + 1] - 1); of_path.push_back(i); }
        rc = fetch_pos(last_idx, last_pos);
        if (rc) return fail(rc);
        for (size_t j = 0; j < last_idx.size(); ++j) {
            path_end_bp[of_path[j]] = last_pos[j] + g->node_len[g->step_handle[last_idx[j]] >> 1];
            s->max_path_bp = std::max(s->max_path_bp, path_end_bp[of_path[j]]);
        }
    }
    {
        // Outbox buckets: power-of-two ranges of node ends, at most 256 of them (a workgroup stages a 64-byte line per
        // bucket in LDS).  One drain workgroup accumulates up to 2^14 ends (128 KiB of LDS); wider buckets, from
        // ~2.1e6 nodes on, are read by 2^(shift - 14) workgroups each (a second bucketing pass is the fix).
        // (Round 3 measured coarser buckets at config 4 — 123 or 62 instead of 245, 26 / 21 KB of LDS per tile workgroup
        // instead of 36, five workgroups per CU instead of four: the warm iterations get 9 % SLOWER with the fifth
        // workgroup, the cooling ones 3 % faster, the drain 30-100 % slower; profiles/r03/bench_variants_call1.txt.)
        // Round 4, with the launch's tail gone (a window's tiles as several work items): at most 128 buckets — 30 KB of LDS per
        // tile workgroup, FIVE per CU — wherever the drain then reads a bucket at most twice (graphs up to 2.1e6 nodes; two
        // drain workgroups per bucket, since 123 would leave half the CUs idle).  Config 4, 245 buckets and four per CU against
        // 123 and five: tile kernel 0.596 -> 0.636 of the roofline, drain 0.49 -> 0.58 ms per iteration, everything 0.554 ->
        // 0.581; six per CU (62 buckets): 0.630 and a drain of 0.89 ms (profiles/r04/bench_probes_late.txt).
        uint32_t ob_shift = 13;
        const char* shift_knob = pgsgd::debug_env("PGSGD_OUTBOX_SHIFT");
        if (shift_knob) ob_shift = (uint32_t)std::min(20, std::max(10, atoi(shift_knob)));  // experiment knob
        auto buckets_at = [&](uint32_t sh) { return ((2 * g->n_nodes - 1) >> sh) + 1; };
        while (buckets_at(ob_shift) > 256) ++ob_shift;
        if (!shift_knob && buckets_at(ob_shift) > 128 && ob_shift < 15) ++ob_shift;
        s->ob.shift = ob_shift;
        s->ob.qbits = pgsgd::outbox_qbits(ob_shift);
        if (const char* e = pgsgd::debug_env("PGSGD_OUTBOX_QBITS"))  // test knob: narrow packed steps, so that most messages take the path of a step too wide for the packed form
            s->ob.qbits = (uint32_t)std::min<int>((int)s->ob.qbits, std::max(2, atoi(e)));
        // A drain workgroup accumulates a whole bucket (at most 2^14 node ends = 128 KiB of LDS; wider buckets in parts).  Where
        // that leaves fewer workgroups than the device has CUs, a bucket's message stream is cut into slices (far_drain_kernel):
        // config 4, 123 buckets -> two slices each, every message unpacked once by a workgroup that sees half of them.
        // (Round 4 ran two PARTS per bucket there: 0.30 ms per launch; slices: see profiles/r05/NOTES.md.)
        s->ob_part_shift = std::min<uint32_t>(ob_shift, 14);
        if (const char* e = pgsgd::debug_env("PGSGD_OUTBOX_PART_SHIFT")) s->ob_part_shift = (uint32_t)std::min<int>((int)ob_shift, std::max(10, atoi(e)));  // experiment knob
        {
            const uint64_t base_wgs = buckets_at(ob_shift) << (ob_shift - s->ob_part_shift);
            int cus = 256;
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
            s->ob_slices = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(4, (uint64_t)cus / std::max<uint64_t>(1, base_wgs)));
            if (const char* e = pgsgd::debug_env("PGSGD_DRAIN_SLICES")) s->ob_slices = (uint32_t)std::min(8, std::max(1, atoi(e)));  // experiment knob
        }
        s->ob.n_buckets = (uint32_t)(((2 * g->n_nodes - 1) >> ob_shift) + 1);
        s->ob_bucket_steps.assign(s->ob.n_buckets, 0);
        for (uint64_t i = 0; i < g->n_nodes; ++i) {
            const uint64_t v = s->node_steps[i];
            s->max_node_steps = std::max(s->max_node_steps, v);
            s->ob_bucket_steps[(2 * i) >> ob_shift] += v;
        }
    }
    timer.lap("step handles up, steps per node (device)");
    hipDeviceProp_t prop;
    S_TRY(hipGetDeviceProperties(&prop, dev));

    s->pf_lds = (g->n_paths + 1) <= pgsgd::kPathLdsCap;
    s->lds_bytes = s->pf_lds ? (size_t)(g->n_paths + 1) * sizeof(uint64_t) : 0;

    // stream count
    if (p->n_streams) {
        s->n_streams = p->n_streams;
    } else {
        int bpc = 0;
        iter_kernel_t k = session_kernel(s, false, 0);
        S_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k, pgsgd::kBlock, s->lds_bytes));
        if (bpc < 1) bpc = 1;
        s->n_streams = (p->flags & PGSGD_FLAG_HOT_NODE_CAP) ? capped_streams(s, prop.multiProcessorCount, bpc, p->min_term_updates)
                                                            : auto_streams(s, prop.multiProcessorCount, bpc);
    }
    // Small lane-bound graphs run an iteration in two passes (pgsgd_kernels.hpp): every stream the GPU holds samples, one
    // workgroup with the lanes the busiest node allows moves the ends in LDS.  Taken with an automatic stream count when
    // the 2N coordinate words fit a compute unit's LDS and the rule allows at most kSplitMaxLanes lanes (beyond, one
    // compute unit's instruction issue is slower than the single-pass kernel's memory round trips: DRB1-3123, 5 632
    // lanes, 9.3 against 6.7 ms; 3 584 lanes 4.4 against 4.2; LPA and chr6.C4, 1 536 / 1 792 lanes, 50 / 42 against
    // 104 / 81; DRB1-3123_unsorted, 128 lanes, 19 against 98: profiles/r03/split_vs_piped.jsonl).
    if (s->fmt == pgsgd::kFmtQ32 && s->upd == pgsgd::kUpdAtomic && p->terms_per_anchor <= 1 && g->n_steps < 0xffffffffull &&
        !(p->flags & (PGSGD_FLAG_HOT_NODE_CAP | PGSGD_FLAG_NO_PIPELINE | PGSGD_FLAG_NO_SPLIT | PGSGD_FLAG_COORD_LOAD_PLAIN | PGSGD_FLAG_ABLATE(15)))) {
        uint64_t max_lanes = kSplitMaxLanes;
        if (const char* e = pgsgd::debug_env("PGSGD_SPLIT_MAX_LANES")) max_lanes = (uint64_t)std::max(0L, atol(e));  // experiment knob
        // parity knob: an explicit stream count takes the two passes too (one stream, one lane = the sequential program)
        const bool forced = p->n_streams && p->n_streams <= (uint32_t)pgsgd::kResidentBlock && pgsgd::debug_env("PGSGD_SPLIT_FORCE");
        const size_t need = (size_t)2 * g->n_nodes * sizeof(uint64_t);
        const size_t have = std::max<size_t>(prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor);
        if ((forced || (!p->n_streams && s->n_streams <= max_lanes)) && need <= have) {
            // (the attribute belongs to the kernel, not to the session: it is set to what the device has, so that sessions of
            // different graphs do not lower it under each other; a request the device turns down is no error — the graph runs
            // the single-pass kernel)
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(pgsgd::apply_terms_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)have) == hipSuccess &&
                            hipFuncSetAttribute(reinterpret_cast<const void*>(pgsgd::sort_apply_terms_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)have) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            if (ok) {
                s->split = true;
                s->resident_lds = need;
                s->split_chunk = kSplitChunkTerms;
                if (const char* e = pgsgd::debug_env("PGSGD_SPLIT_CHUNK")) s->split_chunk = (uint64_t)std::max(1L, atol(e));  // test knob: terms per chunk
                uint32_t lane_cap = kSplitApplyLanes;
                if (const char* e = pgsgd::debug_env("PGSGD_SPLIT_APPLY_LANES")) lane_cap = (uint32_t)std::min<long>(pgsgd::kResidentBlock, std::max(64L, atol(e)));  // experiment knob
                s->apply_lanes = std::min<uint32_t>(s->n_streams, lane_cap);
                if (!forced) {  // sampler streams: what the GPU holds, at least eight terms per stream and iteration
                    int bpc = 0;
                    if (s->pf_lds) S_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, pgsgd::sample_terms_kernel<true>, pgsgd::kBlock, s->lds_bytes));
                    else S_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, pgsgd::sample_terms_kernel<false>, pgsgd::kBlock, 0));
                    const uint64_t full = (uint64_t)prop.multiProcessorCount * (uint64_t)std::max(1, bpc) * pgsgd::kBlock;
                    uint64_t ls = std::min<uint64_t>(full, std::max<uint64_t>(s->apply_lanes, p->min_term_updates / 8));
                    ls = std::max<uint64_t>(64, (ls / 64) * 64);
                    if (ls >= (uint64_t)pgsgd::kBlock) ls = (ls / pgsgd::kBlock) * pgsgd::kBlock;
                    s->n_streams = (uint32_t)std::max<uint64_t>(ls, s->apply_lanes);
                }
            }
        }
    }

    // region-exclusive tiles: with the default coordinate format, update mode and term stream, an
    // automatic stream count, and a graph that is big enough and whose hottest node does not ask for
    // fewer lanes than the GPU holds; everything else runs the per-lane kernel
    if (!(p->flags & (PGSGD_FLAG_NO_TILES | PGSGD_FLAG_COORD_LOAD_PLAIN | PGSGD_FLAG_ABLATE(15))) && s->fmt == pgsgd::kFmtQ32 &&
        s->upd == pgsgd::kUpdAtomic && !p->n_streams && p->terms_per_anchor <= 1) {
        // Region size R = 256 (tiles of 224 steps), 256 lanes per workgroup = one lane per 4 window ends.  A launch has
        // N / 2R work items and a workgroup takes one at a time, so smaller regions fill the chip from smaller graphs
        // (300k nodes: 1.97e10 terms/s against 1.33e10 with R = 512) and cost nothing on large ones (1e6 nodes, five
        // seeds each: 3.10e10 terms/s, stress 0.246 against 2.98e10, 0.255 with R = 512;
        // profiles/r01/tiles_region_256_vs_512.jsonl).  R = 128 with 256 lanes diverges.
        bool region_given = false;
        if (p->flags & PGSGD_FLAG_REGION_128) {  // (a multi-GPU run sharded by region on a graph that 256-node windows would not fill: pgsgd_shard_flags)
            s->region = 128;
            s->tile_steps = 112;
        }
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_REGION")) {  // experiment knob: region size in nodes (a multiple of 8)
            const long r = atol(e);
            if (r >= 32 && r <= 2048 && r % 8 == 0) {
                s->region = (uint32_t)r;
                s->tile_steps = (uint32_t)(r - r / 8);
                region_given = true;
            }
        }
        long steps_given = 0;
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_STEPS")) steps_given = atol(e);  // experiment knob: steps per tile (at most the region size)
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_SUBSTEPS")) {  // experiment knob: window refreshes per iteration
            const long k = atol(e);
            if (k >= 1 && k <= 64) s->tile_substeps = (uint32_t)k;
        }
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_BLOCK")) {  // experiment knob: lanes per workgroup
            const long b = atol(e);
            if (b >= 64 && b <= pgsgd::kTileBlock && b % 64 == 0) s->tile_block = (uint32_t)b;
        }
        const uint64_t cap = 2 * s->n_steps / std::max<uint64_t>(1, s->max_node_steps);
        // LDS of a workgroup: the window, the tile's records, the outbox (staged lines, the waves' queues, counters), the lock bits.  The warm
        // launches hand the rings TWO messages per lane and call (sgd_tile_kernel<.., PUSH = 2>: 64 more queue entries per wave) where that
        // does not cost the session a workgroup per CU — config 4: 30 848 bytes, five per CU either way; 1e7 nodes (153 buckets): 33 240
        // bytes would be four per CU where 30 936 are five, and the fifth workgroup is worth more than the halved calls (+6.7 % against
        // +2 ... 3.5 % of a warm launch) — debug knob PGSGD_TILE_PUSH=1 / 2: one / two whatever it costs.
        auto tile_lds_for = [&](uint32_t push) {
            return (size_t)4 * s->region * sizeof(uint64_t) + (size_t)s->tile_steps * sizeof(uint4) + pgsgd::outbox_lds_bytes(s->ob.n_buckets, pgsgd::tile_wq_cap(push)) +
                   pgsgd::tile_lock_words(s->region) * sizeof(uint32_t);
        };
        int bpc = 0, bpc2 = 0;
        S_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, tile_kernel(pgsgd::kFarTwoSided, pgsgd::kMathFast), (int)s->tile_block, tile_lds_for(1)));
        S_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc2, tile_kernel(pgsgd::kFarTwoSided, pgsgd::kMathFast, false, true, false, true), (int)s->tile_block, tile_lds_for(2)));
        if (bpc < 1) bpc = 1;
        {
            const char* e = pgsgd::debug_env("PGSGD_TILE_PUSH");
            s->tile_push = e ? (atoi(e) == 2 ? 2u : 1u) : bpc2 >= bpc ? 2u : 1u;
            if (s->tile_push == 2) bpc = std::max(1, bpc2);
        }
        s->tile_lds = tile_lds_for(s->tile_push);
        s->tile_lds1 = tile_lds_for(1);
        // the hottest node must leave room for at least four workgroups per CU (the occupancy the kernel
        // was validated at); between that and full residency the grid is cut to the hot-node cap
        const uint64_t cu_lanes = (uint64_t)prop.multiProcessorCount * s->tile_block;
        if (steps_given >= 16 && steps_given <= (long)s->region && region_given) s->tile_steps = (uint32_t)steps_given;
        // (Rounds 3-4 fitted R to the launch — the multiple of 8 in [240, 272] whose work items filled their two or three rounds
        // over the resident workgroups best, R = 248 at config 4 — which a window's tiles as several work items made
        // pointless: build_launch_items.  Measured with those, five workgroups per CU: R = 248 / 256 / 264 / 272 / 288 ->
        // 0.636 / 0.634 / 0.623 / 0.618 / 0.600.)
        if (!region_given && steps_given >= 16 && steps_given <= (long)s->region) s->tile_steps = (uint32_t)steps_given;
        s->tile_lds = tile_lds_for(s->tile_push);
        s->tile_lds1 = tile_lds_for(1);
        // parity knobs: PGSGD_TILE_FORCE=1 runs the tile kernel on a graph of any shape, PGSGD_TILE_GRID and
        // PGSGD_TILE_LANES bound the workgroups of a launch and the lanes of a tile.  One workgroup with one
        // lane is a sequential program that the oracle mirrors bit for bit (tests/test_gpu_parity.py).
        const bool force = pgsgd::debug_env("PGSGD_TILE_FORCE") != nullptr;
        s->tile_forced = force;
        s->snapshot_pass = pgsgd::debug_env("PGSGD_TILE_SNAPSHOT_PASS") != nullptr;
        s->tile_pair_uniform = (p->flags & PGSGD_FLAG_NO_PARTNER_PAIRS) ? 0u : 2u;   // partner quads (pgsgd_tiles.hpp: tile_quad_partner)
        s->tile_lock_mu = (p->flags & PGSGD_FLAG_LOCK_WINDOW_ENDS) ? 0.1f : 0.0f;
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_LOCK_MU")) s->tile_lock_mu = (float)std::max(0.0, atof(e));  // experiment knob: the threshold
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_SNAP_EVERY")) s->tile_snap_every = (uint32_t)std::min(8, std::max(1, atoi(e)));
        s->tile_lane_coin = pgsgd::debug_env("PGSGD_TILE_LANE_COIN") != nullptr;
        s->tile_rotate = pgsgd::debug_env("PGSGD_TILE_ROTATE") != nullptr;
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_UNTIL")) s->tile_until = (uint64_t)std::max(0L, atol(e));
        s->tile_tail = pgsgd::debug_env("PGSGD_TILE_TAIL") != nullptr;
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_FAR_RELAX_MAX")) s->tile_far_relax_max = (float)std::min(2.0, std::max(0.0, atof(e)));
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_FAR_RELAX_SLOPE")) s->tile_far_relax_slope = (float)std::min(2.0, std::max(0.0, atof(e)));
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_FAR_RELAX")) s->tile_far_relax_override = (float)std::min(1.0, std::max(0.0, atof(e)));
        if (s->tile_pair_uniform && pgsgd::debug_env("PGSGD_TILE_PAIRS")) s->tile_pair_uniform = 1;  // A/B knob: the partner pairs of rounds 4-6 (oracle: ORC_TILE_PAIRS)
        if (s->tile_lane_coin) s->tile_pair_uniform = 0;  // (pairs need the wave's lanes on one partner path: an odd lane reads its even neighbour's draw)
        if (const char* e = pgsgd::debug_env("PGSGD_TILE_WQ")) s->tile_wq_threshold = (uint32_t)std::min<int>(64 * pgsgd::kWqPush, std::max(1, atoi(e)));
        // the tile kernel converts path distances through fp64 (term_displacement<true>): positions must stay below 2^52
        bool short_paths = true, paths_32 = true;  // (the fast instance keeps positions as 32-bit words: every path shorter than 2^32 bp)
        for (uint64_t q = 0; q < g->n_paths && short_paths; ++q)
            if (g->path_first[q + 1] > g->path_first[q]) {
                const uint64_t path_end = path_end_bp[q];
                short_paths = path_end < (1ull << 52);
                paths_32 = paths_32 && path_end < (1ull << 32);
            }
        s->tile_math = ((p->flags & PGSGD_FLAG_EXACT_MATH) || !paths_32) ? pgsgd::kMathExact : pgsgd::kMathFast;
        // A schedule of fewer than 15 iterations runs the per-lane kernel.  The tile kernel gives long-range pairs gentle,
        // averaged pulls while the learning rate is above their distance and needs the schedule's length to bring them
        // home: final sampled stress at config 4, tile kernel / per-lane kernel (the reference's rule), for -x 3 / 5 /
        // 8 / 10 / 12 / 15 / 20 / 30: 1374 / 17, 1.07 / 0.31, 0.40 / 0.186, 0.32 / 0.22, 0.265 / 0.225, 0.240 / 0.2285,
        // 0.234 / 0.228, 0.238 / 0.245 (profiles/r03/short_schedules_tile_vs_per_lane.jsonl).
        const bool long_schedule = p->iter_max >= 15;
        if ((force || (cap >= 4 * cu_lanes && long_schedule)) && short_paths && g->n_nodes >= 8ull * s->region && g->n_steps < 0xffffffffull && g->n_nodes < 0x7fffffffull) {
            bpc = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)bpc, std::max<uint64_t>(1, cap / cu_lanes)));
            // Steps per tile: T = R - R / 8.  A tile's share of an iteration is drawn by the workgroup's lanes in trips of 256
            // terms, and T = 204 (eight full trips at ten terms per step, where 217 makes 8.5 and 224 made 8.75) measured 1 %
            // faster (tile kernel's roofline fraction at config 4, R = 248, T = 128 / 153 / 179 / 192 / 204 / 217 / 230: 0.478 /
            // 0.491 / 0.494 / 0.498 / 0.497-0.500 / 0.492 / 0.479, the last with 500 of 202 927 tiles no longer fitting a
            // window; profiles/r03/bench_variants_call26_tile_steps.txt) — and laid the graphs out 1-4 % worse: 256 lanes on
            // 408 node ends instead of 448 are denser concurrent updates (final stress at config 4 0.2177 against 0.2148, at
            // 1e7 nodes 1.08x the per-lane kernel's against 1.04x).  Not taken; PGSGD_TILE_STEPS is the experiment's knob.
            std::vector<RawTile> raw = cut_tiles(g, s->tile_steps);
            rc = device_tile_stats(s->stream, d_handle, s->tile_steps, raw);
            if (rc) return fail(rc);
            // Work order.  By default the items of a colour are one run by decreasing size.  A session whose windows are cut
            // into parts (build_launch_items: launches of few rounds on a full device) takes them in NODE order instead, one run
            // per XCD (TileArgs::chunk), every run cut on its own: the workgroups of an XCD then work on neighbouring windows
            // and find the partner records just outside a tile in their common L2.  Whole windows in that order were slower
            // (rounds 3-4: neighbours in lock step); in parts, with the kernel bound by HBM lines, it is 3 % faster (config 4,
            // driver's window 0.635 -> 0.654, whole schedule 0.644 -> 0.667, same stress: profiles/r04/NOTES.md section 6).
            // PGSGD_TILE_ORDER=region / size forces either (experiments; the parity tests' small sessions are whole windows by size).
            const char* order = pgsgd::debug_env("PGSGD_TILE_ORDER");
            bool by_size = !(order && !strcmp(order, "region"));
            HostTiles ht = group_tiles(raw, g->n_nodes, s->region, by_size);
            if (!order && !pgsgd::debug_env("PGSGD_TILE_GRID") && !pgsgd::debug_env("PGSGD_TILE_SPLIT")) {
                uint64_t n_local_min = ~0ull, windows = 0, tiles = 0;
                for (int colour = 0; colour < 2; ++colour) {
                    uint64_t n_local = 0;
                    for (const pgsgd::WorkItem& wi : ht.items[colour])
                        if (wi.local) { ++n_local; tiles += wi.tile_end - wi.tile_begin; }
                    if (n_local) n_local_min = std::min(n_local_min, n_local);
                    windows += n_local;
                }
                if (windows && pgsgd_tile_parts_for(n_local_min, tiles / windows, (uint64_t)prop.multiProcessorCount * bpc) > 1) {
                    by_size = false;
                    ht = group_tiles(raw, g->n_nodes, s->region, false);
                }
            }
            memcpy(s->item_chunk, ht.chunk, sizeof s->item_chunk);
            if (const char* e = pgsgd::debug_env("PGSGD_TILE_LANES")) {
                const long l = atol(e);
                if (l >= 1)
                    for (pgsgd::Tile& t : ht.tiles) t.lanes = std::min<uint32_t>(t.lanes, (uint32_t)l);
            }
            timer.lap("tile table (device statistics, host grouping)");
            if ((p->flags & PGSGD_FLAG_ONE_SIDED_FAR) && !ht.n_nonlocal) {
                // experiment, not the reference's rule: a far term moves only its first end, by twice the step.
                // Ignored on graphs with window-less tiles.
                s->tile_far = pgsgd::kFarExclusive;
            }
            s->tile_grid = (uint32_t)(prop.multiProcessorCount * bpc);
            if (const char* e = pgsgd::debug_env("PGSGD_TILE_GRID")) {
                const long gr = atol(e);
                if (gr >= 1 && gr <= (long)s->tile_grid) s->tile_grid = (uint32_t)gr;
            }
            s->tile_lanes = s->tile_grid * s->tile_block;
            // A graph whose node ranks do not follow its paths has tiles without a window; all their ends live in
            // global memory and all their terms count as far (learning rate capped).  A few are fine (measured:
            // 10 % in relabelled stretches, same stress); when they are many the cap throttles the whole layout
            // (a randomly numbered graph ends at stress 3e4, profiles/r01/shuffled_graphs.jsonl): per-lane kernel.
            s->tiled = force || 10 * ht.n_nonlocal <= ht.tiles.size();
            if (s->tiled) {
                rc = sample_check_pairs(s, g, fetch_pos);
                if (rc) return fail(rc);
                s->tile_steps_total = ht.steps_total;
                {   // a launch sends at most one message per term of a tile with a window, two per term of a window-less tile
                    double bound = 0;
                    for (int colour = 0; colour < 2; ++colour) {
                        uint64_t with_window = 0, without = 0;
                        for (const pgsgd::WorkItem& wi : ht.items[colour])
                            for (uint32_t ti = wi.tile_begin; ti < wi.tile_end; ++ti) (wi.local ? with_window : without) += ht.tiles[ti].n;
                        bound = std::max(bound, ((double)with_window + 2.0 * (double)without) / (double)std::max<uint64_t>(1, ht.steps_total));
                    }
                    s->ob_msgs_per_term = bound;
                }
                s->n_tiles = ht.tiles.size();
                s->h_tiles = ht.tiles;
                s->n_nonlocal_tiles = ht.n_nonlocal;
                s->n_items[0] = (uint32_t)ht.items[0].size();
                s->n_items[1] = (uint32_t)ht.items[1].size();
                s->n_windowless = (uint32_t)ht.n_nonlocal;
                std::vector<pgsgd::WorkItem> all(ht.items[0]);
                all.insert(all.end(), ht.items[1].begin(), ht.items[1].end());
                s->h_items = all;
                if (const char* e = pgsgd::debug_env("PGSGD_TILE_SPLIT")) s->tile_split_knob = (uint32_t)std::min(256, std::max(1, atoi(e)));
                s->items_one_run = by_size;
                rc = build_launch_items(s);
                if (rc) return fail(rc);
                S_TRY(hipMalloc(&s->d_tiles, std::max<size_t>(1, ht.tiles.size()) * sizeof(pgsgd::Tile)));
                S_TRY(hipMalloc(&s->d_items, std::max<size_t>(1, all.size()) * sizeof(pgsgd::WorkItem)));
                S_TRY(hipMalloc(&s->d_queue, 3 * pgsgd::kItemQueues * sizeof(uint32_t)));
                S_TRY(hipMemset(s->d_queue, 0, 3 * pgsgd::kItemQueues * sizeof(uint32_t)));
                // (the attribute belongs to the kernel, not to the session: always the widest part a drain workgroup accumulates,
                // so that sessions of different graphs do not lower it under each other)
                S_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pgsgd::far_drain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(sizeof(uint64_t) << 14)));
                S_TRY(hipMalloc(&s->d_term0, (ht.tiles.size() + 1) * sizeof(uint64_t)));
                S_TRY(hipMalloc(&s->d_clock, 12 * sizeof(unsigned long long)));  // [6..9]: TileArgs::tail_probe; [10]: TileArgs::term_count
                S_TRY(hipMemset(s->d_clock, 0, 12 * sizeof(unsigned long long)));
                S_TRY(hipMalloc(&s->d_far, 4 * sizeof(unsigned long long)));
                S_TRY(hipMemset(s->d_far, 0, 4 * sizeof(unsigned long long)));
                S_TRY(hipMemcpy(s->d_tiles, ht.tiles.data(), ht.tiles.size() * sizeof(pgsgd::Tile), hipMemcpyHostToDevice));
                {   // what the kernel reads of a tile before its first term, in one 16-byte load (TileArgs::tile_heads)
                    std::vector<uint4> heads(std::max<size_t>(1, ht.tiles.size()));
                    for (size_t i = 0; i < ht.tiles.size(); ++i) {
                        const pgsgd::Tile& t = ht.tiles[i];
                        heads[i] = make_uint4((uint32_t)t.t0, (t.n & 0xffffu) | (std::min<uint32_t>(t.lanes, 0xffffu) << 16), (uint32_t)g->path_first[t.path],
                                              (uint32_t)(g->path_first[t.path + 1] - g->path_first[t.path]));
                    }
                    S_TRY(hipMalloc(&s->d_tile_heads, heads.size() * sizeof(uint4)));
                    S_TRY(hipMemcpy(s->d_tile_heads, heads.data(), heads.size() * sizeof(uint4), hipMemcpyHostToDevice));
                }
                S_TRY(hipMemcpy(s->d_items, all.data(), all.size() * sizeof(pgsgd::WorkItem), hipMemcpyHostToDevice));
            }
        }
    }

    timer.lap("stream, occupancy, tile upload");
    // step records: upload the SoA arrays, pack on the device, drop the staging copies
    {
        S_TRY(hipMalloc(&s->d_recs, g->n_steps * sizeof(uint4)));
        if (s->tiled) S_TRY(hipMalloc(&s->d_recs2, pgsgd::recs2_pieces(g->n_steps) * sizeof(uint4)));
        const int grid = (int)std::min<uint64_t>((g->n_steps + 255) / 256, 256 * 8);
        unsigned int* d_bad = nullptr;
        unsigned int h_bad = 0;
        S_TRY(hipMalloc(&d_bad, sizeof(unsigned int)));
        S_TRY(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), s->stream));
        hipLaunchKernelGGL(pgsgd::build_step_records, dim3(grid), dim3(256), 0, s->stream, d_handle, d_pos, d_len, (uint32_t)g->n_nodes, g->n_steps,
                           s->d_recs, s->d_recs2, d_bad);
        S_TRY(hipGetLastError());
        S_TRY(hipMemcpyAsync(&h_bad, d_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, s->stream));
        S_TRY(hipStreamSynchronize(s->stream));
        (void)hipFree(d_bad);
        if (h_bad) {
            set_error("a step names a node rank outside the graph");
            return fail(PGSGD_E_INVALID);
        }
        (void)hipFree(d_pos);
        d_pos = nullptr;
        (void)hipFree(d_len);
        d_len = nullptr;
    }
    if (s->tiled) {  // the snapshot pass reads the handles (4 bytes per step), not the records: the session keeps them
        s->d_step_handle = d_handle;
        d_handle = nullptr;
    }
    timer.lap("step records (upload + pack)");
    S_TRY(hipMalloc(&s->d_path_first, (g->n_paths + 1) * sizeof(uint64_t)));
    S_TRY(hipMemcpy(s->d_path_first, g->path_first, (g->n_paths + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    {
        const size_t nz = pgsgd_zeta_table_size(p->space, p->space_max, p->space_quantization_step);
        std::vector<double> z(nz);
        rc = pgsgd_zeta_table(p->theta, p->space, p->space_max, p->space_quantization_step, z.data(), nz);
        if (rc) return fail(rc);
        S_TRY(hipMalloc(&s->d_zetas, nz * sizeof(double)));
        S_TRY(hipMemcpy(s->d_zetas, z.data(), nz * sizeof(double), hipMemcpyHostToDevice));
        pgsgd::ZipfConst zc;
        zc.init(p->theta);
        std::vector<double2> zd(nz);
        for (size_t i = 0; i < nz; ++i) zd[i] = make_double2(z[i], pgsgd::zipf_denominator(zc, z[i]));
        S_TRY(hipMalloc(&s->d_zeta_denom, nz * sizeof(double2)));
        S_TRY(hipMemcpy(s->d_zeta_denom, zd.data(), nz * sizeof(double2), hipMemcpyHostToDevice));
    }
    S_TRY(hipMalloc(&s->d_coords, g->n_nodes * 2 * sizeof(uint64_t)));
    S_TRY(hipMemset(s->d_coords, 0, g->n_nodes * 2 * sizeof(uint64_t)));
    S_TRY(hipMalloc(&s->d_rng, (size_t)s->n_streams * 4 * sizeof(uint64_t)));
    S_TRY(hipMalloc(&s->d_delta_max, 2 * sizeof(unsigned int)));
    S_TRY(hipMemset(s->d_delta_max, 0, 2 * sizeof(unsigned int)));
    S_TRY(hipHostMalloc(&s->h_delta_max, 2 * sizeof(unsigned int)));
    s->h_delta_max[0] = s->h_delta_max[1] = 0;
    {
        const int grid = (int)((s->n_streams + 255) / 256);
        hipLaunchKernelGGL(pgsgd::seed_streams_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_rng, s->n_streams,
                           p->seed + (uint64_t)p->stream_offset);
        S_TRY(hipGetLastError());
        S_TRY(hipStreamSynchronize(s->stream));
    }
    pgsgd::DevConst& c = s->dc;
    c.recs = s->d_recs;
    c.path_first = s->d_path_first;
    c.zetas = s->d_zetas;
    c.zeta_denom = s->d_zeta_denom;
    c.coords = s->d_coords;
    c.rng = s->d_rng;
    c.delta_max_bits = s->d_delta_max;
    c.frame_flag = s->d_delta_max + 1;
    c.n_steps = g->n_steps;
    c.n_paths = (uint32_t)g->n_paths;
    c.n_streams = s->n_streams;
    c.n_nodes = (uint32_t)g->n_nodes;
    c.space = p->space;
    c.space_max = p->space_max;
    c.space_quant = p->space_quantization_step;
    c.terms_per_anchor = p->terms_per_anchor ? p->terms_per_anchor : 1;
    c.seed_base = p->seed + (uint64_t)p->stream_offset;
    s->tile_seed_base = c.seed_base;
    c.zc.init(p->theta);
    c.node_steps = nullptr;
    c.hot_scale = (float)((double)s->n_streams / (double)g->n_steps);
    if (!p->n_streams && (p->flags & PGSGD_FLAG_HOT_NODE_CAP) && (double)s->n_streams * (double)s->max_node_steps > (double)g->n_steps) {
        // some node sees more than one concurrent term: cap the learning rate of the terms that touch busy nodes
        S_TRY(hipMalloc(&s->d_node_steps, g->n_nodes * sizeof(uint32_t)));
        S_TRY(hipMemcpy(s->d_node_steps, s->node_steps.data(), g->n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice));
        c.node_steps = s->d_node_steps;
    }
    c.xf.x_off = c.xf.y_off = 0.0;
    c.xf.scale = c.xf.inv_scale = 1.0f;
    if (s->tiled) {  // the tile kernel's Zipf table: one entry per jump length
        uint64_t longest = 0;
        for (uint64_t q = 0; q < g->n_paths; ++q) longest = std::max(longest, g->path_first[q + 1] - g->path_first[q]);
        const uint64_t n_entries = std::min<uint64_t>(p->space, longest) + 1;  // a jump is min(space, steps to the path's end)
        S_TRY(hipMalloc(&s->d_zipf_tab, n_entries * sizeof(double2)));
        hipLaunchKernelGGL(pgsgd::zipf_tab_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, s->stream, s->d_zeta_denom,
                           (uint32_t)std::min<uint64_t>(p->space_max, 0xffffffffull), (uint32_t)std::min<uint64_t>(p->space_quantization_step, 0xffffffffull),
                           c.zc.omt_e, c.zc.omt_frac, (uint32_t)n_entries, s->d_zipf_tab);
        S_TRY(hipGetLastError());
        S_TRY(hipStreamSynchronize(s->stream));
    }
    timer.lap("tables, streams");
    // The drain beside the next launch (pgsgd_session::async_drain): schedules as long as the reference's default.  Measured with the
    // evaluator that has no sampling error (profiles/r06/NOTES.md section 3): free at `-x 30`, costly on short schedules.
    // And graphs whose outbox buckets one drain workgroup holds (up to 2.1e6 nodes): beyond that a bucket is read by 2^(shift - 14)
    // workgroups of 128 KiB of LDS each, which find no room on a CU beside the tile kernel's five — the drain then waits for the
    // launch it was meant to run beside and delays the next (measured at 1e7 nodes: 4.92e10 against 5.11e10 terms/s, stress +0.7 %).
    s->async_drain = s->tiled && p->iter_max >= 30 && !(p->flags & PGSGD_FLAG_SYNC_DRAIN) && s->fmt == pgsgd::kFmtQ32 && s->ob.shift == s->ob_part_shift
                     && s->n_nonlocal_tiles == 0;   // (a window-less tile's every move is a message: none of them may wait a launch)
    if (const char* e = pgsgd::debug_env("PGSGD_ASYNC_FROM")) s->async_from = (uint64_t)std::max(0L, atol(e));
    if (const char* e = pgsgd::debug_env("PGSGD_ASYNC_DRAIN")) s->async_drain = s->tiled && s->fmt == pgsgd::kFmtQ32 && atoi(e) != 0;   // experiment / parity knob
    if (s->tiled && p->min_term_updates) {  // the message pool for iterations of the default length (grown later if a call asks for more)
        rc = ensure_outbox(s, p->min_term_updates, 1);
        if (rc) return fail(rc);
        timer.lap("far-update message pool");
    }
    *out = s;
    return PGSGD_OK;
#undef S_TRY
}

extern "C" void pgsgd_session_destroy(pgsgd_session* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto& ev : s->pending_events) for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev.e[i]);
    for (auto& ev : s->free_events) for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev.e[i]);
    if (s->d_recs) (void)hipFree(s->d_recs);
    if (s->d_path_first) (void)hipFree(s->d_path_first);
    if (s->d_zetas) (void)hipFree(s->d_zetas);
    if (s->d_zeta_denom) (void)hipFree(s->d_zeta_denom);
    if (s->d_terms) (void)hipFree(s->d_terms);
    if (s->d_zipf_tab) (void)hipFree(s->d_zipf_tab);
    if (s->d_node_steps) (void)hipFree(s->d_node_steps);
    if (s->d_coords) (void)hipFree(s->d_coords);
    if (s->d_base) (void)hipFree(s->d_base);
    if (s->d_rng) (void)hipFree(s->d_rng);
    if (s->d_delta_max) (void)hipFree(s->d_delta_max);
    if (s->d_tiles) (void)hipFree(s->d_tiles);
    if (s->d_tile_heads) (void)hipFree(s->d_tile_heads);
    if (s->d_items) (void)hipFree(s->d_items);
    if (s->d_items_split) (void)hipFree(s->d_items_split);
    if (s->d_item_done) (void)hipFree(s->d_item_done);
    if (s->d_queue) (void)hipFree(s->d_queue);
    if (s->d_far) (void)hipFree(s->d_far);
    if (s->d_clock) (void)hipFree(s->d_clock);
    if (s->d_recs2) (void)hipFree(s->d_recs2);
    if (s->d_step_handle) (void)hipFree(s->d_step_handle);
    if (s->ob.pool) (void)hipFree(s->ob.pool);
    if (s->ob.next) (void)hipFree(s->ob.next);
    if (s->ob.fill) (void)hipFree(s->ob.fill);
    if (s->drain_stream) (void)hipStreamSynchronize(s->drain_stream);
    if (s->ob1.pool) (void)hipFree(s->ob1.pool);
    if (s->ob1.fill) (void)hipFree(s->ob1.fill);
    if (s->drain_stream) {   // (ob1.next / ob1.spill are the session's own only when the second outbox was built)
        if (s->ob1.next) (void)hipFree(s->ob1.next);
        if (s->ob1.spill) (void)hipFree(s->ob1.spill);
    }
    for (int c = 0; c < 2; ++c) {
        if (s->d_pend[c]) (void)hipFree(s->d_pend[c]);
        if (s->ev_launch[c]) (void)hipEventDestroy(s->ev_launch[c]);
        if (s->ev_drain[c]) (void)hipEventDestroy(s->ev_drain[c]);
    }
    for (auto& de : s->free_drain_events) { (void)hipEventDestroy(de.a); (void)hipEventDestroy(de.b); }
    for (auto& de : s->pending_drain_events) { (void)hipEventDestroy(de.a); (void)hipEventDestroy(de.b); }
    if (s->drain_stream) (void)hipStreamDestroy(s->drain_stream);
    if (s->d_ob_chunk0) (void)hipFree(s->d_ob_chunk0);
    if (s->d_ob_cap) (void)hipFree(s->d_ob_cap);
    if (s->d_ob_overflow) (void)hipFree(s->d_ob_overflow);
    if (s->d_ob_spill) (void)hipFree(s->d_ob_spill);
    if (s->d_ob_partial) (void)hipFree(s->d_ob_partial);
    if (s->d_term0) (void)hipFree(s->d_term0);
    if (s->h_delta_max) (void)hipHostFree(s->h_delta_max);
    if (s->stream && s->own_stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// Fixed-point frame for kFmtQ32: a power-of-two number of quanta per bp such that 2^32 quanta span
// 8x the larger of the initial layout's extent and the longest path (the scale a path-guided
// layout settles at), centred on the initial layout.
static int choose_xform(pgsgd_session* s, const float* X, const float* Y) {
    const uint64_t n_ends = 2 * s->n_nodes;
    double minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
    for (uint64_t i = 0; i < n_ends; ++i) {
        if (std::isfinite(X[i])) { minx = std::min<double>(minx, X[i]); maxx = std::max<double>(maxx, X[i]); }
        if (std::isfinite(Y[i])) { miny = std::min<double>(miny, Y[i]); maxy = std::max<double>(maxy, Y[i]); }
    }
    if (!(minx <= maxx) || !(miny <= maxy)) { set_error("the initial layout has no finite coordinate"); return PGSGD_E_INVALID; }
    double factor = 8.0;
    if (const char* e = pgsgd::debug_env("PGSGD_FRAME_SPAN")) factor = std::max(1.0, atof(e));  // test knob: a frame this tight must widen itself
    const double extent = std::max({maxx - minx, maxy - miny, (double)s->max_path_bp, 1.0});
    const int span_log2 = (int)std::ceil(std::log2(factor * extent));
    const double span = std::ldexp(1.0, span_log2);
    pgsgd::Xform& xf = s->dc.xf;
    xf.scale = (float)std::ldexp(1.0, 32 - span_log2);
    xf.inv_scale = (float)std::ldexp(1.0, span_log2 - 32);
    xf.x_off = 0.5 * (minx + maxx) - 0.5 * span;
    xf.y_off = 0.5 * (miny + maxy) - 0.5 * span;
    return PGSGD_OK;
}

extern "C" int pgsgd_session_upload_coords(pgsgd_session* s, const float* X, const float* Y) {
    pgsgd::clear_error();
    if (!s || !X || !Y) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    if (pulls_waiting(s)) {  // far pulls of an earlier layout must not reach the new one (their drain also resets the counters)
        const int rc = pgsgd_session_flush(s);
        if (rc) return rc;
    }
    const uint64_t n_ends = 2 * s->n_nodes;
    if (s->fmt == pgsgd::kFmtQ32) {
        const int rc = choose_xform(s, X, Y);
        if (rc) return rc;
        s->frame_doublings = 0;
    }
    if (s->tiled && !s->tile_forced) {
        // 0.1 = long-range distances off by a third on average; `-N d` on a sorted graph measures ~0.01
        const double st = check_pairs_stress(s, X, Y);
        s->warm_per_lane = !(st <= 0.1);
        if (s->params.progress)
            fprintf(stderr, "[odgi::path_linear_sgd_layout] long-range stress of the initial layout %.4g: %s\n", st,
                    s->warm_per_lane ? "per-lane kernel until cooling, tile kernel after" : "tile kernel");
    }
    float *dX = nullptr, *dY = nullptr;
    HIP_TRY(hipMalloc(&dX, n_ends * sizeof(float)));
    HIP_TRY(hipMalloc(&dY, n_ends * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(dX, X, n_ends * sizeof(float), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(dY, Y, n_ends * sizeof(float), hipMemcpyHostToDevice, s->stream));
    const int grid = (int)std::min<uint64_t>((n_ends + 255) / 256, 2048);
    if (s->fmt == pgsgd::kFmtQ32)
        hipLaunchKernelGGL(pgsgd::pack_coords<pgsgd::kFmtQ32>, dim3(grid), dim3(256), 0, s->stream, dX, dY, n_ends, s->dc.xf, s->d_coords);
    else
        hipLaunchKernelGGL(pgsgd::pack_coords<pgsgd::kFmtF32>, dim3(grid), dim3(256), 0, s->stream, dX, dY, n_ends, s->dc.xf, s->d_coords);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    (void)hipFree(dX);
    (void)hipFree(dY);
    s->snap_stale = true;
    // a new layout starts here: nothing of the previous one's run state may reach it — the frame-guard flag (it would
    // double the fresh frame at the first sync), the far-pull counts behind the learning-rate cap of far terms, and the
    // gentle start of the far pulls (tile_far_relax counts iterations of THIS layout; tile_epoch goes on as the seed)
    HIP_TRY(hipMemset(s->d_delta_max, 0, 2 * sizeof(unsigned int)));
    if (s->d_far) HIP_TRY(hipMemset(s->d_far, 0, 4 * sizeof(unsigned long long)));
    s->far_launches[0] = s->far_launches[1] = 0;
    s->relax_iter = 0;
    return PGSGD_OK;
}

// Reading coordinates.  The download entry points return the layout: they first deliver what the last tile launch left
// in the outbox (pgsgd_session_flush; nothing to do for per-lane sessions).  The peek entry points read the words as they
// are — what a snapshot between iterations (path_sgd_layout.cpp:379-408) sees: without the far pulls still waiting.
static int read_coords(pgsgd_session* s, float* X, float* Y, bool flush) {
    pgsgd::clear_error();
    if (!s || !X || !Y) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    if (flush) {
        const int rc = pgsgd_session_flush(s);
        if (rc) return rc;
    }
    const uint64_t n_ends = 2 * s->n_nodes;
    float *dX = nullptr, *dY = nullptr;
    HIP_TRY(hipMalloc(&dX, n_ends * sizeof(float)));
    HIP_TRY(hipMalloc(&dY, n_ends * sizeof(float)));
    const int grid = (int)std::min<uint64_t>((n_ends + 255) / 256, 2048);
    if (s->fmt == pgsgd::kFmtQ32)
        hipLaunchKernelGGL(pgsgd::unpack_coords<pgsgd::kFmtQ32>, dim3(grid), dim3(256), 0, s->stream, s->d_coords, n_ends, s->dc.xf, dX, dY);
    else
        hipLaunchKernelGGL(pgsgd::unpack_coords<pgsgd::kFmtF32>, dim3(grid), dim3(256), 0, s->stream, s->d_coords, n_ends, s->dc.xf, dX, dY);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(X, dX, n_ends * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(Y, dY, n_ends * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    (void)hipFree(dX);
    (void)hipFree(dY);
    return PGSGD_OK;
}

extern "C" int pgsgd_session_download_coords(pgsgd_session* s, float* X, float* Y) { return read_coords(s, X, Y, true); }
extern "C" int pgsgd_session_peek_coords(pgsgd_session* s, float* X, float* Y) { return read_coords(s, X, Y, false); }

extern "C" void* pgsgd_session_coords_ptr(pgsgd_session* s) { return s ? (void*)s->d_coords : nullptr; }

static int read_words(pgsgd_session* s, uint64_t* words, bool flush) {
    pgsgd::clear_error();
    if (!s || !words) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    if (flush) {
        const int rc = pgsgd_session_flush(s);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(words, s->d_coords, s->n_nodes * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PGSGD_OK;
}

// Coordinates in double precision: exactly x_off + q / quanta_per_bp for the fixed-point format (fp32 cannot hold
// 1/16 bp at genome-scale coordinates: its spacing is 2-16 bp from 3e7 bp on), the fp32 words widened otherwise.
extern "C" int pgsgd_session_download_words(pgsgd_session* s, uint64_t* words) { return read_words(s, words, true); }
extern "C" int pgsgd_session_peek_words(pgsgd_session* s, uint64_t* words) { return read_words(s, words, false); }

static int read_coords_f64(pgsgd_session* s, double* X, double* Y, bool flush) {
    pgsgd::clear_error();
    if (!s || !X || !Y) return PGSGD_E_INVALID;
    std::vector<uint64_t> w(2 * s->n_nodes);
    const int rc = read_words(s, w.data(), flush);
    if (rc) return rc;
    const pgsgd::Xform& xf = s->dc.xf;
    for (uint64_t i = 0; i < w.size(); ++i) {
        if (s->fmt == pgsgd::kFmtQ32) {
            X[i] = xf.x_off + (double)(uint32_t)w[i] * (double)xf.inv_scale;
            Y[i] = xf.y_off + (double)(uint32_t)(w[i] >> 32) * (double)xf.inv_scale;
        } else {
            uint32_t lo = (uint32_t)w[i], hi = (uint32_t)(w[i] >> 32);
            float fx, fy;
            memcpy(&fx, &lo, 4);
            memcpy(&fy, &hi, 4);
            X[i] = fx;
            Y[i] = fy;
        }
    }
    return PGSGD_OK;
}
extern "C" int pgsgd_session_download_coords_f64(pgsgd_session* s, double* X, double* Y) { return read_coords_f64(s, X, Y, true); }
extern "C" int pgsgd_session_peek_coords_f64(pgsgd_session* s, double* X, double* Y) { return read_coords_f64(s, X, Y, false); }

extern "C" int pgsgd_session_coord_format(const pgsgd_session* s, int* fixed_point, double* x_off, double* y_off, double* quanta_per_bp) {
    if (!s) return PGSGD_E_INVALID;
    if (fixed_point) *fixed_point = s->fmt == pgsgd::kFmtQ32 ? 1 : 0;
    if (x_off) *x_off = s->dc.xf.x_off;
    if (y_off) *y_off = s->dc.xf.y_off;
    if (quanta_per_bp) *quanta_per_bp = (double)s->dc.xf.scale;
    return PGSGD_OK;
}

extern "C" void* pgsgd_session_stream(pgsgd_session* s) { return s ? (void*)s->stream : nullptr; }

extern "C" int pgsgd_session_set_stream(pgsgd_session* s, void* hip_stream) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    int rc = collect_events(s);
    if (rc) return rc;
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    s->stream = (hipStream_t)hip_stream;
    s->own_stream = false;
    return PGSGD_OK;
}

// lanes of the kernel that runs the cooling half: the tile kernel's resident lanes, or the per-lane streams
extern "C" uint32_t pgsgd_session_n_streams(const pgsgd_session* s) { return !s ? 0 : s->tiled ? s->tile_lanes : s->n_streams; }

// parity hooks for the tile kernel
extern "C" int64_t pgsgd_session_tile_table(const pgsgd_session* s, uint64_t* t0, uint64_t* cum, uint32_t* n, uint32_t* path, uint64_t capacity,
                                            uint64_t* steps_total) {
    if (!s) return PGSGD_E_INVALID;
    if (steps_total) *steps_total = s->tile_steps_total;
    const uint64_t cnt = s->h_tiles.size();
    for (uint64_t i = 0; i < cnt && i < capacity; ++i) {
        if (t0) t0[i] = s->h_tiles[i].t0;
        if (cum) cum[i] = s->h_tiles[i].cum;
        if (n) n[i] = s->h_tiles[i].n;
        if (path) path[i] = s->h_tiles[i].path;
    }
    return (int64_t)cnt;
}

// lanes that work on each tile at once (the tile's hot-node cap, at most the workgroup size): lane l draws the tile's
// terms l, l + lanes, ... from its own stream
extern "C" int64_t pgsgd_session_tile_lanes(const pgsgd_session* s, uint32_t* lanes, uint64_t capacity) {
    if (!s) return PGSGD_E_INVALID;
    const uint64_t cnt = s->h_tiles.size();
    for (uint64_t i = 0; i < cnt && i < capacity; ++i)
        if (lanes) lanes[i] = std::min<uint32_t>(s->h_tiles[i].lanes, s->tile_block);
    return (int64_t)cnt;
}

// work items in launch order (the first *n_first belong to the launch of the even regions, the rest to the odd ones)
extern "C" int64_t pgsgd_session_tile_items(const pgsgd_session* s, uint32_t* tile_begin, uint32_t* tile_end, uint32_t* win0, uint32_t* local,
                                            uint64_t capacity, uint64_t* n_first) {
    if (!s) return PGSGD_E_INVALID;
    const bool split = s->tile_split > 1 && s->shard_world == 1;  // the list the session launches
    const std::vector<pgsgd::WorkItem>& items = split ? s->h_items_split : s->h_items;
    if (n_first) *n_first = split ? s->n_items_split[0] : s->n_items[0];
    const uint64_t cnt = items.size();
    for (uint64_t i = 0; i < cnt && i < capacity; ++i) {
        if (tile_begin) tile_begin[i] = items[i].tile_begin;
        if (tile_end) tile_end[i] = items[i].tile_end;
        if (win0) win0[i] = items[i].win0;
        if (local) local[i] = items[i].local & pgsgd::kItemLocal;
    }
    return (int64_t)cnt;
}

// replay the terms tile `tile` draws in iteration `epoch` (1-based, as counted by the session) of n_terms terms
extern "C" int64_t pgsgd_session_trace_tile_terms(pgsgd_session* s, uint64_t tile, int cooling, uint64_t epoch, uint64_t n_terms,
                                                  uint64_t* out, uint64_t capacity_terms) {
    pgsgd::clear_error();
    if (!s || !out || !s->tiled || tile >= s->h_tiles.size()) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    const pgsgd::Tile t = s->h_tiles[tile];
    auto md = [](uint64_t a, uint64_t b, uint64_t c) { return (uint64_t)(((unsigned __int128)a * b) / c); };
    const uint64_t term_begin = md(t.cum, n_terms, s->tile_steps_total), term_end = md(t.cum + t.n, n_terms, s->tile_steps_total);
    const uint64_t cnt = term_end - term_begin;
    if (cnt > capacity_terms) { set_error("tile has %llu terms, buffer holds %llu", (unsigned long long)cnt, (unsigned long long)capacity_terms); return PGSGD_E_INVALID; }
    if (cnt == 0) return 0;
    uint64_t* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, cnt * 4 * sizeof(uint64_t)));
    pgsgd::IterArgs a;
    a.n_terms = n_terms;
    a.eta = 0.0f;
    a.cooling = cooling ? 1u : 0u;
    a.epoch = epoch;
    const uint32_t lanes = std::min<uint32_t>(t.lanes, s->tile_block);
    hipLaunchKernelGGL(pgsgd::tile_trace_kernel, dim3((lanes + pgsgd::kTileBlock - 1) / pgsgd::kTileBlock), dim3(pgsgd::kTileBlock), 0, s->stream, s->dc, t,
                       tile, lanes, term_begin, term_end, a, s->tile_seed_base, s->tile_pair_uniform, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, cnt * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) { set_error("tile trace failed: %s", hipGetErrorString(e)); return PGSGD_E_HIP; }
    return (int64_t)cnt;
}

// Multi-GPU: this session is rank `rank` of `world`.  by_region 0: every work item's tiles are dealt out to the ranks
// (tile shard); 1: every world-th work item (node region) with all its tiles belongs to this rank — the ranks' private
// windows are then disjoint, which keeps the one-GPU layout quality, but a launch has only (work items / world) items to
// fill the GPU with; -1: by region when that still leaves a launch a thousand windows per rank (BASELINE config 5 at
// G = 8) — with the exact exchange when the coordinates are fixed-point (the default format), with the merge rule otherwise —
// and by tile when it does not; 2: by region with the EXACT exchange — pgsgd_session_iteration_part(c, 2) then
// runs colour c alone and the caller exchanges integer deltas after each colour (pgsgd_session_exchange_exact_begin / _end):
// the ranks' coordinates are then, bit for bit, what one GPU computes.  Returns 0: not a tiled session (the caller shards
// the term count), 1: sharded by tile, 2: by region, 3: by region, exact.
// A sharded session's tile streams are keyed on (seed, iteration, tile, lane) only: every tile is run by exactly one
// rank, and a rank-dependent stream_offset (which the per-lane streams of the ranks need to differ) would make rank r's
// tile t draw the stream of rank 0's tile t + 1024 r.
// What a multi-GPU driver adds to its sessions' flags before it creates them (include/pgsgd.h).  Measured at config 4 with virtual
// ranks (profiles/r06/virtual_ranks_exact_small_regions_config4.jsonl), exact shard, kernels per rank and schedule at G = 1 / 2 / 4 / 8:
// R = 256: 174.5 / 138.6 / 122.0 / 113.3 ms; R = 192: 168 / - / 92.7 / 88.6; R = 128: 190.3 / 108.4 / 85.5 / 78.9 (R = 64 diverges); the exact
// figure of the final layout is the one-GPU one everywhere (0.2049-0.2053).  By tile: 178.4 / 99.5 / 59.6 / 41.4 ms at +7.6 / +13.9 / +20.5 %.
extern "C" uint32_t pgsgd_shard_flags(uint64_t n_nodes, uint32_t world, uint32_t flags) {
    if (world < 2 || (flags & (PGSGD_FLAG_NO_TILES | PGSGD_FLAG_SHARD_TILES | PGSGD_FLAG_FP32_ATOMICS | PGSGD_FLAG_HOGWILD_STORES | PGSGD_FLAG_REGION_128))) return 0;
    const uint64_t w256 = (n_nodes + 511) / 512, w128 = (n_nodes + 255) / 256;   // windows per colour
    if (w256 / world >= PGSGD_SHARD_FULL_WINDOWS) return 0;   // 256-node windows fill every device
    return w128 / world >= PGSGD_SHARD_MIN_WINDOWS ? PGSGD_FLAG_REGION_128 : 0;
}

extern "C" int pgsgd_session_set_shard(pgsgd_session* s, uint32_t rank, uint32_t world, int by_region) {
    pgsgd::clear_error();
    if (!s || world == 0 || rank >= world || by_region > 2) return PGSGD_E_INVALID;
    // -1: the one rule both drivers use (pgsgd_multi.cpp and odgi_amd/distributed.py pass -1 and take what comes back).  It counts
    // WINDOWS (n_items: the session's unsplit work items per colour), not the parts a one-GPU session cuts them into.
    if (by_region < 0) {
        const uint64_t per_rank = s->tiled ? std::min(s->n_items[0], s->n_items[1]) / world : 0;
        if (!s->tiled || (s->params.flags & PGSGD_FLAG_SHARD_TILES)) by_region = 0;
        else if (s->fmt == pgsgd::kFmtQ32) by_region = per_rank >= PGSGD_SHARD_MIN_WINDOWS ? 2 : 0;   // the ranks then hold one GPU's layout bit for bit
        else by_region = per_rank >= PGSGD_SHARD_FULL_WINDOWS ? 1 : 0;
    }
    if (by_region == 2 && (!s->tiled || s->fmt != pgsgd::kFmtQ32)) { set_error("the exact exchange needs a tiled session with fixed-point coordinates"); return PGSGD_E_UNSUPPORTED; }
    if (world > 1 || by_region == 2) {  // a sharded session exchanges after every launch: its far pulls are delivered in front of the next one
        if (pulls_waiting(s)) { const int rc = pgsgd_session_flush(s); if (rc) return rc; }
        s->async_drain = false;
    }
    s->shard_rank = by_region ? rank : 0;
    s->shard_world = by_region ? world : 1;
    s->tshard_rank = by_region ? 0 : rank;
    s->tshard_world = by_region ? 1 : world;
    s->exact_colours = by_region == 2;
    s->tile_seed_base = world > 1 ? s->params.seed : s->params.seed + (uint64_t)s->params.stream_offset;
    if (s->tiled) {
        HIP_TRY(hipStreamSynchronize(s->stream));  // (no launch still reads the lists that are replaced)
        const int rc = build_launch_items(s);
        if (rc) return rc;
    }
    return s->tiled ? (by_region == 2 ? 3 : by_region ? 2 : 1) : 0;
}

extern "C" int pgsgd_session_tile_info(const pgsgd_session* s, uint64_t* n_tiles, uint64_t* n_nonlocal_tiles, uint64_t* n_work_items,
                                       uint32_t* region_nodes, uint32_t* tile_steps) {
    if (!s) return PGSGD_E_INVALID;
    if (n_tiles) *n_tiles = s->tiled ? s->n_tiles : 0;
    if (n_nonlocal_tiles) *n_nonlocal_tiles = s->n_nonlocal_tiles;
    if (n_work_items) *n_work_items = (uint64_t)s->n_items[0] + s->n_items[1];  // (windows and window-less tiles; pgsgd_session_tile_parts: the items launched)
    if (region_nodes) *region_nodes = s->region;
    if (tile_steps) *tile_steps = s->tile_steps;
    return s->tiled ? (s->warm_per_lane ? 2 : 1) : 0;
}

// Parts a window's tiles are cut into (1: whole windows) and the work items the session's two launches take together.
extern "C" int pgsgd_session_tile_parts(const pgsgd_session* s, uint64_t* n_launch_items) {
    if (!s) return PGSGD_E_INVALID;
    if (n_launch_items) *n_launch_items = s->tile_split > 1 ? (uint64_t)s->n_items_split[0] + s->n_items_split[1] : (uint64_t)s->n_items[0] + s->n_items[1];
    return s->tiled ? (int)s->tile_split : 0;
}

// 1: the work items of a colour are in node order, one run per XCD (sessions whose windows are cut into parts); 0: one run by size.
extern "C" int pgsgd_session_tile_order(const pgsgd_session* s) {
    if (!s) return PGSGD_E_INVALID;
    return s->tiled && !s->items_one_run ? 1 : 0;
}

extern "C" int pgsgd_session_tile_math(const pgsgd_session* s) {
    if (!s) return PGSGD_E_INVALID;
    return s->tile_math;
}

namespace pgsgd {
__global__ void tile_displacement_probe_kernel(uint64_t n, float eta, const float* d, const float* dx, const float* dy, const float* cap, float* out_fast, float* out_exact) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float rx, ry, ad;
    tile_displacement<kMathFast>(eta, d[i], dx[i], dy[i], cap[i], rx, ry, ad);
    out_fast[3 * i] = rx; out_fast[3 * i + 1] = ry; out_fast[3 * i + 2] = ad;
    tile_displacement<kMathExact>(eta, d[i], dx[i], dy[i], cap[i], rx, ry, ad);
    out_exact[3 * i] = rx; out_exact[3 * i + 1] = ry; out_exact[3 * i + 2] = ad;
}
}  // namespace pgsgd

extern "C" int pgsgd_debug_tile_displacement(int device, uint64_t n, float eta, const float* d, const float* dx, const float* dy, const float* cap,
                                             float* out_fast, float* out_exact) {
    pgsgd::clear_error();
    if (!n || !d || !dx || !dy || !cap || !out_fast || !out_exact) return PGSGD_E_INVALID;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { set_error("no HIP device"); return PGSGD_E_NODEVICE; }
    HIP_TRY(hipSetDevice(device));
    float* buf = nullptr;
    HIP_TRY(hipMalloc(&buf, 10 * n * sizeof(float)));
    int rc = PGSGD_OK;
    const float* in[4] = {d, dx, dy, cap};
    for (int k = 0; k < 4 && rc == PGSGD_OK; ++k)
        if (hipMemcpy(buf + k * n, in[k], n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = PGSGD_E_HIP;
    if (rc == PGSGD_OK) {
        hipLaunchKernelGGL(pgsgd::tile_displacement_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, eta, buf, buf + n, buf + 2 * n, buf + 3 * n,
                           buf + 4 * n, buf + 7 * n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = PGSGD_E_HIP;
    }
    if (rc == PGSGD_OK && (hipMemcpy(out_fast, buf + 4 * n, 3 * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(out_exact, buf + 7 * n, 3 * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)) rc = PGSGD_E_HIP;
    (void)hipFree(buf);
    if (rc) set_error("tile displacement probe failed on the device");
    return rc;
}

extern "C" int pgsgd_session_split_info(const pgsgd_session* s, uint32_t* apply_lanes) {
    if (!s) return PGSGD_E_INVALID;
    if (apply_lanes) *apply_lanes = s->split ? s->apply_lanes : 0;
    return s->split ? 1 : 0;
}

// The outbox's message pool, sized for calls of n_terms terms: one launch (one colour of one part) sends at most one
// message per term of its tiles (two for window-less tiles), i.e. about 0.5 * n_terms / n_parts when every partner is
// far; shares go to the buckets in proportion to the path steps on their nodes (where partners land), plus one open
// chunk per resident workgroup.  A bucket that still runs out falls back to direct atomics (outbox_push).
static int ensure_outbox(pgsgd_session* s, uint64_t n_terms, uint32_t n_parts) {
    const uint64_t per_call = n_terms / std::max<uint32_t>(1, n_parts * s->tile_substeps * s->tshard_world * s->shard_world) + 1;
    if (s->ob.pool && per_call <= s->ob_budget_terms) return PGSGD_OK;
    if (pulls_waiting(s)) {  // the pool is about to be replaced: deliver what the last launch left in it first
        const int rc = pgsgd_session_flush(s);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->drain_stream) HIP_TRY(hipStreamSynchronize(s->drain_stream));
    if (s->ob.pool) { (void)hipFree(s->ob.pool); s->ob.pool = nullptr; }
    if (s->ob.fill) { (void)hipFree(s->ob.fill); s->ob.fill = nullptr; }
    if (s->ob1.pool) { (void)hipFree(s->ob1.pool); s->ob1.pool = nullptr; }
    if (s->ob1.fill) { (void)hipFree(s->ob1.fill); s->ob1.fill = nullptr; }
    const uint32_t B = s->ob.n_buckets;
    double frac = 1.3 * s->ob_msgs_per_term;  // every partner far, and slack for buckets that draw more than their share
    uint64_t open_chunks = (uint64_t)pgsgd::kObGroup * (s->tile_grid + 4);  // every resident workgroup may hold one partly filled group of chunks per bucket
    if (const char* e = pgsgd::debug_env("PGSGD_OUTBOX_FRACTION")) {  // test knob: a pool this small overflows into direct atomics
        frac = std::max(0.0, atof(e));
        open_chunks = pgsgd::kObGroup;
    }
    uint64_t steps_total = 0;
    for (uint64_t v : s->ob_bucket_steps) steps_total += v;
    std::vector<uint32_t> chunk0(B), cap(B);
    uint64_t total = 0;
    for (uint32_t b = 0; b < B; ++b) {
        const double share = steps_total ? (double)s->ob_bucket_steps[b] / (double)steps_total : 1.0 / B;
        uint64_t c = (uint64_t)std::ceil(frac * (double)per_call * share / (double)pgsgd::kObChunk) + open_chunks;
        c = std::min<uint64_t>(c, pgsgd::kObOverflow - 1);  // chunk ids must fit the workgroups' LDS line words
        if ((total + c) * pgsgd::kObLinesPerChunk > 0xfffffffeull) { set_error("outbox pool exceeds 2^32 lines"); return PGSGD_E_UNSUPPORTED; }
        if (total + c > 0xffffffffull) { set_error("outbox pool exceeds 2^32 chunks"); return PGSGD_E_UNSUPPORTED; }
        chunk0[b] = (uint32_t)total;
        cap[b] = (uint32_t)c;
        total += c;
    }
    hipError_t e = hipMalloc(&s->ob.pool, total * pgsgd::kObChunk * sizeof(unsigned long long));
    if (e != hipSuccess) {
        set_error("outbox pool of %.1f GB: %s", (double)total * pgsgd::kObChunk * 8 / 1e9, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP;
    }
    HIP_TRY(hipMalloc(&s->ob.fill, total * sizeof(uint32_t)));
    HIP_TRY(hipMemset(s->ob.fill, 0, total * sizeof(uint32_t)));
    if (s->async_drain) {  // one pool per region colour: colour c's launch fills its pool while the other colour's is drained
        e = hipMalloc(&s->ob1.pool, total * pgsgd::kObChunk * sizeof(unsigned long long));
        if (e != hipSuccess) {
            set_error("second outbox pool of %.1f GB: %s", (double)total * pgsgd::kObChunk * 8 / 1e9, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP;
        }
        HIP_TRY(hipMalloc(&s->ob1.fill, total * sizeof(uint32_t)));
        HIP_TRY(hipMemset(s->ob1.fill, 0, total * sizeof(uint32_t)));
    }
    if (s->async_drain && !s->drain_stream) {
        const size_t n_ends = 2 * (size_t)s->n_nodes;
        HIP_TRY(hipMalloc(&s->ob1.spill, n_ends * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(s->ob1.spill, 0, n_ends * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&s->ob1.next, B * sizeof(uint32_t)));
        HIP_TRY(hipMemset(s->ob1.next, 0, B * sizeof(uint32_t)));
        for (int c = 0; c < 2; ++c) {
            HIP_TRY(hipMalloc(&s->d_pend[c], (size_t)s->ob_slices * n_ends * sizeof(uint64_t)));
            HIP_TRY(hipEventCreateWithFlags(&s->ev_launch[c], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&s->ev_drain[c], hipEventDisableTiming));
        }
        HIP_TRY(hipStreamCreateWithFlags(&s->drain_stream, hipStreamNonBlocking));
    }
    if (s->ob_slices > 1 && !s->d_ob_partial) {  // (with the pool, where a failure is the create path's to report — not on the iteration path)
        hipError_t e2 = hipMalloc(&s->d_ob_partial, (size_t)s->ob_slices * 2 * s->n_nodes * sizeof(uint64_t));
        if (e2 != hipSuccess) { set_error("drain slices' sums (%u x %.1f MB): %s", s->ob_slices, 16.0 * s->n_nodes / 1e6, hipGetErrorString(e2)); return e2 == hipErrorOutOfMemory ? PGSGD_E_NOMEM : PGSGD_E_HIP; }
    }
    if (!s->d_ob_chunk0) {
        HIP_TRY(hipMalloc(&s->d_ob_overflow, sizeof(unsigned long long)));
        HIP_TRY(hipMemset(s->d_ob_overflow, 0, sizeof(unsigned long long)));
        s->ob.overflow = s->d_ob_overflow;
        HIP_TRY(hipMalloc(&s->d_ob_spill, 2 * s->n_nodes * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(s->d_ob_spill, 0, 2 * s->n_nodes * sizeof(unsigned long long)));
        s->ob.spill = s->d_ob_spill;
        HIP_TRY(hipMalloc(&s->d_ob_chunk0, B * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&s->d_ob_cap, B * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&s->ob.next, B * sizeof(uint32_t)));
        HIP_TRY(hipMemset(s->ob.next, 0, B * sizeof(uint32_t)));
    }
    HIP_TRY(hipMemcpy(s->d_ob_chunk0, chunk0.data(), B * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_ob_cap, cap.data(), B * sizeof(uint32_t), hipMemcpyHostToDevice));
    s->ob.chunk0 = s->d_ob_chunk0;
    s->ob.cap = s->d_ob_cap;
    if (s->async_drain) {  // colour 1's outbox: the shared description with its own pool, chunk counters and spill words
        pgsgd::Outbox o = s->ob;
        o.pool = s->ob1.pool; o.fill = s->ob1.fill; o.next = s->ob1.next; o.spill = s->ob1.spill;
        s->ob1 = o;
    }
    s->ob_budget_terms = per_call;
    s->ob_total_chunks = total;
    if (s->params.progress)
        fprintf(stderr, "[odgi::path_linear_sgd_layout] far-update outbox: %u buckets of %u node ends, %.2f GB message pool\n", B, 1u << s->ob.shift,
                (double)total * pgsgd::kObChunk * 8 / 1e9);
    return PGSGD_OK;
}

// Deliver the far pulls waiting in the outbox (if any): far_drain_kernel adds every bucket's messages up and moves the
// node ends; then the chunk counters, the work queues and `far_next` — the far-pull counter the next launch writes —
// start from zero.
static int drain_outbox(pgsgd_session* s, unsigned long long* far_next) {
    if (!s->ob_pending) {
        if (s->queues_dirty && far_next) {  // (the launches since the last reset drained beside each other: pgsgd_session_flush delivered them, the counters are as they left them)
            hipLaunchKernelGGL(pgsgd::outbox_reset_kernel, dim3(1), dim3(256), 0, s->stream, s->ob.next, 0u, s->d_queue, far_next);
            s->n_kernels++;
            HIP_TRY(hipGetLastError());
            s->queues_dirty = false;
        }
        return PGSGD_OK;
    }
    // (a bucket's parts run on one XCD: the grid is the buckets rounded up to a multiple of the 8 XCDs, times the parts)
    const uint64_t n_ends = 2 * s->n_nodes;
    hipLaunchKernelGGL(pgsgd::far_drain_kernel, dim3(((((s->ob.n_buckets + pgsgd::kItemQueues - 1) / pgsgd::kItemQueues) * pgsgd::kItemQueues) << (s->ob.shift - s->ob_part_shift)) * s->ob_slices), dim3(1024),
                       sizeof(uint64_t) << s->ob_part_shift, s->stream, s->ob, s->d_coords, n_ends, s->ob_part_shift, s->dc.frame_flag, s->ob_slices, s->ob_slices > 1 ? s->d_ob_partial : nullptr);
    s->n_kernels++;
    HIP_TRY(hipGetLastError());
    if (s->ob_slices > 1) {   // the slices' sums reach the coordinates; the same kernel zeroes the chunk counters, the work queues and far_next
        hipLaunchKernelGGL(pgsgd::far_combine_kernel, dim3((unsigned)std::min<uint64_t>((n_ends + 255) / 256, 2048)), dim3(256), 0, s->stream, s->d_coords, s->d_ob_partial, n_ends,
                           s->ob_slices, s->dc.frame_flag, s->ob.next, s->ob.n_buckets, s->d_queue, s->d_queue + pgsgd::kItemQueues, s->d_queue + 2 * pgsgd::kItemQueues, far_next);
    } else {
        hipLaunchKernelGGL(pgsgd::outbox_reset_kernel, dim3((s->ob.n_buckets + 255) / 256), dim3(256), 0, s->stream, s->ob.next, s->ob.n_buckets,
                           s->d_queue, far_next);
    }
    s->n_kernels++;
    HIP_TRY(hipGetLastError());
    s->ob_pending = false;
    s->queues_dirty = false;
    return PGSGD_OK;
}

// ---- the drain beside the next launch (pgsgd_session::async_drain) ----
// Right after colour c's tile launch (already enqueued on the launch stream): far_drain_kernel sums the launch's messages into
// d_pend[c] on the DRAIN stream, beside whatever the launch stream does next — the other colour's launch, which writes the other
// outbox and the coordinates; the drain touches neither.
static int enqueue_drain(pgsgd_session* s, int colour) {
    HIP_TRY(hipEventRecord(s->ev_launch[colour], s->stream));
    HIP_TRY(hipStreamWaitEvent(s->drain_stream, s->ev_launch[colour], 0));
    pgsgd_session::DrainEv de;
    if (!s->free_drain_events.empty()) {
        de = s->free_drain_events.back();
        s->free_drain_events.pop_back();
    } else {
        HIP_TRY(hipEventCreate(&de.a));
        HIP_TRY(hipEventCreate(&de.b));
    }
    const pgsgd::Outbox& ob = colour ? s->ob1 : s->ob;
    const uint64_t n_ends = 2 * s->n_nodes;
    HIP_TRY(hipEventRecord(de.a, s->drain_stream));
    hipLaunchKernelGGL(pgsgd::far_drain_kernel, dim3(((((ob.n_buckets + pgsgd::kItemQueues - 1) / pgsgd::kItemQueues) * pgsgd::kItemQueues) << (ob.shift - s->ob_part_shift)) * s->ob_slices), dim3(1024),
                       sizeof(uint64_t) << s->ob_part_shift, s->drain_stream, ob, s->d_coords, n_ends, s->ob_part_shift, s->dc.frame_flag, s->ob_slices, s->d_pend[colour]);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(de.b, s->drain_stream));
    HIP_TRY(hipEventRecord(s->ev_drain[colour], s->drain_stream));
    s->pending_drain_events.push_back(de);
    s->n_kernels++;
    if (!s->pend_waiting[1 - colour]) s->pend_order = colour;
    s->pend_waiting[colour] = true;
    return PGSGD_OK;
}

// The far pulls colour c's last launch collected reach the coordinates (launch stream; waits for their drain): right before colour
// c's next launch — with for_launch the same kernel zeroes that launch's work queues and the far-pull counter it writes — and for
// whoever needs the coordinates complete (pgsgd_session_flush).
static int deliver_pulls(pgsgd_session* s, int colour, bool for_launch, unsigned long long* far_next) {
    const bool waiting = s->pend_waiting[colour];
    if (!waiting && !for_launch) return PGSGD_OK;
    if (waiting) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_drain[colour], 0));
    const uint64_t n_ends = 2 * s->n_nodes;
    const pgsgd::Outbox& ob = colour ? s->ob1 : s->ob;
    hipLaunchKernelGGL(pgsgd::far_combine_kernel, dim3(waiting ? (unsigned)std::min<uint64_t>((n_ends + 255) / 256, 2048) : 1u), dim3(256), 0, s->stream, s->d_coords,
                       waiting ? s->d_pend[colour] : nullptr, n_ends, s->ob_slices, s->dc.frame_flag, waiting ? ob.next : nullptr, ob.n_buckets,
                       for_launch ? s->d_queue + colour * pgsgd::kItemQueues : nullptr, for_launch && colour == 0 ? s->d_queue + 2 * pgsgd::kItemQueues : nullptr, nullptr,
                       for_launch ? far_next : nullptr);
    HIP_TRY(hipGetLastError());
    s->n_kernels++;
    s->pend_waiting[colour] = false;
    if (for_launch) s->queues_dirty = false;
    return PGSGD_OK;
}

// Apply what is still waiting in the outbox.  A tile launch's far pulls are delivered right before the NEXT launch, so
// after an iteration the coordinates lack the pulls its last launch collected: a run flushes before it reads its result
// (pgsgd_layout_run does), a snapshot between iterations does not.  A no-op for sessions that run the per-lane kernel.
static int flush_for(pgsgd_session* s, int next_colour);
extern "C" int pgsgd_session_flush(pgsgd_session* s) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    if (s->pend_waiting[0] || s->pend_waiting[1]) {  // a session whose drains run beside its launches: the sums of its last two launches wait
        HIP_TRY(hipSetDevice(s->device));
        pgsgd_session::EvSet ev;
        int rc = take_events(s, &ev);
        if (rc) return rc;
        ev.n = 1;
        HIP_TRY(hipEventRecord(ev.e[2], s->stream));
        rc = deliver_pulls(s, s->pend_order, false, nullptr);
        if (rc == PGSGD_OK) rc = deliver_pulls(s, 1 - s->pend_order, false, nullptr);
        HIP_TRY(hipEventRecord(ev.e[3], s->stream));
        s->pending_events.push_back(ev);
        if (rc) return rc;
    }
    // (the counter zeroed is the one the next launch — colour 0 unless it has no items — will write)
    return flush_for(s, s->n_items[0] ? 0 : 1);
}
// next_colour: the colour of the launch that follows (its far-pull counter starts from zero)
static int flush_for(pgsgd_session* s, int next_colour) {
    if (!s->ob_pending) return PGSGD_OK;
    HIP_TRY(hipSetDevice(s->device));
    pgsgd_session::EvSet ev;
    int rc = take_events(s, &ev);
    if (rc) return rc;
    ev.n = 1;
    HIP_TRY(hipEventRecord(ev.e[2], s->stream));
    rc = drain_outbox(s, s->d_far + 2 * next_colour + (s->far_launches[next_colour] & 1u));
    HIP_TRY(hipEventRecord(ev.e[3], s->stream));
    s->pending_events.push_back(ev);
    return rc;
}

extern "C" int pgsgd_session_iteration(pgsgd_session* s, double eta, int cooling, uint64_t n_terms) {
    return pgsgd_session_iteration_part(s, eta, cooling, n_terms, 0, 1);
}

// Part `part` of `n_parts` of an iteration of n_terms terms: the per-lane kernel runs the part-th slice
// of the terms; the tile kernel runs the tiles with index = part (mod n_parts) with their whole share
// (visiting every tile once per iteration keeps the cost of an iteration independent of n_parts).
extern "C" int pgsgd_session_iteration_part(pgsgd_session* s, double eta, int cooling, uint64_t n_terms, uint32_t part, uint32_t n_parts) {
    pgsgd::clear_error();
    if (!s || n_parts == 0 || part >= n_parts) return PGSGD_E_INVALID;
    // a tiled session whose initial layout had no global structure runs the per-lane kernel until cooling
    // (experiment knob PGSGD_TILE_UNTIL=k: the iterations from the k-th on, 0-based, run the per-lane kernel — does the 1e7-node
    // gap come from the late, refining iterations?  profiles/r05/NOTES.md section 7)
    const bool use_tiles = s->tiled && !(s->warm_per_lane && !cooling) && !(s->tile_until && s->relax_iter >= s->tile_until);
    if (!use_tiles && pulls_waiting(s)) {  // (the last tile launch's far pulls must not wait behind per-lane iterations)
        const int rc = pgsgd_session_flush(s);
        if (rc) return rc;
    }
    if (!use_tiles) {
        // a sharded tiled session takes the whole iteration's term count (its tiles' share is applied); while it runs
        // the per-lane kernel (warm phase of a layout without global structure) it applies this device's 1/G of them,
        // whichever way the tiles are sharded
        const uint32_t world = std::max(s->tshard_world, s->shard_world), rank = s->tshard_world > 1 ? s->tshard_rank : s->shard_rank;
        if (s->tiled && world > 1) {
            const uint64_t base = n_terms / world, rem = n_terms % world;
            n_terms = base + (rank < rem ? 1 : 0);
        }
        if (n_parts > 1) {
            const uint64_t base = n_terms / n_parts, rem = n_terms % n_parts;
            n_terms = base + (part < rem ? 1 : 0);
        }
    }
    const bool exact = use_tiles && s->exact_colours;   // part = the colour to run, of n_parts = 2
    if (exact && n_parts != 2) { set_error("a session with the exact exchange runs an iteration as two parts, one per colour"); return PGSGD_E_INVALID; }
    HIP_TRY(hipSetDevice(s->device));
    if (part == 0) {  // iterations started, whichever kernel runs them: tile_epoch seeds the tile streams, relax_iter counts
        s->tile_epoch++;   // the iterations of this layout (pgsgd_session_upload_coords starts it again) for the far pulls' gentle start
        s->relax_iter++;
    }
    if (s->pending_events.size() >= 64) {  // bound the event pool
        HIP_TRY(hipStreamSynchronize(s->stream));
        int rc = collect_events(s);
        if (rc) return rc;
    }
    if (use_tiles) {
        pgsgd::IterArgs a;
        a.n_terms = n_terms;
        a.eta = (float)eta;
        a.cooling = cooling ? 1u : 0u;
        a.epoch = s->tile_epoch;
        const uint32_t tile_parts = exact ? 1u : n_parts;   // the parts the TILES are dealt out to
        int rc = ensure_outbox(s, n_terms, tile_parts);
        if (rc) return rc;
        // the drain beside the next launch: whole iterations of an unsharded session only (pgsgd_session::async_drain)
        const bool async = s->async_drain && s->drain_stream && !exact && n_parts == 1 && s->tile_substeps == 1 && s->shard_world == 1 && s->tshard_world == 1;
        if (s->term0_terms != n_terms) {  // every tile's exact share of this call's terms
            const uint64_t nt = s->n_tiles + 1;
            hipLaunchKernelGGL(pgsgd::tile_terms_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s->stream, s->d_tiles, s->n_tiles,
                               s->tile_steps_total, n_terms, s->d_term0);
            s->n_kernels++;
            HIP_TRY(hipGetLastError());
            s->term0_terms = n_terms;
        }
        // nothing was counted before a colour's first launch: assume three quarters of the partners of this call's
        // terms are far, half of them in each colour's launch
        const double h0 = 0.75 * 0.5 * (double)n_terms / (double)tile_parts / (double)s->tshard_world / (double)(2 * s->n_nodes);
        HIP_TRY(hipMemsetAsync(s->d_delta_max, 0, sizeof(unsigned int), s->stream));  // (the frame-guard flag next to it stays set until the frame is widened)
        // tile subsets: (part, window refresh, tile shard of this device) -> tiles with index = sub (mod n_sub)
        const uint32_t n_sub = tile_parts * s->tile_substeps * s->tshard_world;
        s->n_copies++;
        const int snap_grid = (int)std::min<uint64_t>((s->n_steps + 255) / 256, 256 * 16);
        bool snapshot_taken = exact && part == 1;   // (exact exchange: one pass per iteration, before its first colour — as one GPU does)
        const uint32_t tile_part = exact ? 0u : part;
        for (uint32_t ps = tile_part * s->tile_substeps; ps < (tile_part + 1) * s->tile_substeps; ++ps)
        for (int colour = 0; colour < 2; ++colour) {
            const uint32_t sub = ps * s->tshard_world + s->tshard_rank;
            if (!s->n_items[colour] || (exact && colour != (int)part)) continue;
            pgsgd_session::EvSet ev;
            rc = take_events(s, &ev);
            if (rc) return rc;
            ev.n = 4;
            const uint32_t launches = s->far_launches[colour]++;
            pgsgd::TileArgs ta;
            ta.tiles = s->d_tiles;
            ta.tile_heads = s->d_tile_heads;
            const bool split = s->tile_split > 1;  // the session's own windows, in parts (build_launch_items)
            const uint32_t* n_items_now = split ? s->n_items_split : s->n_items;
            ta.items = (split ? s->d_items_split : s->d_items) + (colour ? n_items_now[0] : 0);
            ta.queue = s->d_queue + colour * pgsgd::kItemQueues;
            memcpy(ta.chunk, s->item_chunk[colour], sizeof ta.chunk);
            ta.item_done = s->d_item_done;
            ta.stamp = ++s->launch_stamp;
            ta.region = s->region;
            ta.tile_steps = s->tile_steps;
            ta.term0 = s->d_term0;
            ta.sub = sub;
            ta.n_sub = n_sub;
            ta.shard_rank = split ? 0 : s->shard_rank;   // (a split list holds this session's windows only)
            ta.shard_world = split ? 1 : s->shard_world;
            // the far-pull count of this colour's previous launch stays on the device: no host round trip
            const bool no_cap = (s->params.flags & PGSGD_FLAG_NO_FAR_CAP) != 0;
            ta.far_mu_cap_first = (no_cap || h0 <= 1.0) ? 1.0f : (float)(1.0 / h0);
            ta.far_from_prev = (launches > 0 && !no_cap) ? 1u : 0u;
            ta.far_relax = s->tile_far_relax_override > 0.0f ? s->tile_far_relax_override : pgsgd::tile_far_relax(s->relax_iter - 1);
            // Sessions whose ranks move the SAME node ends and merge afterwards (tile shard; region shard with the merge rule) count
            // the far pulls of their own share of the terms: G ranks that each deliver a whole projection add up to G of them before
            // the merge rule has averaged anything.  They keep rounds 3-5's half (measured with one projection, eight virtual
            // ranks by tile: stress 3.3x one rank's; profiles/r06/pytest_gpu_call5.log).  The exact exchange is one GPU's arithmetic.
            if (s->tile_far_relax_override <= 0.0f && (s->tshard_world > 1 || (s->shard_world > 1 && !s->exact_colours))) ta.far_relax *= 0.5f;
            if (s->tile_far_relax_override <= 0.0f && (s->tile_far_relax_max > 0.0f || s->tile_far_relax_slope > 0.0f)) {   // (experiment: another ceiling / slope of the ramp)
                const float mx = s->tile_far_relax_max > 0.0f ? s->tile_far_relax_max : 1.0f, sl = s->tile_far_relax_slope > 0.0f ? s->tile_far_relax_slope : 0.2f;
                const uint64_t it = s->relax_iter - 1;
                ta.far_relax = it < 2 ? std::min(mx, sl) : std::min(mx, sl * (float)it);
            }
            ta.far_prev = s->d_far + 2 * colour + ((launches + 1) & 1u);
            ta.far_count = s->d_far + 2 * colour + (launches & 1u);
            ta.recs2 = s->d_recs2;
            ta.seed_base = s->tile_seed_base;
            // (64 per message a lane of the launched instance hands over in one call: two only in the warm windowed instance without locks)
            const bool push2 = s->tile_push == 2 && !a.cooling && !(s->tile_lock_mu > 0.0f);
            ta.wq_threshold = std::min<uint32_t>(s->tile_wq_threshold, push2 ? 128u : 64u);   // (one message per lane and call in a cooling launch, two in a warm one)
            ta.lane_coin = s->tile_lane_coin;
            ta.snap_every = s->tile_snap_every;
            ta.tile_rotate = s->tile_rotate;
            ta.lock_mu = s->tile_lock_mu;
            ta.pair_uniform = s->tile_pair_uniform;
            ta.clock_probe = s->d_clock;
            ta.term_count = s->d_clock ? s->d_clock + 10 : nullptr;
            ta.tail_probe = nullptr;
            if (s->tile_tail && s->d_clock) {  // (debug only: two more memsets per launch)
                ta.tail_probe = s->d_clock + 6;
                HIP_TRY(hipMemsetAsync(s->d_clock + 6, 0, 4 * sizeof(unsigned long long), s->stream));
                if (PGSGD_TILE_ABL != 5) HIP_TRY(hipMemsetAsync(s->d_clock + 8, 0xff, sizeof(unsigned long long), s->stream));   // (instance 5 sums phases in the four words)
            }
            ta.ob = async && colour ? s->ob1 : s->ob;
            pgsgd::TileSampler ts;
            ts.zipf_tab = s->d_zipf_tab;
            ts.space = (uint32_t)std::min<uint64_t>(s->params.space, 0xffffffffull);
            ts.alpha_e = s->dc.zc.alpha_e;
            ts.alpha_frac = s->dc.zc.alpha_frac;
            ts.one_plus_half_pow = s->dc.zc.one_plus_half_pow;
            // colour 0's window-less items (tiles of unsorted stretches; usually none) run in a launch of their own
            const uint32_t windowless = colour == 0 ? s->n_windowless : 0;
            ta.n_items = n_items_now[colour] - windowless;
            if (split) memcpy(ta.chunk, s->item_chunk_split[colour], sizeof ta.chunk);  // the runs of the list in parts
            // The far pulls the launch before this one collected are delivered now, right before the windows are staged:
            // an iteration ends with a launch's window-local terms, not with the arrival of a launch's worth of far
            // pulls (each an average of a dozen long-range pulls — noise at the scale of neighbouring nodes until the
            // next launch's local terms have worked on it; tools/cpu_transient.py).
            HIP_TRY(hipEventRecord(ev.e[2], s->stream));
            if (async) {
                // (a session whose drains run beside its launches: what THIS colour's last launch collected arrives now — its drain
                // ran beside the other colour's launch — and the launch's queues and far-pull counter start from zero; what the
                // launch before this one collected arrives now too if that was a warm launch)
                if (s->ob_pending) { rc = drain_outbox(s, nullptr); if (rc) return rc; }   // (a launch from before the mode was on)
                if (s->pend_waiting[1 - colour] && s->pulls_urgent[1 - colour]) { rc = deliver_pulls(s, 1 - colour, false, nullptr); if (rc) return rc; }
                rc = deliver_pulls(s, colour, true, s->d_far + 2 * colour + (launches & 1u));
            } else {
                if (s->pend_waiting[0] || s->pend_waiting[1]) { rc = pgsgd_session_flush(s); if (rc) return rc; }   // (the mode was turned off: set_shard)
                rc = drain_outbox(s, s->d_far + 2 * colour + (launches & 1u));
            }
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev.e[3], s->stream));
            // Partners outside a window are read from the snapshot pieces of the gather records.  A tile rewrites the pieces
            // of its own steps when its terms are done (sgd_tile_kernel), so a session that runs every tile itself needs
            // the pass over all records only when the coordinates changed behind the tile kernel's back (snap_stale:
            // upload, per-lane iterations, a widened frame, a merge with other devices); a sharded session runs a share
            // of the tiles and takes the pass once per call, after the drain.  (Round 2 took it before every launch of a
            // warm iteration to tame the far pulls delivered at the END of an iteration; delivered at the start of the
            // next launch they need no second refresh, and the tiles' own writes give the same curves as a pass per
            // iteration: tools/cpu_transient.py.)
            const bool sharded = s->shard_world > 1 || s->tshard_world > 1 || tile_parts > 1 || s->tile_substeps > 1 || s->snapshot_pass || exact;  // this launch runs a share of the tiles
            ta.recs2_out = sharded ? nullptr : s->d_recs2;
            if (!snapshot_taken && (sharded || s->snap_stale)) {
                hipLaunchKernelGGL(pgsgd::snapshot_kernel, dim3(snap_grid), dim3(256), 0, s->stream, s->d_step_handle, s->d_coords, s->n_steps, s->d_recs2);
                s->n_kernels++;
                HIP_TRY(hipGetLastError());
                snapshot_taken = true;
            }
            HIP_TRY(hipEventRecord(ev.e[0], s->stream));
            if (ta.n_items) {
                // (the warm instance of a session with two messages per lane and call has the longer wave queues: its launches bring the LDS for them)
                hipLaunchKernelGGL(tile_kernel(s->tile_far, s->tile_math, a.cooling != 0, true, s->tile_lock_mu > 0.0f, push2), dim3(s->tile_grid), dim3(s->tile_block),
                                   push2 ? s->tile_lds : s->tile_lds1, s->stream,
                                   s->dc, ta, ts, a);
                s->n_kernels++;
                HIP_TRY(hipGetLastError());
            }
            if (windowless) {
                pgsgd::TileArgs tw = ta;
                tw.shard_rank = s->shard_rank;
                tw.shard_world = s->shard_world;
                tw.clock_probe = nullptr;
                tw.items = ta.items + ta.n_items;
                tw.n_items = windowless;
                tw.wq_threshold = std::min<uint32_t>(s->tile_wq_threshold, 64u);   // (the window-less instance: one message per lane and call)
                tw.queue = s->d_queue + 2 * pgsgd::kItemQueues;
                tw.chunk[0] = 0;
                for (uint32_t q = 1; q <= pgsgd::kItemQueues; ++q) tw.chunk[q] = windowless;  // one run: every workgroup ends up pulling from it
                hipLaunchKernelGGL(tile_kernel(s->tile_far, s->tile_math, a.cooling != 0, false), dim3(s->tile_grid), dim3(s->tile_block), s->tile_lds1, s->stream,
                                   s->dc, tw, ts, a);
                s->n_kernels++;
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipEventRecord(ev.e[1], s->stream));
            if (async) {
                rc = enqueue_drain(s, colour);   // (drain stream: beside whatever this stream does next)
                if (rc) return rc;
                s->pulls_urgent[colour] = s->relax_iter <= (s->async_from ? s->async_from : pgsgd::kFarGentleIterations);   // (experiment knob PGSGD_ASYNC_FROM=k: late from iteration k on, 0-based)
            } else {
                s->ob_pending = true;
            }
            s->queues_dirty = true;
            s->last_colour = colour;
            s->snap_stale = sharded;
            s->pending_events.push_back(ev);
        }
        HIP_TRY(hipMemcpyAsync(s->h_delta_max, s->d_delta_max, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, s->stream));
        s->n_copies++;
        return PGSGD_OK;
    }
    const bool plain = (s->params.flags & PGSGD_FLAG_COORD_LOAD_PLAIN) != 0;
    const uint32_t abl = (s->params.flags >> 8) & 0xfu;
    iter_kernel_t kernel = session_kernel(s, plain, abl);
    if (!kernel) { set_error("no kernel instance for these debug flags"); return PGSGD_E_UNSUPPORTED; }
    pgsgd_session::EvSet ev;
    {
        const int rc = take_events(s, &ev);
        if (rc) return rc;
    }
    ev.n = 2;
    HIP_TRY(hipMemsetAsync(s->d_delta_max, 0, sizeof(unsigned int), s->stream));  // (the frame-guard flag next to it stays set until the frame is widened)
    pgsgd::IterArgs a;
    s->n_copies++;
    a.n_terms = n_terms;
    a.eta = (float)eta;
    a.cooling = cooling ? 1u : 0u;
    a.epoch = 0;
    const uint32_t block = s->n_streams >= (uint32_t)pgsgd::kBlock ? pgsgd::kBlock : ((s->n_streams + 63) / 64) * 64;
    const uint32_t grid = (s->n_streams + block - 1) / block;
    HIP_TRY(hipEventRecord(ev.e[0], s->stream));
    if (s->split) {
        // chunks of whole rounds of the sampler streams, so that a stream's terms are the same however the call is cut
        const uint64_t chunk = std::max<uint64_t>(s->n_streams, (s->split_chunk / s->n_streams) * s->n_streams);
        const uint64_t want = std::min<uint64_t>(n_terms, chunk);
        if (want > s->terms_cap) {
            HIP_TRY(hipStreamSynchronize(s->stream));
            if (s->d_terms) (void)hipFree(s->d_terms);
            s->d_terms = nullptr;
            s->terms_cap = 0;
            HIP_TRY(hipMalloc(&s->d_terms, want * sizeof(uint4)));
            s->terms_cap = want;
        }
        const uint32_t L = s->apply_lanes;
        for (uint64_t t0 = 0; t0 < n_terms; t0 += chunk) {
            a.n_terms = std::min<uint64_t>(chunk, n_terms - t0);
            if (s->pf_lds) hipLaunchKernelGGL((pgsgd::sample_terms_kernel<true>), dim3(grid), dim3(block), s->lds_bytes, s->stream, s->dc, a.cooling, a.n_terms, s->d_terms);
            else hipLaunchKernelGGL((pgsgd::sample_terms_kernel<false>), dim3(grid), dim3(block), 0, s->stream, s->dc, a.cooling, a.n_terms, s->d_terms);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(pgsgd::apply_terms_resident_kernel, dim3(1), dim3(((L + 63) / 64) * 64), s->resident_lds, s->stream, s->dc, a, s->d_terms, L);
            HIP_TRY(hipGetLastError());
            s->n_kernels += 2;
        }
        s->n_kernels--;  // (counted once more below)
    } else {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), s->lds_bytes, s->stream, s->dc, a);
    }
    s->n_kernels++;
    s->snap_stale = true;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.e[1], s->stream));
    s->pending_events.push_back(ev);
    HIP_TRY(hipMemcpyAsync(s->h_delta_max, s->d_delta_max, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, s->stream));
    s->n_copies++;
    return PGSGD_OK;
}

// Widen the fixed-point frame: same centre, twice the span, half the resolution (see in_frame_guard).
extern "C" int pgsgd_session_reframe(pgsgd_session* s) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    if (s->fmt != pgsgd::kFmtQ32) return PGSGD_OK;
    HIP_TRY(hipSetDevice(s->device));
    if (pulls_waiting(s)) {  // messages in the outbox are steps in quanta of the frame they were computed in
        const int rc = pgsgd_session_flush(s);
        if (rc) return rc;
    }
    const uint64_t n_ends = 2 * s->n_nodes;
    const int grid = (int)std::min<uint64_t>((n_ends + 255) / 256, 2048);
    HIP_TRY(hipMemsetAsync(s->d_delta_max + 1, 0, sizeof(unsigned int), s->stream));  // the guard flag: cleared only here
    hipLaunchKernelGGL(pgsgd::reframe_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_coords, n_ends);
    HIP_TRY(hipGetLastError());
    if (s->d_base) {
        hipLaunchKernelGGL(pgsgd::reframe_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_base, n_ends);
        HIP_TRY(hipGetLastError());
    }
    pgsgd::Xform& xf = s->dc.xf;
    xf.x_off -= 2147483648.0 * (double)xf.inv_scale;  // centre x_off + 2^31 / scale stays where it is
    xf.y_off -= 2147483648.0 * (double)xf.inv_scale;
    xf.scale *= 0.5f;
    xf.inv_scale *= 2.0f;
    s->h_delta_max[1] = 0;
    s->frame_doublings++;
    s->snap_stale = true;
    if (s->params.progress)
        fprintf(stderr, "\n[odgi::path_linear_sgd_layout] a node end reached the outer quarter of the fixed-point frame: frame doubled (now %g quanta per bp)\n",
                (double)xf.scale);
    return PGSGD_OK;
}

// guard_hit: a kernel of the last iteration saw a coordinate in the outer quarter of the frame and the session has not
// widened the frame yet (a sharded session leaves that to its driver: every rank has to do it in the same iteration)
extern "C" int pgsgd_session_frame_status(const pgsgd_session* s, int* guard_hit, uint32_t* doublings) {
    if (!s) return PGSGD_E_INVALID;
    if (guard_hit) *guard_hit = s->fmt == pgsgd::kFmtQ32 && s->h_delta_max[1] != 0;
    if (doublings) *doublings = s->frame_doublings;
    return PGSGD_OK;
}

extern "C" int pgsgd_session_sync(pgsgd_session* s, double* delta_max) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    int rc = collect_events(s);
    if (rc) return rc;
    if (s->fmt == pgsgd::kFmtQ32 && s->h_delta_max[1] && !s->d_base && s->shard_world == 1 && s->tshard_world == 1) {
        rc = pgsgd_session_reframe(s);  // a session on its own widens its frame itself
        if (rc) return rc;
    }
    if (delta_max) {
        float f;
        memcpy(&f, s->h_delta_max, sizeof f);
        *delta_max = (double)f;
    }
    return PGSGD_OK;
}

extern "C" int pgsgd_session_kernel_time(pgsgd_session* s, double* total_ms, uint64_t* launches, int reset) {
    if (!s) return PGSGD_E_INVALID;
    if (total_ms) *total_ms = s->kernel_ms;
    if (launches) *launches = s->launches;
    if (reset) { s->kernel_ms = 0; s->launches = 0; s->aux_ms[0] = s->aux_ms[1] = 0; s->drain_beside_ms = 0; }
    return PGSGD_OK;
}

extern "C" int64_t pgsgd_session_outbox_overflow(pgsgd_session* s) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    if (!s->d_ob_overflow) return 0;
    HIP_TRY(hipSetDevice(s->device));
    unsigned long long v = 0;
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(&v, s->d_ob_overflow, sizeof v, hipMemcpyDeviceToHost));
    return (int64_t)v;
}

// The shader clock the last tile launch ran at, from the two counters its workgroup 0 read when it started and ended
// (shader cycles against the constant 100 MHz reference).  Blocks until the stream is idle.  0 when no tile launch ran.
extern "C" int pgsgd_session_shader_clock(pgsgd_session* s, double* mhz, double* launch_ms) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    if (mhz) *mhz = 0.0;
    if (launch_ms) *launch_ms = 0.0;
    if (!s->d_clock) return PGSGD_OK;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    unsigned long long v[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(v, s->d_clock, sizeof v, hipMemcpyDeviceToHost));
    if (v[3] > v[1] && v[2] > v[0]) {
        const double us = (double)(v[3] - v[1]) / 100.0;  // 100 MHz ticks
        if (mhz) *mhz = (double)(v[2] - v[0]) / us;
        if (launch_ms) *launch_ms = us / 1e3;
    }
    return PGSGD_OK;
}

// Terms of the tile kernel that went for their ends' locks, and terms that found one taken and did nothing (cumulative).
extern "C" int pgsgd_session_tile_conflicts(pgsgd_session* s, uint64_t* locked, uint64_t* lost) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    if (locked) *locked = 0;
    if (lost) *lost = 0;
    if (!s->d_clock) return PGSGD_OK;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(v, s->d_clock, sizeof v, hipMemcpyDeviceToHost));
    if (locked) *locked = v[4];
    if (lost) *lost = v[5];
    return PGSGD_OK;
}

// Terms the session's tile launches have EXECUTED, counted on the device (every wave adds the population count of the lanes
// that finished a term, trip by trip) — the host's accounting (`terms += min_term_updates`, path_sgd_layout.cpp:133-137,370)
// says what was asked for; this says what ran.  Cumulative since the session was created; 0 for a session without tiles.
extern "C" int pgsgd_session_terms_executed(pgsgd_session* s, uint64_t* terms) {
    pgsgd::clear_error();
    if (!s || !terms) return PGSGD_E_INVALID;
    *terms = 0;
    if (!s->d_clock) return PGSGD_OK;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, s->d_clock + 10, sizeof v, hipMemcpyDeviceToHost));
    *terms = v;
    return PGSGD_OK;
}

// Debug knob PGSGD_TILE_TAIL: of the last windowed tile launch, the share of (workgroups x launch duration) that the
// workgroups were alive for — a persistent workgroup lives until the launch has no work item left for it, so what is
// missing is the launch's tail: slots idle while the last items finish.  0 when the knob is off.
extern "C" int pgsgd_session_tile_tail(pgsgd_session* s, double* alive_fraction, double* launch_ms, uint32_t* workgroups) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    if (alive_fraction) *alive_fraction = 0.0;
    if (launch_ms) *launch_ms = 0.0;
    if (workgroups) *workgroups = 0;
    if (!s->d_clock || !s->tile_tail) return PGSGD_OK;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    unsigned long long v[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(v, s->d_clock + 6, sizeof v, hipMemcpyDeviceToHost));
    if (v[3] && v[1] > v[2]) {
        const double span = (double)(v[1] - v[2]);  // 100 MHz ticks from the first start to the last finish
        if (alive_fraction) *alive_fraction = (double)v[0] / (span * (double)v[3]);
        if (launch_ms) *launch_ms = span / 1e5;
        if (workgroups) *workgroups = (uint32_t)v[3];
    }
    return PGSGD_OK;
}

// Profiling hook: the twelve raw words behind pgsgd_session_shader_clock / _tile_conflicts / _tile_tail / _terms_executed.  With the
// profiling instance 5 of the tile kernel (make -C odgi_amd/csrc ../lib/libpgsgd_x5.so; PGSGD_DEBUG=1 PGSGD_LIB=libpgsgd_x5.so
// PGSGD_TILE_TAIL=1) words 4, 5 (cumulative) and 6..9 (the last launch) are the phases of a workgroup's time: tools/gpu_tile_phases.py.
extern "C" int pgsgd_session_probe_words(pgsgd_session* s, uint64_t out[12]) {
    pgsgd::clear_error();
    if (!s || !out) return PGSGD_E_INVALID;
    for (int i = 0; i < 12; ++i) out[i] = 0;
    if (!s->d_clock) return PGSGD_OK;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(out, s->d_clock, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PGSGD_OK;
}

extern "C" int pgsgd_session_launch_counts(const pgsgd_session* s, uint64_t* kernel_launches, uint64_t* copies) {
    if (!s) return PGSGD_E_INVALID;
    if (kernel_launches) *kernel_launches = s->n_kernels;
    if (copies) *copies = s->n_copies;
    return PGSGD_OK;
}

extern "C" int pgsgd_session_aux_time(pgsgd_session* s, double* snapshot_ms, double* drain_ms) {
    if (!s) return PGSGD_E_INVALID;
    if (snapshot_ms) *snapshot_ms = s->aux_ms[0];
    if (drain_ms) *drain_ms = s->aux_ms[1];
    return PGSGD_OK;
}

// Parity hook: the session's step records [first, first + count) as it built them — {handle, node length, position low, position high}
// per step — whether the positions came with the view or were built on the device (pgsgd_graph_view::step_pos == NULL).
extern "C" int pgsgd_session_read_step_records(pgsgd_session* s, uint64_t first, uint64_t count, uint32_t* out) {
    pgsgd::clear_error();
    if (!s || !out || first > s->n_steps || count > s->n_steps - first) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (count) HIP_TRY(hipMemcpy(out, s->d_recs + first, count * sizeof(uint4), hipMemcpyDeviceToHost));
    return PGSGD_OK;
}

// A session whose drains run beside its launches (pgsgd_session::async_drain): *on = whether this one does, *drain_ms = far_drain_kernel's
// time on the DRAIN stream (HIP events; finished drains only — synchronise first for all of them): NOT on the launch stream's critical
// path.  What the launch stream pays for the far pulls (far_combine_kernel in front of a launch) is the drain_ms of pgsgd_session_aux_time.
extern "C" int pgsgd_session_drain_beside(pgsgd_session* s, int* on, double* drain_ms) {
    if (!s) return PGSGD_E_INVALID;
    if (on) *on = s->async_drain && s->drain_stream ? 1 : 0;
    if (drain_ms) {
        if (s->drain_stream) { HIP_TRY(hipStreamSynchronize(s->drain_stream)); (void)collect_drain_events(s); }
        *drain_ms = s->drain_beside_ms;
    }
    return PGSGD_OK;
}

// How the session's far pulls are summed (far_drain_kernel): workgroups per bucket's node range, and per (bucket, part)'s messages.
extern "C" int pgsgd_session_drain_plan(pgsgd_session* s, uint32_t* parts, uint32_t* slices) {
    if (!s) return PGSGD_E_INVALID;
    if (parts) *parts = s->tiled ? 1u << (s->ob.shift - s->ob_part_shift) : 0u;
    if (slices) *slices = s->tiled ? s->ob_slices : 0u;
    return PGSGD_OK;
}

// ---- multi-GPU exchange (see odgi_amd/distributed.py) ------------------------------------------
extern "C" int pgsgd_session_exchange_mark(pgsgd_session* s) {
    pgsgd::clear_error();
    if (!s) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    if (!s->d_base) HIP_TRY(hipMalloc(&s->d_base, s->n_nodes * 2 * sizeof(uint64_t)));
    HIP_TRY(hipMemcpyAsync(s->d_base, s->d_coords, s->n_nodes * 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, s->stream));
    return PGSGD_OK;
}

extern "C" int pgsgd_session_exchange_begin(pgsgd_session* s, void* device_buf_6N_floats) {
    pgsgd::clear_error();
    if (!s || !device_buf_6N_floats) return PGSGD_E_INVALID;
    if (!s->d_base) {
        int rc = pgsgd_session_exchange_mark(s);
        if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(s->device));
    const uint64_t n_ends = 2 * s->n_nodes;
    const int grid = (int)std::min<uint64_t>((n_ends + 255) / 256, 2048);
    if (s->fmt == pgsgd::kFmtQ32)
        hipLaunchKernelGGL(pgsgd::exchange_prepare_kernel<pgsgd::kFmtQ32>, dim3(grid), dim3(256), 0, s->stream, s->d_coords, s->d_base, n_ends, s->dc.xf, (float*)device_buf_6N_floats);
    else
        hipLaunchKernelGGL(pgsgd::exchange_prepare_kernel<pgsgd::kFmtF32>, dim3(grid), dim3(256), 0, s->stream, s->d_coords, s->d_base, n_ends, s->dc.xf, (float*)device_buf_6N_floats);
    HIP_TRY(hipGetLastError());
    return PGSGD_OK;
}

// exchange_begin with a tail: device_buf holds 6N + 2*world floats; slot 6N + rank = this rank's max |Delta| since the
// last iteration call, slot 6N + world + rank = its frame-guard flag, the other ranks' slots zero — after the SUM
// all-reduce every rank holds all of them, so the stop rule and the frame widening need no collective of their own.
extern "C" int pgsgd_session_exchange_begin_stats(pgsgd_session* s, void* device_buf, uint32_t rank, uint32_t world) {
    if (!s || world == 0 || rank >= world) return PGSGD_E_INVALID;
    const int rc = pgsgd_session_exchange_begin(s, device_buf);
    if (rc) return rc;
    hipLaunchKernelGGL(pgsgd::exchange_stats_kernel, dim3((2 * world + 255) / 256), dim3(256), 0, s->stream, s->d_delta_max, rank, world,
                       (float*)device_buf + 6 * s->n_nodes);
    HIP_TRY(hipGetLastError());
    return PGSGD_OK;
}

extern "C" int pgsgd_session_exchange_end(pgsgd_session* s, const void* device_buf_6N_floats, int world_size) {
    pgsgd::clear_error();
    if (!s || !device_buf_6N_floats || world_size < 1 || !s->d_base) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    const uint64_t n_ends = 2 * s->n_nodes;
    const int grid = (int)std::min<uint64_t>((n_ends + 255) / 256, 2048);
    const float inv_world = 1.0f / (float)world_size;
    if (s->fmt == pgsgd::kFmtQ32)
        hipLaunchKernelGGL(pgsgd::exchange_apply_kernel<pgsgd::kFmtQ32>, dim3(grid), dim3(256), 0, s->stream, s->d_coords, s->d_base, n_ends, s->dc.xf, (const float*)device_buf_6N_floats, inv_world);
    else
        hipLaunchKernelGGL(pgsgd::exchange_apply_kernel<pgsgd::kFmtF32>, dim3(grid), dim3(256), 0, s->stream, s->d_coords, s->d_base, n_ends, s->dc.xf, (const float*)device_buf_6N_floats, inv_world);
    HIP_TRY(hipGetLastError());
    s->snap_stale = true;
    return PGSGD_OK;
}

// The exact exchange of a session sharded by region with pgsgd_session_set_shard(.., 2), after pgsgd_session_iteration_part(c, 2):
// begin delivers the launch's far pulls (so that they are part of what this rank changed), writes coords - base as 64-bit
// integers into device_buf[0 .. 2N) and this rank's slots of the tail [2N .. 2N + 3 world): the launch's far-pull count, the
// bits of its max |Delta|, its frame-guard flag; the caller SUMs the buffer over the ranks as 64-bit integers; end sets
// coords = base = base + sum and hands the launch's far-pull count over ALL ranks to the learning-rate cap of the colour's
// next launch.
extern "C" int pgsgd_session_exchange_exact_begin(pgsgd_session* s, void* device_buf, uint32_t rank, uint32_t world) {
    pgsgd::clear_error();
    if (!s || !device_buf || world == 0 || rank >= world) return PGSGD_E_INVALID;
    if (!s->exact_colours || !s->d_base) { set_error("exchange_exact_begin: set_shard(.., 2) and exchange_mark first"); return PGSGD_E_INVALID; }
    HIP_TRY(hipSetDevice(s->device));
    const int c = s->last_colour < 0 ? 0 : s->last_colour;
    const int other = 1 - c;
    int rc = flush_for(s, s->n_items[other] ? other : c);
    if (rc) return rc;
    const uint64_t n_ends = 2 * s->n_nodes;
    const int grid = (int)std::min<uint64_t>((n_ends + 255) / 256, 2048);
    hipLaunchKernelGGL(pgsgd::exchange_exact_prepare_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_coords, s->d_base, n_ends, (uint64_t*)device_buf);
    const unsigned long long* far = s->far_launches[c] ? s->d_far + 2 * c + ((s->far_launches[c] - 1) & 1u) : nullptr;
    hipLaunchKernelGGL(pgsgd::exchange_exact_tail_kernel, dim3((3 * world + 255) / 256), dim3(256), 0, s->stream, s->d_delta_max, far, rank, world,
                       (uint64_t*)device_buf + n_ends);
    HIP_TRY(hipGetLastError());
    return PGSGD_OK;
}

extern "C" int pgsgd_session_exchange_exact_end(pgsgd_session* s, const void* device_buf, uint32_t world) {
    pgsgd::clear_error();
    if (!s || !device_buf || world == 0 || !s->d_base || !s->exact_colours) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    const int c = s->last_colour < 0 ? 0 : s->last_colour;
    const uint64_t n_ends = 2 * s->n_nodes;
    const int grid = (int)std::min<uint64_t>((n_ends + 255) / 256, 2048);
    unsigned long long* far = s->far_launches[c] ? s->d_far + 2 * c + ((s->far_launches[c] - 1) & 1u) : nullptr;
    hipLaunchKernelGGL(pgsgd::exchange_exact_apply_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_coords, s->d_base, n_ends, (const uint64_t*)device_buf, far, world);
    HIP_TRY(hipGetLastError());
    s->snap_stale = true;
    return PGSGD_OK;
}

extern "C" int pgsgd_session_trace_terms(pgsgd_session* s, int cooling, uint64_t terms_per_stream, uint64_t* out) {
    pgsgd::clear_error();
    if (!s || !out) return PGSGD_E_INVALID;
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = (size_t)terms_per_stream * s->n_streams * 4;
    uint64_t* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, n * sizeof(uint64_t)));
    const uint32_t block = s->n_streams >= (uint32_t)pgsgd::kBlock ? pgsgd::kBlock : ((s->n_streams + 63) / 64) * 64;
    const uint32_t grid = (s->n_streams + block - 1) / block;
    const uint64_t seed_base = s->params.seed + (uint64_t)s->params.stream_offset;
    if (s->pf_lds) hipLaunchKernelGGL((pgsgd::trace_kernel<true>), dim3(grid), dim3(block), s->lds_bytes, s->stream, s->dc, cooling ? 1u : 0u, seed_base, terms_per_stream, d_out);
    else hipLaunchKernelGGL((pgsgd::trace_kernel<false>), dim3(grid), dim3(block), 0, s->stream, s->dc, cooling ? 1u : 0u, seed_base, terms_per_stream, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) { set_error("trace launch failed: %s", hipGetErrorString(e)); return PGSGD_E_HIP; }
    return PGSGD_OK;
}

// ---------------------------------------------------------------------------------------------
// the one-shot run: iteration control of path_sgd_layout.cpp:120-163 with exact iteration lengths
int pgsgd_write_lay_f32(const char* path, uint64_t n_ends, const float* X, const float* Y);

namespace pgsgd {
int write_snapshot(const char* name, uint64_t n_nodes, const float* x, const float* y, const uint32_t* new_rank_of_old) {
    if (!new_rank_of_old) return pgsgd_write_lay_f32(name, 2 * n_nodes, x, y);
    std::vector<float> sx(2 * n_nodes), sy(2 * n_nodes);
    for (uint64_t i = 0; i < n_nodes; ++i) {
        const uint64_t n = new_rank_of_old[i];
        sx[2 * i] = x[2 * n]; sx[2 * i + 1] = x[2 * n + 1];
        sy[2 * i] = y[2 * n]; sy[2 * i + 1] = y[2 * n + 1];
    }
    return pgsgd_write_lay_f32(name, 2 * n_nodes, sx.data(), sy.data());
}
}  // namespace pgsgd

static int layout_run_impl(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, double* Xd, double* Yd, pgsgd_stats* stats);
int pgsgd_layout_run_multi(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, double* Xd, double* Yd, pgsgd_stats* stats, const uint32_t* snapshot_names);

extern "C" int pgsgd_layout_run(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, pgsgd_stats* stats) {
    return layout_run_impl(g, p, X, Y, nullptr, nullptr, stats);
}

// The same run with double-precision coordinates in and out: the initial layout is taken in fp32 (it only seeds
// the SGD), the result comes back at the full resolution of the device's fixed-point words.
extern "C" int pgsgd_layout_run_f64(const pgsgd_graph_view* g, const pgsgd_params* p, double* X, double* Y, pgsgd_stats* stats) {
    pgsgd::clear_error();
    if (stats) memset(stats, 0, sizeof *stats);
    if (!g || !X || !Y) return PGSGD_E_INVALID;
    std::vector<float> xf(2 * g->n_nodes), yf(2 * g->n_nodes);
    for (uint64_t i = 0; i < xf.size(); ++i) { xf[i] = (float)X[i]; yf[i] = (float)Y[i]; }
    return layout_run_impl(g, p, xf.data(), yf.data(), X, Y, stats);
}

namespace pgsgd { double step_rank_disorder(const pgsgd_graph_view* g, const uint32_t* new_rank_of_old, uint32_t reach); }  // pgsgd_host.cpp
static int layout_run_named(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, double* Xd, double* Yd, pgsgd_stats* stats, const uint32_t* snapshot_names);

// The run under node ranks that follow the paths, when the caller's do not (pgsgd_graph_path_order; include/pgsgd.h:
// PGSGD_FLAG_NO_RELABEL): node lengths, step handles and the initial layout are renamed, the run is the usual one, the
// coordinates come back under the caller's names.  Ranks are only names — the sampler draws path STEPS, which keep their
// numbers — but names that follow the paths are what lets the tile kernel keep a run of steps' nodes in one LDS window.
static int layout_run_impl(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, double* Xd, double* Yd, pgsgd_stats* stats) {
    pgsgd::clear_error();
    if (stats) memset(stats, 0, sizeof *stats);
    if (!g || !p || !X || !Y) return PGSGD_E_INVALID;
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    const bool may_rename = !(p->flags & (PGSGD_FLAG_NO_RELABEL | PGSGD_FLAG_NO_TILES)) && g->n_nodes >= 4096 &&
                            g->n_steps >= 2 && g->n_steps < 0xffffffffull;
    if (!may_rename || pgsgd::step_rank_disorder(g, nullptr, 128) <= 0.02) return layout_run_named(g, p, X, Y, Xd, Yd, stats, nullptr);
    pgsgd::PhaseTimer timer;
    const uint64_t N = g->n_nodes, S = g->n_steps;
    std::vector<uint32_t> new_of_old(N);
    double d0 = 0, d1 = 0;
    rc = pgsgd_graph_path_order(g, new_of_old.data(), &d0, &d1);
    if (rc) return rc;
    if (!(d1 <= 0.5 * d0)) {  // the paths themselves do not agree on an order: nothing to gain
        if (p->progress) fprintf(stderr, "[odgi::path_linear_sgd_layout] node ranks do not follow the paths (%.1f %% of steps jump) and no order does (%.1f %%): per-lane kernel\n", 100 * d0, 100 * d1);
        return layout_run_named(g, p, X, Y, Xd, Yd, stats, nullptr);
    }
    if (p->progress) fprintf(stderr, "[odgi::path_linear_sgd_layout] node ranks do not follow the paths (%.1f %% of steps jump): laying the graph out under ranks by path position (%.1f %%)\n", 100 * d0, 100 * d1);
    std::vector<uint32_t> len2(N), handle2(S);
    std::vector<float> X2(2 * N), Y2(2 * N);
    for (uint64_t i = 0; i < N; ++i) {
        const uint64_t n = new_of_old[i];
        len2[n] = g->node_len[i];
        X2[2 * n] = X[2 * i]; X2[2 * n + 1] = X[2 * i + 1];
        Y2[2 * n] = Y[2 * i]; Y2[2 * n + 1] = Y[2 * i + 1];
    }
    const unsigned nt = (unsigned)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (!pgsgd::run_threads(nt, [&](unsigned t) {
            for (uint64_t k = S * t / nt; k < S * (t + 1) / nt; ++k) handle2[k] = (new_of_old[g->step_handle[k] >> 1] << 1) | (g->step_handle[k] & 1u);
        })) { set_error("out of memory while renaming the nodes"); return PGSGD_E_NOMEM; }
    pgsgd_graph_view g2 = *g;
    g2.node_len = len2.data();
    g2.step_handle = handle2.data();
    timer.lap("node ranks by path position");
    std::vector<double> Xd2, Yd2;
    if (Xd) { Xd2.resize(2 * N); Yd2.resize(2 * N); }
    rc = layout_run_named(&g2, p, X2.data(), Y2.data(), Xd ? Xd2.data() : nullptr, Xd ? Yd2.data() : nullptr, stats, new_of_old.data());
    if (rc) return rc;
    for (uint64_t i = 0; i < N; ++i) {
        const uint64_t n = new_of_old[i];
        if (Xd) {
            Xd[2 * i] = Xd2[2 * n]; Xd[2 * i + 1] = Xd2[2 * n + 1];
            Yd[2 * i] = Yd2[2 * n]; Yd[2 * i + 1] = Yd2[2 * n + 1];
        } else {
            X[2 * i] = X2[2 * n]; X[2 * i + 1] = X2[2 * n + 1];
            Y[2 * i] = Y2[2 * n]; Y[2 * i + 1] = Y2[2 * n + 1];
        }
    }
    if (stats) stats->relabeled = 1;
    return PGSGD_OK;
}

// snapshot_names: new_rank_of_old of a run under renamed node ranks (snapshots are written under the caller's), or null
static int layout_run_named(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, double* Xd, double* Yd, pgsgd_stats* stats, const uint32_t* snapshot_names) {
    int rc = PGSGD_OK;
    // path_sgd_layout.cpp:64-74: nothing to do unless some path has more than one step
    bool multi = false;
    for (uint64_t i = 0; i < g->n_paths && !multi; ++i) multi = g->path_first[i + 1] - g->path_first[i] > 1;
    if (!multi) {
        int dev;  // still refuse to "succeed" without a device: the product never runs on the CPU
        return pick_device(p->device, &dev);
    }
    // pgsgd_multi.cpp (PGSGD_MULTI_FORCE, a test knob: a one-device run through the multi-GPU driver and a one-rank RCCL communicator)
    if (p->n_devices > 1 || (p->n_devices == 1 && pgsgd::debug_env("PGSGD_MULTI_FORCE"))) return pgsgd_layout_run_multi(g, p, X, Y, Xd, Yd, stats, snapshot_names);
    const auto t0 = std::chrono::steady_clock::now();
    pgsgd::PhaseTimer timer;
    pgsgd_session* s = nullptr;
    rc = pgsgd_session_create(g, p, &s);
    if (rc) return rc;
    timer.lap("session create (total)");
    std::vector<double> etas(p->iter_max + 1);
    if (pgsgd_schedule(p, etas.data(), etas.size()) < 0) { pgsgd_session_destroy(s); return PGSGD_E_INVALID; }
    rc = pgsgd_session_upload_coords(s, X, Y);
    timer.lap("coordinates upload");
    if (rc == PGSGD_OK && p->snapshot && p->snapshot_prefix && s->tiled && !s->warm_per_lane)
        fprintf(stderr, "[odgi::path_linear_sgd_layout] note: this graph runs the tile kernel, whose first iterations treat long-range pairs "
                        "differently from the CPU path (gentle, averaged pulls instead of full projections; the final layout is the same): "
                        "snapshots of iterations 1-15 show a different — smoother — transient.  --gpu-no-tiles gives the CPU path's rule term by term.\n");
    const uint64_t first_cooling = (uint64_t)std::floor(p->cooling_start * (double)p->iter_max);  // :39
    uint64_t iters = 0, terms = 0;
    double dmax = 0;
    uint32_t early = 0;
    std::vector<float> sx, sy;
    for (uint64_t it = 0; rc == PGSGD_OK && it < p->iter_max; ++it) {
        rc = pgsgd_session_iteration(s, etas[it], it >= first_cooling, p->min_term_updates);
        if (rc) break;
        rc = pgsgd_session_sync(s, &dmax);
        if (rc) break;
        ++iters;
        terms += p->min_term_updates;
        if (p->progress)
            pgsgd::progress_line("[odgi::path_linear_sgd_layout] 2D path-guided SGD:", terms, p->iter_max * p->min_term_updates,
                                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        if (it + 1 >= p->iter_max) break;
        if (dmax <= p->delta) {  // :142 (also stops at 0, as upstream notes)
            if (p->progress)
                fprintf(stderr, "\n[odgi::path_linear_sgd_layout] delta_max: %g <= delta: %g. Threshold reached, therefore ending iterations.\n", dmax, p->delta);
            early = 1;
            break;
        }
        if (p->snapshot && p->snapshot_prefix) {  // :379-408: snapshot k after iteration k, k = 1..iter_max-1
            sx.resize(2 * g->n_nodes);
            sy.resize(2 * g->n_nodes);
            rc = pgsgd_session_peek_coords(s, sx.data(), sy.data());  // (between iterations: the last launch's far pulls arrive with the next one)
            if (rc) break;
            const std::string name = std::string(p->snapshot_prefix) + std::to_string(it + 1);
            fprintf(stderr, "[odgi::path_linear_sgd_layout] snapshot thread: Taking snapshot!\n");
            rc = pgsgd::write_snapshot(name.c_str(), g->n_nodes, sx.data(), sy.data(), snapshot_names);
        }
    }
    if (p->progress) fprintf(stderr, "\n");
    if (rc == PGSGD_OK) rc = pgsgd_session_flush(s);  // the far pulls of the last tile launch
    timer.lap("iterations");
    if (rc == PGSGD_OK) rc = Xd ? pgsgd_session_download_coords_f64(s, Xd, Yd) : pgsgd_session_download_coords(s, X, Y);
    timer.lap("coordinates download");
    if (stats) {
        stats->iterations = iters;
        stats->term_updates = terms;
        stats->last_delta_max = dmax;
        stats->n_streams = pgsgd_session_n_streams(s);
        stats->early_stop = early;
        stats->frame_doublings = s->frame_doublings;
        stats->apply_lanes = s->split && !s->tiled ? s->apply_lanes : 0;
        stats->tiled = s->tiled ? (s->warm_per_lane ? 2u : 1u) : 0u;
        pgsgd_session_kernel_time(s, &stats->kernel_ms, nullptr, 0);
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    pgsgd_session_destroy(s);
    timer.lap("session destroy");
    return rc;
}

int pgsgd_write_lay_f32(const char* path, uint64_t n_ends, const float* X, const float* Y) {
    std::vector<double> dx(n_ends), dy(n_ends);
    for (uint64_t i = 0; i < n_ends; ++i) { dx[i] = X[i]; dy[i] = Y[i]; }
    return pgsgd_write_lay(path, n_ends, dx.data(), dy.data());
}


// ---------------------------------------------------------------------------------------------
// 1D path-guided SGD (`odgi sort -Y`, reference src/algorithms/path_sgd.cpp:12-500): SURVEY 8(f) row 2.
// Runs the per-lane 1D kernel on a session's step records, path table, zeta cache and streams.

// reference defaults of `odgi sort -Y` (src/subcommand/sort_main.cpp:313-320,378-414)
extern "C" int pgsgd_sort_params_defaults(const pgsgd_graph_view* g, pgsgd_params* p) {
    int rc = pgsgd_params_defaults(g, p);
    if (rc) return rc;
    uint64_t max_steps = 0, max_bp = 0;
    for (uint64_t i = 0; i < g->n_paths; ++i) {
        const uint64_t b = g->path_first[i], e = g->path_first[i + 1];
        max_steps = std::max(max_steps, e - b);
        if (e > b && g->step_pos) max_bp = std::max(max_bp, g->step_pos[e - 1] + g->node_len[g->step_handle[e - 1] >> 1]);
        if (e > b && !g->step_pos) {   // (a view without step positions: the path's length by a walk)
            uint64_t bp = 0;
            for (uint64_t k = b; k < e; ++k) bp += g->node_len[g->step_handle[k] >> 1];
            max_bp = std::max(max_bp, bp);
        }
    }
    p->iter_max = 100;                                                    // :313
    p->min_term_updates = (uint64_t)(1.0 * (double)g->n_steps);           // :383
    p->eta_max = (double)max_steps * (double)max_steps;                   // :414
    p->space = max_bp;                                                    // :387 get_max_path_length (nucleotides)
    p->space_max = 100;                                                   // :388
    const uint64_t max_dist = std::max<uint64_t>(p->space_max + 1, 100);  // :390-394, MAX_NUMBER_OF_ZIPF_DISTRIBUTIONS = 100
    if (p->space > p->space_max && max_dist > p->space_max)               // :404-410
        p->space_quantization_step = std::max<uint64_t>(2, (uint64_t)std::ceil((double)(p->space - p->space_max) / (double)(max_dist - p->space_max)));
    else
        p->space_quantization_step = 100;
    return PGSGD_OK;
}

static int sort_session(const pgsgd_graph_view* g, const pgsgd_params* p, pgsgd_session** s) {
    pgsgd_params q = *p;
    q.flags |= PGSGD_FLAG_NO_TILES;  // the 1D path has per-lane kernels only (one pass, or two on a small lane-bound graph)
    q.flags &= ~PGSGD_FLAG_HOT_NODE_CAP;
    q.terms_per_anchor = 1;
    q.snapshot = 0;
    return pgsgd_session_create(g, &q, s);
}

static pgsgd::SortArgs sort_args(const pgsgd_session* s, long long* d_x, double scale) {
    pgsgd::SortArgs sa;
    sa.X = d_x;
    sa.frozen = nullptr;
    sa.scale = scale;
    sa.inv_scale = 1.0 / scale;
    sa.zc_cool.init(0.001);  // adj_theta once cooling starts, path_sgd.cpp:195
    sa.n_terms = 0;
    sa.eta = 0;
    sa.cooling = 0;
    (void)s;
    return sa;
}

// X: host fp64 [n_nodes], pre-initialised (the reference starts from the cumulative node length,
// path_sgd.cpp:67-73), updated in place.  Iterations 0..iter_max (:181), cooling when
// iteration > floor(cooling_start*iter_max) (:194), stop when max|Delta| <= delta (:183).
extern "C" int pgsgd_sort_run(const pgsgd_graph_view* g, const pgsgd_params* p, double* X, pgsgd_stats* stats) {
    return pgsgd_sort_run_targets(g, p, nullptr, X, stats);
}

extern "C" int pgsgd_sort_run_targets(const pgsgd_graph_view* g, const pgsgd_params* p, const uint8_t* target_nodes, double* X,
                                      pgsgd_stats* stats) {
    pgsgd::clear_error();
    if (stats) memset(stats, 0, sizeof *stats);
    if (!g || !p || !X) return PGSGD_E_INVALID;
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    bool multi = false;
    for (uint64_t i = 0; i < g->n_paths && !multi; ++i) multi = g->path_first[i + 1] - g->path_first[i] > 1;
    if (!multi) {
        int dev;
        return pick_device(p->device, &dev);
    }
    const auto t0 = std::chrono::steady_clock::now();
    pgsgd_session* s = nullptr;
    rc = sort_session(g, p, &s);
    if (rc) return rc;
    const uint64_t N = g->n_nodes;
    const double scale = 65536.0;  // quanta per bp: +-1.4e14 bp of range in a signed 64-bit word
    long long* d_x = nullptr;
    double* d_f = nullptr;
    uint8_t* d_frozen = nullptr;
    auto cleanup = [&](int code) {
        if (d_x) (void)hipFree(d_x);
        if (d_f) (void)hipFree(d_f);
        if (d_frozen) (void)hipFree(d_frozen);
        pgsgd_session_destroy(s);
        return code;
    };
#define T_TRY(expr)                                                                             \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return cleanup(PGSGD_E_HIP);                                                        \
        }                                                                                       \
    } while (0)
    T_TRY(hipMalloc(&d_x, N * sizeof(long long)));
    T_TRY(hipMalloc(&d_f, N * sizeof(double)));
    T_TRY(hipMemcpyAsync(d_f, X, N * sizeof(double), hipMemcpyHostToDevice, s->stream));
    const int sgrid = (int)std::min<uint64_t>((N + 255) / 256, 2048);
    hipLaunchKernelGGL(pgsgd::sort_pack_kernel, dim3(sgrid), dim3(256), 0, s->stream, d_f, N, scale, d_x);
    T_TRY(hipGetLastError());
    std::vector<double> etas(p->iter_max + 1);
    if (pgsgd_schedule(p, etas.data(), etas.size()) < 0) return cleanup(PGSGD_E_INVALID);
    const uint64_t first_cooling = (uint64_t)std::floor(p->cooling_start * (double)p->iter_max);
    const uint32_t block = s->n_streams >= (uint32_t)pgsgd::kBlock ? pgsgd::kBlock : ((s->n_streams + 63) / 64) * 64;
    const uint32_t grid = (s->n_streams + block - 1) / block;
    pgsgd::SortArgs sa = sort_args(s, d_x, scale);
    if (target_nodes) {
        T_TRY(hipMalloc(&d_frozen, N));
        T_TRY(hipMemcpyAsync(d_frozen, target_nodes, N, hipMemcpyHostToDevice, s->stream));
        sa.frozen = d_frozen;
    }
    const uint64_t chunk = std::max<uint64_t>(s->n_streams, (std::max<uint64_t>(1, s->split_chunk) / s->n_streams) * s->n_streams);  // whole rounds of the sampler streams
    if (s->split) {  // (the moving kernel's LDS attribute was set when the session was created)
        s->terms_cap = std::min<uint64_t>(p->min_term_updates, chunk);
        T_TRY(hipMalloc(&s->d_terms, std::max<uint64_t>(1, s->terms_cap) * sizeof(uint4)));
    }
    hipEvent_t e0, e1;
    T_TRY(hipEventCreate(&e0));
    T_TRY(hipEventCreate(&e1));
    uint64_t iters = 0, terms = 0;
    double dmax = 0, kernel_ms = 0;
    uint32_t early = 0;
    for (uint64_t it = 0; it <= p->iter_max; ++it) {
        sa.n_terms = p->min_term_updates;
        sa.eta = etas[it];
        sa.cooling = it > first_cooling ? 1u : 0u;
        T_TRY(hipMemsetAsync(s->d_delta_max, 0, sizeof(unsigned int), s->stream));
        T_TRY(hipEventRecord(e0, s->stream));
        if (s->split) {  // a small lane-bound graph: sampling apart from moving, as in the layout (pgsgd_kernels.hpp)
            for (uint64_t t0 = 0; t0 < p->min_term_updates; t0 += chunk) {
                sa.n_terms = std::min<uint64_t>(chunk, p->min_term_updates - t0);
                if (s->pf_lds) hipLaunchKernelGGL((pgsgd::sort_sample_terms_kernel<true>), dim3(grid), dim3(block), s->lds_bytes, s->stream, s->dc, sa, s->d_terms);
                else hipLaunchKernelGGL((pgsgd::sort_sample_terms_kernel<false>), dim3(grid), dim3(block), 0, s->stream, s->dc, sa, s->d_terms);
                T_TRY(hipGetLastError());
                hipLaunchKernelGGL(pgsgd::sort_apply_terms_resident_kernel, dim3(1), dim3(((s->apply_lanes + 63) / 64) * 64), N * sizeof(long long), s->stream,
                                   s->dc, sa, s->d_terms, s->apply_lanes);
                T_TRY(hipGetLastError());
            }
        } else if (s->pf_lds) hipLaunchKernelGGL((pgsgd::sort_iteration_kernel<true>), dim3(grid), dim3(block), s->lds_bytes, s->stream, s->dc, sa);
        else hipLaunchKernelGGL((pgsgd::sort_iteration_kernel<false>), dim3(grid), dim3(block), 0, s->stream, s->dc, sa);
        T_TRY(hipGetLastError());
        T_TRY(hipEventRecord(e1, s->stream));
        T_TRY(hipMemcpyAsync(s->h_delta_max, s->d_delta_max, sizeof(unsigned int), hipMemcpyDeviceToHost, s->stream));
        T_TRY(hipStreamSynchronize(s->stream));
        float ms = 0, f;
        T_TRY(hipEventElapsedTime(&ms, e0, e1));
        kernel_ms += ms;
        memcpy(&f, s->h_delta_max, sizeof f);
        dmax = f;
        ++iters;
        terms += p->min_term_updates;
        if (p->progress)
            fprintf(stderr, "\r[odgi::path_linear_sgd] 1D path-guided SGD: iteration %llu/%llu  eta %.4g  delta_max %.4g   ",
                    (unsigned long long)(it + 1), (unsigned long long)(p->iter_max + 1), etas[it], dmax);
        if (it < p->iter_max && dmax <= p->delta) { early = 1; break; }
    }
    if (p->progress) fprintf(stderr, "\n");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    hipLaunchKernelGGL(pgsgd::sort_unpack_kernel, dim3(sgrid), dim3(256), 0, s->stream, d_x, N, 1.0 / scale, d_f);
    T_TRY(hipGetLastError());
    T_TRY(hipMemcpyAsync(X, d_f, N * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    T_TRY(hipStreamSynchronize(s->stream));
    if (stats) {
        stats->iterations = iters;
        stats->term_updates = terms;
        stats->last_delta_max = dmax;
        stats->kernel_ms = kernel_ms;
        stats->n_streams = s->n_streams;
        stats->apply_lanes = s->split ? s->apply_lanes : 0;
        stats->early_stop = early;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return cleanup(PGSGD_OK);
#undef T_TRY
}

// parity hook: out[(j*n_streams+g)*2 + {0,1}] = flat steps a, b of stream g's j-th term (fresh streams)
extern "C" int pgsgd_sort_trace_terms(const pgsgd_graph_view* g, const pgsgd_params* p, int cooling, uint64_t terms_per_stream,
                                      uint64_t* out, uint32_t* n_streams_out) {
    pgsgd::clear_error();
    if (!g || !p || !out) return PGSGD_E_INVALID;
    pgsgd_session* s = nullptr;
    int rc = sort_session(g, p, &s);
    if (rc) return rc;
    if (n_streams_out) *n_streams_out = s->n_streams;
    const size_t n = (size_t)terms_per_stream * s->n_streams * 2;
    uint64_t* d_out = nullptr;
    hipError_t e = hipMalloc(&d_out, n * sizeof(uint64_t));
    if (e == hipSuccess) {
        pgsgd::SortArgs sa = sort_args(s, nullptr, 65536.0);
        sa.cooling = cooling ? 1u : 0u;
        const uint32_t block = s->n_streams >= (uint32_t)pgsgd::kBlock ? pgsgd::kBlock : ((s->n_streams + 63) / 64) * 64;
        const uint32_t grid = (s->n_streams + block - 1) / block;
        const uint64_t seed_base = p->seed + (uint64_t)p->stream_offset;
        if (s->pf_lds) hipLaunchKernelGGL((pgsgd::sort_trace_kernel<true>), dim3(grid), dim3(block), s->lds_bytes, s->stream, s->dc, sa, seed_base, terms_per_stream, d_out);
        else hipLaunchKernelGGL((pgsgd::sort_trace_kernel<false>), dim3(grid), dim3(block), 0, s->stream, s->dc, sa, seed_base, terms_per_stream, d_out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        (void)hipFree(d_out);
    }
    pgsgd_session_destroy(s);
    if (e != hipSuccess) { set_error("1D trace failed: %s", hipGetErrorString(e)); return PGSGD_E_HIP; }
    return PGSGD_OK;
}
