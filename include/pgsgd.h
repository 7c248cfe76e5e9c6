/*
 * pgsgd.h — C ABI of the MI355X-native path-guided SGD 2D layout (the `odgi layout` hot path).
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, owns no global state and
 * never calls exit(): errors come back as negative codes (pgsgd_strerror()).  This is the surface
 * the reference's `--gpu` route would bind instead of `cuda::gpu_layout`
 * (reference: src/cuda/layout.h:65-80, called from src/algorithms/path_sgd_layout.cpp:470-503).
 *
 * Terms:  N = nodes, S = path steps, P = paths.  A "term" is one sampled node-pair update
 * (reference: src/algorithms/path_sgd_layout.cpp:178-375).
 * Coordinates are per node END: X,Y have 2N entries, index 2*rank+0 = node start, 2*rank+1 = node
 * end (reference: src/subcommand/layout_main.cpp:268-269,288; src/algorithms/layout.cpp:76-79).
 */
#ifndef PGSGD_H
#define PGSGD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGSGD_VERSION 1
/* ABI version: bumped whenever a struct of this header grows, an entry point is removed, or one changes what it does to the
 * session.  A binding compares pgsgd_abi_version() with the PGSGD_ABI_VERSION it was built against, and the sizes of the
 * structs it passes with pgsgd_abi_struct_sizes(), ONCE at load time (odgi_amd/_lib.py does; INTEGRATION.md section 6 lists what
 * changed at each version).  5: pgsgd_stats grew by `relabeled` and `tiled` (56 -> 64 bytes: a caller built against the
 * older header would have its stack overwritten by the library's memset of the whole struct);
 * pgsgd_session_download_* deliver the far pulls still waiting before they read (pgsgd_session_peek_* read the words as
 * they are); pgsgd_tile_region_for is gone; pgsgd_session_set_shard(.., -1) chooses the exact exchange itself.
 * 6: a tiled session of a schedule of 30 iterations and more delivers the far pulls of its launches a launch later from the sixth
 * iteration on (their drain runs on a second stream beside the next launch; PGSGD_FLAG_SYNC_DRAIN: as before) — pgsgd_session_peek_* may then lack
 * the pulls of the last TWO launches; the far pulls of a launch amount to one projection, not half (tile kernel; final layouts of
 * short schedules change); pgsgd_graph_view::step_path / step_pos may be NULL.  New entry points (nothing removed):
 * pgsgd_path_stress_near, pgsgd_session_terms_executed, pgsgd_session_drain_beside, pgsgd_session_read_step_records,
 * pgsgd_graph_load_flags, pgsgd_graph_drop_step_index, pgsgd_session_probe_words, pgsgd_tile_quad_partner, pgsgd_session_drain_plan; the lanes of a tile share the
 * uniform partners of a warm trip in quads (one 128-byte line of step records for four terms), not pairs.
 * 7: pgsgd_session_set_shard(.., -1) shards a fixed-point tiled session by region with the exact exchange from PGSGD_SHARD_MIN_WINDOWS
 * (240) windows per device on (a thousand until now; by tile below): a multi-GPU run is one GPU's layout on every graph that can
 * occupy its devices; PGSGD_FLAG_SHARD_TILES asks for the tile shard, PGSGD_FLAG_REGION_128 for 128-node regions; new entry points
 * pgsgd_shard_flags, pgsgd_session_drain_plan. */
#define PGSGD_ABI_VERSION 7
int pgsgd_abi_version(void);
/* sizeof(pgsgd_graph_view), sizeof(pgsgd_params), sizeof(pgsgd_stats) as the LIBRARY was built (any pointer may be NULL) */
void pgsgd_abi_struct_sizes(size_t* graph_view, size_t* params, size_t* stats);

/* ---- error codes ------------------------------------------------------------------------- */
#define PGSGD_OK              0
#define PGSGD_E_INVALID      -1  /* bad argument / inconsistent view                              */
#define PGSGD_E_NODEVICE     -2  /* no usable HIP device / HIP runtime error before launch        */
#define PGSGD_E_HIP          -3  /* HIP runtime error (see pgsgd_last_error)                      */
#define PGSGD_E_NOMEM        -4
#define PGSGD_E_IO           -5  /* file could not be read / written                              */
#define PGSGD_E_FORMAT       -6  /* malformed GFA / .lay                                          */
#define PGSGD_E_NOTOPTIMIZED -7  /* node ids are not exactly 1..N (layout_main.cpp:148-151)       */
#define PGSGD_E_UNSUPPORTED  -8

const char* pgsgd_strerror(int code);
/* thread-local detail text of the last failing call on this thread ("" if none). */
const char* pgsgd_last_error(void);

/* ---- the lowered graph: what graph_t + XP hand to the SGD loop ---------------------------- */
/*
 * Read-only, caller-owned HOST arrays, path-major (steps of path p are
 * [path_first[p], path_first[p+1]) ).  Replaces, for this path:
 *   node_len     <- graph.get_length(handle)                    (src/odgi.cpp:65-71)
 *   path_first   <- XP::get_path_step_count per path, prefixed  (src/algorithms/xp.cpp:375-377)
 *   step_path    <- XP npi_iv  (path id of a step; here 0-based) (xp.cpp:136-148, 434)
 *   step_handle  <- XPPath::handles: 2*node_rank + is_reverse    (xp.cpp:585-595, 691-697)
 *   step_pos     <- XPPath::positions: 0-based bp offset of the step start in its path
 *                                                                (xp.cpp:607-617, 393-397)
 * nr_iv (rank of a step in its path) is implicit: rank = k - path_first[step_path[k]].
 * step_path and step_pos MAY BE NULL (both follow from the other three): the session then uploads 4 bytes per step instead of
 * 12 and builds the positions on the device (a segmented prefix sum over node_len[step_handle >> 1]; the reference's GPU route
 * flattens its paths on the device too, src/cuda/layout.cu:371-410); host code that needs them (the quality figures) walks the paths.
 */
typedef struct pgsgd_graph_view {
    uint64_t n_nodes;
    uint64_t n_steps;
    uint64_t n_paths;
    const uint32_t* node_len;    /* [n_nodes]   */
    const uint64_t* path_first;  /* [n_paths+1] */
    const uint32_t* step_path;   /* [n_steps] or NULL */
    const uint32_t* step_handle; /* [n_steps]   */
    const uint64_t* step_pos;    /* [n_steps] or NULL */
} pgsgd_graph_view;

/* ---- parameters: the argument list of path_linear_sgd_layout_gpu -------------------------- */
/* (reference: src/algorithms/path_sgd_layout.hpp:59-80; cuda::layout_config_t layout.h:65-77) */
#define PGSGD_FLAG_COORD_LOAD_PLAIN   0x1u /* debug: read coordinates through L1/L2 (stale-prone)   */
#define PGSGD_FLAG_FP32_ATOMICS       0x2u /* keep {f32 x, f32 y} words and use four fp32 atomic adds */
                                           /* per term instead of the default 32.32 packed fixed      */
                                           /* point with one 64-bit integer atomic add per node end  */
#define PGSGD_FLAG_HOGWILD_STORES     0x4u /* update by load -> store like the reference CPU loop    */
                                           /* (path_sgd_layout.cpp:360-363) instead of atomic adds   */
#define PGSGD_FLAG_NO_TILES           0x8u /* never use the region-exclusive tile kernel.  By default  */
                                           /* graphs that suit it (large, sorted, no hub node; default  */
                                           /* format/update/term stream, automatic stream count) run     */
                                           /* it: first steps stratified by tile, node windows in LDS   */
#define PGSGD_FLAG_NO_FAR_CAP        0x10u /* tile kernel: do not cap the learning rate of terms whose   */
                                           /* partner lies outside the staged window (debug / A-B)      */
#define PGSGD_FLAG_ONE_SIDED_FAR     0x20u /* tile kernel, experiment (not the reference's update rule): */
                                           /* a term whose partner lies outside the staged window moves */
                                           /* only its first end, by twice the step; no far write;      */
                                           /* ignored on graphs with window-less tiles                  */
#define PGSGD_FLAG_HOT_NODE_CAP      0x40u /* per-lane kernel, experiment: bound the lanes by the bulk of the nodes (2*S /  */
                                           /* s*, nodes busier than s* carry a tenth of the steps) instead of by the       */
                                           /* busiest one, and cap the learning rate of terms on busier nodes at 1/h.      */
                                           /* 12x faster on a hub graph (DRB1-3123_unsorted) at +30 % stress, worse        */
                                           /* everywhere else (profiles/r02/hotcap_per_lane_*.jsonl): never the default    */
#define PGSGD_FLAG_NO_PIPELINE       0x80u /* per-lane kernel: the plain term loop (one term per lane in flight) instead of the    */
                                           /* software-pipelined one (four terms per lane in different stages); same terms, same   */
                                           /* arithmetic, A/B and parity                                                            */
#define PGSGD_FLAG_NO_SPLIT        0x1000u /* per-lane kernel: never run an iteration in two passes (what small lane-bound graphs     */
                                           /* run by default: every stream the GPU holds samples, one workgroup with the lanes the   */
                                           /* busiest node allows moves the ends in LDS); same streams, same arithmetic, A/B, parity  */
#define PGSGD_FLAG_EXACT_MATH      0x2000u /* tile kernel: IEEE divisions, the correctly rounded square root and 64-bit path positions   */
                                           /* in a term's geometry instead of the hardware's reciprocal / reciprocal square root (1 ulp) */
                                           /* and 32-bit positions: the instance the CPU oracle's mirror reproduces bit for bit (parity  */
                                           /* tests).  Chosen automatically when a path is 2^32 bp long or longer                        */
#define PGSGD_FLAG_NO_PARTNER_PAIRS 0x4000u /* tile kernel: every lane keeps its own uniform partner (path_sgd_layout.cpp:235-237).  By default the */
                                           /* lanes of a wave share partners in quads in a warm iteration's uniform trips: lane r of four takes    */
                                           /* flat step lead ^ r of the 128-byte line of four step records its quad's first lane drew (one memory */
                                           /* request for four terms; every partner is still uniform over the path).  A/B and parity             */
#define PGSGD_FLAG_LOCK_WINDOW_ENDS 0x8000u /* tile kernel, option (measured: no effect on the layout, 3 % slower — profiles/r04/NOTES.md): conflict   */
                                           /* resolution on shared node coordinates.  While a term's learning rate is in the projection regime     */
                                           /* (mu >= 0.1) it takes a lock bit on each of its window ends (one LDS atomic OR per end; the lanes of a  */
                                           /* wave that go for one end are served one after the other) and does nothing when an end is taken       */
#define PGSGD_FLAG_NO_RELABEL    0x10000u /* pgsgd_layout_run: keep the caller's node ranks.  By default a graph whose node ranks do not follow its paths   */
                                           /* (more than 2 % of a sample of steps jump over 128 ranks) and would therefore miss the tile kernel is laid out */
                                           /* under ranks ordered by (component, mean path position) when that order at least halves the jumps              */
                                           /* (pgsgd_graph_path_order); the coordinates come back under the caller's ranks.  Not with snapshots (-u)        */
#define PGSGD_FLAG_SYNC_DRAIN    0x20000u /* tile kernel: deliver every launch's far pulls in front of the very next launch.  By default a session of a  */
                                         /* schedule as long as the reference's (iter_max >= 30) sums them on a second stream BESIDE the next     */
                                         /* launch and delivers them a launch later, from the sixth iteration on (+2.7 %, same layout). */
#define PGSGD_FLAG_REGION_128    0x40000u /* tile kernel: regions of 128 nodes (tiles of 112 steps) instead of 256 / 224: twice the windows per launch.  For  */
                                         /* the sessions of a multi-GPU run sharded by region on a graph whose 256-node windows would leave a device     */
                                         /* fewer than a thousand per launch (pgsgd_shard_flags sets it): 9 % slower on one GPU, same layout  */
#define PGSGD_FLAG_SHARD_TILES   0x80000u /* multi-GPU: shard the tile kernel by TILE wherever the rule would shard by region — every device stages every */
                                         /* window and runs every G-th tile: faster on graphs too small to fill G devices with windows (config 4 at G = 8:  */
                                         /* 41 ms of kernels per rank against 79) and NOT one GPU's layout: stress +8 / +14 / +21 % at G = 2 / 4 / 8         */
#define PGSGD_FLAG_ABLATE(n)  (((n) & 0xfu) << 8) /* profiling only: 1 no atomics,                     */
                                                  /* 3 no coordinate loads, 4 neither (results invalid) */

typedef struct pgsgd_params {
    uint64_t iter_max;                    /* -x, default 30                                        */
    uint64_t iter_with_max_learning_rate; /* -F is parsed but the reference passes 0               */
    uint64_t min_term_updates;            /* terms per iteration (-G/-U; default 10*S)             */
    double   delta;                       /* -j, default 0                                         */
    double   eps;                         /* -g, default 0.01                                      */
    double   eta_max;                     /* -v, default (max steps per path)^2                    */
    double   theta;                       /* -a, default 0.99                                      */
    uint64_t space;                       /* -k, default max steps per path                        */
    uint64_t space_max;                   /* -I, default 1000                                      */
    uint64_t space_quantization_step;     /* -l, default 100                                       */
    double   cooling_start;               /* -K, default 0.5                                       */
    uint64_t seed;        /* sampler base seed; stream g seeds Xoshiro256+ with seed+g, exactly as  */
                          /* reference worker tid does with 9399220+tid (path_sgd_layout.cpp:168)   */
    uint32_t n_streams;   /* concurrent sampler streams (GPU lanes). 0 = choose from graph size     */
    uint32_t stream_offset; /* first stream id of this device: rank r of a G-way run uses r*n_streams */
    int32_t  device;      /* HIP device ordinal; -1 = current device                               */
    int32_t  snapshot;    /* 1 = write <prefix><k> (.lay) after iteration k, k=1..iter_max-1        */
    const char* snapshot_prefix;
    int32_t  progress;    /* 1 = progress line on stderr                                           */
    uint32_t flags;       /* PGSGD_FLAG_*                                                          */
    uint32_t terms_per_anchor; /* partners drawn per first step; 0/1 = the reference's term stream.     */
                          /* m > 1 keeps each term's distribution (first step uniform, partner by    */
                          /* the reference's rule) but fetches and updates the first step once per m  */
                          /* terms — fewer scattered memory requests per term                         */
    uint32_t n_devices;   /* GPUs of one node to run on (0/1 = one: `device`).  G > 1: devices device..device+G-1, */
                          /* graph replicated, every step's terms split 1/G per GPU, coordinates merged with an   */
                          /* RCCL all-reduce at every step (pgsgd_layout_run / _f64 only; sessions are per GPU)   */
} pgsgd_params;

#define PGSGD_DEFAULT_SEED 9399220ull

typedef struct pgsgd_stats {
    uint64_t iterations;      /* iterations actually run                                           */
    uint64_t term_updates;    /* terms applied, all iterations                                     */
    double   last_delta_max;  /* max |Delta| seen in the last iteration (early-stop quantity)       */
    double   kernel_ms;       /* sum of update-kernel durations (HIP events on the launch stream)   */
    double   wall_ms;         /* upload + iterations + download                                    */
    uint32_t n_streams;       /* streams actually used                                             */
    uint32_t early_stop;      /* 1 if Delta_max <= delta ended the run (path_sgd_layout.cpp:142)    */
    uint32_t frame_doublings; /* times the fixed-point coordinate frame was widened during the run  */
    uint32_t apply_lanes;     /* iterations in two passes (a small lane-bound graph): the lanes that moved node ends; else 0 */
    uint32_t relabeled;       /* 1: the run renamed the nodes by path position (PGSGD_FLAG_NO_RELABEL turns that off)   */
    uint32_t tiled;           /* 1: the tile kernel ran (2: after per-lane warm iterations); 0: the per-lane kernel     */
} pgsgd_stats;

/* Fill every field of *p with the reference defaults derived from the path index
 * (src/subcommand/layout_main.cpp:153-155,198-204,251-266). */
int pgsgd_params_defaults(const pgsgd_graph_view* g, pgsgd_params* p);
/* Node ranks ordered by (path-connected component, mean bp position of the node's steps): new_rank_of_old[N].  disorder_* :
 * the share of a sample of consecutive path steps whose node ranks differ by more than 128, before and after (may be NULL).
 * What pgsgd_layout_run renames the nodes by when the caller's ranks do not follow the paths.  Host only. */
int pgsgd_graph_path_order(const pgsgd_graph_view* g, uint32_t* new_rank_of_old, double* disorder_before, double* disorder_after);

/* Learning-rate schedule, iter_max+1 doubles (path_sgd_layout.cpp:433-468).  Returns count. */
int64_t pgsgd_schedule(const pgsgd_params* p, double* etas, size_t capacity);

/* Zipf zeta cache (path_sgd_layout.cpp:86-97).  size() then fill(). */
size_t pgsgd_zeta_table_size(uint64_t space, uint64_t space_max, uint64_t space_quant);
int pgsgd_zeta_table(double theta, uint64_t space, uint64_t space_max, uint64_t space_quant,
                     double* zetas, size_t capacity);

/* Initial layout (layout_main.cpp:268-330).  mode: 'd','r','u','g','h'.  seed==0 draws from
 * std::random_device like the reference; any other value is reproducible.  X,Y: [2N] doubles. */
int pgsgd_init_layout(const pgsgd_graph_view* g, char mode, uint64_t seed, double* X, double* Y);

/* ---- one-shot run: the replacement for cuda::gpu_layout ------------------------------------ */
/* X,Y: host fp32 [2N], pre-initialised, updated in place.  Blocking.  Fails with
 * PGSGD_E_NODEVICE when no MI355X-class HIP device is usable: there is no CPU fallback. */
int pgsgd_layout_run(const pgsgd_graph_view* g, const pgsgd_params* p,
                     float* X, float* Y, pgsgd_stats* stats);
/* The same run with double-precision coordinates: the result comes back at the full resolution of the device's
 * fixed-point words (1/16 bp on a 3e7-bp layout, where fp32 is spaced 2-4 bp apart); what the `odgi layout`
 * binary and the C++ shim use for their std::vector<std::atomic<double>> / .lay outputs. */
int pgsgd_layout_run_f64(const pgsgd_graph_view* g, const pgsgd_params* p, double* X, double* Y, pgsgd_stats* stats);

/* ---- session API: the same run, one iteration at a time ------------------------------------ */
/* Used by the multi-GPU driver (one process per GPU; the coordinate all-reduce between eta
 * steps happens outside, on the device buffer) and by the benchmark. */
typedef struct pgsgd_session pgsgd_session;

int pgsgd_session_create(const pgsgd_graph_view* g, const pgsgd_params* p, pgsgd_session** out);
void pgsgd_session_destroy(pgsgd_session* s);
/* host fp32 X,Y [2N]  <->  device coordinate words (one 8-byte word per node end) */
int pgsgd_session_upload_coords(pgsgd_session* s, const float* X, const float* Y);
/* download_* return the LAYOUT: they first deliver what the last tile launch left in the outbox (pgsgd_session_flush; nothing
 * to do in a per-lane session).  peek_* read the coordinates as they are between iterations — what the reference's
 * per-iteration snapshots see (path_sgd_layout.cpp:379-408): without the far pulls that arrive with the next launch. */
int pgsgd_session_download_coords(pgsgd_session* s, float* X, float* Y);
int pgsgd_session_download_coords_f64(pgsgd_session* s, double* X, double* Y); /* exact x_off + q / quanta_per_bp */
int pgsgd_session_peek_coords(pgsgd_session* s, float* X, float* Y);
int pgsgd_session_peek_coords_f64(pgsgd_session* s, double* X, double* Y);
int pgsgd_session_peek_words(pgsgd_session* s, uint64_t* words /* [2N] raw coordinate words */);
/* device pointer to the 2N coordinate words, and their format: fixed_point 1 = {u32 Xq, u32 Yq}
 * with x = x_off + Xq / quanta_per_bp (frame chosen at upload), 0 = {f32 x, f32 y} */
void* pgsgd_session_coords_ptr(pgsgd_session* s);
int pgsgd_session_download_words(pgsgd_session* s, uint64_t* words /* [2N] raw coordinate words, after the flush */);
int pgsgd_session_coord_format(const pgsgd_session* s, int* fixed_point, double* x_off, double* y_off,
                               double* quanta_per_bp);
/* launch stream (hipStream_t as void*); set to make the session launch on a caller stream */
void* pgsgd_session_stream(pgsgd_session* s);
int pgsgd_session_set_stream(pgsgd_session* s, void* hip_stream);
/* Asynchronously run `n_terms` terms with learning rate eta (cooling: 0/1) on the session stream. */
int pgsgd_session_iteration(pgsgd_session* s, double eta, int cooling, uint64_t n_terms);
/* Part `part` of `n_parts` of such an iteration (multi-GPU: one part per exchange).  The per-lane kernel runs
 * the part-th slice of the n_terms terms; the tile kernel runs every n_parts-th tile with its whole share. */
int pgsgd_session_iteration_part(pgsgd_session* s, double eta, int cooling, uint64_t n_terms, uint32_t part, uint32_t n_parts);
/* Wait for the stream; returns max |Delta| of the last iteration in *delta_max (may be NULL). */
int pgsgd_session_sync(pgsgd_session* s, double* delta_max);
/* Tile kernel only (a no-op otherwise): what a tile launch adds to node ends outside its windows is delivered right
 * before the NEXT launch, so between iterations the coordinates lack the far pulls of the last launch.  The download_*
 * entry points call this themselves; a caller that reads the device words directly (pgsgd_session_coords_ptr) calls it
 * first; per-iteration snapshots (path_sgd_layout.cpp:379-408) are taken without it (peek_*). */
int pgsgd_session_flush(pgsgd_session* s);
/* The fixed-point coordinate frame (2^32 quanta per axis, 8x the layout's extent at upload).  Kernels flag any
 * coordinate they see in the outer quarter of the frame; pgsgd_session_sync then doubles the frame (same centre, half
 * the resolution) before the next iteration, so a node end never wraps around.  A session that is part of a multi-GPU
 * run (exchange_mark / set_shard) only reports: its driver calls pgsgd_session_reframe on every rank in the same
 * iteration when any rank's guard_hit is set. */
int pgsgd_session_frame_status(const pgsgd_session* s, int* guard_hit, uint32_t* doublings);
int pgsgd_session_reframe(pgsgd_session* s);
/* Sum of update-kernel durations since creation / last reset, measured with HIP events. */
int pgsgd_session_kernel_time(pgsgd_session* s, double* total_ms, uint64_t* launches, int reset);
/* Tile kernel only: time spent in the two streaming kernels around every tile launch (coordinate snapshot into the
 * step records before it, far-update drain after it), same clock and reset as pgsgd_session_kernel_time. */
int pgsgd_session_aux_time(pgsgd_session* s, double* snapshot_ms, double* drain_ms);
/* kernel launches and memset/copy operations the iteration calls have put on the stream so far (what a step costs
 * besides its dominant kernel: bench.py reports them per step) */
int pgsgd_session_launch_counts(const pgsgd_session* s, uint64_t* kernel_launches, uint64_t* copies);
/* Measurement: the shader clock (MHz) the last tile-kernel launch ran at and that launch's duration as its first
 * workgroup saw it (s_memtime against the constant 100 MHz s_memrealtime); blocks until the stream is idle; 0 when the
 * session has run no tile launch. */
int pgsgd_session_shader_clock(pgsgd_session* s, double* mhz, double* launch_ms);
/* Debug (PGSGD_DEBUG=1 PGSGD_TILE_TAIL=1): the share of (workgroups x duration) of the last windowed tile launch that its
 * persistent workgroups were alive for; the rest is the launch's tail.  Zeros when the knob is off. */
int pgsgd_session_tile_tail(pgsgd_session* s, double* alive_fraction, double* launch_ms, uint32_t* workgroups);
/* Profiling hook: the twelve raw words of the tile kernel's probes (shader clock, conflict counters, tail probe, device-side term
 * count); with the kernel's profiling instance 5 the phases of a workgroup's time (tools/gpu_tile_phases.py).  Zeros without tiles. */
int pgsgd_session_probe_words(pgsgd_session* s, uint64_t out[12]);
/* Tile kernel: terms that went for their window ends' locks so far (conflict resolution on shared node coordinates while
 * the learning rate is in the projection regime), and terms among them that found an end taken and did nothing. */
int pgsgd_session_tile_conflicts(pgsgd_session* s, uint64_t* locked, uint64_t* lost);
/* parity hook: step records [first, first + count) as the session built them: {handle, node length, position low, position high} */
int pgsgd_session_read_step_records(pgsgd_session* s, uint64_t first, uint64_t count, uint32_t* out);
/* whether the session sums its launches' far pulls on a second stream beside the next launch (PGSGD_FLAG_SYNC_DRAIN: never), and
 * far_drain_kernel's time there (ms, HIP events) — off the launch stream's critical path */
int pgsgd_session_drain_beside(pgsgd_session* s, int* on, double* drain_ms);
/* how the session's far pulls are summed (far_drain_kernel): parts = workgroups that share a bucket's node range (1 up to 2.1e6
 * nodes, 2^(shift - 14) beyond: every part looks at every message of the bucket and keeps its own), slices = workgroups that share a
 * (bucket, part)'s messages */
int pgsgd_session_drain_plan(pgsgd_session* s, uint32_t* parts, uint32_t* slices);
/* terms the session's tile launches have executed, counted on the device (cumulative; 0 without tiles) */
int pgsgd_session_terms_executed(pgsgd_session* s, uint64_t* terms);
/* Tile kernel only: far updates that found their bucket's share of the message pool used up and were applied as
 * direct atomic adds instead (0 in normal operation; the pool is sized from the tile table). */
int64_t pgsgd_session_outbox_overflow(pgsgd_session* s);
uint32_t pgsgd_session_n_streams(const pgsgd_session* s);
/* Multi-GPU sharding of the tile kernel.  Every iteration call then takes the FULL term count of the
 * block, of which only the owned tiles' share is applied.
 *   by_region = 0: this device runs tiles rank, rank+world, ... (every work item on every device, each
 *                  visited tile with its whole share of terms: no short term loops);
 *   by_region = 1: this device runs work items (node regions with all their tiles) rank, rank+world, ...
 *                  (disjoint windows across devices; needs >= ~1000 regions per device and launch).
 *   by_region = 2: by region with the EXACT exchange (pgsgd_session_exchange_exact_begin / _end below): an iteration is two
 *                  parts, one per region colour, each followed by an integer exchange — the devices then hold, bit for
 *                  bit, one GPU's coordinates.
 *   by_region -1: the session's rule, the one both drivers use.  Fixed-point coordinates (the default): by region with the exact
 *                  exchange (2) when that leaves every launch at least PGSGD_SHARD_MIN_WINDOWS windows per device (windows, not
 *                  the parts a one-GPU session cuts them into; rounds 4-6 asked for a thousand and sharded config 4 by tile:
 *                  faster and +8..21 % stress — now PGSGD_FLAG_SHARD_TILES), by tile (0) when it does not.  fp32 words: by
 *                  region with the merge rule (1) from a thousand windows per device on, by tile below.
 * Returns 1 (sharded by tile), 2 (by region) or 3 (by region, exact) when the session is tiled, 0 when it runs the
 * per-lane kernel (shard the term count instead), < 0 on error.  The tile streams of a sharded session are keyed on (seed, iteration,
 * tile, lane) — stream_offset, which the devices' per-lane streams need to differ, does not enter them. */
int pgsgd_session_set_shard(pgsgd_session* s, uint32_t rank, uint32_t world, int by_region);
/* Flag bits a multi-GPU driver ORs into the params of its `world` sessions BEFORE it creates them, so that pgsgd_session_set_shard(.., -1)
 * can shard them by region with the exact exchange: PGSGD_FLAG_REGION_128 when 256-node regions would leave a device fewer than
 * PGSGD_SHARD_FULL_WINDOWS windows per launch and 128-node regions leave it at least PGSGD_SHARD_MIN_WINDOWS; 0 otherwise (one device,
 * no tiles, fp32 words, PGSGD_FLAG_SHARD_TILES, a graph too small either way).  Both in-tree drivers call it (pgsgd_multi.cpp,
 * odgi_amd/distributed.py via bench.py). */
#define PGSGD_SHARD_FULL_WINDOWS 1000u
#define PGSGD_SHARD_MIN_WINDOWS   240u
uint32_t pgsgd_shard_flags(uint64_t n_nodes, uint32_t world, uint32_t flags);
/* returns 1 when the session runs the region-exclusive tile kernel, 0 when it runs the per-lane kernel, 2 (known
 * after pgsgd_session_upload_coords) when the initial layout has no global structure — long-range stress above
 * 0.1 — and the iterations before cooling therefore run the per-lane kernel, the cooling ones the tile kernel */
int pgsgd_session_tile_info(const pgsgd_session* s, uint64_t* n_tiles, uint64_t* n_nonlocal_tiles,
                            uint64_t* n_work_items, uint32_t* region_nodes, uint32_t* tile_steps);
/* A tiled session whose launches would have fewer than 24 rounds of work items cuts every window's tiles into k consecutive
 * parts, each a work item that waits for the part before it (a launch's tail — slots idle while the last items finish —
 * shrinks with the items).  Returns k (1: whole windows; 0: per-lane kernel); *n_launch_items: items of the two launches.
 * pgsgd_session_tile_items lists them in launch order for an unsharded session. */
int pgsgd_session_tile_parts(const pgsgd_session* s, uint64_t* n_launch_items);
/* 1 when the session takes its windows in node order, one run of work items per XCD (what a session whose windows are cut
 * into parts does: the workgroups of an XCD share the step records around neighbouring windows in their L2); 0: one run by
 * decreasing size. */
int pgsgd_session_tile_order(const pgsgd_session* s);
/* The rule behind it, pure host arithmetic: 1 when the `windows` of a launch do not fill the resident workgroups once;
 * otherwise what brings the launch to 24 rounds of work items, parts of at least four tiles, at most 16 per window. */
uint32_t pgsgd_tile_parts_for(uint64_t windows, uint64_t tiles_per_window, uint64_t resident_workgroups);
/* The cut itself on caller's arrays (host only, tests): the items of one launch — windows first, the last n_windowless
 * without a window — with every window's tiles in k consecutive parts: part 0 of every window, then part 1, ...; out_flags =
 * bit 0 window | bit 1 another item waits for this one | (1 + index of the item this one waits for) << 2.  Returns the
 * number of items (call with capacity 0 to size the arrays), < 0 on error. */
int64_t pgsgd_tile_split_items(const uint32_t* tile_begin, const uint32_t* tile_end, const uint32_t* win0, uint64_t n_items, uint32_t n_windowless,
                               uint32_t k, uint32_t* out_begin, uint32_t* out_end, uint32_t* out_win0, uint32_t* out_flags, uint64_t capacity);
/* Two rules of the tile kernel's sampler, restated on the host for tests: the Zipf/uniform coin (path_sgd_layout.cpp:205)
 * that the 64 lanes of wave `wave` of a tile share in their trip `trip` of a warm iteration (a SplitMix64 stream per
 * wave, seeded like a lane's generator with lane id 1023 - wave), and the partner (rank in its path) an odd lane takes in
 * a uniform trip: the step sharing a 64-byte unit of the step records with its even neighbour's partner, flat step ^ 1,
 * when that is a step of the path, its own draw otherwise. */
int pgsgd_tile_wave_coin(uint64_t seed_base, uint64_t epoch, uint64_t tile, uint32_t wave, uint64_t trip);
uint32_t pgsgd_tile_pair_partner(uint32_t lead_flat_step, uint32_t path_first_step, uint32_t path_steps, uint32_t own_rank);
/* ... and what sessions run since round 6: lane r (1..3) of four consecutive lanes takes flat step lead ^ r of the 128-byte line of
 * four step records its quad's first lane drew, when that is a step of the path (lane 0 and a cut line: the own draw). */
uint32_t pgsgd_tile_quad_partner(uint32_t lead_flat_step, uint32_t lane_in_quad, uint32_t path_first_step, uint32_t path_steps, uint32_t own_rank);
/* How the session's per-lane launches run: 0 one pass (a lane samples a term and moves its ends), 1 two passes (a small
 * lane-bound graph whose 2N coordinate words fit one compute unit's LDS: pgsgd_session_n_streams() streams sample, one
 * workgroup of *apply_lanes lanes moves the ends in LDS). */
int pgsgd_session_split_info(const pgsgd_session* s, uint32_t* apply_lanes);
/* Which instance of the tile kernel the session runs: 1 = fast math (hardware reciprocal / reciprocal square root, 32-bit
 * path positions; the default), 0 = exact (PGSGD_FLAG_EXACT_MATH, or a path of 2^32 bp or more). */
int pgsgd_session_tile_math(const pgsgd_session* s);
/* Parity hook: the displacement of n terms by both instances of the tile kernel's geometry, on device `device`:
 * inputs eta, d[i] (path distance), dx[i], dy[i] (layout difference, bp), cap[i] (learning-rate cap); out_fast / out_exact
 * [3n] = {r_x, r_y, |Delta|} per term.  The exact form is the oracle's (path_sgd_layout.cpp:280-352 in fp32). */
int pgsgd_debug_tile_displacement(int device, uint64_t n, float eta, const float* d, const float* dx, const float* dy,
                                  const float* cap, float* out_fast, float* out_exact);
/* Multi-GPU exchange between eta steps (or sub-steps); all three run on the session stream.
 *   mark : remember the current coordinates as the exchange base (call once, after upload);
 *   begin: buf[2e..2e+1] = (dx, dy) node end e moved since the base, in bp; buf[4N+e] = dx^2+dy^2;
 *          the caller then all-reduces (SUM) the 6N-float device buffer over the G ranks;
 *   end  : coords = base + S * clamp(Q/|S|^2, 1/G, 1) per node end, base = coords. */
int pgsgd_session_exchange_mark(pgsgd_session* s);
int pgsgd_session_exchange_begin(pgsgd_session* s, void* device_buf_6N_floats);
int pgsgd_session_exchange_end(pgsgd_session* s, const void* device_buf_6N_floats, int world_size);
/* begin, with a tail of 2*world floats behind the 6N: slot 6N + rank = this device's max |Delta| of the last iteration
 * call, slot 6N + world + rank = 1 if it saw a coordinate in the guard band of the fixed-point frame; the other
 * devices' slots are written as zero, so after the SUM all-reduce of all 6N + 2*world floats every device holds every
 * device's values: the reference's stop rule (path_sgd_layout.cpp:142) and the frame widening need no second collective. */
int pgsgd_session_exchange_begin_stats(pgsgd_session* s, void* device_buf_6N_plus_2world_floats, uint32_t rank, uint32_t world);
/* The exact exchange of a tiled session sharded by region with pgsgd_session_set_shard(s, rank, world, 2): an iteration is then
 * pgsgd_session_iteration_part(s, eta, cooling, n_terms, c, 2) for colour c = 0, 1, each followed by
 *   exchange_exact_begin  (delivers the launch's far pulls; device_buf[0 .. 2N) = coords - base as 64-bit integers; tail
 *                          [2N .. 2N + 3 world): in this rank's slots the launch's far-pull count, the bits of its max |Delta|
 *                          (a float), its frame-guard flag — zero in the other ranks' slots),
 *   a SUM all-reduce of the 2N + 3 world 64-bit integers over the ranks,
 *   exchange_exact_end    (coords = base = base + sum; the far-pull count over all ranks goes to the learning-rate cap).
 * The windows a rank runs in one colour's launch are its alone and integer adds commute, so every rank ends with, bit for
 * bit, the coordinates one GPU computes (with its snapshot pass per iteration): no merge rule, 16 N bytes per exchange. */
int pgsgd_session_exchange_exact_begin(pgsgd_session* s, void* device_buf_2N_plus_3world_u64, uint32_t rank, uint32_t world);
int pgsgd_session_exchange_exact_end(pgsgd_session* s, const void* device_buf_2N_plus_3world_u64, uint32_t world);
/* Parity hooks of the tile kernel.  Every term of a tiled iteration is a pure function of (seed +
 * stream_offset, iteration number, tile, lane of the tile, position in the lane's stream); tile_table copies up to `capacity` tiles (in work order:
 * first step, steps before the tile, steps, path) and returns their number; trace_tile_terms replays the
 * terms tile `tile` draws in iteration `epoch` (1-based) of n_terms terms: out[4*j + {0..3}] = {flat step
 * a, flat step b, end offset a, end offset b}; returns the number of terms. */
int64_t pgsgd_session_tile_table(const pgsgd_session* s, uint64_t* t0, uint64_t* cum, uint32_t* n, uint32_t* path,
                                 uint64_t capacity, uint64_t* steps_total);
/* Parity hook: lanes that work on each tile at once (at most the workgroup size; fewer where one node is visited
 * often inside the tile); lane l draws the tile's terms l, l + lanes, ... from its own stream.  Returns the tile count. */
int64_t pgsgd_session_tile_lanes(const pgsgd_session* s, uint32_t* lanes, uint64_t capacity);
/* Parity hook: the work items of the tile kernel in launch order; item i runs tiles [tile_begin[i], tile_end[i])
 * of the tile table on the node window starting at rank win0[i] (local[i] = 0: no window, every end in global
 * memory); the first *n_first items belong to the launch of the even regions.  Returns the item count. */
int64_t pgsgd_session_tile_items(const pgsgd_session* s, uint32_t* tile_begin, uint32_t* tile_end, uint32_t* win0,
                                 uint32_t* local, uint64_t capacity, uint64_t* n_first);
int64_t pgsgd_session_trace_tile_terms(pgsgd_session* s, uint64_t tile, int cooling, uint64_t epoch, uint64_t n_terms,
                                       uint64_t* out, uint64_t capacity_terms);
/* Parity hook: run the sampler only and write, for stream g and its j-th term (j < terms_per_stream),
 * out[(j*n_streams+g)*4 + {0,1,2,3}] = {flat step a, flat step b, end offset a, end offset b}
 * without touching coordinates or the persistent stream states. */
int pgsgd_session_trace_terms(pgsgd_session* s, int cooling, uint64_t terms_per_stream, uint64_t* out);

/* ---- graph input: GFA v1 -> lowered view (gfa_to_handle.cpp:27-217 + xp.cpp:49-175) -------- */
typedef struct pgsgd_graph pgsgd_graph; /* owns its arrays */
int pgsgd_graph_from_gfa(const char* path, int n_threads, pgsgd_graph** out);
/* odgi's native graph file (.og, what graph_t::serialize writes: odgi.cpp:1632-1685, node.cpp:422-435)
 * -> lowered view; path "-" reads standard input.  The graph must be optimized (ids 1..N). */
int pgsgd_graph_from_og(const char* path, int n_threads, pgsgd_graph** out);
/* The reference's input dispatch (utils.cpp:110-134): names ending in "gfa" are GFA, the rest .og. */
int pgsgd_graph_load(const char* path, int n_threads, pgsgd_graph** out);
/* ... with PGSGD_LOAD_NO_STEP_INDEX: step_path and step_pos are not built (GFA) or dropped after the walk (.og): the views carry NULL */
#define PGSGD_LOAD_NO_STEP_INDEX 1u
int pgsgd_graph_load_flags(const char* path, int n_threads, uint32_t flags, pgsgd_graph** out);
/* frees step_path and step_pos of a loaded graph; views taken afterwards carry NULL for them */
int pgsgd_graph_drop_step_index(pgsgd_graph* g);
/* Seeded synthetic "linearised pangenome" (BASELINE.json configs 4/5). */
int pgsgd_graph_synthetic(uint64_t n_nodes, uint64_t n_paths, uint64_t seed, pgsgd_graph** out);
void pgsgd_graph_free(pgsgd_graph* g);
int pgsgd_graph_get_view(const pgsgd_graph* g, pgsgd_graph_view* view);
uint64_t pgsgd_graph_edge_count(const pgsgd_graph* g);
/* edges as handle pairs (2*rank+rev), [2*edge_count] */
const uint64_t* pgsgd_graph_edges(const pgsgd_graph* g);
const char* pgsgd_graph_path_name(const pgsgd_graph* g, uint64_t path);
uint64_t pgsgd_graph_max_path_steps(const pgsgd_graph* g);

/* ---- post-processing and output (layout_main.cpp:388-463; algorithms/layout.cpp) ----------- */
/* Weakly connected components over the edges: comp_of_node[N] (component ids in discovery order,
 * i.e. by lowest node rank: weakly_connected_components.cpp:8-68).  Returns component count. */
int64_t pgsgd_weak_components(uint64_t n_nodes, const uint64_t* edges, uint64_t n_edges,
                              uint32_t* comp_of_node);
/* Stack components vertically with border 1000 (layout_main.cpp:407-435), in place. */
int pgsgd_pack_components(uint64_t n_nodes, const uint32_t* comp_of_node, uint64_t n_comp,
                          double* X, double* Y);
int pgsgd_write_tsv(const char* path, uint64_t n_nodes, const uint32_t* comp_of_node,
                    uint64_t n_comp, const double* X, const double* Y);
/* .lay = f64 min_value + sdsl::enc_vector<> of the bit patterns (layout.cpp:43-61). */
int pgsgd_write_lay(const char* path, uint64_t n_ends, const double* X, const double* Y);
int pgsgd_lay_buffer(uint64_t n_ends, const double* X, const double* Y, uint8_t** buf, size_t* len);
int pgsgd_read_lay(const char* path, uint64_t* n_ends, double** X, double** Y);
void pgsgd_free(void* p);

/* ---- layout quality (no reference equivalent for 2D; odgi stats -s formula restated) -------- */
/* Sampled path stress: mean over sampled same-path end pairs of ((|p_a-p_b| - d)/d)^2. */
int pgsgd_path_stress(const pgsgd_graph_view* g, const double* X, const double* Y,
                      uint64_t n_pairs, uint64_t seed, double* stress);
/* The near pairs' part of that expectation WITHOUT sampling error: every pair of steps <= zmax steps apart, all end choices,
 * weighted by the sampler's probability of drawing it (quality.cpp).  num/mass: [zmax*4], index (z-1)*4 + 2*flip_a + flip_b;
 * hist_step [mod_step] / hist_rank [mod_rank] (optional, may be NULL): the numerator by step rank % mod_step / node rank % mod_rank. */
int pgsgd_path_stress_near(const pgsgd_graph_view* g, const double* X, const double* Y, uint32_t zmax, double theta, uint32_t threads,
                           double* num, double* mass, double* zero_mass, uint32_t mod_step, double* hist_step, uint32_t mod_rank, double* hist_rank);
/* stats_main.cpp:667-716 (2D branch): sum over paths of consecutive-step distances, per node and per bp */
int pgsgd_path_distance(const pgsgd_graph_view* g, const double* X, const double* Y,
                        double* per_node, double* per_bp);

/* ---- 1D path-guided SGD of `odgi sort -Y` (reference src/algorithms/path_sgd.cpp:12-500) -- */
/* Same sampler as the layout, one coordinate per node, the reference's 1D rules (adj_theta = 0.001 once
 * cooling, terms of path distance 0 dropped, iterations 0..iter_max).  X: host fp64 [n_nodes], updated in place. */
int pgsgd_sort_params_defaults(const pgsgd_graph_view* g, pgsgd_params* p); /* sort_main.cpp:313-320,378-414 */
int pgsgd_sort_initial(const pgsgd_graph_view* g, double* X);               /* path_sgd.cpp:67-73 */
int pgsgd_sort_run(const pgsgd_graph_view* g, const pgsgd_params* p, double* X, pgsgd_stats* stats);
/* the same with target sorting (path_sgd.cpp:289-301,392-397): target_nodes[n_nodes] != 0 marks nodes that keep
 * their position; a term between two of them is counted and does nothing.  NULL = no target nodes. */
int pgsgd_sort_run_targets(const pgsgd_graph_view* g, const pgsgd_params* p, const uint8_t* target_nodes, double* X,
                           pgsgd_stats* stats);
int pgsgd_sort_order(uint64_t n_nodes, const double* X, uint64_t* order);   /* path_sgd.cpp:641-650 */
/* weakly connected components ranked by the average id of their nodes (path_sgd.cpp:552-587) */
int pgsgd_sort_component_ranks(uint64_t n_nodes, const uint64_t* edges, uint64_t n_edges, uint32_t* comp_rank_of_node);
/* node ranks by (component rank, position, handle) (path_sgd.cpp:641-650); comp_rank_of_node may be NULL */
int pgsgd_sort_order_components(uint64_t n_nodes, const double* X, const uint32_t* comp_rank_of_node, uint64_t* order);
/* `odgi sort --path-sgd-layout` (path_sgd.cpp:651-672): .lay with X = (pos, pos + node length) per node of `order`, Y = 0 */
int pgsgd_sort_write_lay(const pgsgd_graph_view* g, const double* X, const uint64_t* order, const char* path);
int pgsgd_sort_stress(const pgsgd_graph_view* g, const double* X, uint64_t n_pairs, uint64_t seed, double* stress);
/* parity hook: out[(j*n_streams+g)*2 + {0,1}] = flat steps a, b of fresh stream g's j-th term */
int pgsgd_sort_trace_terms(const pgsgd_graph_view* g, const pgsgd_params* p, int cooling, uint64_t terms_per_stream,
                           uint64_t* out, uint32_t* n_streams);

/* ---- the subcommand: argv as `odgi layout` takes it (layout_main.cpp:18-466) --------------- */
int pgsgd_main_layout(int argc, char** argv);

#ifdef __cplusplus
}
#endif
#endif /* PGSGD_H */
