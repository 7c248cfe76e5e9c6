/*
 * pgsgd_oracle.h — CPU restatement of the reference `odgi layout` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (odgi_amd/, the C-ABI library, the CLI) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * do, and only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: the upstream reference cannot be built here (every deps/ submodule is empty), and
 * the reference has no test or golden vector of this path.  What pins this oracle:
 *   - the libstdc++ distributions are checked against the real libstdc++ of this image
 *     (oracle/check_libstdcxx.cpp);
 *   - index semantics against the known answers of reference src/unittest/pathindex.cpp:22-130;
 *   - the schedule against its closed forms, the layout quality against the one reference-made
 *     output in the tree (test/DRB1-3123_unsorted.og.lay, path stress 0.0871).
 * The third-party arithmetic (Xoshiro-cpp, dirtyzipf) is restated from its published algorithm:
 * "parity unpinned" for those two modules (versions unrecoverable, sources absent).
 */
#ifndef PGSGD_ORACLE_H
#define PGSGD_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* same memory layout as pgsgd_graph_view (include/pgsgd.h) so tests can pass one for the other */
typedef struct orc_graph {
    uint64_t n_nodes, n_steps, n_paths;
    const uint32_t* node_len;
    const uint64_t* path_first;
    const uint32_t* step_path;
    const uint32_t* step_handle;
    const uint64_t* step_pos;
} orc_graph;

typedef struct orc_params {
    uint64_t iter_max;
    uint64_t iter_with_max_learning_rate;
    uint64_t min_term_updates;
    double delta, eps, eta_max, theta;
    uint64_t space, space_max, space_quantization_step;
    double cooling_start;
} orc_params;

typedef struct orc_term {
    uint64_t ka, kb;       /* flat (path-major) step indices of the two steps                  */
    uint32_t off_a, off_b; /* node-end offsets (0 = start, 1 = end)                            */
    uint64_t pos_a, pos_b; /* end-adjusted path positions                                      */
    uint32_t dither;       /* low 32 bits of the draw whose top bit chose end a (unused upstream) */
} orc_term;

/* third-party arithmetic, restated */
void     orc_rng_seed(uint64_t seed, uint64_t s[4]);          /* SplitMix64 -> Xoshiro256+ state */
uint64_t orc_rng_next(uint64_t s[4]);                         /* Xoshiro256+                     */
uint64_t orc_uniform_u64(uint64_t s[4], uint64_t range);      /* libstdc++ uniform_int [0,range) */
double   orc_canonical(uint64_t s[4]);                        /* generate_canonical<double,53>   */
double   orc_fast_precise_pow(double a, double b);            /* dirtyzipf::fast_precise_pow     */
uint64_t orc_zipf(uint64_t s[4], uint64_t n, double theta, double zeta_n); /* dirty zipf in [1,n] */

/* path_sgd_layout.cpp:433-468, :86-97 */
void   orc_schedule(const orc_params* p, double* etas /* [iter_max+1] */);
size_t orc_zeta_size(uint64_t space, uint64_t space_max, uint64_t quant);
void   orc_zetas(double theta, uint64_t space, uint64_t space_max, uint64_t quant, double* zetas);

/* One pass of the sampler (path_sgd_layout.cpp:182-270).  Returns 0 when the reference would
 * `continue` (single-step path), 1 when *t holds a term. */
int orc_sample_term(const orc_graph* g, const orc_params* p, const double* zetas, int cooling,
                    uint64_t s[4], orc_term* t);

/* The device stream definition, serialised: stream i (seed+stream_offset+i) runs terms
 * i, i+n, i+2n, ... of each iteration; here executed round-robin in term order.
 * trace: out[(j*n_streams+g)*4+{0..3}] = {ka,kb,off_a,off_b} for fresh streams (first iteration). */
void orc_trace_terms(const orc_graph* g, const orc_params* p, uint64_t seed, uint32_t n_streams,
                     uint32_t stream_offset, int cooling, uint32_t terms_per_anchor, uint64_t terms_per_stream, uint64_t* out);
/* the terms one tile of the device's tile kernel draws in iteration `epoch`; returns their number */
uint64_t orc_tile_terms(const orc_graph* g, const orc_params* p, uint64_t seed_base, uint64_t epoch, uint64_t n_terms,
                        uint64_t steps_total, uint64_t tile, uint32_t lanes, uint64_t t0, uint64_t cum, uint32_t n, uint32_t path,
                        int cooling, uint32_t share, uint64_t* out);
/* the Zipf/uniform coin the 64 lanes of wave `wave` of a tile share in their trip `trip` of a warm iteration */
int orc_tile_wave_coin(uint64_t seed_base, uint64_t epoch, uint64_t tile, uint32_t wave, uint64_t trip);
/* fp32 mirror of the device arithmetic; bit-exact with the GPU for n_streams == 1 */
void orc_layout_streams_f32(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, int hogwild_stores, uint32_t terms_per_anchor,
                            float* X, float* Y, double* last_delta_max);
/* mirror of the device's default coordinate format: {u32 Xq, u32 Yq} fixed point, x = x_off + Xq/scale,
 * stochastic rounding of each step with the term's spare random bits; X,Y are quantised on entry and
 * de-quantised on exit exactly as the device's upload/download do.  Bit-exact for n_streams == 1. */
/* sequential mirror of the tile kernel run by one workgroup with one lane per tile (see the .c file) */
#define ORC_TILE_DRAIN_AFTER 1u
#define ORC_TILE_TWO_SNAPSHOTS 2u
#define ORC_TILE_NO_FLUSH 4u
#define ORC_TILE_CONSTANT_RELAX 8u
#define ORC_TILE_SNAPSHOT_PASS 16u
#define ORC_TILE_LANE_COIN 32u
#define ORC_TILE_NO_PAIRS 64u
#define ORC_TILE_RELAX_R5 128u
#define ORC_TILE_PAIRS 0x10000u
#define ORC_TILE_DRAIN_BESIDE 0x8000u
void orc_tile_layout_q32(const orc_graph* g, const orc_params* p, uint64_t seed_base,
                         uint64_t n_tiles, const uint64_t* t0, const uint64_t* cum, const uint32_t* tn, const uint32_t* tpath,
                         const uint32_t* tlanes, uint64_t steps_total, uint64_t n_items, uint64_t n_first, const uint32_t* tile_begin,
                         const uint32_t* tile_end, const uint32_t* win0, const uint32_t* local, uint32_t region,
                         double x_off, double y_off, double quanta_per_bp, float* X, float* Y,
                         double* last_delta_max, uint64_t* checksum, uint64_t* far_terms);
void orc_tile_layout_q32_ex(const orc_graph* g, const orc_params* p, uint64_t seed_base,
                         uint64_t n_tiles, const uint64_t* t0, const uint64_t* cum, const uint32_t* tn, const uint32_t* tpath,
                         const uint32_t* tlanes, uint64_t steps_total, uint64_t n_items, uint64_t n_first, const uint32_t* tile_begin,
                         const uint32_t* tile_end, const uint32_t* win0, const uint32_t* local, uint32_t region,
                         double x_off, double y_off, double quanta_per_bp, float* X, float* Y,
                         double* last_delta_max, uint64_t* checksum, uint64_t* far_terms, uint32_t policy, uint64_t stop_after);
void orc_layout_streams_q32(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, int hogwild_stores, uint32_t terms_per_anchor,
                            double x_off, double y_off, double quanta_per_bp, float* X, float* Y, double* last_delta_max,
                            uint64_t* checksum_before_after /* [4]: sum Xq, sum Yq before; after */);
/* same schedule of terms, fp64 arithmetic exactly as path_sgd_layout.cpp:283-363 */
void orc_layout_streams_f64(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, double* X, double* Y);

/* device-concurrency model: rounds of n_streams terms on one snapshot, displacements summed */
void orc_layout_batched_f64(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, double* X, double* Y);

/* The reference itself: Hogwild workers + 1 ms controller (path_sgd_layout.cpp:120-425).
 * max_seconds > 0 bounds the wall time (for the timed CPU baseline). */
typedef struct orc_hogwild_stats {
    uint64_t terms;      /* term updates applied by all workers      */
    uint64_t iterations; /* iterations completed by the controller   */
    double seconds;      /* worker launch -> join                    */
} orc_hogwild_stats;
void orc_layout_hogwild(const orc_graph* g, const orc_params* p, uint32_t nthreads, double max_seconds,
                        double* X, double* Y, orc_hogwild_stats* st);
void orc_layout_hogwild_curve(const orc_graph* g, const orc_params* p, uint32_t nthreads, double max_seconds,
                              double* X, double* Y, orc_hogwild_stats* st,
                              uint64_t n_snap, const uint64_t* snap_iters, double* snapX, double* snapY);

/* quality metrics */
double orc_path_stress_sampled(const orc_graph* g, const double* X, const double* Y,
                               uint64_t n_pairs, uint64_t seed);
/* all same-path step pairs a<b, step-start ends, d = pos_b - pos_a */
void orc_path_stress_near(const orc_graph* g, const double* X, const double* Y, uint32_t zmax, double theta, uint32_t nthreads,
                          double* num, double* mass, double* zero_mass);   /* expectation of the sampled stress over Zipf jumps <= zmax, exactly */
double orc_path_stress_exhaustive(const orc_graph* g, const double* X, const double* Y);
/* odgi stats -s, 2D branch (stats_main.cpp:667-716) */
void orc_path_distance(const orc_graph* g, const double* X, const double* Y, double* per_node, double* per_bp);

/* ---- 1D path-guided SGD of `odgi sort -Y` (reference src/algorithms/path_sgd.cpp:12-500) ---- */
void orc_sort_initial(const orc_graph* g, double* X /* [n_nodes] */);
void orc_sort_trace_terms(const orc_graph* g, const orc_params* p, uint64_t seed, uint32_t n_streams, uint32_t stream_offset,
                          int cooling, uint64_t terms_per_stream, uint64_t* out /* [terms][streams][2] = ka, kb */);
void orc_sort_streams(const orc_graph* g, const orc_params* p, uint64_t seed, uint32_t n_streams, uint32_t stream_offset,
                      double quanta_per_bp, const uint8_t* frozen, double* X, double* last_delta_max);
void orc_sort_hogwild(const orc_graph* g, const orc_params* p, uint32_t nthreads, double max_seconds, const uint8_t* frozen, double* X, orc_hogwild_stats* st);
double orc_sort_stress(const orc_graph* g, const double* X, uint64_t n_pairs, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
