/*
 * pgsgd_oracle.c — CPU restatement of the reference path-guided SGD 2D layout.
 * TEST INFRASTRUCTURE ONLY (see pgsgd_oracle.h).  Plain C11 + pthreads.
 *
 * Each function cites the reference lines it follows (paths relative to the odgi tree).
 * The reference samples a flat step index in NODE-major order (XP nr_iv/npi_iv); the lowered view is
 * PATH-major.  The draw is uniform over all S steps in both, so (path, rank) has the same
 * distribution; rank = k - path_first[path] replaces nr_iv[k]-1.
 */
#define _GNU_SOURCE
#include "pgsgd_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------- */
/* Xoshiro-cpp (Reputeless/Xoshiro-cpp): Xoshiro256Plus(seed) seeds its four words from SplitMix64 */
/* call site: src/algorithms/path_sgd_layout.cpp:168-169                                          */
void orc_rng_seed(uint64_t seed, uint64_t s[4]) {
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        s[i] = z ^ (z >> 31);
    }
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

uint64_t orc_rng_next(uint64_t s[4]) {
    const uint64_t result = s[0] + s[3];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

/* libstdc++ (GCC >= 11) std::uniform_int_distribution<uint64_t>(0, range-1) on a generator whose
 * range is exactly [0, 2^64): Lemire's multiply-shift with rejection (bits/uniform_int_dist.h,
 * _S_nd<unsigned __int128>).  Call sites: path_sgd_layout.cpp:175-176,182,205-206,235-237,253,262.
 * range == 0 means the full 2^64 range (the generator output is returned as is). */
uint64_t orc_uniform_u64(uint64_t s[4], uint64_t range) {
    if (range == 0) return orc_rng_next(s);
    unsigned __int128 product = (unsigned __int128)orc_rng_next(s) * (unsigned __int128)range;
    uint64_t low = (uint64_t)product;
    if (low < range) {
        const uint64_t threshold = (0 - range) % range;
        while (low < threshold) {
            product = (unsigned __int128)orc_rng_next(s) * (unsigned __int128)range;
            low = (uint64_t)product;
        }
    }
    return (uint64_t)(product >> 64);
}

static inline uint64_t orc_flip(uint64_t s[4]) { return orc_uniform_u64(s, 2); }

/* std::generate_canonical<double, 53> on a 64-bit generator: one draw / 2^64, result >= 1 is
 * replaced by nextafter(1, 0) (bits/random.tcc). */
double orc_canonical(uint64_t s[4]) {
    double r = (double)orc_rng_next(s) * 0x1p-64;
    if (r >= 1.0) r = 0x1.fffffffffffffp-1;
    return r;
}

/* dirtyzipf::fast_precise_pow (Ankerl's approximation): exact a^int(b) by squaring times a
 * bit-twiddled approximation of a^frac(b).  Call sites: path_sgd_layout.cpp:90 and inside the
 * distribution (:213-215,226-228); same formula as the in-tree src/cuda/layout.cu:89-113. */
double orc_fast_precise_pow(double a, double b) {
    int e = (int)b;
    union { double d; int32_t x[2]; } u;
    u.d = a;
    u.x[1] = (int32_t)((b - e) * (u.x[1] - 1072632447) + 1072632447);
    u.x[0] = 0;
    double r = 1.0;
    while (e) {
        if (e & 1) r *= a;
        a *= a;
        e >>= 1;
    }
    return r * u.d;
}

/* dirtyzipf::dirty_zipfian_int_distribution<uint64_t>(1, n, theta, zeta_n)(gen)
 * (Gray et al. 1994 as in YCSB; cross-checked with src/cuda/layout.cu:89-113).
 * The final clamp to [1,n] never fires for in-range arithmetic; it only keeps memory safe. */
uint64_t orc_zipf(uint64_t s[4], uint64_t n, double theta, double zeta_n) {
    const double alpha = 1.0 / (1.0 - theta);
    const double zeta2 = orc_fast_precise_pow(1.0, theta) + orc_fast_precise_pow(0.5, theta);
    const double eta = (1.0 - orc_fast_precise_pow(2.0 / (double)n, 1.0 - theta)) / (1.0 - zeta2 / zeta_n);
    const double u = orc_canonical(s);
    const double uz = u * zeta_n;
    if (uz < 1.0) return 1;
    if (uz < 1.0 + orc_fast_precise_pow(0.5, theta)) return 2;
    const double v = 1.0 + (double)n * orc_fast_precise_pow(eta * u - eta + 1.0, alpha);
    uint64_t r = (v >= 1.0 && v < 1.8446744073709552e19) ? (uint64_t)v : 1; /* NaN/neg -> 1 */
    if (r < 1) r = 1;
    if (r > n) r = n;
    return r;
}

/* ------------------------------------------------------------------------------------------- */
/* path_linear_sgd_layout_schedule, path_sgd_layout.cpp:433-468 (w_min = 1/eta_max, w_max = 1)   */
void orc_schedule(const orc_params* p, double* etas) {
    const double w_min = 1.0 / p->eta_max;
    const double w_max = 1.0;
    const double eta_max = 1.0 / w_min;
    const double eta_min = p->eps / w_max;
    const double lambda = log(eta_max / eta_min) / ((double)p->iter_max - 1);
    for (int64_t t = 0; t <= (int64_t)p->iter_max; t++) {
        int64_t a = t - (int64_t)p->iter_with_max_learning_rate;
        if (a < 0) a = -a;
        etas[t] = eta_max * exp(-lambda * (double)a);
    }
}

/* zeta cache, path_sgd_layout.cpp:86-97.  One extra slot is allocated so the reference's
 * out-of-bounds write for space == space_max lands inside the table (it is never read). */
size_t orc_zeta_size(uint64_t space, uint64_t space_max, uint64_t quant) {
    return (size_t)((space <= space_max ? space : space_max + (space - space_max) / quant + 1) + 1) + 1;
}

void orc_zetas(double theta, uint64_t space, uint64_t space_max, uint64_t quant, double* zetas) {
    const size_t n = orc_zeta_size(space, space_max, quant);
    for (size_t i = 0; i < n; ++i) zetas[i] = 0.0;
    double zeta_tmp = 0.0;
    for (uint64_t i = 1; i < space + 1; i++) {
        zeta_tmp += orc_fast_precise_pow(1.0 / (double)i, theta);
        if (i <= space_max) zetas[i] = zeta_tmp;
        if (i >= space_max && (i - space_max) % quant == 0)
            zetas[space_max + 1 + (i - space_max) / quant] = zeta_tmp;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* sampler: path_sgd_layout.cpp:182-270, split where the first step is fixed.  RNG draw order is the
 * reference's: orc_sample_anchor then orc_sample_partner is one reference term. */
typedef struct orc_anchor { uint64_t k, pstart, cnt, s_rank; } orc_anchor;

static int orc_sample_anchor(const orc_graph* g, uint64_t s[4], orc_anchor* a) {
    const uint64_t step_index = orc_uniform_u64(s, g->n_steps);        /* :182 */
    const uint64_t path_i = g->step_path[step_index];                  /* :186 npi_iv */
    a->pstart = g->path_first[path_i];
    a->cnt = g->path_first[path_i + 1] - a->pstart;                    /* :189 */
    if (a->cnt == 1) return 0;                                         /* :190-192 */
    a->k = step_index;
    a->s_rank = step_index - a->pstart;                                /* :200 nr_iv - 1 */
    return 1;
}

static void orc_sample_partner(const orc_graph* g, const orc_params* p, const double* zetas, int cooling,
                               const orc_anchor* a, uint64_t s[4], orc_term* t) {
    const uint64_t s_rank = a->s_rank, path_step_count = a->cnt;
    uint64_t b_rank;
    if (cooling || orc_flip(s)) {                                      /* :205 */
        if ((s_rank > 0 && orc_flip(s)) || s_rank == path_step_count - 1) { /* :206 backward */
            const uint64_t jump_space = p->space < s_rank ? p->space : s_rank;
            uint64_t space = jump_space;
            if (jump_space > p->space_max)
                space = p->space_max + (jump_space - p->space_max) / p->space_quantization_step + 1;
            const uint64_t z_i = orc_zipf(s, jump_space, p->theta, zetas[space]);
            b_rank = s_rank - z_i;
        } else {                                                       /* :219 forward */
            const uint64_t rest = path_step_count - s_rank - 1;
            const uint64_t jump_space = p->space < rest ? p->space : rest;
            uint64_t space = jump_space;
            if (jump_space > p->space_max)
                space = p->space_max + (jump_space - p->space_max) / p->space_quantization_step + 1;
            const uint64_t z_i = orc_zipf(s, jump_space, p->theta, zetas[space]);
            b_rank = s_rank + z_i;
        }
    } else {
        b_rank = orc_uniform_u64(s, path_step_count);                  /* :235-237 */
    }
    t->ka = a->k;
    t->kb = a->pstart + b_rank;
    const uint32_t h_a = g->step_handle[t->ka], h_b = g->step_handle[t->kb]; /* :242-243 */
    uint64_t pos_a = g->step_pos[t->ka], pos_b = g->step_pos[t->kb];   /* :248-249 */
    const uint32_t rev_a = h_a & 1u, rev_b = h_b & 1u;                  /* :252,261 */
    /* flip(0,1) is the top bit of one draw (range 2 never rejects; check_libstdcxx.cpp pins that);
     * the device reuses the low 32 bits of this draw as rounding dither */
    const uint64_t draw_a = orc_rng_next(s);
    t->dither = (uint32_t)draw_a;
    if (draw_a >> 63) {                                                /* :253 */
        pos_a += g->node_len[h_a >> 1];
        t->off_a = !rev_a;
    } else {
        t->off_a = rev_a;
    }
    if (orc_flip(s)) {                                                 /* :262 */
        pos_b += g->node_len[h_b >> 1];
        t->off_b = !rev_b;
    } else {
        t->off_b = rev_b;
    }
    t->pos_a = pos_a;
    t->pos_b = pos_b;
}

/* The tile kernel's sampler (pgsgd_tiles.hpp: below32_hi, pick_stage, partner_stage).  Same term distribution as the
 * reference worker's (:182-270), from TWO 64-bit words per term instead of one word per coin:
 *   word 1  bits 63..32  first step, uniform in the tile (Lemire's method on 32 bits; a rejected word is redrawn whole)
 *           bit 31 Zipf/uniform coin (:205), bit 30 direction coin (:206), bits 29/28 end choices (:253,262),
 *           bits 27..14 / 13..0 rounding dither of the x / y step
 *   word 2  Zipf: the generate_canonical variate; uniform partner (:235-237): Lemire on bits 63..32 over the path's
 *           step count
 * A lane's stream yields its terms' words in term order.  tests/test_oracle_pins.py compares the distribution of these
 * terms with orc_sample_partner's.
 * Round 4: the Zipf/uniform coin (:205) is no longer bit 31 of the lane's word but the WAVE's coin for the trip — the
 * 64 terms that lanes 64w .. 64w+63 of a tile draw in their j-th trip share it (pgsgd_tiles.hpp: tile_coin_seed), so that
 * a wavefront runs one of the two partner paths.  Every term is still a Zipf term with probability 1/2.  The coins of
 * wave w are a SplitMix64 stream seeded like a lane's generator with lane id 1023 - w: trip j takes bit j % 64 of output
 * number j / 64.  coin < 0 (ORC_TILE_LANE_COIN, the vectors committed in rounds 2 and 3): the lane's own bit 31. */
typedef struct orc_tile_pick { orc_anchor an; int zipf, back; uint64_t jump; uint32_t flags; } orc_tile_pick;

static uint64_t orc_splitmix64(uint64_t* x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
int orc_tile_wave_coin(uint64_t seed_base, uint64_t epoch, uint64_t tile, uint32_t wave, uint64_t trip) {
    uint64_t x = seed_base + epoch * 0xd1342543de82ef95ull + ((tile << 10) | (uint64_t)(1023u - wave));
    uint64_t w = 0;
    for (uint64_t k = 0; k <= trip / 64; ++k) w = orc_splitmix64(&x);
    return (int)((w >> (trip % 64)) & 1u);
}
/* the same coins in trip order without the restart: *x is the stream's state (start it at the seed), *w the current word */
static int orc_tile_wave_coin_next(uint64_t* x, uint64_t* w, uint64_t trip) {
    if (trip % 64 == 0) *w = orc_splitmix64(x);
    return (int)((*w >> (trip % 64)) & 1u);
}

static uint32_t orc_below32_hi(uint64_t s[4], uint32_t range, uint32_t* low_half) {
    uint64_t w = orc_rng_next(s);
    uint64_t m = (uint64_t)(uint32_t)(w >> 32) * range;
    if ((uint32_t)m < range) {
        const uint32_t threshold = (0u - range) % range;
        while ((uint32_t)m < threshold) {
            w = orc_rng_next(s);
            m = (uint64_t)(uint32_t)(w >> 32) * range;
        }
    }
    *low_half = (uint32_t)w;
    return (uint32_t)(m >> 32);
}

static void orc_tile_pick_first(const orc_graph* g, const orc_params* p, int cooling, int coin, uint64_t t0, uint32_t tn, uint32_t path,
                                uint64_t s[4], orc_tile_pick* pk) {
    pk->an.pstart = g->path_first[path];
    pk->an.cnt = g->path_first[path + 1] - pk->an.pstart;
    pk->an.k = t0 + orc_below32_hi(s, tn, &pk->flags);
    pk->an.s_rank = pk->an.k - pk->an.pstart;
    pk->zipf = cooling || (coin < 0 ? (int)(pk->flags >> 31) : coin);                          /* :205 */
    pk->back = 0;
    pk->jump = 0;
    if (pk->zipf) {
        pk->back = (pk->an.s_rank > 0 && ((pk->flags >> 30) & 1u)) || pk->an.s_rank == pk->an.cnt - 1;   /* :206 */
        const uint64_t room = pk->back ? pk->an.s_rank : pk->an.cnt - pk->an.s_rank - 1;
        pk->jump = p->space < room ? p->space : room;
    }
}

/* Partner pairs (rounds 4-6) and quads (round 6's second session; pgsgd_tiles.hpp: tile_pair_partner, tile_quad_partner): in a
 * uniform trip the lanes of a tile share the 128-byte line of four step records their group's first lane drew — lane r of a
 * group of `share` lanes (2: pairs, 4: quads) takes flat step lead ^ r, lead = the partner of the group's lane 0, when that is
 * a step of the path; it still draws its own partner first (its stream advances as before) and keeps it otherwise.
 * pair_lead: the group's lane-0 partner (flat step) for the other lanes of the group, ORC_NO_PAIR for lane 0, a Zipf term, or
 * the pipelines of rounds 2 and 3 (ORC_TILE_NO_PAIRS); r: the lane's index in its group (1 for the odd lane of a pair). */
#define ORC_NO_PAIR (~(uint64_t)0)
static void orc_tile_partner(const orc_graph* g, const orc_params* p, const double* zetas, const orc_tile_pick* pk, uint64_t pair_lead, uint32_t r,
                             uint64_t s[4], orc_term* t) {
    uint64_t b_rank;
    if (pk->zipf) {
        uint64_t space = pk->jump;
        if (pk->jump > p->space_max) space = p->space_max + (pk->jump - p->space_max) / p->space_quantization_step + 1;
        const uint64_t z_i = orc_zipf(s, pk->jump, p->theta, zetas[space]);
        b_rank = pk->back ? pk->an.s_rank - z_i : pk->an.s_rank + z_i;
    } else {
        uint32_t unused;
        b_rank = orc_below32_hi(s, (uint32_t)pk->an.cnt, &unused);                             /* :235-237 */
        if (pair_lead != ORC_NO_PAIR && r) {
            const uint32_t twin = ((uint32_t)pair_lead ^ r) - (uint32_t)pk->an.pstart;
            if (twin < (uint32_t)pk->an.cnt) b_rank = twin;
        }
    }
    t->ka = pk->an.k;
    t->kb = pk->an.pstart + b_rank;
    const uint32_t h_a = g->step_handle[t->ka], h_b = g->step_handle[t->kb];
    uint64_t pos_a = g->step_pos[t->ka], pos_b = g->step_pos[t->kb];
    const uint32_t rev_a = h_a & 1u, rev_b = h_b & 1u;
    t->dither = pk->flags & 0x0fffffffu;   /* 14 + 14 bits */
    if ((pk->flags >> 29) & 1u) { pos_a += g->node_len[h_a >> 1]; t->off_a = !rev_a; } else { t->off_a = rev_a; }   /* :253 */
    if ((pk->flags >> 28) & 1u) { pos_b += g->node_len[h_b >> 1]; t->off_b = !rev_b; } else { t->off_b = rev_b; }   /* :262 */
    t->pos_a = pos_a;
    t->pos_b = pos_b;
}

int orc_sample_term(const orc_graph* g, const orc_params* p, const double* zetas, int cooling,
                    uint64_t s[4], orc_term* t) {
    orc_anchor a;
    if (!orc_sample_anchor(g, s, &a)) return 0;
    orc_sample_partner(g, p, zetas, cooling, &a, s, t);
    return 1;
}

/* the device streams redraw on a single-step path so that every stream term is a real term */
static inline void sample_valid(const orc_graph* g, const orc_params* p, const double* zetas,
                                int cooling, uint64_t s[4], orc_term* t) {
    while (!orc_sample_term(g, p, zetas, cooling, s, t)) { }
}
static inline void anchor_valid(const orc_graph* g, uint64_t s[4], orc_anchor* a) {
    while (!orc_sample_anchor(g, s, a)) { }
}

/* update in fp64, exactly path_sgd_layout.cpp:283-363 (single-threaded view of the same loads/stores) */
static inline double update_f64(const orc_graph* g, const orc_term* t, double eta, double* X, double* Y) {
    double term_dist = fabs((double)t->pos_a - (double)t->pos_b);       /* :280-281 */
    if (term_dist == 0) term_dist = 1e-9;                              /* :283-285 */
    const double w_ij = 1.0 / term_dist;                               /* :295-297 */
    double mu = eta * w_ij;                                            /* :301 */
    if (mu > 1) mu = 1;
    const double d_ij = term_dist;
    const uint64_t i = 2 * (uint64_t)(g->step_handle[t->ka] >> 1) + t->off_a; /* :308-324 */
    const uint64_t j = 2 * (uint64_t)(g->step_handle[t->kb] >> 1) + t->off_b;
    double dx = X[i] - X[j];                                           /* :325 */
    const double dy = Y[i] - Y[j];
    if (dx == 0) dx = 1e-9;                                            /* :327-329 */
    const double mag = sqrt(dx * dx + dy * dy);                        /* :335 */
    const double Delta = mu * (mag - d_ij) / 2;                        /* :340 */
    const double r = Delta / mag;                                      /* :350 */
    const double r_x = r * dx, r_y = r * dy;
    X[i] = X[i] - r_x;                                                 /* :360-363 */
    Y[i] = Y[i] - r_y;
    X[j] = X[j] + r_x;
    Y[j] = Y[j] + r_y;
    return fabs(Delta);
}

/* ---- mirrors of the HIP kernel (odgi_amd/csrc/pgsgd_kernels.hpp: sgd_iteration_kernel) ----------
 * One anchor group = one first step with `mt` partners.  The anchor node's two ends are loaded once,
 * each term's anchor-side displacement is applied to that private copy at once and reaches memory as
 * one update per touched end when the group ends; partner-side updates go out term by term.
 * Built with -ffp-contract=off on both sides, so a 1-stream GPU run matches bit for bit. */
static inline float displacement_f32(float eta, uint64_t pos_a, uint64_t pos_b, float dx, float dy, float* r_x, float* r_y) {
    const int64_t diff = (int64_t)pos_a - (int64_t)pos_b;
    float d = (float)(uint64_t)(diff < 0 ? -diff : diff);
    if (d == 0.0f) d = 1e-9f;
    const float w = 1.0f / d;
    float mu = eta * w;
    if (mu > 1.0f) mu = 1.0f;
    if (dx == 0.0f) dx = 1e-9f;
    const float dx2 = dx * dx;
    const float dy2 = dy * dy;
    const float mag = sqrtf(dx2 + dy2);
    const float Delta = (mu * (mag - d)) / 2.0f;
    const float r = Delta / mag;
    *r_x = r * dx;
    *r_y = r * dy;
    return fabsf(Delta);
}

static inline uint64_t q32_shift(uint64_t w, int64_t qx, int64_t qy) {
    return (uint64_t)(uint32_t)((int64_t)(uint32_t)w + qx) | ((uint64_t)(uint32_t)((int64_t)(w >> 32) + qy) << 32);
}

static inline uint32_t q32_quantize(float v, double off, float scale) {
    double q = rint(((double)v - off) * (double)scale);
    q = q < 0.0 ? 0.0 : (q > 4294967295.0 ? 4294967295.0 : q);
    return (uint32_t)q;
}

/* fp32 words {x, y}: X[e], Y[e] per node end */
static float group_f32(const orc_graph* g, const orc_params* p, const double* zetas, int cooling, uint64_t s[4],
                       uint32_t mt, float eta, int stores, float* X, float* Y) {
    orc_anchor an;
    anchor_valid(g, s, &an);
    const uint64_t node_a = (uint64_t)(g->step_handle[an.k] & ~1u);
    float lx[2] = {0.0f, 0.0f}, ly[2] = {0.0f, 0.0f};   /* private copy of the anchor ends, loaded on first use */
    int have[2] = {0, 0};
    float dfx[2] = {0.0f, 0.0f}, dfy[2] = {0.0f, 0.0f};
    int touched[2] = {0, 0};
    float dmax = 0.0f;
    for (uint32_t r = 0; r < mt; ++r) {
        orc_term t;
        orc_sample_partner(g, p, zetas, cooling, &an, s, &t);
        const uint32_t ea = t.off_a;
        const uint64_t j = 2 * (uint64_t)(g->step_handle[t.kb] >> 1) + t.off_b;
        if (!have[ea]) { lx[ea] = X[node_a + ea]; ly[ea] = Y[node_a + ea]; have[ea] = 1; }
        const float dx = lx[ea] - X[j], dy = ly[ea] - Y[j];
        float r_x, r_y;
        const float da = displacement_f32(eta, t.pos_a, t.pos_b, dx, dy, &r_x, &r_y);
        if (da > dmax) dmax = da;
        if (stores && node_a + ea == j) continue;
        touched[ea] = 1;
        lx[ea] = lx[ea] + (-r_x);
        ly[ea] = ly[ea] + (-r_y);
        if (stores) {
            const float bx = X[j] + r_x, by = Y[j] + r_y;
            X[j] = bx; Y[j] = by;
        } else {
            dfx[ea] += -r_x;
            dfy[ea] += -r_y;
            X[j] = X[j] + r_x;
            Y[j] = Y[j] + r_y;
        }
    }
    for (int e = 0; e < 2; ++e) {
        if (!touched[e]) continue;
        if (stores) { X[node_a + e] = lx[e]; Y[node_a + e] = ly[e]; }
        else { X[node_a + e] = X[node_a + e] + dfx[e]; Y[node_a + e] = Y[node_a + e] + dfy[e]; }
    }
    return dmax;
}

/* packed {u32 Xq, u32 Yq} words */
static float group_q32(const orc_graph* g, const orc_params* p, const double* zetas, int cooling, uint64_t s[4],
                       uint32_t mt, float eta, int stores, float scale, float inv_scale, uint64_t* W) {
    orc_anchor an;
    anchor_valid(g, s, &an);
    const uint64_t node_a = (uint64_t)(g->step_handle[an.k] & ~1u);
    uint64_t la[2] = {0, 0}, dq[2] = {0, 0};
    int have[2] = {0, 0};
    int touched[2] = {0, 0};
    float dmax = 0.0f;
    for (uint32_t r = 0; r < mt; ++r) {
        orc_term t;
        orc_sample_partner(g, p, zetas, cooling, &an, s, &t);
        const uint32_t ea = t.off_a;
        const uint64_t j = 2 * (uint64_t)(g->step_handle[t.kb] >> 1) + t.off_b;
        if (!have[ea]) { la[ea] = W[node_a + ea]; have[ea] = 1; }
        const uint64_t wa = la[ea], wb = W[j];
        const float dx = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * inv_scale;
        const float dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * inv_scale;
        float r_x, r_y;
        const float da = displacement_f32(eta, t.pos_a, t.pos_b, dx, dy, &r_x, &r_y);
        if (da > dmax) dmax = da;
        if (stores && node_a + ea == j) continue;
        touched[ea] = 1;
        const float ux = (float)(t.dither & 0xffffu) * (1.0f / 65536.0f);
        const float uy = (float)(t.dither >> 16) * (1.0f / 65536.0f);
        float fx = r_x * scale;
        float fy = r_y * scale;
        fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
        fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
        const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
        if (stores) {
            la[ea] = q32_shift(wa, -qx, -qy);
            W[j] = q32_shift(wb, qx, qy);
        } else {
            const uint64_t delta = (uint64_t)qx + ((uint64_t)qy << 32);
            la[ea] = wa - delta;
            dq[ea] += delta;
            W[j] += delta;
        }
    }
    for (int e = 0; e < 2; ++e) {
        if (!touched[e]) continue;
        if (stores) W[node_a + e] = la[e];
        else W[node_a + e] += (uint64_t)0 - dq[e];
    }
    return dmax;
}

void orc_trace_terms(const orc_graph* g, const orc_params* p, uint64_t seed, uint32_t n_streams,
                     uint32_t stream_offset, int cooling, uint32_t terms_per_anchor, uint64_t terms_per_stream, uint64_t* out) {
    const size_t nz = orc_zeta_size(p->space, p->space_max, p->space_quantization_step);
    double* zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p->theta, p->space, p->space_max, p->space_quantization_step, zetas);
    if (terms_per_anchor == 0) terms_per_anchor = 1;
    for (uint32_t gi = 0; gi < n_streams; ++gi) {
        uint64_t s[4];
        orc_rng_seed(seed + stream_offset + gi, s);
        orc_anchor an;
        for (uint64_t j = 0; j < terms_per_stream; ++j) {
            orc_term t;
            if (j % terms_per_anchor == 0) anchor_valid(g, s, &an);
            orc_sample_partner(g, p, zetas, cooling, &an, s, &t);
            uint64_t* o = out + (j * (uint64_t)n_streams + gi) * 4;
            o[0] = t.ka; o[1] = t.kb; o[2] = t.off_a; o[3] = t.off_b;
        }
    }
    free(zetas);
}

/* Terms of the tile kernel (odgi_amd/csrc/pgsgd_tiles.hpp: sgd_tile_kernel / tile_trace_kernel): the tile with index
 * `tile` in the tile table runs the terms [cum*M/S_tot, (cum+n)*M/S_tot) of an iteration of M = n_terms terms; `lanes`
 * lanes work on it, lane l drawing the tile's terms l, l + lanes, ... from its own generator, seeded with
 * seed_base + epoch*0xd1342543de82ef95 + ((tile << 10) | l); a term draws its first step uniformly in the tile and its
 * partner by the reference rule. */
uint64_t orc_tile_terms(const orc_graph* g, const orc_params* p, uint64_t seed_base, uint64_t epoch, uint64_t n_terms,
                        uint64_t steps_total, uint64_t tile, uint32_t lanes, uint64_t t0, uint64_t cum, uint32_t n, uint32_t path,
                        int cooling, uint32_t share, uint64_t* out) {
    const size_t nz = orc_zeta_size(p->space, p->space_max, p->space_quantization_step);
    double* zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p->theta, p->space, p->space_max, p->space_quantization_step, zetas);
    const uint64_t term_begin = (uint64_t)(((unsigned __int128)cum * n_terms) / steps_total);
    const uint64_t term_end = (uint64_t)(((unsigned __int128)(cum + n) * n_terms) / steps_total);
    /* term q of the tile is drawn by lane (q - term_begin) % lanes in its trip (q - term_begin) / lanes: in term order, one
     * generator per lane, so that an odd lane sees its even neighbour's partner of the same trip */
    uint64_t* streams = (uint64_t*)malloc((size_t)lanes * 4 * sizeof(uint64_t));
    for (uint32_t l = 0; l < lanes; ++l) orc_rng_seed(seed_base + epoch * 0xd1342543de82ef95ull + ((tile << 10) | l), streams + 4 * (size_t)l);
    uint64_t lead = ORC_NO_PAIR;
    const uint32_t waves = (lanes + 63) / 64;
    uint64_t* cx = (uint64_t*)malloc((size_t)waves * sizeof(uint64_t));   /* the waves' coin streams, advanced in trip order */
    uint64_t* cw = (uint64_t*)calloc(waves, sizeof(uint64_t));
    int* coin = (int*)calloc(waves, sizeof(int));
    for (uint32_t w = 0; w < waves; ++w) cx[w] = seed_base + epoch * 0xd1342543de82ef95ull + ((tile << 10) | (uint64_t)(1023u - w));
    for (uint64_t q = term_begin; q < term_end; ++q) {
        const uint32_t lane = (uint32_t)((q - term_begin) % lanes);
        if (lane % 64 == 0) coin[lane / 64] = orc_tile_wave_coin_next(&cx[lane / 64], &cw[lane / 64], (q - term_begin) / lanes);
        orc_tile_pick cur;
        orc_tile_pick_first(g, p, cooling, coin[lane / 64], t0, n, path, streams + 4 * (size_t)lane, &cur);
        orc_term t;
        /* share: lanes per shared line of partner records in a uniform trip (4: quads, what sessions run; 2: the pairs of rounds 4-6; 0, 1: none) */
        const uint32_t r = share > 1 ? lane % share : 0;
        orc_tile_partner(g, p, zetas, &cur, r ? lead : ORC_NO_PAIR, r, streams + 4 * (size_t)lane, &t);
        if (!r) lead = cur.zipf ? ORC_NO_PAIR : t.kb;   /* (the group's lane 0 drew the term before the others, in the same trip) */
        uint64_t* o = out + (q - term_begin) * 4;
        o[0] = t.ka; o[1] = t.kb; o[2] = t.off_a; o[3] = t.off_b;
    }
    free(streams); free(cx); free(cw); free(coin);
    free(zetas);
    return term_end - term_begin;
}

/* iteration control shared by the serialised stream runs: path_sgd_layout.cpp:120-163 with exact
 * iteration lengths (min_term_updates terms each) instead of the 1 ms polling. */
typedef struct stream_run {
    const orc_graph* g; const orc_params* p; double* zetas; double* etas;
    uint64_t* states; uint32_t n_streams;
} stream_run;

static void stream_run_init(stream_run* r, const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset) {
    r->g = g; r->p = p; r->n_streams = n_streams;
    const size_t nz = orc_zeta_size(p->space, p->space_max, p->space_quantization_step);
    r->zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p->theta, p->space, p->space_max, p->space_quantization_step, r->zetas);
    r->etas = (double*)malloc((p->iter_max + 1) * sizeof(double));
    orc_schedule(p, r->etas);
    r->states = (uint64_t*)malloc((size_t)n_streams * 4 * sizeof(uint64_t));
    for (uint32_t i = 0; i < n_streams; ++i) orc_rng_seed(seed + stream_offset + i, r->states + 4 * (size_t)i);
}

static void stream_run_free(stream_run* r) { free(r->zetas); free(r->etas); free(r->states); }

static int has_multistep_path(const orc_graph* g) {
    for (uint64_t pi = 0; pi < g->n_paths; ++pi)
        if (g->path_first[pi + 1] - g->path_first[pi] > 1) return 1;
    return 0;
}

/* group q of an iteration (terms q*m .. q*m+m-1) belongs to stream q % n_streams; groups run in order */
void orc_layout_streams_f32(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, int hogwild_stores, uint32_t terms_per_anchor,
                            float* X, float* Y, double* last_delta_max) {
    if (last_delta_max) *last_delta_max = 0.0;
    if (!has_multistep_path(g)) return;                                /* :64-74 */
    stream_run r;
    stream_run_init(&r, g, p, seed, n_streams, stream_offset);
    const uint64_t m = terms_per_anchor ? terms_per_anchor : 1;
    const uint64_t first_cooling = (uint64_t)floor(p->cooling_start * (double)p->iter_max); /* :39 */
    const uint64_t n_groups = (p->min_term_updates + m - 1) / m;
    for (uint64_t iter = 0; iter < p->iter_max; ++iter) {
        const float eta = (float)r.etas[iter];
        const int cooling = iter >= first_cooling;
        float dmax = 0.0f;
        for (uint64_t q = 0; q < n_groups; ++q) {
            const uint64_t left = p->min_term_updates - q * m;
            const float da = group_f32(g, p, r.zetas, cooling, r.states + 4 * (size_t)(q % n_streams),
                                       (uint32_t)(left < m ? left : m), eta, hogwild_stores, X, Y);
            if (da > dmax) dmax = da;
        }
        if (last_delta_max) *last_delta_max = dmax;
        if (iter + 1 < p->iter_max && (double)dmax <= p->delta) break;  /* :142 */
    }
    stream_run_free(&r);
}

void orc_layout_streams_q32(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, int hogwild_stores, uint32_t terms_per_anchor,
                            double x_off, double y_off, double quanta_per_bp, float* X, float* Y,
                            double* last_delta_max, uint64_t* checksum) {
    if (last_delta_max) *last_delta_max = 0.0;
    const uint64_t n_ends = 2 * g->n_nodes;
    const float scale = (float)quanta_per_bp, inv_scale = (float)(1.0 / quanta_per_bp);
    uint64_t* W = (uint64_t*)malloc(n_ends * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_ends; ++i)
        W[i] = (uint64_t)q32_quantize(X[i], x_off, scale) | ((uint64_t)q32_quantize(Y[i], y_off, scale) << 32);
    if (checksum) {
        checksum[0] = checksum[1] = 0;
        for (uint64_t i = 0; i < n_ends; ++i) { checksum[0] += (uint32_t)W[i]; checksum[1] += W[i] >> 32; }
    }
    if (has_multistep_path(g)) {
        stream_run r;
        stream_run_init(&r, g, p, seed, n_streams, stream_offset);
        const uint64_t m = terms_per_anchor ? terms_per_anchor : 1;
        const uint64_t first_cooling = (uint64_t)floor(p->cooling_start * (double)p->iter_max);
        const uint64_t n_groups = (p->min_term_updates + m - 1) / m;
        for (uint64_t iter = 0; iter < p->iter_max; ++iter) {
            const float eta = (float)r.etas[iter];
            const int cooling = iter >= first_cooling;
            float dmax = 0.0f;
            for (uint64_t q = 0; q < n_groups; ++q) {
                const uint64_t left = p->min_term_updates - q * m;
                const float da = group_q32(g, p, r.zetas, cooling, r.states + 4 * (size_t)(q % n_streams),
                                           (uint32_t)(left < m ? left : m), eta, hogwild_stores, scale, inv_scale, W);
                if (da > dmax) dmax = da;
            }
            if (last_delta_max) *last_delta_max = dmax;
            if (iter + 1 < p->iter_max && (double)dmax <= p->delta) break;
        }
        stream_run_free(&r);
    }
    if (checksum) {
        checksum[2] = checksum[3] = 0;
        for (uint64_t i = 0; i < n_ends; ++i) { checksum[2] += (uint32_t)W[i]; checksum[3] += W[i] >> 32; }
    }
    for (uint64_t i = 0; i < n_ends; ++i) {
        X[i] = (float)(x_off + (double)(uint32_t)W[i] * (double)inv_scale);
        Y[i] = (float)(y_off + (double)(uint32_t)(W[i] >> 32) * (double)inv_scale);
    }
    free(W);
}

void orc_layout_streams_f64(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, double* X, double* Y) {
    if (!has_multistep_path(g)) return;
    stream_run r;
    stream_run_init(&r, g, p, seed, n_streams, stream_offset);
    const uint64_t first_cooling = (uint64_t)floor(p->cooling_start * (double)p->iter_max);
    for (uint64_t iter = 0; iter < p->iter_max; ++iter) {
        const double eta = r.etas[iter];
        const int cooling = iter >= first_cooling;
        double dmax = 0.0;
        for (uint64_t t = 0; t < p->min_term_updates; ++t) {
            uint64_t* s = r.states + 4 * (size_t)(t % n_streams);
            orc_term term;
            sample_valid(g, p, r.zetas, cooling, s, &term);
            const double da = update_f64(g, &term, eta, X, Y);
            if (da > dmax) dmax = da;
        }
        if (iter + 1 < p->iter_max && dmax <= p->delta) break;
    }
    stream_run_free(&r);
}

/* Concurrency model of the device (not part of the reference): the n_streams terms of one round all
 * read the coordinates as they were before the round, and their displacements are summed — what
 * fp32 atomic adds do when n_streams lanes are in flight together.  Pessimistic: on the GPU lanes
 * drift apart and atomics land continuously.  Used on the CPU to bound n_streams per graph size. */
void orc_layout_batched_f64(const orc_graph* g, const orc_params* p, uint64_t seed,
                            uint32_t n_streams, uint32_t stream_offset, double* X, double* Y) {
    if (!has_multistep_path(g)) return;
    stream_run r;
    stream_run_init(&r, g, p, seed, n_streams, stream_offset);
    const uint64_t first_cooling = (uint64_t)floor(p->cooling_start * (double)p->iter_max);
    const uint64_t n_ends = 2 * g->n_nodes;
    double* DX = (double*)calloc(n_ends, sizeof(double));
    double* DY = (double*)calloc(n_ends, sizeof(double));
    uint64_t* touched = (uint64_t*)malloc((size_t)n_streams * 2 * sizeof(uint64_t));
    for (uint64_t iter = 0; iter < p->iter_max; ++iter) {
        const double eta = r.etas[iter];
        const int cooling = iter >= first_cooling;
        for (uint64_t t0 = 0; t0 < p->min_term_updates; t0 += n_streams) {
            const uint64_t nb = (p->min_term_updates - t0 < n_streams) ? p->min_term_updates - t0 : n_streams;
            for (uint64_t b = 0; b < nb; ++b) {
                uint64_t* s = r.states + 4 * (size_t)b;
                orc_term t;
                sample_valid(g, p, r.zetas, cooling, s, &t);
                double term_dist = fabs((double)t.pos_a - (double)t.pos_b);
                if (term_dist == 0) term_dist = 1e-9;
                double mu = eta * (1.0 / term_dist);
                if (mu > 1) mu = 1;
                const uint64_t i = 2 * (uint64_t)(g->step_handle[t.ka] >> 1) + t.off_a;
                const uint64_t j = 2 * (uint64_t)(g->step_handle[t.kb] >> 1) + t.off_b;
                double dx = X[i] - X[j];
                const double dy = Y[i] - Y[j];
                if (dx == 0) dx = 1e-9;
                const double mag = sqrt(dx * dx + dy * dy);
                const double rr = (mu * (mag - term_dist) / 2) / mag;
                DX[i] -= rr * dx; DY[i] -= rr * dy;
                DX[j] += rr * dx; DY[j] += rr * dy;
                touched[2 * b] = i; touched[2 * b + 1] = j;
            }
            for (uint64_t b = 0; b < 2 * nb; ++b) {
                const uint64_t e = touched[b];
                X[e] += DX[e]; Y[e] += DY[e];
                DX[e] = 0; DY[e] = 0;
            }
        }
    }
    free(DX); free(DY); free(touched);
    stream_run_free(&r);
}

/* ------------------------------------------------------------------------------------------- */
/* The reference as it is: Hogwild workers, 1 ms polling controller (path_sgd_layout.cpp:99-425). */
typedef struct hog_shared {
    const orc_graph* g; const orc_params* p;
    const double* zetas; const double* etas;
    double* X; double* Y;
    uint64_t first_cooling_iteration;
    uint64_t term_updates;   /* atomic */
    double eta;              /* atomic */
    int cooling;             /* atomic */
    double Delta_max;        /* atomic */
    int work_todo;           /* atomic */
    uint64_t iteration;
    uint64_t total_terms;    /* atomic: sum of all worker counts */
    double max_seconds;
    struct timespec t0;
    /* convergence curve (test infrastructure): after iteration snap_iters[k] the controller copies X,Y while the
     * workers keep running, exactly like the reference's snapshot thread (path_sgd_layout.cpp:379-408) */
    uint64_t n_snap; const uint64_t* snap_iters; double* snapX; double* snapY; uint64_t snaps_taken;
} hog_shared;

typedef struct hog_worker { hog_shared* sh; uint64_t tid; } hog_worker;

static inline double ld_f64(const double* p) { double v; __atomic_load(p, &v, __ATOMIC_SEQ_CST); return v; }
static inline void st_f64(double* p, double v) { __atomic_store(p, &v, __ATOMIC_SEQ_CST); }

static double seconds_since(const struct timespec* t0) {
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0->tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0->tv_nsec);
}

static void* hog_checker(void* arg) {                                   /* :120-163 */
    hog_shared* sh = (hog_shared*)arg;
    const struct timespec ms = {0, 1000000};
    while (__atomic_load_n(&sh->work_todo, __ATOMIC_SEQ_CST)) {
        if (__atomic_load_n(&sh->term_updates, __ATOMIC_SEQ_CST) > sh->p->min_term_updates) {
            sh->iteration++;
            for (uint64_t k = 0; k < sh->n_snap; ++k)
                if (sh->snap_iters[k] == sh->iteration) {
                    const uint64_t n_ends = 2 * sh->g->n_nodes;
                    for (uint64_t e = 0; e < n_ends; ++e) { sh->snapX[k * n_ends + e] = ld_f64(&sh->X[e]); sh->snapY[k * n_ends + e] = ld_f64(&sh->Y[e]); }
                    sh->snaps_taken++;
                }
            if (sh->iteration >= sh->p->iter_max) {
                __atomic_store_n(&sh->work_todo, 0, __ATOMIC_SEQ_CST);
            } else if (ld_f64(&sh->Delta_max) <= sh->p->delta) {
                __atomic_store_n(&sh->work_todo, 0, __ATOMIC_SEQ_CST);
            } else {
                st_f64(&sh->eta, sh->etas[sh->iteration]);
                st_f64(&sh->Delta_max, sh->p->delta);
                if (sh->iteration >= sh->first_cooling_iteration)
                    __atomic_store_n(&sh->cooling, 1, __ATOMIC_SEQ_CST);
            }
            __atomic_store_n(&sh->term_updates, (uint64_t)0, __ATOMIC_SEQ_CST);
        }
        if (sh->max_seconds > 0 && seconds_since(&sh->t0) > sh->max_seconds)
            __atomic_store_n(&sh->work_todo, 0, __ATOMIC_SEQ_CST);
        nanosleep(&ms, NULL);
    }
    return NULL;
}

static void* hog_work(void* arg) {                                      /* :165-377 */
    hog_worker* w = (hog_worker*)arg;
    hog_shared* sh = w->sh;
    const orc_graph* g = sh->g;
    uint64_t s[4];
    orc_rng_seed(9399220ull + w->tid, s);                               /* :168-169 */
    uint64_t term_updates_local = 0, total_local = 0;
    double* X = sh->X; double* Y = sh->Y;
    while (__atomic_load_n(&sh->work_todo, __ATOMIC_SEQ_CST)) {
        orc_term t;
        const int cooling = __atomic_load_n(&sh->cooling, __ATOMIC_SEQ_CST);
        if (!orc_sample_term(g, sh->p, sh->zetas, cooling, s, &t)) continue;
        double term_dist = fabs((double)t.pos_a - (double)t.pos_b);
        if (term_dist == 0) term_dist = 1e-9;
        const double w_ij = 1.0 / term_dist;
        double mu = ld_f64(&sh->eta) * w_ij;
        if (mu > 1) mu = 1;
        const double d_ij = term_dist;
        const uint64_t i = 2 * (uint64_t)(g->step_handle[t.ka] >> 1) + t.off_a;
        const uint64_t j = 2 * (uint64_t)(g->step_handle[t.kb] >> 1) + t.off_b;
        double dx = ld_f64(&X[i]) - ld_f64(&X[j]);
        const double dy = ld_f64(&Y[i]) - ld_f64(&Y[j]);
        if (dx == 0) dx = 1e-9;
        const double mag = sqrt(dx * dx + dy * dy);
        const double Delta = mu * (mag - d_ij) / 2;
        const double Delta_abs = fabs(Delta);
        while (Delta_abs > ld_f64(&sh->Delta_max)) st_f64(&sh->Delta_max, Delta_abs); /* :345-347 */
        const double r = Delta / mag;
        const double r_x = r * dx, r_y = r * dy;
        st_f64(&X[i], ld_f64(&X[i]) - r_x);                            /* :360-363 load-then-store */
        st_f64(&Y[i], ld_f64(&Y[i]) - r_y);
        st_f64(&X[j], ld_f64(&X[j]) + r_x);
        st_f64(&Y[j], ld_f64(&Y[j]) + r_y);
        term_updates_local++;
        if (term_updates_local >= 1000) {                              /* :367-374 */
            __atomic_fetch_add(&sh->term_updates, term_updates_local, __ATOMIC_SEQ_CST);
            total_local += term_updates_local;
            term_updates_local = 0;
        }
    }
    total_local += term_updates_local;
    __atomic_fetch_add(&sh->total_terms, total_local, __ATOMIC_SEQ_CST);
    return NULL;
}

void orc_layout_hogwild(const orc_graph* g, const orc_params* p, uint32_t nthreads, double max_seconds,
                        double* X, double* Y, orc_hogwild_stats* st) {
    orc_layout_hogwild_curve(g, p, nthreads, max_seconds, X, Y, st, 0, NULL, NULL, NULL);
}

/* the same run, also copying the coordinates after the iterations listed in snap_iters[n_snap] (1-based counts of
 * finished iterations) into snapX/snapY [n_snap][2N] */
void orc_layout_hogwild_curve(const orc_graph* g, const orc_params* p, uint32_t nthreads, double max_seconds,
                              double* X, double* Y, orc_hogwild_stats* st,
                              uint64_t n_snap, const uint64_t* snap_iters, double* snapX, double* snapY) {
    if (st) { st->terms = 0; st->iterations = 0; st->seconds = 0; }
    if (!has_multistep_path(g) || nthreads == 0) return;
    const size_t nz = orc_zeta_size(p->space, p->space_max, p->space_quantization_step);
    double* zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p->theta, p->space, p->space_max, p->space_quantization_step, zetas);
    double* etas = (double*)malloc((p->iter_max + 1) * sizeof(double));
    orc_schedule(p, etas);
    hog_shared sh;
    memset(&sh, 0, sizeof sh);
    sh.g = g; sh.p = p; sh.zetas = zetas; sh.etas = etas; sh.X = X; sh.Y = Y;
    sh.first_cooling_iteration = (uint64_t)floor(p->cooling_start * (double)p->iter_max);
    sh.eta = etas[0];
    sh.cooling = 0; sh.Delta_max = 0; sh.work_todo = 1; sh.iteration = 0;
    sh.max_seconds = max_seconds;
    sh.n_snap = n_snap; sh.snap_iters = snap_iters; sh.snapX = snapX; sh.snapY = snapY;
    clock_gettime(CLOCK_MONOTONIC, &sh.t0);
    pthread_t checker;
    pthread_t* th = (pthread_t*)malloc(nthreads * sizeof(pthread_t));
    hog_worker* ws = (hog_worker*)malloc(nthreads * sizeof(hog_worker));
    pthread_create(&checker, NULL, hog_checker, &sh);
    for (uint32_t t = 0; t < nthreads; ++t) {
        ws[t].sh = &sh; ws[t].tid = t;
        pthread_create(&th[t], NULL, hog_work, &ws[t]);
    }
    for (uint32_t t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    const double secs = seconds_since(&sh.t0);
    pthread_join(checker, NULL);
    if (st) { st->terms = sh.total_terms; st->iterations = sh.iteration; st->seconds = secs; }
    free(th); free(ws); free(zetas); free(etas);
}

/* ------------------------------------------------------------------------------------------- */
/* quality metrics */
double orc_path_stress_sampled(const orc_graph* g, const double* X, const double* Y,
                               uint64_t n_pairs, uint64_t seed) {
    if (!has_multistep_path(g)) return 0.0;
    /* evaluation sampler: the SGD sampler itself in its non-cooling mode (half Zipf, half uniform),
     * with default-like Zipf parameters derived from the graph, on an independent stream */
    uint64_t max_steps = 0;
    for (uint64_t pi = 0; pi < g->n_paths; ++pi) {
        const uint64_t c = g->path_first[pi + 1] - g->path_first[pi];
        if (c > max_steps) max_steps = c;
    }
    orc_params p;
    memset(&p, 0, sizeof p);
    p.theta = 0.99; p.space = max_steps; p.space_max = 1000; p.space_quantization_step = 100;
    const size_t nz = orc_zeta_size(p.space, p.space_max, p.space_quantization_step);
    double* zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p.theta, p.space, p.space_max, p.space_quantization_step, zetas);
    uint64_t s[4];
    orc_rng_seed(seed, s);
    double acc = 0.0;
    uint64_t cnt = 0;
    for (uint64_t n = 0; n < n_pairs; ++n) {
        orc_term t;
        if (!orc_sample_term(g, &p, zetas, 0, s, &t)) continue;
        const double d = fabs((double)t.pos_a - (double)t.pos_b);
        if (d == 0) continue;
        const uint64_t i = 2 * (uint64_t)(g->step_handle[t.ka] >> 1) + t.off_a;
        const uint64_t j = 2 * (uint64_t)(g->step_handle[t.kb] >> 1) + t.off_b;
        const double dx = X[i] - X[j], dy = Y[i] - Y[j];
        const double e = (sqrt(dx * dx + dy * dy) - d) / d;
        acc += e * e;
        cnt++;
    }
    free(zetas);
    return cnt ? acc / (double)cnt : 0.0;
}

/* The near part of the EXPECTATION of orc_path_stress_sampled, term by term: every draw the warm-iteration sampler can make with a
 * Zipf jump of at most zmax steps (path_sgd_layout.cpp:178-269: first step uniform over all steps, Zipf branch 1/2, direction coin —
 * forced at a path's ends, :206-207 — jump z with probability z^-theta / H(min(space, room)), end coins 1/2 each), weighted by its
 * probability.  Written from the sampler's side (first step, direction, jump), where the product's pgsgd_path_stress_near walks
 * unordered pairs: two formulations of one sum, compared in tests/test_host_logic.py.
 * num[(z-1)*4 + 2*flip_first + flip_partner] with the pair oriented (earlier step, later step); *zero_mass: probability of d = 0. */
typedef struct { const orc_graph* g; const double *X, *Y, *H, *zw; uint32_t zmax, tid, nt; uint64_t max_steps; double *num, *mass, zero; } near_work;
static void* near_run(void* arg) {
    near_work* w = (near_work*)arg;
    const orc_graph* g = w->g;
    const double per_step = 1.0 / (double)g->n_steps;
    for (uint64_t pi = 0; pi < g->n_paths; ++pi) {
        const uint64_t b = g->path_first[pi], c = g->path_first[pi + 1] - b;
        if (c < 2) continue;
        for (uint64_t s = c * w->tid / w->nt; s < c * (w->tid + 1) / w->nt; ++s)
            for (int back = 0; back < 2; ++back) {
                /* :206: the jump goes back when (s > 0 and the coin says so) or s is the path's last step */
                const double p_dir = back ? (s == c - 1 ? 1.0 : s > 0 ? 0.5 : 0.0) : (s == 0 ? 1.0 : s < c - 1 ? 0.5 : 0.0);
                if (p_dir == 0.0) continue;
                const uint64_t room = back ? s : c - s - 1;
                const uint64_t jump = room < w->max_steps ? room : w->max_steps;
                for (uint32_t z = 1; z <= w->zmax && z <= jump; ++z) {
                    const uint64_t ka = b + s, kb = back ? ka - z : ka + z;
                    const double pr = per_step * 0.5 * p_dir * w->zw[z] / w->H[jump] * 0.25;
                    for (uint32_t fa = 0; fa < 2; ++fa)
                        for (uint32_t fb = 0; fb < 2; ++fb) {
                            const uint32_t ha = g->step_handle[ka], hb = g->step_handle[kb];
                            const double pa = (double)g->step_pos[ka] + (fa ? (double)g->node_len[ha >> 1] : 0.0);
                            const double pb = (double)g->step_pos[kb] + (fb ? (double)g->node_len[hb >> 1] : 0.0);
                            const double d = fabs(pa - pb);
                            if (d == 0) { w->zero += pr; continue; }
                            const uint64_t i = (uint64_t)(ha ^ fa), j = (uint64_t)(hb ^ fb);
                            const double dx = w->X[i] - w->X[j], dy = w->Y[i] - w->Y[j];
                            const double e = (sqrt(dx * dx + dy * dy) - d) / d;
                            const size_t cl = (size_t)(z - 1) * 4 + (back ? 2 * fb + fa : 2 * fa + fb);   /* (earlier step, later step) */
                            w->num[cl] += pr * e * e;
                            w->mass[cl] += pr;
                        }
                }
            }
    }
    return NULL;
}
void orc_path_stress_near(const orc_graph* g, const double* X, const double* Y, uint32_t zmax, double theta, uint32_t nthreads,
                          double* num, double* mass, double* zero_mass) {
    uint64_t max_steps = 0;
    for (uint64_t pi = 0; pi < g->n_paths; ++pi) {
        const uint64_t c = g->path_first[pi + 1] - g->path_first[pi];
        if (c > max_steps) max_steps = c;
    }
    double* H = (double*)calloc(max_steps + 1, sizeof(double));
    double* zw = (double*)calloc(zmax + 1, sizeof(double));
    for (uint64_t n = 1; n <= max_steps; ++n) H[n] = H[n - 1] + pow((double)n, -theta);
    for (uint32_t z = 1; z <= zmax; ++z) zw[z] = pow((double)z, -theta);
    if (nthreads < 1) nthreads = 1;
    near_work* ws = (near_work*)calloc(nthreads, sizeof(near_work));
    pthread_t* th = (pthread_t*)calloc(nthreads, sizeof(pthread_t));
    for (uint32_t t = 0; t < nthreads; ++t) {
        near_work* w = &ws[t];
        w->g = g; w->X = X; w->Y = Y; w->H = H; w->zw = zw; w->zmax = zmax; w->tid = t; w->nt = nthreads; w->max_steps = max_steps;
        w->num = (double*)calloc((size_t)zmax * 4, sizeof(double));
        w->mass = (double*)calloc((size_t)zmax * 4, sizeof(double));
        pthread_create(&th[t], NULL, near_run, w);
    }
    for (size_t i = 0; i < (size_t)zmax * 4; ++i) num[i] = mass[i] = 0.0;
    *zero_mass = 0.0;
    for (uint32_t t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        for (size_t i = 0; i < (size_t)zmax * 4; ++i) { num[i] += ws[t].num[i]; mass[i] += ws[t].mass[i]; }
        *zero_mass += ws[t].zero;
        free(ws[t].num); free(ws[t].mass);
    }
    free(ws); free(th); free(H); free(zw);
}

double orc_path_stress_exhaustive(const orc_graph* g, const double* X, const double* Y) {
    double acc = 0.0;
    uint64_t cnt = 0;
    for (uint64_t pi = 0; pi < g->n_paths; ++pi) {
        const uint64_t b = g->path_first[pi], e = g->path_first[pi + 1];
        for (uint64_t a = b; a < e; ++a) {
            const uint64_t ia = g->step_handle[a]; /* 2*rank + is_rev = step-start end */
            for (uint64_t c = a + 1; c < e; ++c) {
                const double d = (double)g->step_pos[c] - (double)g->step_pos[a];
                if (d == 0) continue;
                const uint64_t ic = g->step_handle[c];
                const double dx = X[ia] - X[ic], dy = Y[ia] - Y[ic];
                const double r = (sqrt(dx * dx + dy * dy) - d) / d;
                acc += r * r;
                cnt++;
            }
        }
    }
    return cnt ? acc / (double)cnt : 0.0;
}

void orc_path_distance(const orc_graph* g, const double* X, const double* Y, double* per_node, double* per_bp) {
    double sum2d = 0.0;
    uint64_t nodes = 0, bp = 0;
    for (uint64_t pi = 0; pi < g->n_paths; ++pi) {
        const uint64_t b = g->path_first[pi], e = g->path_first[pi + 1];
        for (uint64_t k = b; k < e; ++k) {
            const uint32_t h = g->step_handle[k];
            if (k + 1 < e) {
                const uint32_t i = g->step_handle[k + 1];
                const double dx = X[h] - X[i], dy = Y[h] - Y[i]; /* 2*rank+bit == handle */
                sum2d += sqrt(dx * dx + dy * dy);
            }
            nodes++;
            bp += g->node_len[h >> 1];
        }
    }
    if (per_node) *per_node = nodes ? sum2d / (double)nodes : 0.0;
    if (per_bp) *per_bp = bp ? sum2d / (double)bp : 0.0;
}


/* =========================================================================================== */
/* 1D path-guided SGD of `odgi sort -Y` (reference src/algorithms/path_sgd.cpp:12-500): the sibling of
 * the layout path — same sampler, one coordinate per node, no end choice.  Differences restated here:
 *   - X[n] starts at the cumulative node length in graph order (:67-73);
 *   - the Zipf draw uses adj_theta, which the controller sets to 0.001 once cooling starts (:195,246),
 *     while the zeta cache stays the one of the user's theta (:127-137);
 *   - a term of path distance 0 is dropped without being counted (:320-323);
 *   - the controller stops when iteration > iter_max and cools when iteration > first_cooling (:181,194),
 *     so iter_max + 1 learning rates etas[0..iter_max] are used. */
static int orc_sample_term_1d(const orc_graph* g, const orc_params* p, const double* zetas, int cooling,
                              uint64_t s[4], orc_term* t) {
    orc_anchor a;
    if (!orc_sample_anchor(g, s, &a)) return 0;
    const double theta = cooling ? 0.001 : p->theta;                   /* adj_theta, :195,246 */
    const uint64_t s_rank = a.s_rank, cnt = a.cnt;
    uint64_t b_rank;
    if (cooling || orc_flip(s)) {                                      /* :245 */
        if ((s_rank > 0 && orc_flip(s)) || s_rank == cnt - 1) {        /* :247 */
            const uint64_t jump_space = p->space < s_rank ? p->space : s_rank;
            uint64_t space = jump_space;
            if (jump_space > p->space_max) space = p->space_max + (jump_space - p->space_max) / p->space_quantization_step + 1;
            b_rank = s_rank - orc_zipf(s, jump_space, theta, zetas[space]);
        } else {
            const uint64_t rest = cnt - s_rank - 1;
            const uint64_t jump_space = p->space < rest ? p->space : rest;
            uint64_t space = jump_space;
            if (jump_space > p->space_max) space = p->space_max + (jump_space - p->space_max) / p->space_quantization_step + 1;
            b_rank = s_rank + orc_zipf(s, jump_space, theta, zetas[space]);
        }
    } else {
        b_rank = orc_uniform_u64(s, cnt);                              /* :275-277 */
    }
    t->ka = a.k;
    t->kb = a.pstart + b_rank;
    t->pos_a = g->step_pos[t->ka];                                     /* :312-313 */
    t->pos_b = g->step_pos[t->kb];
    t->off_a = t->off_b = 0;
    t->dither = 0;
    return t->pos_a != t->pos_b;                                       /* :320-323 */
}

static inline double update_1d_f64(const orc_graph* g, const orc_term* t, double eta, double* X) {
    const double term_dist = fabs((double)t->pos_a - (double)t->pos_b);
    double mu = eta * (1.0 / term_dist);
    if (mu > 1) mu = 1;
    const uint64_t i = g->step_handle[t->ka] >> 1, j = g->step_handle[t->kb] >> 1;
    double dx = X[i] - X[j];
    if (dx == 0) dx = 1e-9;
    const double mag = fabs(dx);
    const double Delta = mu * (mag - term_dist) / 2;
    const double r_x = (Delta / mag) * dx;
    X[i] = X[i] - r_x;
    X[j] = X[j] + r_x;
    return fabs(Delta);
}

void orc_sort_initial(const orc_graph* g, double* X) {                 /* :67-73 */
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; ++i) { X[i] = (double)len; len += g->node_len[i]; }
}

void orc_sort_trace_terms(const orc_graph* g, const orc_params* p, uint64_t seed, uint32_t n_streams, uint32_t stream_offset,
                          int cooling, uint64_t terms_per_stream, uint64_t* out) {
    const size_t nz = orc_zeta_size(p->space, p->space_max, p->space_quantization_step);
    double* zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p->theta, p->space, p->space_max, p->space_quantization_step, zetas);
    for (uint32_t gi = 0; gi < n_streams; ++gi) {
        uint64_t s[4];
        orc_rng_seed(seed + stream_offset + gi, s);
        for (uint64_t j = 0; j < terms_per_stream; ++j) {
            orc_term t;
            while (!orc_sample_term_1d(g, p, zetas, cooling, s, &t)) { }
            uint64_t* o = out + (j * (uint64_t)n_streams + gi) * 2;
            o[0] = t.ka; o[1] = t.kb;
        }
    }
    free(zetas);
}

/* mirror of the device's 1D kernel (odgi_amd/csrc/pgsgd_kernels.hpp: sort_iteration_kernel): one signed
 * 64-bit fixed-point word per node (x = q / quanta_per_bp), fp64 arithmetic, round-to-nearest steps.
 * Streams serialised like the 2D mirrors; bit-exact for n_streams == 1. */
void orc_sort_streams(const orc_graph* g, const orc_params* p, uint64_t seed, uint32_t n_streams, uint32_t stream_offset,
                      double quanta_per_bp, const uint8_t* frozen, double* X, double* last_delta_max) {
    if (last_delta_max) *last_delta_max = 0.0;
    if (!has_multistep_path(g)) return;
    stream_run r;
    stream_run_init(&r, g, p, seed, n_streams, stream_offset);
    int64_t* W = (int64_t*)malloc(g->n_nodes * sizeof(int64_t));
    for (uint64_t i = 0; i < g->n_nodes; ++i) W[i] = (int64_t)llrint(X[i] * quanta_per_bp);
    const double inv = 1.0 / quanta_per_bp;
    const uint64_t first_cooling = (uint64_t)floor(p->cooling_start * (double)p->iter_max);
    for (uint64_t iter = 0; iter <= p->iter_max; ++iter) {             /* etas[0..iter_max] */
        const double eta = r.etas[iter];
        const int cooling = iter > first_cooling;
        double dmax = 0.0;
        for (uint64_t t = 0; t < p->min_term_updates; ++t) {
            uint64_t* s = r.states + 4 * (size_t)(t % n_streams);
            orc_term term;
            int move_i = 1, move_j = 1;
            for (;;) {   /* target nodes are looked at before the distance (:289-301 then :320-323) */
                const int distinct = orc_sample_term_1d(g, p, r.zetas, cooling, s, &term);
                if (frozen) {
                    move_i = !frozen[g->step_handle[term.ka] >> 1];
                    move_j = !frozen[g->step_handle[term.kb] >> 1];
                }
                if ((!move_i && !move_j) || distinct) break;
            }
            if (!move_i && !move_j) continue;   /* counted, nothing to move */
            const double term_dist = fabs((double)term.pos_a - (double)term.pos_b);
            double mu = eta * (1.0 / term_dist);
            if (mu > 1) mu = 1;
            const uint64_t i = g->step_handle[term.ka] >> 1, j = g->step_handle[term.kb] >> 1;
            double dx = (double)(W[i] - W[j]) * inv;
            if (dx == 0) dx = 1e-9;
            const double mag = fabs(dx);
            const double Delta = mu * (mag - term_dist) / 2;
            const double r_x = (Delta / mag) * dx;
            const int64_t dq = (int64_t)llrint(r_x * quanta_per_bp);
            if (move_j) W[j] += dq;   /* partner first, then the first step: the device's order */
            if (move_i) W[i] -= dq;
            if (fabs(Delta) > dmax) dmax = fabs(Delta);
        }
        if (last_delta_max) *last_delta_max = dmax;
        if (iter < p->iter_max && dmax <= p->delta) break;             /* :183 */
    }
    for (uint64_t i = 0; i < g->n_nodes; ++i) X[i] = (double)W[i] * inv;
    free(W);
    stream_run_free(&r);
}

/* the reference as it is (Hogwild workers + 1 ms controller), path_sgd.cpp:158-452 */
typedef struct sort_shared {
    const orc_graph* g; const orc_params* p; const double* zetas; const double* etas; double* X; const uint8_t* frozen;
    uint64_t first_cooling_iteration, term_updates, iteration, total_terms;
    double eta, Delta_max, max_seconds; int cooling, work_todo; struct timespec t0;
} sort_shared;
typedef struct sort_worker { sort_shared* sh; uint64_t tid; } sort_worker;

static void* sort_checker(void* arg) {
    sort_shared* sh = (sort_shared*)arg;
    const struct timespec ms = {0, 1000000};
    while (__atomic_load_n(&sh->work_todo, __ATOMIC_SEQ_CST)) {
        if (__atomic_load_n(&sh->term_updates, __ATOMIC_SEQ_CST) > sh->p->min_term_updates) {
            sh->iteration++;
            if (sh->iteration > sh->p->iter_max) {                     /* :181 */
                __atomic_store_n(&sh->work_todo, 0, __ATOMIC_SEQ_CST);
            } else if (ld_f64(&sh->Delta_max) <= sh->p->delta) {
                __atomic_store_n(&sh->work_todo, 0, __ATOMIC_SEQ_CST);
            } else {
                st_f64(&sh->eta, sh->etas[sh->iteration]);
                st_f64(&sh->Delta_max, sh->p->delta);
                if (sh->iteration > sh->first_cooling_iteration) __atomic_store_n(&sh->cooling, 1, __ATOMIC_SEQ_CST);  /* :194 */
            }
            __atomic_store_n(&sh->term_updates, (uint64_t)0, __ATOMIC_SEQ_CST);
        }
        if (sh->max_seconds > 0 && seconds_since(&sh->t0) > sh->max_seconds) __atomic_store_n(&sh->work_todo, 0, __ATOMIC_SEQ_CST);
        nanosleep(&ms, NULL);
    }
    return NULL;
}

static void* sort_work(void* arg) {
    sort_worker* w = (sort_worker*)arg;
    sort_shared* sh = w->sh;
    const orc_graph* g = sh->g;
    uint64_t s[4];
    orc_rng_seed(9399220ull + w->tid, s);
    uint64_t local = 0, total = 0;
    double* X = sh->X;
    while (__atomic_load_n(&sh->work_todo, __ATOMIC_SEQ_CST)) {
        orc_term t;
        const int cooling = __atomic_load_n(&sh->cooling, __ATOMIC_SEQ_CST);
        const int distinct = orc_sample_term_1d(g, sh->p, sh->zetas, cooling, s, &t);
        const uint64_t i = g->step_handle[t.ka] >> 1, j = g->step_handle[t.kb] >> 1;
        const int move_i = !(sh->frozen && sh->frozen[i]), move_j = !(sh->frozen && sh->frozen[j]);   /* :289-296 */
        if (!move_i && !move_j) {                                                                   /* :297-301 */
            if (++local >= 1000) { __atomic_fetch_add(&sh->term_updates, local, __ATOMIC_SEQ_CST); total += local; local = 0; }
            continue;
        }
        if (!distinct) continue;                                                                    /* :320-323 */
        const double term_dist = fabs((double)t.pos_a - (double)t.pos_b);
        double mu = ld_f64(&sh->eta) * (1.0 / term_dist);
        if (mu > 1) mu = 1;
        double dx = ld_f64(&X[i]) - ld_f64(&X[j]);
        if (dx == 0) dx = 1e-9;
        const double mag = fabs(dx);
        const double Delta = mu * (mag - term_dist) / 2;
        const double Delta_abs = fabs(Delta);
        while (Delta_abs > ld_f64(&sh->Delta_max)) st_f64(&sh->Delta_max, Delta_abs);
        const double r_x = (Delta / mag) * dx;
        if (move_i) st_f64(&X[i], ld_f64(&X[i]) - r_x);                                            /* :392-397 */
        if (move_j) st_f64(&X[j], ld_f64(&X[j]) + r_x);
        if (++local >= 1000) { __atomic_fetch_add(&sh->term_updates, local, __ATOMIC_SEQ_CST); total += local; local = 0; }
    }
    total += local;
    __atomic_fetch_add(&sh->total_terms, total, __ATOMIC_SEQ_CST);
    return NULL;
}

void orc_sort_hogwild(const orc_graph* g, const orc_params* p, uint32_t nthreads, double max_seconds, const uint8_t* frozen, double* X,
                      orc_hogwild_stats* st) {
    if (st) { st->terms = 0; st->iterations = 0; st->seconds = 0; }
    if (!has_multistep_path(g) || nthreads == 0) return;
    const size_t nz = orc_zeta_size(p->space, p->space_max, p->space_quantization_step);
    double* zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p->theta, p->space, p->space_max, p->space_quantization_step, zetas);
    double* etas = (double*)malloc((p->iter_max + 1) * sizeof(double));
    orc_schedule(p, etas);
    sort_shared sh;
    memset(&sh, 0, sizeof sh);
    sh.g = g; sh.p = p; sh.zetas = zetas; sh.etas = etas; sh.X = X; sh.frozen = frozen;
    sh.first_cooling_iteration = (uint64_t)floor(p->cooling_start * (double)p->iter_max);
    sh.eta = etas[0]; sh.work_todo = 1; sh.max_seconds = max_seconds;
    clock_gettime(CLOCK_MONOTONIC, &sh.t0);
    pthread_t checker;
    pthread_t* th = (pthread_t*)malloc(nthreads * sizeof(pthread_t));
    sort_worker* ws = (sort_worker*)malloc(nthreads * sizeof(sort_worker));
    pthread_create(&checker, NULL, sort_checker, &sh);
    for (uint32_t t = 0; t < nthreads; ++t) { ws[t].sh = &sh; ws[t].tid = t; pthread_create(&th[t], NULL, sort_work, &ws[t]); }
    for (uint32_t t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    const double secs = seconds_since(&sh.t0);
    pthread_join(checker, NULL);
    if (st) { st->terms = sh.total_terms; st->iterations = sh.iteration; st->seconds = secs; }
    free(th); free(ws); free(zetas); free(etas);
}

/* 1D path stress: mean ((|x_a - x_b| - d)/d)^2 over pairs drawn by the sampler's non-cooling mode */
double orc_sort_stress(const orc_graph* g, const double* X, uint64_t n_pairs, uint64_t seed) {
    if (!has_multistep_path(g)) return 0.0;
    uint64_t max_steps = 0;
    for (uint64_t pi = 0; pi < g->n_paths; ++pi) { const uint64_t c = g->path_first[pi + 1] - g->path_first[pi]; if (c > max_steps) max_steps = c; }
    orc_params p; memset(&p, 0, sizeof p);
    p.theta = 0.99; p.space = max_steps; p.space_max = 100; p.space_quantization_step = 100;
    double* zetas = (double*)malloc(orc_zeta_size(p.space, p.space_max, p.space_quantization_step) * sizeof(double));
    orc_zetas(p.theta, p.space, p.space_max, p.space_quantization_step, zetas);
    uint64_t s[4]; orc_rng_seed(seed, s);
    double acc = 0.0; uint64_t cnt = 0;
    for (uint64_t n = 0; n < n_pairs; ++n) {
        orc_term t;
        if (!orc_sample_term_1d(g, &p, zetas, 0, s, &t)) continue;
        const double d = fabs((double)t.pos_a - (double)t.pos_b);
        const double e = (fabs(X[g->step_handle[t.ka] >> 1] - X[g->step_handle[t.kb] >> 1]) - d) / d;
        acc += e * e; cnt++;
    }
    free(zetas);
    return cnt ? acc / (double)cnt : 0.0;
}

/* ---- mirror of the region-exclusive tile kernel (odgi_amd/csrc/pgsgd_tiles.hpp: sgd_tile_kernel) run by ONE
 * workgroup with ONE lane per tile (PGSGD_TILE_GRID=1, PGSGD_TILE_LANES=1): a sequential program.
 * Per iteration two launches (even regions, odd regions); a launch takes its work items in order; an item
 * stages its window of 4R coordinate words (private copy `win`, and `orig` as staged), runs its tiles' terms in
 * term order — ends inside the window read and move the private copy, ends outside read and move the global
 * words, a term with a partner outside the window is limited to mu <= far_mu_cap, a step that rounds to no
 * quantum sends nothing — and adds win - orig back.  far_mu_cap of a launch is 1 / (far pulls per node end)
 * counted in the previous launch of the same parity (first iteration: 0.75 * 0.5 * n_terms / 2N assumed).
 * The tile table and the work items are the product's (pgsgd_session_tile_table / _tile_items); the test
 * checks them separately as an exact partition of steps and terms. */
/* Relaxation of the far pulls of a launch (pgsgd_tiles.hpp: tile_far_relax): together they amount to this
 * fraction of a projection — one, less in the first five iterations (rounds 3-5: half, reached as slowly: orc_tile_far_relax_r5).
 * ORC_FAR_RELAX="r0,r1,..." (experiments, tools/cpu_transient.py) overrides the schedule:
 * iteration i uses r_i, iterations past the list the last value. */
static float orc_tile_far_relax_r5(uint64_t iter) { return iter < 2 ? 0.1f : iter < 5 ? 0.1f * (float)iter : 0.5f; }   /* 0.1 0.1 0.2 0.3 0.4 0.5 ... */
float orc_tile_far_relax(uint64_t iter) {
    const char* e = getenv("ORC_FAR_RELAX");
    if (e) {
        float v = 0.5f;
        for (uint64_t i = 0; *e; ++i) {
            char* end;
            v = strtof(e, &end);
            if (end == e) break;
            e = *end == ',' ? end + 1 : end;
            if (i == iter) break;
        }
        return v;
    }
    return iter < 2 ? 0.2f : iter < 5 ? 0.2f * (float)iter : 1.0f;   /* 0.2 0.2 0.4 0.6 0.8 1.0 ... (pgsgd_tiles.hpp: tile_far_relax) */
}

static inline float displacement_capped_f32(float eta, uint64_t pos_a, uint64_t pos_b, float dx, float dy, float mu_cap,
                                            float* r_x, float* r_y) {
    const int64_t diff = (int64_t)pos_a - (int64_t)pos_b;
    float d = (float)(uint64_t)(diff < 0 ? -diff : diff);
    if (d == 0.0f) d = 1e-9f;
    const float w = 1.0f / d;
    float mu = eta * w;
    if (mu > mu_cap) mu = mu_cap;
    if (dx == 0.0f) dx = 1e-9f;
    const float dx2 = dx * dx;
    const float dy2 = dy * dy;
    const float mag = sqrtf(dx2 + dy2);
    const float Delta = (mu * (mag - d)) / 2.0f;
    const float r = Delta / mag;
    *r_x = r * dx;
    *r_y = r * dy;
    return fabsf(Delta);
}

/* The launch order of an iteration (pgsgd_session_iteration_part): [deliver the far pulls the launch before collected]
 * [snapshot, first launch of the iteration only] [launch of the even regions] [deliver] [launch of the odd regions]; what the
 * last launch collected is delivered before the first launch of the next iteration, or by pgsgd_session_flush when the run ends.
 * policy (0 = what the product ships; the others are round 2's choices, kept for tools/cpu_transient.py and its pinned vectors):
 *   ORC_TILE_DRAIN_AFTER    a launch's far pulls are delivered right after it (an iteration then ends with the arrival of a
 *                           launch's worth of far pulls instead of with window-local terms)
 *   ORC_TILE_TWO_SNAPSHOTS  the coordinate snapshot is refreshed before both launches of a warm iteration
 *   ORC_TILE_CONSTANT_RELAX the far pulls of a launch amount to half a projection in every iteration (no gentle start)
 *   ORC_TILE_SNAPSHOT_PASS  partners outside a window are read from a snapshot of ALL coordinates taken once per iteration
 *                           (sharded sessions; round 2), not from words the tiles rewrite for their own steps
 *   ORC_TILE_NO_FLUSH       return the coordinates as a snapshot between iterations sees them: without the pulls still waiting
 *   ORC_TILE_LANE_COIN      the Zipf/uniform coin of a warm term is bit 31 of the lane's own word (rounds 2 and 3), not the wave's
 *   ORC_TILE_NO_PAIRS       every lane keeps its own uniform partner (rounds 2 and 3; PGSGD_FLAG_NO_PARTNER_PAIRS)
 *   ORC_TILE_PAIRS          the lanes share uniform partners in pairs (rounds 4-6) instead of quads
 *   ORC_TILE_DRAIN_BESIDE   every region colour has its own outbox and a launch's far pulls are delivered right before the SAME colour's
 *                           next launch — a launch later than by default — from the sixth iteration on (the first five's arrive before
 *                           the very next launch, as by default): what a session does whose drain runs on a second stream beside the
 *                           other colour's launch (pgsgd_session::async_drain: schedules of 30 iterations and more)
 *   ORC_TILE_RELAX_R5       the far pulls' relaxation of rounds 3-5: 0.1 0.1 0.2 0.3 0.4 then half a projection (round 6: 0.2 ... 0.8 then one)
 * stop_after: run only the first stop_after iterations of the schedule (0 = all). */
void orc_tile_layout_q32_ex(const orc_graph* g, const orc_params* p, uint64_t seed_base,
                         uint64_t n_tiles, const uint64_t* t0, const uint64_t* cum, const uint32_t* tn, const uint32_t* tpath,
                         const uint32_t* tlanes, uint64_t steps_total, uint64_t n_items, uint64_t n_first, const uint32_t* tile_begin,
                         const uint32_t* tile_end, const uint32_t* win0, const uint32_t* local, uint32_t region,
                         double x_off, double y_off, double quanta_per_bp, float* X, float* Y,
                         double* last_delta_max, uint64_t* checksum, uint64_t* far_terms, uint32_t policy, uint64_t stop_after);

void orc_tile_layout_q32(const orc_graph* g, const orc_params* p, uint64_t seed_base,
                         uint64_t n_tiles, const uint64_t* t0, const uint64_t* cum, const uint32_t* tn, const uint32_t* tpath,
                         const uint32_t* tlanes, uint64_t steps_total, uint64_t n_items, uint64_t n_first, const uint32_t* tile_begin,
                         const uint32_t* tile_end, const uint32_t* win0, const uint32_t* local, uint32_t region,
                         double x_off, double y_off, double quanta_per_bp, float* X, float* Y,
                         double* last_delta_max, uint64_t* checksum, uint64_t* far_terms) {
    orc_tile_layout_q32_ex(g, p, seed_base, n_tiles, t0, cum, tn, tpath, tlanes, steps_total, n_items, n_first, tile_begin, tile_end, win0, local,
                           region, x_off, y_off, quanta_per_bp, X, Y, last_delta_max, checksum, far_terms, 0, 0);
}

void orc_tile_layout_q32_ex(const orc_graph* g, const orc_params* p, uint64_t seed_base,
                         uint64_t n_tiles, const uint64_t* t0, const uint64_t* cum, const uint32_t* tn, const uint32_t* tpath,
                         const uint32_t* tlanes, uint64_t steps_total, uint64_t n_items, uint64_t n_first, const uint32_t* tile_begin,
                         const uint32_t* tile_end, const uint32_t* win0, const uint32_t* local, uint32_t region,
                         double x_off, double y_off, double quanta_per_bp, float* X, float* Y,
                         double* last_delta_max, uint64_t* checksum, uint64_t* far_terms, uint32_t policy, uint64_t stop_after) {
    (void)n_tiles;
    const int drain_first = !(policy & ORC_TILE_DRAIN_AFTER), one_snapshot = !(policy & ORC_TILE_TWO_SNAPSHOTS);
    const int beside = (policy & ORC_TILE_DRAIN_BESIDE) != 0;
    int pending[2] = {0, 0};   /* a launch's far pulls wait in the outbox (ORC_TILE_DRAIN_BESIDE: in its colour's) */
    int urgent[2] = {0, 0};    /* ORC_TILE_DRAIN_BESIDE: ... and arrive before the very next launch when theirs was a warm one */
    if (last_delta_max) *last_delta_max = 0.0;
    const uint64_t n_ends = 2 * g->n_nodes;
    const float scale = (float)quanta_per_bp, inv_scale = (float)(1.0 / quanta_per_bp);
    uint64_t* W = (uint64_t*)malloc(n_ends * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_ends; ++i)
        W[i] = (uint64_t)q32_quantize(X[i], x_off, scale) | ((uint64_t)q32_quantize(Y[i], y_off, scale) << 32);
    if (checksum) {
        checksum[0] = checksum[1] = 0;
        for (uint64_t i = 0; i < n_ends; ++i) { checksum[0] += (uint32_t)W[i]; checksum[1] += W[i] >> 32; }
    }
    const size_t nz = orc_zeta_size(p->space, p->space_max, p->space_quantization_step);
    double* zetas = (double*)malloc(nz * sizeof(double));
    orc_zetas(p->theta, p->space, p->space_max, p->space_quantization_step, zetas);
    double* etas = (double*)malloc((p->iter_max + 1) * sizeof(double));
    orc_schedule(p, etas);
    const uint32_t win_words = 4 * region;
    uint64_t* win = (uint64_t*)malloc(win_words * sizeof(uint64_t));
    uint64_t* orig = (uint64_t*)malloc(win_words * sizeof(uint64_t));
    /* what a launch sees of node ends outside a window: their words when the iteration began (snapshot_kernel); what it
     * adds to them: collected (the outbox) and applied when the launch is over (far_drain_kernel) */
    uint64_t* snap = (uint64_t*)malloc(n_ends * sizeof(uint64_t));
    uint64_t* outboxes = (uint64_t*)calloc(2 * n_ends, sizeof(uint64_t));   /* (one per region colour; only ORC_TILE_DRAIN_BESIDE uses the second) */
    /* the snapshot words a partner outside the window is read from: a tile rewrites those of its OWN steps when its terms are
     * done (from the window), so a partner is seen as its tile last left it, this iteration or the one before; the words of
     * all steps are taken from the coordinates only when the run starts.  ORC_TILE_SNAPSHOT_PASS: a pass over all node
     * ends once per iteration instead (what a sharded session does, and round 2 did) */
    const int tile_snap = !(policy & ORC_TILE_SNAPSHOT_PASS);
    uint64_t* snapw = tile_snap ? (uint64_t*)malloc(2 * g->n_steps * sizeof(uint64_t)) : NULL;
    if (tile_snap)
        for (uint64_t k = 0; k < g->n_steps; ++k) { snapw[2 * k] = W[g->step_handle[k]]; snapw[2 * k + 1] = W[g->step_handle[k] ^ 1u]; }
    const uint64_t first_cooling = (uint64_t)floor(p->cooling_start * (double)p->iter_max);
    const uint64_t n_terms = p->min_term_updates;
    float far_cap[2];
    {
        const double h = 0.75 * 0.5 * (double)n_terms / (double)n_ends;
        far_cap[0] = far_cap[1] = h > 1.0 ? (float)(1.0 / h) : 1.0f;
    }
    uint64_t far_total = 0;
    for (uint64_t iter = 0; iter < p->iter_max; ++iter) {
        const float eta = (float)etas[iter];
        const int cooling = iter >= first_cooling;
        const uint64_t epoch = iter + 1;
        const float far_relax = (policy & ORC_TILE_CONSTANT_RELAX) ? 0.5f : (policy & ORC_TILE_RELAX_R5) ? orc_tile_far_relax_r5(iter) : orc_tile_far_relax(iter);
        float dmax = 0.0f;
        uint64_t far_count[2] = {0, 0};
        int snap_taken = 0;
        for (int colour = 0; colour < 2; ++colour) {
            const uint64_t ib = colour ? n_first : 0, ie = colour ? n_items : n_first;
            if (ib == ie) continue;
            /* snapshot before every launch of a warm iteration, before the first launch of a cooling one */
            /* what waits arrives now: the last launch's pulls, whatever its colour — or (ORC_TILE_DRAIN_BESIDE) this colour's last launch's */
            for (int c = 0; c < 2; ++c)
                if (pending[c] && (!beside || c == colour || urgent[c])) {
                    uint64_t* ob = outboxes + (size_t)c * n_ends;
                    for (uint64_t i = 0; i < n_ends; ++i) { W[i] += ob[i]; ob[i] = 0; }
                    pending[c] = 0;
                }
            uint64_t* outbox = outboxes + (beside ? (size_t)colour * n_ends : 0);
            if (!snap_taken || (!cooling && !one_snapshot)) { memcpy(snap, W, n_ends * sizeof(uint64_t)); snap_taken = 1; }
            for (uint64_t it = ib; it < ie; ++it) {
                const uint64_t wbase = 2 * (uint64_t)win0[it];
                if (local[it])
                    for (uint32_t i = 0; i < win_words; ++i) win[i] = orig[i] = wbase + i < n_ends ? W[wbase + i] : 0;
                for (uint32_t ti = tile_begin[it]; ti < tile_end[it]; ++ti) {
                    const uint64_t term_begin = (uint64_t)(((unsigned __int128)cum[ti] * n_terms) / steps_total);
                    const uint64_t term_end = (uint64_t)(((unsigned __int128)(cum[ti] + tn[ti]) * n_terms) / steps_total);
                    /* lane l of the tile's lanes draws terms l, l + lanes, ... from its own stream; the mirror runs the
                     * terms in term order (with one lane: the lane's order) */
                    const uint32_t lanes = tlanes[ti];
                    uint64_t* streams = (uint64_t*)malloc((size_t)lanes * 4 * sizeof(uint64_t));
                    for (uint32_t l = 0; l < lanes; ++l)
                        orc_rng_seed(seed_base + epoch * 0xd1342543de82ef95ull + (((uint64_t)ti << 10) | l), streams + 4 * (size_t)l);
                    uint64_t pair_lead = ORC_NO_PAIR;
                    for (uint64_t q = term_begin; q < term_end; ++q) {
                        const uint32_t lane = (uint32_t)((q - term_begin) % lanes);
                        uint64_t* s = streams + 4 * (size_t)lane;
                        orc_tile_pick cur;
                        const int coin = (policy & ORC_TILE_LANE_COIN) ? -1 : orc_tile_wave_coin(seed_base, epoch, ti, lane / 64, (q - term_begin) / lanes);
                        orc_tile_pick_first(g, p, cooling, coin, t0[ti], tn[ti], tpath[ti], s, &cur);
                        orc_term t;
                        const uint32_t share = (policy & ORC_TILE_NO_PAIRS) ? 1u : (policy & ORC_TILE_PAIRS) ? 2u : 4u, r = lane % share;
                        orc_tile_partner(g, p, zetas, &cur, r ? pair_lead : ORC_NO_PAIR, r, s, &t);
                        if (!r) pair_lead = cur.zipf ? ORC_NO_PAIR : t.kb;   /* (the group's lane 0 drew the term before the others, in the same trip) */
                        const uint64_t ea = 2 * (uint64_t)(g->step_handle[t.ka] >> 1) + t.off_a;
                        const uint64_t eb = 2 * (uint64_t)(g->step_handle[t.kb] >> 1) + t.off_b;
                        const int in_a = local[it] && ea >= wbase && ea - wbase < win_words;
                        const int in_b = local[it] && eb >= wbase && eb - wbase < win_words;
                        /* a partner step inside the tile comes with no snapshot (its record is the tile's LDS copy): a
                         * window-less tile then reads the partner's word where it reads the first end's, in global memory */
                        const int b_in_tile = t.kb - t0[ti] < (uint64_t)tn[ti];
                        const uint64_t wa = in_a ? win[ea - wbase] : W[ea];
                        /* experiments (tools/cpu_transient.py): 0x800 far partners are read live, 0x1000 no learning-rate cap of far terms
                         * once cooling has started, 0x2000 no cap at all, 0x4000 far moves are applied at once (no outbox) */
                        const uint64_t wb = in_b ? win[eb - wbase] : (b_in_tile || (policy & 0x800u)) ? W[eb]
                                          : tile_snap ? snapw[2 * t.kb + (eb != g->step_handle[t.kb] ? 1 : 0)] : snap[eb];
                        const float dx = (float)((int64_t)(uint32_t)wa - (int64_t)(uint32_t)wb) * inv_scale;
                        const float dy = (float)((int64_t)(wa >> 32) - (int64_t)(wb >> 32)) * inv_scale;
                        float r_x, r_y;
                        /* far pulls are capped at kFarRelax / h: half a projection per launch in total (pgsgd_tiles.hpp) */
                        const int uncapped = (policy & 0x2000u) || ((policy & 0x1000u) && cooling);
                        const float da = displacement_capped_f32(eta, t.pos_a, t.pos_b, dx, dy, (in_b || uncapped) ? 1.0f : far_cap[colour] * far_relax, &r_x, &r_y);
                        if (da > dmax) dmax = da;
                        const float ux = (float)(t.dither >> 14) * (1.0f / 16384.0f);
                        const float uy = (float)(t.dither & 0x3fffu) * (1.0f / 16384.0f);
                        float fx = r_x * scale;
                        float fy = r_y * scale;
                        fx = fminf(fmaxf(fx + ux, -2147483520.0f), 2147483520.0f);
                        fy = fminf(fmaxf(fy + uy, -2147483520.0f), 2147483520.0f);
                        const int64_t qx = (int64_t)floorf(fx), qy = (int64_t)floorf(fy);
                        if ((qx | qy) == 0) continue;
                        if (!in_b) far_count[colour]++;
                        const uint64_t delta = (uint64_t)qx + ((uint64_t)qy << 32);
                        if (!in_b && (policy & 0x100u)) continue;                 /* experiment: far terms do nothing */
                        if (in_b) win[eb - wbase] += delta; else if (policy & 0x4000u) W[eb] += delta; else if (!(policy & 0x200u)) outbox[eb] += delta;
                        if (!in_b && (policy & 0x400u)) continue;                 /* experiment: far terms move only the partner */
                        if (in_a) win[ea - wbase] -= delta; else outbox[ea] -= delta;
                    }
                    free(streams);
                    if (tile_snap)
                        for (uint64_t k = t0[ti]; k < t0[ti] + tn[ti]; ++k)
                            for (int e = 0; e < 2; ++e) {
                                const uint64_t end = (uint64_t)g->step_handle[k] ^ (uint64_t)e;
                                snapw[2 * k + e] = (local[it] && end >= wbase && end - wbase < win_words) ? win[end - wbase] : W[end];
                            }
                }
                if (local[it])   /* the window's only writer since it was staged: plain stores */
                    for (uint32_t i = 0; i < win_words; ++i)
                        if (wbase + i < n_ends) W[wbase + i] = win[i];
            }
            urgent[colour] = iter < 5;   /* (pgsgd_tiles.hpp: kFarGentleIterations) */
            if (drain_first) pending[beside ? colour : 0] = 1;
            else for (uint64_t i = 0; i < n_ends; ++i) { W[i] += outbox[i]; outbox[i] = 0; }
        }
        for (int colour = 0; colour < 2; ++colour) {
            const double h = (double)far_count[colour] / (double)n_ends;
            far_cap[colour] = h > 1.0 ? (float)(1.0 / h) : 1.0f;
            far_total += far_count[colour];
        }
        if (last_delta_max) *last_delta_max = dmax;
        if (iter + 1 < p->iter_max && (double)dmax <= p->delta) break;
        if (stop_after && iter + 1 >= stop_after) break;
    }
    if (!(policy & ORC_TILE_NO_FLUSH))
        for (int c = 0; c < 2; ++c)
            if (pending[c]) {
                uint64_t* ob = outboxes + (size_t)c * n_ends;
                for (uint64_t i = 0; i < n_ends; ++i) { W[i] += ob[i]; ob[i] = 0; }
            }
    if (far_terms) *far_terms = far_total;
    if (checksum) {
        checksum[2] = checksum[3] = 0;
        for (uint64_t i = 0; i < n_ends; ++i) { checksum[2] += (uint32_t)W[i]; checksum[3] += W[i] >> 32; }
    }
    for (uint64_t i = 0; i < n_ends; ++i) {
        X[i] = (float)(x_off + (double)(uint32_t)W[i] * (double)inv_scale);
        Y[i] = (float)(y_off + (double)(uint32_t)(W[i] >> 32) * (double)inv_scale);
    }
    free(W); free(zetas); free(etas); free(win); free(orig); free(snap); free(outboxes); free(snapw);
}
