// quality.cpp — layout quality figures reported by the CLI and the benchmark.
// The reference has no 2D stress metric; `odgi stats -s` with a layout
// (src/subcommand/stats_main.cpp:667-716, 2D branch) is restated as pgsgd_path_distance, and the
// sampled path stress is the SGD objective evaluated on pairs drawn by the SGD sampler itself.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "pgsgd_internal.hpp"
#include "pgsgd_math.hpp"

extern "C" int pgsgd_path_stress(const pgsgd_graph_view* g, const double* X, const double* Y, uint64_t n_pairs,
                                 uint64_t seed, double* stress) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!X || !Y || !stress) return PGSGD_E_INVALID;
    *stress = 0.0;
    uint64_t max_steps = 0;
    for (uint64_t p = 0; p < g->n_paths; ++p) max_steps = std::max(max_steps, g->path_first[p + 1] - g->path_first[p]);
    if (max_steps < 2) return PGSGD_OK;
    const uint64_t space = max_steps, space_max = 1000, quant = 100;
    std::vector<double> zetas(pgsgd_zeta_table_size(space, space_max, quant));
    rc = pgsgd_zeta_table(0.99, space, space_max, quant, zetas.data(), zetas.size());
    if (rc) return rc;
    pgsgd::ZipfConst zc;
    zc.init(0.99);
    pgsgd::Xoshiro256Plus rng;
    rng.seed(seed);
    double acc = 0.0;
    uint64_t cnt = 0;
    const pgsgd::StepIndex ix(g);
    for (uint64_t n = 0; n < n_pairs; ++n) {
        // the SGD sampler in its non-cooling mode: half Zipf partners, half uniform partners
        const uint64_t k = pgsgd::uniform_below(rng, g->n_steps);
        const uint64_t p = ix.path[k];
        const uint64_t pstart = g->path_first[p], c = g->path_first[p + 1] - pstart;
        if (c == 1) continue;
        const uint64_t s_rank = k - pstart;
        uint64_t b_rank;
        if (pgsgd::coin(rng)) {
            const bool back = (s_rank > 0 && pgsgd::coin(rng)) || s_rank == c - 1;
            const uint64_t room = back ? s_rank : c - s_rank - 1;
            const uint64_t jump = std::min(space, room);
            const uint64_t z = pgsgd::zipf(rng, zc, jump, zetas[pgsgd::zeta_index(jump, space_max, quant)]);
            b_rank = back ? s_rank - z : s_rank + z;
        } else {
            b_rank = pgsgd::uniform_below(rng, c);
        }
        const uint64_t kb = pstart + b_rank;
        const uint32_t ha = g->step_handle[k], hb = g->step_handle[kb];
        uint64_t pa = ix.pos[k], pb = ix.pos[kb];
        uint32_t oa = ha & 1u, ob = hb & 1u;
        if (pgsgd::coin(rng)) { pa += g->node_len[ha >> 1]; oa ^= 1u; }
        if (pgsgd::coin(rng)) { pb += g->node_len[hb >> 1]; ob ^= 1u; }
        const double d = std::fabs((double)pa - (double)pb);
        if (d == 0) continue;
        const uint64_t i = (uint64_t)(ha & ~1u) | oa, j = (uint64_t)(hb & ~1u) | ob;
        const double dx = X[i] - X[j], dy = Y[i] - Y[j];
        const double e = (std::sqrt(dx * dx + dy * dy) - d) / d;
        acc += e * e;
        ++cnt;
    }
    *stress = cnt ? acc / (double)cnt : 0.0;
    return PGSGD_OK;
}

// The NEAR part of the sampled path stress, exhaustively: every pair of steps at most zmax steps apart on a path, all four
// end choices, weighted by the probability with which the warm-iteration sampler (and pgsgd_path_stress) draws it.  The sampled
// stress is a mean of squared RELATIVE errors: pairs a few bp apart dominate it and its distribution is so heavy-tailed that a
// hundred pairs of a 2e6-pair sample carry a third of the figure (profiles/r06/NOTES.md) — two layouts of equal quality can differ
// by 10 % on one fixed sample.  This sum has no sampling error.
//   num[(z-1)*4 + 2*flip_a + flip_b] = sum over pairs (k, k+z) of P(pair) * ((|p_a - p_b| - d) / d)^2      (d > 0)
//   mass[...]                        = sum of P(pair) over the same pairs                                  (d > 0)
//   zero_mass                        = probability of drawing a pair with d = 0 among these (skipped by the metric)
// P(pair): first step uniform over all steps, Zipf branch 1/2, direction 1/2 (1 at a path's ends), jump z with probability
// z^-theta / H(min(space, room)), either end choice 1/2 — from both sides of the pair.  sum(num) / (1 - P(d = 0) overall) is
// the near pairs' exact contribution to the expected sampled stress.
// hist_step[m] / hist_rank[m] (optional): the same numerator by (rank of the earlier step in its path) % mod_step and by
// (node rank of the earlier step) % mod_rank — where along tiles / regions the stress sits.
extern "C" int pgsgd_path_stress_near(const pgsgd_graph_view* g, const double* X, const double* Y, uint32_t zmax, double theta, uint32_t threads,
                                      double* num, double* mass, double* zero_mass, uint32_t mod_step, double* hist_step, uint32_t mod_rank, double* hist_rank) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!X || !Y || !num || !mass || zmax == 0 || zmax > 64) return PGSGD_E_INVALID;
    uint64_t max_steps = 0;
    for (uint64_t p = 0; p < g->n_paths; ++p) max_steps = std::max(max_steps, g->path_first[p + 1] - g->path_first[p]);
    std::vector<double> H(max_steps + 1, 0.0);   // H[n] = sum_{k<=n} k^-theta (the sampler's space is the longest path)
    for (uint64_t n = 1; n <= max_steps; ++n) H[n] = H[n - 1] + std::pow((double)n, -theta);
    std::vector<double> zw(zmax + 1, 0.0);
    for (uint32_t z = 1; z <= zmax; ++z) zw[z] = std::pow((double)z, -theta);
    const uint32_t T = std::max<uint32_t>(1, threads);
    const size_t nc = (size_t)zmax * 4;
    std::vector<std::vector<double>> t_num(T, std::vector<double>(nc, 0.0)), t_mass(T, std::vector<double>(nc, 0.0)), t_hs(T), t_hr(T);
    std::vector<double> t_zero(T, 0.0);
    for (uint32_t t = 0; t < T; ++t) { if (hist_step) t_hs[t].assign(mod_step, 0.0); if (hist_rank) t_hr[t].assign(mod_rank, 0.0); }
    const double inv_steps = 1.0 / (double)g->n_steps;
    const pgsgd::StepIndex ix(g);
    auto work = [&](uint32_t t) {
        for (uint64_t p = 0; p < g->n_paths; ++p) {
            const uint64_t b = g->path_first[p], c = g->path_first[p + 1] - b;
            if (c < 2) continue;
            const uint64_t lo = c * t / T, hi = c * (t + 1) / T;
            for (uint64_t s = lo; s < hi; ++s) {
                const uint64_t k = b + s;
                const uint32_t ha = g->step_handle[k];
                const double la = (double)g->node_len[ha >> 1], pa0 = (double)ix.pos[k];
                // k as the first step, jumping forward: direction coin 1/2, or certain at the path's first step
                const double fwd_a = (s == 0 ? 1.0 : s == c - 1 ? 0.0 : 0.5) / H[std::min<uint64_t>(max_steps, c - s - 1)];
                for (uint32_t z = 1; z <= zmax && s + z < c; ++z) {
                    const uint64_t kb = k + z, sb = s + z;
                    const uint32_t hb = g->step_handle[kb];
                    const double lb = (double)g->node_len[hb >> 1], pb0 = (double)ix.pos[kb];
                    const double back_b = (sb == c - 1 ? 1.0 : 0.5) / H[std::min<uint64_t>(max_steps, sb)];   // kb as the first step, jumping back
                    const double w = inv_steps * 0.5 * zw[z] * (fwd_a + back_b) * 0.25;
                    for (uint32_t f = 0; f < 4; ++f) {
                        const uint32_t fa = f >> 1, fb = f & 1u;
                        const double d = std::fabs((pa0 + (fa ? la : 0.0)) - (pb0 + (fb ? lb : 0.0)));
                        if (d == 0) { t_zero[t] += w; continue; }
                        const uint64_t i = (uint64_t)(ha ^ fa), j = (uint64_t)(hb ^ fb);
                        const double dx = X[i] - X[j], dy = Y[i] - Y[j];
                        const double e = (std::sqrt(dx * dx + dy * dy) - d) / d;
                        const size_t cl = (size_t)(z - 1) * 4 + f;
                        t_num[t][cl] += w * e * e;
                        t_mass[t][cl] += w;
                        if (hist_step) t_hs[t][s % mod_step] += w * e * e;
                        if (hist_rank) t_hr[t][(ha >> 1) % mod_rank] += w * e * e;
                    }
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    for (size_t i = 0; i < nc; ++i) { num[i] = mass[i] = 0.0; for (uint32_t t = 0; t < T; ++t) { num[i] += t_num[t][i]; mass[i] += t_mass[t][i]; } }
    if (zero_mass) { *zero_mass = 0.0; for (uint32_t t = 0; t < T; ++t) *zero_mass += t_zero[t]; }
    if (hist_step) for (uint32_t m = 0; m < mod_step; ++m) { hist_step[m] = 0.0; for (uint32_t t = 0; t < T; ++t) hist_step[m] += t_hs[t][m]; }
    if (hist_rank) for (uint32_t m = 0; m < mod_rank; ++m) { hist_rank[m] = 0.0; for (uint32_t t = 0; t < T; ++t) hist_rank[m] += t_hr[t][m]; }
    return PGSGD_OK;
}

extern "C" int pgsgd_path_distance(const pgsgd_graph_view* g, const double* X, const double* Y, double* per_node, double* per_bp) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!X || !Y) return PGSGD_E_INVALID;
    double sum2d = 0.0;
    uint64_t nodes = 0, bp = 0;
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        const uint64_t b = g->path_first[p], e = g->path_first[p + 1];
        for (uint64_t k = b; k < e; ++k) {
            const uint32_t h = g->step_handle[k];
            if (k + 1 < e) {
                const uint32_t i = g->step_handle[k + 1];
                const double dx = X[h] - X[i], dy = Y[h] - Y[i];
                sum2d += std::sqrt(dx * dx + dy * dy);
            }
            ++nodes;
            bp += g->node_len[h >> 1];
        }
    }
    if (per_node) *per_node = nodes ? sum2d / (double)nodes : 0.0;
    if (per_bp) *per_bp = bp ? sum2d / (double)bp : 0.0;
    return PGSGD_OK;
}


// ---- 1D (odgi sort -Y) helpers ----------------------------------------------------------------
// Node order from 1D positions: by position, ties by handle (path_sgd.cpp:641-650; the component
// key of the reference's comparator reads a vector it has just cleared, :587, so only these two act).
extern "C" int pgsgd_sort_order(uint64_t n_nodes, const double* X, uint64_t* order) {
    pgsgd::clear_error();
    if (!X || !order) return PGSGD_E_INVALID;
    for (uint64_t i = 0; i < n_nodes; ++i) order[i] = i;
    std::sort(order, order + n_nodes, [&](uint64_t a, uint64_t b) { return X[a] < X[b] || (X[a] == X[b] && a < b); });
    return PGSGD_OK;
}

// path_sgd.cpp:552-587: weakly connected components, ranked by the average id of their nodes (ties: discovery order)
extern "C" int pgsgd_sort_component_ranks(uint64_t n_nodes, const uint64_t* edges, uint64_t n_edges, uint32_t* comp_rank_of_node) {
    pgsgd::clear_error();
    if (!comp_rank_of_node || (n_edges && !edges)) return PGSGD_E_INVALID;
    const int64_t n_comp = pgsgd_weak_components(n_nodes, edges, n_edges, comp_rank_of_node);
    if (n_comp < 0) return (int)n_comp;
    std::vector<uint64_t> id_sum((size_t)n_comp, 0), size((size_t)n_comp, 0);
    for (uint64_t i = 0; i < n_nodes; ++i) { id_sum[comp_rank_of_node[i]] += i + 1; size[comp_rank_of_node[i]]++; }
    std::vector<std::pair<double, uint64_t>> by_avg;
    for (int64_t c = 0; c < n_comp; ++c) by_avg.emplace_back((double)id_sum[c] / (double)size[c], (uint64_t)c);
    std::sort(by_avg.begin(), by_avg.end());
    std::vector<uint32_t> rank((size_t)n_comp);
    for (size_t r = 0; r < by_avg.size(); ++r) rank[by_avg[r].second] = (uint32_t)r;
    for (uint64_t i = 0; i < n_nodes; ++i) comp_rank_of_node[i] = rank[comp_rank_of_node[i]];
    return PGSGD_OK;
}

// path_sgd.cpp:641-650: by component, then position, then handle.  (The reference's comparator lets the handle
// tie-break apply across components when two positions are exactly equal; positions of distinct nodes differ,
// so the strict lexicographic order is used.  Its component key reads a vector it cleared at :587, whose
// storage still holds the values.)
extern "C" int pgsgd_sort_order_components(uint64_t n_nodes, const double* X, const uint32_t* comp_rank_of_node, uint64_t* order) {
    pgsgd::clear_error();
    if (!X || !order) return PGSGD_E_INVALID;
    if (!comp_rank_of_node) return pgsgd_sort_order(n_nodes, X, order);
    for (uint64_t i = 0; i < n_nodes; ++i) order[i] = i;
    std::sort(order, order + n_nodes, [&](uint64_t a, uint64_t b) {
        if (comp_rank_of_node[a] != comp_rank_of_node[b]) return comp_rank_of_node[a] < comp_rank_of_node[b];
        return X[a] < X[b] || (X[a] == X[b] && a < b);
    });
    return PGSGD_OK;
}

// path_sgd.cpp:651-672 (`odgi sort --path-sgd-layout`): the sorted nodes' 1D layout as a .lay file:
// X = (position, position + node length) of each node in `order`, Y = 0.
extern "C" int pgsgd_sort_write_lay(const pgsgd_graph_view* g, const double* X, const uint64_t* order, const char* path) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!X || !order || !path) return PGSGD_E_INVALID;
    std::vector<double> sx(2 * g->n_nodes), sy(2 * g->n_nodes, 0.0);
    for (uint64_t i = 0; i < g->n_nodes; ++i) {
        if (order[i] >= g->n_nodes) { pgsgd::set_error("order names a node outside the graph"); return PGSGD_E_INVALID; }
        sx[2 * i] = X[order[i]];
        sx[2 * i + 1] = X[order[i]] + (double)g->node_len[order[i]];
    }
    return pgsgd_write_lay(path, 2 * g->n_nodes, sx.data(), sy.data());
}

// reference start: X[rank] = cumulative node length in graph order (path_sgd.cpp:67-73)
extern "C" int pgsgd_sort_initial(const pgsgd_graph_view* g, double* X) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!X) return PGSGD_E_INVALID;
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; ++i) { X[i] = (double)len; len += g->node_len[i]; }
    return PGSGD_OK;
}

// 1D path stress: mean ((|x_a - x_b| - d)/d)^2 over pairs drawn by the sampler's non-cooling mode (step starts)
extern "C" int pgsgd_sort_stress(const pgsgd_graph_view* g, const double* X, uint64_t n_pairs, uint64_t seed, double* stress) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!X || !stress) return PGSGD_E_INVALID;
    *stress = 0.0;
    uint64_t max_steps = 0;
    for (uint64_t p = 0; p < g->n_paths; ++p) max_steps = std::max(max_steps, g->path_first[p + 1] - g->path_first[p]);
    if (max_steps < 2) return PGSGD_OK;
    const uint64_t space = max_steps, space_max = 100, quant = 100;
    std::vector<double> zetas(pgsgd_zeta_table_size(space, space_max, quant));
    rc = pgsgd_zeta_table(0.99, space, space_max, quant, zetas.data(), zetas.size());
    if (rc) return rc;
    pgsgd::ZipfConst zc;
    zc.init(0.99);
    pgsgd::Xoshiro256Plus rng;
    rng.seed(seed);
    double acc = 0.0;
    uint64_t cnt = 0;
    const pgsgd::StepIndex ix(g);
    for (uint64_t n = 0; n < n_pairs; ++n) {
        const uint64_t k = pgsgd::uniform_below(rng, g->n_steps);
        const uint64_t p = ix.path[k];
        const uint64_t pstart = g->path_first[p], c = g->path_first[p + 1] - pstart;
        if (c == 1) continue;
        const uint64_t s_rank = k - pstart;
        uint64_t b_rank;
        if (pgsgd::coin(rng)) {
            const bool back = (s_rank > 0 && pgsgd::coin(rng)) || s_rank == c - 1;
            const uint64_t room = back ? s_rank : c - s_rank - 1;
            const uint64_t jump = std::min(space, room);
            const uint64_t z = pgsgd::zipf(rng, zc, jump, zetas[pgsgd::zeta_index(jump, space_max, quant)]);
            b_rank = back ? s_rank - z : s_rank + z;
        } else {
            b_rank = pgsgd::uniform_below(rng, c);
        }
        const uint64_t kb = pstart + b_rank;
        const double d = std::fabs((double)ix.pos[k] - (double)ix.pos[kb]);
        if (d == 0) continue;
        const double e = (std::fabs(X[g->step_handle[k] >> 1] - X[g->step_handle[kb] >> 1]) - d) / d;
        acc += e * e;
        ++cnt;
    }
    *stress = cnt ? acc / (double)cnt : 0.0;
    return PGSGD_OK;
}
