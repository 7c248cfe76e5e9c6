// quality.cpp — layout quality figures reported by the CLI and the benchmark.
// The reference has no 2D stress metric; `odgi stats -s` with a layout
// (src/subcommand/stats_main.cpp:667-716, 2D branch) is restated as pgsgd_path_distance, and the
// sampled path stress is the SGD objective evaluated on pairs drawn by the SGD sampler itself.
#include <algorithm>
#include <cmath>
#include <cstring>
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
    for (uint64_t n = 0; n < n_pairs; ++n) {
        // the SGD sampler in its non-cooling mode: half Zipf partners, half uniform partners
        const uint64_t k = pgsgd::uniform_below(rng, g->n_steps);
        const uint64_t p = g->step_path[k];
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
        uint64_t pa = g->step_pos[k], pb = g->step_pos[kb];
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
    for (uint64_t n = 0; n < n_pairs; ++n) {
        const uint64_t k = pgsgd::uniform_below(rng, g->n_steps);
        const uint64_t p = g->step_path[k];
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
        const double d = std::fabs((double)g->step_pos[k] - (double)g->step_pos[kb]);
        if (d == 0) continue;
        const double e = (std::fabs(X[g->step_handle[k] >> 1] - X[g->step_handle[kb] >> 1]) - d) / d;
        acc += e * e;
        ++cnt;
    }
    *stress = cnt ? acc / (double)cnt : 0.0;
    return PGSGD_OK;
}
