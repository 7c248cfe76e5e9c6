// pgsgd_handlegraph.hpp — the reference-side shim: `path_linear_sgd_layout_gpu` with the reference's
// own signature (src/algorithms/path_sgd_layout.hpp:59-80), implemented over the C ABI of pgsgd.h.
//
// It is what replaces the body of the reference's `path_linear_sgd_layout_gpu`
// (src/algorithms/path_sgd_layout.cpp:470-503, which fills cuda::layout_config_t and calls
// cuda::gpu_layout): instead it lowers `graph` into the flat path-step index — the same walk the
// reference's CUDA host code does (src/cuda/layout.cu:325-410) — and calls pgsgd_layout_run.
//
// Header-only and duck-typed: libhandlegraph is not needed to compile it.  `Graph` must offer the
// PathHandleGraph calls used below (get_node_count, get_length, for_each_handle, for_each_path_handle,
// get_step_count, for_each_step_in_path, get_handle_of_step); handles must be odgi's packed integers (2*rank + is_reverse,
// handlegraph::number_bool_packing, pinned in-tree by src/algorithms/layout.cpp:76-79) reachable
// through an ADL `as_integer(handle)`.  `PathIndex` (xp::XP) is accepted for signature
// compatibility and not read: like the reference's CUDA route, the GPU path builds its own index.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "pgsgd.h"

namespace pgsgd {

// graph_t + paths -> pgsgd_graph_view (owned arrays)
struct lowered_graph {
    std::vector<uint32_t> node_len, step_path, step_handle;
    std::vector<uint64_t> path_first, step_pos;
    pgsgd_graph_view view() const {
        pgsgd_graph_view v;
        v.n_nodes = node_len.size();
        v.n_steps = step_handle.size();
        v.n_paths = path_first.size() - 1;
        v.node_len = node_len.data();
        v.path_first = path_first.data();
        v.step_path = step_path.data();
        v.step_handle = step_handle.data();
        v.step_pos = step_pos.data();
        return v;
    }
};

// Paths are independent, so they are walked on `n_threads` host threads (the reference's CUDA route does
// the same walk under OpenMP, src/cuda/layout.cu:371-410); the graph is only read.
template <class PathHandle, class Graph>
lowered_graph lower_graph(const Graph& graph, uint64_t n_threads = 1) {
    lowered_graph lg;
    lg.node_len.assign(graph.get_node_count(), 0);
    graph.for_each_handle([&](const auto& h) {
        lg.node_len[as_integer(h) >> 1] = (uint32_t)graph.get_length(h);  // src/odgi.cpp:65-71
    });
    std::vector<PathHandle> paths;  // path ids in creation order (odgi.cpp:261-270)
    graph.for_each_path_handle([&](const PathHandle& path) { paths.push_back(path); });
    lg.path_first.assign(paths.size() + 1, 0);
    for (size_t p = 0; p < paths.size(); ++p) lg.path_first[p + 1] = lg.path_first[p] + graph.get_step_count(paths[p]);
    lg.step_path.resize(lg.path_first.back());
    lg.step_handle.resize(lg.path_first.back());
    lg.step_pos.resize(lg.path_first.back());
    std::atomic<size_t> next{0};
    auto walk = [&]() {
        for (size_t p = next.fetch_add(1); p < paths.size(); p = next.fetch_add(1)) {
            uint64_t pos = 0, k = lg.path_first[p];
            graph.for_each_step_in_path(paths[p], [&](const auto& step) {
                const uint64_t hi = as_integer(graph.get_handle_of_step(step));
                lg.step_path[k] = (uint32_t)p;
                lg.step_handle[k] = (uint32_t)hi;
                lg.step_pos[k] = pos;  // xp.cpp:607-617
                pos += lg.node_len[hi >> 1];
                ++k;
            });
        }
    };
    std::vector<std::thread> th;
    for (uint64_t t = 1; t < n_threads && t < paths.size(); ++t) th.emplace_back(walk);
    walk();
    for (auto& t : th) t.join();
    return lg;
}

}  // namespace pgsgd

namespace odgi {
namespace algorithms {

// Same parameter list as the reference.  X, Y: 2*node_count doubles, pre-initialised, updated in
// place.  Errors print a message and exit(1), as the reference's GPU route does
// (src/cuda/layout.cu:6-13,320-323).
template <class Graph, class PathIndex, class PathHandle>
void path_linear_sgd_layout_gpu(const Graph& graph, const PathIndex& /*path_index*/,
                                const std::vector<PathHandle>& /*path_sgd_use_paths*/, const uint64_t& iter_max,
                                const uint64_t& iter_with_max_learning_rate, const uint64_t& min_term_updates,
                                const double& delta, const double& eps, const double& eta_max, const double& theta,
                                const uint64_t& space, const uint64_t& space_max, const uint64_t& space_quantization_step,
                                const double& cooling_start, const uint64_t& nthreads, const bool& progress,
                                const bool& snapshot, const std::string& snapshot_prefix,
                                std::vector<std::atomic<double>>& X, std::vector<std::atomic<double>>& Y) {
    const pgsgd::lowered_graph lg = pgsgd::lower_graph<PathHandle>(graph, nthreads);
    const pgsgd_graph_view view = lg.view();
    pgsgd_params p;
    if (pgsgd_params_defaults(&view, &p) != PGSGD_OK) {
        std::fprintf(stderr, "[odgi::path_linear_sgd_layout_gpu] error: %s\n", pgsgd_last_error());
        std::exit(1);
    }
    p.iter_max = iter_max;
    p.iter_with_max_learning_rate = iter_with_max_learning_rate;
    p.min_term_updates = min_term_updates;
    p.delta = delta;
    p.eps = eps;
    p.eta_max = eta_max;
    p.theta = theta;
    p.space = space;
    p.space_max = space_max;
    p.space_quantization_step = space_quantization_step;
    p.cooling_start = cooling_start;
    p.progress = progress ? 1 : 0;
    p.snapshot = snapshot ? 1 : 0;
    p.snapshot_prefix = snapshot ? snapshot_prefix.c_str() : nullptr;
    std::vector<double> xd(X.size()), yd(Y.size());
    for (size_t i = 0; i < X.size(); ++i) {
        xd[i] = X[i].load();
        yd[i] = Y[i].load();
    }
    pgsgd_stats st;
    const int rc = pgsgd_layout_run_f64(&view, &p, xd.data(), yd.data(), &st);  // full resolution of the device's coordinates
    if (rc != PGSGD_OK) {
        std::fprintf(stderr, "[odgi::path_linear_sgd_layout_gpu] error: %s: %s\n", pgsgd_strerror(rc), pgsgd_last_error());
        std::exit(1);
    }
    for (size_t i = 0; i < X.size(); ++i) {
        X[i].store(xd[i]);
        Y[i].store(yd[i]);
    }
}

// The 1D sibling with the reference's parameter list (src/algorithms/path_sgd.hpp:38-55, `odgi sort -Y`):
// returns the nodes' 1D positions.  `target_nodes` (optional, the reference passes it to
// path_linear_sgd_order: path_sgd.hpp:63-86) freezes the marked nodes.  Snapshots of intermediate
// positions (temp files upstream) are not produced.
template <class Graph, class PathIndex, class PathHandle>
std::vector<double> path_linear_sgd_gpu(const Graph& graph, const PathIndex& /*path_index*/,
                                        const std::vector<PathHandle>& /*path_sgd_use_paths*/, const uint64_t& iter_max,
                                        const uint64_t& iter_with_max_learning_rate, const uint64_t& min_term_updates,
                                        const double& delta, const double& eps, const double& eta_max, const double& theta,
                                        const uint64_t& space, const uint64_t& space_max, const uint64_t& space_quantization_step,
                                        const double& cooling_start, const uint64_t& nthreads, const bool& progress,
                                        const bool& /*snapshot*/, std::vector<std::string>& /*snapshots*/,
                                        const std::vector<bool>* target_nodes = nullptr) {
    const pgsgd::lowered_graph lg = pgsgd::lower_graph<PathHandle>(graph, nthreads);
    const pgsgd_graph_view view = lg.view();
    pgsgd_params p;
    if (pgsgd_sort_params_defaults(&view, &p) != PGSGD_OK) {
        std::fprintf(stderr, "[odgi::path_linear_sgd_gpu] error: %s\n", pgsgd_last_error());
        std::exit(1);
    }
    p.iter_max = iter_max;
    p.iter_with_max_learning_rate = iter_with_max_learning_rate;
    p.min_term_updates = min_term_updates;
    p.delta = delta;
    p.eps = eps;
    p.eta_max = eta_max;
    p.theta = theta;
    p.space = space;
    p.space_max = space_max;
    p.space_quantization_step = space_quantization_step;
    p.cooling_start = cooling_start;
    p.progress = progress ? 1 : 0;
    std::vector<double> X(view.n_nodes);
    pgsgd_sort_initial(&view, X.data());  // path_sgd.cpp:67-73
    std::vector<uint8_t> frozen;
    if (target_nodes) frozen.assign(target_nodes->begin(), target_nodes->end());
    pgsgd_stats st;
    const int rc = pgsgd_sort_run_targets(&view, &p, target_nodes ? frozen.data() : nullptr, X.data(), &st);
    if (rc != PGSGD_OK) {
        std::fprintf(stderr, "[odgi::path_linear_sgd_gpu] error: %s: %s\n", pgsgd_strerror(rc), pgsgd_last_error());
        std::exit(1);
    }
    return X;
}

}  // namespace algorithms
}  // namespace odgi
