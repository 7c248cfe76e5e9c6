// Compile-and-run check of include/pgsgd_handlegraph.hpp against a mock graph type that offers the
// handlegraph calls the shim uses (the 4-node graph of reference src/unittest/pathindex.cpp:22-57,
// a few times over so that the layout has something to do).
// Prints the lowered index and, when a GPU is present, runs the layout; exit code 0 = ok,
// 3 = no device (expected on the CPU-only container).
#include <cstdio>
#include <functional>

#include "pgsgd_handlegraph.hpp"

namespace mock {
struct handle_t { uint64_t v; };
struct path_handle_t { uint64_t v; };
struct step_handle_t { uint64_t path, rank; };
inline uint64_t as_integer(const handle_t& h) { return h.v; }

struct graph_t {
    std::vector<uint32_t> len;
    std::vector<std::vector<handle_t>> paths;
    uint64_t get_node_count() const { return len.size(); }
    uint64_t get_length(const handle_t& h) const { return len[h.v >> 1]; }
    void for_each_handle(const std::function<void(const handle_t&)>& f) const {
        for (uint64_t i = 0; i < len.size(); ++i) f(handle_t{2 * i});
    }
    void for_each_path_handle(const std::function<void(const path_handle_t&)>& f) const {
        for (uint64_t p = 0; p < paths.size(); ++p) f(path_handle_t{p + 1});
    }
    void for_each_step_in_path(const path_handle_t& p, const std::function<void(const step_handle_t&)>& f) const {
        for (uint64_t r = 0; r < paths[p.v - 1].size(); ++r) f(step_handle_t{p.v, r});
    }
    handle_t get_handle_of_step(const step_handle_t& s) const { return paths[s.path - 1][s.rank]; }
    uint64_t get_step_count(const path_handle_t& p) const { return paths[p.v - 1].size(); }
};
struct xp_t {};
}  // namespace mock

int main() {
    mock::graph_t g;
    const int copies = 200;
    for (int c = 0; c < copies; ++c) {
        const uint32_t l[4] = {4, 1, 2, 7};  // AGGA, A, TC, TCTCAGG
        for (uint32_t x : l) g.len.push_back(x);
    }
    std::vector<mock::handle_t> p5, p5m;
    for (int c = 0; c < copies; ++c) {
        const uint64_t b = 4ull * c;
        p5.push_back({2 * (b + 0)}); p5.push_back({2 * (b + 2)}); p5.push_back({2 * (b + 3)});          // 1+,3+,4+
        p5m.push_back({2 * (b + 0)}); p5m.push_back({2 * (b + 3)}); p5m.push_back({2 * (b + 2) + 1});  // 1+,4+,3-
    }
    g.paths = {p5, p5m};
    const pgsgd::lowered_graph lg = pgsgd::lower_graph<mock::path_handle_t>(g, 2);
    std::printf("N=%zu S=%zu P=%zu pos=%llu,%llu,%llu handle_last=%u\n", lg.node_len.size(), lg.step_handle.size(), lg.path_first.size() - 1,
                (unsigned long long)lg.step_pos[0], (unsigned long long)lg.step_pos[1], (unsigned long long)lg.step_pos[2], lg.step_handle[3 * copies + 2]);
    if (lg.step_pos[1] != 4 || lg.step_pos[2] != 6 || lg.step_handle[3 * copies + 2] != 5) return 2;
    const pgsgd_graph_view view = lg.view();
    pgsgd_params p;
    pgsgd_params_defaults(&view, &p);
    std::vector<double> X0(2 * g.len.size()), Y0(X0.size());
    pgsgd_init_layout(&view, 'd', 7, X0.data(), Y0.data());
    std::vector<std::atomic<double>> X(X0.size()), Y(Y0.size());
    for (size_t i = 0; i < X0.size(); ++i) { X[i].store(X0[i]); Y[i].store(Y0[i]); }
    pgsgd_session* probe = nullptr;
    if (pgsgd_session_create(&view, &p, &probe) == PGSGD_E_NODEVICE) { std::printf("no device\n"); return 3; }
    pgsgd_session_destroy(probe);
    std::vector<mock::path_handle_t> use;
    odgi::algorithms::path_linear_sgd_layout_gpu(g, mock::xp_t{}, use, p.iter_max, (uint64_t)0, p.min_term_updates, p.delta, p.eps, p.eta_max,
                                                 p.theta, p.space, p.space_max, p.space_quantization_step, p.cooling_start, (uint64_t)2, false,
                                                 false, std::string(), X, Y);
    double before = 0, after = 0;
    std::vector<double> X1(X0.size()), Y1(Y0.size());
    for (size_t i = 0; i < X0.size(); ++i) { X1[i] = X[i].load(); Y1[i] = Y[i].load(); }
    pgsgd_path_stress(&view, X0.data(), Y0.data(), 100000, 1, &before);
    pgsgd_path_stress(&view, X1.data(), Y1.data(), 100000, 1, &after);
    std::printf("stress %.4f -> %.4f\n", before, after);
    if (!(after < before)) return 4;
    // the 1D sibling: frozen nodes stay where the initial order put them, the others move
    std::vector<bool> targets(g.len.size(), false);
    for (size_t i = 0; i < targets.size(); i += 5) targets[i] = true;
    std::vector<std::string> snapshots;
    pgsgd_params sp;
    pgsgd_sort_params_defaults(&view, &sp);
    const std::vector<double> X1d = odgi::algorithms::path_linear_sgd_gpu(g, mock::xp_t{}, use, sp.iter_max, (uint64_t)0, sp.min_term_updates, sp.delta,
                                                                          sp.eps, sp.eta_max, sp.theta, sp.space, sp.space_max,
                                                                          sp.space_quantization_step, sp.cooling_start, (uint64_t)2, false, false,
                                                                          snapshots, &targets);
    std::vector<double> Xi(g.len.size());
    pgsgd_sort_initial(&view, Xi.data());
    size_t moved = 0;
    for (size_t i = 0; i < Xi.size(); ++i) {
        if (targets[i] && X1d[i] != Xi[i]) return 5;
        moved += X1d[i] != Xi[i];
    }
    std::printf("1D: %zu of %zu nodes moved\n", moved, Xi.size());
    return moved > 0 ? 0 : 6;
}
