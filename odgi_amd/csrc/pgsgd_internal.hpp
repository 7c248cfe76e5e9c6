// pgsgd_internal.hpp — types shared by the host translation units of libpgsgd.
#pragma once
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/pgsgd.h"

namespace pgsgd {

// thread-local error text behind pgsgd_last_error()
void set_error(const char* fmt, ...);
void clear_error();

// PGSGD_TIMING=1: wall-clock of the set-up phases on stderr (where the time of a run goes besides the kernels)
struct PhaseTimer {
    const bool on = getenv("PGSGD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pgsgd timing] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};


}  // namespace pgsgd

// The owning counterpart of pgsgd_graph_view: what `graph_t` + `XP` are lowered to.
struct pgsgd_graph {
    uint64_t n_nodes = 0;
    std::vector<uint32_t> node_len;     // [N]
    std::vector<uint64_t> path_first;   // [P+1]
    std::vector<uint32_t> step_path;    // [S]
    std::vector<uint32_t> step_handle;  // [S]
    std::vector<uint64_t> step_pos;     // [S]
    std::vector<uint64_t> edges;        // [2E] handle pairs
    std::vector<std::string> path_names;
    uint64_t n_paths() const { return path_first.empty() ? 0 : path_first.size() - 1; }
    uint64_t n_steps() const { return step_handle.size(); }
};

int pgsgd_validate_view(const pgsgd_graph_view* g);
