// pgsgd_internal.hpp — types shared by the host translation units of libpgsgd.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/pgsgd.h"

namespace pgsgd {

// thread-local error text behind pgsgd_last_error()
void set_error(const char* fmt, ...);
void clear_error();

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
