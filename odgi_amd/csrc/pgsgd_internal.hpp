// pgsgd_internal.hpp — types shared by the host translation units of libpgsgd.
#pragma once
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pgsgd.h"

namespace pgsgd {

// thread-local error text behind pgsgd_last_error()
void set_error(const char* fmt, ...);
void clear_error();

// Experiment and test knobs (PGSGD_TILE_*, PGSGD_OUTBOX_*, PGSGD_FRAME_SPAN, PGSGD_MULTI_*): environment variables that
// change what the library computes or how it is laid out.  They are read when a session (or a multi-GPU run) is set up,
// never on a launch path, and ONLY in a process that also sets PGSGD_DEBUG=1 — a stray variable in a user's environment
// cannot change a layout.  tests/ and tools/ set PGSGD_DEBUG=1.
inline const char* debug_env(const char* name) {
    const char* dbg = getenv("PGSGD_DEBUG");
    if (!dbg || dbg[0] != '1') return nullptr;
    return getenv(name);
}

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
namespace pgsgd {
// std::vector whose resize() leaves new elements uninitialised: the per-step arrays (hundreds of MB) are written once,
// by threads, right after they are sized — zero-filling them first is a serial pass over memory of the same length
template <class T>
struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using raw_vector = std::vector<T, default_init_allocator<T>>;
}  // namespace pgsgd

struct pgsgd_graph {
    uint64_t n_nodes = 0;
    std::vector<uint32_t> node_len;     // [N]
    std::vector<uint64_t> path_first;   // [P+1]
    pgsgd::raw_vector<uint32_t> step_path;    // [S]
    pgsgd::raw_vector<uint32_t> step_handle;  // [S]
    pgsgd::raw_vector<uint64_t> step_pos;     // [S]
    std::vector<uint64_t> edges;        // [2E] handle pairs
    std::vector<std::string> path_names;
    uint64_t n_paths() const { return path_first.empty() ? 0 : path_first.size() - 1; }
    uint64_t n_steps() const { return step_handle.size(); }
};

int pgsgd_validate_view(const pgsgd_graph_view* g);
