// pgsgd_internal.hpp — types shared by the host translation units of libpgsgd.
#pragma once
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <new>
#include <atomic>
#include <string>
#include <thread>
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

// body(t) for t = 0 .. n-1 on n threads (t = 0 on the caller), joined before returning.  Nothing escapes: a thread that
// cannot be started has its index run by the caller (a worker-pool body then simply finds no work left), an exception
// in a body (std::bad_alloc while a worker grows its vectors) is caught, and either makes the call return false — a
// joinable std::thread that is destroyed, or an exception leaving a thread, would end the process in std::terminate,
// where the single-threaded readers returned PGSGD_E_NOMEM.
template <class F>
inline bool run_threads(unsigned n, F&& body) {
    std::atomic<bool> ok{true};
    auto guarded = [&](unsigned t) {
        try { body(t); } catch (...) { ok.store(false); }
    };
    std::vector<std::thread> th;
    unsigned started = 1;
    try {
        th.reserve(n > 1 ? n - 1 : 0);
        for (unsigned t = 1; t < n; ++t) { th.emplace_back(guarded, t); ++started; }
    } catch (...) {  // (std::system_error: no more threads; nothing is lost, the caller runs the rest)
    }
    guarded(0u);
    for (unsigned t = started; t < n; ++t) guarded(t);
    for (auto& x : th) x.join();
    return ok.load();
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
// The `-P` progress line in the form of the reference's ProgressMeter (src/algorithms/progress.hpp:52-68 with the banner of
// path_sgd_layout.cpp:45-46): "\r<banner> <percent>% @ <rate> bp/s elapsed: DD:HH:MM:SS remain: DD:HH:MM:SS" — the rate is
// term updates per second; the reference's meter labels it bp/s, and a drop-in keeps the label.  One line per iteration.
inline void progress_line(const char* banner, uint64_t completed, uint64_t total, double elapsed_s) {
    const double rate = elapsed_s > 0 ? (double)completed / elapsed_s : 0.0;
    const double remain_s = completed > 0 && rate > 0 ? (double)(total - completed) / rate : 0.0;
    auto dhms = [](double sec, int out[4]) {
        const long t = (long)sec;
        out[0] = (int)(t / 86400); out[1] = (int)(t % 86400 / 3600); out[2] = (int)(t % 3600 / 60); out[3] = (int)(t % 60);
    };
    int e[4], r[4];
    dhms(elapsed_s, e);
    dhms(remain_s, r);
    fprintf(stderr, "\r%s %5.2f%% @ %.2e bp/s elapsed: %02d:%02d:%02d:%02d remain: %02d:%02d:%02d:%02d", banner,
            total ? 100.0 * (double)completed / (double)total : 100.0, rate, e[0], e[1], e[2], e[3], r[0], r[1], r[2], r[3]);
}

// A snapshot between iterations (path_sgd_layout.cpp:379-408) written under the CALLER's node ranks: a run that laid the graph
// out under ranks by path position (pgsgd_layout_run: new_rank_of_old) names its snapshots back, so that `-u` changes
// neither the kernel a graph runs nor what the files mean.  new_rank_of_old null: the coordinates are written as they are.
int write_snapshot(const char* name, uint64_t n_nodes, const float* x, const float* y, const uint32_t* new_rank_of_old);
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

namespace pgsgd {
uint32_t load_flags();   // of the pgsgd_graph_load_flags call in progress on this thread (gfa_lower.cpp)
// step_pos / step_path of a view for host code that reads them at random steps (the quality figures): the view's arrays, or — when
// the view carries none (pgsgd_graph_view::step_pos / step_path may be NULL) — built here by one walk over the paths (xp.cpp:607-617).
struct StepIndex {
    const uint64_t* pos;
    const uint32_t* path;
    std::vector<uint64_t> own_pos;
    std::vector<uint32_t> own_path;
    explicit StepIndex(const pgsgd_graph_view* g) : pos(g->step_pos), path(g->step_path) {
        if (pos && path) return;
        if (!pos) own_pos.resize(g->n_steps);
        if (!path) own_path.resize(g->n_steps);
        for (uint64_t p = 0; p < g->n_paths; ++p) {
            uint64_t bp = 0;
            for (uint64_t k = g->path_first[p]; k < g->path_first[p + 1]; ++k) {
                if (!pos) own_pos[k] = bp;
                if (!path) own_path[k] = (uint32_t)p;
                bp += g->node_len[g->step_handle[k] >> 1];
            }
        }
        if (!pos) pos = own_pos.data();
        if (!path) path = own_path.data();
    }
};
}  // namespace pgsgd
