// pgsgd_multi.cpp — the multi-GPU run behind the C ABI: pgsgd_layout_run with params.n_devices > 1.
//
// North star / SURVEY 8(e): the graph is replicated read-only on every GPU, the terms of every learning-rate step are
// split 1/G per GPU, and the coordinates are merged with an RCCL all-reduce over xGMI at every step.  One host thread
// drives each device (one session each: graph upload, kernels and exchange on that device's stream); the merge is the
// one of odgi_amd/distributed.py (begin: what this rank moved since the last exchange and its squared length, SUM
// all-reduce of the fused buffer, end: base + S * clamp(Q / |S|^2, 1/G, 1)), so a C or C++ caller gets the same run
// the one-process-per-GPU Python driver gives.  The reference has no multi-device path (src/cuda/layout.cu is
// single-GPU; its NCCLCHECK macro is unused).
//
// One collective per exchange and nothing else between the ranks inside an iteration: the tail of the fused buffer
// carries every rank's max |Delta| (the reference's stop rule, path_sgd_layout.cpp:142) and frame-guard flag in a slot
// of its own, so the SUM hands all of them to everybody; the host threads meet once per iteration, where the ranks'
// error states are OR-ed inside the barrier and every rank acts on the same value.
//
// RCCL is bound at run time (dlopen "librccl.so"; types and enumerators from <rccl/rccl.h>): the library must not
// pull a second HIP runtime into a process that already has one (PyTorch ships its own), and single-GPU users need no
// RCCL at all.  Test knobs (PGSGD_DEBUG=1 only): PGSGD_MULTI_HOST_REDUCE=1 replaces the collective by a sum through
// pinned host memory, with several "devices" mapped to the one GPU of the test box (RCCL refuses two ranks on one
// device); PGSGD_MULTI_FORCE=1 sends an n_devices = 1 run through this driver — one rank, a one-rank communicator, the
// same ncclAllReduce calls — which is how the RCCL binding is executed on a single-GPU box.
#include <hip/hip_runtime_api.h>
// RCCL's declarations come from its header when the development package is there; a build box without it still builds
// the library (single-GPU users need RCCL neither at build nor at run time): the few declarations the exchange uses are
// then restated here, with the values rccl.h gives them (ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint64 = 5, ncclFloat32 = 7, ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pgsgd_internal.hpp"

using pgsgd::set_error;

int pgsgd_write_lay_f32(const char* path, uint64_t n_ends, const float* X, const float* Y);  // pgsgd_session.hip

namespace {

// the few RCCL entry points the exchange needs, with the signatures rccl.h declares
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) init_all = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool load() {
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        init_all = (decltype(init_all))dlsym(lib, "ncclCommInitAll");
        destroy = (decltype(destroy))dlsym(lib, "ncclCommDestroy");
        all_reduce = (decltype(all_reduce))dlsym(lib, "ncclAllReduce");
        error_string = (decltype(error_string))dlsym(lib, "ncclGetErrorString");
        return init_all && destroy && all_reduce;
    }
};

// A reusable barrier for the G driver threads that also agrees on a flag: wait(f) returns the OR of the f of every rank
// in this round — computed by the last arriver under the lock, so every rank leaves with the same value and acts on it
// (a rank that reads a shared "failed" word on its own after a barrier can see a later failure the others did not).
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t round = 0;
    bool acc = false, result = false;
    explicit Barrier(int n_) : n(n_) {}
    bool wait(bool flag = false) {
        std::unique_lock<std::mutex> lk(m);
        acc = acc || flag;
        const uint64_t r = round;
        if (++waiting == n) {
            result = acc;
            acc = false;
            waiting = 0;
            ++round;
            cv.notify_all();
            return result;
        }
        cv.wait(lk, [&] { return round != r; });
        return result;  // (the next round cannot complete before this rank has returned and arrived again)
    }
};

struct Shared {
    int G;
    const pgsgd_graph_view* g;
    const pgsgd_params* p;
    std::vector<int> devices;
    std::vector<double> etas;
    uint64_t first_cooling;
    Barrier barrier;
    // the host threads are all created before any of them starts: a thread that cannot be created must not leave the
    // others waiting at a barrier for it
    std::mutex start_m;
    std::condition_variable start_cv;
    int start = 0;                      // 0: wait, 1: go, 2: give up
    Rccl* rccl = nullptr;               // null: host-staged sum
    std::vector<ncclComm_t> comms;
    std::vector<float*> host_buf;       // [2G] pinned, host-staged sum only
    std::vector<std::vector<uint64_t>> host_tail64;  // [G][2G] the exact exchange's tail as read back
    std::vector<int> rc;                // [G] first error of each rank
    std::vector<std::string> err;       // [G]
    // what rank 0 found when it set its session up; the ranks' sessions are built from the same graph and layout, so
    // it holds for all of them (and a rank whose set-up failed never uses it: everybody leaves after the first barrier)
    bool tiled = false, warm_per_lane = false;
    const uint32_t* snapshot_names = nullptr;  // new_rank_of_old of a run under renamed node ranks: snapshots go out under the caller's
    const float* X0;
    const float* Y0;
    float* X;
    float* Y;
    double* Xd;
    double* Yd;
    pgsgd_stats stats{};
    Shared(int G_) : G(G_), barrier(G_) {}
};

#define R_TRY(expr)                                                                 \
    do {                                                                            \
        const int _rc = (expr);                                                     \
        if (_rc != PGSGD_OK && !sh.rc[r]) {                                         \
            sh.rc[r] = _rc;                                                         \
            sh.err[r] = pgsgd_last_error();                                         \
        }                                                                           \
    } while (0)
#define R_HIP(expr)                                                                 \
    do {                                                                            \
        const hipError_t _e = (expr);                                               \
        if (_e != hipSuccess && !sh.rc[r]) {                                        \
            sh.rc[r] = PGSGD_E_HIP;                                                 \
            sh.err[r] = std::string(#expr) + ": " + hipGetErrorString(_e);          \
        }                                                                           \
    } while (0)

// One rank = one device.  Every rank runs the same control flow; whether to go on is decided at barriers, from the OR
// of the ranks' error states, so all of them leave a loop in the same place and nobody waits for ever.
void rank_main(Shared& sh, int r) {
    if (r != 0) {
        std::unique_lock<std::mutex> lk(sh.start_m);
        sh.start_cv.wait(lk, [&] { return sh.start != 0; });
        if (sh.start == 2) return;
    }
    const int G = sh.G;
    const pgsgd_params& p0 = *sh.p;
    const uint64_t N = sh.g->n_nodes;
    const size_t buf_floats = 6 * N + 2 * (size_t)G;  // S (4N), Q (2N), then every rank's max |Delta| and frame-guard flag
    const size_t buf_words = 2 * N + 3 * (size_t)G;   // the exact exchange: 2N integer deltas, then every rank's far-pull count, max |Delta| bits, guard flag
    static_assert(sizeof(uint64_t) == 2 * sizeof(float), "the two exchanges share one buffer");
    const size_t buf_bytes = std::max(buf_floats * sizeof(float), buf_words * sizeof(uint64_t));
    pgsgd_params p = p0;
    p.device = sh.devices[r];
    // Sampler streams: the per-lane kernel's lane l draws from seed + stream_offset + l, so ranks take disjoint blocks of
    // 2^20 ids (a GPU runs < 2^20 lanes).  The tile kernel seeds by (seed + stream_offset, iteration, tile, lane) and
    // every tile belongs to exactly one rank: there the offset must NOT depend on the rank, or rank r's tile t would
    // draw the stream of rank 0's tile t + 1024 r (the tile index enters as tile << 10) — pgsgd_session_set_shard
    // takes the rank's offset out of a sharded session's tile seeds.
    p.stream_offset = p0.stream_offset + (uint32_t)r * (1u << 20);
    p.n_devices = 1;
    // (regions of 128 nodes where 256-node windows would not fill the devices: the rule then shards by region with the exact
    // exchange; not under the test knob that forces a way of sharding on a graph of any size)
    if (!pgsgd::debug_env("PGSGD_MULTI_SHARD")) p.flags |= pgsgd_shard_flags(N, (uint32_t)G, p.flags);
    p.snapshot = 0;  // snapshots are written by rank 0 below, from the merged coordinates
    p.progress = r == 0 ? p0.progress : 0;
    pgsgd_session* s = nullptr;
    float* buf = nullptr;
    float* h_tail = nullptr;  // pinned: the tails of the exchanges of one iteration (up to four)
    R_HIP(hipSetDevice(sh.devices[r]));
    R_TRY(pgsgd_session_create(sh.g, &p, &s));
    if (s) R_TRY(pgsgd_session_upload_coords(s, sh.X0, sh.Y0));
    if (s) R_TRY(pgsgd_session_exchange_mark(s));
    bool engine_sharded = false, exact = false;  // exact: region shard, an integer exchange after each colour's launch
    if (s && !sh.rc[r]) {
        const int info = pgsgd_session_tile_info(s, nullptr, nullptr, nullptr, nullptr, nullptr);
        // by tile, or by node region when that leaves every launch enough work items (decided by the session: -1)
        int want = -1;
        if (const char* e = pgsgd::debug_env("PGSGD_MULTI_SHARD"))  // test knob: tiles / regions / exact whatever the graph's size
            want = !strcmp(e, "tiles") ? 0 : !strcmp(e, "regions") ? 1 : !strcmp(e, "exact") ? 2 : -1;
        if (want > 0 && info <= 0) want = -1;   // (the knob asks for a way of sharding TILES: a per-lane session shards its term count)
        // (-1: the session's own rule — by region with the exact exchange when a launch keeps a thousand windows per rank and
        // the coordinates are fixed-point, by region with the merge rule when they are not, by tile otherwise; the
        // torch.distributed route, odgi_amd/distributed.py, passes the same -1)
        int rcs = pgsgd_session_set_shard(s, (uint32_t)r, (uint32_t)G, want);
        if (rcs < 0) R_TRY(rcs);
        engine_sharded = rcs > 0;
        exact = rcs == 3;
        if (r == 0) {
            sh.tiled = info > 0;
            sh.warm_per_lane = info == 2;
        }
        R_HIP(hipMalloc((void**)&buf, buf_bytes));
        R_HIP(hipHostMalloc((void**)&h_tail, 4 * 2 * (size_t)G * sizeof(float)));
    }
    hipStream_t stream = s ? (hipStream_t)pgsgd_session_stream(s) : nullptr;
    auto cleanup = [&] {
        if (buf) (void)hipFree(buf);
        if (h_tail) (void)hipHostFree(h_tail);
        if (s) pgsgd_session_destroy(s);
    };
    if (sh.barrier.wait(sh.rc[r] != 0)) {  // somebody's set-up failed: everybody leaves here
        cleanup();
        return;
    }
    const uint64_t M = p0.min_term_updates;
    const uint64_t my_terms = engine_sharded ? M : M / G + ((uint64_t)r < M % G ? 1 : 0);
    // one exchange: what this rank moved, every rank's statistics in the tail, SUM over the ranks, merge
    auto exchange = [&](float* tail_out) -> bool {  // false: the run cannot go on (decided together)
        if (!sh.rc[r]) R_TRY(pgsgd_session_exchange_begin_stats(s, buf, (uint32_t)r, (uint32_t)G));
        if (sh.rccl) {
            // (no barrier: a rank in trouble still issues its collective — the buffer and the communicator exist — and the
            // ranks decide together at the end of the iteration)
            const ncclResult_t e = sh.rccl->all_reduce(buf, buf, buf_floats, ncclFloat32, ncclSum, sh.comms[r], stream);
            if (e != ncclSuccess && !sh.rc[r]) {
                sh.rc[r] = PGSGD_E_HIP;
                sh.err[r] = std::string("ncclAllReduce: ") + (sh.rccl->error_string ? sh.rccl->error_string(e) : "error");
            }
        } else {  // sum through pinned host memory (tests on a single GPU)
            if (!sh.rc[r]) R_HIP(hipMemcpyAsync(sh.host_buf[r], buf, buf_floats * sizeof(float), hipMemcpyDeviceToHost, stream));
            if (!sh.rc[r]) R_HIP(hipStreamSynchronize(stream));
            if (sh.barrier.wait(sh.rc[r] != 0)) return false;
            float* mine = sh.host_buf[G + r];  // a second slot per rank for the sum
            for (size_t i = 0; i < buf_floats; ++i) {
                float acc = 0;
                for (int q = 0; q < G; ++q) acc += sh.host_buf[q][i];
                mine[i] = acc;
            }
            sh.barrier.wait();  // everybody has read the ranks' buffers
            R_HIP(hipMemcpyAsync(buf, mine, buf_floats * sizeof(float), hipMemcpyHostToDevice, stream));
        }
        if (!sh.rc[r]) R_TRY(pgsgd_session_exchange_end(s, buf, G));
        if (!sh.rc[r]) R_HIP(hipMemcpyAsync(tail_out, buf + 6 * N, 2 * (size_t)G * sizeof(float), hipMemcpyDeviceToHost, stream));
        return true;
    };
    // the exact exchange after one colour's launch of a region-sharded run: far pulls delivered, 64-bit integer deltas
    // summed over the ranks, every rank ends with one GPU's coordinates (pgsgd_session_exchange_exact_*); the tail read
    // back is {max |Delta| bits, guard flag} x G as floats, like the other exchange's
    auto exchange_exact = [&](float* tail_out) -> bool {
        uint64_t* wbuf = reinterpret_cast<uint64_t*>(buf);
        if (!sh.rc[r]) R_TRY(pgsgd_session_exchange_exact_begin(s, wbuf, (uint32_t)r, (uint32_t)G));
        if (sh.rccl) {
            const ncclResult_t e = sh.rccl->all_reduce(wbuf, wbuf, buf_words, ncclUint64, ncclSum, sh.comms[r], stream);
            if (e != ncclSuccess && !sh.rc[r]) {
                sh.rc[r] = PGSGD_E_HIP;
                sh.err[r] = std::string("ncclAllReduce: ") + (sh.rccl->error_string ? sh.rccl->error_string(e) : "error");
            }
        } else {  // sum through pinned host memory (tests on a single GPU)
            if (!sh.rc[r]) R_HIP(hipMemcpyAsync(sh.host_buf[r], wbuf, buf_words * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            if (!sh.rc[r]) R_HIP(hipStreamSynchronize(stream));
            if (sh.barrier.wait(sh.rc[r] != 0)) return false;
            uint64_t* mine = reinterpret_cast<uint64_t*>(sh.host_buf[G + r]);
            for (size_t i = 0; i < buf_words; ++i) {
                uint64_t acc = 0;
                for (int q = 0; q < G; ++q) acc += reinterpret_cast<const uint64_t*>(sh.host_buf[q])[i];
                mine[i] = acc;
            }
            sh.barrier.wait();  // everybody has read the ranks' buffers
            R_HIP(hipMemcpyAsync(wbuf, mine, buf_words * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        }
        if (!sh.rc[r]) R_TRY(pgsgd_session_exchange_exact_end(s, wbuf, (uint32_t)G));
        // the tail's words: the low halves of the max |Delta| slots are the floats' bits; guard slots are 0 / 1
        if (!sh.rc[r]) R_HIP(hipMemcpyAsync(sh.host_tail64[r].data(), wbuf + 2 * N + G, 2 * (size_t)G * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        if (!sh.rc[r]) R_HIP(hipStreamSynchronize(stream));
        for (int q = 0; q < G; ++q) {
            const uint32_t bits = (uint32_t)sh.host_tail64[r][(size_t)q];
            float f;
            memcpy(&f, &bits, sizeof f);
            tail_out[q] = f;
            tail_out[G + q] = sh.host_tail64[r][(size_t)G + q] ? 1.0f : 0.0f;
        }
        return true;
    };
    uint64_t iters = 0;
    uint32_t early = 0;
    double dmax_all = 0;
    bool broken = false;
    std::vector<float> sx, sy;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t it = 0; it < p0.iter_max && !broken; ++it) {
        const bool cooling = it >= sh.first_cooling;
        // the per-lane kernel needs four exchanges per iteration to keep the one-GPU quality, the tile kernel one
        // (measured with virtual ranks)
        const bool tiles_now = sh.tiled && !(sh.warm_per_lane && !cooling);
        const uint32_t blocks = !tiles_now ? 4u : exact ? 2u : 1u;   // exact exchange: one part per region colour
        for (uint32_t b = 0; b < blocks && !broken; ++b) {
            if (!sh.rc[r]) R_TRY(pgsgd_session_iteration_part(s, sh.etas[it], cooling ? 1 : 0, my_terms, b, blocks));
            broken = !((exact && tiles_now) ? exchange_exact(h_tail + (size_t)b * 2 * G) : exchange(h_tail + (size_t)b * 2 * G));
        }
        if (!broken && !sh.rc[r]) R_TRY(pgsgd_session_sync(s, nullptr));
        // the one meeting of the iteration: did anybody fail?
        if (sh.barrier.wait(broken || sh.rc[r] != 0)) {
            broken = true;
            break;
        }
        // every rank holds every rank's max |Delta| and frame-guard flag of every exchange: the same decisions everywhere
        double dmax = 0;
        bool any_guard = false;
        for (uint32_t b = 0; b < blocks; ++b)
            for (int q = 0; q < G; ++q) {
                dmax = std::max(dmax, (double)h_tail[(size_t)b * 2 * G + q]);
                any_guard = any_guard || h_tail[(size_t)b * 2 * G + G + q] != 0.0f;
            }
        dmax_all = dmax;
        if (any_guard) R_TRY(pgsgd_session_reframe(s));
        ++iters;
        if (r == 0 && p0.progress)
            pgsgd::progress_line("[odgi::path_linear_sgd_layout] 2D path-guided SGD:", iters * M, p0.iter_max * M,
                                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        if (it + 1 >= p0.iter_max) break;
        if (dmax_all <= p0.delta) {  // path_sgd_layout.cpp:142 — the same decision on every rank
            early = 1;
            break;
        }
        if (r == 0 && p0.snapshot && p0.snapshot_prefix && !sh.rc[r]) {  // :379-408: snapshot k after iteration k, k = 1..iter_max-1
            sx.resize(2 * N);
            sy.resize(2 * N);
            R_TRY(pgsgd_session_peek_coords(s, sx.data(), sy.data()));  // rank 0's copy = the merged coordinates, as they are between iterations
            const std::string name = std::string(p0.snapshot_prefix) + std::to_string(it + 1);
            fprintf(stderr, "[odgi::path_linear_sgd_layout] snapshot thread: Taking snapshot!\n");
            if (!sh.rc[r]) R_TRY(pgsgd::write_snapshot(name.c_str(), N, sx.data(), sy.data(), sh.snapshot_names));
        }
    }
    if (r == 0 && p0.progress) fprintf(stderr, "\n");
    // the far pulls of every rank's last tile launch: delivered, then merged like any other move
    bool failed = broken;  // (a rank that left the loop on a barrier's verdict left it with every other rank)
    if (!failed) failed = sh.barrier.wait(sh.rc[r] != 0);
    if (!failed && sh.tiled && !exact) {  // (an exact exchange delivers its launch's far pulls itself: nothing is left)
        R_TRY(pgsgd_session_flush(s));
        failed = !exchange(h_tail);
        if (!failed && !sh.rc[r]) R_TRY(pgsgd_session_sync(s, nullptr));
        if (!failed) failed = sh.barrier.wait(sh.rc[r] != 0);
    }
    if (r == 0 && !failed) {
        if (sh.Xd) R_TRY(pgsgd_session_download_coords_f64(s, sh.Xd, sh.Yd));
        else R_TRY(pgsgd_session_download_coords(s, sh.X, sh.Y));
        sh.stats.iterations = iters;
        sh.stats.term_updates = iters * M;
        sh.stats.last_delta_max = dmax_all;
        sh.stats.n_streams = pgsgd_session_n_streams(s);
        sh.stats.early_stop = early;
        sh.stats.tiled = sh.tiled ? (sh.warm_per_lane ? 2u : 1u) : 0u;
        uint32_t doublings = 0;
        (void)pgsgd_session_frame_status(s, nullptr, &doublings);
        sh.stats.frame_doublings = doublings;
        (void)pgsgd_session_kernel_time(s, &sh.stats.kernel_ms, nullptr, 0);
        sh.stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    cleanup();
}

}  // namespace

// X,Y as in pgsgd_layout_run (fp32) or, when Xd/Yd are given, doubles out (pgsgd_layout_run_f64).
int pgsgd_layout_run_multi(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, double* Xd, double* Yd, pgsgd_stats* stats, const uint32_t* snapshot_names) {
    pgsgd::clear_error();
    if (stats) memset(stats, 0, sizeof *stats);
    const bool forced_single = p && p->n_devices == 1 && pgsgd::debug_env("PGSGD_MULTI_FORCE") != nullptr;
    if (!g || !p || !X || !Y || (p->n_devices < 2 && !forced_single)) return PGSGD_E_INVALID;
    const int G = (int)p->n_devices;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device available; the layout kernels run on MI355X only, there is no CPU fallback");
        return PGSGD_E_NODEVICE;
    }
    const bool host_reduce = pgsgd::debug_env("PGSGD_MULTI_HOST_REDUCE") != nullptr;
    if (G > count && !host_reduce) {
        set_error("%d GPUs requested, %d present", G, count);
        return PGSGD_E_NODEVICE;
    }
    Shared sh(G);
    sh.g = g;
    sh.p = p;
    sh.snapshot_names = snapshot_names;
    sh.devices.resize(G);
    const int first = p->device < 0 ? 0 : p->device;
    for (int r = 0; r < G; ++r) sh.devices[r] = host_reduce ? (first + r) % count : first + r;  // virtual ranks may share a GPU
    if (!host_reduce && first + G > count) { set_error("devices %d..%d requested, %d present", first, first + G - 1, count); return PGSGD_E_NODEVICE; }
    sh.etas.resize(p->iter_max + 1);
    if (pgsgd_schedule(p, sh.etas.data(), sh.etas.size()) < 0) return PGSGD_E_INVALID;
    sh.first_cooling = (uint64_t)std::floor(p->cooling_start * (double)p->iter_max);
    sh.rc.assign(G, 0);
    sh.err.assign(G, "");
    sh.host_tail64.assign((size_t)G, std::vector<uint64_t>(2 * (size_t)G, 0));
    sh.X0 = X; sh.Y0 = Y; sh.X = X; sh.Y = Y; sh.Xd = Xd; sh.Yd = Yd;
    Rccl rccl;
    auto release = [&] {
        if (sh.rccl) for (ncclComm_t c : sh.comms) if (c) (void)rccl.destroy(c);
        for (float* b : sh.host_buf) if (b) (void)hipHostFree(b);
    };
    if (!host_reduce) {
        if (!rccl.load()) { set_error("n_devices = %d needs RCCL: librccl.so could not be loaded (%s)", G, dlerror()); return PGSGD_E_UNSUPPORTED; }
        sh.comms.assign(G, nullptr);
        const ncclResult_t e = rccl.init_all(sh.comms.data(), G, sh.devices.data());
        if (e != ncclSuccess) {
            set_error("ncclCommInitAll on %d devices: %s", G, rccl.error_string ? rccl.error_string(e) : "error");
            return PGSGD_E_HIP;
        }
        sh.rccl = &rccl;
    } else {
        sh.host_buf.assign(2 * (size_t)G, nullptr);
        for (size_t i = 0; i < sh.host_buf.size(); ++i)
            if (hipHostMalloc((void**)&sh.host_buf[i], std::max((6 * g->n_nodes + 2 * (size_t)G) * sizeof(float), (2 * g->n_nodes + 3 * (size_t)G) * sizeof(uint64_t))) != hipSuccess) {
                set_error("pinned exchange buffers");
                release();
                return PGSGD_E_NOMEM;
            }
    }
    {
        std::vector<std::thread> th;
        bool all_started = true;
        try {
            for (int r = 1; r < G; ++r) th.emplace_back(rank_main, std::ref(sh), r);
        } catch (...) {
            all_started = false;
        }
        {
            std::lock_guard<std::mutex> lk(sh.start_m);
            sh.start = all_started ? 1 : 2;
        }
        sh.start_cv.notify_all();
        if (all_started) rank_main(sh, 0);
        for (auto& t : th) t.join();
        if (!all_started) {
            release();
            set_error("could not start %d host threads for the GPU ranks", G - 1);
            return PGSGD_E_NOMEM;
        }
    }
    release();
    for (int r = 0; r < G; ++r)
        if (sh.rc[r]) {
            set_error("GPU rank %d: %s", r, sh.err[r].c_str());
            return sh.rc[r];
        }
    if (stats) *stats = sh.stats;
    return PGSGD_OK;
}
