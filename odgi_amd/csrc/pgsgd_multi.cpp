// pgsgd_multi.cpp — the multi-GPU run behind the C ABI: pgsgd_layout_run with params.n_devices > 1.
//
// North star / SURVEY 8(e): the graph is replicated read-only on every GPU, the terms of every learning-rate step are
// split 1/G per GPU, and the coordinates are merged with an RCCL all-reduce over xGMI at every step.  One host thread
// drives each device (one session each: graph upload, kernels and exchange on that device's stream); the merge is the
// one of odgi_amd/distributed.py (begin: what this rank moved since the last exchange and its squared length, SUM
// all-reduce of the fused 6N-float buffer, end: base + S * clamp(Q / |S|^2, 1/G, 1)), so a C or C++ caller gets the
// same run the one-process-per-GPU Python driver gives.  The reference has no multi-device path (src/cuda/layout.cu
// is single-GPU; its NCCLCHECK macro is unused).
//
// RCCL is bound at run time (dlopen "librccl.so"): the library must not pull a second HIP runtime into a process
// that already has one (PyTorch ships its own), and single-GPU users need no RCCL at all.  PGSGD_MULTI_HOST_REDUCE=1
// replaces the collective by a sum through pinned host memory — what the single-GPU test box runs, with several
// "devices" mapped to the one GPU it has (RCCL refuses two ranks on one device).
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
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

namespace {

// the few RCCL entry points the exchange needs (rccl.h: ncclFloat32 = 7, ncclSum = 0)
typedef void* ncclComm_t;
typedef int (*ncclCommInitAll_t)(ncclComm_t*, int, const int*);
typedef int (*ncclCommDestroy_t)(ncclComm_t);
typedef int (*ncclAllReduce_t)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef const char* (*ncclGetErrorString_t)(int);
struct Rccl {
    void* lib = nullptr;
    ncclCommInitAll_t init_all = nullptr;
    ncclCommDestroy_t destroy = nullptr;
    ncclAllReduce_t all_reduce = nullptr;
    ncclGetErrorString_t error_string = nullptr;
    bool load() {
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        init_all = (ncclCommInitAll_t)dlsym(lib, "ncclCommInitAll");
        destroy = (ncclCommDestroy_t)dlsym(lib, "ncclCommDestroy");
        all_reduce = (ncclAllReduce_t)dlsym(lib, "ncclAllReduce");
        error_string = (ncclGetErrorString_t)dlsym(lib, "ncclGetErrorString");
        return init_all && destroy && all_reduce;
    }
};

// a reusable barrier for the G driver threads
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t round = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t r = round;
        if (++waiting == n) {
            waiting = 0;
            ++round;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return round != r; });
        }
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
    Rccl* rccl = nullptr;               // null: host-staged sum
    std::vector<ncclComm_t> comms;
    std::vector<float*> host_buf;       // [G] pinned, host-staged sum only
    std::vector<double> dmax;           // [G] per iteration
    std::vector<int> guard;             // [G]
    std::vector<int> rc;                // [G] first error of each rank
    std::vector<std::string> err;       // [G]
    std::atomic<int> failed{0};
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
            sh.failed.store(1);                                                     \
        }                                                                           \
    } while (0)
#define R_HIP(expr)                                                                 \
    do {                                                                            \
        const hipError_t _e = (expr);                                               \
        if (_e != hipSuccess && !sh.rc[r]) {                                        \
            sh.rc[r] = PGSGD_E_HIP;                                                 \
            sh.err[r] = std::string(#expr) + ": " + hipGetErrorString(_e);          \
            sh.failed.store(1);                                                     \
        }                                                                           \
    } while (0)

// One rank = one device.  Every rank runs the same control flow and meets the others at the same barriers whatever
// happens (a rank that failed keeps walking through the barriers and skips the work), so nobody waits for ever.
void rank_main(Shared& sh, int r) {
    const int G = sh.G;
    const pgsgd_params& p0 = *sh.p;
    const uint64_t N = sh.g->n_nodes;
    pgsgd_params p = p0;
    p.device = sh.devices[r];
    p.stream_offset = p0.stream_offset + (uint32_t)r * (1u << 20);  // disjoint sampler stream ids per rank (a GPU runs < 2^20 streams)
    p.n_devices = 1;
    p.snapshot = 0;
    p.progress = r == 0 ? p0.progress : 0;
    pgsgd_session* s = nullptr;
    float* buf = nullptr;
    R_HIP(hipSetDevice(sh.devices[r]));
    R_TRY(pgsgd_session_create(sh.g, &p, &s));
    if (s) R_TRY(pgsgd_session_upload_coords(s, sh.X0, sh.Y0));
    if (s) R_TRY(pgsgd_session_exchange_mark(s));
    bool engine_sharded = false, warm_per_lane = false, tiled = false;
    if (s && !sh.rc[r]) {
        const int rcs = pgsgd_session_set_shard(s, (uint32_t)r, (uint32_t)G, 0);
        if (rcs < 0) R_TRY(rcs);
        engine_sharded = rcs == 1;
        const int info = pgsgd_session_tile_info(s, nullptr, nullptr, nullptr, nullptr, nullptr);
        tiled = info > 0;
        warm_per_lane = info == 2;
        R_HIP(hipMalloc((void**)&buf, 6 * N * sizeof(float)));
    }
    hipStream_t stream = s ? (hipStream_t)pgsgd_session_stream(s) : nullptr;
    sh.barrier.wait();  // everybody is set up (or has failed)
    const uint64_t M = p0.min_term_updates;
    const uint64_t my_terms = engine_sharded ? M : M / G + ((uint64_t)r < M % G ? 1 : 0);
    uint64_t iters = 0;
    uint32_t early = 0;
    double dmax_all = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t it = 0; it < p0.iter_max; ++it) {
        const bool cooling = it >= sh.first_cooling;
        // the per-lane kernel needs four exchanges per iteration to keep the one-GPU quality, the tile kernel one
        // (measured with virtual ranks)
        const uint32_t blocks = (!tiled || (warm_per_lane && !cooling)) ? 4u : 1u;
        double dmax = 0;
        for (uint32_t b = 0; b < blocks; ++b) {
            const bool ok = !sh.failed.load();
            if (ok) R_TRY(pgsgd_session_iteration_part(s, sh.etas[it], cooling ? 1 : 0, my_terms, b, blocks));
            if (ok) R_TRY(pgsgd_session_exchange_begin(s, buf));
            if (sh.rccl) {
                // every rank must issue the collective or none: decide together
                sh.barrier.wait();
                if (!sh.failed.load()) {
                    const int e = sh.rccl->all_reduce(buf, buf, 6 * N, 7 /* ncclFloat32 */, 0 /* ncclSum */, sh.comms[r], stream);
                    if (e != 0 && !sh.rc[r]) {
                        sh.rc[r] = PGSGD_E_HIP;
                        sh.err[r] = std::string("ncclAllReduce: ") + (sh.rccl->error_string ? sh.rccl->error_string(e) : "error");
                        sh.failed.store(1);
                    }
                }
            } else {  // sum through pinned host memory (tests on a single GPU)
                if (ok) R_HIP(hipMemcpyAsync(sh.host_buf[r], buf, 6 * N * sizeof(float), hipMemcpyDeviceToHost, stream));
                if (ok) R_HIP(hipStreamSynchronize(stream));
                sh.barrier.wait();
                if (!sh.failed.load()) {
                    float* mine = sh.host_buf[G + r];  // a second slot per rank for the sum
                    const uint64_t n = 6 * N;
                    for (uint64_t i = 0; i < n; ++i) {
                        float acc = 0;
                        for (int q = 0; q < G; ++q) acc += sh.host_buf[q][i];
                        mine[i] = acc;
                    }
                }
                sh.barrier.wait();  // everybody has read the ranks' buffers
                if (!sh.failed.load()) R_HIP(hipMemcpyAsync(buf, sh.host_buf[G + r], 6 * N * sizeof(float), hipMemcpyHostToDevice, stream));
            }
            if (!sh.failed.load()) R_TRY(pgsgd_session_exchange_end(s, buf, G));
            double d = 0;
            if (!sh.failed.load()) R_TRY(pgsgd_session_sync(s, &d));
            dmax = std::max(dmax, d);
        }
        // max |Delta| over the ranks (the reference's stop rule) and their frame-guard flags, through host memory
        int hit = 0;
        if (s && !sh.failed.load()) (void)pgsgd_session_frame_status(s, &hit, nullptr);
        sh.dmax[r] = dmax;
        sh.guard[r] = hit;
        sh.barrier.wait();
        dmax_all = *std::max_element(sh.dmax.begin(), sh.dmax.end());
        const bool any_guard = std::any_of(sh.guard.begin(), sh.guard.end(), [](int v) { return v != 0; });
        const bool failed = sh.failed.load() != 0;
        sh.barrier.wait();  // everybody has read dmax / guard / failed before anyone writes them again
        if (failed) break;
        if (any_guard) R_TRY(pgsgd_session_reframe(s));
        ++iters;
        if (r == 0 && p0.progress)
            fprintf(stderr, "\r[odgi::path_linear_sgd_layout] 2D path-guided SGD on %d GPUs: iteration %llu/%llu  eta %.4g  delta_max %.4g   ", G,
                    (unsigned long long)(it + 1), (unsigned long long)p0.iter_max, sh.etas[it], dmax_all);
        if (it + 1 >= p0.iter_max) break;
        if (dmax_all <= p0.delta) {  // path_sgd_layout.cpp:142 — the same decision on every rank
            early = 1;
            break;
        }
    }
    if (r == 0 && p0.progress) fprintf(stderr, "\n");
    if (r == 0 && s && !sh.failed.load()) {
        if (sh.Xd) R_TRY(pgsgd_session_download_coords_f64(s, sh.Xd, sh.Yd));
        else R_TRY(pgsgd_session_download_coords(s, sh.X, sh.Y));
        sh.stats.iterations = iters;
        sh.stats.term_updates = iters * M;
        sh.stats.last_delta_max = dmax_all;
        sh.stats.n_streams = pgsgd_session_n_streams(s);
        sh.stats.early_stop = early;
        uint32_t doublings = 0;
        (void)pgsgd_session_frame_status(s, nullptr, &doublings);
        sh.stats.frame_doublings = doublings;
        (void)pgsgd_session_kernel_time(s, &sh.stats.kernel_ms, nullptr, 0);
        sh.stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (buf) (void)hipFree(buf);
    if (s) pgsgd_session_destroy(s);
}

}  // namespace

// X,Y as in pgsgd_layout_run (fp32) or, when Xd/Yd are given, doubles out (pgsgd_layout_run_f64).
int pgsgd_layout_run_multi(const pgsgd_graph_view* g, const pgsgd_params* p, float* X, float* Y, double* Xd, double* Yd, pgsgd_stats* stats) {
    pgsgd::clear_error();
    if (stats) memset(stats, 0, sizeof *stats);
    if (!g || !p || !X || !Y || p->n_devices < 2) return PGSGD_E_INVALID;
    const int G = (int)p->n_devices;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device available; the layout kernels run on MI355X only, there is no CPU fallback");
        return PGSGD_E_NODEVICE;
    }
    const bool host_reduce = getenv("PGSGD_MULTI_HOST_REDUCE") != nullptr;
    if (G > count && !host_reduce) {
        set_error("%d GPUs requested, %d present", G, count);
        return PGSGD_E_NODEVICE;
    }
    Shared sh(G);
    sh.g = g;
    sh.p = p;
    sh.devices.resize(G);
    const int first = p->device < 0 ? 0 : p->device;
    for (int r = 0; r < G; ++r) sh.devices[r] = host_reduce ? (first + r) % count : first + r;  // virtual ranks may share a GPU
    if (!host_reduce && first + G > count) { set_error("devices %d..%d requested, %d present", first, first + G - 1, count); return PGSGD_E_NODEVICE; }
    sh.etas.resize(p->iter_max + 1);
    if (pgsgd_schedule(p, sh.etas.data(), sh.etas.size()) < 0) return PGSGD_E_INVALID;
    sh.first_cooling = (uint64_t)std::floor(p->cooling_start * (double)p->iter_max);
    sh.dmax.assign(G, 0);
    sh.guard.assign(G, 0);
    sh.rc.assign(G, 0);
    sh.err.assign(G, "");
    sh.X0 = X; sh.Y0 = Y; sh.X = X; sh.Y = Y; sh.Xd = Xd; sh.Yd = Yd;
    Rccl rccl;
    if (!host_reduce) {
        if (!rccl.load()) { set_error("n_devices = %d needs RCCL: librccl.so could not be loaded (%s)", G, dlerror()); return PGSGD_E_UNSUPPORTED; }
        sh.comms.resize(G);
        const int e = rccl.init_all(sh.comms.data(), G, sh.devices.data());
        if (e != 0) { set_error("ncclCommInitAll on %d devices: %s", G, rccl.error_string ? rccl.error_string(e) : "error"); return PGSGD_E_HIP; }
        sh.rccl = &rccl;
    } else {
        sh.host_buf.assign(2 * G, nullptr);
        for (int i = 0; i < 2 * G; ++i)
            if (hipHostMalloc((void**)&sh.host_buf[i], 6 * g->n_nodes * sizeof(float)) != hipSuccess) { set_error("pinned exchange buffers"); return PGSGD_E_NOMEM; }
    }
    std::vector<std::thread> th;
    for (int r = 1; r < G; ++r) th.emplace_back(rank_main, std::ref(sh), r);
    rank_main(sh, 0);
    for (auto& t : th) t.join();
    if (sh.rccl) for (ncclComm_t c : sh.comms) (void)rccl.destroy(c);
    for (float* b : sh.host_buf) if (b) (void)hipHostFree(b);
    for (int r = 0; r < G; ++r)
        if (sh.rc[r]) {
            set_error("GPU rank %d: %s", r, sh.err[r].c_str());
            return sh.rc[r];
        }
    if (stats) *stats = sh.stats;
    return PGSGD_OK;
}
