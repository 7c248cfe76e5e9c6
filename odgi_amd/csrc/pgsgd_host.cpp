// pgsgd_host.cpp — device-independent host logic of the layout path: error reporting, reference
// defaults, learning-rate schedule, Zipf zeta cache, initial layouts.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>

#include "pgsgd_internal.hpp"
#include "pgsgd_math.hpp"

namespace pgsgd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
void clear_error() { g_err[0] = 0; }
}  // namespace pgsgd

extern "C" const char* pgsgd_last_error(void) { return pgsgd::g_err; }

extern "C" int pgsgd_abi_version(void) { return PGSGD_ABI_VERSION; }
extern "C" void pgsgd_abi_struct_sizes(size_t* graph_view, size_t* params, size_t* stats) {
    if (graph_view) *graph_view = sizeof(pgsgd_graph_view);
    if (params) *params = sizeof(pgsgd_params);
    if (stats) *stats = sizeof(pgsgd_stats);
}

extern "C" const char* pgsgd_strerror(int code) {
    switch (code) {
        case PGSGD_OK: return "ok";
        case PGSGD_E_INVALID: return "invalid argument";
        case PGSGD_E_NODEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case PGSGD_E_HIP: return "HIP runtime error";
        case PGSGD_E_NOMEM: return "out of memory";
        case PGSGD_E_IO: return "I/O error";
        case PGSGD_E_FORMAT: return "malformed input";
        case PGSGD_E_NOTOPTIMIZED: return "graph is not optimized (node ids must be exactly 1..N)";
        case PGSGD_E_UNSUPPORTED: return "unsupported";
        default: return "unknown error";
    }
}

int pgsgd_validate_view(const pgsgd_graph_view* g) {
    using pgsgd::set_error;
    if (!g) { set_error("graph view is NULL"); return PGSGD_E_INVALID; }
    if (g->n_nodes == 0 || g->n_nodes > 0x7fffffffull) { set_error("n_nodes %llu out of range [1, 2^31)", (unsigned long long)g->n_nodes); return PGSGD_E_INVALID; }
    if (g->n_paths >= 0xffffffffull) { set_error("too many paths"); return PGSGD_E_INVALID; }
    // (step_path and step_pos may be NULL: both follow from path_first, step_handle and node_len — the session builds the positions on
    // the device, host code that needs them walks the paths)
    if (!g->node_len || !g->path_first || (g->n_steps && !g->step_handle)) {
        set_error("graph view has NULL arrays");
        return PGSGD_E_INVALID;
    }
    if (g->path_first[0] != 0 || g->path_first[g->n_paths] != g->n_steps) {
        set_error("path_first must start at 0 and end at n_steps");
        return PGSGD_E_INVALID;
    }
    for (uint64_t p = 0; p < g->n_paths; ++p)
        if (g->path_first[p + 1] < g->path_first[p]) { set_error("path_first not monotone at %llu", (unsigned long long)p); return PGSGD_E_INVALID; }
    return PGSGD_OK;
}

static uint64_t max_path_steps(const pgsgd_graph_view* g) {
    uint64_t m = 0;
    for (uint64_t p = 0; p < g->n_paths; ++p) m = std::max(m, g->path_first[p + 1] - g->path_first[p]);
    return m;
}

// reference: src/subcommand/layout_main.cpp:153-155,198-204,251-266 (all paths used: no -f file)
extern "C" int pgsgd_params_defaults(const pgsgd_graph_view* g, pgsgd_params* p) {
    pgsgd::clear_error();
    if (!p) return PGSGD_E_INVALID;
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    std::memset(p, 0, sizeof *p);
    const uint64_t max_steps = max_path_steps(g);
    p->iter_max = 30;
    p->iter_with_max_learning_rate = 0;
    p->min_term_updates = (uint64_t)(10.0 * (double)g->n_steps);
    p->delta = 0;
    p->eps = 0.01;
    p->eta_max = (double)max_steps * (double)max_steps;
    p->theta = 0.99;
    p->space = max_steps;
    p->space_max = 1000;
    p->space_quantization_step = 100;
    p->cooling_start = 0.5;
    p->seed = PGSGD_DEFAULT_SEED;
    p->n_streams = 0;
    p->stream_offset = 0;
    p->device = -1;
    p->terms_per_anchor = 1;
    p->n_devices = 1;
    return PGSGD_OK;
}

// reference: path_linear_sgd_layout_schedule, src/algorithms/path_sgd_layout.cpp:433-468,
// called with w_min = 1/eta_max, w_max = 1 (:76-84)
extern "C" int64_t pgsgd_schedule(const pgsgd_params* p, double* etas, size_t capacity) {
    pgsgd::clear_error();
    if (!p || !etas || p->iter_max == 0) return PGSGD_E_INVALID;
    if (capacity < p->iter_max + 1) { pgsgd::set_error("schedule needs iter_max+1 slots"); return PGSGD_E_INVALID; }
    const double w_min = 1.0 / p->eta_max, w_max = 1.0;
    const double eta_max = 1.0 / w_min;
    const double eta_min = p->eps / w_max;
    const double lambda = std::log(eta_max / eta_min) / ((double)p->iter_max - 1);
    for (int64_t t = 0; t <= (int64_t)p->iter_max; t++)
        etas[t] = eta_max * std::exp(-lambda * (double)std::llabs(t - (int64_t)p->iter_with_max_learning_rate));
    return (int64_t)p->iter_max + 1;
}

// reference: path_sgd_layout.cpp:86-97.  One slot more than the reference allocates, so its
// out-of-bounds write at space == space_max (index space_max+1) lands in the table.
extern "C" size_t pgsgd_zeta_table_size(uint64_t space, uint64_t space_max, uint64_t quant) {
    if (quant == 0) quant = 1;
    return (size_t)((space <= space_max ? space : space_max + (space - space_max) / quant + 1) + 1) + 1;
}

extern "C" int pgsgd_zeta_table(double theta, uint64_t space, uint64_t space_max, uint64_t quant,
                                double* zetas, size_t capacity) {
    pgsgd::clear_error();
    if (!zetas || quant == 0) return PGSGD_E_INVALID;
    const size_t n = pgsgd_zeta_table_size(space, space_max, quant);
    if (capacity < n) { pgsgd::set_error("zeta table needs %zu slots", n); return PGSGD_E_INVALID; }
    std::fill(zetas, zetas + n, 0.0);
    double zeta_tmp = 0.0;
    for (uint64_t i = 1; i < space + 1; i++) {
        zeta_tmp += pgsgd::fast_precise_pow(1.0 / (double)i, theta);
        if (i <= space_max) zetas[i] = zeta_tmp;
        if (i >= space_max && (i - space_max) % quant == 0) zetas[space_max + 1 + (i - space_max) / quant] = zeta_tmp;
    }
    return PGSGD_OK;
}

// Hilbert curve index -> (x,y), reference: src/algorithms/hilbert.hpp:5-41.  The reference keeps
// its loop variables in `int`; 2N stays below 2^31 here (validated), so plain 64-bit is identical.
static void hilbert_rot(uint64_t n, uint64_t* x, uint64_t* y, uint64_t rx, uint64_t ry) {
    if (ry == 0) {
        if (rx == 1) {
            *x = n - 1 - *x;
            *y = n - 1 - *y;
        }
        const uint64_t t = *x;
        *x = *y;
        *y = t;
    }
}
static void hilbert_d2xy(uint64_t n, uint64_t d, uint64_t* x, uint64_t* y) {
    uint64_t t = d;
    *x = *y = 0;
    for (uint64_t s = 1; s < n; s *= 2) {
        const uint64_t rx = 1 & (t / 2);
        const uint64_t ry = 1 & (t ^ rx);
        hilbert_rot(s, x, y, rx, ry);
        *x += s * rx;
        *y += s * ry;
        t /= 4;
    }
}

// reference: src/subcommand/layout_main.cpp:268-330.  seed == 0 -> std::random_device, as upstream.
extern "C" int pgsgd_init_layout(const pgsgd_graph_view* g, char mode, uint64_t seed, double* X, double* Y) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!X || !Y) return PGSGD_E_INVALID;
    const uint64_t N = g->n_nodes;
    std::mt19937 rng;
    if (seed == 0) {
        std::random_device dev;
        rng.seed(dev());
    } else {
        rng.seed((std::mt19937::result_type)seed);
    }
    std::uniform_real_distribution<double> uniform_noise(0, std::sqrt((double)(N * 2)));
    std::normal_distribution<double> gaussian_noise(0, std::sqrt((double)(N * 2)));
    uint64_t total_length = 0;
    for (uint64_t i = 0; i < N; ++i) total_length += g->node_len[i];
    std::uniform_real_distribution<double> uniform_noise_in_length(0, (double)total_length);
    uint64_t len = 0;
    const uint64_t square_space = N * 2;
    uint64_t x, y;
    for (uint64_t i = 0; i < N; ++i) {
        const uint64_t pos = 2 * i;
        switch (mode) {
            case 'g':
                X[pos] = gaussian_noise(rng);
                Y[pos] = gaussian_noise(rng);
                X[pos + 1] = gaussian_noise(rng);
                Y[pos + 1] = gaussian_noise(rng);
                break;
            case 'u':
                X[pos] = (double)len;
                Y[pos] = uniform_noise(rng);
                len += g->node_len[i];
                X[pos + 1] = (double)len;
                Y[pos + 1] = uniform_noise(rng);
                break;
            case 'r':
                X[pos] = uniform_noise_in_length(rng);
                Y[pos] = uniform_noise_in_length(rng);
                X[pos + 1] = uniform_noise_in_length(rng);
                Y[pos + 1] = uniform_noise_in_length(rng);
                break;
            case 'h':
                hilbert_d2xy(square_space, pos, &x, &y);
                X[pos] = (double)x;
                Y[pos] = (double)y;
                hilbert_d2xy(square_space, pos + 1, &x, &y);
                X[pos + 1] = (double)x;
                Y[pos + 1] = (double)y;
                break;
            default:  // 'd'
                X[pos] = (double)len;
                Y[pos] = gaussian_noise(rng);
                len += g->node_len[i];
                X[pos + 1] = (double)len;
                Y[pos + 1] = gaussian_noise(rng);
        }
    }
    return PGSGD_OK;
}

// ---------------------------------------------------------------------------------------------
// Node order by path position.  The tile kernel (what runs at 6e10 terms/s) needs node ranks that follow the paths: a run
// of path steps must visit a run of node ranks.  `odgi layout` is normally fed a sorted graph (odgi sort), but nothing
// in the reference requires it, and a graph whose ids are in another order would fall back to the per-lane kernel
// (1e10 terms/s).  The ranks are only names: pgsgd_layout_run renames them for the duration of a run when that helps —
// nodes ordered by (path-connected component, mean bp position of the node's steps) — and hands the coordinates back
// under the caller's names.  Whether it helps is measured on a sample of consecutive steps: the share of steps whose
// successor lies more than `reach` ranks away.
namespace pgsgd {
double step_rank_disorder(const pgsgd_graph_view* g, const uint32_t* new_rank_of_old, uint32_t reach) {
    const uint64_t S = g->n_steps;
    if (S < 2) return 0.0;
    const uint64_t K = std::min<uint64_t>(S - 1, 200000);
    uint64_t seen = 0, far = 0;
    for (uint64_t i = 0; i < K; ++i) {
        const uint64_t k = (uint64_t)(((unsigned __int128)i * (S - 1)) / K);
        // k and k + 1 on one path: k + 1 is not a path's first step
        if (std::binary_search(g->path_first, g->path_first + g->n_paths + 1, k + 1)) continue;
        uint32_t a = g->step_handle[k] >> 1, b = g->step_handle[k + 1] >> 1;
        if (new_rank_of_old) { a = new_rank_of_old[a]; b = new_rank_of_old[b]; }
        ++seen;
        far += (a > b ? a - b : b - a) > reach;
    }
    return seen ? (double)far / (double)seen : 0.0;
}
}  // namespace pgsgd

extern "C" int pgsgd_graph_path_order(const pgsgd_graph_view* g, uint32_t* new_rank_of_old, double* disorder_before, double* disorder_after) {
    pgsgd::clear_error();
    int rc = pgsgd_validate_view(g);
    if (rc) return rc;
    if (!new_rank_of_old) return PGSGD_E_INVALID;
    const uint64_t N = g->n_nodes, S = g->n_steps;
    for (uint64_t k = 0; k < S; ++k)
        if ((g->step_handle[k] >> 1) >= N) { pgsgd::set_error("a step names a node rank outside the graph"); return PGSGD_E_INVALID; }
    // components of "visited by a common path, transitively": union-find over consecutive steps
    std::vector<uint32_t> parent(N);
    for (uint64_t i = 0; i < N; ++i) parent[i] = (uint32_t)i;
    auto find = [&](uint32_t x) {
        while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
        return x;
    };
    std::vector<double> sum(N, 0.0);
    std::vector<uint32_t> cnt(N, 0);
    for (uint64_t p = 0; p < g->n_paths; ++p) {   // (path by path: positions and "same path" need neither step_pos nor step_path)
        uint64_t pos = 0;
        for (uint64_t k = g->path_first[p]; k < g->path_first[p + 1]; ++k) {
            const uint32_t n = g->step_handle[k] >> 1;
            sum[n] += (double)pos;
            pos += g->node_len[n];
            cnt[n]++;
            if (k + 1 < g->path_first[p + 1]) {
                const uint32_t a = find(n), b = find(g->step_handle[k + 1] >> 1);
                if (a != b) parent[a < b ? b : a] = a < b ? a : b;   // the smaller rank is the component's name
            }
        }
    }
    struct Key { uint32_t comp; double pos; uint32_t old; };
    std::vector<Key> keys(N);
    for (uint64_t i = 0; i < N; ++i) keys[i] = Key{find((uint32_t)i), cnt[i] ? sum[i] / (double)cnt[i] : 0.0, (uint32_t)i};
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
        return a.comp != b.comp ? a.comp < b.comp : a.pos != b.pos ? a.pos < b.pos : a.old < b.old;
    });
    for (uint64_t r = 0; r < N; ++r) new_rank_of_old[keys[r].old] = (uint32_t)r;
    if (disorder_before) *disorder_before = pgsgd::step_rank_disorder(g, nullptr, 128);
    if (disorder_after) *disorder_after = pgsgd::step_rank_disorder(g, new_rank_of_old, 128);
    return PGSGD_OK;
}
