// gfa_lower.cpp — GFA v1 -> the flat path-step index the SGD kernels read.
//
// Replaces, for `odgi layout`, the chain  gfa_to_handle (src/gfa_to_handle.cpp:27-217)
//   -> graph_t -> XP::from_handle_graph (src/algorithms/xp.cpp:49-175, XPPath :543-649).
// Only S, L and P lines are consumed (gfa_to_handle.cpp:45,72,108,193); segment names must be
// non-negative integers (:75-79); duplicate ids (:89-93) and path steps on missing nodes (:162-170)
// are errors; node rank = id-1 and the graph must be "optimized" = ids exactly 1..N
// (src/odgi.cpp:752-758, enforced by layout_main.cpp:148-151); paths are numbered in P-line order
// (odgi.cpp:1233-1234); the position of a step is the bp offset of its start in the path
// (xp.cpp:607-617).  Instead of exit(1) every failure is an error code.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <thread>
#include <tuple>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "pgsgd_internal.hpp"

namespace {

struct Line { const char* b; const char* e; };

inline const char* next_tab(const char* p, const char* e) {
    while (p < e && *p != '\t') ++p;
    return p;
}

bool parse_uint(const char* b, const char* e, uint64_t* out) {
    if (b >= e) return false;
    uint64_t v = 0;
    for (const char* p = b; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        const uint64_t d = (uint64_t)(*p - '0');
        if (v > (UINT64_MAX - d) / 10) return false;  // does not fit 64 bits: not an id
        v = v * 10 + d;
    }
    *out = v;
    return true;
}

}  // namespace

extern "C" int pgsgd_graph_from_gfa(const char* path, int n_threads, pgsgd_graph** out) {
    using pgsgd::set_error;
    pgsgd::clear_error();
    if (!path || !out) return PGSGD_E_INVALID;
    *out = nullptr;
    pgsgd::PhaseTimer timer;
    // The file is mapped, not copied: the parse below reads every byte once or twice, from the page cache.
    struct Mapping {
        const char* p = nullptr;
        size_t n = 0;
        std::string fallback;  // pipes and other things that cannot be mapped are read
        ~Mapping() { if (p && fallback.empty() && n) munmap(const_cast<char*>(p), n); }
    } file;
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) { set_error("cannot open '%s'", path); return PGSGD_E_IO; }
        struct stat st;
        void* m = MAP_FAILED;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m != MAP_FAILED) {
                file.p = (const char*)m;
                file.n = (size_t)st.st_size;
                (void)madvise(m, file.n, MADV_SEQUENTIAL);
            }
        }
        if (m == MAP_FAILED) {
            try {
                char chunk[1 << 16];
                ssize_t r;
                while ((r = read(fd, chunk, sizeof chunk)) > 0) file.fallback.append(chunk, (size_t)r);
                if (r < 0) { close(fd); set_error("cannot read '%s'", path); return PGSGD_E_IO; }
            } catch (...) { close(fd); return PGSGD_E_NOMEM; }
            file.p = file.fallback.data();
            file.n = file.fallback.size();
        }
        close(fd);
    }
    struct { const char* p; size_t n; const char* data() const { return p; } size_t size() const { return n; } } buf{file.p, file.n};

    timer.lap("gfa: read file");
    const int nt_lines = std::max(1, std::min(n_threads, 64));
    bool threads_ok = true;  // false: a worker ran out of memory (checked after every parallel section)
    // fn(t, begin, end) on contiguous slices of [0, n): on threads when there is enough to split
    auto parallel_slices = [&](uint64_t n, uint64_t min_per_thread, auto&& fn) {
        const unsigned k = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)nt_lines, n / std::max<uint64_t>(1, min_per_thread)));
        if (!pgsgd::run_threads(k, [&](unsigned t) { fn(t, n * t / k, n * (t + 1) / k); })) threads_ok = false;
        return k;
    };
    // lines: the file is cut at newlines into one range per thread, every range lists its S, L and P lines, the lists
    // are joined in file order
    std::vector<Line> s_lines, l_lines, p_lines;
    {
        const char* base = buf.data();
        const char* end = base + buf.size();
        struct Lists { std::vector<Line> s, l, p; };
        std::vector<Lists> part((size_t)nt_lines);
        const unsigned k = parallel_slices(buf.size(), 1 << 22, [&](unsigned t, uint64_t b0, uint64_t b1) {
            // a range starts at the first line that begins inside it
            const char* p = base + b0;
            if (b0 > 0) {
                const char* nl = (const char*)memchr(p - 1, '\n', (size_t)(end - (p - 1)));
                p = nl ? nl + 1 : end;
            }
            const char* stop = base + b1;  // lines that BEGIN before stop belong to this range
            Lists& L = part[t];
            while (p < end && p < stop) {
                const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
                const char* le = nl ? nl : end;
                const char* e = le;
                if (e > p && e[-1] == '\r') --e;
                if (e - p >= 2 && p[1] == '\t') {
                    if (p[0] == 'S') L.s.push_back({p, e});
                    else if (p[0] == 'L') L.l.push_back({p, e});
                    else if (p[0] == 'P') L.p.push_back({p, e});
                }
                p = nl ? nl + 1 : end;
            }
        });
        for (unsigned t = 0; t < k; ++t) {
            s_lines.insert(s_lines.end(), part[t].s.begin(), part[t].s.end());
            l_lines.insert(l_lines.end(), part[t].l.begin(), part[t].l.end());
            p_lines.insert(p_lines.end(), part[t].p.begin(), part[t].p.end());
        }
    }
    if (s_lines.empty()) { set_error("'%s' has no S lines", path); return PGSGD_E_FORMAT; }
    if (!threads_ok) { set_error("out of memory while reading '%s'", path); return PGSGD_E_NOMEM; }
    timer.lap("gfa: split lines");

    auto g = new pgsgd_graph();
    const uint64_t N = s_lines.size();
    g->n_nodes = N;
    g->node_len.assign(N, 0);
    // nodes and edges: slices of the line lists on threads; the error reported is the first in file order
    struct LineError { uint64_t line = UINT64_MAX; int code = 0; std::string msg; };
    auto first_error = [](const std::vector<LineError>& errs) {
        const LineError* best = nullptr;
        for (const LineError& e : errs)
            if (e.code && (!best || e.line < best->line)) best = &e;
        return best;
    };
    {
        std::vector<uint64_t> seen(N, 0);  // 1 + the S line that claimed the id (0: none yet)
        std::vector<LineError> errs((size_t)nt_lines);
        std::vector<uint64_t> lo((size_t)nt_lines, UINT64_MAX), hi((size_t)nt_lines, 0);
        const unsigned k = parallel_slices(N, 1 << 16, [&](unsigned t, uint64_t b0, uint64_t b1) {
            uint64_t min_id = UINT64_MAX, max_id = 0;
            for (uint64_t i = b0; i < b1; ++i) {
                const Line& ln = s_lines[i];
                const char* nb = ln.b + 2;
                const char* ne = next_tab(nb, ln.e);
                uint64_t id;
                if (!parse_uint(nb, ne, &id)) {
                    errs[t].line = i;
                    errs[t].code = PGSGD_E_FORMAT;
                    errs[t].msg = "segment name '" + std::string(nb, ne) + "' is not a non-negative integer node id";
                    break;
                }
                min_id = std::min(min_id, id);
                max_id = std::max(max_id, id);
                const char* sb = ne < ln.e ? ne + 1 : ln.e;
                const char* se = next_tab(sb, ln.e);
                if (id >= 1 && id <= N) {
                    if (const uint64_t prev = __atomic_exchange_n(&seen[id - 1], i + 1, __ATOMIC_RELAXED)) {
                        // the duplicate is the LATER of the two lines, whichever thread got there first (a reader going
                        // through the file in order reports it there); this slice goes on unless that line is its own
                        const uint64_t dup = std::max(i, prev - 1);
                        if (dup < errs[t].line) {
                            errs[t].line = dup;
                            errs[t].code = PGSGD_E_FORMAT;
                            errs[t].msg = "duplicate node id " + std::to_string(id);
                        }
                        if (dup == i) break;
                        continue;
                    }
                    g->node_len[id - 1] = (uint32_t)(se - sb);
                }
            }
            lo[t] = min_id;
            hi[t] = max_id;
        });
        if (const LineError* e = first_error(errs)) {
            set_error("%s", e->msg.c_str());
            const int code = e->code;
            delete g;
            return code;
        }
        uint64_t min_id = UINT64_MAX, max_id = 0;
        for (unsigned t = 0; t < k; ++t) { min_id = std::min(min_id, lo[t]); max_id = std::max(max_id, hi[t]); }
        if (!(min_id == 1 && max_id == N)) {
            set_error("the graph is not optimized: node ids span [%llu, %llu] for %llu nodes", (unsigned long long)min_id,
                      (unsigned long long)max_id, (unsigned long long)N);
            delete g;
            return PGSGD_E_NOTOPTIMIZED;
        }
    }
    if (!threads_ok) { set_error("out of memory while reading '%s'", path); delete g; return PGSGD_E_NOMEM; }
    timer.lap("gfa: S lines");
    // edges (only needed for the weakly-connected-component post step)
    {
        std::vector<std::vector<uint64_t>> part((size_t)nt_lines);
        std::vector<LineError> errs((size_t)nt_lines);
        const unsigned k = parallel_slices(l_lines.size(), 1 << 16, [&](unsigned t, uint64_t b0, uint64_t b1) {
            std::vector<uint64_t>& out = part[t];
            out.reserve((size_t)(b1 - b0) * 2);
            for (uint64_t i = b0; i < b1; ++i) {
                const Line& ln = l_lines[i];
                const char* f[5];
                const char* fe[5];
                const char* p = ln.b + 2;
                int nf = 0;
                while (nf < 4 && p <= ln.e) {
                    const char* tb = next_tab(p, ln.e);
                    f[nf] = p; fe[nf] = tb; ++nf;
                    p = tb + 1;
                }
                if (nf < 4 || fe[0] == f[0]) continue;  // gfa_to_handle.cpp:111 skips empty sources
                uint64_t a, b;
                if (!parse_uint(f[0], fe[0], &a) || !parse_uint(f[2], fe[2], &b) || a < 1 || a > N || b < 1 || b > N) {
                    errs[t].line = i;
                    errs[t].code = PGSGD_E_FORMAT;
                    errs[t].msg = "edge '" + std::string(f[0], fe[0]) + " <--> " + std::string(f[2], fe[2]) + "' names a missing node";
                    break;
                }
                const uint64_t ra = (fe[1] > f[1] && f[1][0] == '-') ? 1 : 0;
                const uint64_t rb = (fe[3] > f[3] && f[3][0] == '-') ? 1 : 0;
                out.push_back(2 * (a - 1) + ra);
                out.push_back(2 * (b - 1) + rb);
            }
        });
        if (const LineError* e = first_error(errs)) {
            set_error("%s", e->msg.c_str());
            const int code = e->code;
            delete g;
            return code;
        }
        size_t total = 0;
        for (unsigned t = 0; t < k; ++t) total += part[t].size();
        g->edges.reserve(total);
        for (unsigned t = 0; t < k; ++t) g->edges.insert(g->edges.end(), part[t].begin(), part[t].end());
    }
    if (!threads_ok) { set_error("out of memory while reading '%s'", path); delete g; return PGSGD_E_NOMEM; }
    timer.lap("gfa: L lines");
    // graph_t::create_edge ignores an edge that exists (odgi.cpp:611-631), and a -> b is the same edge as
    // flip(b) -> flip(a): keep the first occurrence of each, in file order
    {
        const uint64_t E = g->edges.size() / 2;
        struct Key { uint64_t a, b, idx; };
        std::vector<Key> keys(E);
        for (uint64_t i = 0; i < E; ++i) {
            const uint64_t a = g->edges[2 * i], b = g->edges[2 * i + 1];
            const bool swap = std::make_pair(b ^ 1, a ^ 1) < std::make_pair(a, b);
            keys[i] = {swap ? (b ^ 1) : a, swap ? (a ^ 1) : b, i};
        }
        std::sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return std::tie(x.a, x.b, x.idx) < std::tie(y.a, y.b, y.idx); });
        std::vector<uint64_t> keep;
        for (uint64_t i = 0; i < E; ++i)
            if (i == 0 || keys[i].a != keys[i - 1].a || keys[i].b != keys[i - 1].b) keep.push_back(keys[i].idx);
        if (keep.size() != E) {
            std::sort(keep.begin(), keep.end());
            std::vector<uint64_t> uniq;
            uniq.reserve(2 * keep.size());
            for (uint64_t i : keep) { uniq.push_back(g->edges[2 * i]); uniq.push_back(g->edges[2 * i + 1]); }
            g->edges.swap(uniq);
        }
    }
    timer.lap("gfa: distinct edges");
    // Paths are numbered in file order.  The segment list of a P line is cut into chunks at commas, so that one
    // very long path (a reference chromosome next to fragmented assemblies) is parsed by all threads too:
    // pass 1 counts the steps of every chunk, pass 2 parses them straight into the step arrays and sums the bp of
    // each chunk, pass 3 turns the sums into the bp offset of every step (xp.cpp:607-617).
    const uint64_t P = p_lines.size();
    g->path_names.resize(P);
    struct Chunk { uint64_t path; const char* b; const char* e; uint64_t steps = 0, first = 0, bp = 0, pos0 = 0; int err = 0; std::string msg; };
    std::vector<Chunk> chunks;
    std::vector<uint64_t> first_chunk(P + 1, 0);
    const size_t kChunkBytes = 1 << 18;
    for (uint64_t i = 0; i < P; ++i) {
        const Line& ln = p_lines[i];
        const char* nb = ln.b + 2;
        const char* ne = next_tab(nb, ln.e);
        g->path_names[i].assign(nb, ne);
        const char* sb = ne < ln.e ? ne + 1 : ln.e;
        const char* se = next_tab(sb, ln.e);
        first_chunk[i] = chunks.size();
        const char* b = sb;
        while (b < se) {
            const char* e = (size_t)(se - b) > kChunkBytes ? (const char*)memchr(b + kChunkBytes, ',', (size_t)(se - b) - kChunkBytes) : nullptr;
            e = e ? e + 1 : se;  // a chunk ends right after a comma
            Chunk c;
            c.path = i;
            c.b = b;
            c.e = e;
            chunks.push_back(c);
            b = e;
        }
    }
    first_chunk[P] = chunks.size();
    const int nt = std::max(1, std::min(n_threads, 64));
    auto for_each_chunk = [&](auto&& fn) {
        std::atomic<uint64_t> next{0};
        auto body = [&]() {
            for (uint64_t ci = next.fetch_add(1); ci < chunks.size(); ci = next.fetch_add(1)) fn(chunks[ci]);
        };
        if (!pgsgd::run_threads((unsigned)std::max<size_t>(1, std::min<size_t>((size_t)nt, chunks.size())), [&](unsigned) { body(); })) threads_ok = false;
    };
    // a token is what lies between commas; empty tokens and the placeholder "*" are not steps
    // pass 1: steps = tokens - empty tokens - "*" tokens, each counted by a loop without carried state (the compiler
    // vectorises them; a memchr per 7-byte token costs more than the token)
    for_each_chunk([&](Chunk& c) {
        const char* b = c.b;
        const size_t n = (size_t)(c.e - c.b);
        if (n == 0) { c.steps = 0; return; }
        uint64_t commas = 0, empty = (b[0] == ',') + (b[n - 1] == ','), star = 0;
        commas += (b[0] == ',') + (n > 1 && b[n - 1] == ',');
        for (size_t i = 1; i + 1 < n; ++i) {  // one sweep: a comma, a comma after a comma, a "*" between commas
            const bool cm = b[i] == ',', after = b[i - 1] == ',';
            commas += cm;
            empty += cm & after;
            star += (b[i] == '*') & after & (b[i + 1] == ',');
        }
        if (n > 1) empty += (b[n - 1] == ',') & (b[n - 2] == ',');
        star += (b[0] == '*') & (n == 1 || b[1] == ',');
        if (n > 1) star += (b[n - 1] == '*') & (b[n - 2] == ',');
        c.steps = commas + 1 - empty - star;
    });
    g->path_first.assign(P + 1, 0);
    {
        uint64_t k = 0;
        for (uint64_t i = 0; i < P; ++i) {
            g->path_first[i] = k;
            for (uint64_t ci = first_chunk[i]; ci < first_chunk[i + 1]; ++ci) { chunks[ci].first = k; k += chunks[ci].steps; }
        }
        g->path_first[P] = k;
    }
    const uint64_t S = g->path_first[P];
    // PGSGD_LOAD_NO_STEP_INDEX (pgsgd_graph_load_flags): step_path and step_pos are not built — 12 of the index's 16 bytes per step —
    // the session builds the positions on the device from the handles, host code that needs them walks the paths (pgsgd::StepIndex)
    const bool index = !(pgsgd::load_flags() & PGSGD_LOAD_NO_STEP_INDEX);
    if (index) g->step_path.resize(S);
    g->step_handle.resize(S);
    if (index) g->step_pos.resize(S);
    for_each_chunk([&](Chunk& c) {
        uint64_t k = c.first, bp = 0;
        const std::string& name = g->path_names[c.path];
        // the general form of a token, with the reference's error cases (gfa_to_handle.cpp:150-170)
        auto slow_token = [&](const char* p, const char* te) {
            const char orient = te[-1];
            uint64_t id;
            if ((orient != '+' && orient != '-') || !parse_uint(p, te - 1, &id)) {
                c.err = PGSGD_E_FORMAT;
                c.msg = "malformed path segment '" + std::string(p, te) + "' in path '" + name + "'";
                return false;
            }
            if (id < 1 || id > N) {
                c.err = PGSGD_E_FORMAT;
                c.msg = "path '" + name + "' visits missing node '" + std::string(p, te - 1) + "'";
                return false;
            }
            if (index) g->step_path[k] = (uint32_t)c.path;
            g->step_handle[k] = (uint32_t)(2 * (id - 1) + (orient == '-' ? 1 : 0));
            bp += g->node_len[id - 1];
            ++k;
            return true;
        };
        // the usual form — up to 18 digits, an orientation, a comma or the end — is parsed in one sweep over its bytes;
        // anything else (empty tokens, "*", errors) goes through slow_token
        const char* p = c.b;
        while (p < c.e) {
            const char* t = p;
            uint64_t id = 0;
            while (p < c.e && (unsigned)(*p - '0') < 10u && p - t < 18) id = id * 10 + (uint64_t)(*p++ - '0');
            if (p > t && p < c.e && (*p == '+' || *p == '-') && (p + 1 == c.e || p[1] == ',') && id >= 1 && id <= N) {
                if (index) g->step_path[k] = (uint32_t)c.path;
                g->step_handle[k] = (uint32_t)(2 * (id - 1) + (*p == '-' ? 1 : 0));
                bp += g->node_len[id - 1];
                ++k;
                p += 2;  // past the orientation and the comma
                continue;
            }
            const char* cm = (const char*)memchr(t, ',', (size_t)(c.e - t));
            const char* te = cm ? cm : c.e;
            if (te > t && !(te - t == 1 && *t == '*'))
                if (!slow_token(t, te)) return;
            p = cm ? cm + 1 : c.e;
        }
        c.bp = bp;
    });
    if (!threads_ok) { set_error("out of memory while reading '%s'", path); delete g; return PGSGD_E_NOMEM; }
    timer.lap("gfa: P lines (threads)");
    for (const Chunk& c : chunks)  // the first error in file order
        if (c.err) {
            set_error("%s", c.msg.c_str());
            const int rc = c.err;
            delete g;
            return rc;
        }
    for (uint64_t i = 0; i < P; ++i) {
        uint64_t pos = 0;
        for (uint64_t ci = first_chunk[i]; ci < first_chunk[i + 1]; ++ci) { chunks[ci].pos0 = pos; pos += chunks[ci].bp; }
    }
    if (index) for_each_chunk([&](Chunk& c) {
        uint64_t pos = c.pos0;
        for (uint64_t k = c.first; k < c.first + c.steps; ++k) {
            g->step_pos[k] = pos;
            pos += g->node_len[g->step_handle[k] >> 1];
        }
    });
    if (!threads_ok) { set_error("out of memory while reading '%s'", path); delete g; return PGSGD_E_NOMEM; }
    timer.lap("gfa: step index (threads)");
    *out = g;
    return PGSGD_OK;
}

// Seeded synthetic "linearised pangenome" (BASELINE.json configs 4 and 5): a sorted
// backbone of N nodes with geometric node lengths (mean 32 bp, clipped to [1,4096]); 5 % of the
// nodes are the alternative allele of the node before them (a bubble: a haplotype takes one of the
// two); each of the P haplotype paths walks the backbone, drops a node with p = 0.02 (deletion) and
// traverses a node in reverse with p = 0.01.
extern "C" int pgsgd_graph_synthetic(uint64_t n_nodes, uint64_t n_paths, uint64_t seed, pgsgd_graph** out) {
    pgsgd::clear_error();
    if (!out || n_nodes < 2 || n_nodes > 0x7fffffffull || n_paths < 1) return PGSGD_E_INVALID;
    auto g = new pgsgd_graph();
    const uint64_t N = n_nodes;
    g->n_nodes = N;
    g->node_len.resize(N);
    auto sm = [](uint64_t& x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto unit = [&](uint64_t& x) { return (double)(sm(x) >> 11) * 0x1p-53; };
    std::vector<uint8_t> is_alt(N, 0);
    {
        uint64_t x = seed ^ 0x6a09e667f3bcc909ull;
        const double q = 1.0 / 32.0;  // geometric, mean 32
        const double lq = std::log1p(-q);
        for (uint64_t i = 0; i < N; ++i) {
            const double u = unit(x);
            double len = std::floor(std::log1p(-u) / lq) + 1.0;
            if (len < 1) len = 1;
            if (len > 4096) len = 4096;
            g->node_len[i] = (uint32_t)len;
            if (i >= 1 && !is_alt[i - 1] && unit(x) < 0.05) is_alt[i] = 1;
        }
        is_alt[N - 1] = 0;  // keep the last backbone node shared
    }
    // backbone edges: i -> i+1 for every consecutive choice (alt pairs get the four bubble edges)
    for (uint64_t i = 0; i + 1 < N; ++i) {
        if (is_alt[i + 1]) {  // bubble {i, i+1}: predecessor handled below, successor i+2
            if (i + 2 < N) {
                g->edges.push_back(2 * i); g->edges.push_back(2 * (i + 2));
                g->edges.push_back(2 * (i + 1)); g->edges.push_back(2 * (i + 2));
            }
        } else if (!is_alt[i]) {
            g->edges.push_back(2 * i); g->edges.push_back(2 * (i + 1));
            if (i + 2 < N && is_alt[i + 2]) { g->edges.push_back(2 * i); g->edges.push_back(2 * (i + 2)); }
        }
    }
    const uint64_t P = n_paths;
    std::vector<std::vector<uint32_t>> walks(P);
    const int nt = (int)std::min<uint64_t>(P, std::max(1u, std::thread::hardware_concurrency()));
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const uint64_t pi = next.fetch_add(1);
            if (pi >= P) break;
            uint64_t x = seed * 0x9e3779b97f4a7c15ull + 0x1234567ull + pi * 0xd1342543de82ef95ull;
            auto& w = walks[pi];
            w.reserve((size_t)((double)N * 0.95));
            uint64_t i = 0;
            while (i < N) {
                uint64_t node = i;
                if (i + 1 < N && is_alt[i + 1]) {
                    if (sm(x) >> 63) node = i + 1;
                    i += 2;
                } else {
                    i += 1;
                }
                if (unit(x) < 0.02) continue;  // deletion
                const uint32_t rev = unit(x) < 0.01 ? 1u : 0u;
                w.push_back((uint32_t)(2 * node + rev));
            }
        }
    };
    if (!pgsgd::run_threads((unsigned)nt, [&](unsigned) { worker(); })) {
        delete g;
        pgsgd::set_error("out of memory while generating the synthetic graph");
        return PGSGD_E_NOMEM;
    }
    g->path_first.assign(P + 1, 0);
    g->path_names.resize(P);
    for (uint64_t pi = 0; pi < P; ++pi) {
        g->path_first[pi + 1] = g->path_first[pi] + walks[pi].size();
        g->path_names[pi] = "hap" + std::to_string(pi);
    }
    const uint64_t S = g->path_first[P];
    g->step_path.resize(S);
    g->step_handle.resize(S);
    g->step_pos.resize(S);
    next = 0;
    auto filler = [&]() {
        for (;;) {
            const uint64_t pi = next.fetch_add(1);
            if (pi >= P) break;
            uint64_t k = g->path_first[pi], pos = 0;
            for (uint32_t h : walks[pi]) {
                g->step_path[k] = (uint32_t)pi;
                g->step_handle[k] = h;
                g->step_pos[k] = pos;
                pos += g->node_len[h >> 1];
                ++k;
            }
            std::vector<uint32_t>().swap(walks[pi]);
        }
    };
    if (!pgsgd::run_threads((unsigned)nt, [&](unsigned) { filler(); })) {
        delete g;
        pgsgd::set_error("out of memory while generating the synthetic graph");
        return PGSGD_E_NOMEM;
    }
    *out = g;
    return PGSGD_OK;
}

extern "C" void pgsgd_graph_free(pgsgd_graph* g) { delete g; }

extern "C" int pgsgd_graph_get_view(const pgsgd_graph* g, pgsgd_graph_view* v) {
    if (!g || !v) return PGSGD_E_INVALID;
    v->n_nodes = g->n_nodes;
    v->n_steps = g->n_steps();
    v->n_paths = g->n_paths();
    v->node_len = g->node_len.data();
    v->path_first = g->path_first.data();
    v->step_path = g->step_path.size() ? g->step_path.data() : nullptr;   // (NULL: not built, or dropped — pgsgd_graph_drop_step_index)
    v->step_handle = g->step_handle.data();
    v->step_pos = g->step_pos.size() ? g->step_pos.data() : nullptr;
    return PGSGD_OK;
}

// The two arrays that follow from the others (step_path from path_first; step_pos from step_handle and node_len): freed, the views
// handed out from now on carry NULL for them.  The session then uploads 4 bytes per step instead of 12 and builds the positions on
// the device.
extern "C" int pgsgd_graph_drop_step_index(pgsgd_graph* g) {
    if (!g) return PGSGD_E_INVALID;
    pgsgd::raw_vector<uint32_t>().swap(g->step_path);
    pgsgd::raw_vector<uint64_t>().swap(g->step_pos);
    return PGSGD_OK;
}

namespace pgsgd {
static thread_local uint32_t t_load_flags = 0;
uint32_t load_flags() { return t_load_flags; }
}  // namespace pgsgd
extern "C" int pgsgd_graph_load_flags(const char* path, int n_threads, uint32_t flags, pgsgd_graph** out) {
    pgsgd::t_load_flags = flags;
    const int rc = pgsgd_graph_load(path, n_threads, out);
    pgsgd::t_load_flags = 0;
    if (rc == PGSGD_OK && (flags & PGSGD_LOAD_NO_STEP_INDEX)) (void)pgsgd_graph_drop_step_index(*out);   // (.og: the walk built them)
    return rc;
}

extern "C" uint64_t pgsgd_graph_edge_count(const pgsgd_graph* g) { return g ? g->edges.size() / 2 : 0; }
extern "C" const uint64_t* pgsgd_graph_edges(const pgsgd_graph* g) { return g ? g->edges.data() : nullptr; }
extern "C" const char* pgsgd_graph_path_name(const pgsgd_graph* g, uint64_t p) {
    return (g && p < g->path_names.size()) ? g->path_names[p].c_str() : nullptr;
}
extern "C" uint64_t pgsgd_graph_max_path_steps(const pgsgd_graph* g) {
    uint64_t m = 0;
    if (g)
        for (uint64_t p = 0; p + 1 < g->path_first.size(); ++p) m = std::max(m, g->path_first[p + 1] - g->path_first[p]);
    return m;
}
