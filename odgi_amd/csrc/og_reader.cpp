// og_reader.cpp — odgi's native graph file (`.og`) -> the flat path-step index the SGD kernels read.
//
// `odgi layout -i graph.og` is the reference's primary input (utils.cpp:110-134: GFAz by magic, names
// ending in "gfa" as GFA, anything else as .og).  The file is what graph_t::serialize writes
// (src/odgi.cpp:1632-1685 members, src/node.cpp:422-435 per node):
//
//   u32 magic 1988148666, big-endian (libhandlegraph Serializable; odgi.cpp:1629-1631)
//   u64 max_node_id, min_node_id, node_count, edge_count, path_count, path_handle_next, id_increment
//   node_count x node record:
//       u64 sequence length, the sequence bytes, u64 node id (0 = deleted node)
//       3 packed vectors: edges, decoding, paths
//   path_count x path record: u64 step count, first step (2 x u64), last step (2 x u64),
//       u64 name length, the name bytes
//
// A packed vector (DYNAMIC's hacked_vector; the library is absent from /root/reference/deps, the layout
// below was read off the fixture test/DRB1-3123_sorted.og and consumes it to the last byte) is
//       u64 word count, the 64-bit words, u64 mask, u64 element count, u8 bits per element, u8 elements per word
// element i = (words[i / per_word] >> ((i % per_word) * bits)) & mask.
//
// Node records (src/node.hpp:26-84, node.cpp:57-117):
//   edges    pairs (other node id, type), type = other_rev | on_rev << 1 | to_curr << 2; an edge is stored
//            on both of its nodes (once for a self loop); the copy with to_curr = 0 is "this node, on_rev
//            -> other node, other_rev" (odgi.cpp:632-647), so those copies list every edge exactly once;
//   decoding deltas of neighbour node ids: 0 = this node, odd = id + (delta >> 1), even = id - (delta >> 1);
//   paths    6 values per step on this node: path id, type (is_rev | is_start << 1 | is_end << 2 | is_del << 3),
//            index into `decoding` of the previous step's node, rank of the previous step on that node,
//            the same two for the next step.
// A path is walked from its first step (node handle = 2 * rank + is_rev, step rank on the node) through
// the next pointers (odgi.cpp:393-423); node rank = id - id_increment - 1 (odgi.cpp:40-42).
// `odgi layout` needs an optimized graph (ids exactly 1..N, layout_main.cpp:148-151): anything else is
// PGSGD_E_NOTOPTIMIZED here.  Instead of exit(1) every failure is an error code.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <thread>

#include "pgsgd_internal.hpp"

namespace {

struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    uint64_t u64() {
        if (!ok || end - p < 8) { ok = false; return 0; }
        uint64_t v;
        memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    const uint8_t* skip(uint64_t n) {
        if (!ok || (uint64_t)(end - p) < n) { ok = false; return nullptr; }
        const uint8_t* q = p;
        p += n;
        return q;
    }
};

struct PackedVec {  // a view into the file buffer
    const uint8_t* words = nullptr;
    uint64_t n_words = 0, mask = 0, size = 0;
    uint32_t bits = 0, per_word = 0;
    bool read(Cursor& c) {
        n_words = c.u64();
        if (!c.ok || n_words > (uint64_t)(c.end - c.p) / 8) { c.ok = false; return false; }
        words = c.skip(n_words * 8);
        mask = c.u64();
        size = c.u64();
        const uint8_t* wb = c.skip(2);
        if (!c.ok) return false;
        bits = wb[0];
        per_word = wb[1];
        // every element must lie inside the words that are there
        if (size && (per_word == 0 || bits == 0 || bits > 64 || (uint64_t)per_word * bits > 64 || (size + per_word - 1) / per_word > n_words)) {
            c.ok = false;
            return false;
        }
        return true;
    }
    uint64_t at(uint64_t i) const {
        uint64_t w;
        memcpy(&w, words + 8 * (i / per_word), 8);
        return (w >> ((i % per_word) * bits)) & mask;
    }
};

struct NodeRec {
    PackedVec decoding, paths;
};

}  // namespace

extern "C" int pgsgd_graph_from_og(const char* path, int n_threads, pgsgd_graph** out) {
    using pgsgd::set_error;
    pgsgd::clear_error();
    if (!path || !out) return PGSGD_E_INVALID;
    *out = nullptr;
    std::string buf;
    if (strcmp(path, "-") == 0) {  // `-i -`: an .og on standard input (layout_main.cpp:113-138)
        char chunk[1 << 16];
        size_t n;
        while ((n = fread(chunk, 1, sizeof chunk, stdin)) > 0) buf.append(chunk, n);
    } else {
        std::ifstream in(path, std::ios::binary | std::ios::ate);
        if (!in) { set_error("cannot open '%s'", path); return PGSGD_E_IO; }
        const std::streamoff size = in.tellg();
        try {
            buf.resize((size_t)size);
        } catch (...) { return PGSGD_E_NOMEM; }
        in.seekg(0);
        if (size > 0 && !in.read(&buf[0], size)) { set_error("cannot read '%s'", path); return PGSGD_E_IO; }
    }

    Cursor c{(const uint8_t*)buf.data(), (const uint8_t*)buf.data() + buf.size()};
    const uint8_t* m = c.skip(4);
    if (!m || !(m[0] == 0x76 && m[1] == 0x80 && m[2] == 0xbd && m[3] == 0xba)) {
        set_error("'%s' is not a graph in ODGI format (magic number mismatch)", path);
        return PGSGD_E_FORMAT;
    }
    const uint64_t max_id = c.u64(), min_id = c.u64(), N = c.u64(), n_edges = c.u64(), P = c.u64();
    (void)c.u64();  // path_handle_next
    const uint64_t id_increment = c.u64();
    if (!c.ok) { set_error("'%s' is truncated (header)", path); return PGSGD_E_FORMAT; }
    if (N > (uint64_t)buf.size() / 16 || P > (uint64_t)buf.size() / 48) { set_error("'%s': implausible node or path count", path); return PGSGD_E_FORMAT; }
    if (N == 0) { set_error("'%s' has no nodes", path); return PGSGD_E_FORMAT; }
    if (N > (1ull << 31) - 1) { set_error("'%s': %llu nodes exceed the 2^31-1 the step index can address", path, (unsigned long long)N); return PGSGD_E_UNSUPPORTED; }
    if (!(min_id == 1 && max_id == N && id_increment == 0)) {  // graph_t::is_optimized, odgi.cpp:752-758
        set_error("the graph is not optimized: node ids span [%llu, %llu] for %llu nodes (id increment %llu)", (unsigned long long)min_id,
                  (unsigned long long)max_id, (unsigned long long)N, (unsigned long long)id_increment);
        return PGSGD_E_NOTOPTIMIZED;
    }

    auto g = new pgsgd_graph();
    auto fail = [&](int code) {
        delete g;
        return code;
    };
    g->n_nodes = N;
    std::vector<NodeRec> recs;
    try {
        g->node_len.assign(N, 0);
        g->edges.reserve(2 * std::min<uint64_t>(n_edges, buf.size()));
        recs.resize(N);
    } catch (...) { return fail(PGSGD_E_NOMEM); }
    for (uint64_t i = 0; i < N; ++i) {
        const uint64_t seq_len = c.u64();
        c.skip(seq_len);
        const uint64_t id = c.u64();
        PackedVec edges;
        if (!c.ok || !edges.read(c) || !recs[i].decoding.read(c) || !recs[i].paths.read(c)) {
            set_error("'%s' is truncated or corrupt in node record %llu", path, (unsigned long long)i);
            return fail(PGSGD_E_FORMAT);
        }
        if (id != i + 1) {  // a deleted node (id 0) or ids that do not follow the ranks
            set_error("the graph is not optimized: node record %llu holds id %llu", (unsigned long long)i, (unsigned long long)id);
            return fail(PGSGD_E_NOTOPTIMIZED);
        }
        if (seq_len > UINT32_MAX) { set_error("node %llu is longer than 2^32-1 bp", (unsigned long long)id); return fail(PGSGD_E_UNSUPPORTED); }
        g->node_len[i] = (uint32_t)seq_len;
        if (edges.size % 2 || recs[i].paths.size % 6) { set_error("'%s': malformed edge or step list on node %llu", path, (unsigned long long)id); return fail(PGSGD_E_FORMAT); }
        for (uint64_t e = 0; e < edges.size; e += 2) {
            const uint64_t other = edges.at(e), type = edges.at(e + 1);
            if (other < 1 || other > N) { set_error("'%s': node %llu has an edge to missing node %llu", path, (unsigned long long)id, (unsigned long long)other); return fail(PGSGD_E_FORMAT); }
            if (type & 4u) continue;  // to_curr: the other node holds the copy that is listed
            g->edges.push_back(2 * i + ((type >> 1) & 1u));
            g->edges.push_back(2 * (other - 1) + (type & 1u));
        }
    }
    struct PathMeta { uint64_t length, first_handle, first_rank; };
    std::vector<PathMeta> meta(P);
    g->path_names.resize(P);
    g->path_first.assign(P + 1, 0);
    for (uint64_t j = 0; j < P; ++j) {
        meta[j].length = c.u64();
        meta[j].first_handle = c.u64();
        meta[j].first_rank = c.u64();
        (void)c.u64();  // last step
        (void)c.u64();
        const uint64_t k = c.u64();
        const uint8_t* nm = c.skip(k);
        if (!c.ok) { set_error("'%s' is truncated in path record %llu", path, (unsigned long long)j); return fail(PGSGD_E_FORMAT); }
        g->path_names[j].assign((const char*)nm, (size_t)k);
        if (meta[j].length > (uint64_t)buf.size()) { set_error("'%s': implausible step count of path '%s'", path, g->path_names[j].c_str()); return fail(PGSGD_E_FORMAT); }
        g->path_first[j + 1] = g->path_first[j] + meta[j].length;
    }
    const uint64_t S = g->path_first[P];
    uint64_t step_records = 0;  // every step of every path has one 6-value record on its node
    for (uint64_t i = 0; i < N; ++i) step_records += recs[i].paths.size / 6;
    if (S > step_records) { set_error("'%s': the paths claim %llu steps, the nodes hold %llu", path, (unsigned long long)S, (unsigned long long)step_records); return fail(PGSGD_E_FORMAT); }
    try {
        g->step_path.resize(S);
        g->step_handle.resize(S);
        g->step_pos.resize(S);
    } catch (...) { return fail(PGSGD_E_NOMEM); }

    // walk the paths (independent of each other) through the next-step pointers
    std::atomic<uint64_t> next_path{0};
    std::atomic<int> err{0};
    std::vector<std::string> msgs(P);
    auto walker = [&]() {
        for (;;) {
            const uint64_t j = next_path.fetch_add(1);
            if (j >= P) break;
            uint64_t node = meta[j].first_handle >> 1, rank = meta[j].first_rank, pos = 0;
            uint64_t k = g->path_first[j];
            for (uint64_t s = 0; s < meta[j].length; ++s, ++k) {
                if (node >= N || 6 * rank + 5 >= recs[node].paths.size) {
                    msgs[j] = "path '" + g->path_names[j] + "' points outside the graph at step " + std::to_string(s);
                    err = PGSGD_E_FORMAT;
                    break;
                }
                const PackedVec& pv = recs[node].paths;
                const uint64_t type = pv.at(6 * rank + 1);
                if (pv.at(6 * rank) != j + 1 || (type & 8u)) {  // path handle j+1 (odgi.cpp:1713); deleted step
                    msgs[j] = "path '" + g->path_names[j] + "' is broken at step " + std::to_string(s);
                    err = PGSGD_E_FORMAT;
                    break;
                }
                g->step_path[k] = (uint32_t)j;
                g->step_handle[k] = (uint32_t)(2 * node + (type & 1u));
                g->step_pos[k] = pos;  // xp.cpp:607-617
                pos += g->node_len[node];
                const bool last = s + 1 == meta[j].length;
                if (((type >> 2) & 1u) != (last ? 1u : 0u)) {
                    msgs[j] = "path '" + g->path_names[j] + "' ends at step " + std::to_string(s) + " of " + std::to_string(meta[j].length);
                    err = PGSGD_E_FORMAT;
                    break;
                }
                if (last) break;
                const uint64_t di = pv.at(6 * rank + 4);
                if (di >= recs[node].decoding.size) {
                    msgs[j] = "path '" + g->path_names[j] + "' has a bad neighbour index at step " + std::to_string(s);
                    err = PGSGD_E_FORMAT;
                    break;
                }
                const uint64_t delta = recs[node].decoding.at(di), id = node + 1;
                const uint64_t next_id = delta == 0 ? id : (delta & 1u) ? id + (delta >> 1) : id - (delta >> 1);  // node.hpp:43-51
                rank = pv.at(6 * rank + 5);
                node = next_id - 1;  // next_id = 0 wraps and fails the range check above
            }
        }
    };
    {
        const int nt = std::max(1, std::min(n_threads, 64));
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(walker);
        walker();
        for (auto& t : th) t.join();
    }
    if (err) {
        for (const std::string& msg : msgs)
            if (!msg.empty()) { set_error("'%s': %s", path, msg.c_str()); break; }
        return fail(err);
    }
    *out = g;
    return PGSGD_OK;
}

// The reference's input dispatch (src/utils.cpp:110-134): names ending in "gfa" are GFA v1, everything
// else is read as .og.  GFAz (detected by magic upstream) is not supported.
extern "C" int pgsgd_graph_load(const char* path, int n_threads, pgsgd_graph** out) {
    if (!path || !out) return PGSGD_E_INVALID;
    const size_t n = strlen(path);
    // The reference also takes GFAz (utils.cpp:110-121: a magic word and a codec that both live in deps/GFAz, an empty
    // directory in the reference tree; no fixture either): refused by name with a message instead of being read as .og.
    if (n >= 5 && strcmp(path + n - 5, ".gfaz") == 0) {
        pgsgd::set_error("GFAz input is not supported (the codec is a dependency absent from the reference tree and there is no fixture to pin a "
                         "reader against): decompress to GFA or build an .og with `odgi build`");
        return PGSGD_E_UNSUPPORTED;
    }
    if (n >= 3 && strcmp(path + n - 3, "gfa") == 0) return pgsgd_graph_from_gfa(path, n_threads, out);
    return pgsgd_graph_from_og(path, n_threads, out);
}
