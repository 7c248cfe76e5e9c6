// lay_io.cpp — `.lay` writer/reader and TSV writer for `odgi layout` output.
//
// .lay (src/algorithms/layout.cpp:43-66): f64 min_value, then an sdsl::enc_vector<> (Elias-delta
// coder, sample density 128) over the IEEE-754 bit patterns of X[i]-min, Y[i]-min interleaved.
// sdsl-lite is not available, so the container is restated from its on-disk form, which the
// reference fixture test/DRB1-3123_unsorted.og.lay pins byte for byte (tests/test_host_logic.py):
//   u64 n | z: u64 bit_len, u8 width(=1), ceil(bit_len/64) u64 | samples: u64 bit_len, u8 width w,
//   ceil(bit_len/64) u64.   samples holds 2*ceil(n/128)+2 w-bit ints: (value at 128k, bit offset of
//   block k in z) pairs and a final (0, z_bits+1).  Non-sample elements are Elias-delta codes of
//   (v[i]-v[i-1]) mod 2^64, LSB first; a delta of 0 is coded as 2^64 (length 65).
// TSV (layout.cpp:10-35): header idx/X/Y/component, two rows per node, 16 significant digits.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <charconv>
#include <cstring>
#include <thread>
#include <fstream>
#include <limits>

#include "pgsgd_internal.hpp"

namespace {

inline int hi_bit(uint64_t x) { return 63 - __builtin_clzll(x); }  // x != 0

struct BitWriter {
    std::vector<uint64_t> w;
    uint64_t nbits = 0;
    void reserve_bits(uint64_t n) { w.assign((size_t)((n + 63) / 64), 0); }
    void put(uint64_t v, int len) {  // low `len` bits of v, LSB first; len in [1,64]
        if (len < 64) v &= (1ull << len) - 1;
        const uint64_t word = nbits >> 6;
        const int off = (int)(nbits & 63);
        w[(size_t)word] |= v << off;
        if (off + len > 64) w[(size_t)word + 1] |= v >> (64 - off);
        nbits += (uint64_t)len;
    }
};

inline uint64_t delta_len(uint64_t x) {  // elias_delta::encoding_length
    const int len_1 = x ? hi_bit(x) : 64;
    return (uint64_t)len_1 + ((uint64_t)hi_bit((uint64_t)len_1 + 1) << 1) + 1;
}

inline void delta_put(BitWriter& bw, uint64_t x) {
    int len, lenlen;
    if (!x) {
        len = 65;
        lenlen = 7;
    } else {
        len = hi_bit(x) + 1;
        lenlen = hi_bit((uint64_t)len) + 1;
    }
    bw.put(1ull << (lenlen - 1), lenlen);  // lenlen-1 zeros, then a one
    if (lenlen > 1) {
        bw.put((uint64_t)len, lenlen - 1);  // len without its leading one
        if (len > 1) bw.put(x, len - 1);    // x without its leading one
    }
}

struct PackedInts {  // sdsl int_vector<0> with a run-time width
    std::vector<uint64_t> w;
    int width = 1;
    uint64_t count = 0;
    void init(uint64_t n, int wd) {
        width = wd;
        count = n;
        w.assign((size_t)((n * (uint64_t)wd + 63) / 64), 0);
    }
    void set(uint64_t i, uint64_t v) {
        const uint64_t bit = i * (uint64_t)width;
        const uint64_t word = bit >> 6;
        const int off = (int)(bit & 63);
        if (width < 64) v &= (1ull << width) - 1;
        w[(size_t)word] |= v << off;
        if (off + width > 64) w[(size_t)word + 1] |= v >> (64 - off);
    }
    uint64_t get(uint64_t i) const {
        const uint64_t bit = i * (uint64_t)width;
        const uint64_t word = bit >> 6;
        const int off = (int)(bit & 63);
        uint64_t v = w[(size_t)word] >> off;
        if (off + width > 64) v |= w[(size_t)word + 1] << (64 - off);
        if (width < 64) v &= (1ull << width) - 1;
        return v;
    }
};

void append(std::vector<uint8_t>& out, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    out.insert(out.end(), b, b + n);
}

int encode_lay(uint64_t n_ends, const double* X, const double* Y, std::vector<uint8_t>& out) {
    double min_value = std::numeric_limits<double>::max();  // layout.hpp:27
    for (uint64_t i = 0; i < n_ends; ++i) min_value = std::min(X[i], min_value);
    for (uint64_t i = 0; i < n_ends; ++i) min_value = std::min(Y[i], min_value);
    const uint64_t n = 2 * n_ends;
    std::vector<uint64_t> vals((size_t)n);
    for (uint64_t i = 0; i < n_ends; ++i) {
        const double x = X[i] - min_value, y = Y[i] - min_value;
        memcpy(&vals[2 * i], &x, 8);
        memcpy(&vals[2 * i + 1], &y, 8);
    }
    // enc_vector: every dens-th value is a sample (stored with the bit offset of what follows it), the others are
    // Elias-delta codes of differences.  The blocks between samples are independent once their bit offsets are known:
    // pass 1 sizes them (threads), a prefix sum places them, pass 2 encodes ranges of blocks on threads into private
    // word arrays that start at a word boundary of the final bit string and are OR-ed into it afterwards (neighbouring
    // ranges share a word).
    const uint64_t dens = 128;
    const uint64_t samples = (n + dens - 1) / dens;
    uint64_t z_size = 0, max_sample = 0;
    std::vector<uint64_t> block_bits((size_t)samples + 1, 0);
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({32, std::thread::hardware_concurrency(), samples / 64 + 1}));
    bool threads_ok = true;
    auto on_threads = [&](auto&& body) {  // body(thread index): contiguous ranges of blocks
        if (!pgsgd::run_threads(nt, body)) threads_ok = false;
    };
    auto range = [&](unsigned t) { return std::make_pair(samples * t / nt, samples * (t + 1) / nt); };
    on_threads([&](unsigned t) {
        const auto [b0, b1] = range(t);
        for (uint64_t b = b0; b < b1; ++b) {
            uint64_t bits = 0;
            const uint64_t e = std::min(n, (b + 1) * dens);
            for (uint64_t i = b * dens + 1; i < e; ++i) bits += delta_len(vals[i] - vals[i - 1]);
            block_bits[(size_t)b + 1] = bits;
        }
    });
    if (!threads_ok) { pgsgd::set_error("out of memory while encoding the layout"); return PGSGD_E_NOMEM; }
    for (uint64_t b = 0; b < samples; ++b) {
        if (max_sample < vals[b * dens]) max_sample = vals[b * dens];
        block_bits[(size_t)b + 1] += block_bits[(size_t)b];  // -> bit offset of block b + 1
    }
    z_size = block_bits[(size_t)samples];
    PackedInts sv;
    BitWriter z;
    uint64_t z_bits_total = z_size;
    if (n) {
        const int width = (max_sample > z_size + 1) ? hi_bit(max_sample) + 1 : hi_bit(z_size + 1) + 1;
        sv.init(2 * samples + 2, width);
        z.reserve_bits(z_size);
        uint64_t si = 0;
        for (uint64_t b = 0; b < samples; ++b) {
            sv.set(si++, vals[b * dens]);
            sv.set(si++, block_bits[(size_t)b]);
        }
        sv.set(si++, 0);
        sv.set(si++, z_bits_total + 1);
        std::vector<BitWriter> part(nt);
        on_threads([&](unsigned t) {
            const auto [b0, b1] = range(t);
            if (b0 == b1) return;
            BitWriter& w = part[t];
            const uint64_t first_bit = block_bits[(size_t)b0], last_bit = block_bits[(size_t)b1];
            w.reserve_bits((first_bit & 63) + (last_bit - first_bit));
            w.nbits = first_bit & 63;  // bit 0 of the private array is a word boundary of the final string
            for (uint64_t b = b0; b < b1; ++b) {
                const uint64_t e = std::min(n, (b + 1) * dens);
                for (uint64_t i = b * dens + 1; i < e; ++i) delta_put(w, vals[i] - vals[i - 1]);
            }
        });
        if (!threads_ok) { pgsgd::set_error("out of memory while encoding the layout"); return PGSGD_E_NOMEM; }
        for (unsigned t = 0; t < nt; ++t) {
            const auto [b0, b1] = range(t);
            if (b0 == b1) continue;
            const size_t w0 = (size_t)(block_bits[(size_t)b0] >> 6);
            for (size_t k = 0; k < part[t].w.size(); ++k) z.w[w0 + k] |= part[t].w[k];
        }
        z.nbits = z_size;
    } else {
        sv.init(0, 64);  // empty int_vector<0> keeps its default width of 64
    }
    out.clear();
    append(out, &min_value, 8);
    append(out, &n, 8);
    const uint64_t zb = n ? z_bits_total : 0;
    const uint8_t zw = n ? 1 : 64;
    append(out, &zb, 8);
    append(out, &zw, 1);
    append(out, z.w.data(), z.w.size() * 8);
    const uint64_t sb = sv.count * (uint64_t)sv.width;
    const uint8_t sw = (uint8_t)sv.width;
    append(out, &sb, 8);
    append(out, &sw, 1);
    append(out, sv.w.data(), sv.w.size() * 8);
    return PGSGD_OK;
}

struct BitReader {
    const uint64_t* w;
    uint64_t nbits, pos;
    bool bit() {
        const bool b = (w[pos >> 6] >> (pos & 63)) & 1;
        ++pos;
        return b;
    }
    uint64_t get(int len) {  // len in [0,64]
        if (len == 0) return 0;
        const uint64_t word = pos >> 6;
        const int off = (int)(pos & 63);
        uint64_t v = w[word] >> off;
        if (off + len > 64) v |= w[word + 1] << (64 - off);
        if (len < 64) v &= (1ull << len) - 1;
        pos += (uint64_t)len;
        return v;
    }
};

}  // namespace

extern "C" int pgsgd_lay_buffer(uint64_t n_ends, const double* X, const double* Y, uint8_t** buf, size_t* len) {
    pgsgd::clear_error();
    if (!buf || !len || (n_ends && (!X || !Y))) return PGSGD_E_INVALID;
    std::vector<uint8_t> out;
    int rc = encode_lay(n_ends, X, Y, out);
    if (rc) return rc;
    *buf = (uint8_t*)malloc(out.size() ? out.size() : 1);
    if (!*buf) { pgsgd::set_error("out of memory while encoding the layout"); return PGSGD_E_NOMEM; }
    memcpy(*buf, out.data(), out.size());
    *len = out.size();
    return PGSGD_OK;
}

extern "C" int pgsgd_write_lay(const char* path, uint64_t n_ends, const double* X, const double* Y) {
    pgsgd::clear_error();
    if (!path) return PGSGD_E_INVALID;
    std::vector<uint8_t> out;
    int rc = encode_lay(n_ends, X, Y, out);
    if (rc) return rc;
    FILE* f = strcmp(path, "-") == 0 ? stdout : fopen(path, "wb");
    if (!f) { pgsgd::set_error("cannot write '%s'", path); return PGSGD_E_IO; }
    const size_t wr = fwrite(out.data(), 1, out.size(), f);
    if (f != stdout) fclose(f); else fflush(f);
    if (wr != out.size()) { pgsgd::set_error("short write to '%s'", path); return PGSGD_E_IO; }
    return PGSGD_OK;
}

extern "C" int pgsgd_read_lay(const char* path, uint64_t* n_ends, double** Xo, double** Yo) {
    using pgsgd::set_error;
    pgsgd::clear_error();
    if (!path || !n_ends || !Xo || !Yo) return PGSGD_E_INVALID;
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) { set_error("cannot open '%s'", path); return PGSGD_E_IO; }
    const size_t size = (size_t)in.tellg();
    std::vector<uint8_t> b(size);
    in.seekg(0);
    if (size && !in.read((char*)b.data(), (std::streamsize)size)) return PGSGD_E_IO;
    size_t p = 0;
    auto need = [&](size_t k) { return p + k <= size; };
    if (!need(8 + 8 + 9)) { set_error("'%s' is too short for a .lay", path); return PGSGD_E_FORMAT; }
    double min_value;
    uint64_t n;
    memcpy(&min_value, &b[p], 8); p += 8;
    memcpy(&n, &b[p], 8); p += 8;
    uint64_t zbits; uint8_t zw;
    memcpy(&zbits, &b[p], 8); p += 8;
    zw = b[p++];
    if (zbits > 8 * (uint64_t)size) { set_error("truncated .lay (deltas)"); return PGSGD_E_FORMAT; }
    const size_t zwords = (size_t)((zbits + 63) / 64);
    if (!need(zwords * 8 + 9)) { set_error("truncated .lay (deltas)"); return PGSGD_E_FORMAT; }
    std::vector<uint64_t> z(zwords + 1, 0);
    memcpy(z.data(), &b[p], zwords * 8); p += zwords * 8;
    uint64_t sbits; uint8_t sw;
    memcpy(&sbits, &b[p], 8); p += 8;
    sw = b[p++];
    if (sbits > 8 * (uint64_t)size) { set_error("truncated .lay (samples)"); return PGSGD_E_FORMAT; }
    const size_t swords = (size_t)((sbits + 63) / 64);
    if (!need(swords * 8) || sw == 0 || sw > 64 || (n && zw != 1)) { set_error("truncated or malformed .lay (samples)"); return PGSGD_E_FORMAT; }
    PackedInts sv;
    sv.width = sw;
    sv.count = sbits / sw;
    sv.w.assign(swords + 1, 0);
    memcpy(sv.w.data(), &b[p], swords * 8); p += swords * 8;
    if (n % 2) { set_error(".lay holds an odd number of values"); return PGSGD_E_FORMAT; }
    if (n > zbits + sv.count) { set_error(".lay claims more values than its streams can hold"); return PGSGD_E_FORMAT; }  // >= 1 bit each
    const uint64_t blocks = (n + 127) / 128;
    if (n && sv.count < 2 * blocks + 2) { set_error(".lay sample table too small"); return PGSGD_E_FORMAT; }
    const uint64_t ends = n / 2;
    double* X = (double*)malloc((ends ? ends : 1) * 8);
    double* Y = (double*)malloc((ends ? ends : 1) * 8);
    if (!X || !Y) { free(X); free(Y); pgsgd::set_error("out of memory while decoding the layout"); return PGSGD_E_NOMEM; }
    BitReader br{z.data(), zbits, 0};
    uint64_t v = 0;
    auto corrupt = [&](const char* what) {
        free(X);
        free(Y);
        set_error(".lay %s", what);
        return PGSGD_E_FORMAT;
    };
    for (uint64_t i = 0; i < n; ++i) {
        if (i % 128 == 0) {
            v = sv.get(2 * (i / 128));
            br.pos = sv.get(2 * (i / 128) + 1);
            if (br.pos > zbits) return corrupt("block offset beyond the delta stream");
        } else {
            // Elias delta: L zeros, a one, L bits of the length, length-1 bits of the value; L <= 6, length <= 65
            int L = 0;
            for (;;) {
                if (br.pos >= zbits || L > 6) return corrupt("delta stream overrun");
                if (br.bit()) break;
                ++L;
            }
            uint64_t x;
            if (L == 0) {
                x = 1;
            } else {
                if (br.pos + (uint64_t)L > zbits) return corrupt("delta stream overrun");
                const uint64_t len = br.get(L) | (1ull << L);
                if (len > 65 || br.pos + (len - 1) > zbits) return corrupt("delta stream overrun");
                const uint64_t u = br.get((int)len - 1);
                x = len - 1 >= 64 ? u : (u | (1ull << (len - 1)));  // len 65 encodes 2^64 == 0
            }
            v += x;
        }
        double d;
        memcpy(&d, &v, 8);
        d += min_value;  // layout.cpp:86-96
        if (i & 1) Y[i / 2] = d; else X[i / 2] = d;
    }
    *n_ends = ends;
    *Xo = X;
    *Yo = Y;
    return PGSGD_OK;
}

extern "C" void pgsgd_free(void* p) { free(p); }

// ---- weakly connected components + vertical stacking + TSV -----------------------------------

// weakly_connected_components.cpp:8-68: components are discovered in node-rank order, so their ids
// follow each component's lowest rank; orientation of the edges is irrelevant.
extern "C" int64_t pgsgd_weak_components(uint64_t n_nodes, const uint64_t* edges, uint64_t n_edges, uint32_t* comp) {
    pgsgd::clear_error();
    if (!comp || (n_edges && !edges)) return PGSGD_E_INVALID;
    std::vector<uint32_t> parent((size_t)n_nodes);
    for (uint64_t i = 0; i < n_nodes; ++i) parent[i] = (uint32_t)i;
    auto find = [&](uint32_t x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    };
    for (uint64_t e = 0; e < n_edges; ++e) {
        const uint64_t a = edges[2 * e] >> 1, b = edges[2 * e + 1] >> 1;
        if (a >= n_nodes || b >= n_nodes) { pgsgd::set_error("edge %llu out of range", (unsigned long long)e); return PGSGD_E_INVALID; }
        uint32_t ra = find((uint32_t)a), rb = find((uint32_t)b);
        if (ra != rb) {
            if (ra < rb) parent[rb] = ra; else parent[ra] = rb;  // root = lowest rank
        }
    }
    std::vector<uint32_t> id_of_root((size_t)n_nodes, UINT32_MAX);
    uint32_t next = 0;
    for (uint64_t i = 0; i < n_nodes; ++i) {
        const uint32_t r = find((uint32_t)i);
        if (id_of_root[r] == UINT32_MAX) id_of_root[r] = next++;  // r == lowest rank, reached first
        comp[i] = id_of_root[r];
    }
    return (int64_t)next;
}

// layout_main.cpp:401-435 with coord_range_2d_t of draw.hpp:35-55, including its max_* start value
// of numeric_limits<double>::min() (the smallest positive double).
extern "C" int pgsgd_pack_components(uint64_t n_nodes, const uint32_t* comp, uint64_t n_comp, double* X, double* Y) {
    pgsgd::clear_error();
    if (!comp || !X || !Y) return PGSGD_E_INVALID;
    struct Range {
        double min_x = std::numeric_limits<double>::max(), max_x = std::numeric_limits<double>::min();
        double min_y = std::numeric_limits<double>::max(), max_y = std::numeric_limits<double>::min();
        double x_offset = 0, y_offset = 0;
    };
    std::vector<Range> ranges((size_t)n_comp);
    for (uint64_t i = 0; i < n_nodes; ++i) {
        if (comp[i] >= n_comp) return PGSGD_E_INVALID;
        Range& r = ranges[comp[i]];
        for (uint64_t j = 2 * i; j <= 2 * i + 1; ++j) {
            if (X[j] < r.min_x) r.min_x = X[j];
            if (X[j] > r.max_x) r.max_x = X[j];
            if (Y[j] < r.min_y) r.min_y = Y[j];
            if (Y[j] > r.max_y) r.max_y = Y[j];
        }
    }
    const double border = 1000.0;
    double curr_y_offset = border;
    for (Range& r : ranges) {
        r.x_offset = r.min_x - border;
        r.y_offset = curr_y_offset - r.min_y;
        curr_y_offset += (r.max_y - r.min_y) + border;
    }
    for (uint64_t i = 0; i < n_nodes; ++i) {
        const Range& r = ranges[comp[i]];
        for (uint64_t j = 2 * i; j <= 2 * i + 1; ++j) {
            X[j] -= r.x_offset;
            Y[j] += r.y_offset;
        }
    }
    return PGSGD_OK;
}

extern "C" int pgsgd_write_tsv(const char* path, uint64_t n_nodes, const uint32_t* comp, uint64_t n_comp,
                               const double* X, const double* Y) {
    pgsgd::clear_error();
    if (!path || !comp || !X || !Y) return PGSGD_E_INVALID;
    FILE* f = strcmp(path, "-") == 0 ? stdout : fopen(path, "w");
    if (!f) { pgsgd::set_error("cannot write '%s'", path); return PGSGD_E_IO; }
    fprintf(f, "idx\tX\tY\tcomponent\n");
    // rows grouped by component, nodes of a component in rank order (handles sorted by integer)
    std::vector<uint64_t> start((size_t)n_comp + 1, 0);
    for (uint64_t i = 0; i < n_nodes; ++i) start[comp[i] + 1]++;
    for (uint64_t c = 0; c < n_comp; ++c) start[c + 1] += start[c];
    std::vector<uint32_t> order((size_t)n_nodes);
    {
        std::vector<uint64_t> cur(start.begin(), start.end() - 1);
        for (uint64_t i = 0; i < n_nodes; ++i) order[(size_t)cur[comp[i]]++] = (uint32_t)i;
    }
    // rows are formatted on host threads into per-chunk buffers (std::to_chars, general format, precision 16 =
    // what `out << std::setprecision(16)` prints upstream, layout.cpp:23), then written in order
    std::vector<uint32_t> comp_of_row((size_t)n_nodes);
    for (uint64_t c = 0; c < n_comp; ++c)
        for (uint64_t k = start[c]; k < start[c + 1]; ++k) comp_of_row[(size_t)k] = (uint32_t)c;
    const uint64_t chunk = 1 << 16;
    const uint64_t n_chunks = (n_nodes + chunk - 1) / chunk;
    std::vector<std::string> out((size_t)n_chunks);
    std::atomic<uint64_t> next{0};
    auto format = [&]() {
        char num[64];
        for (uint64_t ci = next.fetch_add(1); ci < n_chunks; ci = next.fetch_add(1)) {
            std::string& o = out[(size_t)ci];
            o.reserve(chunk * 2 * 48);
            const uint64_t e = std::min(n_nodes, (ci + 1) * chunk);
            for (uint64_t k = ci * chunk; k < e; ++k) {
                const uint64_t pos = 2 * (uint64_t)order[(size_t)k];
                for (uint64_t end = pos; end < pos + 2; ++end) {
                    auto put_u = [&](uint64_t v) { o.append(num, std::to_chars(num, num + sizeof num, v).ptr); };
                    auto put_d = [&](double v) { o.append(num, std::to_chars(num, num + sizeof num, v, std::chars_format::general, 16).ptr); };
                    put_u(end); o.push_back('\t');
                    put_d(X[end]); o.push_back('\t');
                    put_d(Y[end]); o.push_back('\t');
                    put_u(comp_of_row[(size_t)k]); o.push_back('\n');
                }
            }
        }
    };
    {
        const unsigned nt = n_chunks > 1 ? std::max(1u, std::min<unsigned>({64u, std::thread::hardware_concurrency(), (unsigned)n_chunks})) : 1u;
        if (!pgsgd::run_threads(nt, [&](unsigned) { format(); })) {
            pgsgd::set_error("out of memory while formatting the TSV");
            if (f != stdout) fclose(f);
            return PGSGD_E_NOMEM;
        }
    }
    for (const std::string& o : out)
        if (fwrite(o.data(), 1, o.size(), f) != o.size()) {
            pgsgd::set_error("short write to '%s'", path);
            if (f != stdout) fclose(f);
            return PGSGD_E_IO;
        }
    if (f != stdout) fclose(f); else fflush(f);
    return PGSGD_OK;
}
