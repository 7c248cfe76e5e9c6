// layout_main.cpp — the `odgi layout` subcommand on top of the MI355X kernels.
//
// Same flags, defaults, messages and outputs as the reference subcommand
// (src/subcommand/layout_main.cpp:18-466): flags :28-96, mandatory checks :113-125, "not optimized"
// exit :148-151, derived defaults :198-204,251-266, initial layout :268-330, SGD dispatch :333-387,
// component stacking :401-435, TSV/.lay output :438-463.  The SGD always runs on the GPU (`--gpu`
// is accepted for command-line compatibility); input is odgi's native `.og` or GFA v1, dispatched by
// file name like the reference (src/utils.cpp:110-134).  `-X FILE` (a serialized XP index,
// src/algorithms/xp.cpp:247-324) is refused with a message and exit code 1: the file is a dump of
// sdsl-lite structures (csa_wt, enc_vector, bit_vector + rank/select supports) whose on-disk layout
// is defined by a dependency that is absent from the reference tree, and the reference holds no XP
// file to pin a reader against; the index is lowered from the graph input instead, which is what
// the reference does when -X is not given (layout_main.cpp:220-227).
// Additions: `--seed N` (reproducible initial layout and sampler; upstream's flag is commented
// out, :76-80), `--gpu-streams N`, `--device N`, `--gpus N` (multi-GPU, pgsgd_multi.cpp), `--stress` (print layout
// quality to stderr).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "pgsgd_internal.hpp"

namespace {

struct Opt { char s; const char* l; bool has_value; const char* meta; const char* help; };

const Opt kOpts[] = {
    {'i', "idx", true, "FILE", "Load the succinct variation graph in ODGI format from this *FILE* (the default extension is .og); a file name ending in gfa is read as GFAv1; - reads .og from standard input."},
    {'o', "out", true, "FILE", "Write the layout coordinates to this FILE in .lay binary format."},
    {'T', "tsv", true, "FILE", "Write the layout in TSV format to this FILE."},
    {'X', "path-index", true, "FILE", "Load the path index from this FILE (not supported: the index is lowered from the graph input)."},
    {'C', "temp-dir", true, "PATH", "directory for temporary files (accepted; no temporary files are written)"},
    {'f', "path-sgd-use-paths", true, "FILE", "Line separated list of paths used to derive the default term count, Zipf space and eta_max."},
    {'N', "layout-initialization", true, "C", "Layout initialization mode: d) node rank in X, gaussian noise in Y (default); r) uniform noise in X and Y; u) node rank in X, uniform noise in Y; g) gaussian noise in X and Y; h) Hilbert curve."},
    {'G', "path-sgd-min-term-updates-paths", true, "N", "Minimum number of terms per iteration as a multiple of total path steps (default: 10)."},
    {'U', "path-sgd-min-term-updates-nodes", true, "N", "Minimum number of terms per iteration as a multiple of the number of nodes."},
    {'j', "path-sgd-delta", true, "N", "Threshold of the maximum displacement at which to stop (default: 0)."},
    {'g', "path-sgd-eta", true, "N", "The final learning rate (default: 0.01)."},
    {'v', "path-sgd-eta-max", true, "N", "The first and maximum learning rate (default: squared longest path step count)."},
    {'a', "path-sgd-zipf-theta", true, "N", "Theta of the Zipfian distribution that samples the second node of a term (default: 0.99)."},
    {'x', "path-sgd-iter-max", true, "N", "The maximum number of iterations (default: 30)."},
    {'K', "path-sgd-cooling", true, "N", "Use this fraction of the iterations for layout annealing (default: 0.5)."},
    {'F', "path-sgd-iteration-max-learning-rate", true, "N", "Iteration where the learning rate is max (parsed; as upstream, 0 is always used)."},
    {'k', "path-sgd-zipf-space", true, "N", "Maximum space size of the Zipfian distribution (default: max path step count)."},
    {'I', "path-sgd-zipf-space-max", true, "N", "Space size beyond which quantization occurs (default: 1000)."},
    {'l', "path-sgd-zipf-space-quantization-step", true, "N", "Quantization step beyond the maximum space size (default: 100)."},
    {'u', "path-sgd-snapshot", true, "STRING", "Prefix to which each snapshot layout of an iteration is written (default: NONE)."},
    {'t', "threads", true, "N", "Number of host threads for graph loading."},
    {0, "gpu", false, "", "Enable computation with GPU (always on in this build)."},
    {0, "seed", true, "N", "Seed for the initial layout and the sampler streams (default: random initial layout, sampler seed 9399220)."},
    {0, "gpu-streams", true, "N", "Number of concurrent sampler streams on the GPU (default: from graph size)."},
    {0, "gpus", true, "N", "Run on N GPUs of this node (devices --device .. --device+N-1): graph replicated, every iteration's terms split 1/N per GPU, coordinates merged with an RCCL all-reduce at every iteration (default: 1)."},
    {0, "device", true, "N", "HIP device ordinal (default: current)."},
    {0, "gpu-no-tiles", false, "", "Always use the per-lane kernel (one reference worker stream per GPU lane), never the tiled one."},
    {0, "gpu-shard-tiles", false, "", "Several GPUs (--gpus): share the tiles of every window among the GPUs instead of the windows. Faster on graphs below ~5M nodes, and not one GPU's layout (stress +8..21 %)."},
    {0, "gpu-sync-drain", false, "", "Tiled kernel: deliver every launch's far updates before the very next launch (default for schedules shorter than 30 iterations)."},
    {0, "gpu-terms-per-anchor", true, "N", "Partners drawn per sampled first step (default: 1 = the reference's term stream)."},
    {0, "gpu-exact-math", false, "", "Tile kernel: IEEE divisions and square root in a term's geometry instead of the hardware's reciprocal / reciprocal square root (1 ulp)."},
    {0, "gpu-no-partner-pairs", false, "", "Tile kernel: every lane keeps its own uniform partner (default: the lanes of a wavefront share them in quads of neighbouring steps)."},
    {0, "gpu-no-relabel", false, "", "Keep the graph's node ranks even if they do not follow the paths (default: such a graph is laid out under ranks by path position, which lets the tile kernel run)."},
    {0, "gpu-lock-window-ends", false, "", "Tile kernel: conflict resolution on shared node ends (measured: no effect on the layout)."},
    {0, "stress", false, "", "Print sampled path stress and the odgi-stats 2D path distance of the result to stderr."},
    {'P', "progress", false, "", "Write the current progress to stderr."},
    {'h', "help", false, "", "Print a help summary for odgi layout."},
};

void print_help(FILE* f) {
    fprintf(f, "  odgi layout {OPTIONS}\n\n    Establish 2D layouts of the graph using path-guided stochastic gradient\n"
               "    descent. The graph must be sorted and id-compacted.\n\n  OPTIONS:\n\n");
    for (const Opt& o : kOpts) {
        std::string flag = "      ";
        if (o.s) { flag += "-"; flag += o.s; if (o.has_value) flag += std::string("[") + o.meta + "]"; flag += ", "; }
        flag += std::string("--") + o.l;
        if (o.has_value) flag += std::string("=[") + o.meta + "]";
        fprintf(f, "%s\n          %s\n", flag.c_str(), o.help);
    }
    fflush(f);
}

struct Parsed {
    std::map<std::string, std::string> v;  // by long name
    bool has(const char* k) const { return v.count(k) != 0; }
    const std::string& get(const char* k) const { return v.at(k); }
};

// accepts -iFILE, -i FILE, -i=FILE, --idx FILE, --idx=FILE like args.hxx
int parse_cli(int argc, char** argv, Parsed& out, std::string& err) {
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const Opt* opt = nullptr;
        std::string value;
        bool have_value = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            for (const Opt& o : kOpts) if (name == o.l) opt = &o;
            if (!opt) { err = "Flag could not be matched: " + name; return 1; }
            if (eq != std::string::npos) { value = a.substr(eq + 1); have_value = true; }
        } else if (a.size() >= 2 && a[0] == '-' && a[1] != '-') {
            for (const Opt& o : kOpts) if (o.s && a[1] == o.s) opt = &o;
            if (!opt) { err = std::string("Flag could not be matched: '") + a[1] + "'"; return 1; }
            if (a.size() > 2) {
                if (!opt->has_value) { err = "Flag '" + a.substr(1, 1) + "' takes no value"; return 1; }
                value = a.substr(a[2] == '=' ? 3 : 2);
                have_value = true;
            }
        } else {
            err = "Passed in argument, but no positional arguments were ready to receive it: " + a;
            return 1;
        }
        if (opt->has_value && !have_value) {
            if (i + 1 >= argc) { err = std::string("Flag '") + opt->l + "' requires an argument but received none"; return 1; }
            value = argv[++i];
        }
        out.v[opt->l] = value;
    }
    return 0;
}

bool to_double(const std::string& s, double* d) {
    char* e = nullptr;
    *d = strtod(s.c_str(), &e);
    return e && *e == 0 && !s.empty();
}
bool to_u64(const std::string& s, uint64_t* u) {
    char* e = nullptr;
    if (s.empty() || s[0] == '-') return false;
    *u = strtoull(s.c_str(), &e, 10);
    return e && *e == 0;
}

}  // namespace

extern "C" int pgsgd_main_layout(int argc, char** argv) {
    // `odgi layout ...` or `layout ...`: drop the subcommand word like main_layout does (:21-26)
    std::vector<char*> av;
    av.push_back((char*)"odgi layout");
    int first = 1;
    if (argc > 1 && strcmp(argv[1], "layout") == 0) first = 2;
    for (int i = first; i < argc; ++i) av.push_back(argv[i]);
    Parsed a;
    std::string err;
    if (parse_cli((int)av.size(), av.data(), a, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        print_help(stderr);
        return 1;
    }
    if (a.has("help")) { print_help(stdout); return 0; }
    if (av.size() == 1) { print_help(stdout); return 1; }
    if (!a.has("idx")) {
        fprintf(stderr, "[odgi::layout] error: Please specify an input file from where to load the graph via -i=[FILE], --idx=[FILE].\n");
        return 1;
    }
    if (!a.has("out") && !a.has("tsv")) {
        fprintf(stderr, "[odgi::layout] error: Please specify an output file to where to store the layout via -o/--out=[FILE] or -T/--tsv=[FILE].\n");
        return 1;
    }
    auto bad = [&](const char* flag) {
        fprintf(stderr, "[odgi::layout] error: could not parse the value of --%s.\n", flag);
        return 1;
    };
    uint64_t num_threads = 1;
    if (a.has("threads") && !to_u64(a.get("threads"), &num_threads)) return bad("threads");
    if (num_threads == 0) num_threads = 1;
    if (a.has("path-index")) {
        fprintf(stderr, "[odgi::layout] error: -X/--path-index is not supported by this build (serialized XP files are sdsl-lite dumps; there is no "
                        "reference fixture to pin a reader against). Leave -X out: the path index is built from the graph input, as the reference does without -X.\n");
        return 1;
    }
    // utils.cpp:110-134: names ending in "gfa" are built from GFA, anything else (and "-" = stdin) is .og
    const std::string infile = a.get("idx");
    pgsgd_graph* graph = nullptr;
    // step_path / step_pos are not built on the host: the session builds the positions on the device from the handles (4 bytes per
    // step over PCIe instead of 12); `--stress` walks the paths for its own figures (pgsgd::StepIndex)
    int rc = pgsgd_graph_load_flags(infile.c_str(), (int)num_threads, PGSGD_LOAD_NO_STEP_INDEX, &graph);
    if (rc == PGSGD_E_NOTOPTIMIZED) {
        fprintf(stderr, "[odgi::layout] error: the graph is not optimized. Please run 'odgi sort' using -O, --optimize.\n");
        return 1;
    }
    if (rc) {
        fprintf(stderr, "[odgi::layout] error: %s: %s\n", pgsgd_strerror(rc), pgsgd_last_error());
        return 1;
    }
    pgsgd_graph_view view;
    pgsgd_graph_get_view(graph, &view);
    auto finish = [&](int code) {
        pgsgd_graph_free(graph);
        return code;
    };

    // parameters: layout_main.cpp:153-155,198-204
    pgsgd_params p;
    memset(&p, 0, sizeof p);
    p.eps = 0.01;
    if (a.has("path-sgd-eta") && !to_double(a.get("path-sgd-eta"), &p.eps)) return finish(bad("path-sgd-eta"));
    p.delta = 0;
    if (a.has("path-sgd-delta") && !to_double(a.get("path-sgd-delta"), &p.delta)) return finish(bad("path-sgd-delta"));
    if (a.has("path-sgd-min-term-updates-paths") && a.has("path-sgd-min-term-updates-nodes")) {
        fprintf(stderr, "[odgi::layout] error: there can only be one argument provided for the minimum number of term updates in the path guided 1D SGD."
                        "Please either use -G=[N], path-sgd-min-term-updates-paths=[N] or -U=[N], path-sgd-min-term-updates-nodes=[N].\n");
        return finish(1);
    }
    p.iter_max = 30;
    if (a.has("path-sgd-iter-max") && !to_u64(a.get("path-sgd-iter-max"), &p.iter_max)) return finish(bad("path-sgd-iter-max"));
    uint64_t ignored_F = 0;
    if (a.has("path-sgd-iteration-max-learning-rate") && !to_u64(a.get("path-sgd-iteration-max-learning-rate"), &ignored_F))
        return finish(bad("path-sgd-iteration-max-learning-rate"));
    p.iter_with_max_learning_rate = 0;  // :340,369 pass a literal 0
    p.theta = 0.99;
    if (a.has("path-sgd-zipf-theta") && !to_double(a.get("path-sgd-zipf-theta"), &p.theta)) return finish(bad("path-sgd-zipf-theta"));
    p.cooling_start = 0.5;
    if (a.has("path-sgd-cooling") && !to_double(a.get("path-sgd-cooling"), &p.cooling_start)) return finish(bad("path-sgd-cooling"));
    if (p.iter_max == 0 || !(p.theta < 1.0)) {
        fprintf(stderr, "[odgi::layout] error: need --path-sgd-iter-max >= 1 and --path-sgd-zipf-theta < 1.\n");
        return finish(1);
    }

    // paths that define the derived defaults (:229-250)
    std::vector<uint64_t> use_paths;
    if (a.has("path-sgd-use-paths")) {
        std::map<std::string, uint64_t> by_name;
        for (uint64_t i = 0; i < view.n_paths; ++i) by_name[pgsgd_graph_path_name(graph, i)] = i;
        std::ifstream in(a.get("path-sgd-use-paths"));
        std::string buf;
        while (std::getline(in, buf)) {
            auto it = by_name.find(buf);
            if (it != by_name.end()) {
                use_paths.push_back(it->second);
            } else {
                fprintf(stderr, "[odgi::layout] error: path '%s' as was given by -f=[FILE], --path-sgd-use-paths=[FILE]"
                                " is not present in the graph. Please remove this path from the file and restart 'odgi sort'.", buf.c_str());
            }
        }
    } else {
        for (uint64_t i = 0; i < view.n_paths; ++i) use_paths.push_back(i);
    }
    uint64_t sum_path_step_count = 0, max_path_step_count = 0;
    for (uint64_t pi : use_paths) {
        const uint64_t c = view.path_first[pi + 1] - view.path_first[pi];
        sum_path_step_count += c;
        max_path_step_count = std::max(max_path_step_count, c);
    }
    double G = 0, U = 0;
    if (a.has("path-sgd-min-term-updates-paths") && !to_double(a.get("path-sgd-min-term-updates-paths"), &G)) return finish(bad("path-sgd-min-term-updates-paths"));
    if (a.has("path-sgd-min-term-updates-nodes") && !to_double(a.get("path-sgd-min-term-updates-nodes"), &U)) return finish(bad("path-sgd-min-term-updates-nodes"));
    if (G != 0) p.min_term_updates = (uint64_t)(G * (double)sum_path_step_count);          // :252-253
    else if (U != 0) p.min_term_updates = (uint64_t)(U * (double)view.n_nodes);              // :255-256
    else p.min_term_updates = (uint64_t)(10.0 * (double)sum_path_step_count);               // :258
    uint64_t k = 0, I = 0, l = 0;
    double v = 0;
    if (a.has("path-sgd-zipf-space") && !to_u64(a.get("path-sgd-zipf-space"), &k)) return finish(bad("path-sgd-zipf-space"));
    if (a.has("path-sgd-eta-max") && !to_double(a.get("path-sgd-eta-max"), &v)) return finish(bad("path-sgd-eta-max"));
    if (a.has("path-sgd-zipf-space-max") && !to_u64(a.get("path-sgd-zipf-space-max"), &I)) return finish(bad("path-sgd-zipf-space-max"));
    if (a.has("path-sgd-zipf-space-quantization-step") && !to_u64(a.get("path-sgd-zipf-space-quantization-step"), &l)) return finish(bad("path-sgd-zipf-space-quantization-step"));
    p.space = k ? std::min(k, max_path_step_count) : max_path_step_count;                    // :262
    p.eta_max = v != 0 ? v : (double)max_path_step_count * (double)max_path_step_count;      // :263
    p.space_max = I ? std::min(p.space, I) : 1000;                                           // :265
    p.space_quantization_step = l ? std::max<uint64_t>(2, l) : 100;                          // :266
    p.snapshot = a.has("path-sgd-snapshot") ? 1 : 0;
    std::string snapshot_prefix = p.snapshot ? a.get("path-sgd-snapshot") : "";
    p.snapshot_prefix = p.snapshot ? snapshot_prefix.c_str() : nullptr;
    p.progress = a.has("progress") ? 1 : 0;
    uint64_t seed = 0;
    if (a.has("seed") && !to_u64(a.get("seed"), &seed)) return finish(bad("seed"));
    p.seed = a.has("seed") ? seed : PGSGD_DEFAULT_SEED;
    uint64_t streams = 0, device = 0;
    if (a.has("gpu-streams") && !to_u64(a.get("gpu-streams"), &streams)) return finish(bad("gpu-streams"));
    p.n_streams = (uint32_t)streams;
    uint64_t tpa = 1;
    if (a.has("gpu-terms-per-anchor") && !to_u64(a.get("gpu-terms-per-anchor"), &tpa)) return finish(bad("gpu-terms-per-anchor"));
    p.terms_per_anchor = (uint32_t)std::max<uint64_t>(1, tpa);
    if (a.has("gpu-no-tiles")) p.flags |= PGSGD_FLAG_NO_TILES;
    if (a.has("gpu-sync-drain")) p.flags |= PGSGD_FLAG_SYNC_DRAIN;
    if (a.has("gpu-shard-tiles")) p.flags |= PGSGD_FLAG_SHARD_TILES;
    if (a.has("gpu-exact-math")) p.flags |= PGSGD_FLAG_EXACT_MATH;
    if (a.has("gpu-no-partner-pairs")) p.flags |= PGSGD_FLAG_NO_PARTNER_PAIRS;
    if (a.has("gpu-no-relabel")) p.flags |= PGSGD_FLAG_NO_RELABEL;
    if (a.has("gpu-lock-window-ends")) p.flags |= PGSGD_FLAG_LOCK_WINDOW_ENDS;
    p.device = -1;
    if (a.has("device")) {
        if (!to_u64(a.get("device"), &device)) return finish(bad("device"));
        p.device = (int32_t)device;
    }
    uint64_t gpus = 1;
    if (a.has("gpus") && (!to_u64(a.get("gpus"), &gpus) || gpus == 0 || gpus > 64)) return finish(bad("gpus"));
    p.n_devices = (uint32_t)gpus;

    // initial layout (:268-330)
    char init_mode = 'd';
    if (a.has("layout-initialization")) {
        const std::string& m = a.get("layout-initialization");
        if (m.size() != 1) return finish(bad("layout-initialization"));
        init_mode = m[0];
    }
    const uint64_t n_ends = 2 * view.n_nodes;
    std::vector<double> X(n_ends), Y(n_ends);
    rc = pgsgd_init_layout(&view, init_mode, a.has("seed") ? (seed ? seed : 1) : 0, X.data(), Y.data());
    if (rc) { fprintf(stderr, "[odgi::layout] error: %s\n", pgsgd_last_error()); return finish(1); }

    // the SGD itself, on the GPU (:333-387)
    bool any_multi = max_path_step_count > 1;
    pgsgd_stats st;
    memset(&st, 0, sizeof st);
    if (any_multi && p.space >= 1) {
        rc = pgsgd_layout_run_f64(&view, &p, X.data(), Y.data(), &st);  // coordinates at the device's full resolution
        if (rc) {
            fprintf(stderr, "[odgi::layout] error: %s: %s\n", pgsgd_strerror(rc), pgsgd_last_error());
            return finish(1);
        }
        if (p.progress) {
            fprintf(stderr, "[odgi::layout] %llu term updates in %.1f ms of kernel time (%.3g terms/s) on %u GPU streams",
                    (unsigned long long)st.term_updates, st.kernel_ms, st.kernel_ms > 0 ? 1e3 * (double)st.term_updates / st.kernel_ms : 0.0, st.n_streams);
            if (st.apply_lanes) fprintf(stderr, " (sampling; %u lanes of one workgroup moved the node ends)", st.apply_lanes);
            fprintf(stderr, "\n");
        }
    }

    if (a.has("stress")) {
        double stress = 0, per_node = 0, per_bp = 0;
        pgsgd_path_stress(&view, X.data(), Y.data(), 2000000, 0x5eedull, &stress);
        pgsgd_path_distance(&view, X.data(), Y.data(), &per_node, &per_bp);
        fprintf(stderr, "[odgi::layout] sampled path stress: %.6g; sum path nodes distances (2D): %.6g per node, %.6g per bp\n", stress, per_node, per_bp);
    }

    // refine by weakly connected components (:401-435)
    std::vector<uint32_t> comp(view.n_nodes);
    const int64_t n_comp = pgsgd_weak_components(view.n_nodes, pgsgd_graph_edges(graph), pgsgd_graph_edge_count(graph), comp.data());
    if (n_comp < 0) { fprintf(stderr, "[odgi::layout] error: %s\n", pgsgd_last_error()); return finish(1); }
    pgsgd_pack_components(view.n_nodes, comp.data(), (uint64_t)n_comp, X.data(), Y.data());

    if (a.has("tsv") && !a.get("tsv").empty()) {
        rc = pgsgd_write_tsv(a.get("tsv").c_str(), view.n_nodes, comp.data(), (uint64_t)n_comp, X.data(), Y.data());
        if (rc) { fprintf(stderr, "[odgi::layout] error: %s\n", pgsgd_last_error()); return finish(1); }
    }
    if (a.has("out") && !a.get("out").empty()) {
        rc = pgsgd_write_lay(a.get("out").c_str(), n_ends, X.data(), Y.data());
        if (rc) { fprintf(stderr, "[odgi::layout] error: %s\n", pgsgd_last_error()); return finish(1); }
    }
    return finish(0);
}
