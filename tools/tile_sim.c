/* tile_sim.c — CPU simulation of a tile-scheduled PG-SGD (experiment behind DESIGN.md's "what
 * would move the roofline further").  Not part of the product or the oracle; it borrows the
 * oracle's sampler by including its source.
 *
 * Schedule: steps of each path are cut into tiles of T steps; per iteration every tile gets
 * terms_per_iter * T_tile / S anchors drawn uniformly inside the tile (stratified), partners by
 * the reference rule.  A tile works on a PRIVATE copy of the coordinates of the nodes in its rank
 * window [rmin, rmin+W); partners outside the tile's steps / window are read from and atomically
 * added to the global arrays.  C tiles form a concurrent window: they all stage from the global
 * state at window start and add their private deltas back at window end (maximal staleness).
 * usage: tile_sim graph.bin T W C visits  -> prints stress
 */
#include "../oracle/pgsgd_oracle.c"
#include <stdio.h>

typedef struct { uint64_t t0, n; uint32_t path; uint64_t rmin; } tile_t;

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s graph.bin T W C visits\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    uint64_t hdr[3];
    if (!f || fread(hdr, 8, 3, f) != 3) return 1;
    orc_graph g; g.n_nodes = hdr[0]; g.n_steps = hdr[1]; g.n_paths = hdr[2];
    uint32_t* node_len = malloc(4 * g.n_nodes); uint64_t* path_first = malloc(8 * (g.n_paths + 1));
    uint32_t* step_path = malloc(4 * g.n_steps); uint32_t* step_handle = malloc(4 * g.n_steps); uint64_t* step_pos = malloc(8 * g.n_steps);
    double* X = malloc(16 * g.n_nodes); double* Y = malloc(16 * g.n_nodes);
    if (fread(node_len, 4, g.n_nodes, f) != g.n_nodes || fread(path_first, 8, g.n_paths + 1, f) != g.n_paths + 1 ||
        fread(step_path, 4, g.n_steps, f) != g.n_steps || fread(step_handle, 4, g.n_steps, f) != g.n_steps ||
        fread(step_pos, 8, g.n_steps, f) != g.n_steps || fread(X, 8, 2 * g.n_nodes, f) != 2 * g.n_nodes ||
        fread(Y, 8, 2 * g.n_nodes, f) != 2 * g.n_nodes) return 1;
    fclose(f);
    g.node_len = node_len; g.path_first = path_first; g.step_path = step_path; g.step_handle = step_handle; g.step_pos = step_pos;
    const uint64_t T = strtoull(argv[2], 0, 10), W = strtoull(argv[3], 0, 10), C = strtoull(argv[4], 0, 10), visits = strtoull(argv[5], 0, 10);
    uint64_t max_steps = 0;
    for (uint64_t p = 0; p < g.n_paths; ++p) if (path_first[p + 1] - path_first[p] > max_steps) max_steps = path_first[p + 1] - path_first[p];
    orc_params p; memset(&p, 0, sizeof p);
    p.iter_max = 30; p.min_term_updates = 10 * g.n_steps; p.eps = 0.01; p.eta_max = (double)max_steps * (double)max_steps; p.theta = 0.99;
    p.space = max_steps; p.space_max = 1000; p.space_quantization_step = 100; p.cooling_start = 0.5;
    double* zetas = malloc(8 * orc_zeta_size(p.space, p.space_max, p.space_quantization_step));
    orc_zetas(p.theta, p.space, p.space_max, p.space_quantization_step, zetas);
    double* etas = malloc(8 * (p.iter_max + 1)); orc_schedule(&p, etas);
    /* tiles, path-major */
    uint64_t n_tiles = 0;
    for (uint64_t pi = 0; pi < g.n_paths; ++pi) { uint64_t c = path_first[pi + 1] - path_first[pi]; if (c > 1) n_tiles += (c + T - 1) / T; }
    tile_t* tiles = malloc(sizeof(tile_t) * n_tiles); uint64_t ti = 0;
    for (uint64_t pi = 0; pi < g.n_paths; ++pi) {
        const uint64_t b = path_first[pi], c = path_first[pi + 1] - b;
        if (c <= 1) continue;
        for (uint64_t t0 = 0; t0 < c; t0 += T) {
            tile_t t; t.t0 = b + t0; t.n = c - t0 < T ? c - t0 : T; t.path = (uint32_t)pi; t.rmin = UINT64_MAX;
            for (uint64_t k = t.t0; k < t.t0 + t.n; ++k) if ((step_handle[k] >> 1) < t.rmin) t.rmin = step_handle[k] >> 1;
            tiles[ti++] = t;
        }
    }
    double* PX = malloc(16 * W * C); double* PY = malloc(16 * W * C);   /* private copies, 2W ends per tile */
    double* GX = malloc(16 * g.n_nodes); double* GY = malloc(16 * g.n_nodes);
    uint64_t s[4]; orc_rng_seed(9399220, s);
    uint64_t local_a = 0, local_b = 0, total = 0;
    for (uint64_t iter = 0; iter < p.iter_max; ++iter) {
        const double eta = etas[iter]; const int cooling = iter >= 15;
        for (uint64_t v = 0; v < visits; ++v)
        for (uint64_t w0 = 0; w0 < n_tiles; w0 += C) {
            const uint64_t wc = n_tiles - w0 < C ? n_tiles - w0 : C;
            memcpy(GX, X, 16 * g.n_nodes); memcpy(GY, Y, 16 * g.n_nodes);
            for (uint64_t j = 0; j < wc; ++j) {
                const tile_t* t = &tiles[w0 + j];
                double* px = PX + 2 * W * j; double* py = PY + 2 * W * j;
                for (uint64_t e = 0; e < 2 * W; ++e) { const uint64_t ge = 2 * t->rmin + e; px[e] = ge < 2 * g.n_nodes ? GX[ge] : 0; py[e] = ge < 2 * g.n_nodes ? GY[ge] : 0; }
                const uint64_t n_terms = (uint64_t)((double)p.min_term_updates * (double)t->n / (double)g.n_steps / (double)visits + 0.5);
                for (uint64_t q = 0; q < n_terms; ++q) {
                    orc_anchor an; an.pstart = path_first[t->path]; an.cnt = path_first[t->path + 1] - an.pstart;
                    an.k = t->t0 + orc_uniform_u64(s, t->n); an.s_rank = an.k - an.pstart;
                    orc_term tm; orc_sample_partner(&g, &p, zetas, cooling, &an, s, &tm);
                    const uint64_t i = 2 * (uint64_t)(step_handle[tm.ka] >> 1) + tm.off_a, jj = 2 * (uint64_t)(step_handle[tm.kb] >> 1) + tm.off_b;
                    const int la = i >= 2 * t->rmin && i < 2 * (t->rmin + W);
                    const int lb = tm.kb >= t->t0 && tm.kb < t->t0 + t->n && jj >= 2 * t->rmin && jj < 2 * (t->rmin + W);
                    double* xa = la ? &px[i - 2 * t->rmin] : &X[i]; double* ya = la ? &py[i - 2 * t->rmin] : &Y[i];
                    double* xb = lb ? &px[jj - 2 * t->rmin] : &X[jj]; double* yb = lb ? &py[jj - 2 * t->rmin] : &Y[jj];
                    local_a += la; local_b += lb; total++;
                    double d = fabs((double)tm.pos_a - (double)tm.pos_b); if (d == 0) d = 1e-9;
                    double mu = eta / d; if (mu > 1) mu = 1;
                    double dx = *xa - *xb, dy = *ya - *yb; if (dx == 0) dx = 1e-9;
                    const double mag = sqrt(dx * dx + dy * dy), r = (mu * (mag - d) / 2) / mag;
                    *xa -= r * dx; *ya -= r * dy; *xb += r * dx; *yb += r * dy;
                }
            }
            for (uint64_t j = 0; j < wc; ++j) {   /* flush private deltas */
                const tile_t* t = &tiles[w0 + j];
                const double* px = PX + 2 * W * j; const double* py = PY + 2 * W * j;
                for (uint64_t e = 0; e < 2 * W; ++e) { const uint64_t ge = 2 * t->rmin + e; if (ge < 2 * g.n_nodes) { X[ge] += px[e] - GX[ge]; Y[ge] += py[e] - GY[ge]; } }
            }
        }
    }
    printf("{\"T\": %llu, \"W\": %llu, \"C\": %llu, \"visits\": %llu, \"tiles\": %llu, \"local_a\": %.3f, \"local_b\": %.3f, \"stress\": %.4f}\n",
           (unsigned long long)T, (unsigned long long)W, (unsigned long long)C, (unsigned long long)visits, (unsigned long long)n_tiles,
           (double)local_a / total, (double)local_b / total, orc_path_stress_sampled(&g, X, Y, 1000000, 0x5eed));
    return 0;
}
