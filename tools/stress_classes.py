#!/usr/bin/env python3
"""Where a layout's sampled path stress sits: the SGD sampler's pairs (warm mode: half Zipf, half uniform partners, both end
coins — what `path_stress` draws, path_sgd_layout.cpp:178-269) classified by step distance, end choices and path distance,
with every class's CONTRIBUTION to the total (sum of squared relative errors / all counted pairs).  The pairs depend only on
(graph, n_pairs, seed): two layouts of one graph — GPU tile kernel, GPU per-lane kernel, CPU oracle, on different machines —
are scored on the same pairs, class by class.

    classes(g, X, Y, n_pairs=4_000_000, seed=7) -> {"total": s, "pairs": n, "by": {class: [count, contribution, mean]}}

Test / experiment infrastructure (numpy only); not part of the product."""
import numpy as np

TILE_STEPS = 224
REGION = 256


def _zipf_table(theta, n):
    return np.cumsum(1.0 / np.arange(1, n + 1, dtype=np.float64) ** theta)


def sample_pairs(g, n_pairs, seed, theta=0.99):
    """ka, kb (flat steps), flip_a, flip_b, zipf flag; the reference's warm-iteration sampler, vectorised."""
    rs = np.random.RandomState(seed)
    first = g.path_first.astype(np.int64)
    cnt = np.diff(first)
    ka = (rs.rand(n_pairs) * g.n_steps).astype(np.int64)
    pth = np.searchsorted(first, ka, side="right") - 1
    pstart, c = first[pth], cnt[pth]
    s_rank = ka - pstart
    zipf = rs.rand(n_pairs) < 0.5
    back = ((s_rank > 0) & (rs.rand(n_pairs) < 0.5)) | (s_rank == c - 1)
    room = np.where(back, s_rank, c - s_rank - 1)
    H = _zipf_table(theta, int(cnt.max()))
    jump = np.maximum(room, 1)
    z = np.searchsorted(H, rs.rand(n_pairs) * H[jump - 1], side="right") + 1
    z = np.minimum(z, jump)
    b_rank = np.where(zipf, np.where(back, s_rank - z, s_rank + z), (rs.rand(n_pairs) * c).astype(np.int64))
    ok = (c > 1) & (b_rank >= 0) & (b_rank < c)
    kb = pstart + np.clip(b_rank, 0, c - 1)
    fa = rs.rand(n_pairs) < 0.5
    fb = rs.rand(n_pairs) < 0.5
    return ka[ok], kb[ok], fa[ok], fb[ok], zipf[ok], pstart[ok]


def classes(g, X, Y, n_pairs=4_000_000, seed=7):
    ka, kb, fa, fb, zipf, pstart = sample_pairs(g, n_pairs, seed)
    sh, sp, nl = g.step_handle, g.step_pos, g.node_len
    ha, hb = sh[ka].astype(np.int64), sh[kb].astype(np.int64)
    pa = sp[ka].astype(np.float64) + np.where(fa, nl[ha >> 1], 0)
    pb = sp[kb].astype(np.float64) + np.where(fb, nl[hb >> 1], 0)
    ea, eb = ha ^ fa, hb ^ fb
    d = np.abs(pa - pb)
    keep = d > 0
    mag = np.hypot(X[ea] - X[eb], Y[ea] - Y[eb])
    e2 = np.where(keep, ((mag - np.where(keep, d, 1.0)) / np.where(keep, d, 1.0)) ** 2, 0.0)
    n = int(keep.sum())
    dz = np.abs(kb - ka)
    zc = np.select([~zipf, dz <= 1, dz <= 3, dz <= 30, dz <= 1000], ["uniform", "z1", "z2-3", "z4-30", "z31-1000"], "z>1000")
    fc = np.where(fa, "e", "s").astype(object) + np.where(fb, "e", "s").astype(object)
    # the pair as (earlier step, later step): "se" = start of the earlier node, end of the later one
    swap = kb < ka
    fc = np.where(swap, np.where(fb, "e", "s").astype(object) + np.where(fa, "e", "s").astype(object), fc)
    dc = np.select([d <= 2, d <= 5, d <= 20, d <= 100], ["d<=2", "d3-5", "d6-20", "d21-100"], "d>100")
    tile_cut = ((ka - pstart) // TILE_STEPS) != ((kb - pstart) // TILE_STEPS)
    win_cut = ((ha >> 1) // REGION) != ((hb >> 1) // REGION)
    by = {}

    def add(name, m):
        m = m & keep
        c = int(m.sum())
        if c:
            s = float(e2[m].sum())
            by[name] = [c, s / n, s / c]

    for z in ("z1", "z2-3", "z4-30", "z31-1000", "z>1000", "uniform"):
        add(z, zc == z)
    for f in ("ss", "se", "es", "ee"):
        add("z1/" + f, (zc == "z1") & (fc == f))
        add("z2-3/" + f, (zc == "z2-3") & (fc == f))
    for dd in ("d<=2", "d3-5", "d6-20", "d21-100", "d>100"):
        add("z1/" + dd, (zc == "z1") & (dc == dd))
        add("z2-3/" + dd, (zc == "z2-3") & (dc == dd))
    near = (zc == "z1") | (zc == "z2-3") | (zc == "z4-30")
    add("z<=30/tile-cut", near & tile_cut)
    add("z<=30/region-cut", near & win_cut)
    add("z<=30/same-tile-and-region", near & ~tile_cut & ~win_cut)
    # the heavy tail: how much of the total the worst pairs carry
    srt = np.sort(e2[keep])[::-1]
    tail = {f"top{k}": float(srt[:k].sum() / n) for k in (100, 1000, 10000, 100000) if k < n}
    # the two ends of one node against its length (never a term of its own: held by the terms of its neighbours)
    rs = np.random.RandomState(seed + 1)
    nodes = (rs.rand(min(n_pairs, 2_000_000)) * g.n_nodes).astype(np.int64)
    ln = nl[nodes].astype(np.float64)
    seg = np.hypot(X[2 * nodes] - X[2 * nodes + 1], Y[2 * nodes] - Y[2 * nodes + 1])
    node = {"mean_rel_err2": float(np.mean(((seg - ln) / ln) ** 2)), "mean_ratio": float(np.mean(seg / ln)),
            "short(len<=2)_rel_err2": float(np.mean((((seg - ln) / ln) ** 2)[ln <= 2])) if (ln <= 2).any() else None}
    return {"total": float(e2.sum() / n), "pairs": n, "by": by, "tail": tail, "node_segments": node}


def diff_table(a, b, names=("a", "b")):
    """printable comparison of two classes() results on the same pairs"""
    rows = [f"{'class':28s} {'count':>9s} {names[0]:>12s} {names[1]:>12s} {'b-a':>11s}  (contribution to the total)"]
    rows.append(f"{'TOTAL':28s} {a['pairs']:9d} {a['total']:12.5f} {b['total']:12.5f} {b['total'] - a['total']:+11.5f}")
    for k in a["by"]:
        if k in b["by"]:
            rows.append(f"{k:28s} {a['by'][k][0]:9d} {a['by'][k][1]:12.5f} {b['by'][k][1]:12.5f} {b['by'][k][1] - a['by'][k][1]:+11.5f}")
    for k in a["tail"]:
        rows.append(f"{'tail ' + k:28s} {'':9s} {a['tail'][k]:12.5f} {b['tail'].get(k, float('nan')):12.5f}")
    rows.append(f"node segments: {names[0]} {a['node_segments']}  {names[1]} {b['node_segments']}")
    return "\n".join(rows)
