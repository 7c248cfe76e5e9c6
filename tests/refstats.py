"""Distributional statistics of a 2D layout, used to compare layouts with the one layout file the reference holds
(test/DRB1-3123_unsorted.og.lay) two-sidedly — not only "stress at most ...":
  stress      exhaustive path stress (the oracle's evaluator)
  per_node, per_bp   `odgi stats -s` 2D path distance per node / per bp
  adj         10th / 50th / 90th percentile of |p_a - p_b| / d over all pairs of consecutive path steps
  zipf        the same over 256 000 pairs drawn by the reference's sampler in its cooling mode (Zipf partners)
  extent, aspect     sqrt of the larger eigenvalue of the coordinate covariance; ratio of the two (rotation invariant)
"""
import numpy as np


def zipf_pairs(orc, og, oparams, n_streams=64, per_stream=4000, seed=12345):
    return orc.trace_terms(og, oparams, seed, n_streams, 0, True, per_stream).reshape(-1, 4)


def layout_stats(orc, g, og, X, Y, terms):
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    sh = np.asarray(g.step_handle).astype(np.int64)
    pos = np.asarray(g.step_pos).astype(np.int64)
    nl = np.asarray(g.node_len).astype(np.int64)
    pf = np.asarray(g.path_first).astype(np.int64)
    out = {"stress": orc.path_stress_exhaustive(og, X, Y)}
    out["per_node"], out["per_bp"] = orc.path_distance(og, X, Y)
    ks = np.concatenate([np.arange(pf[i], pf[i + 1] - 1) for i in range(len(pf) - 1)])
    a, b = sh[ks], sh[ks + 1]
    d = (pos[ks + 1] - pos[ks]).astype(np.float64)
    r = np.hypot(X[a] - X[b], Y[a] - Y[b]) / np.maximum(d, 1e-9)
    out["adj"] = [float(np.percentile(r, q)) for q in (10, 50, 90)]
    ka, kb, oa_, ob = (terms[:, i].astype(np.int64) for i in range(4))
    ea, eb = (sh[ka] & ~1) | oa_, (sh[kb] & ~1) | ob
    pa = pos[ka] + np.where((sh[ka] & 1) != oa_, nl[sh[ka] >> 1], 0)
    pb = pos[kb] + np.where((sh[kb] & 1) != ob, nl[sh[kb] >> 1], 0)
    d = np.abs(pa - pb).astype(np.float64)
    m = d > 0
    r = np.hypot(X[ea] - X[eb], Y[ea] - Y[eb])[m] / d[m]
    out["zipf"] = [float(np.percentile(r, q)) for q in (10, 50, 90)]
    ev = np.linalg.eigvalsh(np.cov(np.stack([X, Y])))
    out["extent"] = float(np.sqrt(ev[1]))
    out["aspect"] = float(np.sqrt(ev[1] / max(ev[0], 1e-12)))
    return out


# what the reference's file measures (tests/test_oracle_pins.py asserts these from the file itself)
FIXTURE = {"stress": 0.08709, "per_node": 9.59988, "per_bp": 1.28546, "adj": [0.880, 1.214, 2.293], "zipf": [0.882, 1.147, 1.593],
           "extent": 3251.9, "aspect": 4.151}


def mean_stats(runs):
    out = {}
    for k in runs[0]:
        v = np.array([r[k] for r in runs], dtype=np.float64)
        out[k] = v.mean(0).tolist() if v.ndim > 1 else float(v.mean())
    return out


def assert_matches_fixture(m, what, stress_band=0.03):
    """Two-sided bands around the reference file's statistics for a mean of three runs WITHOUT a cooling phase (see
    test_reference_fixture_is_reproduced_two_sided).  Bands from nine CPU-restatement runs (3 seeds x 1/2/4 threads):
    stress 0.0868..0.0886, per node 9.50..9.60, Zipf-pair percentiles 0.885..0.894 / 1.147..1.151 / 1.581..1.595,
    adjacent-pair median 1.212..1.217, 10th percentile 0.883..0.936, extent 3251..3276."""
    f = FIXTURE
    rel = lambda a, b: abs(a / b - 1.0)
    assert rel(m["stress"], f["stress"]) <= stress_band, (what, m)
    assert rel(m["per_node"], f["per_node"]) <= 0.02 and rel(m["per_bp"], f["per_bp"]) <= 0.02, (what, m)
    for i, band in enumerate((0.03, 0.01, 0.02)):
        assert rel(m["zipf"][i], f["zipf"][i]) <= band, (what, "zipf", i, m)
    assert rel(m["adj"][1], f["adj"][1]) <= 0.01 and rel(m["adj"][0], f["adj"][0]) <= 0.07 and rel(m["adj"][2], f["adj"][2]) <= 0.20, (what, m)
    assert rel(m["extent"], f["extent"]) <= 0.015, (what, m)
