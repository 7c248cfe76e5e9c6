"""The committed CPU reference distributions (tests/golden/cpu_reference_distributions.json): every configuration a
statistical GPU test looks up is there, with enough runs, and the file's summaries follow from its runs.  Runs without a
GPU: a key that drifted (a default that changed, a graph that is built differently) fails HERE, not on the GPU box."""
import numpy as np
import pytest

import cpu_reference as cr


def _fixture_keys(oa, graphs):
    from odgi_amd.sort import sort_params_defaults
    keys = []
    for name in ("DRB1-3123", "LPA", "chr6.C4"):
        keys.append(cr.key(name, oa.LayoutParams.defaults(graphs(name))))
    g = graphs("chr6.C4")
    for theta in (0.5, 0.9, 0.99, 0.999):
        for K in (0.25, 0.5, 0.75):
            keys.append(cr.key("chr6.C4", oa.LayoutParams.defaults(g, theta=theta, cooling_start=K), "h"))
    keys.append(cr.key("5000-paths", oa.LayoutParams.defaults(cr.many_paths_graph(oa))))
    for name in ("DRB1-3123", "DRB1-3123_unsorted", "chr6.C4"):
        keys.append(cr.key("1d:" + name, sort_params_defaults(graphs(name)), "1d"))
    g1, _ = cr.shuffled_linear_graph(oa, n_nodes=20000, n_paths=8)
    keys.append(cr.key("1d:shuffled-20000", sort_params_defaults(g1, iter_max=30, min_term_updates=10 * g1.n_steps), "1d"))
    return keys


def test_every_configuration_the_gpu_tests_look_up_is_committed(oa, graphs):
    db = cr.load()
    keys = _fixture_keys(oa, graphs)
    g = cr.synthetic_300k(oa)
    keys.append(cr.key("synthetic-300k", oa.LayoutParams.defaults(g, min_term_updates=3 * g.n_steps)))
    for k in keys:
        assert k in db["entries"], f"{k!r} has no committed CPU distribution: run tools/make_cpu_reference_distributions.py"
        e = db["entries"][k]
        assert e["stress"]["n"] >= 8, (k, e["stress"]["n"])
    assert len(set(keys)) == len(keys) == 21


def test_summaries_follow_from_the_runs_and_bands_are_sane():
    db = cr.load()
    for k, e in db["entries"].items():
        for metric in ("stress", "path_distance", "order_quality"):
            if metric not in e:
                continue
            d = e[metric]
            again = cr.summarize(d["runs"])
            for f in ("mean", "sigma", "min", "max", "median", "sigma_robust"):
                assert again[f] == pytest.approx(d[f], rel=1e-12, abs=1e-15), (k, metric, f)
            lo, hi = cr.band(d)
            assert lo < d["median"] < hi and lo > 0, (k, metric, lo, hi)
            # no vacuous side: the lower bound is never below 0.4 of the centre (the widest committed distribution, chr6.C4 -N h
            # theta 0.5 K 0.75, has its lowest run at 0.53 of its median), the upper never above 1.6 of it
            assert lo >= 0.4 * d["median"] and hi <= 1.6 * d["median"], (k, metric, lo, hi, d["median"])
            # the default band is never tighter than 10 % either way, and a distribution's own median lies inside it
            assert hi - d["median"] >= 0.0999 * d["median"] and d["median"] - lo >= 0.0999 * d["median"]
    # what the old tests did — ONE live run as the yardstick — would have failed the band of its own distribution in some
    # configurations: the reason the distributions are committed
    wide = [k for k, e in db["entries"].items() if (e["stress"]["max"] - e["stress"]["min"]) > 0.2 * e["stress"]["median"]]
    assert wide, "expected some configurations whose CPU runs scatter by more than 20 %"


def test_a_missing_configuration_is_an_error_not_a_reroll(oa, graphs):
    p = oa.LayoutParams.defaults(graphs("DRB1-3123"), theta=0.123)
    with pytest.raises(KeyError):
        cr.entry("DRB1-3123", p)
    assert np.isfinite(cr.entry("DRB1-3123", oa.LayoutParams.defaults(graphs("DRB1-3123")))["stress"]["median"])


def test_committed_yardsticks_were_rolled_with_todays_oracle():
    """Every committed yardstick of the CPU restatement's Hogwild loop names the text of the oracle functions it follows from
    (cpu_reference.hogwild_source_id: generator, sampler, schedule, the 2D and 1D loops).  A change to any of them leaves the
    yardstick stale — and fails here, in the CPU suite, not on the GPU box whose statistical tests and smoke() read the files.
    Every entry also says how many threads rolled it (the restatement's result moves with its thread count)."""
    import json
    import os
    sid = cr.hogwild_source_id()
    seen = 0
    for name in cr.YARDSTICK_FILES:
        path = os.path.join(cr.GOLDEN, name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            d = json.load(f)
        assert d.get("oracle_hogwild_source_id") == sid, f"{name} was rolled with oracle sources {d.get('oracle_hogwild_source_id')}, today's are {sid}: regenerate it ({d.get('generator')})"
        seen += 1
        if "entries" in d:
            assert all(int(e.get("threads", 0)) >= 1 for e in d["entries"].values()), name
        else:
            assert int(d.get("threads", 0)) >= 1, name
    assert seen >= 4
