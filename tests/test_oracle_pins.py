"""Pins the CPU oracle before it is trusted as the checker (CPU only).

Pins, strongest first: the real libstdc++ (distributions), the reference's committed layout
test/DRB1-3123_unsorted.og.lay (quality bar), reference src/unittest/pathindex.cpp known answers,
closed forms of the schedule, a third pure-Python restatement, and the committed golden vectors.
"""
import os
import subprocess

import numpy as np
import pytest

import pyref
from conftest import GOLDEN, ROOT


def test_libstdcxx_distributions_match_real_libstdcxx(orc):
    exe = os.path.join(ROOT, "oracle", "_build", "check_libstdcxx")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_splitmix64_known_answers(orc):
    import ctypes as C
    # published SplitMix64 vectors for seed 1234567 (Vigna's reference / Rosetta Code)
    want = [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431]
    s = (C.c_uint64 * 4)()
    orc.lib().orc_rng_seed(1234567, s)
    assert list(s) == want


def test_rng_and_distributions_match_python_restatement(orc):
    import ctypes as C
    for seed in (9399220, 9399221, 42):
        s = (C.c_uint64 * 4)()
        orc.lib().orc_rng_seed(seed, s)
        g = pyref.Xoshiro256Plus(seed)
        assert list(s) == g.s
        for _ in range(200):
            assert orc.lib().orc_rng_next(s) == g.next()
        for rng in (2, 3, 12, 35059, 202806):
            for _ in range(100):
                assert orc.lib().orc_uniform_u64(s, rng) == pyref.uniform_below(g, rng)
        for _ in range(100):
            assert orc.lib().orc_canonical(s) == pyref.canonical(g)


def test_fast_precise_pow_and_zipf_match_python_restatement(orc):
    import ctypes as C
    lib = orc.lib()
    rs = np.random.RandomState(1)
    for a, b in [(1.0, 0.99), (0.5, 0.99), (2.0 / 3100, 0.01), (0.3, 100.0), (1.0 / 7, 0.99), (0.99, 100.00000000000009)]:
        assert lib.orc_fast_precise_pow(a, b) == pyref.fast_precise_pow(a, b)
    for a, b in zip(rs.uniform(1e-6, 1.0, 200), rs.uniform(0.0, 3.0, 200)):
        assert lib.orc_fast_precise_pow(a, b) == pyref.fast_precise_pow(a, b)
    # the "dirty" bias: the fractional-part estimate of a^0 is ~0.97, not 1
    assert 0.9 < lib.orc_fast_precise_pow(0.5, 2.0) / 0.25 < 1.0
    z = orc.zetas(0.99, 3100, 1000, 100)
    for n, zi in [(1, 1), (2, 2), (3, 3), (17, 17), (1000, 1000), (1500, 1006), (3100, 1022)]:
        s = (C.c_uint64 * 4)()
        lib.orc_rng_seed(77 + n, s)
        g = pyref.Xoshiro256Plus(77 + n)
        for _ in range(300):
            got = lib.orc_zipf(s, n, 0.99, z[zi])
            assert got == pyref.zipf(g, n, 0.99, float(z[zi]))
            assert 1 <= got <= n


def test_schedule_closed_forms(orc):
    # path_sgd_layout.cpp:444-459 with iter_with_max_lr = 0: etas[0] = eta_max, etas[iter_max-1] = eps
    for eta_max, eps, iters in [(9.61e6, 0.01, 30), (4.797e8, 0.01, 30), (100.0, 0.5, 2), (8.6e6, 0.001, 100)]:
        p = orc.params(iter_max=iters, iter_with_max_learning_rate=0, eps=eps, eta_max=eta_max)
        e = orc.schedule(p)
        assert len(e) == iters + 1
        assert e[0] == pytest.approx(eta_max, rel=1e-12)
        assert e[iters - 1] == pytest.approx(eps, rel=1e-9)
        assert np.all(np.diff(e) < 0)
    p = orc.params(iter_max=10, iter_with_max_learning_rate=3, eps=0.01, eta_max=1000.0)
    e = orc.schedule(p)
    assert np.argmax(e) == 3 and e[2] == pytest.approx(e[4])


def test_zeta_table_rule(orc):
    theta, space, smax, q = 0.99, 3100, 1000, 100
    z = orc.zetas(theta, space, smax, q)
    assert len(z) == smax + (space - smax) // q + 1 + 1 + 1  # reference size + the guard slot
    pw = [orc.lib().orc_fast_precise_pow(1.0 / i, theta) for i in range(1, space + 1)]
    cs = np.cumsum(pw)  # same left-to-right accumulation
    run = 0.0
    exact = []
    for v in pw:
        run += v
        exact.append(run)
    assert z[0] == 0.0
    assert np.array_equal(z[1:smax + 1], np.array(exact[:smax]))
    for k in range(0, (space - smax) // q + 1):  # quantised part: zeta at the LOWER edge of each bucket
        assert z[smax + 1 + k] == exact[smax + k * q - 1]
    assert np.all(np.diff(z[1:smax + 1]) > 0)
    del cs
    # space == space_max: the reference writes one slot past its table; ours has room for it
    z2 = orc.zetas(theta, 1000, 1000, 100)
    assert len(z2) == 1000 + 1 + 1 and z2[1001] == z2[1000]
    # space < space_max
    z3 = orc.zetas(theta, 12, 1000, 100)
    assert len(z3) == 14 and np.array_equal(z3[1:13], z[1:13])


def test_golden_vectors_regression(orc, ographs):
    gv = np.load(os.path.join(GOLDEN, "golden_vectors.npz"))
    for name, max_steps in (("DRB1-3123", 3100), ("LPA", 21901), ("chr6.C4", 2932)):
        p = orc.params(iter_max=30, iter_with_max_learning_rate=0, eps=0.01, eta_max=float(max_steps) ** 2)
        assert np.array_equal(orc.schedule(p), gv[f"etas/{name}"])
        assert np.array_equal(orc.zetas(0.99, max_steps, 1000, 100), gv[f"zetas/{name}"])
    assert np.array_equal(orc.zetas(0.5, 2932, 1000, 100), gv["zetas/theta0.5_space2932"])
    g = ographs("DRB1-3123")
    p = orc.params(iter_max=30, min_term_updates=10 * g.n_steps, eps=0.01, eta_max=3100.0 ** 2, theta=0.99,
                   space=3100, space_max=1000, space_quantization_step=100, cooling_start=0.5)
    warm = orc.trace_terms(g, p, 9399220, 1, 0, False, 1000)[:, 0, :]
    cool = orc.trace_terms(g, p, 9399220, 1, 0, True, 1000)[:, 0, :]
    assert np.array_equal(warm, gv["terms/DRB1-3123/warm"])
    assert np.array_equal(cool, gv["terms/DRB1-3123/cooling"])


def test_sampler_terms_are_valid_and_follow_the_reference_rules(orc, ographs):
    g = ographs("chr6.C4")  # 104 031 reverse steps, 90 paths
    p = orc.params(iter_max=30, min_term_updates=1, eps=0.01, eta_max=1.0, theta=0.99, space=2932, space_max=1000,
                   space_quantization_step=100, cooling_start=0.5)
    for cooling in (False, True):
        t = orc.trace_terms(g, p, 5, 64, 0, cooling, 400).reshape(-1, 4)
        ka, kb = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
        assert ka.max() < g.n_steps and kb.max() < g.n_steps
        assert np.array_equal(g.step_path[ka], g.step_path[kb])       # partner on the same path
        assert set(np.unique(t[:, 2])) <= {0, 1} and set(np.unique(t[:, 3])) <= {0, 1}
        if cooling:
            assert np.all(ka != kb)                                   # a Zipf jump is >= 1
            jump = np.abs(ka - kb)
            assert np.mean(jump <= 10) > 0.25                         # Zipf: mass near the origin
        # a is uniform over all steps: mean flat index near S/2
        assert abs(ka.mean() / g.n_steps - 0.5) < 0.02


def test_pathindex_known_answers(oa, orc, tmp_path):
    """reference src/unittest/pathindex.cpp:22-130 — 4 nodes AGGA/A/TC/TCTCAGG, three paths."""
    gfa = tmp_path / "pi.gfa"
    gfa.write_text("H\tVN:Z:1.0\nS\t1\tAGGA\nS\t2\tA\nS\t3\tTC\nS\t4\tTCTCAGG\n"
                   "L\t1\t+\t2\t+\t0M\nL\t2\t+\t3\t+\t0M\nL\t2\t+\t4\t+\t0M\nL\t3\t+\t4\t+\t0M\nL\t1\t+\t4\t+\t0M\n"
                   "L\t4\t+\t3\t-\t0M\nL\t4\t+\t3\t+\t0M\n"
                   "P\t5\t1+,3+,4+\t*\nP\t5-\t1+,4+,3+\t*\nP\t5-m\t1+,4+,3-\t*\n")
    g = oa.Graph.from_gfa(gfa)
    assert (g.n_nodes, g.n_paths, g.n_steps) == (4, 3, 9)
    assert list(g.node_len) == [4, 1, 2, 7]
    assert g.path_names == ["5", "5-", "5-m"]
    # get_position_of_step: path "5" steps start at 0, 4, 6 (pathindex.cpp:127-130)
    assert list(g.step_pos[0:3]) == [0, 4, 6]
    assert list(g.step_pos[3:6]) == [0, 4, 11]
    assert list(g.step_handle[6:9]) == [0, 6, 5]      # n3_m = 2*2+1
    assert list(g.step_path) == [0, 0, 0, 1, 1, 1, 2, 2, 2]
    # node-major grouping of steps (np_bv: 1 at each node's first step, :83-97) is derivable
    order = np.argsort(g.step_handle >> 1, kind="stable")
    ranks = (g.step_handle >> 1)[order]
    np_bv = np.r_[1, (np.diff(ranks) != 0).astype(int)]
    assert np_bv.sum() == 3 and len(np_bv) == 9       # node 2 is on no path
    # from_arrays derives the same positions
    g2 = oa.Graph.from_arrays(g.node_len, g.path_first, g.step_handle)
    assert np.array_equal(g2.step_pos, g.step_pos) and np.array_equal(g2.step_path, g.step_path)


def test_reference_layout_fixture_quality_bar(oa, orc, graphs, ographs):
    """The one reference-made output: test/DRB1-3123_unsorted.og.lay (BASELINE 1b)."""
    lay = oa.Layout.load(os.path.join(GOLDEN, "DRB1-3123_unsorted.og.lay"))
    g = ographs("DRB1-3123_unsorted")
    assert lay.size() == 2 * 3214
    assert min(lay.X.min(), lay.Y.min()) == 1000.0
    assert orc.path_stress_exhaustive(g, lay.X, lay.Y) == pytest.approx(0.08709, abs(1e-4))
    per_node, per_bp = orc.path_distance(g, lay.X, lay.Y)
    assert per_node == pytest.approx(9.59988, abs=1e-4) and per_bp == pytest.approx(1.28546, abs=1e-4)


def test_reference_fixture_is_reproduced_two_sided(oa, orc, graphs, ographs):
    """The reference's one layout file against the oracle, TWO-sided and on more than one number.  Round 2 could only
    say "the oracle is at least as good" (0.0755 against the file's 0.0871, a 13 % gap no seed or thread count closes:
    nine runs give 0.0750 .. 0.0758).  A sweep of the generating parameters (tools/reference_pin_sweep.py) finds the
    one that does: WITHOUT the cooling phase (`-K 1`: never switch to Zipf-only partners; the file is older than the
    `-K` option's default of 0.5) the restatement reproduces the file — stress, `odgi stats -s`, the distribution of
    layout distance over path distance for adjacent steps and for Zipf-sampled pairs, the layout's extent — within the
    bands of tests/refstats.py, from three initial layouts, with the docs' `--threads 2`
    (docs/rst/tutorials/sort_layout.rst:365; the command the reference keeps for this very file, scripts/GIFs_doc.sh:24, is
    `odgi layout -i DRB1-3123_unsorted.og -o DRB1-3123_unsorted.og.lay -P --threads 2 -u ...`: the defaults of its day, two
    threads).  With the default cooling phase the same restatement is 13 % better, also asserted two-sided."""
    import refstats
    g, og = graphs("DRB1-3123_unsorted"), ographs("DRB1-3123_unsorted")
    lay = oa.Layout.load(os.path.join(GOLDEN, "DRB1-3123_unsorted.og.lay"))
    terms = refstats.zipf_pairs(orc, og, orc.params_from(oa.LayoutParams.defaults(g)))
    fx = refstats.layout_stats(orc, g, og, lay.X, lay.Y, terms)
    for k, v in refstats.FIXTURE.items():   # the constants the GPU test compares with are the file's
        assert np.allclose(fx[k], v, rtol=2e-3), (k, fx[k], v)
    runs = {1.0: [], 0.5: []}
    for cs in runs:
        p = oa.LayoutParams.defaults(g, cooling_start=cs)
        for seed in (7, 8, 9):
            X0, Y0 = oa.initial_layout(g, "d", seed=seed)
            X, Y, _ = orc.layout_hogwild(og, orc.params_from(p), 2, X0, Y0)
            runs[cs].append(refstats.layout_stats(orc, g, og, X, Y, terms))
    no_cooling, default = refstats.mean_stats(runs[1.0]), refstats.mean_stats(runs[0.5])
    print("reference file      ", fx)
    print("oracle, no cooling  ", no_cooling)
    print("oracle, default -K  ", default)
    refstats.assert_matches_fixture(no_cooling, "oracle without cooling")
    assert 0.0735 <= default["stress"] <= 0.0775 and 8.8 <= default["per_node"] <= 9.1, default


def test_oracle_hogwild_reaches_reference_quality(oa, orc, graphs, ographs):
    """The oracle's restatement of the reference loop lays DRB1-3123_unsorted out as well as the
    reference did (0.0871): this pins the whole restated chain end to end, statistically."""
    g, og = graphs("DRB1-3123_unsorted"), ographs("DRB1-3123_unsorted")
    p = oa.LayoutParams.defaults(g)
    X0, Y0 = oa.initial_layout(g, "d", seed=7)
    assert orc.path_stress_exhaustive(og, X0, Y0) > 1000
    X, Y, st = orc.layout_hogwild(og, orc.params_from(p), 4, X0, Y0)
    assert st["iterations"] == 30 and st["terms"] >= 30 * p.min_term_updates
    assert orc.path_stress_exhaustive(og, X, Y) < 0.0871 * 1.25
    # the serialised stream schedule (what the GPU runs) gives the same quality
    X, Y = orc.layout_streams_f64(og, orc.params_from(p), 9399220, 64, X0, Y0)
    assert orc.path_stress_exhaustive(og, X, Y) < 0.0871 * 1.25
    Xf, Yf, dmax = orc.layout_streams_f32(og, orc.params_from(p), 9399220, 64, X0, Y0)
    assert orc.path_stress_exhaustive(og, Xf, Yf) < 0.0871 * 1.25 and dmax > 0


def tile_mirror_case(orc, pyref, policy=0):
    """The committed tile-mirror case (also called by tests/golden/make_golden.py): DRB1-3123, region 64,
    6 iterations of 2*S terms, deterministic initial layout (X = cumulative bp at node ends, Y = a fixed
    pattern), frame 16 quanta per bp around it.  Returns the final coordinates and bookkeeping.
    policy 0 = the launch order the product ships; TILE_DRAIN_AFTER | TILE_TWO_SNAPSHOTS = round 2's."""
    from conftest import parse_gfa_py
    d = parse_gfa_py(os.path.join(GOLDEN, "DRB1-3123.gfa"))
    g = orc.Graph(d["node_len"], d["path_first"], d["step_path"], d["step_handle"], d["step_pos"])
    counts = np.diff(d["path_first"].astype(np.int64))
    max_steps = int(counts.max())
    p = orc.params(iter_max=6, iter_with_max_learning_rate=0, min_term_updates=2 * g.n_steps, delta=0.0, eps=0.01,
                   eta_max=float(max_steps) ** 2, theta=0.99, space=max_steps, space_max=1000, space_quantization_step=100,
                   cooling_start=0.5)
    tiles, items = pyref.build_tiles_py(d["path_first"], d["step_handle"], 64, 56)
    ends = np.cumsum(np.repeat(d["node_len"].astype(np.float64), 2) * np.tile([0.0, 1.0], g.n_nodes))
    X0 = ends.astype(np.float32)
    Y0 = (((np.arange(2 * g.n_nodes) * 2654435761) % 1000) / 10.0 - 50.0).astype(np.float32)
    X, Y, dmax, ck, far = orc.tile_layout_q32(g, p, 9399220, tiles, items, 64, X0, Y0, -float(1 << 27), -float(1 << 27), 16.0, policy=policy)
    return dict(X=X, Y=Y, dmax=np.array([dmax]), checksum=ck, far=np.array([far], dtype=np.uint64),
                n_tiles=np.array([len(tiles["t0"])]), n_items=np.array([len(items["local"])]),
                n_windowless=np.array([int((items["local"] == 0).sum())]))


def test_tile_mirror_golden_regression(orc):
    """The oracle's mirror of the tile kernel against its committed output (the GPU test compares the kernel
    with the same mirror), and the invariants of a run: coordinate sums conserved, some far partners."""
    gv = np.load(os.path.join(GOLDEN, "golden_vectors.npz"))
    got = tile_mirror_case(orc, pyref)
    for k, v in got.items():
        assert np.array_equal(v, gv[f"tile_mirror/{k}"]), k
    ck = got["checksum"]
    assert (ck[0], ck[1]) == (ck[2], ck[3]) and got["far"][0] > 0 and got["n_windowless"][0] > 0
    # round 2's launch order (far pulls delivered right after their launch, two snapshots per warm iteration) is still
    # in the mirror as a policy, and still gives the vectors committed in round 2: the terms and their arithmetic did
    # not change with the order of the launches
    old = tile_mirror_case(orc, pyref, policy=orc.TILE_ROUND2)
    for k, v in old.items():
        assert np.array_equal(v, gv[f"tile_mirror_r2/{k}"]), k
    assert not np.array_equal(old["X"], got["X"])
    # round 3's pipeline (the Zipf/uniform coin of a warm term per lane; round 4 draws it per wave and trip) likewise
    # still gives the vectors committed in round 3
    r3 = tile_mirror_case(orc, pyref, policy=orc.TILE_ROUND3)
    for k, v in r3.items():
        assert np.array_equal(v, gv[f"tile_mirror_r3/{k}"]), k
    assert not np.array_equal(r3["X"], got["X"])
    # rounds 4-5 (today's sampler; the far pulls of a launch ramped 0.1 0.1 0.2 0.3 0.4 to half a projection where round 6
    # ramps 0.2 .. 0.8 to one) still give the vectors committed in round 5
    r5 = tile_mirror_case(orc, pyref, policy=orc.TILE_ROUND5)
    for k, v in r5.items():
        assert np.array_equal(v, gv[f"tile_mirror_r5/{k}"]), k
    assert not np.array_equal(r5["X"], got["X"])


def test_tile_sampler_draws_the_reference_term_distribution(orc, ographs):
    """The tile kernel's sampler takes its coins from the bits of one word (two words per term) where the reference
    worker draws a word per coin, and (round 4) the Zipf/uniform coin of a warm term is the wave's for the trip: the 64
    terms a wave draws together share it.  Same distribution of terms: partner offsets and end choices of the tile
    sampler's terms against the reference-order sampler's (path_sgd_layout.cpp:182-270) on the same path, warm and
    cooling.  Terms of one wave-trip are independent GIVEN their coin, so the warm comparison is made three ways that
    are each a valid test: one term per wave-trip against the reference's warm terms (independent draws of the mixture);
    all terms of the Zipf trips against the reference's cooling terms (all Zipf); and the coins themselves as fair."""
    from scipy.stats import chi2_contingency
    g = ographs("DRB1-3123")  # 12 paths of ~3000 steps
    lens = np.diff(g.path_first)
    path = int(np.argmax(lens))
    first, L = int(g.path_first[path]), int(lens[path])
    p = orc.params(iter_max=30, min_term_updates=1, eps=0.01, eta_max=1.0, theta=0.99, space=3100, space_max=1000,
                   space_quantization_step=100, cooling_start=0.5)
    edges = np.array([1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 1 << 20])

    def table(t):
        d = t[:, 1] - t[:, 0]
        mag = np.digitize(np.abs(d), edges)                        # 0: the same step (uniform partner only)
        return np.bincount(mag * 2 + (d < 0), minlength=2 * (len(edges) + 1))

    def same_distribution(x, y, what):
        counts = np.stack([table(x), table(y)])
        counts = counts[:, counts.sum(axis=0) >= 20]
        chi2, pval, dof, _ = chi2_contingency(counts)
        assert pval > 1e-3, (what, chi2, dof, pval)

    ref_zipf = None
    for cooling in (True, False):
        ref = orc.trace_terms(g, p, 77, 256, 0, cooling, 6000).reshape(-1, 4).astype(np.int64)
        ref = ref[g.step_path[ref[:, 0]] == path]                      # first step uniform over this path's steps
        lanes, M = 64, 400000 if cooling else 64 * 100000
        til = orc.tile_terms(g, p, 12345, 3, M, L, 0, lanes, first, 0, L, path, cooling, capacity=M).astype(np.int64)
        assert len(til) == M and len(ref) > 50000
        assert til[:, 0].min() >= first and til[:, 0].max() < first + L and np.array_equal(g.step_path[til[:, 1]], g.step_path[til[:, 0]])
        if cooling:
            same_distribution(ref, til, "cooling")
            ref_zipf = ref
        else:
            # term q of the tile is drawn by lane q % lanes in its trip q // lanes (one wave: lanes = 64)
            trips = M // lanes
            coins = np.array([orc.tile_wave_coin(12345, 3, 0, 0, j) for j in range(trips)])
            assert abs(coins.mean() - 0.5) < 4 * 0.5 / np.sqrt(trips), coins.mean()
            assert abs(np.mean(coins[1:] == coins[:-1]) - 0.5) < 4 * 0.5 / np.sqrt(trips)   # no serial dependence
            one_per_trip = til[np.arange(trips) * lanes + (np.arange(trips) * 7) % lanes]
            same_distribution(ref, one_per_trip, "warm, one term per wave-trip")
            zipf_trips = til[np.repeat(coins == 1, lanes)]
            same_distribution(ref_zipf, zipf_trips[:400000], "warm, the Zipf trips")
            # a uniform trip's partners are uniform over the path's steps, whatever the first step
            uni = til[np.repeat(coins == 0, lanes)][:400000]
            kb = np.bincount((uni[:, 1] - first) * 16 // L, minlength=16)
            # (partner quads: lanes 1..3 of four consecutive lanes take their first lane's line — the first lanes' draws are the independent ones)
            lead = uni[0::4]
            kb_lead = np.bincount((lead[:, 1] - first) * 16 // L, minlength=16)
            assert np.all(np.abs(kb_lead - len(lead) / 16) < 5 * np.sqrt(len(lead) / 16)), kb_lead
            assert np.all(np.abs(kb - len(uni) / 16) < 5 * np.sqrt(4 * len(uni) / 16)), kb
            # the lanes of a wave share a 128-byte line of four step records in a uniform trip (pgsgd_tiles.hpp: tile_quad_partner): lane r's
            # partner is flat step lead ^ r unless that step lies outside the path
            for r in (1, 2, 3):
                other = uni[r::4]
                twin = lead[:len(other), 1] ^ r
                inside = (twin >= first) & (twin < first + L)
                assert np.array_equal(other[inside, 1], twin[inside]) and inside.mean() > 0.999
                assert np.all((other[~inside, 1] >= first) & (other[~inside, 1] < first + L))
                # ... and is itself uniform over the path's steps, with fair end choices of its own
                kb_r = np.bincount((other[:, 1] - first) * 16 // L, minlength=16)
                assert np.all(np.abs(kb_r - len(other) / 16) < 5 * np.sqrt(len(other) / 16)), (r, kb_r)
            # the pairs of rounds 4-6 (share=2) are still what the oracle draws when asked: odd lanes take their even neighbour's twin
            til2 = orc.tile_terms(g, p, 12345, 3, 64 * 2000, L, 0, lanes, first, 0, L, path, cooling, capacity=64 * 2000, share=2).astype(np.int64)
            uni2 = til2[np.repeat(coins[:2000] == 0, lanes)]
            even, odd = uni2[0::2], uni2[1::2]
            twin = even[:, 1] ^ 1
            inside = (twin >= first) & (twin < first + L)
            assert np.array_equal(odd[inside, 1], twin[inside]) and inside.mean() > 0.999
            # (the streams are the same under either rule: the quads' first lanes and the pairs' even lanes of the same trips drew the same partners)
            uni4 = til[:64 * 2000][np.repeat(coins[:2000] == 0, lanes)]
            assert np.array_equal(uni4[0::4, 1], uni2[0::4, 1]) and np.array_equal(uni4[:, 0], uni2[:, 0])
            til = til[:400000]
            M = len(til)
        # first steps: uniform over the path in both
        pos = np.stack([np.bincount((t[:, 0] - first) * 16 // L, minlength=16) for t in (ref, til)])
        assert chi2_contingency(pos)[1] > 1e-3
        # end choices: two independent fair coins (offset = orientation xor coin; the orientation is a step property)
        rev_a, rev_b = g.step_handle[til[:, 0]] & 1, g.step_handle[til[:, 1]] & 1
        ca, cb = til[:, 2] ^ rev_a, til[:, 3] ^ rev_b
        joint = np.bincount(ca * 2 + cb, minlength=4)
        assert np.all(np.abs(joint - M / 4) < 5 * np.sqrt(M * 3 / 16)), joint
