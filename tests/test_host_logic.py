"""Host logic of the product on the CPU: GFA lowering, defaults, schedule/zeta tables vs the
oracle, initial layouts, .lay/TSV, component packing, CLI argument handling, C-ABI surface."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, ROOT, parse_gfa_py


@pytest.mark.parametrize("name", list(FIXTURES))
def test_gfa_lowering_matches_independent_parser(oa, graphs, name):
    g = graphs(name)
    N, S, P, max_steps = FIXTURES[name]
    assert (g.n_nodes, g.n_steps, g.n_paths, g.max_path_steps()) == (N, S, P, max_steps)
    d = parse_gfa_py(os.path.join(GOLDEN, name + ".gfa"))
    for k in ("node_len", "path_first", "step_path", "step_handle", "step_pos"):
        assert np.array_equal(getattr(g, k), d[k]), k
    assert np.array_equal(g.edges, d["edges"])
    assert g.path_names == d["names"]


def test_fixture_facts_from_survey(graphs):
    g = graphs("DRB1-3123")
    assert int((g.step_handle & 1).sum()) == 3096 and int(g.node_len.sum()) == 21997
    g = graphs("chr6.C4")
    assert int((g.step_handle & 1).sum()) == 104031


def test_gfa_errors_are_codes_not_exits(oa, tmp_path):
    from odgi_amd._lib import PgsgdError
    with pytest.raises(PgsgdError) as e:
        oa.Graph.from_gfa(tmp_path / "missing.gfa")
    assert e.value.code == -5
    bad = tmp_path / "gap.gfa"
    bad.write_text("S\t1\tA\nS\t3\tC\nP\tx\t1+,3+\t*\n")
    with pytest.raises(PgsgdError) as e:
        oa.Graph.from_gfa(bad)
    assert e.value.code == -7  # not optimized: ids are not 1..N (layout_main.cpp:148-151)
    bad.write_text("S\ts1\tA\n")
    with pytest.raises(PgsgdError) as e:
        oa.Graph.from_gfa(bad)
    assert e.value.code == -6
    bad.write_text("S\t1\tA\nS\t2\tC\nP\tx\t1+,7+\t*\n")
    with pytest.raises(PgsgdError) as e:
        oa.Graph.from_gfa(bad)
    assert e.value.code == -6 and "missing node" in str(e.value)
    bad.write_text("S\t1\tA\nS\t1\tC\n")
    with pytest.raises(PgsgdError):
        oa.Graph.from_gfa(bad)
    # ragged but legal: empty path, single-step path, CRLF, unknown line types, W lines ignored
    ok = tmp_path / "ok.gfa"
    ok.write_text("H\tVN:Z:1.0\r\nS\t1\tACGT\r\nS\t2\tG\r\nW\tx\t0\tc\t0\t5\t>1>2\r\nP\tempty\t*\t*\r\nP\tone\t2-\t*\r\nP\ttwo\t1+,2-\t*\r\n")
    g = oa.Graph.from_gfa(ok)
    assert (g.n_nodes, g.n_paths, g.n_steps) == (2, 3, 3)
    assert list(g.path_first) == [0, 0, 1, 3] and list(g.step_handle) == [3, 0, 3] and list(g.step_pos) == [0, 0, 4]


def test_gfa_errors_are_the_first_in_file_order_on_any_thread_count(oa, tmp_path):
    """S lines are parsed on threads (slices of 65536+ lines each): a duplicate node id is reported for the LATER of its
    two lines, whichever thread met it first, and the error returned is the first one in file order — so one thread and
    eight threads give the same message."""
    from odgi_amd._lib import PgsgdError
    n = 300_000
    ids = [str(i + 1) for i in range(n)]
    ids[250_000] = "5"                      # duplicate of line 4's id, a quarter of a million lines later
    f = tmp_path / "dup.gfa"
    f.write_text("".join(f"S\t{i}\tA\n" for i in ids))
    msgs = set()
    for threads in (1, 8, 5):
        with pytest.raises(PgsgdError) as e:
            oa.Graph.from_gfa(f, threads=threads)
        assert e.value.code == -6
        msgs.add(str(e.value).split("(")[-1])
    assert len(msgs) == 1 and "duplicate node id 5" in msgs.pop()
    ids[120_000] = "x7"                     # an earlier error in file order wins, wherever the threads' slices fall
    f.write_text("".join(f"S\t{i}\tA\n" for i in ids))
    for threads in (1, 8):
        with pytest.raises(PgsgdError) as e:
            oa.Graph.from_gfa(f, threads=threads)
        assert "x7" in str(e.value)


def test_defaults_match_reference_rules(oa, graphs):
    # C1 DRB1-3123 -> min_term_updates 350 590, eta_max 9.61e6, space 3100
    p = oa.LayoutParams.defaults(graphs("DRB1-3123"))
    assert (p.iter_max, p.min_term_updates, p.eta_max, p.space, p.space_max, p.space_quantization_step) == \
        (30, 350590, 3100.0 ** 2, 3100, 1000, 100)
    assert (p.theta, p.eps, p.delta, p.cooling_start, p.seed) == (0.99, 0.01, 0.0, 0.5, 9399220)
    assert p.first_cooling_iteration() == 15
    p = oa.LayoutParams.defaults(graphs("LPA"))
    assert p.min_term_updates == 2028060 and p.eta_max == 21901.0 ** 2
    p = oa.LayoutParams.defaults(graphs("chr6.C4"))
    assert p.min_term_updates == 1712080 and p.space == 2932


def test_schedule_and_zetas_bitwise_equal_to_oracle(oa, orc, graphs):
    for name in ("DRB1-3123", "LPA", "chr6.C4"):
        p = oa.LayoutParams.defaults(graphs(name))
        assert np.array_equal(oa.path_linear_sgd_layout_schedule(p), orc.schedule(orc.params_from(p)))
        assert np.array_equal(oa.zeta_table(p.theta, p.space, p.space_max, p.space_quantization_step),
                              orc.zetas(p.theta, p.space, p.space_max, p.space_quantization_step))
    for theta, space, smax, q in [(0.5, 2932, 1000, 100), (0.999, 1000, 1000, 100), (0.9, 7, 1000, 2), (0.99, 50000, 10, 7)]:
        assert np.array_equal(oa.zeta_table(theta, space, smax, q), orc.zetas(theta, space, smax, q))


def test_product_sampler_equals_oracle_sampler_on_host(oa, orc, graphs, ographs):
    """pgsgd_path_stress draws its pairs with the product's own host build of the device sampler
    (pgsgd_math.hpp); the oracle's evaluator draws them with the C restatement.  Bit-equal results
    mean the two samplers produced the same 200k terms."""
    for name in ("DRB1-3123", "chr6.C4"):
        g, og = graphs(name), ographs(name)
        X, Y = oa.initial_layout(g, "h")
        a = oa.path_stress(g, X, Y, 200000, seed=123)
        b = orc.path_stress_sampled(og, X, Y, 200000, seed=123)
        assert a == b and a > 0
        assert oa.path_distance(g, X, Y) == orc.path_distance(og, X, Y)


def test_initial_layouts(oa, graphs):
    g = graphs("DRB1-3123")
    N = g.n_nodes
    X, Y = oa.initial_layout(g, "d", seed=3)
    cs = np.r_[0, np.cumsum(g.node_len.astype(np.int64))]
    assert np.array_equal(X[0::2], cs[:-1]) and np.array_equal(X[1::2], cs[1:])   # layout_main.cpp:321-326
    assert abs(Y.std() - np.sqrt(2 * N)) < 0.1 * np.sqrt(2 * N)
    X2, Y2 = oa.initial_layout(g, "d", seed=3)
    assert np.array_equal(Y, Y2)                                                   # seeded -> reproducible
    _, Y3 = oa.initial_layout(g, "d", seed=0)
    assert not np.array_equal(Y, Y3)                                               # seed 0 = random_device
    X, Y = oa.initial_layout(g, "u", seed=3)
    assert Y.min() >= 0 and Y.max() <= np.sqrt(2 * N)
    X, Y = oa.initial_layout(g, "r", seed=3)
    assert X.max() <= g.node_len.sum() and X.min() >= 0
    X, Y = oa.initial_layout(g, "g", seed=3)
    assert abs(X.mean()) < 5 and abs(X.std() - np.sqrt(2 * N)) < 0.1 * np.sqrt(2 * N)
    # Hilbert: deterministic; d2xy(n, d) for the first indices (hilbert.hpp:30-41)
    X, Y = oa.initial_layout(g, "h")
    assert [(X[i], Y[i]) for i in range(4)] == [(0, 0), (1, 0), (1, 1), (0, 1)]
    assert len({(x, y) for x, y in zip(X, Y)}) == 2 * N                            # a curve: no repeats


def test_lay_roundtrip_and_reference_bytes(oa, tmp_path):
    raw = open(os.path.join(GOLDEN, "DRB1-3123_unsorted.og.lay"), "rb").read()
    lay = oa.Layout.load(os.path.join(GOLDEN, "DRB1-3123_unsorted.og.lay"))
    assert lay.to_bytes() == raw              # our writer reproduces sdsl::enc_vector's bytes exactly
    rs = np.random.RandomState(0)
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 1000):
        X = rs.normal(0, 1e4, n)
        Y = rs.normal(0, 1e4, n)
        X[n // 2] = Y[n // 2]                 # equal neighbours: a zero delta (coded as 2^64)
        if n > 3:
            X[1] = X[0]
            Y[0] = X[0]
        f = tmp_path / f"r{n}.lay"
        oa.Layout(X, Y).serialize(f)
        back = oa.Layout.load(f)
        m = min(X.min(), Y.min())
        # values are stored relative to the minimum and restored by adding it back (layout.cpp:86-96)
        assert np.array_equal(back.X, (X - m) + m) and np.array_equal(back.Y, (Y - m) + m)
    from odgi_amd._lib import PgsgdError
    (tmp_path / "short.lay").write_bytes(raw[:1000])
    with pytest.raises(PgsgdError):
        oa.Layout.load(tmp_path / "short.lay")


def test_components_pack_and_tsv(oa, tmp_path):
    from odgi_amd import layout as L
    # two components: {1,2,4} and {3,5}; discovery order = lowest rank first
    gfa = tmp_path / "c.gfa"
    gfa.write_text("S\t1\tAA\nS\t2\tC\nS\t3\tGGG\nS\t4\tT\nS\t5\tA\nL\t1\t+\t2\t-\t0M\nL\t4\t+\t2\t+\t0M\nL\t5\t-\t3\t+\t0M\n"
                   "P\ta\t1+,2-,4+\t*\nP\tb\t3+,5-\t*\n")
    g = oa.Graph.from_gfa(gfa)
    comp, n = L.weak_components(g)
    assert n == 2 and list(comp) == [0, 0, 1, 0, 1]
    X = np.array([5, 6, 7, 8, -3, -2, 9, 10, -1, 0], dtype=np.float64)
    Y = np.array([1, 2, 3, 4, -9, -8, 5, 6, -7, -6], dtype=np.float64)
    Xp, Yp = X.copy(), Y.copy()
    comp, n = L.pack_components(g, Xp, Yp)
    # layout_main.cpp:407-435: x -= min_x - 1000 ; y += curr_y_offset - min_y ; offset += height + 1000
    c0 = [0, 1, 2, 3, 6, 7]
    c1 = [4, 5, 8, 9]
    assert np.allclose(Xp[c0], X[c0] - (5 - 1000)) and np.allclose(Yp[c0], Y[c0] + (1000 - 1))
    h0 = 6 - 1
    # component 1 lies entirely at negative y: max_y keeps its start value DBL_MIN (draw.hpp:37-39)
    assert np.allclose(Xp[c1], X[c1] - (-3 - 1000)) and np.allclose(Yp[c1], Y[c1] + (1000 + h0 + 1000 - (-9)))
    tsv = tmp_path / "o.tsv"
    L.write_tsv(tsv, g, comp, n, Xp, Yp)
    rows = tsv.read_text().splitlines()
    assert rows[0] == "idx\tX\tY\tcomponent"
    assert [r.split("\t")[0] for r in rows[1:]] == ["0", "1", "2", "3", "6", "7", "4", "5", "8", "9"]
    assert [r.split("\t")[3] for r in rows[1:]] == ["0"] * 6 + ["1"] * 4
    assert float(rows[1].split("\t")[1]) == Xp[0]


def test_c_abi_exports_every_declared_symbol(oa):
    from odgi_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pgsgd.h")).read()
    declared = set(re.findall(r"\b(pgsgd_[a-z0-9_]+)\s*\(", hdr))
    bound = {n for n, _, _ in _lib.SIGNATURES}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(_lib.lib, name)
    assert C.sizeof(_lib.GraphView) == 64 and C.sizeof(_lib.Stats) == 64 and C.sizeof(_lib.Params) == 136


def test_shard_flags_rule(oa):
    """pgsgd_shard_flags — what a multi-GPU driver ORs into its sessions' flags before it creates them: 128-node regions where
    256-node windows would leave a device fewer than a thousand per launch and 128-node ones leave it at least 240 (the threshold
    from which pgsgd_session_set_shard(.., -1) shards by region with the exact exchange); nothing for one device, without tiles,
    with fp32 words or Hogwild stores, when the tile shard is asked for, or when the flag is already there."""
    from odgi_amd import _lib
    f, R = _lib.lib.pgsgd_shard_flags, _lib.FLAG_REGION_128
    assert _lib.lib.pgsgd_abi_version() == 7 and R == 0x40000 and _lib.FLAG_SHARD_TILES == 0x80000
    # windows per colour: N / 512 with 256-node regions, N / 256 with 128-node ones
    assert [f(1_000_000, w, 0) for w in (1, 2, 4, 8, 16, 17)] == [0, R, R, R, R, 0]      # 3907 / 16 = 244 >= 240 > 3907 / 17
    assert [f(10_000_000, w, 0) for w in (2, 8, 19, 20)] == [0, 0, 0, R]                 # 19532 / 19 = 1028 >= 1000 > 19532 / 20
    assert f(122_880, 2, 0) == R and f(122_879, 2, 0) == R and f(122_624, 2, 0) == 0     # 480 windows of 128-node regions = 240 per device
    for flags in (_lib.FLAG_NO_TILES, _lib.FLAG_SHARD_TILES, 0x2, 0x4, R):
        assert f(1_000_000, 8, flags) == 0
    assert f(1_000_000, 8, _lib.FLAG_SYNC_DRAIN | 0x2000) == R                            # other flags do not matter
    g = oa.Graph.synthetic(3000, 4, seed=3)
    assert oa.shard_flags(g, 8) == 0 and oa.shard_flags(g, 1) == 0


def test_no_cpu_fallback_without_a_device(oa, graphs):
    """On a machine without a GPU the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from odgi_amd._lib import PgsgdError
    g = graphs("DRB1-3123")
    p = oa.LayoutParams.defaults(g)
    X, Y = oa.initial_layout(g, "h")
    X0 = X.copy()
    with pytest.raises(PgsgdError) as e:
        oa.path_linear_sgd_layout_gpu(g, p, X, Y)
    assert e.value.code == -2 and np.array_equal(X, X0)
    with pytest.raises(PgsgdError):
        oa.LayoutSession(g, p)


def test_cli_argument_handling(oa, tmp_path, capfd):
    assert oa.main_layout(["-h"]) == 0
    assert "odgi layout" in capfd.readouterr().out
    assert oa.main_layout([]) == 1
    assert oa.main_layout(["-o", str(tmp_path / "x.lay")]) == 1
    assert "Please specify an input file" in capfd.readouterr().err
    assert oa.main_layout(["-i", os.path.join(GOLDEN, "t.gfa")]) == 1
    assert "Please specify an output file" in capfd.readouterr().err
    assert oa.main_layout(["-i", os.path.join(GOLDEN, "t.gfa"), "-o", "x", "-G", "1", "-U", "1"]) == 1
    assert oa.main_layout(["--bogus"]) == 1
    bad = tmp_path / "gap.gfa"
    bad.write_text("S\t1\tA\nS\t3\tC\n")
    assert oa.main_layout(["-i", str(bad), "-o", str(tmp_path / "x.lay")]) == 1
    assert "not optimized" in capfd.readouterr().err
    assert oa.main_layout(["-i", "graph.og", "-o", "x"]) == 1
    capfd.readouterr()
    # -X FILE (a serialized XP index, xp.cpp:247-324) is refused explicitly, before anything is read: no reference-held
    # XP file exists to pin a reader of the sdsl-lite dump against
    assert oa.main_layout(["-i", os.path.join(GOLDEN, "t.gfa"), "-o", str(tmp_path / "x.lay"), "-X", str(tmp_path / "no.xp")]) == 1
    err = capfd.readouterr().err
    assert "-X/--path-index is not supported" in err and "Leave -X out" in err
    # GFAz (utils.cpp:110-121) likewise: codec and magic word live in an absent dependency; refused by name
    assert oa.main_layout(["-i", str(tmp_path / "graph.gfaz"), "-o", str(tmp_path / "x.lay")]) == 1
    assert "GFAz input is not supported" in capfd.readouterr().err
    import torch
    if not torch.cuda.is_available():  # a valid command line still ends in a loud device error
        assert oa.main_layout(["-i", os.path.join(GOLDEN, "t.gfa"), "-o", str(tmp_path / "t.lay")]) == 1
        assert "no usable HIP device" in capfd.readouterr().err


def test_session_refuses_a_graph_without_a_multi_step_path(oa):
    """path_sgd_layout.cpp:64-74: with no path of two or more steps there is no term to sample (the sampler would draw
    first steps for ever, :182-192).  The session API refuses such a graph up front, before it asks for a device."""
    from odgi_amd import _lib
    node_len = np.array([3, 5, 2], dtype=np.uint32)
    g = oa.Graph.from_arrays(node_len, np.array([0, 1, 2], dtype=np.uint64), np.array([0, 4], dtype=np.uint32))
    p = oa.LayoutParams(iter_max=3, min_term_updates=10, eta_max=1.0, space=1, n_streams=64)
    with pytest.raises(_lib.PgsgdError) as e:
        oa.LayoutSession(g, p)
    assert e.value.code == _lib.E_INVALID and "more than one step" in str(e.value)


def _build_shim_mock(tmp_path):
    import subprocess
    exe = tmp_path / "shim_mock"
    libdir = os.path.join(ROOT, "odgi_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_mock.cpp"),
                           "-o", str(exe), "-L" + libdir, "-lpgsgd", "-Wl,-rpath," + libdir])
    return exe


def test_reference_signature_shim_compiles_and_lowers(tmp_path):
    """include/pgsgd_handlegraph.hpp: path_linear_sgd_layout_gpu with the reference's signature over a
    mock PathHandleGraph; the lowering reproduces pathindex.cpp's step positions (0, 4, 6)."""
    import subprocess
    import torch
    r = subprocess.run([str(_build_shim_mock(tmp_path))], capture_output=True, text=True)
    assert "pos=0,4,6 handle_last=5" in r.stdout
    assert r.returncode == (0 if torch.cuda.is_available() else 3), r.stdout + r.stderr


def test_sort_order_by_component_then_position_and_1d_lay(oa, tmp_path):
    """path_linear_sgd_order's ordering rule (path_sgd.cpp:552-587,641-650): components ranked by the average id
    of their nodes, inside a component by position, ties by handle; and the 1D .lay of `--path-sgd-layout`."""
    from odgi_amd import sort as osort
    # two components: nodes {0, 3, 4} (ids 1,4,5: average 3.33) and {1, 2} (ids 2,3: average 2.5) -> {1,2} first
    edges = np.array([[0, 6], [6, 8], [2, 4]], dtype=np.uint64)            # handles 2*rank
    g = oa.Graph.from_arrays(np.array([5, 1, 2, 3, 4], dtype=np.uint32), np.array([0, 3, 5], dtype=np.uint64),
                             np.array([0, 6, 8, 2, 4], dtype=np.uint32), edges=edges)
    cr = osort.component_ranks(g)
    assert cr.tolist() == [1, 0, 0, 1, 1]
    X = np.array([7.0, 9.0, 9.0, 1.0, 7.0])
    assert osort.order_from_positions(X).tolist() == [3, 0, 4, 1, 2]          # position, then handle
    order = osort.order_from_positions(X, cr)
    assert order.tolist() == [1, 2, 3, 0, 4]                                  # component {1,2}, then 1.0 < 7.0 (rank 0 before 4)
    from odgi_amd._lib import lib, check
    import ctypes as C
    f = tmp_path / "sorted.lay"
    check(lib.pgsgd_sort_write_lay(C.byref(g.view), X.ctypes.data_as(C.POINTER(C.c_double)), order.ctypes.data_as(C.POINTER(C.c_uint64)),
                                   str(f).encode()), "write")
    L = oa.Layout.load(f)
    want_x = np.array([9.0, 10.0, 9.0, 11.0, 1.0, 4.0, 7.0, 12.0, 7.0, 11.0])   # (pos, pos + node length) in the new order
    assert np.array_equal(L.X, want_x) and np.array_equal(L.Y, np.zeros(10))


def test_tsv_rows_are_printf_16_significant_digits(oa, tmp_path):
    """layout.cpp:10-35: `idx X Y component`, two rows per node, rows grouped by component, numbers as the
    reference's stream with precision 16 prints them (= printf %.16g), also for awkward values."""
    import ctypes as C
    from odgi_amd._lib import lib
    n = 70_000                                              # more than one formatting chunk
    rs = np.random.RandomState(2)
    X = rs.rand(2 * n) * 3e7
    Y = (rs.rand(2 * n) - 0.5) * 1e4
    X[:12] = [0.0, 1.0, -1.0, 1e-300, 1e300, 123456789012345678.0, 0.1, 1.0 / 3.0, 2.5e-5, 1000.0, 1e16, 123456.7890123456789]
    comp = (np.arange(n) % 3).astype(np.uint32)
    f = tmp_path / "o.tsv"
    assert lib.pgsgd_write_tsv(str(f).encode(), n, comp.ctypes.data_as(C.POINTER(C.c_uint32)), 3, X.ctypes.data_as(C.POINTER(C.c_double)),
                               Y.ctypes.data_as(C.POINTER(C.c_double))) == 0
    lines = f.read_text().split("\n")
    assert lines[0] == "idx\tX\tY\tcomponent" and lines[-1] == "" and len(lines) == 2 * n + 2
    order = np.concatenate([np.where(comp == c)[0] for c in range(3)])
    want = []
    for i in order:
        for e in (2 * i, 2 * i + 1):
            want.append("%d\t%.16g\t%.16g\t%d" % (e, X[e], Y[e], comp[i]))
    assert lines[1:-1] == want


def test_gfa_and_lay_readers_survive_corruption(oa, tmp_path):
    """Random damage to a GFA or a .lay file ends in an error code or in a consistent result, never in a crash."""
    from odgi_amd._lib import PgsgdError
    rs = np.random.RandomState(7)
    raw = bytearray(open(os.path.join(GOLDEN, "t.gfa"), "rb").read() + open(os.path.join(GOLDEN, "k.gfa"), "rb").read()[:0])
    big = bytearray(open(os.path.join(GOLDEN, "DRB1-3123_unsorted.gfa"), "rb").read())
    f = tmp_path / "f.gfa"
    n_ok = n_err = 0
    for t in range(120):
        b = bytearray(big if t % 2 else raw)
        for _ in range(rs.randint(1, 8)):
            i = rs.randint(0, len(b))
            b[i] = rs.choice([9, 10, 43, 45, 44, 42, 48 + rs.randint(10), rs.randint(256)])
        if t % 7 == 0:
            b = b[: rs.randint(1, len(b))]
        f.write_bytes(bytes(b))
        try:
            g = oa.Graph.from_gfa(f, threads=3)
        except PgsgdError as e:
            assert e.code in (-5, -6, -7, -8)
            n_err += 1
            continue
        n_ok += 1
        assert int((g.step_handle >> 1).max(initial=0)) < g.n_nodes and g.path_first[-1] == g.n_steps
        lens = g.node_len[g.step_handle >> 1].astype(np.uint64)
        for p in range(g.n_paths):
            a, e = int(g.path_first[p]), int(g.path_first[p + 1])
            assert np.array_equal(g.step_pos[a:e], np.cumsum(lens[a:e]) - lens[a:e])
    assert n_ok > 0 and n_err > 0
    lay = bytearray(open(os.path.join(GOLDEN, "DRB1-3123_unsorted.og.lay"), "rb").read())
    fl = tmp_path / "f.lay"
    n_ok = n_err = 0
    for t in range(120):
        b = bytearray(lay)
        for _ in range(rs.randint(1, 6)):
            b[rs.randint(0, len(b))] = rs.randint(256)
        if t % 5 == 0:
            b = b[: rs.randint(1, len(b))]
        if t % 11 == 0:
            b[8:16] = rs.bytes(8)                      # the element count
        fl.write_bytes(bytes(b))
        try:
            L = oa.Layout.load(fl)
            n_ok += 1
            assert len(L.X) == len(L.Y)
        except PgsgdError as e:
            assert e.code in (-4, -5, -6)
            n_err += 1
    assert n_ok > 0 and n_err > 0


def test_readers_under_sanitizers(tmp_path):
    """The GFA, .og and .lay readers built with AddressSanitizer + UBSan and fed damaged files (byte flips,
    truncation, extreme values in length fields): no sanitizer report, no abnormal exit."""
    import random
    import subprocess
    csrc = os.path.join(ROOT, "odgi_amd", "csrc")
    exe = tmp_path / "fuzz_readers"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "include"), "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "fuzz_readers.cpp")] +
                          [os.path.join(csrc, f) for f in ("gfa_lower.cpp", "og_reader.cpp", "lay_io.cpp", "pgsgd_host.cpp")] + ["-lpthread"])
    rnd = random.Random(11)
    extreme = [0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFC18, 1 << 63, 1 << 32, (1 << 31) - 1, 0]
    sources = {"gfa": "DRB1-3123_unsorted.gfa", "og": "DRB1-3123_sorted.og", "lay": "DRB1-3123_unsorted.og.lay"}
    env = dict(os.environ, ASAN_OPTIONS="allocator_may_return_null=1:detect_leaks=1")
    for kind, name in sources.items():
        blob = open(os.path.join(GOLDEN, name), "rb").read()
        files = []
        for t in range(90):
            b = bytearray(blob)
            for _ in range(rnd.randint(0, 6)):
                b[rnd.randrange(len(b))] = rnd.randrange(256) if kind != "gfa" else rnd.choice([9, 10, 43, 45, 44, 42, 48 + rnd.randrange(10)])
            r = rnd.random()
            if r < 0.15:
                b = b[: rnd.randint(1, len(b))]
            elif r < 0.7 and kind != "gfa":
                i = rnd.choice([4, 8, 12, 16, 20, 24, 28, 36, 44, 52, rnd.randint(60, len(b) - 8), len(b) - rnd.randint(8, 120)])
                b[i:i + 8] = rnd.choice(extreme).to_bytes(8, "little")
            elif r < 0.5:
                i = rnd.randrange(len(b))
                b[i:i] = str(rnd.choice([0, 18446744073709551615, 99999999999999999999999, 4294967296])).encode()
            f = tmp_path / f"{kind}{t}"
            f.write_bytes(bytes(b))
            files.append(str(f))
        r = subprocess.run([str(exe), kind] + files, capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-3000:]


def test_gfa_path_lines_with_empty_tokens_placeholders_and_chunk_borders(oa, tmp_path):
    """P lines are parsed in chunks by threads: a count sweep (tokens - empty tokens - "*" placeholders), then a
    one-sweep parse with a general path for unusual tokens.  Random small graphs with empty tokens and placeholders, and
    one path long enough to be cut into several chunks, against a plain Python parse."""
    rs = np.random.RandomState(7)
    for trial in range(40):
        N, P = int(rs.randint(1, 30)), int(rs.randint(1, 5))
        lines = ["H\tVN:Z:1.0"] + [f"S\t{i + 1}\t{'A' * int(rs.randint(1, 5))}" for i in range(N)]
        want = []
        for p in range(P):
            toks, steps = [], []
            n_tok = 120000 if (trial == 0 and p == 0) else int(rs.randint(0, 12))   # 120k tokens: ~0.5 MB, three chunks
            for _ in range(n_tok):
                r = rs.rand()
                if r < 0.1:
                    toks.append("")
                elif r < 0.2:
                    toks.append("*")
                else:
                    i, o = int(rs.randint(1, N + 1)), "+-"[int(rs.randint(2))]
                    toks.append(f"{i}{o}")
                    steps.append(2 * (i - 1) + (o == "-"))
            lines.append(f"P\tp{p}\t" + ",".join(toks) + "\t*")
            want.append(steps)
        fn = tmp_path / "t.gfa"
        fn.write_text("\n".join(lines) + "\n")
        g = oa.Graph.from_gfa(str(fn), 3)
        node_len = np.array([len(l.split("\t")[2]) for l in lines[1:N + 1]])
        for p in range(P):
            a, b = int(g.path_first[p]), int(g.path_first[p + 1])
            assert list(g.step_handle[a:b]) == want[p], (trial, p)
            pos = np.concatenate([[0], np.cumsum(node_len[np.array(want[p], dtype=np.int64) >> 1])])[:-1] if want[p] else np.zeros(0)
            assert np.array_equal(g.step_pos[a:b], pos.astype(np.uint64))


def test_gfa_lowering_is_the_same_on_one_thread_and_on_many(oa, tmp_path):
    """The file is cut at newlines into one range per thread (from 4 MB per range on), S and L lines are parsed on
    slices: a 10 MB GFA with CRLF line ends in places, lowered with 1 and with 5 threads."""
    rs = np.random.RandomState(11)
    N, P = 150000, 6
    lens = rs.randint(1, 30, size=N)
    with open(tmp_path / "big.gfa", "w", newline="") as f:
        f.write("H\tVN:Z:1.0\n")
        for i in range(N):
            f.write(f"S\t{i + 1}\t{'C' * int(lens[i])}" + ("\r\n" if i % 1000 == 7 else "\n"))
        for i in range(N - 1):
            f.write(f"L\t{i + 1}\t+\t{i + 2}\t{'-' if i % 97 == 0 else '+'}\t0M\n")
        f.write("# a comment line\n")
        for p in range(P):
            ids = np.nonzero(rs.rand(N) < 0.9)[0] + 1
            f.write(f"P\tq{p}\t" + ",".join(f"{i}{'-' if i % 13 == 0 else '+'}" for i in ids) + "\t*\n")
    assert os.path.getsize(tmp_path / "big.gfa") > 9_000_000
    a = oa.Graph.from_gfa(str(tmp_path / "big.gfa"), 1)
    b = oa.Graph.from_gfa(str(tmp_path / "big.gfa"), 5)
    assert a.n_nodes == b.n_nodes == N and a.n_steps == b.n_steps
    for k in ("node_len", "path_first", "step_path", "step_handle", "step_pos"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert np.array_equal(a.node_len, lens)
    assert a.edges.shape == (N - 1, 2) and np.array_equal(a.edges, b.edges)
    assert np.array_equal(a.edges[:3], [[0, 3], [2, 4], [4, 6]])   # 1+ -> 2- (every 97th edge), 2+ -> 3+, 3+ -> 4+


def test_tile_windows_are_cut_into_parts_when_a_launch_has_few_rounds(oa):
    """A tiled launch hands its work items to the resident workgroups in rounds and ends with a tail: slots idle while the last
    items finish.  A session whose windows fill the device at least once but fewer than 24 times cuts every window's tiles
    into consecutive parts — 24 rounds, parts of at least four tiles, at most 16 per window.  Host arithmetic."""
    from odgi_amd._lib import lib
    f = lib.pgsgd_tile_parts_for
    assert f(2016, 53, 1280) == 13 and f(2016, 53, 1024) == 13       # BASELINE config 4 on MI355X (five / four workgroups per CU): by length
    assert f(20161, 53, 1024) == 2                                   # config 5: twenty rounds already
    assert f(50000, 53, 1024) == 1 and f(500, 53, 1024) == 1         # many rounds; not one full round
    assert f(2016, 7, 1024) == 1 and f(2016, 200, 1024) == 13        # short windows stay whole; long ones: 24 rounds (ceil(24 * 1024 / 2016))
    assert f(1024, 1000, 1024) == 16 and f(0, 53, 1024) == 1 and f(2016, 53, 0) == 1


def test_tile_sampler_rules_host_copies_match_the_oracle(orc):
    """The wave coin and the partner pair of the tile kernel's sampler (pgsgd_tiles.hpp: tile_coin_seed,
    tile_pair_partner), through the library's host copies, against the oracle's restatement: the same coins for any
    (seed, iteration, tile, wave, trip), fair and serially independent; the odd lane's partner is its even neighbour's
    twin in the 64-byte unit when that is a step of the path and its own draw otherwise (rounds 4-6), and lane r of a quad takes
    step lead ^ r of its first lane's 128-byte line (what sessions run since round 6)."""
    from odgi_amd import _lib
    lib = _lib.lib
    rs = np.random.RandomState(3)
    for _ in range(200):
        seed, epoch, tile, wave, trip = int(rs.randint(1, 1 << 40)), int(rs.randint(1, 31)), int(rs.randint(0, 1 << 22)), int(rs.randint(0, 4)), int(rs.randint(0, 5000))
        assert lib.pgsgd_tile_wave_coin(seed, epoch, tile, wave, trip) == orc.tile_wave_coin(seed, epoch, tile, wave, trip)
    coins = np.array([lib.pgsgd_tile_wave_coin(9399220, 7, 1234, 2, j) for j in range(20000)])
    assert abs(coins.mean() - 0.5) < 4 * 0.5 / np.sqrt(len(coins))
    assert abs(np.mean(coins[1:] == coins[:-1]) - 0.5) < 4 * 0.5 / np.sqrt(len(coins))
    # different waves, tiles and iterations have different coin streams
    other = np.array([lib.pgsgd_tile_wave_coin(9399220, 7, 1234, 3, j) for j in range(2000)])
    assert 0.4 < np.mean(other == coins[:2000]) < 0.6
    # partner pairs: path of 10 steps starting at flat step 7 (odd: its first step has no twin inside the path)
    first, cnt = 7, 10
    for lead_rank in range(cnt):
        got = lib.pgsgd_tile_pair_partner(first + lead_rank, first, cnt, 3)
        twin = ((first + lead_rank) ^ 1) - first
        assert got == (twin if 0 <= twin < cnt else 3), (lead_rank, got)
    # every step is some step's twin exactly once, but for the path's unpaired ends
    twins = [lib.pgsgd_tile_pair_partner(first + r, first, cnt, 99) for r in range(cnt)]
    assert sorted(t for t in twins if t != 99) == [r for r in range(cnt) if 0 <= ((first + r) ^ 1) - first < cnt]
    # partner quads (what sessions run): lane r of a quad takes flat step lead ^ r when that is a step of the path, lane 0 and a cut line keep the own draw
    for lead_rank in range(cnt):
        assert lib.pgsgd_tile_quad_partner(first + lead_rank, 0, first, cnt, 3) == 3
        for r in (1, 2, 3):
            got = lib.pgsgd_tile_quad_partner(first + lead_rank, r, first, cnt, 3)
            twin = ((first + lead_rank) ^ r) - first
            assert got == (twin if 0 <= twin < cnt else 3), (lead_rank, r, got)
    # for every r the map lead -> lead ^ r permutes the steps whose line lies inside the path: each is taken exactly once
    for r in (1, 2, 3):
        taken = [lib.pgsgd_tile_quad_partner(first + k, r, first, cnt, 99) for k in range(cnt)]
        inside = [k for k in range(cnt) if 0 <= ((first + k) ^ r) - first < cnt]
        assert sorted(t for t in taken if t != 99) == inside


def test_path_order_renames_nodes_along_the_paths(oa):
    """pgsgd_graph_path_order: ranks by (path-connected component, mean bp position).  A sorted pangenome whose nodes are
    renumbered at random comes back in an order that follows the paths again (hardly a step jumps over 128 ranks, where
    nearly all did), components stay apart, and a sorted graph is left (almost) as it is.  Host only."""
    from odgi_amd import _lib
    import ctypes as C
    g = oa.Graph.synthetic(50_000, 8, seed=5)
    u32p, f64 = C.POINTER(C.c_uint32), C.c_double

    def order(graph):
        new = np.zeros(graph.n_nodes, dtype=np.uint32)
        d0, d1 = f64(), f64()
        rc = _lib.lib.pgsgd_graph_path_order(C.byref(graph.view), new.ctypes.data_as(u32p), C.byref(d0), C.byref(d1))
        assert rc == 0
        return new, d0.value, d1.value

    new, d0, d1 = order(g)
    assert sorted(new.tolist()) == list(range(g.n_nodes)) and d0 < 0.01 and d1 < 0.01
    # mean path position follows the synthetic graph's own order up to local swaps (nodes no path visits are components of
    # their own and go behind the component they sat in)
    visited = np.zeros(g.n_nodes, dtype=bool)
    visited[g.step_handle >> 1] = True
    assert np.abs(new[visited].astype(np.int64) - np.arange(int(visited.sum()))).max() < 256
    perm = np.random.RandomState(3).permutation(g.n_nodes)
    new_len = np.empty_like(g.node_len)
    new_len[perm] = g.node_len
    h = g.step_handle
    gr = oa.Graph.from_arrays(new_len, g.path_first, (perm[h >> 1].astype(np.uint32) << 1) | (h & 1))
    new_r, d0, d1 = order(gr)
    assert d0 > 0.9 and d1 < 0.01
    # the recovered order is the sorted graph's (up to ties of the mean position, which go by the names): new_r[perm[i]] ~ new[i]
    assert np.abs(new_r[perm][visited].astype(np.int64) - new[visited].astype(np.int64)).max() <= 8
    # two components (two copies of the graph side by side, node ranks interleaved): each keeps to itself
    n = g.n_nodes
    len2 = np.repeat(g.node_len, 2)
    first2 = np.concatenate([g.path_first, g.path_first[1:] + g.n_steps])
    h2 = np.concatenate([((h >> 1) * 2 << 1) | (h & 1), (((h >> 1) * 2 + 1) << 1) | (h & 1)]).astype(np.uint32)
    g2 = oa.Graph.from_arrays(len2, first2, h2)
    new2, d0, d1 = order(g2)
    comp_a, comp_b = new2[0::2][visited], new2[1::2][visited]   # (the nodes no path visits are components of their own, at the end)
    assert comp_a.max() < comp_b.min() or comp_b.max() < comp_a.min()
    assert d1 < 0.01


def test_tile_windows_cut_into_parts_keep_order_and_dependencies(oa):
    """The cut of a launch's work items (pgsgd_tile_split_items = what a session launches) against the independent restatement
    in tests/pyref.py, and its invariants: every part waits for the part before it of the same window, which sits earlier in
    the queue and ends where this one begins; the parts of a window cover its tiles exactly once, in order; window-less items
    stay whole at the end."""
    import ctypes as C
    from odgi_amd._lib import lib
    import pyref
    g = oa.Graph.synthetic(3000, 4, seed=3)
    tiles, items = pyref.build_tiles_py(g.path_first, g.step_handle, 64, 56)
    u32p = C.POINTER(C.c_uint32)
    ptr = lambda a: a.ctypes.data_as(u32p)
    for colour, (lo, hi) in enumerate(((0, items["n_first"]), (items["n_first"], len(items["local"])))):
        tb, te, w0 = (np.ascontiguousarray(items[k][lo:hi]) for k in ("tile_begin", "tile_end", "win0"))
        n_windowless = int((items["local"][lo:hi] == 0).sum())
        for k in (1, 2, 3, 7, 64):
            cnt = lib.pgsgd_tile_split_items(ptr(tb), ptr(te), ptr(w0), len(tb), n_windowless, k, None, None, None, None, 0)
            ob, oe, ow, of = (np.zeros(cnt, dtype=np.uint32) for _ in range(4))
            assert lib.pgsgd_tile_split_items(ptr(tb), ptr(te), ptr(w0), len(tb), n_windowless, k, ptr(ob), ptr(oe), ptr(ow), ptr(of), cnt) == cnt
            _, cut = pyref.build_tiles_py(g.path_first, g.step_handle, 64, 56, split=k)
            clo, chi = (0, cut["n_first"]) if colour == 0 else (cut["n_first"], len(cut["local"]))
            assert np.array_equal(ob, cut["tile_begin"][clo:chi]) and np.array_equal(oe, cut["tile_end"][clo:chi]) and np.array_equal(ow, cut["win0"][clo:chi])
            assert np.array_equal(of & 1, cut["local"][clo:chi])
            dep, has_next = (of >> 2).astype(np.int64) - 1, (of >> 1) & 1
            waited_for = np.zeros(cnt, dtype=bool)
            for i in range(cnt):
                if dep[i] >= 0:
                    assert dep[i] < i and ow[dep[i]] == ow[i] and oe[dep[i]] == ob[i] and has_next[dep[i]]
                    assert not waited_for[dep[i]]
                    waited_for[dep[i]] = True
            assert np.array_equal(waited_for, has_next.astype(bool))
            assert np.all(oe > ob) and (k > 1 or cnt == len(tb))
            covered = np.zeros(int(te.max()) + 1, dtype=np.int32)
            for b, e in zip(ob, oe):
                covered[b:e] += 1
            want = np.zeros_like(covered)
            for b, e in zip(tb, te):
                want[b:e] += 1
            assert np.array_equal(covered, want)
    assert lib.pgsgd_tile_split_items(None, None, None, 0, 0, 2, None, None, None, None, 0) < 0


def test_views_without_step_positions_and_the_exact_evaluator(oa, orc, tmp_path):
    """step_path and step_pos of a view may be NULL (include/pgsgd.h): both follow from path_first, step_handle and node_len.  A graph
    loaded without them (PGSGD_LOAD_NO_STEP_INDEX, what `odgi layout` does) or stripped of them gives the same defaults, the same
    node order by path position and the same quality figures as the full view; and the product's exact near-pair evaluator
    (pgsgd_path_stress_near: unordered pairs) equals the oracle's (orc_path_stress_near: the sampler's draws) — two formulations of
    one sum — on layouts of two graphs.  Host only."""
    import ctypes as C
    from odgi_amd import _lib
    gfa = os.path.join(GOLDEN, "DRB1-3123.gfa")
    g, lean = oa.Graph.from_gfa(gfa, threads=3), oa.Graph.load_lean(gfa, threads=3)
    assert lean.step_pos is None and lean.step_path is None and g.step_pos is not None
    for f in ("node_len", "path_first", "step_handle"):
        assert np.array_equal(getattr(g, f), getattr(lean, f)), f
    stripped = oa.Graph.synthetic(20_000, 6, seed=9)
    full = oa.Graph.synthetic(20_000, 6, seed=9)
    stripped.drop_step_index()
    assert stripped.step_pos is None and stripped.step_path is None
    for a, b in ((g, lean), (full, stripped)):
        pa, pb = oa.LayoutParams.defaults(a), oa.LayoutParams.defaults(b)
        assert (pa.iter_max, pa.min_term_updates, pa.space, pa.eta_max) == (pb.iter_max, pb.min_term_updates, pb.space, pb.eta_max)
        X0, Y0 = oa.initial_layout(a, "d", seed=4)
        rs = np.random.RandomState(2)
        X, Y = X0.astype(np.float64) + 3 * rs.randn(len(X0)), Y0.astype(np.float64) + 3 * rs.randn(len(X0))
        assert oa.path_stress(a, X, Y, 200_000, seed=3) == oa.path_stress(b, X, Y, 200_000, seed=3)
        assert oa.path_distance(a, X, Y) == oa.path_distance(b, X, Y)
        na, nb = oa.path_stress_near(a, X, Y, zmax=4, threads=2, mod_step=7, mod_rank=5), oa.path_stress_near(b, X, Y, zmax=4, threads=3)
        assert np.allclose(na["num"], nb["num"], rtol=1e-12) and abs(na["near"] - nb["near"]) <= 1e-12 * na["near"]
        assert np.isclose(na["hist_step"].sum(), na["num"].sum(), rtol=1e-12) and np.isclose(na["hist_rank"].sum(), na["num"].sum(), rtol=1e-12)
        no = orc.path_stress_near(orc.Graph.from_product(a), X, Y, zmax=4, threads=2)
        assert np.allclose(na["num"], no["num"], rtol=1e-11) and np.allclose(na["mass"], no["mass"], rtol=1e-11) and abs(na["zero_mass"] - no["zero_mass"]) < 1e-13
        # and it is a part of what the sampled evaluator estimates, growing with the reach (on a CONVERGED layout four steps of
        # reach carry 99.9 % of the figure; on this perturbed initial layout the pairs further apart still carry a share)
        n64 = oa.path_stress_near(a, X, Y, zmax=64, threads=2)["near"]
        assert na["near"] < n64 < 1.15 * oa.path_stress(a, X, Y, 4_000_000, seed=5)
        order = []
        for gr in (a, b):
            new = np.zeros(gr.n_nodes, dtype=np.uint32)
            d0, d1 = C.c_double(), C.c_double()
            assert _lib.lib.pgsgd_graph_path_order(C.byref(gr.view), new.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(d0), C.byref(d1)) == 0
            order.append((new, d0.value, d1.value))
        assert np.array_equal(order[0][0], order[1][0]) and order[0][1:] == order[1][1:]
