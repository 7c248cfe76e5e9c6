"""The .og reader (odgi's native graph file as input of `odgi layout`), CPU only.

Pins: the reference's own fixture test/DRB1-3123_sorted.og (copied to tests/golden/) is read to the last
byte; its header is the known answer of SURVEY 8f-3 (3214 nodes, 4380 edges, 12 paths); the same graph
exists as test/DRB1-3123_unsorted.gfa, so path names, step counts, the bp offset of every step, the
spelled sequence of every path and the number of distinct edges must agree across the two formats; an
independent pure-Python decoder (tests/pyref.py) must agree with the C++ reader element by element."""
import hashlib
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
import pyref

OG = os.path.join(GOLDEN, "DRB1-3123_sorted.og")
GFA = os.path.join(GOLDEN, "DRB1-3123_unsorted.gfa")
_COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def _spell_gfa(path):
    seq, out = {}, {}
    for line in open(path, "rb"):
        t = line.rstrip(b"\n").split(b"\t")
        if t[0] == b"S":
            seq[t[1]] = t[2]
        elif t[0] == b"P":
            out[t[1].decode()] = b"".join(seq[x[:-1]].translate(_COMP)[::-1] if x[-1:] == b"-" else seq[x[:-1]] for x in t[2].split(b","))
    return out


def test_og_fixture_header_and_cross_format_pins(oa):
    g = oa.Graph.from_og(OG)
    h = oa.Graph.from_gfa(GFA)
    assert (g.n_nodes, g.n_steps, g.n_paths, len(g.edges)) == (3214, 21882, 12, 4380)   # .og header: SURVEY 8f-3
    assert len(h.edges) == 4380            # 6243 L lines of the GFA name 4380 distinct edges (graph_t::create_edge)
    assert g.path_names == h.path_names
    assert np.array_equal(g.path_first, h.path_first)
    assert np.array_equal(g.step_pos, h.step_pos)           # node order differs, the paths' bp offsets cannot
    assert np.array_equal(g.step_path, h.step_path)
    assert sorted(g.node_len.tolist()) == sorted(h.node_len.tolist())
    assert int(g.node_len[g.step_handle >> 1].sum()) == 163416
    # Graph.load follows the reference's dispatch by file name
    assert oa.Graph.load(OG).n_steps == 21882 and oa.Graph.load(GFA).n_steps == 21882


def test_og_reader_matches_independent_decoder_and_spells_the_gfa_paths(oa):
    ref = pyref.decode_og(OG)
    g = oa.Graph.from_og(OG, threads=4)
    assert ref["header"] == [3214, 1, 3214, 4380, 12, 12, 0]
    assert g.node_len.tolist() == ref["node_len"]
    assert [tuple(e) for e in g.edges.tolist()] == ref["edges"]
    pf = g.path_first.astype(np.int64)
    spelled = _spell_gfa(GFA)
    for j, (name, handles) in enumerate(ref["paths"]):
        assert g.path_names[j] == name
        assert g.step_handle[pf[j]:pf[j + 1]].tolist() == handles
        s = b"".join(ref["node_seq"][hd >> 1].translate(_COMP)[::-1] if hd & 1 else ref["node_seq"][hd >> 1] for hd in handles)
        assert hashlib.md5(s).hexdigest() == hashlib.md5(spelled[name]).hexdigest()


def test_og_reader_rejects_bad_input(oa, tmp_path):
    from odgi_amd import _lib
    raw = bytearray(open(OG, "rb").read())

    def code_of(data, name):
        f = tmp_path / name
        f.write_bytes(bytes(data))
        with pytest.raises(_lib.PgsgdError) as e:
            oa.Graph.from_og(f)
        return e.value.code

    assert code_of(b"\x00\x01\x02\x03" + bytes(raw[4:]), "magic.og") == _lib.E_FORMAT
    assert code_of(raw[:40], "hdr.og") == _lib.E_FORMAT
    assert code_of(raw[:len(raw) // 2], "half.og") == _lib.E_FORMAT
    assert code_of(raw[:-3], "tail.og") == _lib.E_FORMAT
    notopt = bytearray(raw)
    struct.pack_into("<Q", notopt, 4 + 8, 2)   # min_node_id = 2
    assert code_of(notopt, "notopt.og") == _lib.E_NOTOPTIMIZED
    # a path whose first step points at a node that does not exist
    broken = bytearray(raw)
    ref_len = len(raw)
    # last path record: ... u64 length, first (2 x u64), last (2 x u64), u64 name length, name
    name_len = len(b"gi|157702218:147985-163915")
    first_off = ref_len - name_len - 8 - 16 - 16
    struct.pack_into("<Q", broken, first_off, 2 * 999999)
    assert code_of(broken, "ptr.og") == _lib.E_FORMAT
    with pytest.raises(_lib.PgsgdError):
        oa.Graph.from_og(tmp_path / "missing.og")


def test_og_reader_survives_corruption(oa, tmp_path):
    """Random byte damage must end in an error code or in a graph that is consistent, never in a crash."""
    from odgi_amd import _lib
    raw = bytearray(open(OG, "rb").read())
    rs = np.random.RandomState(1)
    f = tmp_path / "f.og"
    n_ok = n_err = 0
    for t in range(150):
        b = bytearray(raw)
        for _ in range(rs.randint(1, 6)):
            b[rs.randint(4, len(b))] = rs.randint(256)
        if t % 5 == 0:
            i = rs.randint(60, len(b) - 8)
            b[i:i + 8] = rs.bytes(8)
        f.write_bytes(bytes(b))
        try:
            g = oa.Graph.from_og(f)
        except _lib.PgsgdError as e:
            assert e.code in (_lib.E_FORMAT, _lib.E_NOTOPTIMIZED, _lib.E_UNSUPPORTED, _lib.E_NOMEM)
            n_err += 1
            continue
        n_ok += 1
        assert g.n_nodes == 3214 and int((g.step_handle >> 1).max()) < g.n_nodes
        assert g.path_first[-1] == g.n_steps and np.all(np.diff(g.path_first.astype(np.int64)) >= 0)
    assert n_ok > 0 and n_err > 0
