import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # native pieces are built in-tree; build them once if a fresh checkout has none
    lib = os.path.join(ROOT, "odgi_amd", "lib", "libpgsgd.so")
    orc = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oa():
    import odgi_amd
    return odgi_amd


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


FIXTURES = {
    # name: (N, S, P, max steps/path)   — SURVEY.md section 4 fixture table
    "DRB1-3123": (4955, 35059, 12, 3100),
    "LPA": (3751, 202806, 13, 21901),
    "chr6.C4": (1748, 171208, 90, 2932),
    "DRB1-3123_unsorted": (3214, 21882, 12, 2007),
}


@pytest.fixture(scope="session")
def graphs(oa):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = oa.Graph.from_gfa(os.path.join(GOLDEN, name + ".gfa"), threads=4)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def ographs(graphs, orc):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = orc.Graph.from_product(graphs(name))
        return cache[name]
    return get


def parse_gfa_py(path):
    """Independent (pure Python) lowering of a GFA, used to cross-check the C++ loader."""
    node_len, paths, edges = {}, [], []
    seen_edges = set()
    with open(path) as f:
        for line in f:
            t = line.rstrip("\n").split("\t")
            if t[0] == "S":
                node_len[int(t[1])] = len(t[2])
            elif t[0] == "L":
                a, b = 2 * (int(t[1]) - 1) + (t[2] == "-"), 2 * (int(t[3]) - 1) + (t[4] == "-")
                key = min((a, b), (b ^ 1, a ^ 1))   # a -> b is the same edge as flip(b) -> flip(a) (odgi.cpp:611-631)
                if key not in seen_edges:
                    seen_edges.add(key)
                    edges.append((a, b))
            elif t[0] == "P":
                steps = [(int(s[:-1]), s[-1] == "-") for s in t[2].split(",") if s and s != "*"]
                paths.append((t[1], steps))
    n = len(node_len)
    nl = np.array([node_len[i + 1] for i in range(n)], dtype=np.uint32)
    path_first, step_path, step_handle, step_pos = [0], [], [], []
    for pi, (_, steps) in enumerate(paths):
        pos = 0
        for nid, rev in steps:
            step_path.append(pi)
            step_handle.append(2 * (nid - 1) + int(rev))
            step_pos.append(pos)
            pos += node_len[nid]
        path_first.append(len(step_handle))
    return dict(node_len=nl, path_first=np.array(path_first, dtype=np.uint64),
                step_path=np.array(step_path, dtype=np.uint32), step_handle=np.array(step_handle, dtype=np.uint32),
                step_pos=np.array(step_pos, dtype=np.uint64), edges=np.array(edges, dtype=np.uint64).reshape(-1, 2),
                names=[p[0] for p in paths])
