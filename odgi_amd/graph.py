"""Graph — the lowered `graph_t` + `xp::XP` the layout path reads (include/pgsgd.h: pgsgd_graph_view)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Graph:
    """Flat path-step index: node_len[N], path_first[P+1], step_path/step_handle/step_pos[S].

    Build with Graph.from_gfa / Graph.from_og / Graph.load (reference: gfa_to_handle or
    graph_t::deserialize, + XP::from_handle_graph), Graph.synthetic
    (BASELINE configs 4/5) or Graph.from_arrays (any caller that already walked its own graph_t).
    """

    def __init__(self):
        self._handle = None
        self._arrays = None
        self.view = _lib.GraphView()
        self.edges = np.zeros((0, 2), dtype=np.uint64)
        self.path_names = []

    # -- constructors ------------------------------------------------------------------------
    @classmethod
    def from_gfa(cls, path, threads=1):
        g = cls()
        h = C.c_void_p()
        check(lib.pgsgd_graph_from_gfa(str(path).encode(), int(threads), C.byref(h)), f"from_gfa({path})")
        g._adopt(h)
        return g

    @classmethod
    def from_og(cls, path, threads=1):
        """odgi's native graph file (graph_t::serialize, src/odgi.cpp:1632-1685); must be optimized."""
        g = cls()
        h = C.c_void_p()
        check(lib.pgsgd_graph_from_og(str(path).encode(), int(threads), C.byref(h)), f"from_og({path})")
        g._adopt(h)
        return g

    @classmethod
    def load(cls, path, threads=1):
        """The reference's input dispatch (src/utils.cpp:110-134): *gfa -> GFA v1, anything else -> .og."""
        g = cls()
        h = C.c_void_p()
        check(lib.pgsgd_graph_load(str(path).encode(), int(threads), C.byref(h)), f"load({path})")
        g._adopt(h)
        return g

    @classmethod
    def load_lean(cls, path, threads=1):
        """Graph.load without step_path / step_pos (PGSGD_LOAD_NO_STEP_INDEX): the session builds the positions on the device."""
        g = cls()
        h = C.c_void_p()
        check(lib.pgsgd_graph_load_flags(str(path).encode(), int(threads), 1, C.byref(h)), f"load_lean({path})")
        g._adopt(h)
        return g

    def drop_step_index(self):
        """Free step_path and step_pos (they follow from path_first, step_handle and node_len); the view carries NULL for them:
        sessions upload 4 bytes per step instead of 12 and build the positions on the device, host metrics walk the paths."""
        if self._handle:
            check(lib.pgsgd_graph_drop_step_index(self._handle), "drop_step_index")
            check(lib.pgsgd_graph_get_view(self._handle, C.byref(self.view)), "get_view")
        else:
            self.view.step_path = None
            self.view.step_pos = None
        return self

    @classmethod
    def synthetic(cls, n_nodes, n_paths, seed=42):
        g = cls()
        h = C.c_void_p()
        check(lib.pgsgd_graph_synthetic(int(n_nodes), int(n_paths), int(seed), C.byref(h)), "synthetic")
        g._adopt(h)
        return g

    @classmethod
    def from_arrays(cls, node_len, path_first, step_handle, step_pos=None, step_path=None, edges=None, path_names=None):
        g = cls()
        # copies: the arguments may be views into another Graph's native arrays, which die with that Graph
        node_len = np.array(node_len, dtype=np.uint32, order="C")
        path_first = np.array(path_first, dtype=np.uint64, order="C")
        step_handle = np.array(step_handle, dtype=np.uint32, order="C")
        n_paths = len(path_first) - 1
        if step_path is None:
            step_path = np.repeat(np.arange(n_paths, dtype=np.uint32), np.diff(path_first).astype(np.int64))
        if step_pos is None:  # xp.cpp:607-617: bp offset of each step start within its path
            lens = node_len[step_handle >> 1].astype(np.uint64)
            csum = np.cumsum(lens) - lens
            first = csum[np.minimum(path_first[:-1], max(len(csum) - 1, 0)).astype(np.int64)] if len(csum) else csum
            step_pos = csum - np.repeat(first, np.diff(path_first).astype(np.int64)) if len(csum) else csum
        step_path = np.array(step_path, dtype=np.uint32, order="C")
        step_pos = np.array(step_pos, dtype=np.uint64, order="C")
        g._arrays = (node_len, path_first, step_path, step_handle, step_pos)
        v = g.view
        v.n_nodes, v.n_steps, v.n_paths = len(node_len), len(step_handle), n_paths
        v.node_len = _ptr(node_len, C.c_uint32)
        v.path_first = _ptr(path_first, C.c_uint64)
        v.step_path = _ptr(step_path, C.c_uint32)
        v.step_handle = _ptr(step_handle, C.c_uint32)
        v.step_pos = _ptr(step_pos, C.c_uint64)
        if edges is not None:
            g.edges = np.ascontiguousarray(edges, dtype=np.uint64).reshape(-1, 2)
        g.path_names = list(path_names) if path_names is not None else [f"p{i}" for i in range(n_paths)]
        return g

    def _adopt(self, h):
        self._handle = h
        check(lib.pgsgd_graph_get_view(h, C.byref(self.view)), "get_view")
        ne = lib.pgsgd_graph_edge_count(h)
        if ne:
            self.edges = np.ctypeslib.as_array(lib.pgsgd_graph_edges(h), shape=(ne, 2)).copy()
        self.path_names = [lib.pgsgd_graph_path_name(h, i).decode() for i in range(self.view.n_paths)]

    def __del__(self):
        if getattr(self, "_handle", None):
            lib.pgsgd_graph_free(self._handle)
            self._handle = None

    # -- accessors ---------------------------------------------------------------------------
    n_nodes = property(lambda s: int(s.view.n_nodes))
    n_steps = property(lambda s: int(s.view.n_steps))
    n_paths = property(lambda s: int(s.view.n_paths))

    def _arr(self, ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,))

    node_len = property(lambda s: s._arr(s.view.node_len, s.n_nodes, np.uint32))
    path_first = property(lambda s: s._arr(s.view.path_first, s.n_paths + 1, np.uint64))
    step_path = property(lambda s: s._arr(s.view.step_path, s.n_steps, np.uint32) if s.view.step_path else None)
    step_handle = property(lambda s: s._arr(s.view.step_handle, s.n_steps, np.uint32))
    step_pos = property(lambda s: s._arr(s.view.step_pos, s.n_steps, np.uint64) if s.view.step_pos else None)

    def path_step_counts(self):
        return np.diff(self.path_first.astype(np.int64))

    def max_path_steps(self):
        c = self.path_step_counts()
        return int(c.max()) if len(c) else 0
