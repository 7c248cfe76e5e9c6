"""1D path-guided SGD of `odgi sort -Y` (reference src/algorithms/path_sgd.hpp / path_sgd.cpp) over the C ABI.

`path_linear_sgd` returns the 1D node positions, `path_linear_sgd_order` the node order derived from
them, like the reference functions of the same names; the work runs on the GPU (libpgsgd.so)."""
import ctypes as C
import dataclasses

import numpy as np

from . import _lib
from ._lib import lib, check
from .graph import Graph
from .layout import LayoutParams

_F64P = C.POINTER(C.c_double)


def sort_params_defaults(graph: Graph, **overrides) -> LayoutParams:
    """Defaults of `odgi sort -Y` (src/subcommand/sort_main.cpp:313-320,378-414)."""
    p = _lib.Params()
    check(lib.pgsgd_sort_params_defaults(C.byref(graph.view), C.byref(p)), "sort_params_defaults")
    out = LayoutParams(iter_max=p.iter_max, iter_with_max_learning_rate=p.iter_with_max_learning_rate,
                       min_term_updates=p.min_term_updates, delta=p.delta, eps=p.eps, eta_max=p.eta_max, theta=p.theta,
                       space=p.space, space_max=p.space_max, space_quantization_step=p.space_quantization_step,
                       cooling_start=p.cooling_start, seed=p.seed)
    return dataclasses.replace(out, **overrides)


def sort_initial(graph: Graph):
    X = np.zeros(graph.n_nodes, dtype=np.float64)
    check(lib.pgsgd_sort_initial(C.byref(graph.view), X.ctypes.data_as(_F64P)), "sort_initial")
    return X


def path_linear_sgd(graph: Graph, params: LayoutParams, X=None, target_nodes=None):
    """1D positions of the nodes (path_sgd.cpp:12-500). Returns (X, stats).  `target_nodes` (bool per node,
    the reference's target sorting): nodes that keep their position."""
    X = sort_initial(graph) if X is None else np.ascontiguousarray(X, dtype=np.float64).copy()
    st = _lib.Stats()
    p = params.to_c()
    if target_nodes is None:
        tp = None
    else:
        tn = np.ascontiguousarray(target_nodes, dtype=np.uint8)
        if len(tn) != graph.n_nodes:
            raise ValueError("target_nodes needs one entry per node")
        tp = tn.ctypes.data_as(C.POINTER(C.c_uint8))
    check(lib.pgsgd_sort_run_targets(C.byref(graph.view), C.byref(p), tp, X.ctypes.data_as(_F64P), C.byref(st)), "sort_run")
    return X, {f: getattr(st, f) for f, _ in _lib.Stats._fields_}


def component_ranks(graph: Graph):
    """Weakly connected components of the graph ranked by the average id of their nodes (path_sgd.cpp:552-587)."""
    edges = np.ascontiguousarray(graph.edges, dtype=np.uint64).reshape(-1)
    out = np.zeros(graph.n_nodes, dtype=np.uint32)
    check(lib.pgsgd_sort_component_ranks(graph.n_nodes, edges.ctypes.data_as(C.POINTER(C.c_uint64)), len(edges) // 2,
                                         out.ctypes.data_as(C.POINTER(C.c_uint32))), "sort_component_ranks")
    return out


def order_from_positions(X, comp_ranks=None):
    """Node ranks by (component rank, position, handle) (path_sgd.cpp:641-650)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    order = np.zeros(len(X), dtype=np.uint64)
    cr = None if comp_ranks is None else np.ascontiguousarray(comp_ranks, dtype=np.uint32)
    check(lib.pgsgd_sort_order_components(len(X), X.ctypes.data_as(_F64P), None if cr is None else cr.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          order.ctypes.data_as(C.POINTER(C.c_uint64))), "sort_order")
    return order


def path_linear_sgd_order(graph: Graph, params: LayoutParams, target_nodes=None, layout_out=None):
    """Node ranks in their new order (path_sgd.cpp:503-686, without the graph rewrite): by weak component
    (ranked by average node id), then position, then handle.  `layout_out`: also write the sorted nodes' 1D
    layout as a .lay file (`odgi sort --path-sgd-layout`, :651-672).  Returns (order, X, stats)."""
    X, st = path_linear_sgd(graph, params, target_nodes=target_nodes)
    order = order_from_positions(X, component_ranks(graph) if len(graph.edges) else None)
    if layout_out is not None:
        check(lib.pgsgd_sort_write_lay(C.byref(graph.view), X.ctypes.data_as(_F64P), order.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       str(layout_out).encode()), "sort_write_lay")
    return order, X, st


def sort_stress(graph: Graph, X, n_pairs=1_000_000, seed=0x5eed):
    X = np.ascontiguousarray(X, dtype=np.float64)
    s = C.c_double()
    check(lib.pgsgd_sort_stress(C.byref(graph.view), X.ctypes.data_as(_F64P), n_pairs, seed, C.byref(s)), "sort_stress")
    return s.value


def trace_terms_1d(graph: Graph, params: LayoutParams, cooling, terms_per_stream):
    """Parity hook: uint64 [terms_per_stream, n_streams, 2] = (ka, kb) of fresh streams."""
    if not params.n_streams:
        raise ValueError("trace_terms_1d needs an explicit n_streams")
    out = np.zeros((terms_per_stream, params.n_streams, 2), dtype=np.uint64)
    n = C.c_uint32()
    p = params.to_c()
    check(lib.pgsgd_sort_trace_terms(C.byref(graph.view), C.byref(p), 1 if cooling else 0, terms_per_stream,
                                     out.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(n)), "sort_trace_terms")
    assert n.value == params.n_streams
    return out
