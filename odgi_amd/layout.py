"""Host-side mirror of the reference's layout interface over the C ABI (include/pgsgd.h).

Names and argument meaning follow src/algorithms/path_sgd_layout.hpp:32-80 and
src/algorithms/layout.hpp:24-40; the work happens in libpgsgd.so on the GPU.
"""
import ctypes as C
import dataclasses
import math
import sys

import numpy as np

from . import _lib
from ._lib import lib, check
from .graph import Graph

_F32P = C.POINTER(C.c_float)
_F64P = C.POINTER(C.c_double)


def _f64(a):
    return a.ctypes.data_as(_F64P)


@dataclasses.dataclass
class LayoutParams:
    """Argument list of path_linear_sgd_layout[_gpu] (path_sgd_layout.hpp:59-80) plus GPU knobs."""
    iter_max: int = 30
    iter_with_max_learning_rate: int = 0
    min_term_updates: int = 0
    delta: float = 0.0
    eps: float = 0.01
    eta_max: float = 0.0
    theta: float = 0.99
    space: int = 0
    space_max: int = 1000
    space_quantization_step: int = 100
    cooling_start: float = 0.5
    seed: int = _lib.DEFAULT_SEED
    n_streams: int = 0
    stream_offset: int = 0
    device: int = -1
    snapshot_prefix: str = ""
    progress: bool = False
    flags: int = 0
    terms_per_anchor: int = 1
    n_devices: int = 1   # GPUs of one node for the one-call run (C++ threads + RCCL inside the library)

    @classmethod
    def defaults(cls, graph: Graph, **overrides):
        """layout_main.cpp:198-204,251-266 for a graph whose paths are all used."""
        p = _lib.Params()
        check(lib.pgsgd_params_defaults(C.byref(graph.view), C.byref(p)), "params_defaults")
        out = cls(iter_max=p.iter_max, iter_with_max_learning_rate=p.iter_with_max_learning_rate,
                  min_term_updates=p.min_term_updates, delta=p.delta, eps=p.eps, eta_max=p.eta_max,
                  theta=p.theta, space=p.space, space_max=p.space_max,
                  space_quantization_step=p.space_quantization_step, cooling_start=p.cooling_start,
                  seed=p.seed)
        return dataclasses.replace(out, **overrides)

    def first_cooling_iteration(self):
        return int(math.floor(self.cooling_start * float(self.iter_max)))  # path_sgd_layout.cpp:39

    def to_c(self):
        p = _lib.Params()
        for f in ("iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps", "eta_max",
                  "theta", "space", "space_max", "space_quantization_step", "cooling_start", "seed",
                  "n_streams", "stream_offset", "device", "flags", "terms_per_anchor", "n_devices"):
            setattr(p, f, getattr(self, f))
        p.snapshot = 1 if self.snapshot_prefix else 0
        self._prefix_bytes = self.snapshot_prefix.encode() if self.snapshot_prefix else None
        p.snapshot_prefix = self._prefix_bytes
        p.progress = 1 if self.progress else 0
        return p


def path_linear_sgd_layout_schedule(params: LayoutParams):
    """etas[0..iter_max] (path_sgd_layout.cpp:433-468)."""
    etas = np.zeros(params.iter_max + 1, dtype=np.float64)
    p = params.to_c()
    n = lib.pgsgd_schedule(C.byref(p), _f64(etas), len(etas))
    if n < 0:
        check(int(n), "schedule")
    return etas


def zeta_table(theta, space, space_max, space_quantization_step):
    """Zipf zeta cache (path_sgd_layout.cpp:86-97)."""
    n = lib.pgsgd_zeta_table_size(space, space_max, space_quantization_step)
    z = np.zeros(n, dtype=np.float64)
    check(lib.pgsgd_zeta_table(theta, space, space_max, space_quantization_step, _f64(z), n), "zeta_table")
    return z


def initial_layout(graph: Graph, mode="d", seed=0):
    """Initial X,Y [2N] float64 (layout_main.cpp:268-330); seed 0 = std::random_device as upstream."""
    X = np.zeros(2 * graph.n_nodes, dtype=np.float64)
    Y = np.zeros(2 * graph.n_nodes, dtype=np.float64)
    check(lib.pgsgd_init_layout(C.byref(graph.view), mode.encode()[:1], int(seed), _f64(X), _f64(Y)), "init_layout")
    return X, Y


def path_linear_sgd_layout_gpu(graph: Graph, params: LayoutParams, X, Y):
    """The `--gpu` entry (path_sgd_layout.hpp:59-80): X,Y [2N] pre-initialised, updated IN PLACE.

    float64 arrays (the reference's vector<atomic<double>>) come back at the full resolution of the device's
    fixed-point coordinates (pgsgd_layout_run_f64); float32 arrays are used as they are (pgsgd_layout_run).
    Returns the run statistics.
    """
    if X.shape != (2 * graph.n_nodes,) or Y.shape != X.shape:
        raise ValueError("X and Y must have 2*node_count entries")
    st = _lib.Stats()
    p = params.to_c()
    if X.dtype == np.float32 and Y.dtype == np.float32:
        Xf, Yf = np.ascontiguousarray(X), np.ascontiguousarray(Y)
        check(lib.pgsgd_layout_run(C.byref(graph.view), C.byref(p), Xf.ctypes.data_as(_F32P), Yf.ctypes.data_as(_F32P),
                                   C.byref(st)), "layout_run")
    else:
        Xf, Yf = np.ascontiguousarray(X, dtype=np.float64), np.ascontiguousarray(Y, dtype=np.float64)
        f64p = C.POINTER(C.c_double)
        check(lib.pgsgd_layout_run_f64(C.byref(graph.view), C.byref(p), Xf.ctypes.data_as(f64p), Yf.ctypes.data_as(f64p),
                                       C.byref(st)), "layout_run_f64")
    X[...] = Xf
    Y[...] = Yf
    return {f: getattr(st, f) for f, _ in _lib.Stats._fields_}


class LayoutSession:
    """One GPU's resident graph + coordinates; runs the SGD one learning-rate step at a time."""

    def __init__(self, graph: Graph, params: LayoutParams):
        self.graph = graph
        self.params = params
        self._h = C.c_void_p()
        self._cparams = params.to_c()
        check(lib.pgsgd_session_create(C.byref(graph.view), C.byref(self._cparams), C.byref(self._h)), "session_create")

    def close(self):
        if self._h:
            lib.pgsgd_session_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def n_streams(self):
        return int(lib.pgsgd_session_n_streams(self._h))

    def upload(self, X, Y):
        Xf = np.ascontiguousarray(X, dtype=np.float32)
        Yf = np.ascontiguousarray(Y, dtype=np.float32)
        check(lib.pgsgd_session_upload_coords(self._h, Xf.ctypes.data_as(_F32P), Yf.ctypes.data_as(_F32P)), "upload")

    def flush(self):
        """Deliver the far pulls of the last tile launch (pgsgd_session_flush; a no-op for the per-lane kernel)."""
        check(lib.pgsgd_session_flush(self._h), "flush")

    def download(self, flush=True):
        """Coordinates as fp32 [2N] X, Y (pgsgd_session_download_coords: the far pulls of the last tile launch delivered).
        flush=False: as a snapshot between iterations sees them (pgsgd_session_peek_coords), without those pulls."""
        n = 2 * self.graph.n_nodes
        X = np.zeros(n, dtype=np.float32)
        Y = np.zeros(n, dtype=np.float32)
        f = lib.pgsgd_session_download_coords if flush else lib.pgsgd_session_peek_coords
        check(f(self._h, X.ctypes.data_as(_F32P), Y.ctypes.data_as(_F32P)), "download")
        return X, Y

    def download_f64(self, flush=True):
        """Coordinates in double precision: exactly x_off + q / quanta_per_bp of the fixed-point words."""
        n = 2 * self.graph.n_nodes
        X, Y = np.zeros(n, dtype=np.float64), np.zeros(n, dtype=np.float64)
        f64p = C.POINTER(C.c_double)
        f = lib.pgsgd_session_download_coords_f64 if flush else lib.pgsgd_session_peek_coords_f64
        check(f(self._h, X.ctypes.data_as(f64p), Y.ctypes.data_as(f64p)), "download_f64")
        return X, Y

    def download_words(self, flush=True):
        """Raw device coordinate words, uint64 [2N] (see coord_format)."""
        w = np.zeros(2 * self.graph.n_nodes, dtype=np.uint64)
        f = lib.pgsgd_session_download_words if flush else lib.pgsgd_session_peek_words
        check(f(self._h, w.ctypes.data_as(C.POINTER(C.c_uint64))), "download_words")
        return w

    def tile_info(self):
        """dict(tiled, warm_per_lane, n_tiles, n_nonlocal_tiles, n_work_items, region_nodes, tile_steps, fast_math, parts,
        n_launch_items) of the session (parts: consecutive work items a window's tiles are cut into, pgsgd_session_tile_parts)
        (tiled=False: per-lane kernel; warm_per_lane, known after upload(): the initial layout had no global
        structure, so the iterations before cooling run the per-lane kernel)."""
        a, b, c_, r, t = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        on = lib.pgsgd_session_tile_info(self._h, C.byref(a), C.byref(b), C.byref(c_), C.byref(r), C.byref(t))
        n_launch = C.c_uint64()
        parts = lib.pgsgd_session_tile_parts(self._h, C.byref(n_launch))
        return dict(tiled=bool(on), warm_per_lane=(on == 2), n_tiles=a.value, n_nonlocal_tiles=b.value, n_work_items=c_.value, region_nodes=r.value, tile_steps=t.value,
                    fast_math=lib.pgsgd_session_tile_math(self._h) == 1, parts=max(0, parts), n_launch_items=n_launch.value,
                    xcd_runs=lib.pgsgd_session_tile_order(self._h) == 1)

    def split_info(self):
        """dict(split, apply_lanes): whether per-lane iterations run in two passes (a small lane-bound graph: n_streams
        streams sample, one workgroup of apply_lanes lanes moves the ends in LDS)."""
        lanes = C.c_uint32()
        mode = lib.pgsgd_session_split_info(self._h, C.byref(lanes))
        return dict(split=mode > 0, apply_lanes=int(lanes.value))

    def tile_table(self):
        """Tiles in work order: dict of arrays t0, cum, n, path, lanes (lanes that work on the tile at once, one term
        stream each), and steps_total."""
        cnt = lib.pgsgd_session_tile_table(self._h, None, None, None, None, 0, None)
        t0, cum = np.zeros(cnt, dtype=np.uint64), np.zeros(cnt, dtype=np.uint64)
        n, path = np.zeros(cnt, dtype=np.uint32), np.zeros(cnt, dtype=np.uint32)
        tot = C.c_uint64()
        u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        lib.pgsgd_session_tile_table(self._h, t0.ctypes.data_as(u64p), cum.ctypes.data_as(u64p), n.ctypes.data_as(u32p),
                                     path.ctypes.data_as(u32p), cnt, C.byref(tot))
        lanes = np.zeros(cnt, dtype=np.uint32)
        lib.pgsgd_session_tile_lanes(self._h, lanes.ctypes.data_as(u32p), cnt)
        return dict(t0=t0, cum=cum, n=n, path=path, lanes=lanes, steps_total=tot.value)


    def tile_items(self):
        """Work items in launch order: dict of arrays tile_begin, tile_end, win0, local, and n_first (items of the
        first launch, the even regions)."""
        cnt = lib.pgsgd_session_tile_items(self._h, None, None, None, None, 0, None)
        arrs = [np.zeros(cnt, dtype=np.uint32) for _ in range(4)]
        nf = C.c_uint64()
        u32p = C.POINTER(C.c_uint32)
        lib.pgsgd_session_tile_items(self._h, *[a.ctypes.data_as(u32p) for a in arrs], cnt, C.byref(nf))
        return dict(tile_begin=arrs[0], tile_end=arrs[1], win0=arrs[2], local=arrs[3], n_first=nf.value)
    def trace_tile_terms(self, tile, cooling, epoch, n_terms, capacity=1 << 16):
        """Replay of the terms one tile draws in iteration `epoch`: uint64 [terms, 4] = (ka, kb, off_a, off_b)."""
        out = np.zeros((capacity, 4), dtype=np.uint64)
        cnt = lib.pgsgd_session_trace_tile_terms(self._h, int(tile), 1 if cooling else 0, int(epoch), int(n_terms),
                                                 out.ctypes.data_as(C.POINTER(C.c_uint64)), capacity)
        if cnt < 0:
            check(int(cnt), "trace_tile_terms")
        return out[:cnt]

    def coord_format(self):
        """(fixed_point, x_off, y_off, quanta_per_bp) of the device coordinate words."""
        fp, xo, yo, q = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        check(lib.pgsgd_session_coord_format(self._h, C.byref(fp), C.byref(xo), C.byref(yo), C.byref(q)), "coord_format")
        return bool(fp.value), xo.value, yo.value, q.value

    def use_torch_stream(self):
        """Launch on torch's current stream so torch ops and collectives order with the kernels."""
        import torch
        check(lib.pgsgd_session_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "set_stream")

    def iteration(self, eta, cooling, n_terms):
        check(lib.pgsgd_session_iteration(self._h, float(eta), 1 if cooling else 0, int(n_terms)), "iteration")

    def iteration_part(self, eta, cooling, n_terms, part, n_parts):
        """Part `part` of `n_parts` of an iteration of n_terms terms (one part per multi-GPU exchange)."""
        check(lib.pgsgd_session_iteration_part(self._h, float(eta), 1 if cooling else 0, int(n_terms), int(part), int(n_parts)),
              "iteration_part")

    def sync(self):
        d = C.c_double()
        check(lib.pgsgd_session_sync(self._h, C.byref(d)), "sync")
        return d.value

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        check(lib.pgsgd_session_kernel_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0), "kernel_time")
        return ms.value, n.value

    def frame_status(self):
        """(guard_hit, doublings): whether the last iteration saw a coordinate in the outer quarter of the fixed-point
        frame that the session has not answered yet (sharded sessions leave the widening to their driver), and how
        often the frame was doubled so far."""
        hit, n = C.c_int(), C.c_uint32()
        check(lib.pgsgd_session_frame_status(self._h, C.byref(hit), C.byref(n)), "frame_status")
        return bool(hit.value), int(n.value)

    def reframe(self):
        check(lib.pgsgd_session_reframe(self._h), "reframe")

    def aux_time(self):
        """(snapshot_ms, drain_ms): time in the streaming kernels around the tile launches since the last reset."""
        a, b = C.c_double(), C.c_double()
        check(lib.pgsgd_session_aux_time(self._h, C.byref(a), C.byref(b)), "aux_time")
        return a.value, b.value

    def shader_clock(self):
        """(MHz, ms): the shader clock the last tile-kernel launch ran at and its duration as its first workgroup saw it."""
        mhz, ms = C.c_double(), C.c_double()
        check(lib.pgsgd_session_shader_clock(self._h, C.byref(mhz), C.byref(ms)), "shader_clock")
        return mhz.value, ms.value

    def tile_tail(self):
        """(alive, ms, workgroups): with PGSGD_DEBUG=1 PGSGD_TILE_TAIL=1, the share of workgroups x duration of the last
        windowed tile launch that its workgroups were alive for (the rest is the launch's tail), its duration, its grid."""
        a, b, n = C.c_double(), C.c_double(), C.c_uint32()
        check(lib.pgsgd_session_tile_tail(self._h, C.byref(a), C.byref(b), C.byref(n)), "tile_tail")
        return a.value, b.value, n.value

    def tile_conflicts(self):
        """(locked, lost): tile-kernel terms that went for their window ends' locks so far, and those that lost one."""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib.pgsgd_session_tile_conflicts(self._h, C.byref(a), C.byref(b)), "tile_conflicts")
        return a.value, b.value

    def step_records(self, first=0, count=None):
        """The session's step records [first, first + count) as a (count, 4) uint32 array {handle, node length, position low,
        position high} — built from the view's positions, or on the device when the view carries none (parity hook)."""
        count = self.graph.n_steps - first if count is None else count
        out = np.zeros((count, 4), dtype=np.uint32)
        check(lib.pgsgd_session_read_step_records(self._h, int(first), int(count), out.ctypes.data_as(C.POINTER(C.c_uint32))), "step_records")
        return out

    def drain_beside(self):
        """(on, ms): whether the session sums its launches' far pulls on a second stream beside the next launch (schedules of
        30 iterations and more, unsharded; PGSGD_FLAG_SYNC_DRAIN: never), and far_drain_kernel's time on that stream so far."""
        on, ms = C.c_int(), C.c_double()
        check(lib.pgsgd_session_drain_beside(self._h, C.byref(on), C.byref(ms)), "drain_beside")
        return bool(on.value), ms.value

    def drain_plan(self):
        """(parts, slices): workgroups per bucket's node range and per (bucket, part)'s messages in far_drain_kernel."""
        parts, slices = C.c_uint32(), C.c_uint32()
        check(lib.pgsgd_session_drain_plan(self._h, C.byref(parts), C.byref(slices)), "drain_plan")
        return parts.value, slices.value

    def probe_words(self):
        """Profiling hook: the twelve raw words of the tile kernel's probes (pgsgd_session_probe_words)."""
        out = (C.c_uint64 * 12)()
        check(lib.pgsgd_session_probe_words(self._h, out), "probe_words")
        return [int(v) for v in out]

    def terms_executed(self):
        """Terms the session's tile launches have executed so far, counted on the device (0 for a session without tiles)."""
        n = C.c_uint64()
        check(lib.pgsgd_session_terms_executed(self._h, C.byref(n)), "terms_executed")
        return int(n.value)

    def launch_counts(self):
        """(kernel launches, memsets + copies) the session's iterations have put on the stream."""
        k, c = C.c_uint64(), C.c_uint64()
        check(lib.pgsgd_session_launch_counts(self._h, C.byref(k), C.byref(c)), "launch_counts")
        return int(k.value), int(c.value)

    def outbox_overflow(self):
        """Far updates applied as direct atomics because the message pool share of their bucket was used up."""
        n = lib.pgsgd_session_outbox_overflow(self._h)
        if n < 0:
            check(int(n), "outbox_overflow")
        return int(n)

    def trace_terms(self, cooling, terms_per_stream):
        """Sampler-only parity hook: uint64 [terms_per_stream, n_streams, 4] = (ka, kb, off_a, off_b)."""
        out = np.zeros((terms_per_stream, self.n_streams, 4), dtype=np.uint64)
        check(lib.pgsgd_session_trace_terms(self._h, 1 if cooling else 0, terms_per_stream,
                                            out.ctypes.data_as(C.POINTER(C.c_uint64))), "trace_terms")
        return out


class Layout:
    """algorithms::layout::Layout (layout.hpp:24-40): X,Y per node end, `.lay` and TSV forms."""

    def __init__(self, X=None, Y=None):
        self.X = np.zeros(0) if X is None else np.ascontiguousarray(X, dtype=np.float64)
        self.Y = np.zeros(0) if Y is None else np.ascontiguousarray(Y, dtype=np.float64)

    def size(self):
        return len(self.X)

    def serialize(self, path):
        check(lib.pgsgd_write_lay(str(path).encode(), len(self.X), _f64(self.X), _f64(self.Y)), "write_lay")

    def to_bytes(self):
        buf, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        check(lib.pgsgd_lay_buffer(len(self.X), _f64(self.X), _f64(self.Y), C.byref(buf), C.byref(n)), "lay_buffer")
        try:
            return C.string_at(buf, n.value)
        finally:
            lib.pgsgd_free(buf)

    @classmethod
    def load(cls, path):
        n, px, py = C.c_uint64(), _F64P(), _F64P()
        check(lib.pgsgd_read_lay(str(path).encode(), C.byref(n), C.byref(px), C.byref(py)), "read_lay")
        try:
            X = np.ctypeslib.as_array(px, shape=(n.value,)).copy() if n.value else np.zeros(0)
            Y = np.ctypeslib.as_array(py, shape=(n.value,)).copy() if n.value else np.zeros(0)
        finally:
            lib.pgsgd_free(px)
            lib.pgsgd_free(py)
        return cls(X, Y)

    def get_X(self):
        return self.X

    def get_Y(self):
        return self.Y


def shard_flags(graph: Graph, world, flags=0):
    """Flag bits a multi-GPU driver ORs into the params of its `world` sessions before it creates them (pgsgd_shard_flags):
    FLAG_REGION_128 where 256-node regions would leave a device fewer than a thousand windows per launch — the sessions' rule
    (set_shard) then shards by region with the exact exchange: G ranks hold one GPU's layout bit for bit."""
    return int(lib.pgsgd_shard_flags(int(graph.n_nodes), int(world), int(flags)))


def weak_components(graph: Graph):
    comp = np.zeros(graph.n_nodes, dtype=np.uint32)
    e = np.ascontiguousarray(graph.edges, dtype=np.uint64)
    n = lib.pgsgd_weak_components(graph.n_nodes, e.ctypes.data_as(C.POINTER(C.c_uint64)), len(e),
                                  comp.ctypes.data_as(C.POINTER(C.c_uint32)))
    if n < 0:
        check(int(n), "weak_components")
    return comp, int(n)


def pack_components(graph: Graph, X, Y):
    """layout_main.cpp:401-435, in place on float64 X,Y; returns (comp_of_node, n_components)."""
    comp, n = weak_components(graph)
    check(lib.pgsgd_pack_components(graph.n_nodes, comp.ctypes.data_as(C.POINTER(C.c_uint32)), n, _f64(X), _f64(Y)), "pack")
    return comp, n


def write_tsv(path, graph: Graph, comp, n_comp, X, Y):
    check(lib.pgsgd_write_tsv(str(path).encode(), graph.n_nodes, comp.ctypes.data_as(C.POINTER(C.c_uint32)), n_comp,
                              _f64(X), _f64(Y)), "write_tsv")


def path_stress(graph: Graph, X, Y, n_pairs=1_000_000, seed=0x5eed):
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    s = C.c_double()
    check(lib.pgsgd_path_stress(C.byref(graph.view), _f64(X), _f64(Y), n_pairs, seed, C.byref(s)), "path_stress")
    return s.value


def path_stress_near(graph: Graph, X, Y, zmax=3, theta=0.99, threads=0, mod_step=0, mod_rank=0):
    """The near pairs' part of the expected sampled stress without sampling error (pgsgd_path_stress_near): every pair of
    steps at most zmax apart, all end choices, weighted by the sampler's probability.  Returns dict(num [zmax, 2, 2], mass
    [zmax, 2, 2], zero_mass, near = num.sum() / (1 - zero_mass): the pairs' contribution to the expectation when pairs
    beyond zmax are ignored in the normalisation's d = 0 share; hist_step / hist_rank when asked for)."""
    import os
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    num, mass = np.zeros(zmax * 4), np.zeros(zmax * 4)
    zero = C.c_double()
    hs = np.zeros(mod_step) if mod_step else None
    hr = np.zeros(mod_rank) if mod_rank else None
    threads = threads or min(64, os.cpu_count() or 1)
    check(lib.pgsgd_path_stress_near(C.byref(graph.view), _f64(X), _f64(Y), zmax, theta, threads, _f64(num), _f64(mass), C.byref(zero),
                                     mod_step, _f64(hs) if mod_step else None, mod_rank, _f64(hr) if mod_rank else None), "path_stress_near")
    out = dict(num=num.reshape(zmax, 2, 2), mass=mass.reshape(zmax, 2, 2), zero_mass=zero.value, near=float(num.sum() / (1.0 - zero.value)))
    if mod_step:
        out["hist_step"] = hs
    if mod_rank:
        out["hist_rank"] = hr
    return out


def path_distance(graph: Graph, X, Y):
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    a, b = C.c_double(), C.c_double()
    check(lib.pgsgd_path_distance(C.byref(graph.view), _f64(X), _f64(Y), C.byref(a), C.byref(b)), "path_distance")
    return a.value, b.value


def main_layout(argv=None):
    """`odgi layout` with the reference's flags (layout_main.cpp:18-466). argv without the program name."""
    argv = list(sys.argv[1:] if argv is None else argv)
    full = [b"odgi", b"layout"] + [a.encode() for a in argv]
    arr = (C.c_char_p * len(full))(*full)
    return int(lib.pgsgd_main_layout(len(full), arr))
