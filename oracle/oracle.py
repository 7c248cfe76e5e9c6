"""ctypes wrapper of the CPU oracle (oracle/pgsgd_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by odgi_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force=False):
    need = [os.path.join(_BUILD, f) for f in ("liboracle.so", "liboracle_fast.so", "check_libstdcxx")]
    if force or not all(os.path.exists(f) for f in need):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return need


class OrcGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_steps", C.c_uint64), ("n_paths", C.c_uint64),
                ("node_len", C.POINTER(C.c_uint32)), ("path_first", C.POINTER(C.c_uint64)),
                ("step_path", C.POINTER(C.c_uint32)), ("step_handle", C.POINTER(C.c_uint32)),
                ("step_pos", C.POINTER(C.c_uint64))]


class OrcParams(C.Structure):
    _fields_ = [("iter_max", C.c_uint64), ("iter_with_max_learning_rate", C.c_uint64),
                ("min_term_updates", C.c_uint64), ("delta", C.c_double), ("eps", C.c_double),
                ("eta_max", C.c_double), ("theta", C.c_double), ("space", C.c_uint64),
                ("space_max", C.c_uint64), ("space_quantization_step", C.c_uint64),
                ("cooling_start", C.c_double)]


class HogStats(C.Structure):
    _fields_ = [("terms", C.c_uint64), ("iterations", C.c_uint64), ("seconds", C.c_double)]


_F64P = C.POINTER(C.c_double)
_F32P = C.POINTER(C.c_float)
_U64P = C.POINTER(C.c_uint64)
_U32P = C.POINTER(C.c_uint32)
_U8P = C.POINTER(C.c_uint8)


def _load(name):
    build()
    lib = C.CDLL(os.path.join(_BUILD, name))
    lib.orc_rng_seed.argtypes = [C.c_uint64, _U64P]
    lib.orc_rng_next.argtypes = [_U64P]
    lib.orc_rng_next.restype = C.c_uint64
    lib.orc_uniform_u64.argtypes = [_U64P, C.c_uint64]
    lib.orc_uniform_u64.restype = C.c_uint64
    lib.orc_canonical.argtypes = [_U64P]
    lib.orc_canonical.restype = C.c_double
    lib.orc_fast_precise_pow.argtypes = [C.c_double, C.c_double]
    lib.orc_fast_precise_pow.restype = C.c_double
    lib.orc_zipf.argtypes = [_U64P, C.c_uint64, C.c_double, C.c_double]
    lib.orc_zipf.restype = C.c_uint64
    lib.orc_schedule.argtypes = [C.POINTER(OrcParams), _F64P]
    lib.orc_zeta_size.argtypes = [C.c_uint64] * 3
    lib.orc_zeta_size.restype = C.c_size_t
    lib.orc_zetas.argtypes = [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, _F64P]
    lib.orc_trace_terms.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint32, C.c_uint32,
                                    C.c_int, C.c_uint32, C.c_uint64, _U64P]
    lib.orc_tile_terms.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _U64P]
    lib.orc_tile_terms.restype = C.c_uint64
    lib.orc_layout_streams_f32.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint32,
                                           C.c_uint32, C.c_int, C.c_uint32, _F32P, _F32P, _F64P]
    lib.orc_tile_layout_q32.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64,
                                        C.c_uint64, _U64P, _U64P, _U32P, _U32P, _U32P, C.c_uint64, C.c_uint64, C.c_uint64,
                                        _U32P, _U32P, _U32P, _U32P, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                        _F32P, _F32P, C.POINTER(C.c_double), _U64P, C.POINTER(C.c_uint64)]
    lib.orc_tile_layout_q32.restype = None
    lib.orc_tile_layout_q32_ex.argtypes = lib.orc_tile_layout_q32.argtypes + [C.c_uint32, C.c_uint64]
    lib.orc_tile_layout_q32_ex.restype = None
    lib.orc_layout_streams_q32.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint32,
                                           C.c_uint32, C.c_int, C.c_uint32, C.c_double, C.c_double, C.c_double, _F32P, _F32P, _F64P, _U64P]
    lib.orc_layout_streams_f64.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint32,
                                           C.c_uint32, _F64P, _F64P]
    lib.orc_layout_batched_f64.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint32,
                                           C.c_uint32, _F64P, _F64P]
    lib.orc_layout_hogwild.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint32, C.c_double, _F64P, _F64P,
                                       C.POINTER(HogStats)]
    lib.orc_layout_hogwild_curve.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint32, C.c_double, _F64P, _F64P,
                                             C.POINTER(HogStats), C.c_uint64, _U64P, _F64P, _F64P]
    lib.orc_sort_initial.argtypes = [C.POINTER(OrcGraph), _F64P]
    lib.orc_sort_trace_terms.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint32, C.c_uint32, C.c_int,
                                         C.c_uint64, _U64P]
    lib.orc_sort_streams.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint64, C.c_uint32, C.c_uint32, C.c_double,
                                     _U8P, _F64P, _F64P]
    lib.orc_sort_hogwild.argtypes = [C.POINTER(OrcGraph), C.POINTER(OrcParams), C.c_uint32, C.c_double, _U8P, _F64P, C.POINTER(HogStats)]
    lib.orc_sort_stress.argtypes = [C.POINTER(OrcGraph), _F64P, C.c_uint64, C.c_uint64]
    lib.orc_sort_stress.restype = C.c_double
    lib.orc_path_stress_sampled.argtypes = [C.POINTER(OrcGraph), _F64P, _F64P, C.c_uint64, C.c_uint64]
    lib.orc_path_stress_sampled.restype = C.c_double
    lib.orc_path_stress_exhaustive.argtypes = [C.POINTER(OrcGraph), _F64P, _F64P]
    lib.orc_path_stress_exhaustive.restype = C.c_double
    lib.orc_path_distance.argtypes = [C.POINTER(OrcGraph), _F64P, _F64P, _F64P, _F64P]
    return lib


_libs = {}


FAST_FLAGS = "-Ofast -fPIC -std=c11 -funroll-all-loops -pipe"   # oracle/Makefile FAST: the reference's Release flags (CMakeLists.txt:85) minus -march=native
_fast_name = ["liboracle_fast.so"]


def build_fast_native():
    """Compile the timed build of the oracle ON THIS HOST with the reference's full Release flags, -march=native included
    (oracle/_build/liboracle_fast_native.so), and make it what lib(fast=True) loads.  bench.py's cpu_baseline calls this on the box
    it times on; returns the flags used (the portable build's when gcc is missing or fails)."""
    out = os.path.join(_BUILD, "liboracle_fast_native.so")
    try:
        os.makedirs(_BUILD, exist_ok=True)
        subprocess.check_call(["gcc"] + FAST_FLAGS.split() + ["-march=native", "-shared", "-o", out, os.path.join(_HERE, "pgsgd_oracle.c"), "-lm", "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _fast_name[0] = "liboracle_fast_native.so"
        return FAST_FLAGS + " -march=native (built on this host)"
    except Exception:  # noqa: BLE001
        return FAST_FLAGS + " (portable build: no -march=native)"


def lib(fast=False):
    name = _fast_name[0] if fast else "liboracle.so"
    if name not in _libs:
        _libs[name] = _load(name)
    return _libs[name]


class Graph:
    """Holds numpy arrays and the C view over them (same layout as pgsgd_graph_view)."""

    def __init__(self, node_len, path_first, step_path, step_handle, step_pos):
        self.node_len = np.ascontiguousarray(node_len, dtype=np.uint32)
        self.path_first = np.ascontiguousarray(path_first, dtype=np.uint64)
        self.step_path = np.ascontiguousarray(step_path, dtype=np.uint32)
        self.step_handle = np.ascontiguousarray(step_handle, dtype=np.uint32)
        self.step_pos = np.ascontiguousarray(step_pos, dtype=np.uint64)
        v = OrcGraph()
        v.n_nodes, v.n_steps, v.n_paths = len(self.node_len), len(self.step_handle), len(self.path_first) - 1
        v.node_len = self.node_len.ctypes.data_as(C.POINTER(C.c_uint32))
        v.path_first = self.path_first.ctypes.data_as(_U64P)
        v.step_path = self.step_path.ctypes.data_as(C.POINTER(C.c_uint32))
        v.step_handle = self.step_handle.ctypes.data_as(C.POINTER(C.c_uint32))
        v.step_pos = self.step_pos.ctypes.data_as(_U64P)
        self.view = v

    @classmethod
    def from_product(cls, g):
        """Copy the arrays of an odgi_amd.Graph (tests build the graph once, with the product loader)."""
        return cls(g.node_len.copy(), g.path_first.copy(), g.step_path.copy(), g.step_handle.copy(), g.step_pos.copy())

    n_nodes = property(lambda s: len(s.node_len))
    n_steps = property(lambda s: len(s.step_handle))
    n_paths = property(lambda s: len(s.path_first) - 1)


def params(**kw):
    p = OrcParams()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def params_from(lp):
    """From an odgi_amd.LayoutParams-like object."""
    return params(iter_max=lp.iter_max, iter_with_max_learning_rate=lp.iter_with_max_learning_rate,
                  min_term_updates=lp.min_term_updates, delta=lp.delta, eps=lp.eps, eta_max=lp.eta_max,
                  theta=lp.theta, space=lp.space, space_max=lp.space_max,
                  space_quantization_step=lp.space_quantization_step, cooling_start=lp.cooling_start)


def schedule(p):
    etas = np.zeros(p.iter_max + 1)
    lib().orc_schedule(C.byref(p), etas.ctypes.data_as(_F64P))
    return etas


def zetas(theta, space, space_max, quant):
    n = lib().orc_zeta_size(space, space_max, quant)
    z = np.zeros(n)
    lib().orc_zetas(theta, space, space_max, quant, z.ctypes.data_as(_F64P))
    return z


def trace_terms(g, p, seed, n_streams, stream_offset, cooling, terms_per_stream, terms_per_anchor=1):
    out = np.zeros((terms_per_stream, n_streams, 4), dtype=np.uint64)
    lib().orc_trace_terms(C.byref(g.view), C.byref(p), seed, n_streams, stream_offset, 1 if cooling else 0,
                          terms_per_anchor, terms_per_stream, out.ctypes.data_as(_U64P))
    return out


def tile_terms(g, p, seed_base, epoch, n_terms, steps_total, tile, lanes, t0, cum, n, path, cooling, capacity=1 << 16, share=4):
    """Terms of tile number `tile` of the tile table, worked on by `lanes` lanes (one stream per lane), in term order.  share: the lanes
    that share a line of partner records in a uniform trip (4: quads, what sessions run; 2: the pairs of rounds 4-6; 1: none)."""
    out = np.zeros((capacity, 4), dtype=np.uint64)
    cnt = lib().orc_tile_terms(C.byref(g.view), C.byref(p), seed_base, epoch, n_terms, steps_total, int(tile), int(lanes), int(t0), int(cum),
                               int(n), int(path), 1 if cooling else 0, int(share), out.ctypes.data_as(_U64P))
    return out[:cnt]


def layout_streams_f32(g, p, seed, n_streams, X, Y, stream_offset=0, stores=False, terms_per_anchor=1):
    X = np.ascontiguousarray(X, dtype=np.float32).copy()
    Y = np.ascontiguousarray(Y, dtype=np.float32).copy()
    d = C.c_double()
    lib().orc_layout_streams_f32(C.byref(g.view), C.byref(p), seed, n_streams, stream_offset, 1 if stores else 0, terms_per_anchor,
                                 X.ctypes.data_as(_F32P), Y.ctypes.data_as(_F32P), C.byref(d))
    return X, Y, d.value


def layout_streams_q32(g, p, seed, n_streams, X, Y, x_off, y_off, quanta_per_bp, stream_offset=0, stores=False,
                       terms_per_anchor=1):
    """Mirror of the device's packed fixed-point path. Returns X, Y (fp32), last delta_max, checksums[4]."""
    X = np.ascontiguousarray(X, dtype=np.float32).copy()
    Y = np.ascontiguousarray(Y, dtype=np.float32).copy()
    d = C.c_double()
    ck = np.zeros(4, dtype=np.uint64)
    lib().orc_layout_streams_q32(C.byref(g.view), C.byref(p), seed, n_streams, stream_offset, 1 if stores else 0, terms_per_anchor,
                                 x_off, y_off, quanta_per_bp,
                                 X.ctypes.data_as(_F32P), Y.ctypes.data_as(_F32P), C.byref(d), ck.ctypes.data_as(_U64P))
    return X, Y, d.value, ck


TILE_DRAIN_AFTER, TILE_TWO_SNAPSHOTS, TILE_NO_FLUSH, TILE_CONSTANT_RELAX, TILE_SNAPSHOT_PASS, TILE_LANE_COIN, TILE_NO_PAIRS, TILE_RELAX_R5 = 1, 2, 4, 8, 16, 32, 64, 128
TILE_PAIRS = 0x10000           # uniform partners shared by pairs of lanes (rounds 4-6) instead of quads
TILE_DRAIN_BESIDE = 0x8000   # per-colour outboxes, pulls delivered before the same colour's next launch (sessions of >= 30 iterations)
TILE_ROUND2 = TILE_DRAIN_AFTER | TILE_TWO_SNAPSHOTS | TILE_CONSTANT_RELAX | TILE_SNAPSHOT_PASS | TILE_LANE_COIN | TILE_NO_PAIRS   # the launch order, far-pull policy and per-lane coin of round 2
TILE_ROUND3 = TILE_LANE_COIN | TILE_NO_PAIRS | TILE_RELAX_R5   # round 3's pipeline: today's launch order, the Zipf/uniform coin per lane, far pulls ramping to half a projection
TILE_ROUND5 = TILE_RELAX_R5 | TILE_PAIRS   # rounds 4-5: wave coin, partner pairs; far pulls ramping 0.1 .. 0.5 (round 6: 0.2 .. 1.0, partner quads)


def tile_wave_coin(seed_base, epoch, tile, wave, trip):
    """The Zipf/uniform coin that the lanes of one wave of a tile share in one trip of a warm iteration."""
    f = lib().orc_tile_wave_coin
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]
    f.restype = C.c_int
    return int(f(int(seed_base), int(epoch), int(tile), int(wave), int(trip)))


def tile_layout_q32(g, p, seed_base, tiles, items, region, X, Y, x_off, y_off, quanta_per_bp, policy=0, stop_after=0):
    """Sequential mirror of the tile kernel (one workgroup, one lane per tile).  `tiles` / `items` are the dicts of
    LayoutSession.tile_table() / tile_items(); tiles["lanes"] (lanes per tile, default 1 each) selects the term
    streams.  policy / stop_after: see orc_tile_layout_q32_ex (0, 0 = the product's run).
    Returns X, Y (fp32), last delta_max, checksums[4], far terms."""
    X = np.ascontiguousarray(X, dtype=np.float32).copy()
    Y = np.ascontiguousarray(Y, dtype=np.float32).copy()
    d, far = C.c_double(), C.c_uint64()
    ck = np.zeros(4, dtype=np.uint64)
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
    t0, cum, tn, tp = u64(tiles["t0"]), u64(tiles["cum"]), u32(tiles["n"]), u32(tiles["path"])
    tl = u32(tiles["lanes"]) if "lanes" in tiles else np.ones(len(t0), dtype=np.uint32)
    tb, te, w0, lo = u32(items["tile_begin"]), u32(items["tile_end"]), u32(items["win0"]), u32(items["local"])
    lib().orc_tile_layout_q32_ex(C.byref(g.view), C.byref(p), seed_base, len(t0), t0.ctypes.data_as(_U64P), cum.ctypes.data_as(_U64P),
                              tn.ctypes.data_as(_U32P), tp.ctypes.data_as(_U32P), tl.ctypes.data_as(_U32P), int(tiles["steps_total"]), len(tb),
                              int(items["n_first"]),
                              tb.ctypes.data_as(_U32P), te.ctypes.data_as(_U32P), w0.ctypes.data_as(_U32P), lo.ctypes.data_as(_U32P),
                              int(region), x_off, y_off, quanta_per_bp, X.ctypes.data_as(_F32P), Y.ctypes.data_as(_F32P),
                              C.byref(d), ck.ctypes.data_as(_U64P), C.byref(far), int(policy), int(stop_after))
    return X, Y, d.value, ck, far.value


def layout_streams_f64(g, p, seed, n_streams, X, Y, stream_offset=0):
    X = np.ascontiguousarray(X, dtype=np.float64).copy()
    Y = np.ascontiguousarray(Y, dtype=np.float64).copy()
    lib().orc_layout_streams_f64(C.byref(g.view), C.byref(p), seed, n_streams, stream_offset,
                                 X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P))
    return X, Y


def layout_batched_f64(g, p, seed, n_streams, X, Y, stream_offset=0):
    """Model of the device's concurrency: n_streams terms read one snapshot, their deltas are summed."""
    X = np.ascontiguousarray(X, dtype=np.float64).copy()
    Y = np.ascontiguousarray(Y, dtype=np.float64).copy()
    lib().orc_layout_batched_f64(C.byref(g.view), C.byref(p), seed, n_streams, stream_offset,
                                 X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P))
    return X, Y


def layout_hogwild(g, p, nthreads, X, Y, max_seconds=0.0, fast=False):
    X = np.ascontiguousarray(X, dtype=np.float64).copy()
    Y = np.ascontiguousarray(Y, dtype=np.float64).copy()
    st = HogStats()
    lib(fast).orc_layout_hogwild(C.byref(g.view), C.byref(p), nthreads, max_seconds,
                                 X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P), C.byref(st))
    return X, Y, {"terms": st.terms, "iterations": st.iterations, "seconds": st.seconds}


def layout_hogwild_curve(g, p, nthreads, X, Y, snap_iters, fast=False):
    """layout_hogwild that also returns the coordinates after the iterations in snap_iters (1-based): X, Y, stats,
    snapX [k, 2N], snapY [k, 2N]."""
    X = np.ascontiguousarray(X, dtype=np.float64).copy()
    Y = np.ascontiguousarray(Y, dtype=np.float64).copy()
    it = np.ascontiguousarray(snap_iters, dtype=np.uint64)
    sx = np.zeros((len(it), len(X)))
    sy = np.zeros((len(it), len(X)))
    st = HogStats()
    lib(fast).orc_layout_hogwild_curve(C.byref(g.view), C.byref(p), nthreads, 0.0, X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P),
                                       C.byref(st), len(it), it.ctypes.data_as(_U64P), sx.ctypes.data_as(_F64P), sy.ctypes.data_as(_F64P))
    return X, Y, {"terms": st.terms, "iterations": st.iterations, "seconds": st.seconds}, sx, sy


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def path_stress_sampled(g, X, Y, n_pairs=1_000_000, seed=0x5eed):
    X, Y = _d(X), _d(Y)
    return lib().orc_path_stress_sampled(C.byref(g.view), X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P), n_pairs, seed)


def path_stress_near(g, X, Y, zmax=4, theta=0.99, threads=0):
    """The near pairs' exact contribution to the expectation of path_stress_sampled (orc_path_stress_near):
    dict(num [zmax, 2, 2], mass [zmax, 2, 2], zero_mass, near = num.sum() / (1 - zero_mass))."""
    X, Y = _d(X), _d(Y)
    num, mass, zero = np.zeros(zmax * 4), np.zeros(zmax * 4), C.c_double()
    f = lib().orc_path_stress_near
    f.restype = None
    f.argtypes = [C.POINTER(OrcGraph), _F64P, _F64P, C.c_uint32, C.c_double, C.c_uint32, _F64P, _F64P, C.POINTER(C.c_double)]
    f(C.byref(g.view), X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P), zmax, theta, threads or (os.cpu_count() or 1),
      num.ctypes.data_as(_F64P), mass.ctypes.data_as(_F64P), C.byref(zero))
    return {"num": num.reshape(zmax, 2, 2), "mass": mass.reshape(zmax, 2, 2), "zero_mass": zero.value, "near": float(num.sum() / (1.0 - zero.value))}


def path_stress_exhaustive(g, X, Y):
    X, Y = _d(X), _d(Y)
    return lib().orc_path_stress_exhaustive(C.byref(g.view), X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P))


def path_distance(g, X, Y):
    X, Y = _d(X), _d(Y)
    a, b = C.c_double(), C.c_double()
    lib().orc_path_distance(C.byref(g.view), X.ctypes.data_as(_F64P), Y.ctypes.data_as(_F64P), C.byref(a), C.byref(b))
    return a.value, b.value


# ---- 1D path-guided SGD (odgi sort -Y) ---------------------------------------------------------
def sort_initial(g):
    X = np.zeros(g.n_nodes)
    lib().orc_sort_initial(C.byref(g.view), X.ctypes.data_as(_F64P))
    return X


def sort_trace_terms(g, p, seed, n_streams, stream_offset, cooling, terms_per_stream):
    out = np.zeros((terms_per_stream, n_streams, 2), dtype=np.uint64)
    lib().orc_sort_trace_terms(C.byref(g.view), C.byref(p), seed, n_streams, stream_offset, 1 if cooling else 0, terms_per_stream,
                               out.ctypes.data_as(_U64P))
    return out


def sort_streams(g, p, seed, n_streams, X, quanta_per_bp=65536.0, stream_offset=0, frozen=None):
    X = _d(X).copy()
    d = C.c_double()
    fz = None if frozen is None else np.ascontiguousarray(frozen, dtype=np.uint8)
    lib().orc_sort_streams(C.byref(g.view), C.byref(p), seed, n_streams, stream_offset, quanta_per_bp,
                           None if fz is None else fz.ctypes.data_as(_U8P), X.ctypes.data_as(_F64P), C.byref(d))
    return X, d.value


def sort_hogwild(g, p, nthreads, X, max_seconds=0.0, fast=False, frozen=None):
    X = _d(X).copy()
    st = HogStats()
    fz = None if frozen is None else np.ascontiguousarray(frozen, dtype=np.uint8)
    lib(fast).orc_sort_hogwild(C.byref(g.view), C.byref(p), nthreads, max_seconds, None if fz is None else fz.ctypes.data_as(_U8P),
                               X.ctypes.data_as(_F64P), C.byref(st))
    return X, {"terms": st.terms, "iterations": st.iterations, "seconds": st.seconds}


def sort_stress(g, X, n_pairs=1_000_000, seed=0x5eed):
    X = _d(X)
    return lib().orc_sort_stress(C.byref(g.view), X.ctypes.data_as(_F64P), n_pairs, seed)
