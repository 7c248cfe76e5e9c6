"""ctypes binding of libpgsgd.so (the C ABI declared in include/pgsgd.h).

The library is the product: HIP kernels for gfx950 plus the host code around them.  There is no
Python or CPU implementation of the layout to fall back to: if the shared object is missing the
import fails, and every compute entry point returns PGSGD_E_NODEVICE without a HIP device.
"""
import ctypes as C
import os

# torch ships its own HIP runtime with the same soname as /opt/rocm's; importing it first makes
# libpgsgd bind to the runtime that owns torch's device memory and streams.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpgsgd.so")
if os.environ.get("PGSGD_DEBUG", "")[:1] == "1" and os.environ.get("PGSGD_LIB"):
    # experiment knob (tools/): another build of the same sources, e.g. a different register budget of the tile kernel
    LIB_PATH = os.path.join(_HERE, "lib", os.environ["PGSGD_LIB"])
if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C odgi_amd/csrc` (hipcc --offload-arch=gfx950). There is no fallback implementation.")
lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

u32, u64, i32, i64, f64 = C.c_uint32, C.c_uint64, C.c_int32, C.c_int64, C.c_double
P = C.POINTER


class GraphView(C.Structure):
    _fields_ = [("n_nodes", u64), ("n_steps", u64), ("n_paths", u64),
                ("node_len", P(u32)), ("path_first", P(u64)), ("step_path", P(u32)),
                ("step_handle", P(u32)), ("step_pos", P(u64))]


class Params(C.Structure):
    _fields_ = [("iter_max", u64), ("iter_with_max_learning_rate", u64), ("min_term_updates", u64),
                ("delta", f64), ("eps", f64), ("eta_max", f64), ("theta", f64),
                ("space", u64), ("space_max", u64), ("space_quantization_step", u64),
                ("cooling_start", f64), ("seed", u64), ("n_streams", u32), ("stream_offset", u32),
                ("device", i32), ("snapshot", i32), ("snapshot_prefix", C.c_char_p),
                ("progress", i32), ("flags", u32), ("terms_per_anchor", u32), ("n_devices", u32)]


class Stats(C.Structure):
    _fields_ = [("iterations", u64), ("term_updates", u64), ("last_delta_max", f64),
                ("kernel_ms", f64), ("wall_ms", f64), ("n_streams", u32), ("early_stop", u32),
                ("frame_doublings", u32), ("apply_lanes", u32), ("relabeled", u32), ("tiled", u32)]


ABI_VERSION = 7   # include/pgsgd.h: PGSGD_ABI_VERSION this binding was written against


def _check_abi():
    """Once, at load time: the library's ABI version and the sizes of the structs this binding passes by pointer.  A library
    built from another header would read or write past them (pgsgd_layout_run memsets the whole pgsgd_stats)."""
    try:
        lib.pgsgd_abi_version.restype = C.c_int
        got = lib.pgsgd_abi_version()
    except AttributeError:
        raise ImportError(f"{LIB_PATH} predates PGSGD_ABI_VERSION: rebuild it from these sources (make -C odgi_amd/csrc)")
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {got}, this binding expects {ABI_VERSION}: rebuild the library")
    sizes = [C.c_size_t(), C.c_size_t(), C.c_size_t()]
    lib.pgsgd_abi_struct_sizes.restype = None
    lib.pgsgd_abi_struct_sizes(C.byref(sizes[0]), C.byref(sizes[1]), C.byref(sizes[2]))
    want = (C.sizeof(GraphView), C.sizeof(Params), C.sizeof(Stats))
    if tuple(v.value for v in sizes) != want:
        raise ImportError(f"struct sizes differ: library {[v.value for v in sizes]}, binding {list(want)} (graph view, params, stats)")


_check_abi()

FLAG_COORD_LOAD_PLAIN = 0x1
FLAG_FP32_ATOMICS = 0x2
FLAG_HOGWILD_STORES = 0x4
FLAG_NO_TILES = 0x8
FLAG_NO_FAR_CAP = 0x10
FLAG_ONE_SIDED_FAR = 0x20
FLAG_HOT_NODE_CAP = 0x40
FLAG_NO_PIPELINE = 0x80
FLAG_NO_SPLIT = 0x1000
FLAG_EXACT_MATH = 0x2000
FLAG_NO_PARTNER_PAIRS = 0x4000
FLAG_LOCK_WINDOW_ENDS = 0x8000
FLAG_NO_RELABEL = 0x10000
FLAG_SYNC_DRAIN = 0x20000
FLAG_REGION_128 = 0x40000
FLAG_SHARD_TILES = 0x80000
DEFAULT_SEED = 9399220
# error codes of include/pgsgd.h
E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_IO, E_FORMAT, E_NOTOPTIMIZED, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7, -8

# every symbol include/pgsgd.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("pgsgd_strerror", C.c_char_p, [C.c_int]),
    ("pgsgd_last_error", C.c_char_p, []),
    ("pgsgd_params_defaults", C.c_int, [P(GraphView), P(Params)]),
    ("pgsgd_schedule", i64, [P(Params), P(f64), C.c_size_t]),
    ("pgsgd_zeta_table_size", C.c_size_t, [u64, u64, u64]),
    ("pgsgd_zeta_table", C.c_int, [f64, u64, u64, u64, P(f64), C.c_size_t]),
    ("pgsgd_init_layout", C.c_int, [P(GraphView), C.c_char, u64, P(f64), P(f64)]),
    ("pgsgd_layout_run", C.c_int, [P(GraphView), P(Params), P(C.c_float), P(C.c_float), P(Stats)]),
    ("pgsgd_layout_run_f64", C.c_int, [P(GraphView), P(Params), P(f64), P(f64), P(Stats)]),
    ("pgsgd_session_create", C.c_int, [P(GraphView), P(Params), P(C.c_void_p)]),
    ("pgsgd_session_destroy", None, [C.c_void_p]),
    ("pgsgd_session_upload_coords", C.c_int, [C.c_void_p, P(C.c_float), P(C.c_float)]),
    ("pgsgd_session_download_coords", C.c_int, [C.c_void_p, P(C.c_float), P(C.c_float)]),
    ("pgsgd_session_download_coords_f64", C.c_int, [C.c_void_p, P(f64), P(f64)]),
    ("pgsgd_session_coords_ptr", C.c_void_p, [C.c_void_p]),
    ("pgsgd_session_download_words", C.c_int, [C.c_void_p, P(u64)]),
    ("pgsgd_session_peek_coords", C.c_int, [C.c_void_p, P(C.c_float), P(C.c_float)]),
    ("pgsgd_session_peek_coords_f64", C.c_int, [C.c_void_p, P(f64), P(f64)]),
    ("pgsgd_session_peek_words", C.c_int, [C.c_void_p, P(u64)]),
    ("pgsgd_session_coord_format", C.c_int, [C.c_void_p, P(C.c_int), P(f64), P(f64), P(f64)]),
    ("pgsgd_session_stream", C.c_void_p, [C.c_void_p]),
    ("pgsgd_session_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pgsgd_session_iteration", C.c_int, [C.c_void_p, f64, C.c_int, u64]),
    ("pgsgd_session_iteration_part", C.c_int, [C.c_void_p, f64, C.c_int, u64, u32, u32]),
    ("pgsgd_session_sync", C.c_int, [C.c_void_p, P(f64)]),
    ("pgsgd_session_flush", C.c_int, [C.c_void_p]),
    ("pgsgd_session_frame_status", C.c_int, [C.c_void_p, P(C.c_int), P(u32)]),
    ("pgsgd_session_reframe", C.c_int, [C.c_void_p]),
    ("pgsgd_session_kernel_time", C.c_int, [C.c_void_p, P(f64), P(u64), C.c_int]),
    ("pgsgd_abi_version", C.c_int, []),
    ("pgsgd_abi_struct_sizes", None, [P(C.c_size_t), P(C.c_size_t), P(C.c_size_t)]),
    ("pgsgd_session_aux_time", C.c_int, [C.c_void_p, P(f64), P(f64)]),
    ("pgsgd_session_launch_counts", C.c_int, [C.c_void_p, P(u64), P(u64)]),
    ("pgsgd_session_shader_clock", C.c_int, [C.c_void_p, P(f64), P(f64)]),
    ("pgsgd_session_tile_tail", C.c_int, [C.c_void_p, P(f64), P(f64), P(C.c_uint32)]),
    ("pgsgd_session_tile_parts", C.c_int, [C.c_void_p, P(C.c_uint64)]),
    ("pgsgd_session_tile_order", C.c_int, [C.c_void_p]),
    ("pgsgd_tile_parts_for", u32, [u64, u64, u64]),
    ("pgsgd_tile_split_items", C.c_int64, [P(u32), P(u32), P(u32), u64, u32, u32, P(u32), P(u32), P(u32), P(u32), u64]),
    ("pgsgd_session_tile_conflicts", C.c_int, [C.c_void_p, P(u64), P(u64)]),
    ("pgsgd_session_terms_executed", C.c_int, [C.c_void_p, P(u64)]),
    ("pgsgd_session_read_step_records", C.c_int, [C.c_void_p, u64, u64, P(u32)]),
    ("pgsgd_session_drain_beside", C.c_int, [C.c_void_p, P(C.c_int), P(f64)]),
    ("pgsgd_shard_flags", u32, [u64, u32, u32]),
    ("pgsgd_session_drain_plan", C.c_int, [C.c_void_p, P(C.c_uint32), P(C.c_uint32)]),
    ("pgsgd_session_probe_words", C.c_int, [C.c_void_p, P(u64)]),
    ("pgsgd_session_outbox_overflow", i64, [C.c_void_p]),
    ("pgsgd_session_n_streams", u32, [C.c_void_p]),
    ("pgsgd_session_exchange_mark", C.c_int, [C.c_void_p]),
    ("pgsgd_session_exchange_begin", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pgsgd_session_exchange_end", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("pgsgd_session_exchange_begin_stats", C.c_int, [C.c_void_p, C.c_void_p, u32, u32]),
    ("pgsgd_session_tile_table", i64, [C.c_void_p, P(u64), P(u64), P(u32), P(u32), u64, P(u64)]),
    ("pgsgd_session_tile_lanes", i64, [C.c_void_p, P(u32), u64]),
    ("pgsgd_session_tile_items", i64, [C.c_void_p, P(u32), P(u32), P(u32), P(u32), u64, P(u64)]),
    ("pgsgd_session_trace_tile_terms", i64, [C.c_void_p, u64, C.c_int, u64, u64, P(u64), u64]),
    ("pgsgd_session_set_shard", C.c_int, [C.c_void_p, u32, u32, C.c_int]),
    ("pgsgd_session_exchange_exact_begin", C.c_int, [C.c_void_p, C.c_void_p, u32, u32]),
    ("pgsgd_session_exchange_exact_end", C.c_int, [C.c_void_p, C.c_void_p, u32]),
    ("pgsgd_session_tile_info", C.c_int, [C.c_void_p, P(u64), P(u64), P(u64), P(u32), P(u32)]),
    ("pgsgd_session_split_info", C.c_int, [C.c_void_p, P(u32)]),
    ("pgsgd_session_tile_math", C.c_int, [C.c_void_p]),
    ("pgsgd_debug_tile_displacement", C.c_int, [C.c_int, u64, C.c_float, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float)]),
    ("pgsgd_graph_path_order", C.c_int, [P(GraphView), P(u32), P(f64), P(f64)]),
    ("pgsgd_tile_wave_coin", C.c_int, [u64, u64, u64, u32, u64]),
    ("pgsgd_tile_pair_partner", u32, [u32, u32, u32, u32]),
    ("pgsgd_tile_quad_partner", u32, [u32, u32, u32, u32, u32]),
    ("pgsgd_session_trace_terms", C.c_int, [C.c_void_p, C.c_int, u64, P(u64)]),
    ("pgsgd_graph_from_gfa", C.c_int, [C.c_char_p, C.c_int, P(C.c_void_p)]),
    ("pgsgd_graph_from_og", C.c_int, [C.c_char_p, C.c_int, P(C.c_void_p)]),
    ("pgsgd_graph_load", C.c_int, [C.c_char_p, C.c_int, P(C.c_void_p)]),
    ("pgsgd_graph_load_flags", C.c_int, [C.c_char_p, C.c_int, C.c_uint32, P(C.c_void_p)]),
    ("pgsgd_graph_drop_step_index", C.c_int, [C.c_void_p]),
    ("pgsgd_graph_synthetic", C.c_int, [u64, u64, u64, P(C.c_void_p)]),
    ("pgsgd_graph_free", None, [C.c_void_p]),
    ("pgsgd_graph_get_view", C.c_int, [C.c_void_p, P(GraphView)]),
    ("pgsgd_graph_edge_count", u64, [C.c_void_p]),
    ("pgsgd_graph_edges", P(u64), [C.c_void_p]),
    ("pgsgd_graph_path_name", C.c_char_p, [C.c_void_p, u64]),
    ("pgsgd_graph_max_path_steps", u64, [C.c_void_p]),
    ("pgsgd_weak_components", i64, [u64, P(u64), u64, P(u32)]),
    ("pgsgd_pack_components", C.c_int, [u64, P(u32), u64, P(f64), P(f64)]),
    ("pgsgd_write_tsv", C.c_int, [C.c_char_p, u64, P(u32), u64, P(f64), P(f64)]),
    ("pgsgd_write_lay", C.c_int, [C.c_char_p, u64, P(f64), P(f64)]),
    ("pgsgd_lay_buffer", C.c_int, [u64, P(f64), P(f64), P(P(C.c_uint8)), P(C.c_size_t)]),
    ("pgsgd_read_lay", C.c_int, [C.c_char_p, P(u64), P(P(f64)), P(P(f64))]),
    ("pgsgd_free", None, [C.c_void_p]),
    ("pgsgd_path_stress", C.c_int, [P(GraphView), P(f64), P(f64), u64, u64, P(f64)]),
    ("pgsgd_path_stress_near", C.c_int, [P(GraphView), P(f64), P(f64), C.c_uint32, C.c_double, C.c_uint32, P(f64), P(f64), P(f64), C.c_uint32, P(f64), C.c_uint32, P(f64)]),
    ("pgsgd_path_distance", C.c_int, [P(GraphView), P(f64), P(f64), P(f64), P(f64)]),
    ("pgsgd_sort_params_defaults", C.c_int, [P(GraphView), P(Params)]),
    ("pgsgd_sort_initial", C.c_int, [P(GraphView), P(f64)]),
    ("pgsgd_sort_run", C.c_int, [P(GraphView), P(Params), P(f64), P(Stats)]),
    ("pgsgd_sort_run_targets", C.c_int, [P(GraphView), P(Params), P(C.c_uint8), P(f64), P(Stats)]),
    ("pgsgd_sort_order", C.c_int, [u64, P(f64), P(u64)]),
    ("pgsgd_sort_component_ranks", C.c_int, [u64, P(u64), u64, P(u32)]),
    ("pgsgd_sort_order_components", C.c_int, [u64, P(f64), P(u32), P(u64)]),
    ("pgsgd_sort_write_lay", C.c_int, [P(GraphView), P(f64), P(u64), C.c_char_p]),
    ("pgsgd_sort_stress", C.c_int, [P(GraphView), P(f64), u64, u64, P(f64)]),
    ("pgsgd_sort_trace_terms", C.c_int, [P(GraphView), P(Params), C.c_int, u64, P(u64), P(u32)]),
    ("pgsgd_main_layout", C.c_int, [C.c_int, P(C.c_char_p)]),
]
for _name, _res, _args in SIGNATURES:
    _f = getattr(lib, _name)  # AttributeError here = the library does not export what the header declares
    _f.restype = _res
    _f.argtypes = _args


class PgsgdError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        detail = lib.pgsgd_last_error().decode(errors="replace")  # messages may quote bytes of a damaged file
        super().__init__(f"{where}: {lib.pgsgd_strerror(code).decode()}" + (f" ({detail})" if detail else ""))


def check(code, where):
    if code != 0:
        raise PgsgdError(code, where)
