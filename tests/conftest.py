import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the library reads its test and experiment knobs (PGSGD_TILE_*, PGSGD_OUTBOX_*, PGSGD_MULTI_*, PGSGD_FRAME_SPAN) only in a
# process that sets PGSGD_DEBUG=1 (odgi_amd/csrc/pgsgd_internal.hpp: debug_env); a test that sets none of them runs
# the product path as a user gets it
os.environ["PGSGD_DEBUG"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # native pieces are built in-tree; build them once if a fresh checkout has none
    lib = os.path.join(ROOT, "odgi_amd", "lib", "libpgsgd.so")
    orc = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    import __graft_entry__
    if not (os.path.exists(lib) and os.path.exists(orc)):
        __graft_entry__.build()
    elif __graft_entry__.source_id() != __graft_entry__.built_id():
        # never test a library older than its sources: rebuild where a compiler exists, fail loudly where not
        try:
            __graft_entry__.build()
        except Exception as e:  # noqa: BLE001
            raise pytest.UsageError(f"libpgsgd.so is stale (built from {__graft_entry__.built_id()}, sources are "
                                    f"{__graft_entry__.source_id()}) and cannot be rebuilt here: {e}")


# GPU run order.  The driver runs `pytest -x -q -m gpu`: the first failure hides everything behind it, so tests run in
# order of DETERMINISM, not of headline — every bit-exact test first (a failure there is a defect, never scatter), then
# deterministic properties (accounting, checksums, conservation, plans), then statistical bands (mean of three GPU runs
# against COMMITTED CPU distributions / curves — nothing on the CPU side is re-rolled on the GPU box), then whatever goes
# through a subprocess (CLI, torchrun, the C++ shim) last.  Lower = earlier; unnamed tests get 25 (deterministic properties).
GPU_ORDER = [
    # --- bit-exact against the oracle, golden vectors or another route
    (0, "test_tile_kernel_one_workgroup_one_lane_is_bit_exact_with_oracle_mirror"),
    (1, "test_tiled_kernel_terms_bit_exact_and_tile_table"),
    (2, "test_sampler_"),
    (3, "test_one_stream_run_is_bit_exact"),
    (4, "test_1d_sampler_streams_bit_exact"),
    (4, "test_1d_one_stream_run_bit_exact"),
    (4, "test_1d_target_nodes_bit_exact_and_frozen"),
    (4, "test_reference_unit_test_paths_one_node_long"),
    (5, "test_region_shard_with_the_exact_exchange_is_one_gpu_bit_for_bit"),
    (5, "test_exchange_kernels_match_the_merge_rule_word_for_word"),
    (5, "test_cpp_multi_gpu_run_with_the_exact_exchange"),
    (6, "test_double_precision_download_is_exact"),
    (6, "test_fixed_point_frame"),
    (7, "test_tile_kernel_fast_math"),
    (8, "test_single_step_and_ragged_paths"),
    (8, "test_step_positions_built_on_the_device"),
    (30, "test_two_pass_iterations_of_small_lane_bound_graphs"),   # bit-exact parts + a GPU-vs-GPU band
    # --- deterministic properties: accounting, checksums, conservation, kernel plans (BASELINE configs 4 and 5 by size)
    (20, "test_synthetic_million_node_properties"),
    (21, "test_config5_size_properties"),
    (22, "test_tiled_kernel_with_unsorted_stretches"),
    (22, "test_tiled_kernel_with_tandem_repeats"),
    (22, "test_outbox_overflow"),
    (22, "test_pending_far_pulls"),
    # --- statistical: committed yardsticks, two-sided bands
    (40, "test_tile_kernel_against_the_reference_rule_at_config4"),
    (41, "test_tiled_kernel_matches_per_lane_kernel_and_oracle"),
    (42, "test_full_layout_stress_matches_cpu_oracle"),
    (43, "test_reference_layout_quality_bar"),
    (43, "test_reference_fixture_statistics_two_sided_on_gpu"),
    (44, "test_hilbert_init_theta_sweep_and_cooling"),
    (45, "test_many_paths"),
    (46, "test_1d_two_pass_iterations"),
    (46, "test_1d_layout_and_order_match_oracle"),
    (47, "test_tile_kernel_conflict_resolution"),
    (50, "test_tile_sharded_virtual_ranks"),
    (50, "test_virtual_rank_stress_band"),
    (50, "test_cpp_multi_gpu_run_with_two_virtual_devices"),
    (54, "test_config5_size_schedules_against_the_committed_cpu_points"),
    (55, "test_config5_size_whole_schedule_against_the_per_lane_kernel"),
    # --- subprocesses: CLI, the C++ shim, torchrun
    (90, "test_cli_"),
    (90, "test_reference_signature_shim"),
    (90, "test_cpp_multi_gpu_run_writes_snapshots"),
    (90, "test_cpp_multi_gpu_driver_executes_rccl_with_one_rank"),
    (90, "test_bench_runs_the_rccl_exchange"),
]


def _gpu_rank(item):
    for rank, prefix in GPU_ORDER:
        if item.name.startswith(prefix):
            return rank
    return 25


def _native_fingerprint(path):
    import hashlib
    import time
    try:
        st = os.stat(path)
        with open(path, "rb") as f:
            h = hashlib.sha256(f.read()).hexdigest()[:16]
        return f"{os.path.relpath(path, ROOT)} sha256:{h} {st.st_size} B built {time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime(st.st_mtime))}Z"
    except OSError as e:
        return f"{os.path.relpath(path, ROOT)} MISSING ({e})"


def pytest_report_header(config):
    """Which native objects this run loads: a stale .so on the GPU box must be visible in the log.  The hash of
    the sources the library was built from is stored next to it by the Makefile (lib/BUILD_ID) and compared."""
    lines = ["native: " + _native_fingerprint(os.path.join(ROOT, "odgi_amd", "lib", "libpgsgd.so")),
             "native: " + _native_fingerprint(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))]
    import __graft_entry__
    want, have = __graft_entry__.source_id(), __graft_entry__.built_id()
    lines.append(f"native: sources {want} / built from {have}" + ("" if want == have else "  ** STALE BUILD **"))
    return lines


def pytest_collection_modifyitems(config, items):
    import torch
    items.sort(key=_gpu_rank)   # stable: file order inside a rank
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
