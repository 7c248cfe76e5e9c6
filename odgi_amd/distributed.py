"""Multi-GPU path-guided SGD: replicated graph, term-sharded iterations, RCCL all-reduce exchanges.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  Every rank holds the
whole lowered graph and a full copy of the coordinates.  In each iteration (learning-rate step)
rank r applies 1/G of the iteration's terms with its own sampler streams (disjoint stream ids across
ranks), in `exchanges_per_iteration` parts (per-lane kernel: slices of 1/G of the terms; tile kernel:
the tiles with index = part*G + r (mod B*G), each with its whole share of the iteration's terms).
After each part the ranks exchange what they changed since the previous exchange:

    begin : buf[0..4N) = coords - base ; buf[4N..6N) = |delta of each node end|^2 ;
            buf[6N + r] = rank r's max|Delta|, buf[6N + G + r] = its frame-guard flag     (HIP kernels)
    all-reduce(SUM) of the one fused buffer of 6N + 2G floats over the G ranks            (RCCL)
    end   : coords = base + S * clamp(Q/|S|^2, 1/G, 1) per node end ; base = coords      (HIP kernel)

A tiled engine sharded by node region exchanges EXACTLY instead (HipEngine.set_shard: chosen when a launch keeps a
thousand work items per rank): an iteration is one launch per region colour, each followed by

    begin : deliver the launch's far pulls; buf[0..2N) = coords - base as 64-bit integers; tail: far-pull count, max|Delta|
            bits and frame-guard flag in this rank's slots                                                      (HIP kernels)
    all-reduce(SUM) of the 2N + 3G 64-bit integers                                                              (RCCL)
    end   : coords = base = base + sum; the far-pull count over all ranks goes to the learning-rate cap         (HIP kernel)

Windows of one colour are disjoint and integer adds commute: every rank ends with, bit for bit, one GPU's coordinates.

The merge of the other modes (tile shard, per-lane kernel) is not a plain sum: with the early learning rates every term is a full projection
(mu = 1), each rank alone already moves a node end all the way, and summing G such deltas
overshoots G-fold (measured: divergence at G = 2 on DRB1-3123).  The factor f = Q/|S|^2 is 1/G when
the ranks' deltas agree (-> their mean) and 1 when they are uncorrelated small steps (-> their sum);
validated by stress at G = 1,2,4,8.  Every rank's max|Delta| (the reference's stop rule) and
frame-guard flag ride in a slot of their own behind the 6N floats — the other ranks write zero there — so
the one SUM hands all of them to everybody: ONE collective per exchange, none besides.  The reference has no multi-device path (src/cuda/layout.cu is single-GPU,
its NCCLCHECK macro is unused), so this exchange is new; SURVEY 8(e).
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from ._lib import lib, check
from .layout import LayoutParams, LayoutSession, path_linear_sgd_layout_schedule


def shard_terms(n_terms, world_size, rank):
    """Terms of one block owned by `rank`: the first n_terms % G ranks take one extra."""
    base, rem = divmod(int(n_terms), int(world_size))
    return base + (1 if rank < rem else 0)


def split_blocks(n_terms, blocks):
    """Terms of one iteration split into `blocks` exchange blocks (sizes differ by at most 1)."""
    base, rem = divmod(int(n_terms), int(blocks))
    return [base + (1 if b < rem else 0) for b in range(blocks)]


class HipEngine:
    """The product engine: a LayoutSession launching on torch's current stream, so the RCCL
    all-reduce of the exchange buffer (a torch tensor) is ordered with the kernels."""

    def __init__(self, graph, params: LayoutParams, X, Y):
        self.session = LayoutSession(graph, params)
        self.session.upload(X, Y)
        self.session.use_torch_stream()
        self.n_nodes = graph.n_nodes
        self.tiled = bool(self.session.tile_info()["tiled"])
        self.device = torch.device("cuda", torch.cuda.current_device() if params.device < 0 else params.device)

    def new_exchange_buffer(self, world=1):
        return torch.empty(6 * self.n_nodes + 2 * world, dtype=torch.float32, device=self.device)

    def iteration(self, eta, cooling, n_terms):
        self.session.iteration(eta, cooling, n_terms)

    def iteration_part(self, eta, cooling, n_terms, part, n_parts):
        self.session.iteration_part(eta, cooling, n_terms, part, n_parts)

    def sync(self):
        return self.session.sync()

    def flush(self):
        self.session.flush()

    def warm_per_lane(self):
        return bool(self.session.tile_info()["warm_per_lane"])

    def set_shard(self, rank, world, by_region=None):
        """by_region None: by node region with the exact exchange when that leaves a launch a thousand WINDOWS per
        rank (fixed-point coordinates; the merge rule otherwise), by tile when it does not — decided by the session, the
        same rule the C++ driver uses; True / False force region (merge rule) / tile; "exact" forces region with the exact
        exchange (pgsgd_session_set_shard).  True when the engine shards by tile or by region itself (tile kernel):
        iteration() then takes the full term count of a block; False when the caller must shard the term count
        (per-lane kernel)."""
        if by_region is None:
            mode = -1   # the session's own rule (pgsgd_session_set_shard), the one `odgi layout --gpus N` (pgsgd_multi.cpp) uses too
        else:
            mode = 2 if by_region == "exact" else 1 if by_region else 0
        rc = lib.pgsgd_session_set_shard(self.session._h, int(rank), int(world), mode)
        if rc < 0:
            check(rc, "set_shard")
        self.shard_mode = {0: "terms", 1: "tiles", 2: "regions", 3: "regions-exact"}[rc]
        return rc > 0

    def new_exact_exchange_buffer(self, world=1):
        return torch.zeros(2 * self.n_nodes + 3 * world, dtype=torch.int64, device=self.device)

    def exchange_exact_begin(self, buf, rank=0, world=1):
        check(lib.pgsgd_session_exchange_exact_begin(self.session._h, C.c_void_p(buf.data_ptr()), int(rank), int(world)), "exchange_exact_begin")

    def exchange_exact_end(self, buf, world):
        check(lib.pgsgd_session_exchange_exact_end(self.session._h, C.c_void_p(buf.data_ptr()), int(world)), "exchange_exact_end")

    def exchange_mark(self):
        check(lib.pgsgd_session_exchange_mark(self.session._h), "exchange_mark")

    def exchange_begin(self, buf, rank=0, world=1):
        check(lib.pgsgd_session_exchange_begin_stats(self.session._h, C.c_void_p(buf.data_ptr()), int(rank), int(world)), "exchange_begin")

    def exchange_end(self, buf, world):
        check(lib.pgsgd_session_exchange_end(self.session._h, C.c_void_p(buf.data_ptr()), int(world)), "exchange_end")

    def reframe(self):
        self.session.reframe()

    def snapshot(self):
        """Coordinates as a snapshot between iterations sees them (path_sgd_layout.cpp:379-408)."""
        return self.session.download(flush=False)

    def result(self):
        return self.session.download()

    def close(self):
        self.session.close()


class DistributedLayout:
    """Drives one engine per rank through the schedule with the exchange between blocks."""

    def __init__(self, params: LayoutParams, engine, group=None, exchanges_per_iteration=None, region_shard=None, tile_shard=True,
                 snapshot_prefix=None, force_exchange=False):
        self.params = params
        self.engine = engine
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.backend = dist.get_backend(group) if self.distributed else None
        # force_exchange: a one-rank process group still runs the exchange (prepare, all-reduce, merge: the identity up to
        # a quantum) — how the RCCL path is executed on a single-GPU box
        self.exchanging = self.world > 1 or (bool(force_exchange) and self.distributed)
        self.snapshot_prefix = snapshot_prefix
        if exchanges_per_iteration is None:
            # measured with G virtual ranks on one MI355X (profiles/r01/virtual_ranks_*.jsonl): the per-lane
            # kernel (small graphs) needs 4 exchanges per iteration to keep G = 1 quality, the tile kernel is
            # insensitive between 1 and 4 (stress within 5-17 % of G = 1 either way) and every exchange is a
            # 24 MB all-reduce against ~3 ms of kernels per iteration and rank at G = 8: one per iteration
            exchanges_per_iteration = 1 if self.world == 1 else (1 if getattr(engine, "tiled", False) else 4)
        self.blocks = max(1, int(exchanges_per_iteration)) if self.world > 1 else 1
        self.etas = path_linear_sgd_layout_schedule(params)
        self.first_cooling = int(math.floor(params.cooling_start * float(params.iter_max)))
        self.iterations_done = 0
        self.stopped_early = False
        self._buf = None
        self._ebuf = None
        # Every rank applies 1/G of each iteration's terms.  Per-lane kernel: 1/G of the term count, in B
        # slices.  Tile kernel: part b of an iteration runs the tiles with index = b*G + rank (mod B*G), each
        # with its whole share of the iteration's terms, so a visited tile always has a full term loop and
        # the cost of an iteration grows neither with the number of exchanges nor with G; or (region shard)
        # every G-th node region with all its tiles: the ranks' private windows are then disjoint, which keeps
        # the one-GPU layout quality, but a launch has only N/2RG work items — chosen by the engine when that
        # is still a thousand per rank (BASELINE config 5 at G = 8), region_shard=True/False forces it.
        self.engine_sharded = False
        if self.exchanging:
            self._buf = engine.new_exchange_buffer(self.world)
            engine.exchange_mark()
            # (a one-rank group under force_exchange shards too — trivially — so that a single-GPU box executes the very
            # collectives of a multi-rank run, the integer all-reduce of the exact exchange included)
            if hasattr(engine, "set_shard") and (region_shard or (tile_shard and getattr(engine, "tiled", False))):
                self.engine_sharded = bool(engine.set_shard(self.rank, self.world, by_region=region_shard))

    def _iteration_terms(self):
        """Term count this rank passes for one iteration: everything when the engine shards by tile or
        region (only the owned tiles' share is applied), 1/G of it otherwise."""
        M = self.params.min_term_updates
        return M if self.engine_sharded else shard_terms(M, self.world, self.rank)

    def my_terms(self):
        """Terms this rank applies per iteration (for an engine that shards itself: its expected share)."""
        if self.engine_sharded:
            return self.params.min_term_updates // self.world
        return shard_terms(self.params.min_term_updates, self.world, self.rank)

    def _exchange_exact(self):
        """The exchange of a region-sharded engine after one colour's launch: 64-bit integer deltas, one SUM all-reduce;
        returns (max|Delta| over the ranks, any frame-guard flag) from its tail."""
        import struct
        eng, G = self.engine, self.world
        if self._ebuf is None:
            self._ebuf = eng.new_exact_exchange_buffer(G)
        eng.exchange_exact_begin(self._ebuf, self.rank, G)
        dist.all_reduce(self._ebuf, op=dist.ReduceOp.SUM, group=self.group)
        eng.exchange_exact_end(self._ebuf, G)
        tail = self._ebuf[-2 * G:].tolist()   # (waits for the stream: the exchange's one synchronisation)
        dmax = max(struct.unpack("<f", struct.pack("<I", int(v) & 0xffffffff))[0] for v in tail[:G])
        return dmax, any(v != 0 for v in tail[G:])

    def _exchange(self):
        """One fused all-reduce; returns (max|Delta| over the ranks, any frame-guard flag) from its tail."""
        eng, G = self.engine, self.world
        eng.exchange_begin(self._buf, self.rank, G)
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
        eng.exchange_end(self._buf, G)
        tail = self._buf[-2 * G:].tolist()   # (waits for the stream: the step's one synchronisation)
        return max(tail[:G]), any(v != 0.0 for v in tail[G:])

    def step(self, it):
        """Iteration `it` (0-based) on every rank with its exchanges.  Returns global max|Delta|."""
        eng = self.engine
        dmax, guard = 0.0, False
        # a tiled engine whose initial layout had no global structure runs the per-lane kernel until cooling: that
        # phase takes the per-lane kernel's four exchanges per iteration
        cooling = it >= self.first_cooling
        blocks = self.blocks
        if self.world > 1 and not cooling and getattr(eng, "warm_per_lane", lambda: False)():
            blocks = max(blocks, 4)
        exact = self.exchanging and getattr(eng, "shard_mode", "") == "regions-exact" and not (not cooling and getattr(eng, "warm_per_lane", lambda: False)())
        if exact:
            # region shard with the exact exchange: one colour, its far pulls delivered, integer deltas summed over the ranks —
            # twice per iteration; every rank then holds what one GPU would (no merge rule)
            for colour in range(2):
                eng.iteration_part(self.etas[it], cooling, self._iteration_terms(), colour, 2)
                d, g = self._exchange_exact()
                dmax, guard = max(dmax, d), guard or g
            blocks = 0
        for b in range(blocks):
            eng.iteration_part(self.etas[it], cooling, self._iteration_terms(), b, blocks)
            if self.exchanging:
                d, g = self._exchange()
                dmax, guard = max(dmax, d), guard or g
            else:
                dmax = max(dmax, eng.sync())
        if self.exchanging:
            eng.sync()
            # when any rank saw a coordinate in the outer quarter of the fixed-point frame, every rank widens its
            # frame before the next iteration (the flags came with the exchange: the same decision everywhere)
            if guard and hasattr(eng, "reframe"):
                eng.reframe()
        self.iterations_done = it + 1
        return dmax

    def finish(self):
        """The far pulls of every rank's last tile launch: delivered, then merged like any other move."""
        eng = self.engine
        if hasattr(eng, "flush"):
            eng.flush()   # (nothing left to deliver after an exact exchange: it drained the launch's far pulls itself)
            if self.exchanging and getattr(eng, "shard_mode", "") != "regions-exact":
                self._exchange()
                eng.sync()

    def run(self):
        """The whole schedule with the reference's stop rules (path_sgd_layout.cpp:139-149) and, on rank 0, its
        snapshots (:379-408: `prefix + k` after iteration k, k = 1 .. iter_max - 1)."""
        p = self.params
        for it in range(p.iter_max):
            dmax = self.step(it)
            if it + 1 >= p.iter_max:
                break
            if dmax <= p.delta:
                self.stopped_early = True
                break
            if self.snapshot_prefix and self.rank == 0 and hasattr(self.engine, "snapshot"):
                from .layout import Layout
                X, Y = self.engine.snapshot()
                Layout(X, Y).serialize(f"{self.snapshot_prefix}{it + 1}")
        self.finish()
        return self.iterations_done
