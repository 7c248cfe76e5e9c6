"""The multi-GPU driver on CPU: world_size-2 gloo processes, oracle-backed engine stub.

The product engine is the HIP session; here an engine with the same interface computes each rank's
share with the CPU oracle, so the test exercises exactly the driver's sharding, stream offsets,
delta all-reduce and stop rule."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


class OracleEngine:
    """Same interface as odgi_amd.distributed.HipEngine, computing on the CPU with the oracle."""

    def __init__(self, og, params, rank, X, Y, n_streams=8):
        from oracle import oracle as orc
        self.orc, self.og, self.params, self.rank, self.n_streams = orc, og, params, rank, n_streams
        c = np.stack([X[0::2], Y[0::2], X[1::2], Y[1::2]], axis=1)
        self.coords = torch.from_numpy(c.astype(np.float32).copy())
        self.calls = 0

    def iteration(self, eta, cooling, n_terms):
        orc = self.orc
        c = self.coords.numpy().copy()
        X = np.empty(2 * len(c)); Y = np.empty(2 * len(c))
        X[0::2], Y[0::2], X[1::2], Y[1::2] = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
        p = self.params
        # iter_max 2 with eta_max == eps keeps the schedule flat at eta; the huge delta stops after iteration 0
        op = orc.params(iter_max=2, iter_with_max_learning_rate=0, min_term_updates=n_terms, delta=1e300,
                        eps=eta, eta_max=eta, theta=p.theta, space=p.space, space_max=p.space_max,
                        space_quantization_step=p.space_quantization_step, cooling_start=0.0 if cooling else 2.0)
        # one iteration at learning rate eta, streams of this rank and this call
        seed = p.seed + 1000 * self.calls
        Xn, Yn = orc.layout_streams_f64(self.og, op, seed, self.n_streams, X, Y, stream_offset=self.rank * self.n_streams)
        self.calls += 1
        out = np.stack([Xn[0::2], Yn[0::2], Xn[1::2], Yn[1::2]], axis=1).astype(np.float32)
        self.coords.copy_(torch.from_numpy(out))
        self._dmax = float(np.abs(out - c).max())

    def iteration_part(self, eta, cooling, n_terms, part, n_parts):
        base, rem = divmod(n_terms, n_parts)
        self.iteration(eta, cooling, base + (1 if part < rem else 0))

    def sync(self):
        return self._dmax

    # the exchange of odgi_amd/csrc/pgsgd_kernels.hpp (exchange_prepare/apply kernels), in numpy
    def new_exchange_buffer(self, world=1):
        return torch.zeros(6 * len(self.coords) + 2 * world, dtype=torch.float32)

    def exchange_mark(self):
        self.base = self.coords.clone()

    def exchange_begin(self, buf, rank=0, world=1):
        n = len(self.coords)
        d = self.coords - self.base
        buf[:4 * n] = d.reshape(-1)
        buf[4 * n:6 * n] = torch.stack([d[:, 0] ** 2 + d[:, 1] ** 2, d[:, 2] ** 2 + d[:, 3] ** 2], dim=1).reshape(-1)
        buf[6 * n:] = 0.0                      # the tail: a slot per rank for max|Delta| and one for the frame-guard flag
        buf[6 * n + rank] = self._dmax

    def exchange_end(self, buf, world):
        self.coords.copy_(merge_rule(self.base, buf, world))
        self.base = self.coords.clone()


def merge_rule(base, buf, world):
    n = len(base)
    S = buf[:4 * n].reshape(n, 4)
    Q = buf[4 * n:6 * n].reshape(n, 2)
    S2 = torch.stack([S[:, 0] ** 2 + S[:, 1] ** 2, S[:, 2] ** 2 + S[:, 3] ** 2], dim=1)
    f = torch.where(S2 > 0, torch.clamp(Q / torch.clamp(S2, min=1e-38), 1.0 / world, 1.0), torch.ones_like(S2))
    return base + S * f.repeat_interleave(2, dim=1)


def _setup(world):
    sys.path.insert(0, ROOT)
    import odgi_amd as oa
    from oracle import oracle as orc
    g = oa.Graph.from_gfa(os.path.join(GOLDEN, "DRB1-3123.gfa"))
    og = orc.Graph.from_product(g)
    p = oa.LayoutParams.defaults(g, iter_max=8, min_term_updates=150001)
    X0, Y0 = oa.initial_layout(g, "d", seed=4)
    return oa, orc, g, og, p, X0, Y0


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oa, orc, g, og, p, X0, Y0 = _setup(world)
        from odgi_amd.distributed import DistributedLayout, shard_terms
        eng = OracleEngine(og, p, rank, X0, Y0)
        drv = DistributedLayout(p, eng)
        assert drv.my_terms() == shard_terms(p.min_term_updates, world, rank)
        n = drv.run()
        np.save(os.path.join(outdir, f"coords_{rank}.npy"), eng.coords.numpy())
        np.save(os.path.join(outdir, f"iters_{rank}.npy"), np.array([n]))
    finally:
        dist.destroy_process_group()


def test_shard_terms():
    from odgi_amd.distributed import shard_terms
    from odgi_amd.distributed import split_blocks
    for n, w in [(20001, 2), (7, 8), (5 * 10 ** 8, 8), (0, 4)]:
        parts = [shard_terms(n, w, r) for r in range(w)]
        assert sum(parts) == n and max(parts) - min(parts) <= 1
        blocks = split_blocks(n, 4)
        assert sum(blocks) == n and max(blocks) - min(blocks) <= 1
        assert sum(shard_terms(b, w, r) for b in blocks for r in range(w)) == n


def test_two_rank_delta_allreduce_matches_single_process_merge(tmp_path):
    world, port = 2, 29611
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    c0 = np.load(tmp_path / "coords_0.npy")
    c1 = np.load(tmp_path / "coords_1.npy")
    assert np.array_equal(c0, c1)                       # every rank ends with the same coordinates
    assert int(np.load(tmp_path / "iters_0.npy")[0]) == 8
    # expected: the same exchanges computed in one process (both rank engines run in turn)
    oa, orc, g, og, p, X0, Y0 = _setup(world)
    from odgi_amd.distributed import DistributedLayout, shard_terms, split_blocks
    engines = [OracleEngine(og, p, r, X0, Y0) for r in range(world)]
    etas = oa.path_linear_sgd_layout_schedule(p)
    cur = engines[0].coords.clone()
    for it in range(p.iter_max):
        for b in range(4):
            total = torch.zeros(6 * len(cur) + 2)
            for r, e in enumerate(engines):
                e.coords.copy_(cur)
                e.base = cur.clone()
                e.iteration_part(etas[it], it >= p.first_cooling_iteration(), shard_terms(p.min_term_updates, world, r), b, 4)
                buf = e.new_exchange_buffer()
                e.exchange_begin(buf)
                total += buf
            cur = merge_rule(cur, total, world)
    assert np.allclose(c0, cur.numpy(), rtol=0, atol=1e-3 * np.abs(cur.numpy()).max())
    # and the merged layout is as good as the one a single rank computes with all the terms
    def stress(c):
        X = np.empty(2 * g.n_nodes); Y = np.empty(2 * g.n_nodes)
        X[0::2], Y[0::2], X[1::2], Y[1::2] = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
        return orc.path_stress_sampled(og, X, Y, 200000)
    single = OracleEngine(og, p, 0, X0, Y0)
    DistributedLayout(p, single).run()
    s_two, s_one, s_init = stress(c0), stress(single.coords.numpy()), orc.path_stress_sampled(og, X0, Y0, 200000)
    assert s_two < 0.01 * s_init and s_two < 1.5 * s_one + 0.05, (s_two, s_one, s_init)


def test_single_process_driver_without_process_group():
    oa, orc, g, og, p, X0, Y0 = _setup(1)
    from odgi_amd.distributed import DistributedLayout
    eng = OracleEngine(og, p, 0, X0, Y0)
    drv = DistributedLayout(p, eng)
    assert drv.world == 1 and drv.my_terms() == p.min_term_updates
    assert drv.run() == 8


class ExactStubEngine:
    """The interface of a tiled HipEngine sharded by region with the exact exchange (shard_mode "regions-exact"), in numpy:
    `words` are 2N 64-bit coordinate words; the launch of colour c moves the words of the blocks (of 64 words) this rank
    owns in that colour by a deterministic function of (iteration, colour, word index, current value) — wrapping 64-bit
    arithmetic, like the device's packed adds — and sends a "far pull" to a word that another rank owns.  What the driver
    must get right: one launch per colour, each followed by one integer all-reduce; tails decoded; no exchange left over."""
    tiled = True

    def __init__(self, n_words, rank, world):
        self.n, self.rank, self.world = n_words, rank, world
        self.words = (np.arange(n_words, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(7)
        self.pending = np.zeros(n_words, dtype=np.uint64)   # far pulls not yet delivered (the outbox)
        self.shard_mode = "terms"
        self.launches = []
        self.far_seen = []

    def set_shard(self, rank, world, by_region=None):
        self.shard_mode = "regions-exact"
        return True

    def warm_per_lane(self):
        return False

    def new_exact_exchange_buffer(self, world=1):
        return torch.zeros(self.n + 3 * world, dtype=torch.int64)

    def new_exchange_buffer(self, world=1):
        return torch.zeros(8, dtype=torch.float32)

    def exchange_mark(self):
        self.base = self.words.copy()

    def iteration_part(self, eta, cooling, n_terms, part, n_parts):
        assert n_parts == 2 and part in (0, 1)
        colour = part
        self.launches.append((float(eta), bool(cooling), colour))
        idx = np.arange(self.n, dtype=np.uint64)
        block = idx // np.uint64(64)
        mine = ((block % np.uint64(2)) == np.uint64(colour)) & (((block // np.uint64(2)) % np.uint64(self.world)) == np.uint64(self.rank))
        k = np.uint64(len(self.launches))
        with np.errstate(over="ignore"):
            step = (self.words * np.uint64(6364136223846793005) + idx * k + np.uint64(colour)) >> np.uint64(40)
            self.words = np.where(mine, self.words + step - np.uint64(1 << 23), self.words)
            # far pulls: every owned word pushes on the word half the array away (someone else's, or the other colour's)
            tgt = (idx + np.uint64(self.n // 2 + 64)) % np.uint64(self.n)
            np.add.at(self.pending, tgt[mine].astype(np.int64), (step[mine] >> np.uint64(3)))
        self._far = int(mine.sum())
        self._dmax = float(eta) * (1.0 + 0.125 * self.rank)

    def exchange_exact_begin(self, buf, rank=0, world=1):
        with np.errstate(over="ignore"):
            self.words = self.words + self.pending       # deliver the launch's far pulls
            self.pending[:] = 0
            delta = self.words - self.base
        out = np.zeros(self.n + 3 * world, dtype=np.uint64)
        out[: self.n] = delta
        out[self.n + rank] = self._far
        out[self.n + world + rank] = np.float32(self._dmax).view(np.uint32)
        buf.copy_(torch.from_numpy(out.view(np.int64)))

    def exchange_exact_end(self, buf, world):
        tot = buf.numpy().view(np.uint64)
        with np.errstate(over="ignore"):
            self.words = self.base + tot[: self.n]
        self.base = self.words.copy()
        self.far_seen.append(int(tot[self.n: self.n + world].sum()))

    def sync(self):
        return self._dmax

    def flush(self):
        assert not self.pending.any()    # an exact exchange delivers its launch's far pulls itself


def _exact_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import odgi_amd as oa
        from odgi_amd.distributed import DistributedLayout
        g = oa.Graph.from_gfa(os.path.join(GOLDEN, "DRB1-3123.gfa"))
        p = oa.LayoutParams.defaults(g, iter_max=5, min_term_updates=1000)
        eng = ExactStubEngine(4096, rank, world)
        drv = DistributedLayout(p, eng)
        assert drv.engine_sharded and eng.shard_mode == "regions-exact"
        dmaxes = [drv.step(it) for it in range(p.iter_max)]
        drv.finish()
        np.save(os.path.join(outdir, f"words_{rank}.npy"), eng.words)
        np.save(os.path.join(outdir, f"dmax_{rank}.npy"), np.array(dmaxes))
        np.save(os.path.join(outdir, f"far_{rank}.npy"), np.array(eng.far_seen))
        assert [c for _, _, c in eng.launches] == [0, 1] * p.iter_max
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_exact_exchange_equals_one_rank(tmp_path, world):
    """DistributedLayout with an engine sharded by region and the exact exchange, world 2 and world 4 over gloo: one launch per
    region colour, one 64-bit integer all-reduce after each, tails decoded (max |Delta| as float bits), nothing exchanged at the
    end — and every rank ends with exactly the words ONE rank owning everything computes."""
    port = 29617 + world
    mp.spawn(_exact_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    w0 = np.load(tmp_path / "words_0.npy")
    for r in range(1, world):
        assert np.array_equal(w0, np.load(tmp_path / f"words_{r}.npy")), r
    sys.path.insert(0, ROOT)
    import odgi_amd as oa
    from odgi_amd.distributed import DistributedLayout
    g = oa.Graph.from_gfa(os.path.join(GOLDEN, "DRB1-3123.gfa"))
    p = oa.LayoutParams.defaults(g, iter_max=5, min_term_updates=1000)
    etas = oa.path_linear_sgd_layout_schedule(p)
    one = ExactStubEngine(4096, 0, 1)
    one.exchange_mark()
    one.shard_mode = "regions-exact"
    buf = one.new_exact_exchange_buffer(1)
    for it in range(p.iter_max):
        for colour in range(2):
            one.iteration_part(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates, colour, 2)
            one.exchange_exact_begin(buf, 0, 1)
            one.exchange_exact_end(buf, 1)
    assert np.array_equal(w0, one.words)
    # every rank saw every rank's max |Delta| (rank 1's is the larger by construction) and the far-pull counts of all ranks
    d0, d1 = np.load(tmp_path / "dmax_0.npy"), np.load(tmp_path / f"dmax_{world - 1}.npy")
    assert np.array_equal(d0, d1) and np.allclose(d0, np.float32(1.0 + 0.125 * (world - 1)) * etas[: p.iter_max].astype(np.float32), rtol=1e-6)
    assert np.array_equal(np.load(tmp_path / "far_0.npy"), np.array(one.far_seen))
