"""Multi-GPU path-guided SGD: replicated graph, term-sharded iterations, one all-reduce per eta step.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  Every rank holds the
whole lowered graph and a full copy of the coordinates.  In each iteration (learning-rate step)
rank r applies its 1/G share of the iteration's terms with its own sampler streams
(stream ids r*L .. r*L+L-1, disjoint across ranks), then the ranks exchange what they changed:

    delta_r = coords_r - coords_start          (fp32, 4N values)
    all-reduce(sum) over ranks                 (one fused buffer: 16 MB at N = 1e6)
    coords  = coords_start + sum_r delta_r

which is what one GPU running all G shares with atomic adds would have produced, up to the
staleness of not seeing the other ranks' updates inside the iteration.  The early-stop quantity
max|Delta| is all-reduced with MAX.  The reference has no multi-device path (src/cuda/layout.cu is
single-GPU, its NCCLCHECK macro is unused), so this exchange is new; SURVEY 8(e).
"""
import math

import torch
import torch.distributed as dist

from .layout import LayoutParams, LayoutSession, path_linear_sgd_layout_schedule


def shard_terms(n_terms, world_size, rank):
    """Terms of one iteration owned by `rank`: the first n_terms % G ranks take one extra."""
    base, rem = divmod(int(n_terms), int(world_size))
    return base + (1 if rank < rem else 0)


class HipEngine:
    """The product engine: a LayoutSession whose coordinate buffer is a torch tensor."""

    def __init__(self, graph, params: LayoutParams, X, Y):
        self.session = LayoutSession(graph, params)
        self.session.upload(X, Y)
        self.coords = self.session.coords_tensor()
        self.session.use_torch_stream()

    def iteration(self, eta, cooling, n_terms):
        self.session.iteration(eta, cooling, n_terms)

    def sync(self):
        return self.session.sync()

    def result(self):
        c = self.coords.detach().cpu().numpy()
        X = c[:, [0, 2]].reshape(-1).copy()
        Y = c[:, [1, 3]].reshape(-1).copy()
        return X, Y

    def close(self):
        self.session.close()


class DistributedLayout:
    """Drives one engine per rank through the schedule with the delta all-reduce between eta steps."""

    def __init__(self, params: LayoutParams, engine, group=None):
        self.params = params
        self.engine = engine
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.etas = path_linear_sgd_layout_schedule(params)
        self.first_cooling = int(math.floor(params.cooling_start * float(params.iter_max)))
        self._start = torch.empty_like(engine.coords)
        self.iterations_done = 0
        self.stopped_early = False

    def my_terms(self):
        return shard_terms(self.params.min_term_updates, self.world, self.rank)

    def step(self, it):
        """Iteration `it` (0-based) on every rank, then the exchange.  Returns global max|Delta|."""
        eng = self.engine
        coords = eng.coords
        if self.world > 1:
            self._start.copy_(coords)
        eng.iteration(self.etas[it], it >= self.first_cooling, self.my_terms())
        if self.world > 1:
            # delta in place: coords <- coords - start; all-reduce; coords <- start + sum
            coords.sub_(self._start)
            dist.all_reduce(coords, op=dist.ReduceOp.SUM, group=self.group)
            coords.add_(self._start)
        dmax = eng.sync()
        if self.world > 1:
            t = torch.tensor([dmax], dtype=torch.float64, device=coords.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            dmax = float(t.item())
        self.iterations_done = it + 1
        return dmax

    def run(self):
        """The whole schedule with the reference's stop rules (path_sgd_layout.cpp:139-149)."""
        p = self.params
        for it in range(p.iter_max):
            dmax = self.step(it)
            if it + 1 >= p.iter_max:
                break
            if dmax <= p.delta:
                self.stopped_early = True
                break
        return self.iterations_done
