"""CPU, 2 processes, gloo: the N > 1 plumbing of bench.py (ngp/sharding.py) -- view assignment, throughput reduction
(sum of samples, max of time), row-band gather.  The data path itself has no collective (rays shard, model replicas)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import importlib
    importlib.import_module("nerf-navigation_amd")
    from ngp import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert sharding.world() == (rank, world)
        views = sharding.pose_indices(rank, world, 4)
        # each rank "renders" its views: samples proportional to the view id, time proportional to the rank
        samples = sum(1000 + v for v in views)
        total, t_max = sharding.reduce_throughput(samples, 1.0 + rank, torch.device("cpu"))
        lo, hi = sharding.row_band(rank, world, 24)
        band = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
        full = sharding.gather_rows(band)
        out.put((rank, views, total, t_max, full.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    views = [r[1] for r in res]
    assert sorted(views[0] + views[1]) == list(range(8)) and not set(views[0]) & set(views[1])
    want_total = float(sum(1000 + v for v in range(8)))
    for rank, _, total, t_max, full in res:
        assert total == want_total and t_max == 2.0                  # sum over ranks / max over ranks, identical on every rank
        assert full == [[float(i)] * 3 for i in range(24)]           # bands reassemble into the full image in row order


def _grad_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import importlib
    importlib.import_module("nerf-navigation_amd")
    from ngp.train import GradExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        table = torch.nn.Parameter(torch.zeros(5000, 2))             # "big": reduced in place
        w1 = torch.nn.Parameter(torch.zeros(64, 32))
        w2 = torch.nn.Parameter(torch.zeros(16))
        frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
        table.grad = torch.full_like(table, float(rank + 1))
        w1.grad = torch.arange(64 * 32, dtype=torch.float32).view(64, 32) * (rank + 1)
        w2.grad = None                                               # a parameter that got no gradient on this rank
        ex = GradExchange([table, w1, w2, frozen], big_numel=4096)
        ex()
        out.put((rank, float(table.grad.mean()), w1.grad.clone(), w2.grad.clone(), len(ex.big), len(ex.small)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_exchange():
    """config 5's only data-path collective: gradients averaged over ranks, the table in place, the MLP in one bucket"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = torch.arange(64 * 32, dtype=torch.float32).view(64, 32)
    for rank, tmean, w1g, w2g, nbig, nsmall in res:
        assert tmean == 1.5 and nbig == 1 and nsmall == 2            # (1 + 2) / 2
        assert torch.equal(w1g, base * 1.5) and torch.equal(w2g, torch.zeros(16))
