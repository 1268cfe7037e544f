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


class _MockField(torch.nn.Module):
    """stands in for NGPFieldFF on the CPU: a big "table" (reduced in place) and small "weights" (bucketed)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)                                            # identical replicas
        self.table = torch.nn.Parameter(torch.randn(4096, 2) * 0.1)
        self.w = torch.nn.Parameter(torch.randn(3, 2) * 0.5)

    def get_params(self, lr):
        return [{"params": [self.table], "lr": lr}, {"params": [self.w], "lr": lr}]


class _MockRenderer(torch.nn.Module):
    """the surface NGPTrainer.step touches (ngp/render.py: run_cuda, update_extra_state, grid_seed, field) without a GPU: the image
    is a differentiable function of the parameters and the rays; the grid refresh is a deterministic function of (grid_seed,
    iter_density, parameters), like the native one"""

    def __init__(self, poison=None):
        super().__init__()
        self.field = _MockField()
        self.register_buffer("density_bitfield", torch.zeros(64, dtype=torch.uint8))
        self.grid_seed, self.iter_density, self.poison, self.calls = 0, 0, poison, 0

    def run_cuda(self, rays_o, rays_d, bg_color=1, perturb=False, force_all_rays=False, **kw):
        idx = (rays_o[0, :, 0].abs() * 1000).long() % 4096
        feat = self.field.table[idx] * rays_d[0, :, :2]
        image = torch.sigmoid(feat @ self.field.w.t())[None]
        if self.poison is not None and self.calls == self.poison:       # an overflow on THIS rank only
            image = image * float("inf")
        self.calls += 1
        return {"image": image}

    def update_extra_state(self):
        g = torch.Generator().manual_seed(self.grid_seed * 1000 + self.iter_density)
        noise = torch.rand(64, generator=g)
        self.density_bitfield = ((self.field.table.detach()[:64, 0] + noise) > 0.5).to(torch.uint8)
        self.iter_density += 1


def _trainer_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import importlib
    importlib.import_module("nerf-navigation_amd")
    from ngp.train import NGPTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ren = _MockRenderer(poison=7 if rank == 1 else None)
        tr = NGPTrainer(ren, lr=1e-2, iters=100, fp16=True, update_extra_interval=4, seed=5, ema_decay=0.95, steps_per_epoch=6)
        tr.exchange.big = [ren.field.table]                             # 8 K elements: force the in-place route for the table
        tr.exchange.small = [ren.field.w]
        g = torch.Generator().manual_seed(100 + rank)                   # every rank draws its OWN ray batch
        scales, skipped = [], []
        for step in range(20):
            o = torch.rand(1, 256, 3, generator=g)
            d = torch.randn(1, 256, 3, generator=g)
            target = torch.rand(1, 256, 3, generator=g)
            before = ren.field.w.detach().clone()
            loss = tr.step(o, d, target)
            skipped.append(bool(torch.equal(before, ren.field.w.detach())))
            scales.append(float(tr.scaler.get_scale()))
        # plain bytes through the queue: tensors would travel as shared-memory handles that die with this process
        ema = b"".join(s.numpy().tobytes() for s in tr.ema.shadow) + bytes([tr.ema.num_updates])
        out.put((rank, ren.field.table.detach().numpy().tobytes(), ren.field.w.detach().numpy().tobytes(),
                 ren.density_bitfield.numpy().tobytes(), scales, skipped, ren.iter_density, ema))
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_replicas_stay_identical():
    """config 5, rehearsed on CPU: NGPTrainer.step x 20 on two ranks with different ray batches and ONE all-reduce per step.  After
    every step the replicas hold identical parameters; an overflow on rank 1 only (step 7) makes BOTH ranks skip that step and halve
    the loss scale (the inf travels with the summed gradients); the seeded grid refresh leaves identical bitfields."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=90) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, t0, w0, b0, s0, k0, it0, e0), (_, t1, w1, b1, s1, k1, it1, e1) = res
    assert t0 == t1 and w0 == w1 and b0 == b1 and it0 == it1 == 5
    assert e0 == e1 and e0[-1] == 3                                      # the weight averages (3 epochs of 6 steps) are identical without communication
    assert s0 == s1 and k0 == k1
    assert k0[7] and not any(k0[:7]) and not any(k0[8:])                 # the poisoned step is skipped on both ranks, and only that one
    assert s0[7] == 0.5 * s0[6]
    fresh = _MockField()
    assert fresh.w.detach().numpy().tobytes() != w0                                # and training did move the parameters


def _deliver_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import importlib
    importlib.import_module("nerf-navigation_amd")
    from ngp.train import GradExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(50 + rank)
        table = torch.nn.Parameter(torch.zeros(6000, 2))
        ws, wc = torch.nn.Parameter(torch.zeros(7168)), torch.nn.Parameter(torch.zeros(11264))
        other = torch.nn.Parameter(torch.zeros(10))                        # a parameter autograd filled (the after-the-backward route)
        # this rank's (loss-scaled) gradients: the table's as the native scatter writes it for the exchange = half, pre-divided by the world size
        gt = torch.randn(6000, 2, generator=g) * 300.0
        gt[::7] *= 1e-4                                                     # small entries next to large ones
        gw = torch.randn(7168 + 11264, generator=g)
        other.grad = torch.full((10,), float(rank + 1))
        ex = GradExchange([table, ws, wc, other], big_numel=4096)
        assert ex.active() and ex.world_size() == world
        ex.begin_step()
        ex.deliver([ws, wc], (gw / world).clone())
        full = (gt / world).to(torch.float16)
        for lo, hi in ((4000, 6000), (500, 4000), (0, 500), (0, 0)):           # the native scatter's order: finest levels (the last rows) first
            ex.deliver_rows(table, full, lo, hi)
        ex()
        out.put((rank, gt.numpy().tobytes(), gw.numpy().tobytes(), table.grad.numpy().tobytes(), ws.grad.numpy().tobytes(), wc.grad.numpy().tobytes(),
                 other.grad.numpy().tobytes(), str(table.grad.dtype), ex.stats["allreduce_bytes"]))
    finally:
        dist.destroy_process_group()


def test_two_rank_delivered_half_table_gradient_equals_the_float32_exchange_to_half_precision():
    """The overlapped exchange: the native backward delivers the weight bucket and the HALF table gradient, pre-divided by the
    world size; the result is the float32 mean to within half-precision rounding (each rank's share and their sum are rounded to half: <= 1.5 half ulp),
    identical on both ranks, widened to the parameter's dtype; a parameter that autograd filled still takes the after-the-backward route."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_deliver_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = lambda b: np.frombuffer(b, np.float32)                              # noqa: E731
    gt = [f(r[1]) for r in res]
    gw = [f(r[2]) for r in res]
    mean_t, mean_w = (gt[0].astype(np.float64) + gt[1]) / 2, (gw[0].astype(np.float64) + gw[1]) / 2
    for r in res:
        assert r[7] == "torch.float32"
        assert r[3] == res[0][3] and r[4] == res[0][4] and r[5] == res[0][5]            # identical on both ranks
        got = f(r[3]).astype(np.float64)
        # three roundings to half (11 significant bits): each rank's share g_r / 2, then their sum -- half an ulp each, at the magnitude of the
        # value that is rounded (the shares may be larger than their mean when they nearly cancel)
        bound = 2.0 ** -11 * (np.abs(gt[0]) / 2 + np.abs(gt[1]) / 2 + np.abs(mean_t)) + 3 * 2.0 ** -25
        assert np.all(np.abs(got - mean_t) <= 1.01 * bound)
        assert np.abs(got - mean_t).max() > 0                                           # (it IS a half-precision exchange)
        assert np.allclose(np.concatenate([f(r[4]), f(r[5])]), mean_w, rtol=1e-6, atol=1e-7)
        assert np.array_equal(f(r[6]), np.full(10, 1.5, np.float32))
        assert r[8] == 6000 * 2 * 2 + (7168 + 11264) * 4 + 10 * 4                       # half table + float32 bucket + the small float32 bucket


class _FakeScatterLib:
    """libngp_hip's binned-scatter entry points as host functions that fill the output rows they would have written with a value that depends on
    the rank's batch size: enough to run gridencoder.grid.table_gradient_binned's CONTROL FLOW (which collectives, in which order) without a GPU"""

    def __init__(self, offsets):
        self.bounds = [int(v) for v in offsets]

    def ngp_grid_scatter_binned_workspace(self, B, L):
        return 64

    def _value(self, B):
        return 0.0 if B == 0 else float(B % 1000) / 8.0

    def ngp_grid_scatter_binned_phase(self, phase, grad, inputs, offsets, out, B, L, lo, hi, *rest):
        if phase == 2:
            out.tensor[self.bounds[lo]:self.bounds[hi]] = self._value(B) * rest[6]         # out_scale
        return 0

    def ngp_grid_scatter_binned(self, grad, inputs, offsets, out, B, L, *rest):
        out.tensor[:] = self._value(B) * rest[6]
        return 0


def _straddle_worker(rank, world, port, out, sizes, groups):
    sys.path.insert(0, ROOT)
    import importlib
    importlib.import_module("nerf-navigation_amd")
    import ngp_hip
    from gridencoder import grid as G
    from ngp import workload as W
    from ngp.train import GradExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offsets = torch.from_numpy(W.grid_offsets()[0])
        ngp_hip.lib = lambda: _FakeScatterLib(offsets)
        table = torch.nn.Parameter(torch.zeros(int(offsets[-1]), 2))
        ex = GradExchange([table], level_groups=groups)
        B = sizes[rank]
        dummy = torch.empty(1)
        posted = []

        def on_group(full, lo, hi):
            posted.append((lo, hi))
            ex.deliver_rows(table, full, lo, hi)
        ex.begin_step()                                                     # (the rank-invariance check runs here)
        G.table_gradient_binned(dummy, dummy, offsets, B, 16, 8 / 15, 16, 0, False, out_dtype=torch.float16, out_scale=1.0 / world, on_group=on_group,
                                groups=ex.level_groups)
        ex()
        out.put((rank, posted, float(table.grad.min()), float(table.grad.max())))
    finally:
        dist.destroy_process_group()


def _run_straddle(sizes, groups=2):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_straddle_worker, args=(r, world, port, q, sizes, groups)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=90) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_collective_schedule_does_not_depend_on_the_ranks_batch_sizes():
    """B is a rank's OWN sample count.  One rank above the one-pass limit of the grouped scatter (2^22 points) and one below, or one
    rank with an empty batch, must still post the same all-reduces (same row ranges, same order) -- a mismatch hangs RCCL or mixes up gradients."""
    for sizes in ((100, (1 << 22) + 1), (0, 100), ((1 << 22) + 1, 0)):
        res = _run_straddle(sizes)
        assert res[0][1] == res[1][1] and len(res[0][1]) == 2 and res[0][1][0][1] == 6328848 and res[0][1][-1][0] == 0
        want = sum((0.0 if B == 0 else (B % 1000) / 8.0) / 2 for B in sizes)
        for r in res:
            assert r[2] == r[3] == want, (sizes, r)                            # the mean of the two ranks' "gradients" on every row, on both ranks
    res = _run_straddle((5, 7), groups=4)
    assert res[0][1] == res[1][1] and len(res[0][1]) == 4


def _mismatch_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import importlib
    importlib.import_module("nerf-navigation_amd")
    from ngp.train import GradExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = GradExchange([torch.nn.Parameter(torch.zeros(8))], level_groups=2 + rank)
        try:
            ex.begin_step()
            out.put((rank, "no error"))
        except RuntimeError as e:
            out.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_level_group_mismatch_is_an_error_not_a_hang():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("ranks disagree" in msg for _, msg in res), res
