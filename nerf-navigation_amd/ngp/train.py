"""Training step harness for the hot path (BASELINE configs 3 and 5).

Mirrors the optimisation recipe of main_nerf.py:124-135 / nerf/utils.py:774-794 -- Adam(betas 0.9/0.99, eps 1e-15),
lr decayed by 0.1^(step/iters), fp16 autocast with a GradScaler, `update_extra_state` every 16 steps, MSE loss on
4096-ray batches -- and adds what the reference never activated: data-parallel replicas that draw
their own ray batches and exchange gradients with ONE all-reduce per step.

Gradient exchange: the hash-table gradient (12.7 M float32 = 50.6 MB at bound 2) is dense and contiguous, so it is
all-reduced in place -- no bucket copy; the MLP weights (< 80 KB) travel in one small flat bucket.  On 8 MI355X the
xGMI links are point to point (7 x ~153 GB/s per GPU), so a ring moves 2*(7/8)*50.6 MB = 89 MB per GPU per step
(~0.6 ms at one link pair); RCCL picks the algorithm, we only keep the payload in two messages.
The GradScaler's overflow decision is made identical on every replica by reducing the found-inf flag with the gradients
(a replica that skipped a step while the others did not would diverge).
"""
import torch
import torch.distributed as dist

from . import sharding


class GradExchange:
    """Averages `.grad` of the given parameters across ranks: large tensors in place, the rest through one flat bucket."""

    def __init__(self, params, big_numel=1 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.big = [p for p in self.params if p.numel() >= big_numel]
        self.small = [p for p in self.params if p.numel() < big_numel]
        self.bucket = None

    def __call__(self):
        rank, world = sharding.world()
        if not sharding.collectives_on():
            return
        inv = 1.0 / world
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
            p.grad.mul_(inv)
        if self.small:
            n = sum(p.numel() for p in self.small)
            if self.bucket is None or self.bucket.numel() != n or self.bucket.device != self.small[0].device:
                self.bucket = torch.empty(n, dtype=torch.float32, device=self.small[0].device)
            off = 0
            for p in self.small:
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                self.bucket[off:off + p.numel()].copy_(g.reshape(-1))
                off += p.numel()
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM)
            self.bucket.mul_(inv)
            off = 0
            for p in self.small:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(self.bucket[off:off + p.numel()].view_as(p))
                off += p.numel()


class NGPTrainer:
    def __init__(self, renderer, lr=1e-2, iters=30000, fp16=True, update_extra_interval=16, seed=0, fused_adam=None):
        self.ren = renderer
        self.fp16 = fp16
        self.iters = iters
        self.update_extra_interval = update_extra_interval
        # the device type comes from the model so that the 2-rank CPU rehearsal (tests/test_distributed_cpu.py, gloo) runs the very same step
        self.device_type = next(renderer.parameters()).device.type
        # Same update rule as the reference's Adam; on the GPU the fused implementation (one launch per parameter group instead of ~20
        # foreach launches) takes the GradScaler's found-inf flag ON THE DEVICE, so `scaler.step` no longer reads it back: the host is
        # not stalled once per step and can queue the next step's launches behind the running one.
        if fused_adam is None:
            fused_adam = self.device_type == "cuda"
        self.opt = torch.optim.Adam(renderer.field.get_params(lr), betas=(0.9, 0.99), eps=1e-15, fused=bool(fused_adam))
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda it: 0.1 ** min(it / iters, 1))
        self.scaler = torch.amp.GradScaler(self.device_type, enabled=fp16)
        self.exchange = GradExchange(list(renderer.field.parameters()))
        self.global_step = 0
        # the same pcg32 seed on every rank keeps the density grids of the replicas identical
        renderer.grid_seed = int(seed)

    def step(self, rays_o, rays_d, target, bg_color=1, **march):
        """One optimisation step on a [1, N, 3] ray batch; returns the (unscaled) loss as a tensor."""
        ren = self.ren.train()
        if self.global_step % self.update_extra_interval == 0:
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                ren.update_extra_state()
        self.opt.zero_grad(set_to_none=True)
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
            out = ren.run_cuda(rays_o, rays_d, bg_color=bg_color, perturb=True, force_all_rays=False, **march)
            loss = torch.nn.functional.mse_loss(out["image"], target)
        self.scaler.scale(loss).backward()
        self.exchange()                              # gradients are still scaled; the scale is identical on every rank
        self.scaler.step(self.opt)
        self.scaler.update()
        if hasattr(ren.field, "mark_updated"):
            ren.field.mark_updated()                 # parameters changed (fused Adam does not say so through `_version`)
        self.sched.step()
        self.global_step += 1
        return loss.detach()
