"""Training step harness for the hot path (BASELINE configs 3 and 5).

Mirrors the optimisation recipe of main_nerf.py:124-135 / nerf/utils.py:774-794 -- Adam(betas 0.9/0.99, eps 1e-15),
lr decayed by 0.1^(step/iters), fp16 autocast with a GradScaler, `update_extra_state` every 16 steps, MSE loss on
4096-ray batches, an exponential moving average of the weights (decay 0.95) that evaluation renders with -- and adds what
the reference never activated: data-parallel replicas that draw their own ray batches and exchange
gradients once per step.

Gradient exchange (`GradExchange`).  Payload per step and rank: the hash-table gradient (12.7 M values) + the MLP weights
(18 K values).  Two routes:
  * the field's native training step (ngp/field.py `_field_train`) DELIVERS its gradients to the exchange while the backward
    is still running: the MLP bucket (74 KB) as soon as the field's backward kernels are queued -- its all-reduce runs
    underneath the table scatter --, and the table gradient as the HALF tensor the scatter writes, pre-divided by the world
    size (25.3 MB instead of 50.6 MB of float32; the sum is then the mean and cannot overflow where a single rank's
    gradient did not), all-reduced asynchronously right behind the scatter.  `finish()` waits and widens into `.grad`.
    On 8 MI355X the xGMI links are point to point (7 x ~153 GB/s per GPU): a ring moves 2*(7/8)*25.3 MB = 44 MB per GPU per
    step over one link pair (~0.3 ms), a direct reduce-scatter + all-gather 3.2 MB per link per phase; RCCL picks.
  * anything else (the op-by-op graph, other fields, the CPU rehearsal) is reduced after the backward: large tensors in
    place, the rest through one flat bucket.
The GradScaler's overflow decision is identical on every replica by construction: the all-reduce sums the still-scaled
gradients, so an inf or NaN on one rank is one on all ranks (a replica that skipped a step while the others did not would
diverge).
"""
import contextlib
import ctypes

import torch
import torch.distributed as dist

from . import sharding


class GradExchange:
    """Averages the gradients of `params` across ranks.  `deliver(param, grad)` starts the all-reduce of a gradient that a
    native backward produced out of band (`grad` already divided by the world size, any float dtype); `__call__()` after the
    backward reduces what autograd left in `.grad` and completes the delivered ones.  `stats` counts the bytes and (with
    `timing=True`, CUDA only) the host-visible wait of the last step."""

    def __init__(self, params, big_numel=1 << 20, timing=False, level_groups=None):
        """level_groups: in how many pieces (groups of levels, finest first) the native scatter hands the table gradient over (gridencoder/grid.py
        table_gradient_binned).  It shapes the sequence of all-reduces, so it must be the same on every rank: the first active step checks that
        (`_check_rank_invariants`).  Default: NGP_LEVEL_GROUPS of the launching environment, else 2."""
        import os
        self.level_groups = int(os.environ.get("NGP_LEVEL_GROUPS", "2")) if level_groups is None else int(level_groups)
        self._checked = False
        self.params = [p for p in params if p.requires_grad]
        self.big = [p for p in self.params if p.numel() >= big_numel]
        self.small = [p for p in self.params if p.numel() < big_numel]
        self.bucket = None
        self.pending = []                  # (params, delivered tensor, work handle)
        self.partial = {}                  # id(param) -> (param, full gradient tensor, work handles of its row groups)
        self.timing = timing
        self.stats = {"allreduce_bytes": 0, "exposed_ms": None}
        self._events = None

    # ---- out-of-band route -------------------------------------------------------------------------------------------
    def active(self):
        return sharding.collectives_on()

    def world_size(self):
        return sharding.world()[1]

    def deliver(self, params, grad):
        """grad: ONE contiguous tensor holding this rank's gradients of `params` (a parameter or a list of them, in order), ALREADY divided by
        the world size, any float dtype; starts its SUM all-reduce and returns at once"""
        params = [params] if isinstance(params, torch.Tensor) else list(params)
        assert grad.numel() == sum(p.numel() for p in params)
        work = dist.all_reduce(grad, op=dist.ReduceOp.SUM, async_op=True)
        self.stats["allreduce_bytes"] += grad.numel() * grad.element_size()
        self.pending.append((params, grad, work))

    def deliver_rows(self, param, full, row_lo, row_hi):
        """rows [row_lo, row_hi) of `full` (this rank's whole gradient of `param`, pre-divided by the world size, any float dtype) are final: start their
        all-reduce.  The native scatter hands the table over one group of levels at a time (gridencoder.grid.table_gradient_binned(on_group=...)), so
        the collective of one group runs while the next is still being summed; only the last, smallest group (the coarsest levels) is exposed."""
        if row_hi <= row_lo:
            return
        piece = full[row_lo:row_hi]
        work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True)
        self.stats["allreduce_bytes"] += piece.numel() * piece.element_size()
        entry = self.partial.setdefault(id(param), (param, full, []))
        assert entry[1] is full, "deliver_rows: one gradient tensor per parameter and step"
        entry[2].append(work)

    def _finish_pending(self):
        delivered = set()
        for param, full, works in self.partial.values():
            for work in works:
                work.wait()
            g = full.view_as(param)
            param_grad = g.to(param.dtype) if g.dtype != param.dtype else g
            if param.grad is None:
                param.grad = param_grad
            else:
                param.grad.copy_(param_grad)
            delivered.add(id(param))
        self.partial = {}
        for params, grad, work in self.pending:
            work.wait()                    # on the GPU: makes the current stream wait for the collective; the host does not block
            flat, off = grad.reshape(-1), 0
            for p in params:
                g = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                g = g.to(p.dtype) if g.dtype != p.dtype else (g if len(params) == 1 else g.clone())
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.copy_(g)
                delivered.add(id(p))
        self.pending = []
        return delivered

    # ---- after the backward ------------------------------------------------------------------------------------------
    def __call__(self):
        rank, world = sharding.world()
        if not sharding.collectives_on():
            self.pending, self.partial = [], {}
            return
        if self.timing and torch.cuda.is_available():
            self._events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._events[0].record()
        done = self._finish_pending()
        inv = 1.0 / world
        for p in self.big:
            if id(p) in done:
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
            self.stats["allreduce_bytes"] += p.grad.numel() * p.grad.element_size()
            p.grad.mul_(inv)
        small = [p for p in self.small if id(p) not in done]
        if small:
            n = sum(p.numel() for p in small)
            if self.bucket is None or self.bucket.numel() != n or self.bucket.device != small[0].device:
                self.bucket = torch.empty(n, dtype=torch.float32, device=small[0].device)
            off = 0
            for p in small:
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                self.bucket[off:off + p.numel()].copy_(g.reshape(-1))
                off += p.numel()
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM)
            self.stats["allreduce_bytes"] += n * 4
            self.bucket.mul_(inv)
            off = 0
            for p in small:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(self.bucket[off:off + p.numel()].view_as(p))
                off += p.numel()
        if self._events is not None:
            self._events[1].record()

    def begin_step(self):
        self.stats["allreduce_bytes"] = 0
        if not self._checked and self.active():
            self._check_rank_invariants()

    def _check_rank_invariants(self):
        """once, collectively: everything that decides WHICH all-reduces a step posts must agree across ranks -- the level-group count and the parameter
        sizes.  A mismatch (say NGP_LEVEL_GROUPS set for one process only) would otherwise surface as a hang inside RCCL."""
        dev = self.params[0].device if self.params else torch.device("cpu")
        mine = torch.tensor([self.level_groups, len(self.params), sum(p.numel() for p in self.params)], dtype=torch.int64, device=dev)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError(f"GradExchange: ranks disagree on (level_groups, parameter count, parameter elements): min {lo.tolist()} max {hi.tolist()}")
        self._checked = True

    def exposed_ms(self):
        """GPU time between the end of the backward and the last gradient being ready (events on the compute stream): what the step
        waits for the collectives.  None when timing is off."""
        if self._events is None:
            return None
        torch.cuda.synchronize()
        return self._events[0].elapsed_time(self._events[1])


class _Ctx:
    """What the body of a torch.autograd.Function asks of its `ctx`, for calling `forward` / `backward` directly: NGPTrainer's direct step runs the SAME
    function bodies as the autograd graph, in the order the engine would, without building the graph (five Function.apply + the engine are ~0.35 ms of host
    time per step, and the step is host-bound)."""
    __slots__ = ("saved_tensors", "needs_input_grad", "__dict__")

    def __init__(self, needs_input_grad=()):
        self.saved_tensors, self.needs_input_grad = (), tuple(needs_input_grad)

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


def _body(fn):
    """the undecorated forward / backward of a Function (custom_fwd / custom_bwd keep it as __wrapped__): the direct step runs outside autocast on float32
    operands, which is what those decorators arrange"""
    return getattr(fn, "__wrapped__", fn)


class _LeanLambdaLR(torch.optim.lr_scheduler.LambdaLR):
    """main_nerf.py:131 `LambdaLR(optimizer, lambda iter: 0.1 ** min(iter / opt.iters, 1))` with a `step()` that only does the arithmetic (the stock one
    spends 0.1 ms per call on bookkeeping: a tenth of this training step's host time).  Same state, same state_dict, same learning rates."""

    def step(self, epoch=None):
        if epoch is not None:
            return super().step(epoch)
        self._step_count += 1
        self.last_epoch += 1
        lrs = [base * fn(self.last_epoch) for base, fn in zip(self.base_lrs, self.lr_lambdas)]
        for group, lr in zip(self.optimizer.param_groups, lrs):
            group["lr"] = lr
        self._last_lr = lrs


class _mse_head(torch.autograd.Function):
    """loss = mean((pred - target)^2) (nerf/utils.py:450,480) and loss * scale (:789 scaler.scale(loss)) in ONE launch that also leaves the
    gradient with respect to pred for a unit incoming gradient; backward is one more launch (csrc/train_head.hip).  Returns (loss, scaled loss);
    differentiate the scaled one.  `scale`: a 1-element float32 device tensor or None."""
    _ws = {}

    @staticmethod
    def forward(ctx, pred, target, scale):
        import ngp_hip as _hip
        p, t = pred.contiguous(), target.contiguous().to(torch.float32)
        dev = p.device
        ws = _mse_head._ws.get(dev)
        if ws is None:
            ws = _mse_head._ws[dev] = torch.zeros(int(_hip.lib().ngp_mse_head_workspace()), dtype=torch.uint8, device=dev)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        unit = torch.empty_like(p)
        _hip.check(_hip.lib().ngp_mse_head_forward(_hip.ptr(p), _hip.ptr(t), p.numel(), _hip.ptr(scale), _hip.ptr(loss), _hip.ptr(unit), _hip.ptr(ws),
                                                   ws.numel(), _hip.stream()), "mse_head_forward")
        ctx.save_for_backward(unit, scale)
        ctx.shape = pred.shape
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g_loss, g_scaled):
        import ngp_hip as _hip
        unit, scale = ctx.saved_tensors
        g = torch.empty_like(unit)
        gl = g_loss.contiguous().float() if g_loss is not None else None
        gs = g_scaled.contiguous().float() if g_scaled is not None else None
        _hip.check(_hip.lib().ngp_mse_head_backward(_hip.ptr(unit), _hip.ptr(gl), _hip.ptr(gs), _hip.ptr(scale), unit.numel(), _hip.ptr(g),
                                                    _hip.stream()), "mse_head_backward")
        return g.view(ctx.shape), None, None


class WeightEMA:
    """Exponential moving average of the parameters, as the reference's Trainer keeps it (nerf/utils.py:324-325: torch_ema's
    ExponentialMovingAverage(model.parameters(), decay=0.95); updated once per epoch :814-815 and per GUI burst :683-684; evaluation and the
    "best" checkpoint render with the averaged weights :723-733, :851-853, :983-994).  torch_ema is a third-party package (environment.yml:24,
    unpinned; absent from this image): its published algorithm (torch_ema 0.3, ema.py) is restated here --
        update():  num_updates += 1;  d = min(decay, (1 + num_updates) / (10 + num_updates));  shadow -= (1 - d) * (shadow - param)
        store() / copy_to() / restore():  park the live parameters, install the shadow ones, put the live ones back
    -- parity unpinned by reference fixtures (the package cannot be imported here); tests/test_train_host.py checks the recurrence against
    its closed form.  Replicas hold identical parameters, so their averages are identical without communication."""

    def __init__(self, parameters, decay=0.95, use_num_updates=True):
        self.params = [p for p in parameters if p.requires_grad]
        self.decay = float(decay)
        self.num_updates = 0 if use_num_updates else None
        self.shadow = [p.detach().clone() for p in self.params]
        self.collected = None

    @torch.no_grad()
    def update(self):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        one_minus = 1.0 - decay
        tmp = torch._foreach_sub(self.shadow, [p.detach() for p in self.params])      # (shadow - param) * (1 - d), subtracted: torch_ema's order
        torch._foreach_mul_(tmp, one_minus)
        torch._foreach_sub_(self.shadow, tmp)

    @torch.no_grad()
    def store(self):
        self.collected = [p.detach().clone() for p in self.params]

    @torch.no_grad()
    def copy_to(self):
        for p, s in zip(self.params, self.shadow):
            p.copy_(s)

    @torch.no_grad()
    def restore(self):
        if self.collected is None:
            raise RuntimeError("WeightEMA.restore() without a store()")
        for p, c in zip(self.params, self.collected):
            p.copy_(c)
        self.collected = None

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow, "collected_params": self.collected}

    def load_state_dict(self, state):
        self.decay, self.num_updates = state["decay"], state["num_updates"]
        with torch.no_grad():
            for s, v in zip(self.shadow, state["shadow_params"]):
                s.copy_(v)


class NGPTrainer:
    def __init__(self, renderer, lr=1e-2, iters=30000, fp16=True, update_extra_interval=16, seed=0, fused_adam=None, ema_decay=0.95,
                 steps_per_epoch=None, time_exchange=False, native_adam=None, direct=True):
        """ema_decay: None disables the average (main_nerf.py:135 passes 0.95).  steps_per_epoch: the reference updates the average once per
        epoch = once per pass over the training views (nerf/utils.py:814-815); with a number here `step()` calls `end_epoch()` itself.
        native_adam: Adam + GradScaler as three native launches (ngp/optim.py, csrc/adam.hip); default on a GPU model, never on a CPU one
        (the gloo rehearsal runs torch's classes: same recurrences, same state_dict layouts).
        direct: run forward, loss and backward of the default GPU configuration (FFMLP field, fp16, native optimiser, scalar background) as straight-line
        calls of the autograd functions' bodies instead of through the autograd engine (`_direct_forward_backward`); False = always autograd."""
        self.direct, self._one = bool(direct), None
        self.ren = renderer
        self.fp16 = fp16
        self.iters = iters
        self.update_extra_interval = update_extra_interval
        self.steps_per_epoch = steps_per_epoch
        # the device type comes from the model so that the 2-rank CPU rehearsal (tests/test_distributed_cpu.py, gloo) runs the very same step
        self.device_type = next(renderer.parameters()).device.type
        # Same update rule as the reference's Adam; on the GPU the fused implementation (one launch per parameter group instead of ~20
        # foreach launches) takes the GradScaler's found-inf flag ON THE DEVICE, so `scaler.step` no longer reads it back: the host is
        # not stalled once per step and can queue the next step's launches behind the running one.
        if fused_adam is None:
            fused_adam = self.device_type == "cuda"
        if native_adam is None:
            # the native launch takes up to ADAM_MAX_TENSORS float32 tensors (csrc/adam.hip); a field with more of them (deeper nn.Linear stacks, a
            # background network) keeps torch's Adam + GradScaler, as before the native optimiser existed
            import ngp_hip as _hip
            field_params = list(renderer.field.parameters())
            native_adam = (self.device_type == "cuda" and len(field_params) <= _hip.ADAM_MAX_TENSORS
                           and all(p.dtype == torch.float32 and p.is_contiguous() for p in field_params))
        self.native_adam = bool(native_adam)
        if self.native_adam:
            # ... and the native one (default): the non-finite check, GradScaler's decisions and recurrence, Adam and the float16 copy of the
            # parameters the next forward reads, in three launches with nothing read back (0.20 -> 0.09 ms of the step, ngp/optim.py)
            from .optim import NativeAdam, NativeScaler
            self.opt = NativeAdam(renderer.field.get_params(lr), betas=(0.9, 0.99), eps=1e-15, scaler_enabled=fp16)
            self.scaler = NativeScaler(self.opt)
        else:
            self.opt = torch.optim.Adam(renderer.field.get_params(lr), betas=(0.9, 0.99), eps=1e-15, fused=bool(fused_adam))
            self.scaler = torch.amp.GradScaler(self.device_type, enabled=fp16)
        self.sched = _LeanLambdaLR(self.opt, lambda it: 0.1 ** min(it / iters, 1))
        self.exchange = GradExchange(list(renderer.field.parameters()), timing=time_exchange)
        self.ema = WeightEMA(renderer.field.parameters(), decay=ema_decay) if ema_decay is not None else None
        self.global_step = 0
        # the same pcg32 seed on every rank keeps the density grids of the replicas identical
        renderer.grid_seed = int(seed)

    # ---- the direct step: the autograd graph's function bodies, called in order ---------------------------------------------------------------
    def _direct_applies(self, ren, field, rays_o, rays_d, target, bg_color, march):
        """The direct step calls the undecorated bodies, so nothing casts or checks the operands for it (custom_fwd(cast_inputs=float32) and the autograd
        route's `image.shape == target.shape` do that elsewhere): every operand must already be what the kernels read -- float32, on the GPU, rays_d shaped
        like rays_o, one target colour per ray.  Anything else takes the autograd route, which casts or raises."""
        from .field import NGPFieldFF
        p = getattr(getattr(field, "encoder", None), "embeddings", None)
        return (self.direct and self.native_adam and self.fp16 and isinstance(field, NGPFieldFF) and field.fused_training and ren.cuda_ray
                and getattr(ren, "bg_radius", -1) <= 0 and rays_o.is_cuda and rays_o.dtype == torch.float32 and target.dtype == torch.float32
                and rays_d.is_cuda and rays_d.dtype == torch.float32 and rays_d.shape == rays_o.shape and rays_o.shape[-1] == 3
                and target.is_cuda and target.numel() == rays_o.numel() and target.shape[-1] == 3
                and isinstance(bg_color, (int, float)) and p is not None and p.requires_grad and p.dtype == torch.float32
                and field.sigma_net.weights.requires_grad and field.color_net.weights.requires_grad and field._fused_shape_ok()
                and set(march) <= {"dt_gamma", "max_steps"} and rays_o.numel() > 0)

    @torch.no_grad()
    def _direct_forward_backward(self, ren, field, rays_o, rays_d, target, bg_color, dt_gamma=0, max_steps=1024):
        """`run_cuda`'s training branch (ngp/render.py, nerf/renderer.py:282-321), the loss and the whole backward as straight-line calls of the function
        bodies the autograd route runs (raymarching._march_rays_train, field._field_train, raymarching._composite_rays_train; render._mix_background and
        _mse_head, forward and backward, as the one launch ngp_train_head_direct): same values in the same order; the gradients end up in `.grad` (or with the exchange, as there).  Returns the loss."""
        import raymarching
        from raymarching import raymarching as RM
        from .field import _field_train
        o, d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        nears, fars = raymarching.near_far_from_aabb(o, d, ren._aabb(), ren.min_near)
        counter = ren.step_counter[ren.local_step % 16]
        counter.zero_()
        ren.local_step += 1
        xyzs, dirs, deltas, rays = _body(RM._march_rays_train.forward)(_Ctx(), o, d, ren.bound, ren.density_bitfield, ren.cascade, ren.grid_size, nears, fars,
                                                                      counter, ren.mean_count, True, 128, False, dt_gamma, max_steps)
        emb, w_s, w_c = field.encoder.embeddings, field.sigma_net.weights, field.color_net.weights
        c_field = _Ctx((False, False, True, True, True, False))
        sigmas, rgbs = _body(_field_train.forward)(c_field, xyzs, dirs, emb, w_s, w_c, field)
        scale_sigma = ren.density_scale != 1
        c_comp = _Ctx((True, True, False, False))
        weights_sum, depth, image = _body(RM._composite_rays_train.forward)(c_comp, ren.density_scale * sigmas if scale_sigma else sigmas, rgbs, deltas, rays)
        # background mix, loss, and the loss's backward down to the compositor's two incoming gradients in ONE launch (the values of _mix_background and
        # _mse_head run forward and backward, bit for bit); the same launch clears what the compositor's backward expects zeroed
        import ngp_hip as _hip
        L, dev, M, N = _hip.lib(), image.device, xyzs.shape[0], weights_sum.shape[0]
        mixed = torch.empty(N, 3, dtype=torch.float32, device=dev)
        both = torch.empty(2, dtype=torch.float32, device=dev)
        g_image, g_ws = torch.empty(N, 3, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev)
        sig_in = c_comp.saved_tensors[0]
        g_sig, g_rgb = torch.empty_like(sig_in), torch.empty(sig_in.shape[0], 3, dtype=torch.float32, device=dev)
        clear = [(g_sig, g_sig.numel() * 4), (g_rgb, g_rgb.numel() * 4)]
        clear = [(t, b) for t, b in clear if b > 0 and t.data_ptr() % 16 == 0 and b % 16 == 0]
        if len(clear) < 2:                           # (an empty batch, or a size that is not a multiple of 16 bytes: the torch fills)
            g_sig.zero_(); g_rgb.zero_()
            clear = []
        ws_head = _mse_head._ws.get(dev)
        if ws_head is None:
            ws_head = _mse_head._ws[dev] = torch.zeros(int(L.ngp_mse_head_workspace()), dtype=torch.uint8, device=dev)
        zp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t, _ in clear], *([None] * (3 - len(clear))))
        zb = (ctypes.c_uint64 * 3)(*[b for _, b in clear], *([0] * (3 - len(clear))))
        tgt = target.contiguous().view(-1)
        _hip.check(L.ngp_train_head_direct(_hip.ptr(weights_sum), _hip.ptr(image), None, 0, float(bg_color), _hip.ptr(tgt),
                                           _hip.ptr(self.opt._state_on(dev)[:1]), N, _hip.ptr(mixed), _hip.ptr(both), _hip.ptr(g_image), _hip.ptr(g_ws),
                                           ctypes.cast(zp, ctypes.c_void_p), ctypes.cast(zb, ctypes.c_void_p), len(clear), _hip.ptr(ws_head), ws_head.numel(),
                                           _hip.stream()), "train_head_direct")
        # ---- the rest of the backward, in the engine's order ----
        c_comp.cleared_grads = (g_sig, g_rgb)
        g_sig, g_rgb, _, _ = _body(RM._composite_rays_train.backward)(c_comp, g_ws, None, g_image)
        if scale_sigma:
            g_sig = g_sig * ren.density_scale
        grads = _body(_field_train.backward)(c_field, g_sig, g_rgb)
        if grads[2] is not None:                     # without a gradient exchange: what autograd's AccumulateGrad would do on cleared gradients
            emb.grad, w_s.grad, w_c.grad = grads[2].view_as(emb), grads[3].view_as(w_s), grads[4].view_as(w_c)
        field._param_epoch += 1                      # the post-accumulate-grad hooks of the autograd route (ngp/field.py _watch_parameters, gridencoder drop_half_table)
        field._mirror_ok = False
        emb._ngp_half = None
        return both[0]

    def step(self, rays_o, rays_d, target, bg_color=1, **march):
        """One optimisation step on a [1, N, 3] ray batch; returns the (unscaled) loss as a tensor."""
        ren = self.ren if self.ren.training else self.ren.train()
        if self.global_step % self.update_extra_interval == 0:
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                ren.update_extra_state()
        for group in self.opt.param_groups:              # = opt.zero_grad(set_to_none=True) without its profiler range and foreach bookkeeping (30 us per step)
            for p in group["params"]:
                p.grad = None
        self.exchange.begin_step()
        field = ren.field
        if hasattr(field, "grad_sink"):
            field.grad_sink = self.exchange if self.exchange.active() else None     # the native backward hands its gradients over as they appear
        try:
            if self._direct_applies(ren, field, rays_o, rays_d, target, bg_color, march):
                loss = self._direct_forward_backward(ren, field, rays_o, rays_d, target, bg_color, **march)
            else:
                with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                    out = ren.run_cuda(rays_o, rays_d, bg_color=bg_color, perturb=True, force_all_rays=False, **march)
                    image = out["image"]
                    if self.native_adam and image.is_cuda and image.dtype == torch.float32 and image.numel() > 0 and image.shape == target.shape:
                        # loss, scaled loss and d loss / d image in one launch (the scale stays on the device: ngp/optim.py)
                        scale = self.opt._state_on(image.device)[:1] if self.fp16 else None
                        loss, scaled = _mse_head.apply(image, target, scale)
                    else:
                        loss = torch.nn.functional.mse_loss(image, target)
                        scaled = None
                (scaled if scaled is not None else self.scaler.scale(loss)).backward()
        finally:
            if hasattr(field, "grad_sink"):
                field.grad_sink = None
        self.exchange()                              # gradients are still scaled; the scale is identical on every rank
        if self.native_adam:
            # the update launch also writes the half copies the field's native kernels read (otherwise the next forward converts the table again)
            mirrored = self.fp16 and getattr(field, "fused_training", False) and hasattr(field, "half_mirrors") and field._fused_shape_ok()
            self.opt.half_mirrors = field.half_mirrors() if mirrored else {}
            self.opt.native_step()                   # = scaler.step(opt) + scaler.update()
            if mirrored:
                field.mirrors_are_current()
            elif hasattr(field, "mark_updated"):
                field.mark_updated()
        else:
            self.scaler.step(self.opt)
            self.scaler.update()
            if hasattr(field, "mark_updated"):
                field.mark_updated()                 # parameters changed (fused Adam does not say so through `_version`)
        self.sched.step()
        self.global_step += 1
        if self.steps_per_epoch and self.global_step % self.steps_per_epoch == 0:
            self.end_epoch()
        return loss.detach()

    def state_dict(self):
        """the optimisation state under the keys of the reference's full checkpoint (nerf/utils.py:944-958), each in its class's own layout"""
        state = {"global_step": self.global_step, "optimizer": self.opt.state_dict(), "lr_scheduler": self.sched.state_dict(),
                 "scaler": self.scaler.state_dict()}
        if self.ema is not None:
            state["ema"] = self.ema.state_dict()
        return state

    def load_state_dict(self, state):
        """nerf/utils.py:1036-1060 (there every part is optional and a failure to load one is a warning; here a part that is present must load)"""
        self.global_step = int(state.get("global_step", self.global_step))
        if "optimizer" in state:
            self.opt.load_state_dict(state["optimizer"])
        if "lr_scheduler" in state:
            self.sched.load_state_dict(state["lr_scheduler"])
        if "scaler" in state and state["scaler"]:
            self.scaler.load_state_dict(state["scaler"])
        if self.ema is not None and "ema" in state:
            self.ema.load_state_dict(state["ema"])
        field = self.ren.field
        if hasattr(field, "mark_updated"):
            field.mark_updated()

    def end_epoch(self):
        """nerf/utils.py:814-815: the moving average follows the weights once per epoch"""
        if self.ema is not None:
            self.ema.update()

    @contextlib.contextmanager
    def eval_weights(self):
        """`with trainer.eval_weights(): renderer.render_fused(...)` renders with the averaged weights (nerf/utils.py:851-853, :933-934) and puts
        the live ones back afterwards; without an average it is a no-op"""
        field = self.ren.field
        if self.ema is None:
            yield
            return
        self.ema.store()
        self.ema.copy_to()
        if hasattr(field, "mark_updated"):
            field.mark_updated()
        try:
            yield
        finally:
            self.ema.restore()
            if hasattr(field, "mark_updated"):
                field.mark_updated()
