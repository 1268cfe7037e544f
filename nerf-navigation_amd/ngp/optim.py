"""The optimiser half of the training step as native launches: `torch.optim.Adam` + `torch.cuda.amp.GradScaler` the way the reference drives them
(main_nerf.py:126 `Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)`; nerf/utils.py:329 `GradScaler(enabled=fp16)`; :789-791
`scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`).

Both are torch library code in the reference, not its own; they are restated natively because on this path they are 0.2 ms of a 1.55 ms step
(csrc/adam.hip has the accounting): one pass that looks for non-finite gradients, one thread that takes GradScaler's decisions ON THE DEVICE (the host
never reads found_inf back), one pass that updates p, exp_avg, exp_avg_sq and writes the float16 copy of the parameters the next autocast forward
reads.  Arithmetic, skip rule, scale recurrence and the state_dict layouts are torch's (tests/test_gpu_adam.py steps both side by side).

There is no CPU path: on a CPU model (the gloo rehearsal) ngp/train.py uses torch's own classes."""
import ctypes

import torch

import ngp_hip as _hip


class NativeAdam(torch.optim.Optimizer):
    """Drop-in for `torch.optim.Adam(params, lr, betas, eps)` (weight_decay 0, amsgrad off: the reference's settings) on float32 CUDA parameters,
    with the loss scaling of `torch.amp.GradScaler(init_scale, growth_factor, backoff_factor, growth_interval, enabled)` built in:

        loss = opt.scale_loss(loss); loss.backward(); opt.step()          # = scaler.scale(loss).backward(); scaler.step(opt); scaler.update()

    `param_groups` / `state` / `state_dict()` have torch.optim.Adam's layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so LR schedulers and the
    reference's checkpoints (nerf/utils.py:950-957, :1040-1060) work unchanged; `scaler_state_dict()` has GradScaler.state_dict()'s.

    Two restrictions against torch's class (`NGPTrainer` falls back to torch.optim.Adam + GradScaler where they would bite):
      * at most `ngp_hip.ADAM_MAX_TENSORS` (16) parameter tensors per optimiser (one launch takes them all; the reference's fields have 3 or 6);
      * ONE step count for all parameters: every parameter is updated in every unskipped step, as in the reference's training loop, where every
        parameter receives a gradient in every step.  torch keeps a count per parameter, so a parameter that receives its first gradient later would get another
        bias correction there; `load_state_dict` refuses a state whose per-parameter counts differ."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, scaler_enabled=True, init_scale=2.0 ** 16, growth_factor=2.0,
                 backoff_factor=0.5, growth_interval=2000):
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"NativeAdam: invalid hyper-parameters lr={lr} betas={betas} eps={eps}")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                        differentiable=False, fused=None)
        super().__init__(params, defaults)
        self.scaler_enabled = bool(scaler_enabled)
        self.init_scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval)
        self._dev_state = None               # float32 [32] on the parameters' device (include/ngp_hip.h NGP_ADAM_STATE_*)
        self._pending = None                 # (scale, growth_tracker, step) loaded before the device is known
        self.half_mirrors = {}               # parameter -> float16 tensor of the same shape that every step refreshes (ngp/field.py half_mirrors())
        n = sum(len(g["params"]) for g in self.param_groups)
        if n > _hip.ADAM_MAX_TENSORS:
            raise ValueError(f"NativeAdam: {n} parameter tensors, the native step takes {_hip.ADAM_MAX_TENSORS} per launch")
        for g in self.param_groups:
            for p in g["params"]:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise ValueError("NativeAdam: parameters must be contiguous float32 CUDA tensors (there is no CPU path)")

    # ---- device state --------------------------------------------------------------------------------------------------------------------
    def _state_on(self, device):
        if self._dev_state is None or self._dev_state.device != device:
            old = self._pending if self._dev_state is None else self._read_state()
            s = torch.zeros(_hip.ADAM_STATE_WORDS, dtype=torch.float32, device=device)
            scale, tracker, step = old if old is not None else (self.init_scale, 0, 0)
            s[_hip.ADAM_STATE_SCALE] = scale
            i = s.view(torch.int32)
            i[_hip.ADAM_STATE_GROWTH_TRACKER] = int(tracker)
            i[_hip.ADAM_STATE_STEP] = int(step)
            self._dev_state, self._pending = s, None
        return self._dev_state

    def _read_state(self):
        """(scale, growth_tracker, step) -- synchronises"""
        if self._dev_state is None:
            return self._pending if self._pending is not None else (self.init_scale, 0, 0)
        host = self._dev_state.cpu()
        i = host.view(torch.int32)
        return float(host[_hip.ADAM_STATE_SCALE]), int(i[_hip.ADAM_STATE_GROWTH_TRACKER]), int(i[_hip.ADAM_STATE_STEP])

    def _device(self):
        return self.param_groups[0]["params"][0].device

    # ---- GradScaler's surface ------------------------------------------------------------------------------------------------------------
    def scale_loss(self, loss):
        """GradScaler.scale(): loss * scale, the scale staying on the device"""
        if not self.scaler_enabled:
            return loss
        return loss * self._state_on(loss.device)[_hip.ADAM_STATE_SCALE]

    def get_scale(self):
        return self._read_state()[0] if self.scaler_enabled else 1.0

    def found_inf(self):
        """1.0 if the last step() met a non-finite gradient and left the parameters alone (a device tensor: reading it synchronises)"""
        return self._state_on(self._device())[_hip.ADAM_STATE_FOUND_INF]

    def step_count(self):
        return self._read_state()[2]

    def scaler_state_dict(self):
        """torch.amp.GradScaler.state_dict()'s keys (nerf/utils.py:956 stores it as checkpoint['scaler'])"""
        if not self.scaler_enabled:
            return {}
        scale, tracker, _ = self._read_state()
        return {"scale": scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": tracker}

    def load_scaler_state_dict(self, state):
        if not self.scaler_enabled or not state:
            return
        self.growth_factor, self.backoff_factor = float(state["growth_factor"]), float(state["backoff_factor"])
        self.growth_interval = int(state["growth_interval"])
        _, _, step = self._read_state()
        self._write_state(float(state["scale"]), int(state["_growth_tracker"]), step)

    def _write_state(self, scale, tracker, step):
        if self._dev_state is None:
            self._pending = (scale, tracker, step)
            return
        self._dev_state[_hip.ADAM_STATE_SCALE] = scale
        i = self._dev_state.view(torch.int32)
        i[_hip.ADAM_STATE_GROWTH_TRACKER] = int(tracker)
        i[_hip.ADAM_STATE_STEP] = int(step)

    # ---- torch.optim.Adam's state_dict layout ------------------------------------------------------------------------------------------------
    def state_dict(self):
        step = float(self._read_state()[2])
        for st in self.state.values():
            if "step" in st:
                st["step"] = torch.tensor(step, dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = {int(float(st["step"])) for st in self.state.values() if "step" in st}
        if len(steps) > 1:
            raise ValueError(f"NativeAdam.load_state_dict: the parameters carry different step counts {sorted(steps)}; the native step keeps one")
        scale, tracker, _ = self._read_state()
        self._write_state(scale, tracker, steps.pop() if steps else 0)
        for st in self.state.values():               # torch casts state to the parameter's dtype/device; keep `step` a CPU scalar like torch.optim.Adam's default
            if "step" in st:
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32).cpu()

    # ---- the step --------------------------------------------------------------------------------------------------------------------------
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError("NativeAdam.step: closures are not supported (GradScaler.step does not support them either)")
        return self.native_step()

    @torch.no_grad()
    def native_step(self):
        """step() without the wrappers torch.optim.Optimizer and the LR scheduler put around it (profiler ranges, hooks: ~40 us of host time per call)"""
        betas, eps = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        rows, keep = [], []
        for g in self.param_groups:
            if tuple(g["betas"]) != tuple(betas) or g["eps"] != eps or g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                raise RuntimeError("NativeAdam: one (betas, eps) for all groups, weight_decay 0, amsgrad / maximize off (the reference's settings)")
            for p in g["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                if grad.dtype != torch.float32 or grad.device != p.device or grad.is_sparse:
                    raise RuntimeError("NativeAdam: gradients must be dense float32 tensors on the parameter's device")
                grad = grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                half = self.half_mirrors.get(p)
                if half is not None and (half.dtype != torch.float16 or half.shape != p.shape or half.device != p.device or not half.is_contiguous()):
                    raise RuntimeError("NativeAdam: a half mirror must be a contiguous float16 tensor of the parameter's shape on its device")
                rows.append((p, grad, st["exp_avg"], st["exp_avg_sq"], half, float(g["lr"])))
                keep.append(grad)
        if not rows:
            return None
        device = rows[0][0].device
        if any(r[0].device != device for r in rows):
            raise RuntimeError("NativeAdam: all parameters on one device")
        tensors = (_hip.ngp_adam_tensor_t * len(rows))()
        for t, (p, grad, m, v, half, lr) in zip(tensors, rows):
            t.param, t.grad, t.exp_avg, t.exp_avg_sq = p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr()
            t.half_copy = half.data_ptr() if half is not None else None
            t.n, t.lr = p.numel(), lr
        hyper = _hip.ngp_adam_hyper_t(float(betas[0]), float(betas[1]), float(eps), self.growth_factor, self.backoff_factor, self.growth_interval,
                                      int(self.scaler_enabled))
        state = self._state_on(device)
        with _hip.timed("adam_step"):
            _hip.check(_hip.lib().ngp_adam_step(ctypes.cast(tensors, ctypes.c_void_p), len(rows), ctypes.cast(ctypes.pointer(hyper), ctypes.c_void_p),
                                                _hip.ptr(state), _hip.stream()), "adam_step")
        del keep
        return None


class NativeScaler:
    """The `trainer.scaler` of a trainer whose optimiser is a NativeAdam: GradScaler's reading surface over the optimiser's device state"""

    def __init__(self, opt):
        self.opt = opt

    def is_enabled(self):
        return self.opt.scaler_enabled

    def scale(self, loss):
        return self.opt.scale_loss(loss)

    def get_scale(self):
        return self.opt.get_scale()

    def state_dict(self):
        return self.opt.scaler_state_dict()

    def load_state_dict(self, state):
        self.opt.load_scaler_state_dict(state)
