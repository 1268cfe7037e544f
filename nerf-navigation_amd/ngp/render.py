"""Renderer harness over the drop-in ops: the counterpart of nerf/renderer.py's NeRFRenderer.

Three render paths:
  run_cuda(...)      the reference's occupancy-grid path, op by op (nerf/renderer.py:257-379): the parity anchor and
                     what a torch-ngp caller gets from the drop-in packages;
  render_fused(...)  one launch per frame (csrc/render_fused.hip) -- the MI355X-native fast path for inference;
  run(...)           the fixed-step path used by the nav loop (nerf/renderer.py:125-254, simulate.py:163-166).
plus update_extra_state / mark_untrained_grid (density-grid maintenance, nerf/renderer.py:381-537) on the native ops of
csrc/density_grid.hip.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

import ngp_hip as _hip
import raymarching
from gridencoder import grid as _grid


def sample_pdf(bins, weights, n_samples, det=False, generator=None):
    """nerf/renderer.py:12-46: draw n_samples per ray from the piecewise-constant pdf `weights` [B, T-1] over the intervals of
    `bins` [B, T] by inverting its CDF; det=True uses the midpoints of n_samples equal probability slices."""
    pdf = weights + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=cdf.device).expand(cdf.shape[0], n_samples)
    else:
        u = torch.rand(cdf.shape[0], n_samples, device=cdf.device, generator=generator)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return b_lo + (u - c_lo) / span * (b_hi - b_lo)


class _mix_background(torch.autograd.Function):
    """nerf/renderer.py:318-319 (and :370-371 after the inference loop) in one launch each way (csrc/train_head.hip):
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color;   depth = torch.clamp(depth - nears, min=0) / (fars - nears)
    bg: a Python number, or a float32 tensor [3] / [N,3] that needs no gradient.  Backward: grad_weights_sum = -(g . bg), the gradient of `image` is g
    itself; `depth`'s is dropped, as the compositor drops it (raymarching/raymarching.py:270: grad_depth is never used)."""

    @staticmethod
    def forward(ctx, weights_sum, depth, image, nears, fars, bg):
        N = weights_sum.shape[0]
        ws, dp, im = weights_sum.contiguous(), depth.contiguous(), image.contiguous()
        if isinstance(bg, torch.Tensor):
            bg_t = bg.to(torch.float32).contiguous()
            bg_rows, bg_value = (1 if bg_t.numel() == 3 else N), 0.0
        else:
            bg_t, bg_rows, bg_value = None, 0, float(bg)
        out_image = torch.empty(N, 3, dtype=torch.float32, device=ws.device)
        out_depth = torch.empty(N, dtype=torch.float32, device=ws.device)
        _hip.check(_hip.lib().ngp_train_mix_forward(_hip.ptr(ws), _hip.ptr(dp), _hip.ptr(im), _hip.ptr(nears.contiguous()), _hip.ptr(fars.contiguous()),
                                                    _hip.ptr(bg_t), bg_rows, bg_value, N, _hip.ptr(out_image), _hip.ptr(out_depth), _hip.stream()),
                   "train_mix_forward")
        ctx.bg = (bg_t, bg_rows, bg_value, N)
        ctx.mark_non_differentiable(out_depth)
        return out_image, out_depth

    @staticmethod
    def backward(ctx, g_image, g_depth):
        bg_t, bg_rows, bg_value, N = ctx.bg
        g_image = g_image.contiguous().float()
        g_ws = torch.empty(N, dtype=torch.float32, device=g_image.device)
        _hip.check(_hip.lib().ngp_train_mix_backward(_hip.ptr(g_image), _hip.ptr(bg_t), bg_rows, bg_value, N, _hip.ptr(g_ws), _hip.stream()),
                   "train_mix_backward")
        return g_ws, None, g_image, None, None, None


def mix_background(weights_sum, depth, image, nears, fars, bg_color):
    """the two lines that end both branches of run_cuda; one native launch when the operands allow it, the reference's torch ops otherwise (a background
    MODEL's colours need their own gradient; CPU tensors)"""
    native = (weights_sum.is_cuda and weights_sum.dtype == torch.float32 and image.dtype == torch.float32 and depth.dtype == torch.float32
              and weights_sum.dim() == 1 and image.shape == (weights_sum.shape[0], 3))
    if isinstance(bg_color, torch.Tensor):
        native = native and not bg_color.requires_grad and bg_color.device == weights_sum.device and tuple(bg_color.shape) in ((3,), (weights_sum.shape[0], 3))
    else:
        native = native and isinstance(bg_color, (int, float))
    if native:
        return _mix_background.apply(weights_sum, depth, image, nears, fars, bg_color)
    image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    depth = torch.clamp(depth - nears, min=0) / (fars - nears)
    return image, depth


class NGPRenderer(nn.Module):
    def __init__(self, field, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=0.01, grid_size=128, bg_radius=-1):
        super().__init__()
        self._mean_host, self._mean_dev = 0, None
        self.field = field
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = grid_size
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius                          # > 0: the field's background model colours what the rays do not hit (:61,233-236)
        if bg_radius > 0 and getattr(field, "bg_net", None) is None:
            raise ValueError("bg_radius > 0 needs a field with a background model (NGPField(bg_radius=...)); the reference has none for --ff")
        self.cuda_ray = cuda_ray
        aabb = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, grid_size ** 3]))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * grid_size ** 3 // 8, dtype=torch.uint8))
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.grid_seed = 0                             # pcg32 seed of the density-grid refresh (same on every replica)
            self._grid_ws = None
            self.mean_density = 0
            self.iter_density = 0
            self.mean_count = 0
            self.local_step = 0

    # the field's interface, as NeRFRenderer subclasses expose it
    def forward(self, x, d):
        return self.field(x, d)

    def density(self, x):
        return self.field.density(x)

    def color(self, x, d, mask=None, **kwargs):
        return self.field.color(x, d, mask=mask, **kwargs)

    def background(self, x, d):
        return self.field.background(x, d)

    def _bg_color(self, rays_o, rays_d, bg_color):
        """nerf/renderer.py:233-238 / :273-278: the background model's colour per ray when there is one, else bg_color, else white"""
        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            return self.background(sph, rays_d)
        return 1 if bg_color is None else bg_color

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.step_counter.zero_()
        self.mean_density = self.iter_density = self.mean_count = self.local_step = 0

    def _aabb(self):
        return self.aabb_train if self.training else self.aabb_infer

    # ------------------------------------------------------------------------------------------------------------
    # occupancy-grid path, op by op
    # ------------------------------------------------------------------------------------------------------------
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 trace=None, fused_field=None, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        with torch.no_grad():
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self._aabb(), self.min_near)
        bg_color = self._bg_color(rays_o, rays_d, bg_color)
        results = {}

        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb,
                                                                    128, force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            if self.density_scale != 1:                                # (x * 1 is x: two launches per step that change nothing)
                sigmas = self.density_scale * sigmas
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays)
            image, depth = mix_background(weights_sum, depth, image, nears, fars, bg_color)
            results["weights_sum"] = weights_sum
        else:
            weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
            depth = torch.zeros(N, dtype=torch.float32, device=device)
            image = torch.zeros(N, 3, dtype=torch.float32, device=device)
            n_alive = N
            rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            step = 0
            while step < max_steps:
                n_alive = rays_alive.shape[0]
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                            self.density_bitfield, self.cascade, self.grid_size, nears, fars,
                                                            128, perturb, dt_gamma, max_steps)
                if fused_field:
                    # the reference's loop and schedule, but encoder + both MLPs + activations in ONE launch per iteration
                    # (ngp_field_forward: sigma already times the renderer's density_scale) instead of ~25 small ones
                    sigmas, rgbs = self.field.forward_fused(xyzs, dirs, density_scale=self.density_scale)
                elif fused_field is None:
                    # the reference's two lines (nerf/renderer.py:360-361); NGPFieldFF.forward itself takes the one-launch route when it applies
                    sigmas, rgbs = self(xyzs, dirs)
                    sigmas = self.density_scale * sigmas
                else:
                    # fused_field=False: every op of the field separately (the parity anchor of the drop-in ops)
                    keep = getattr(self.field, "fused_inference", None)
                    if keep is not None:
                        self.field.fused_inference = False
                    try:
                        sigmas, rgbs = self(xyzs, dirs)
                    finally:
                        if keep is not None:
                            self.field.fused_inference = keep
                    sigmas = self.density_scale * sigmas
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
                if trace is not None:
                    trace.append((n_alive, n_step, int((deltas[:, 0] > 0).sum().item())))
                # stable compaction on the device; the new count reaches the host through two pinned words the kernel writes (no stream synchronisation:
                # the reference's boolean mask, nerf/renderer.py:365, pays one + a copy per iteration)
                packed, _, kept = raymarching.compact_alive(rays_alive, n_alive, count=True)
                rays_alive = packed[:kept]
                step += n_step
            image, depth = mix_background(weights_sum, depth, image, nears, fars, bg_color)
            results["weights_sum"] = weights_sum

        results["depth"] = depth.view(*prefix)
        results["image"] = image.view(*prefix, 3)
        return results

    # ------------------------------------------------------------------------------------------------------------
    # one launch per frame
    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def render_fused(self, rays_o, rays_d, dt_gamma=0, bg_color=None, max_steps=1024, image_width=0, **kwargs):
        """Inference only.  Returns image / depth / weights_sum like run_cuda plus `stats`, a 4-int device tensor:
        [ray-samples evaluated, rays that hit the max_steps cap, rays with >= 1 sample, 16-column MFMA tiles evaluated].
        image_width: width of the image when the rays are a full row-major image (enables 8x8 tile traversal)."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, device = rays_o.shape[0], rays_o.device
        image = torch.empty(N, 3, dtype=torch.float32, device=device)
        depth = torch.empty(N, dtype=torch.float32, device=device)
        weights_sum = torch.empty(N, dtype=torch.float32, device=device)
        stats = torch.empty(4, dtype=torch.int32, device=device)
        if bg_color is None:
            bg_color = 1
        bg = (ctypes.c_float * 3)(*([float(bg_color)] * 3 if np.isscalar(bg_color) else [float(v) for v in bg_color]))
        aabb = (ctypes.c_float * 6)(*[float(v) for v in self._aabb().tolist()])
        L = _hip.lib()
        ws = _hip.workspace(L.ngp_render_frame_workspace(N), device)
        f = self.field.fused_state(self.density_scale)
        _hip.check(L.ngp_render_frame(ctypes.byref(f), _hip.ptr(rays_o), _hip.ptr(rays_d), N, int(image_width), aabb, self.min_near,
                                      _hip.ptr(self.density_bitfield), self.cascade, self.grid_size, dt_gamma, max_steps, bg,
                                      _hip.ptr(image), _hip.ptr(depth), _hip.ptr(weights_sum), _hip.ptr(stats),
                                      _hip.ptr(ws), ws.numel(), _hip.stream()), "render_frame")
        out = {"image": image.view(*prefix, 3), "depth": depth.view(*prefix), "weights_sum": weights_sum, "stats": stats}
        if kwargs.get("return_workspace"):
            out["workspace"] = ws                          # debug builds (-DRV_COUNTERS) leave counters behind the ray queue
        return out

    @torch.no_grad()
    def render_fused_camera(self, pose, intrinsics, H, W, dt_gamma=0, bg_color=None, max_steps=1024):
        """render_fused for the H x W rays of one camera (pose [4,4] or [3,4] cam2world, intrinsics fx, fy, cx, cy) without
        materialising them: get_rays (nerf/utils.py:53-116) runs inside the frame kernel.  Same results, bit for bit, as
        `render_fused(*get_rays_native(...), image_width=W)`."""
        device = self.density_bitfield.device
        N = int(H) * int(W)
        image = torch.empty(N, 3, dtype=torch.float32, device=device)
        depth = torch.empty(N, dtype=torch.float32, device=device)
        weights_sum = torch.empty(N, dtype=torch.float32, device=device)
        stats = torch.empty(4, dtype=torch.int32, device=device)
        if bg_color is None:
            bg_color = 1
        bg = (ctypes.c_float * 3)(*([float(bg_color)] * 3 if np.isscalar(bg_color) else [float(v) for v in bg_color]))
        aabb = (ctypes.c_float * 6)(*[float(v) for v in self._aabb().tolist()])
        pose_h, intr_h = _hip.camera_args(pose, intrinsics)
        L = _hip.lib()
        ws = _hip.workspace(L.ngp_render_frame_workspace(N), device)
        f = self.field.fused_state(self.density_scale)
        _hip.check(L.ngp_render_frame_camera(ctypes.byref(f), pose_h, intr_h, int(H), int(W), aabb, self.min_near,
                                             _hip.ptr(self.density_bitfield), self.cascade, self.grid_size, dt_gamma, max_steps, bg,
                                             _hip.ptr(image), _hip.ptr(depth), _hip.ptr(weights_sum), _hip.ptr(stats),
                                             _hip.ptr(ws), ws.numel(), _hip.stream()), "render_frame_camera")
        return {"image": image.view(H, W, 3), "depth": depth.view(H, W), "weights_sum": weights_sum, "stats": stats}

    @torch.no_grad()
    def render_fused_cameras(self, poses, intrinsics, H, W, dt_gamma=0, bg_color=None, max_steps=1024):
        """P frames in ONE launch (ngp_render_frames_camera): poses [P,4,4] cam2world (tensor, array or nested lists), one set of intrinsics.
        The frame kernel's ramp and drain are paid once per launch instead of once per frame; every pixel equals `render_fused_camera`'s, bit
        for bit.  Returns image [P,H,W,3], depth [P,H,W], weights_sum [P,H*W], stats (summed over the frames)."""
        device = self.density_bitfield.device
        if isinstance(poses, torch.Tensor):
            poses = poses.detach().cpu().numpy()
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4))
        P, N = poses.shape[0], int(H) * int(W)
        image = torch.empty(P * N, 3, dtype=torch.float32, device=device)
        depth = torch.empty(P * N, dtype=torch.float32, device=device)
        weights_sum = torch.empty(P * N, dtype=torch.float32, device=device)
        stats = torch.empty(4, dtype=torch.int32, device=device)
        if bg_color is None:
            bg_color = 1
        bg = (ctypes.c_float * 3)(*([float(bg_color)] * 3 if np.isscalar(bg_color) else [float(v) for v in bg_color]))
        aabb = (ctypes.c_float * 6)(*[float(v) for v in self._aabb().tolist()])
        intr = (ctypes.c_float * 4)(*[float(v) for v in intrinsics])
        L = _hip.lib()
        ws = _hip.workspace(L.ngp_render_frames_workspace(P, N), device)
        f = self.field.fused_state(self.density_scale)
        _hip.check(L.ngp_render_frames_camera(ctypes.byref(f), poses.ctypes.data_as(ctypes.c_void_p), P, intr, int(H), int(W), aabb, self.min_near,
                                              _hip.ptr(self.density_bitfield), self.cascade, self.grid_size, dt_gamma, max_steps, bg,
                                              _hip.ptr(image), _hip.ptr(depth), _hip.ptr(weights_sum), _hip.ptr(stats),
                                              _hip.ptr(ws), ws.numel(), _hip.stream()), "render_frames_camera")
        return {"image": image.view(P, H, W, 3), "depth": depth.view(P, H, W), "weights_sum": weights_sum.view(P, N), "stats": stats}

    # ------------------------------------------------------------------------------------------------------------
    # fixed-step path (nav loop)
    # ------------------------------------------------------------------------------------------------------------
    def _ray_weights(self, z_vals, sample_dist, sigma):
        """nerf/renderer.py:206-210: alpha compositing weights of sorted samples; the last interval is `sample_dist` long"""
        deltas = torch.cat([z_vals[..., 1:] - z_vals[..., :-1], sample_dist * torch.ones_like(z_vals[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * self.density_scale * sigma)
        trans = torch.cumprod(torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1), dim=-1)[..., :-1]
        return alphas * trans, deltas

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, generator=None, **kwargs):
        """nerf/renderer.py:125-254: fixed-step sampling between the AABB hits, optional importance resampling (sample_pdf), torch
        compositing.  The nav loop calls it with num_steps=512, upsample_steps=0 (simulate.py:122-123); the default 128 + 128 is the
        reference's.  `generator` seeds perturb / the training-mode resampling (the reference uses the global RNG)."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        aabb = self._aabb()
        with torch.no_grad():
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears = nears.unsqueeze(-1)
        fars = fars.unsqueeze(-1)
        z_vals = torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand((N, num_steps))
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device, generator=generator) - 0.5) * sample_dist

        def points(z):
            p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
            return torch.min(torch.max(p, aabb[:3]), aabb[3:])

        xyzs = points(z_vals)
        dens = {k: v.view(N, num_steps, -1) for k, v in self.density(xyzs.reshape(-1, 3)).items()}

        if upsample_steps > 0:
            with torch.no_grad():
                w, deltas = self._ray_weights(z_vals, sample_dist, dens["sigma"].squeeze(-1))
                z_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z = sample_pdf(z_mid, w[:, 1:-1], upsample_steps, det=not self.training, generator=generator).detach()
                new_xyzs = points(new_z)
            new_dens = {k: v.view(N, upsample_steps, -1) for k, v in self.density(new_xyzs.reshape(-1, 3)).items()}
            z_vals, order = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
            xyzs = torch.gather(torch.cat([xyzs, new_xyzs], dim=1), 1, order.unsqueeze(-1).expand(-1, -1, 3))
            for k in dens:
                both = torch.cat([dens[k], new_dens[k]], dim=1)
                dens[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))

        weights, _ = self._ray_weights(z_vals, sample_dist, dens["sigma"].squeeze(-1))
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        flat = {k: v.reshape(-1, v.shape[-1]) for k, v in dens.items()}
        mask = weights > 1e-4
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **flat).view(N, -1, 3)

        weights_sum = weights.sum(dim=-1)
        ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z_vals, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        bg_color = self._bg_color(rays_o, rays_d, bg_color)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "weights_sum": weights_sum}

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """nerf/renderer.py:542-575: stage in ray batches only on the fixed-step path."""
        _run = self.run_cuda if self.cuda_ray else self.run
        B, N = rays_o.shape[:2]
        if staged and not self.cuda_ray:
            depth = torch.empty((B, N), device=rays_o.device)
            image = torch.empty((B, N, 3), device=rays_o.device)
            for b in range(B):
                for head in range(0, N, max_ray_batch):
                    tail = min(head + max_ray_batch, N)
                    r = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                    depth[b:b + 1, head:tail] = r["depth"]
                    image[b:b + 1, head:tail] = r["image"]
            return {"depth": depth, "image": image}
        return _run(rays_o, rays_d, **kwargs)

    # ------------------------------------------------------------------------------------------------------------
    # density-grid maintenance
    # ------------------------------------------------------------------------------------------------------------
    def load_density_grid(self, grid):
        """Install a [cascade, H^3] grid (numpy or tensor) and rebuild the bitfield with update_extra_state's rule."""
        g = torch.as_tensor(grid, dtype=torch.float32, device=self.density_grid.device)
        self.density_grid.copy_(g)
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, thresh, self.density_bitfield)
        return thresh

    # mean_density lives on the device after a native grid refresh; the host value is fetched only when somebody asks for it
    @property
    def mean_density(self):
        if self._mean_dev is not None:
            self._mean_host = float(self._mean_dev.item())
            self._mean_dev = None
        return self._mean_host

    @mean_density.setter
    def mean_density(self, value):
        self._mean_host = value
        self._mean_dev = None

    def _grid_workspace(self):
        need = _hip.lib().ngp_density_grid_workspace(self.cascade, self.grid_size)
        dev = self.density_bitfield.device
        if self._grid_ws is None or self._grid_ws.numel() < need or self._grid_ws.device != dev:
            self._grid_ws = _hip.workspace(need, dev)
        return self._grid_ws

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, seed=None):
        """nerf/renderer.py:446-537 on the native ops of csrc/density_grid.hip: sample points (full sweep for the first 16 calls,
        then H^3/4 random + H^3/4 occupied cells per cascade) -> density query in chunks of S^3 points -> scatter / EMA / mean /
        packbits on the device, no host synchronisation for the grid (mean_density stays on the device until it is read).
        Random numbers come from pcg32(seed = self.grid_seed, stream = iter_density): identical on every replica."""
        if not self.cuda_ray:
            return
        L = _hip.lib()
        dev = self.density_bitfield.device
        cas, H = self.cascade, self.grid_size
        partial = self.iter_density >= 16
        seed = self.grid_seed if seed is None else seed
        n = int(L.ngp_density_grid_points(cas, H, int(partial)))
        ws = self._grid_workspace()
        xyzs = torch.empty(n, 3, dtype=torch.float32, device=dev)
        cells = torch.empty(n, dtype=torch.int32, device=dev) if partial else None
        grid = self.density_grid.view(-1)
        with torch.cuda.device(dev):
            _hip.check(L.ngp_density_grid_sample(_hip.ptr(grid), cas, H, float(self.bound), int(partial), int(seed), int(self.iter_density),
                                                 _hip.ptr(xyzs), _hip.ptr(cells), _hip.ptr(ws), ws.numel(), _hip.stream()), "density_grid_sample")
            sigmas = torch.empty(n, dtype=torch.float32, device=dev)
            chunk = max(int(S), 1) ** 3
            fused = getattr(self.field, "density_sigma", None)      # NGPFieldFF: two native launches where they apply (ngp/field.py), else its own op chain
            for head in range(0, n, chunk):
                tail = min(head + chunk, n)
                with _grid.level_major_forward():          # random points: one level's table at a time stays in L2 (gridencoder/grid.py)
                    sigmas[head:tail] = fused(xyzs[head:tail]) if fused is not None else self.density(xyzs[head:tail])["sigma"].reshape(-1).detach().float()
            mean = torch.empty(1, dtype=torch.float32, device=dev)
            _hip.check(L.ngp_density_grid_update(_hip.ptr(sigmas), _hip.ptr(cells), n, float(self.density_scale), float(decay),
                                                 float(self.density_thresh), cas, H, _hip.ptr(grid), _hip.ptr(self.density_bitfield),
                                                 _hip.ptr(mean), _hip.ptr(ws), ws.numel(), _hip.stream()), "density_grid_update")
        self._mean_dev = mean
        self.iter_density += 1

        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """nerf/renderer.py:381-442 as one launch (ngp_mark_untrained_grid): cells no training camera sees get density -1."""
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        dev = self.density_bitfield.device
        poses = poses.to(device=dev, dtype=torch.float32).reshape(-1, 4, 4).contiguous()
        fx, fy, cx, cy = (float(v) for v in intrinsic)
        with torch.cuda.device(dev):
            _hip.check(_hip.lib().ngp_mark_untrained_grid(_hip.ptr(poses), poses.shape[0], fx, fy, cx, cy, self.cascade, self.grid_size,
                                                          float(self.bound), _hip.ptr(self.density_grid.view(-1)), _hip.stream()),
                       "mark_untrained_grid")
