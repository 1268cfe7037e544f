"""Renderer harness over the drop-in ops: the counterpart of nerf/renderer.py's NeRFRenderer.

Three render paths:
  run_cuda(...)      the reference's occupancy-grid path, op by op (nerf/renderer.py:257-379): the parity anchor and
                     what a torch-ngp caller gets from the drop-in packages;
  render_fused(...)  one launch per frame (csrc/render_fused.hip) -- the MI355X-native fast path for inference;
  run(...)           the fixed-step path used by the nav loop (nerf/renderer.py:125-254, simulate.py:163-166).
plus update_extra_state / mark_untrained_grid (density-grid maintenance, nerf/renderer.py:381-537).
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

import ngp_hip as _hip
import raymarching


class NGPRenderer(nn.Module):
    def __init__(self, field, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=0.01, grid_size=128):
        super().__init__()
        self.field = field
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = grid_size
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = -1
        self.cuda_ray = cuda_ray
        aabb = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, grid_size ** 3]))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * grid_size ** 3 // 8, dtype=torch.uint8))
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
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
                 trace=None, fused_field=False, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        with torch.no_grad():
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self._aabb(), self.min_near)
        if bg_color is None:
            bg_color = 1
        results = {}

        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb,
                                                                    128, force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
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
                    # (ngp_field_forward: sigma already times the field's density_scale) instead of ~25 small ones
                    sigmas, rgbs = self.field.forward_fused(xyzs, dirs)
                else:
                    sigmas, rgbs = self(xyzs, dirs)
                    sigmas = self.density_scale * sigmas
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
                if trace is not None:
                    trace.append((n_alive, n_step, int((deltas[:, 0] > 0).sum().item())))
                # stable compaction on the device; one 4-byte readback for the new count (the reference's boolean mask
                # costs the same sync, nerf/renderer.py:365)
                packed, cnt = raymarching.compact_alive(rays_alive, n_alive)
                rays_alive = packed[: int(cnt.item())]
                step += n_step
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
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
        f = self.field.fused_state()
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
        f = self.field.fused_state()
        _hip.check(L.ngp_render_frame_camera(ctypes.byref(f), pose_h, intr_h, int(H), int(W), aabb, self.min_near,
                                             _hip.ptr(self.density_bitfield), self.cascade, self.grid_size, dt_gamma, max_steps, bg,
                                             _hip.ptr(image), _hip.ptr(depth), _hip.ptr(weights_sum), _hip.ptr(stats),
                                             _hip.ptr(ws), ws.numel(), _hip.stream()), "render_frame_camera")
        return {"image": image.view(H, W, 3), "depth": depth.view(H, W), "weights_sum": weights_sum, "stats": stats}

    # ------------------------------------------------------------------------------------------------------------
    # fixed-step path (nav loop)
    # ------------------------------------------------------------------------------------------------------------
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=0, bg_color=None, perturb=False, **kwargs):
        """nerf/renderer.py:125-254 with upsample_steps == 0 (the nav configuration, simulate.py:122-123)."""
        if upsample_steps > 0:
            raise NotImplementedError("upsample_steps > 0 (sample_pdf) is not on the navigation path")
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
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist
        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])

        density_outputs = self.density(xyzs.reshape(-1, 3))
        for k, v in density_outputs.items():
            density_outputs[k] = v.view(N, num_steps, -1)

        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * self.density_scale * density_outputs["sigma"].squeeze(-1))
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]

        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        for k, v in density_outputs.items():
            density_outputs[k] = v.view(-1, v.shape[-1])
        mask = weights > 1e-4
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **density_outputs).view(N, -1, 3)

        weights_sum = weights.sum(dim=-1)
        ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z_vals, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        if bg_color is None:
            bg_color = 1
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

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, generator=None):
        """nerf/renderer.py:446-537.  `generator` seeds the jitter / sampling so replicas stay identical."""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        H = self.grid_size
        tmp_grid = -torch.ones_like(self.density_grid)

        def rand_like(t):
            return torch.rand(t.shape, device=dev, dtype=t.dtype, generator=generator)

        def query(coords, indices, cas):
            xyzs = 2 * coords.float() / (H - 1) - 1
            bound = min(2 ** cas, self.bound)
            half_grid_size = bound / H
            cas_xyzs = xyzs * (bound - half_grid_size)
            cas_xyzs += (rand_like(cas_xyzs) * 2 - 1) * half_grid_size
            sigmas = self.density(cas_xyzs)["sigma"].reshape(-1).detach().float()
            sigmas *= self.density_scale
            tmp_grid[cas, indices] = sigmas

        if self.iter_density < 16:
            ar = torch.arange(H, dtype=torch.int32, device=dev).split(S)
            for xs in ar:
                for ys in ar:
                    for zs in ar:
                        xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                        coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                        indices = raymarching.morton3D(coords).long()
                        for cas in range(self.cascade):
                            query(coords, indices, cas)
        else:
            N = H ** 3 // 4
            for cas in range(self.cascade):
                coords = torch.randint(0, H, (N, 3), device=dev, generator=generator)
                indices = raymarching.morton3D(coords).long()
                occ_indices = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                rand_mask = torch.randint(0, occ_indices.shape[0], [N], dtype=torch.long, device=dev, generator=generator)
                occ_indices = occ_indices[rand_mask]
                occ_coords = raymarching.morton3D_invert(occ_indices)
                query(torch.cat([coords, occ_coords], dim=0), torch.cat([indices, occ_indices], dim=0), cas)

        valid_mask = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid[valid_mask] = torch.maximum(self.density_grid[valid_mask] * decay, tmp_grid[valid_mask])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1
        density_thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)

        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """nerf/renderer.py:381-442: cells no training camera sees get density -1."""
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        dev = self.density_bitfield.device
        H = self.grid_size
        B = poses.shape[0]
        fx, fy, cx, cy = intrinsic
        count = torch.zeros_like(self.density_grid)
        poses = poses.to(dev)
        ar = torch.arange(H, dtype=torch.int32, device=dev).split(S)
        for xs in ar:
            for ys in ar:
                for zs in ar:
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    indices = raymarching.morton3D(coords).long()
                    world_xyzs = (2 * coords.float() / (H - 1) - 1).unsqueeze(0)
                    for cas in range(self.cascade):
                        bound = min(2 ** cas, self.bound)
                        half_grid_size = bound / H
                        cas_world_xyzs = world_xyzs * (bound - half_grid_size)
                        for head in range(0, B, S):
                            tail = min(head + S, B)
                            cam_xyzs = cas_world_xyzs - poses[head:tail, :3, 3].unsqueeze(1)
                            cam_xyzs = cam_xyzs @ poses[head:tail, :3, :3]
                            mask_z = cam_xyzs[:, :, 2] > 0
                            mask_x = torch.abs(cam_xyzs[:, :, 0]) < cx / fx * cam_xyzs[:, :, 2] + half_grid_size * 2
                            mask_y = torch.abs(cam_xyzs[:, :, 1]) < cy / fy * cam_xyzs[:, :, 2] + half_grid_size * 2
                            count[cas, indices] += (mask_z & mask_x & mask_y).sum(0).reshape(-1)
        self.density_grid[count == 0] = -1
