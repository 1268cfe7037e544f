"""The callers on the navigation side of the hot path (8f rows 3 and 4).

  get_rays          nerf/utils.py:53-116 in torch on the device: full image, uniform random pixels, error-map sampling
  NavQueries        the three lambdas simulate.py hands to the planner and the pose filter (simulate.py:340-347):
                      density_fn(x)            = density(x.reshape(-1,3) @ rot)['sigma'].reshape(x.shape[:-1])
                      render_fn(rays_o, rays_d) = render(..., staged=True, bg_color=1., perturb=False)
                      get_rays_fn(pose)
                    with ONE difference that changes no result: the model is frozen (requires_grad_(False)).  The nav loop
                    differentiates w.r.t. body points and camera poses only (nav/quad_plot.py:237, nav/estimator_helpers.py:316);
                    with trainable parameters autograd also builds the 50 MB table gradient (128 atomics per point) and
                    the weight-gradient GEMMs on every call, which the reference pays and throws away.  Frozen, the encoder's
                    backward skips the scatter (gridencoder/grid.py) and autograd skips the dW GEMMs.
Everything runs in fp32 without autocast, as simulate.py does.
"""
import torch

ROT = ((0., 0., 1.), (1., 0., 0.), (0., 1., 0.))           # simulate.py:340: NeRF camera looks along +z, Blender along -z


def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, generator=None):
    """nerf/utils.py:53-116.  poses [B,4,4] cam2world, intrinsics (fx, fy, cx, cy) -> dict(rays_o, rays_d [B,N,3], inds ...).
    `generator` seeds the random branches (the reference uses the global torch RNG)."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    jj, ii = torch.meshgrid(torch.linspace(0, H - 1, H, device=device), torch.linspace(0, W - 1, W, device=device), indexing="ij")
    i = ii.reshape(1, H * W).expand(B, H * W) + 0.5          # custom_meshgrid(...).t(): row-major over the image
    j = jj.reshape(1, H * W).expand(B, H * W) + 0.5
    results = {}
    if N > 0:
        N = min(N, H * W)
        if error_map is None:
            inds = torch.randint(0, H * W, size=[N], device=device, generator=generator).expand(B, N)       # may duplicate
        else:
            inds_coarse = torch.multinomial(error_map.to(device), N, replacement=False, generator=generator)   # [B,N] in [0, 128*128)
            inds_x, inds_y = inds_coarse // 128, inds_coarse % 128
            sx, sy = H / 128, W / 128
            inds_x = (inds_x * sx + torch.rand(B, N, device=device, generator=generator) * sx).long().clamp(max=H - 1)
            inds_y = (inds_y * sy + torch.rand(B, N, device=device, generator=generator) * sy).long().clamp(max=W - 1)
            inds = inds_x * W + inds_y
            results["inds_coarse"] = inds_coarse
        i = torch.gather(i, -1, inds)
        j = torch.gather(j, -1, inds)
        results["inds"] = inds
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = torch.stack((xs, ys, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    results["rays_o"] = rays_o
    results["rays_d"] = rays_d
    return results


def get_rays_native(pose, intrinsics, H, W, inds=None, device=None):
    """get_rays for ONE camera as a native op (ngp_get_rays, csrc/ngp_camera.h): pose [4,4] cam2world, full image when
    `inds` is None, else the pixels `inds` [N] (int64, row-major).  Returns rays_o, rays_d [N,3] float32 on `device`.
    Equal to `get_rays` to ~1 ulp (torch may associate the norm and the 3x3 product differently)."""
    import ngp_hip as _hip                                   # the C-ABI library: fails loudly when it is not built
    if device is None:
        device = inds.device if inds is not None else (pose.device if isinstance(pose, torch.Tensor) and pose.is_cuda else torch.device("cuda"))
    N = int(H) * int(W) if inds is None else int(inds.numel())
    rays_o = torch.empty(N, 3, dtype=torch.float32, device=device)
    rays_d = torch.empty(N, 3, dtype=torch.float32, device=device)
    if inds is not None:
        inds = inds.to(device=device, dtype=torch.int64).contiguous()
    pose_h, intr_h = _hip.camera_args(pose, intrinsics)
    with torch.cuda.device(device):
        _hip.check(_hip.lib().ngp_get_rays(pose_h, intr_h, int(H), int(W), _hip.ptr(inds) if inds is not None else None, N,
                                           _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.stream()), "get_rays")
    return rays_o, rays_d


class NavQueries:
    """density_fn / render_fn / get_rays_fn for Planner and Estimator, bound to a frozen renderer."""

    def __init__(self, renderer, intrinsics, H, W, num_steps=512, upsample_steps=0, max_ray_batch=4096, freeze=True):
        self.renderer = renderer.eval()
        if freeze:
            for p in self.renderer.parameters():
                p.requires_grad_(False)
        self.intrinsics, self.H, self.W = intrinsics, H, W
        self.render_kwargs = dict(num_steps=num_steps, upsample_steps=upsample_steps, max_ray_batch=max_ray_batch)
        self._rot = None

    def _rot_on(self, device):
        if self._rot is None or self._rot.device != device:
            self._rot = torch.tensor(ROT, device=device, dtype=torch.float32)
        return self._rot

    def density_fn(self, x):
        return self.renderer.density(x.reshape(-1, 3) @ self._rot_on(x.device))["sigma"].reshape(x.shape[:-1])

    def render_fn(self, rays_o, rays_d):
        return self.renderer.render(rays_o, rays_d, staged=True, bg_color=1.0, perturb=False, **self.render_kwargs)

    def get_rays_fn(self, pose):
        return get_rays(pose, self.intrinsics, self.H, self.W)


class _GraphedPointwise(torch.autograd.Function):
    """sigma = f(x) for a point-wise f whose value AND per-point gradient were produced by one graph replay: the backward is
    grad_x[i] = grad_sigma[i] * dsigma_i/dx_i (a point's density depends on that point only)."""

    @staticmethod
    def forward(ctx, x, owner):
        sigma, jac = owner._replay(x)
        ctx.save_for_backward(jac)
        return sigma

    @staticmethod
    @torch.autograd.function.once_differentiable          # like the reference's encoder backward: no second derivative
    def backward(ctx, grad_sigma):
        (jac,) = ctx.saved_tensors
        return grad_sigma.reshape(-1, 1) * jac, None


class GraphedDensity:
    """The planner's query (`density_fn` on a fixed number of body points + its gradient, nav/quad_plot.py:224-250) is
    launch-bound: ~15 kernels of a few microseconds each.  This captures value and per-point gradient into ONE hipGraph
    (torch.cuda.CUDAGraph) and replays it per call; the returned sigma is differentiable w.r.t. the points through the
    captured per-point gradient.  Same kernels and the same sigma bits as `NavQueries.density_fn`; the gradient is bit-identical
    for a plain sum and equal to rounding when the caller weights the densities; the model must stay frozen
    and unchanged while the graph is in use (re-create the object after loading new weights).

        dens = GraphedDensity(queries, n_points=10000)
        sigma = dens(x)                 # x [..., 3] with prod(shape[:-1]) == n_points; sigma.sum().backward() works
    """

    def __init__(self, queries, n_points, device=None):
        self.q = queries
        device = device or next(queries.renderer.parameters()).device
        if any(p.requires_grad for p in queries.renderer.parameters()):
            raise ValueError("GraphedDensity needs a frozen model (NavQueries(freeze=True))")
        self.n = int(n_points)
        self._x = torch.zeros(self.n, 3, device=device, requires_grad=True)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):                       # warm-up outside the capture (lazy initialisation, allocator)
            for _ in range(2):
                s = self.q.density_fn(self._x)
                torch.autograd.grad(s.sum(), self._x)
        torch.cuda.current_stream(device).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._sigma = self.q.density_fn(self._x)
            (self._jac,) = torch.autograd.grad(self._sigma.sum(), self._x)

    def _replay(self, x):
        with torch.no_grad():
            self._x.copy_(x.reshape(self.n, 3))
        self._graph.replay()
        return self._sigma.clone(), self._jac.clone()        # the static buffers are overwritten by the next replay

    def __call__(self, x):
        if x.numel() != 3 * self.n:
            raise ValueError(f"GraphedDensity was captured for {self.n} points, got {x.numel() // 3}")
        flat = x.reshape(self.n, 3)
        return _GraphedPointwise.apply(flat, self).reshape(x.shape[:-1])
