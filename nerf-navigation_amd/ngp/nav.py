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
    def backward(ctx, grad_sigma):
        # The reference's rule: the encoder's backward is a native op writing into a fresh tensor, so under create_graph=True the
        # gradient w.r.t. the points is a CONSTANT (no graph to the points, none to grad_sigma) and no error is raised -- not @once_differentiable,
        # which would make torch.autograd.functional.hessian (nav/estimator_helpers.py:384) fail where the reference returns a matrix.
        (jac,) = ctx.saved_tensors
        return (grad_sigma.detach().reshape(-1, 1) * jac).detach(), None


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


# ------------------------------------------------------------------------------------------------------------------------
# the same queries on the fused float32 kernels of csrc/nav_field.hip: one launch forward, one launch backward
# ------------------------------------------------------------------------------------------------------------------------
class _NativeField:
    """ngp_nav_field_t + the prepared (transposed) weights for a default-architecture field (ngp.field.NGPField: nn.Linear layers,
    32-64-16 | 31-64-64-3, 16 x 2 hash grid).  Rebuilt when a parameter's storage or version changes."""

    def __init__(self, renderer):
        import ctypes
        import numpy as np
        import ngp_hip as _hip
        self._hip, self._ct, self._np = _hip, ctypes, np
        self.renderer = renderer
        self.field = renderer.field
        f = self.field
        ok = (hasattr(f.sigma_net, "__len__") and len(f.sigma_net) == 2 and len(f.color_net) == 3
              and tuple(f.sigma_net[0].weight.shape) == (64, 32) and tuple(f.sigma_net[1].weight.shape) == (16, 64)
              and tuple(f.color_net[0].weight.shape) == (64, 31) and tuple(f.color_net[1].weight.shape) == (64, 64)
              and tuple(f.color_net[2].weight.shape) == (3, 64) and f.encoder.num_levels == 16 and f.encoder.level_dim == 2
              and f.encoder.input_dim == 3 and f.encoder.gridtype == "hash" and not f.encoder.align_corners
              and f.encoder.embeddings.dtype == torch.float32)
        if not ok:
            raise RuntimeError("the fused nav queries support the default field only (hash grid 16 x 2 float32, Linear 32-64-16 | 31-64-64-3)")
        self._key = None
        self._offsets = (ctypes.c_int32 * 17)(*[int(v) for v in f.encoder.offsets.cpu().tolist()])

    def _params(self):
        f = self.field
        return [f.encoder.embeddings, f.sigma_net[0].weight, f.sigma_net[1].weight, f.color_net[0].weight, f.color_net[1].weight, f.color_net[2].weight]

    def struct(self):
        """(ngp_nav_field_t, prepared-weights tensor); the transposes are redone only when a parameter changed"""
        _hip, ct = self._hip, self._ct
        ps = self._params()
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps) + (float(self.field.bound), float(self.renderer.density_scale), getattr(self.field, "_param_epoch", 0))
        if key != self._key:
            dev = ps[0].device
            for p in ps:
                if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                    raise RuntimeError("the fused nav queries need contiguous float32 parameters on the GPU")
            L = _hip.lib()
            self._prep = _hip.workspace(L.ngp_nav_field_workspace(), dev)
            self._struct = _hip.ngp_nav_field_t(ps[0].data_ptr(), ct.cast(self._offsets, ct.c_void_p), ps[1].data_ptr(), ps[2].data_ptr(),
                                                ps[3].data_ptr(), ps[4].data_ptr(), ps[5].data_ptr(), 16, self.field.encoder.base_resolution,
                                                float(self._np.log2(self.field.encoder.per_level_scale)), float(self.field.bound),
                                                float(self.renderer.density_scale))
            with torch.cuda.device(dev):
                _hip.check(L.ngp_nav_field_prepare(ct.byref(self._struct), _hip.ptr(self._prep), self._prep.numel(), _hip.stream()), "nav_field_prepare")
            self._key = key
        return self._struct, self._prep


class _nav_density(torch.autograd.Function):
    """sigma = trunc_exp(h0(x)) for [M,3] points in one launch; backward to the points in one launch (ngp_nav_density_*).  Like the
    reference's encoder backward this is a first-order op whose gradient carries NO graph under create_graph=True: every
    path from the points to sigma crosses the grid encoder (gridencoder/grid.py:61-87), so the reference's gradient is graph-less too."""

    @staticmethod
    def forward(ctx, x, owner):
        import ctypes
        import ngp_hip as _hip
        x = x.contiguous().float()
        M = x.shape[0]
        sigma = torch.empty(M, dtype=torch.float32, device=x.device)
        st, prep = owner.struct()
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib().ngp_nav_density_forward(ctypes.byref(st), _hip.ptr(prep), _hip.ptr(x), M, _hip.ptr(sigma), None, _hip.stream()),
                       "nav_density_forward")
        ctx.save_for_backward(x)
        ctx.owner = owner
        return sigma

    @staticmethod
    def backward(ctx, grad_sigma):
        import ctypes
        import ngp_hip as _hip
        (x,) = ctx.saved_tensors
        grad_sigma = grad_sigma.detach()
        M = x.shape[0]
        gx = torch.empty_like(x)
        st, prep = ctx.owner.struct()
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib().ngp_nav_density_backward(ctypes.byref(st), _hip.ptr(prep), _hip.ptr(x), M, _hip.ptr(grad_sigma.contiguous().float()),
                                                           None, _hip.ptr(gx), _hip.stream()), "nav_density_backward")
        return gx, None


class _nav_density_vj(torch.autograd.Function):
    """sigma AND d sigma / d x for [M,3] points in ONE launch (ngp_nav_density_value_jac: a point's 16 levels over the four waves of a workgroup; the planner's
    batch sizes), the axis change `x @ rot` of simulate.py:340 folded in.  backward = grad_sigma * jac, a constant of any second pass like the reference's
    encoder backward."""

    @staticmethod
    def forward(ctx, x, owner, rot9):
        import ctypes
        import ngp_hip as _hip
        x = x.contiguous().float()
        M = x.shape[0]
        sigma = torch.empty(M, dtype=torch.float32, device=x.device)
        jac = torch.empty(M, 3, dtype=torch.float32, device=x.device)
        st, prep = owner.struct()
        rot = (ctypes.c_float * 9)(*rot9) if rot9 is not None else None
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib().ngp_nav_density_value_jac(ctypes.byref(st), _hip.ptr(prep), _hip.ptr(x), M, rot, _hip.ptr(sigma), _hip.ptr(jac), _hip.stream()),
                       "nav_density_value_jac")
        ctx.save_for_backward(jac)
        return sigma

    @staticmethod
    def backward(ctx, grad_sigma):
        (jac,) = ctx.saved_tensors
        return (grad_sigma.detach().reshape(-1, 1) * jac).detach(), None, None


class _nav_run(torch.autograd.Function):
    """NeRFRenderer.run(num_steps, upsample_steps = 0, perturb = False) for [N,3] rays: one launch forward, one backward to the rays.
    Second derivatives (the pose filter's Hessian, nav/estimator_helpers.py:384): in the reference every path from the rays to the image crosses an
    encoder (xyzs -> grid encoder, dirs -> SH encoder; near / far are computed under no_grad, nerf/renderer.py:140-141), whose backward returns
    graph-less tensors; so d L / d rays is a constant of the second pass there, and the tensors this backward fills through the C ABI are exactly that.
    tests/test_gpu_nav_golden.py checks the 12 x 12 Hessian against the executed reference's."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, owner, num_steps, bg):
        import ctypes
        import raymarching
        import ngp_hip as _hip
        rays_o, rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()
        N = rays_o.shape[0]
        ren = owner.renderer
        aabb_t = ren._aabb()
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb_t, ren.min_near)
        aabb = (ctypes.c_float * 6)(*[float(v) for v in aabb_t.tolist()])
        bgc = (ctypes.c_float * 3)(*bg)
        image = torch.empty(N, 3, dtype=torch.float32, device=rays_o.device)
        depth = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        ws = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        st, prep = owner.struct()
        L = _hip.lib()
        # the per-sample record the backward consumes (108 B per sample); not kept when nobody will ask for a gradient
        need = any(ctx.needs_input_grad[:2])
        saved = _hip.workspace(L.ngp_nav_run_saved_bytes(N, int(num_steps)), rays_o.device) if need else None
        with torch.cuda.device(rays_o.device), _hip.timed("nav_run_forward"):
            _hip.check(L.ngp_nav_run_forward(ctypes.byref(st), _hip.ptr(prep), _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars),
                                             N, int(num_steps), aabb, bgc, _hip.ptr(image), _hip.ptr(depth), _hip.ptr(ws),
                                             _hip.ptr(saved), saved.numel() if need else 0, _hip.stream()), "nav_run_forward")
        ctx.save_for_backward(rays_o, rays_d, nears, fars, saved if need else torch.empty(0, device=rays_o.device))
        ctx.owner, ctx.num_steps, ctx.bg, ctx.aabb = owner, int(num_steps), bgc, aabb
        return image, depth, ws

    @staticmethod
    def backward(ctx, g_image, g_depth, g_ws):
        import ctypes
        import ngp_hip as _hip
        rays_o, rays_d, nears, fars, saved = ctx.saved_tensors
        g_image, g_depth, g_ws = g_image.detach(), g_depth.detach(), g_ws.detach()
        N = rays_o.shape[0]
        go, gd = torch.empty_like(rays_o), torch.empty_like(rays_d)
        st, prep = ctx.owner.struct()
        with torch.cuda.device(rays_o.device), _hip.timed("nav_run_backward"):
            _hip.check(_hip.lib().ngp_nav_run_backward(ctypes.byref(st), _hip.ptr(prep), _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars),
                                                       N, ctx.num_steps, ctx.aabb, ctx.bg, _hip.ptr(g_image.contiguous().float()),
                                                       _hip.ptr(g_depth.contiguous().float()), _hip.ptr(g_ws.contiguous().float()),
                                                       _hip.ptr(saved), saved.numel(), _hip.ptr(go), _hip.ptr(gd), _hip.stream()), "nav_run_backward")
        return go, gd, None, None, None


class NativeNavQueries(NavQueries):
    """NavQueries on the fused float32 kernels (csrc/nav_field.hip): `density_fn` = one launch (+ one for its gradient), `render_fn` = one
    launch per max_ray_batch rays (+ one for the gradient to the rays) instead of ~130 launches per filter iteration.  Same interface and
    the same results to float32 rounding (tests/test_gpu_nav_native.py compares both with the CPU oracle).  Needs the default field
    (ngp.field.NGPField) in float32 and upsample_steps == 0 (what simulate.py uses); anything else: use NavQueries."""

    def __init__(self, renderer, intrinsics, H, W, num_steps=512, upsample_steps=0, max_ray_batch=4096, freeze=True):
        super().__init__(renderer, intrinsics, H, W, num_steps=num_steps, upsample_steps=upsample_steps, max_ray_batch=max_ray_batch, freeze=freeze)
        if upsample_steps != 0:
            raise ValueError("NativeNavQueries implements upsample_steps == 0 (the navigation configuration); use NavQueries otherwise")
        self.native = _NativeField(self.renderer)
        self.num_steps, self.max_ray_batch = int(num_steps), int(max_ray_batch)

    # The one-lane-per-point kernels walk a point's 16 levels serially: a small batch (the planner's 10,000 body points = 40 workgroups) is bound by that
    # lane's 16 dependent gather round trips.  Below this many points the query is ONE launch of the level-parallel value-and-Jacobian kernel
    # (ngp_nav_density_value_jac, the axis change folded in); the A* occupancy query (10^6 points, nav/quad_plot.py:65-79) and the filter are far above it.
    NATIVE_MIN_POINTS = 32768
    _ROT9 = tuple(v for row in ROT for v in row)

    def density_fn(self, x):
        if x.numel() // 3 < self.NATIVE_MIN_POINTS:
            return _nav_density_vj.apply(x.reshape(-1, 3), self.native, self._ROT9).reshape(x.shape[:-1])
        pts = x.reshape(-1, 3) @ self._rot_on(x.device)
        return _nav_density.apply(pts, self.native).reshape(x.shape[:-1])

    def density_fn_chain(self, x):
        """the level-parallel op chain (NavQueries.density_fn) whatever the batch size (tests, timing)"""
        return super().density_fn(x)

    def density_fn_native(self, x):
        """the fused kernels whatever the batch size (tests, timing)"""
        pts = x.reshape(-1, 3) @ self._rot_on(x.device)
        return _nav_density.apply(pts, self.native).reshape(x.shape[:-1])

    def render_fn(self, rays_o, rays_d):
        """rays [B, N, 3] -> {'image' [B,N,3], 'depth' [B,N]} like NeRFRenderer.render(staged=True, bg_color=1, perturb=False)"""
        B, N = rays_o.shape[:2]
        images, depths = [], []
        for b in range(B):
            for head in range(0, N, self.max_ray_batch):
                tail = min(head + self.max_ray_batch, N)
                img, dep, _ = _nav_run.apply(rays_o[b, head:tail], rays_d[b, head:tail], self.native, self.num_steps, (1.0, 1.0, 1.0))
                images.append(img)
                depths.append(dep)
        return {"image": torch.cat(images).view(B, N, 3), "depth": torch.cat(depths).view(B, N)}
