"""CPU restatement of the reference's Python callers around the native ops, in torch on the CPU so that autograd supplies
the gradients the GPU tests check (images, d image / d rays, d sigma / d x, weight and table gradients).

TEST INFRASTRUCTURE ONLY -- never imported by the product package (nerf-navigation_amd/), never timed as the product.

What is restated, with the reference lines each function follows:
  grid_encode          gridencoder/src/gridencoder.cu:75-170 + gridencoder/grid.py:140-156 as index arithmetic + gathers; autograd through
                       the fractional position gives the reference's dy_dx (gridencoder.cu:173-222) and through the gather its
                       scatter-add table gradient (:227-343)
  sh_encode            shencoder/src/shencoder.cu:50-122 -- built from Legendre coefficients like oracle/sh_oracle.py (pinned against the
                       reference's 256 polynomials by tests/golden/sh_deg8.npz)
  DefaultField         nerf/network.py:95-191 : forward / density / color(mask) with bias-free Linear layers, fp32
  trunc_exp            activation.py:5-18
  sample_pdf           nerf/renderer.py:12-46
  run                  nerf/renderer.py:125-254 (fixed-step renderer of the nav loop; upsample_steps >= 0)
  composite_rays_train raymarching/raymarching.py:233-283 as an autograd.Function over oracle/ngp_oracle.c's forward/backward
  run_cuda_train       nerf/renderer.py:282-323 (training branch: march_rays_train -> field -> composite_rays_train -> bg mix)
  update_extra_state   nerf/renderer.py:446-537 with the random numbers INJECTED as arrays (the reference draws them from torch's global
                       RNG; the native ops draw them from pcg32 streams, `grid_update_randoms` below states the stream layout)
  mark_untrained_grid  nerf/renderer.py:381-442
  psnr_meter           nerf/utils.py:185-219

Pinning: the reference ships no fixtures for any of these (it has no tests) and its renderer cannot be imported here (trimesh,
raymarching CUDA extension): PARITY UNPINNED by reference outputs.  tests/test_oracle_callers.py pins them by the relations the
reference itself implies: grid_encode == the C oracle's kernel restatement; sh_encode == the golden polynomial table; run()'s
weights == composite_rays_train's (SURVEY 8c relation 1); sample_pdf == the inverse-CDF definition; update_extra_state's scatter ==
a dense recomputation.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from numpy.polynomial import legendre as _leg

from . import ngp_oracle as O

PRIMES = (1, 2654435761, 805459861)                     # gridencoder.cu:42 (first three)
MASK32 = 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------------------------------------
def _grid_index(loc, hashmap_size, resolution, gridtype, align_corners):
    """get_grid_index (gridencoder.cu:54-72) on int64 tensors holding uint32 values."""
    stride, index = 1, torch.zeros_like(loc[0])
    step = resolution if align_corners else resolution + 1
    d = 0
    while d < len(loc) and stride <= hashmap_size:
        index = (index + loc[d] * stride) & MASK32
        stride *= step
        d += 1
    if gridtype == 0 and stride > hashmap_size:
        index = torch.zeros_like(loc[0])
        for k, c in enumerate(loc):
            index = index ^ ((c * PRIMES[k]) & MASK32)              # fast_hash: uint32 wrap-around multiply, xor (:35-51)
    return index % hashmap_size


def grid_encode(x, embeddings, offsets, per_level_scale, base_resolution=16, bound=1.0, gridtype=0, align_corners=False):
    """GridEncoder.forward (grid.py:140-156): x [B,3] world coordinates in [-bound, bound] -> [B, L*C] in the dtype of `embeddings`
    (float32 or float64).  Differentiable w.r.t. x and embeddings."""
    dt = embeddings.dtype
    # grid.py:144 `(inputs + bound) / (2 * bound)` as the reference evaluates it on its GPU: torch divides a tensor by a host scalar by multiplying
    # with the scalar's reciprocal in the tensor's dtype (ATen BinaryDivTrueKernel.cu).  Exact for a power-of-two 2 * bound.
    x01 = (x.to(dt) + bound) * torch.tensor(1.0, dtype=dt).div(torch.tensor(2 * bound, dtype=dt))
    B, D = x01.shape
    L = len(offsets) - 1
    S = np.float32(np.log2(per_level_scale))                          # grid.py:33: narrowed to float at the boundary
    oob = ((x01 < 0) | (x01 > 1)).any(dim=-1, keepdim=True)           # gridencoder.cu:100-123
    scales, resolutions = O.grid_level_table(L, S, base_resolution)   # :125-127 in float32 with libm's exp2f, as the C restatement
    outs = []
    for level in range(L):
        size = int(offsets[level + 1] - offsets[level])
        scale, resolution = scales[level], int(resolutions[level])
        pos = x01 * float(scale) + (0.0 if align_corners else 0.5)
        cell = torch.floor(pos.detach())
        frac = pos - cell
        cell = cell.long().clamp(min=0)                               # oob rows are zeroed below; keep their indices in range
        acc = torch.zeros(B, embeddings.shape[1], dtype=dt)
        for corner in range(1 << D):
            w = torch.ones(B, dtype=dt)
            loc = []
            for d in range(D):
                if corner & (1 << d):
                    w = w * frac[:, d]
                    loc.append(cell[:, d] + 1)
                else:
                    w = w * (1 - frac[:, d])
                    loc.append(cell[:, d])
            rows = _grid_index(loc, size, resolution, gridtype, align_corners) + int(offsets[level])
            acc = acc + w.unsqueeze(-1) * embeddings[rows]
        outs.append(torch.where(oob, torch.zeros_like(acc), acc))
    return torch.cat(outs, dim=-1)


def _sh_tables(degree):
    """[(l, m, norm, power-series coefficients of d^m P_l / dz^m)] for l < degree, 0 <= m <= l."""
    rows = []
    for l in range(degree):
        for m in range(l + 1):
            n = math.sqrt((1.0 if m == 0 else 2.0) * (2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            if m & 1:
                n = -n
            p = _leg.Legendre.basis(l)
            if m:
                p = p.deriv(m)
            rows.append((l, m, n, _leg.leg2poly(p.coef)))
    return rows


def sh_encode(d, degree=4):
    """SHEncoder.forward (shencoder/sphere_harmonics.py:75-87; polynomials shencoder.cu:50-122): d [B,3] -> [B, degree^2]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    re, im = [torch.ones_like(x)], [torch.zeros_like(x)]
    for _ in range(degree):
        re, im = re + [re[-1] * x - im[-1] * y], im + [re[-1] * y + im[-1] * x]
    cols = [None] * (degree * degree)
    for l, m, n, coef in _sh_tables(degree):
        q = torch.zeros_like(z)
        for c in coef[::-1]:
            q = q * z + float(c)
        cols[l * l + l + m] = n * q * re[m]
        if m:
            cols[l * l + l - m] = n * q * im[m]
    return torch.stack(cols, dim=-1)


class _GridEncodeFirstOrder(torch.autograd.Function):
    """_grid_encode (gridencoder/grid.py:19-87) with the reference's DIFFERENTIATION RULE: its backward hands `grad` to a native op that writes
    into fresh tensors (`grad_inputs = torch.zeros_like(inputs)` :76, `@once_differentiable` commented out :62), so under create_graph=True the
    gradients w.r.t. inputs and table carry NO graph -- neither back to the inputs nor to `grad`.  torch.autograd.functional.hessian
    (nav/estimator_helpers.py:384) therefore sees the encoder as a function whose gradient is a constant.  Values = `grid_encode`."""

    @staticmethod
    def forward(ctx, x, embeddings, offsets, per_level_scale, base_resolution, bound, gridtype, align_corners):
        ctx.save_for_backward(x, embeddings)
        ctx.args = (offsets, per_level_scale, base_resolution, bound, gridtype, align_corners)
        return grid_encode(x.detach(), embeddings.detach(), *ctx.args)

    @staticmethod
    def backward(ctx, grad):
        x, emb = ctx.saved_tensors
        need = [ctx.needs_input_grad[0], ctx.needs_input_grad[1]]
        with torch.enable_grad():
            xi, ei = x.detach().requires_grad_(need[0]), emb.detach().requires_grad_(need[1])
            y = grid_encode(xi, ei, *ctx.args)
            wrt = [t for t, n in zip((xi, ei), need) if n]
            got = list(torch.autograd.grad(y, wrt, grad.detach(), allow_unused=True)) if wrt else []
        out = [None, None]
        for k in (0, 1):
            if need[k]:
                g = got.pop(0)
                out[k] = (torch.zeros_like((x, emb)[k]) if g is None else g).detach()
        return (out[0], out[1]) + (None,) * 6


def grid_encode_first_order(x, embeddings, offsets, per_level_scale, base_resolution=16, bound=1.0, gridtype=0, align_corners=False):
    """`grid_encode` as the reference's autograd.Function differentiates it: first derivatives only, graph-less (see _GridEncodeFirstOrder)"""
    return _GridEncodeFirstOrder.apply(x, embeddings, offsets, per_level_scale, base_resolution, bound, gridtype, align_corners)


class _SHEncodeFirstOrder(torch.autograd.Function):
    """_sh_encoder (shencoder/sphere_harmonics.py:14-54): same rule -- `grad_inputs = torch.zeros_like(inputs)` filled by the native op (:49-51)."""

    @staticmethod
    def forward(ctx, d, degree):
        ctx.save_for_backward(d)
        ctx.degree = degree
        return sh_encode(d.detach(), degree)

    @staticmethod
    def backward(ctx, grad):
        (d,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        with torch.enable_grad():
            di = d.detach().requires_grad_(True)
            (g,) = torch.autograd.grad(sh_encode(di, ctx.degree), di, grad.detach())
        return g.detach(), None


def sh_encode_first_order(d, degree=4):
    return _SHEncodeFirstOrder.apply(d, degree)


class _TruncExp(torch.autograd.Function):
    """activation.py:5-18"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


trunc_exp = _TruncExp.apply


# ------------------------------------------------------------------------------------------------------------------------
# the default field (nn.Linear layers)
# ------------------------------------------------------------------------------------------------------------------------
class DefaultField:
    """NeRFNetwork of nerf/network.py: sigma net Linear(32,64), Linear(64,16); colour net Linear(31,64), Linear(64,64), Linear(64,3);
    no biases (:45,66).  Parameters are plain leaf tensors so tests read their .grad."""

    def __init__(self, embeddings, offsets, per_level_scale, sigma_weights, color_weights, bound, dtype=torch.float32, ff_layout=False,
                 first_order_encoders=False):
        """ff_layout: the FFMLP variant of nerf/network_ff.py:51-77 -- colour input cat(SH16, geo15, one zero column) = 32 wide (:67-68),
        colour output 16 wide of which [:3] is used (:72-74); any number of hidden layers in either net.
        first_order_encoders: differentiate the encoders like the reference's autograd.Functions (graph-less gradients, N3) -- needed whenever a
        caller asks for second derivatives (the pose filter's Hessian); first derivatives are the same either way."""
        self.ff_layout = ff_layout
        self._grid = grid_encode_first_order if first_order_encoders else grid_encode
        self._sh = sh_encode_first_order if first_order_encoders else sh_encode
        as_t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype).clone().requires_grad_(True)        # noqa: E731
        self.embeddings = as_t(embeddings)
        self.offsets = [int(v) for v in offsets]
        self.per_level_scale = float(per_level_scale)
        self.sigma_weights = [as_t(w) for w in sigma_weights]
        self.color_weights = [as_t(w) for w in color_weights]
        self.bound = float(bound)
        self.dtype = dtype

    def parameters(self):
        return [self.embeddings] + self.sigma_weights + self.color_weights

    @staticmethod
    def _mlp(weights, h):
        for i, w in enumerate(weights):
            h = F.linear(h, w)
            if i != len(weights) - 1:
                h = F.relu(h)
        return h

    def density(self, x):                                               # network.py:125-143
        h = self._mlp(self.sigma_weights, self._grid(x, self.embeddings, self.offsets, self.per_level_scale, bound=self.bound))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):          # network.py:163-191
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=self.dtype)
            if not mask.any():
                return rgbs
            d, geo_feat = d[mask], geo_feat[mask]
        cin = [self._sh(d.to(self.dtype)), geo_feat]
        if self.ff_layout:
            cin.append(torch.zeros_like(geo_feat[..., :1]))
        h = torch.sigmoid(self._mlp(self.color_weights, torch.cat(cin, dim=-1))[..., :3])
        if mask is not None:
            rgbs = rgbs.clone()
            rgbs[mask] = h
            return rgbs
        return h

    def background(self, sph, d, embeddings_bg, offsets_bg, per_level_scale_bg, bg_weights):
        """NeRFNetwork.background (nerf/network.py:145-161): 2-D hash grid (4 levels, base 16) of the sphere coordinates in [-1,1] ++ SH16(d)
        -> bias-free Linear layers -> sigmoid.  The 2-D grid goes through the C restatement (no gradient: used for images only)."""
        x01 = ((sph.numpy().astype(np.float32) + np.float32(1)) / np.float32(2)).astype(np.float32)
        feats, _ = O.grid_encode_forward(x01, np.asarray(embeddings_bg, np.float32), np.asarray(offsets_bg, np.int32), per_level_scale_bg, 16, False, 0, False)
        enc = torch.from_numpy(np.ascontiguousarray(feats.transpose(1, 0, 2).reshape(x01.shape[0], -1))).to(self.dtype)
        h = torch.cat([sh_encode(d.to(self.dtype)), enc], dim=-1)
        return torch.sigmoid(self._mlp([torch.as_tensor(np.asarray(w), dtype=self.dtype) for w in bg_weights], h))

    def forward(self, x, d):                                            # network.py:95-123
        out = self.density(x)
        return out["sigma"], self.color(x, d, geo_feat=out["geo_feat"])

    __call__ = forward


# ------------------------------------------------------------------------------------------------------------------------
# run(): the fixed-step renderer of the nav loop
# ------------------------------------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, n_samples, det=True, u=None):
    """nerf/renderer.py:12-46.  bins [B,T], weights [B,T-1] -> [B,n_samples].  `u` injects the uniforms of the det=False branch."""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, n_samples, dtype=cdf.dtype).expand(cdf.shape[0], n_samples)
    elif u is None:
        raise ValueError("det=False needs the uniforms")
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b0, b1 = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


def _weights(z_vals, sample_dist, sigma, density_scale):
    """nerf/renderer.py:206-210"""
    deltas = torch.cat([z_vals[..., 1:] - z_vals[..., :-1], sample_dist * torch.ones_like(z_vals[..., :1])], dim=-1)
    alphas = 1 - torch.exp(-deltas * density_scale * sigma)
    shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
    return alphas * torch.cumprod(shifted, dim=-1)[..., :-1], deltas


def run(field, rays_o, rays_d, bound, num_steps=128, upsample_steps=0, bg_color=1.0, min_near=0.2, density_scale=1.0, training=False,
        perturb_u=None, pdf_u=None):
    """NeRFRenderer.run (nerf/renderer.py:125-254).  rays_* torch [N,3] (may require grad).  perturb_u: the uniforms of :153 or None.
    Returns dict(image [N,3], depth [N], weights_sum [N], weights [N,T], mask [N,T])."""
    dt = rays_o.dtype
    N = rays_o.shape[0]
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=dt)
    n, f = O.near_far_from_aabb(rays_o.detach().numpy().astype(np.float32), rays_d.detach().numpy().astype(np.float32),
                                aabb.numpy().astype(np.float32), min_near)                                          # :140-141 no_grad
    nears, fars = torch.from_numpy(n).to(dt).unsqueeze(-1), torch.from_numpy(f).to(dt).unsqueeze(-1)
    z_vals = torch.linspace(0.0, 1.0, num_steps, dtype=dt).unsqueeze(0).expand(N, num_steps)
    z_vals = nears + (fars - nears) * z_vals
    sample_dist = (fars - nears) / num_steps
    if perturb_u is not None:
        z_vals = z_vals + (perturb_u - 0.5) * sample_dist
    xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
    xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
    dens = {k: v.view(N, num_steps, -1) for k, v in field.density(xyzs.reshape(-1, 3)).items()}

    if upsample_steps > 0:                                              # :172-204
        with torch.no_grad():
            w, deltas = _weights(z_vals, sample_dist, dens["sigma"].squeeze(-1), density_scale)
            mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
            new_z = sample_pdf(mid, w[:, 1:-1], upsample_steps, det=not training, u=pdf_u)
            new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z.unsqueeze(-1)
            new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])
        new_dens = {k: v.view(N, upsample_steps, -1) for k, v in field.density(new_xyzs.reshape(-1, 3)).items()}
        z_vals, order = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
        xyzs = torch.gather(torch.cat([xyzs, new_xyzs], dim=1), 1, order.unsqueeze(-1).expand(-1, -1, 3))
        for k in dens:
            both = torch.cat([dens[k], new_dens[k]], dim=1)
            dens[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))

    weights, _ = _weights(z_vals, sample_dist, dens["sigma"].squeeze(-1), density_scale)
    T = z_vals.shape[1]
    dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
    flat = {k: v.reshape(-1, v.shape[-1]) for k, v in dens.items()}
    mask = weights > 1e-4                                               # :217
    rgbs = field.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **flat).view(N, T, 3)
    weights_sum = weights.sum(dim=-1)
    ori_z = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
    depth = torch.sum(weights * ori_z, dim=-1)
    image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    return {"image": image, "depth": depth, "weights_sum": weights_sum, "weights": weights, "mask": mask}


# ------------------------------------------------------------------------------------------------------------------------
# run_cuda, training branch
# ------------------------------------------------------------------------------------------------------------------------
class _CompositeTrain(torch.autograd.Function):
    """_composite_rays_train (raymarching/raymarching.py:233-283) over the C restatement of raymarching.cu:506-699."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays):
        ws, depth, image = O.composite_rays_train_forward(sigmas.detach().numpy(), rgbs.detach().numpy(), deltas.numpy(), rays.numpy())
        ctx.save_for_backward(sigmas.detach(), rgbs.detach(), deltas, rays, torch.from_numpy(ws), torch.from_numpy(image))
        return torch.from_numpy(ws), torch.from_numpy(depth), torch.from_numpy(image)

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):                          # grad_depth is ignored, as in the reference (:268)
        sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
        gs, gc = O.composite_rays_train_backward(g_ws.contiguous().numpy(), g_image.contiguous().numpy(), sigmas.numpy(), rgbs.numpy(),
                                                 deltas.numpy(), rays.numpy(), ws.numpy(), image.numpy())
        return torch.from_numpy(gs), torch.from_numpy(gc), None, None


def composite_rays_train(sigmas, rgbs, deltas, rays):
    return _CompositeTrain.apply(sigmas.float(), rgbs.float(), deltas, rays)


def run_cuda_train(field, rays_o, rays_d, bitfield, bound, cascade, H=128, min_near=0.2, density_scale=1.0, dt_gamma=0.0, max_steps=1024,
                   bg_color=1.0, perturb=False, mean_count=-1, force_all_rays=False, counter=None):
    """nerf/renderer.py:282-323.  rays numpy [N,3].  Returns dict(image, depth, weights_sum (torch, differentiable w.r.t. the field's
    parameters), xyzs, dirs, deltas, rays (numpy), sigmas, rgbs (torch))."""
    rays_o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(rays_o, rays_d, aabb, min_near)
    if counter is None:
        counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = O.march_rays_train(rays_o, rays_d, bound, bitfield, cascade, H, nears, fars, counter, mean_count, perturb,
                                                  128, force_all_rays, dt_gamma, max_steps)
    sigmas, rgbs = field(torch.from_numpy(xyzs), torch.from_numpy(dirs))
    sigmas = density_scale * sigmas
    ws, depth, image = composite_rays_train(sigmas, rgbs, torch.from_numpy(deltas), torch.from_numpy(rays))
    image = image + (1 - ws).unsqueeze(-1) * bg_color
    depth = torch.clamp(depth - torch.from_numpy(nears), min=0) / torch.from_numpy(fars - nears)
    return dict(image=image, depth=depth, weights_sum=ws, xyzs=xyzs, dirs=dirs, deltas=deltas, rays=rays, sigmas=sigmas, rgbs=rgbs,
                counter=counter)


# ------------------------------------------------------------------------------------------------------------------------
# density-grid maintenance
# ------------------------------------------------------------------------------------------------------------------------
GRID_RNG_STRIDE = 16                                                   # pcg32 draws reserved per sample (csrc/density_grid.hip)


def grid_update_randoms(seed, iteration, cascade, H, partial, n_occ=None):
    """The random numbers the NATIVE ops draw (csrc/density_grid.hip), restated on the oracle's pcg32 (raymarching/src/pcg32.h:44-205):
    one stream per update, `pcg32(seed, seq = iteration)`, sample e uses draws [16 e, 16 e + 16):
      full sweep   : e = cas * H^3 + morton index;           draws 0..2 = jitter (next_float, pcg32.h:107-116)
      partial sweep: e = cas * (H^3/4) + i, i < H^3/4;       draws 0..2 = random cell (next_uint * H >> 32 per axis), 3 = pick among the
                     occupied cells (next_uint * n_occ >> 32), 4..6 = jitter of the random cell, 7..9 = jitter of the occupied cell
    Returns dict of numpy arrays in that layout (only what the mode uses)."""
    H3 = H ** 3
    n = cascade * (H3 // 4 if partial else H3)
    u = O.pcg32_stream(seed, iteration, n * GRID_RNG_STRIDE).reshape(n, GRID_RNG_STRIDE)
    flt = ((u >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    if not partial:
        return {"noise": flt[:, 0:3].reshape(cascade, H3, 3)}
    N = H3 // 4
    coords = ((u[:, 0:3].astype(np.uint64) * np.uint64(H)) >> np.uint64(32)).astype(np.int32).reshape(cascade, N, 3)
    out = {"coords": coords, "pick_u": u[:, 3].reshape(cascade, N), "noise_rand": flt[:, 4:7].reshape(cascade, N, 3),
           "noise_occ": flt[:, 7:10].reshape(cascade, N, 3)}
    if n_occ is not None:
        out["pick"] = np.stack([((out["pick_u"][c].astype(np.uint64) * np.uint64(max(int(n_occ[c]), 1))) >> np.uint64(32)).astype(np.int64)
                                for c in range(cascade)])
    return out


def grid_sample_positions(coords, noise, cas, bound, H):
    """nerf/renderer.py:473-483 in binary32, one rounding per written operation: coords int [n,3], noise [n,3] in [0,1)."""
    f = np.float32
    xyzs = f(2) * coords.astype(f) / f(H - 1) - f(1)
    b = float(min(2 ** cas, bound))
    half, span = f(b / H), f(b - b / H)                                  # Python floats in the reference: double, then narrowed
    return (xyzs * span + (noise.astype(f) * f(2) - f(1)) * half).astype(f)


def update_extra_state(density_fn, density_grid, bound, density_thresh, iter_density, randoms, density_scale=1.0, decay=0.95, H=128):
    """nerf/renderer.py:446-531 (grid part).  density_fn(xyzs [n,3] f32) -> sigma [n] f32 (before density_scale).
    `randoms`: full sweep {"noise" [cas,H^3,3] indexed by Morton index}; partial {"coords" [cas,N,3], "pick" [cas,N] (index into the
    ascending list of occupied cells), "noise_rand", "noise_occ" [cas,N,3]}.  Duplicate indices of the partial sweep resolve to the
    LARGEST sigma (the reference's `tmp_grid[cas, indices] = sigmas` keeps an arbitrary one of them; the native op keeps the maximum).
    Returns (density_grid, bitfield, mean_density, density_thresh used, tmp_grid)."""
    f = np.float32
    cascade = density_grid.shape[0]
    H3 = H ** 3
    grid = np.array(density_grid, dtype=f)
    tmp = -np.ones_like(grid)
    if iter_density < 16:
        ar = np.arange(H, dtype=np.int32)
        coords = np.stack(np.meshgrid(ar, ar, ar, indexing="ij"), axis=-1).reshape(-1, 3)               # custom_meshgrid order (:460)
        indices = O.morton3D(coords).astype(np.int64)
        for cas in range(cascade):
            pts = grid_sample_positions(coords, randoms["noise"][cas][indices], cas, bound, H)
            tmp[cas, indices] = density_fn(pts).astype(f) * f(density_scale)
    else:
        for cas in range(cascade):
            coords = np.asarray(randoms["coords"][cas], np.int32)
            indices = O.morton3D(coords).astype(np.int64)
            occ = np.flatnonzero(grid[cas] > 0)                                                           # :493
            if occ.size:
                occ_idx = occ[np.asarray(randoms["pick"][cas], np.int64)]
                occ_coords = O.morton3D_invert(occ_idx.astype(np.int32))
                coords = np.concatenate([coords, occ_coords])
                indices = np.concatenate([indices, occ_idx])
                noise = np.concatenate([randoms["noise_rand"][cas], randoms["noise_occ"][cas]])
            else:
                noise = randoms["noise_rand"][cas]
            pts = grid_sample_positions(coords, noise, cas, bound, H)
            sig = density_fn(pts).astype(f) * f(density_scale)
            keep = sig >= 0                                                                               # NaN / negative never win the max
            np.maximum.at(tmp[cas], indices[keep], sig[keep])
    valid = (grid >= 0) & (tmp >= 0)                                                                      # :523-524
    grid[valid] = np.maximum(grid[valid] * f(decay), tmp[valid])
    mean_density = float(f(np.mean(np.clip(grid, 0, None), dtype=np.float64)))                            # :525 (float64 sum, one rounding)
    thresh = min(mean_density, density_thresh)
    bitfield = O.packbits(grid, thresh)
    return grid, bitfield, mean_density, thresh, tmp


def mark_untrained_grid(density_grid, poses, intrinsic, bound, H=128):
    """nerf/renderer.py:381-442: count, per cell and cascade, the cameras whose frustum holds the cell centre; 0 -> density -1.
    binary32 with one rounding per operation in the order csrc/density_grid.hip uses: ((p - t) . R[:,k]) summed x, y, z."""
    f = np.float32
    fx, fy, cx, cy = (float(v) for v in intrinsic)
    tan_x, tan_y = f(cx / fx), f(cy / fy)                                 # Python floats in the reference (:432-433): double, then narrowed
    poses = np.asarray(poses, f)
    cascade = density_grid.shape[0]
    grid = np.array(density_grid, dtype=f)
    idx = np.arange(H ** 3, dtype=np.int32)
    coords = O.morton3D_invert(idx)
    world = f(2) * coords.astype(f) / f(H - 1) - f(1)
    for cas in range(cascade):
        b = float(min(2 ** cas, bound))
        half = f(b / H)
        p = world * f(b - b / H)
        count = np.zeros(H ** 3, np.int64)
        for pose in poses:
            q = p - pose[:3, 3]
            cam = [(q[:, 0] * pose[0, k] + q[:, 1] * pose[1, k]) + q[:, 2] * pose[2, k] for k in range(3)]
            ok = (cam[2] > 0) & (np.abs(cam[0]) < tan_x * cam[2] + half * f(2)) & (np.abs(cam[1]) < tan_y * cam[2] + half * f(2))
            count += ok
        grid[cas, count == 0] = -1
    return grid


def psnr_meter(preds, truths):
    """PSNRMeter (nerf/utils.py:185-219): per update -10 log10(mean((p - t)^2)), measure() = arithmetic mean over updates."""
    vals = [-10 * np.log10(np.mean((np.asarray(p, np.float64) - np.asarray(t, np.float64)) ** 2)) for p, t in zip(preds, truths)]
    return float(np.mean(vals))
