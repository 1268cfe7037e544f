"""Drop-in `raymarching` package: the autograd Functions of the reference's raymarching/raymarching.py, backed
by libngp_hip.so (csrc/raymarching.hip) instead of the `_raymarching` CUDA extension.

Same names, argument order, defaults, dtype casts (float32 via custom_fwd), return values and in-place
behaviour as the reference (file:line cited per function).  Differences, all supersets:
  * kernels run on torch's current HIP stream (the reference used the legacy default stream);
  * march_rays_train allocates slots in ray order (deterministic) instead of atomic arrival order.
"""
import ctypes
import functools
import threading
import time

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import ngp_hip as _hip


def _f32_forward(fwd):
    """`custom_fwd(device_type="cuda", cast_inputs=torch.float32)` for a Function that has NO backward: under autocast the half / bfloat16 CUDA tensor
    arguments are converted to float32, exactly the set torch.amp casts.  The stock decorator walks every argument through a recursive `_cast` and
    enters an `autocast(enabled=False)` context: ~25 us of host time per call, three calls per iteration of the inference loop, on the critical path right
    after that loop's one synchronisation (profiles/r16_dropin_summary.md).  The bodies below call native code only, so there is nothing for a disabled
    autocast to protect, and without a backward nothing reads `ctx._fwd_used_autocast`."""
    @functools.wraps(fwd)
    def wrapper(ctx, *args):
        if torch.is_autocast_enabled("cuda"):
            args = tuple(a.float() if (isinstance(a, torch.Tensor) and a.is_cuda and (a.dtype == torch.float16 or a.dtype == torch.bfloat16)) else a
                         for a in args)
        return fwd(ctx, *args)
    return wrapper


def _public(fn_cls):
    """the package-level callable of a forward-only Function: `Function.apply` when autograd is recording (so that differentiating through it fails as in
    the reference), the forward body itself under no_grad (inference loops: ~10 us of autograd bookkeeping per call saved)"""
    apply, fwd = fn_cls.apply, fn_cls.forward

    @functools.wraps(fwd)
    def call(*args):
        if torch.is_grad_enabled():
            return apply(*args)
        return fwd(None, *args)
    return call

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train", "march_rays", "composite_rays", "compact_alive"]


def _rays(t):
    """[..., 3] -> contiguous float32 [N, 3] on the GPU (the reference moves CPU inputs over: raymarching.py:34-35).  The kernels read float32:
    custom_fwd(cast_inputs=float32) converts half / bfloat16 inputs under autocast but never float64 ones (torch.amp's cast skips them) and nothing
    outside autocast, where the reference would hand the kernel another dtype's bytes; here any other floating dtype is converted."""
    if not t.is_cuda:
        t = t.cuda()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous().view(-1, 3)


class _near_far_from_aabb(Function):
    """reference: raymarching/raymarching.py:19-49"""

    @staticmethod
    @_f32_forward
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        aabb = aabb.to(rays_o.device).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        _hip.check(_hip.lib().ngp_near_far_from_aabb(_hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(aabb), N, min_near,
                                                     _hip.ptr(nears), _hip.ptr(fars), _hip.stream()), "near_far_from_aabb")
        return nears, fars


near_far_from_aabb = _public(_near_far_from_aabb)


class _sph_from_ray(Function):
    """reference: raymarching/raymarching.py:52-80"""

    @staticmethod
    @_f32_forward
    def forward(ctx, rays_o, rays_d, radius):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        _hip.check(_hip.lib().ngp_sph_from_ray(_hip.ptr(rays_o), _hip.ptr(rays_d), radius, N, _hip.ptr(coords), _hip.stream()),
                   "sph_from_ray")
        return coords


sph_from_ray = _public(_sph_from_ray)


class _morton3D(Function):
    """reference: raymarching/raymarching.py:83-102"""

    @staticmethod
    def forward(ctx, coords):
        if not coords.is_cuda:
            coords = coords.cuda()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        c = coords.int().contiguous()
        _hip.check(_hip.lib().ngp_morton3D(_hip.ptr(c), N, _hip.ptr(indices), _hip.stream()), "morton3D")
        return indices


morton3D = _public(_morton3D)


class _morton3D_invert(Function):
    """reference: raymarching/raymarching.py:106-124"""

    @staticmethod
    def forward(ctx, indices):
        if not indices.is_cuda:
            indices = indices.cuda()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        i = indices.int().contiguous()
        _hip.check(_hip.lib().ngp_morton3D_invert(_hip.ptr(i), N, _hip.ptr(coords), _hip.stream()), "morton3D_invert")
        return coords


morton3D_invert = _public(_morton3D_invert)


class _packbits(Function):
    """reference: raymarching/raymarching.py:129-153 (writes into `bitfield` when given)"""

    @staticmethod
    @_f32_forward
    def forward(ctx, grid, thresh, bitfield=None):
        if not grid.is_cuda:
            grid = grid.cuda()
        grid = grid.contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        _hip.check(_hip.lib().ngp_packbits(_hip.ptr(grid), N, thresh, _hip.ptr(bitfield), _hip.stream()), "packbits")
        return bitfield


packbits = _public(_packbits)


class _march_rays_train(Function):
    """reference: raymarching/raymarching.py:161-228 (same M / mean_count / align logic, same D2H read of the counter)"""

    @staticmethod
    @_f32_forward
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        if not density_bitfield.is_cuda:
            density_bitfield = density_bitfield.cuda()
        density_bitfield = density_bitfield.contiguous()
        dev = rays_o.device

        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count

        # the reference allocates these with torch.zeros (raymarching.py:205-207: three whole-buffer fills per step); the native call zeroes the slots no
        # ray fills itself (ngp_march_rays_train_filled: they form one tail), so the contents are the same
        xyzs = torch.empty(M, 3, dtype=rays_o.dtype, device=dev)
        dirs = torch.empty(M, 3, dtype=rays_o.dtype, device=dev)
        deltas = torch.empty(M, 2, dtype=rays_o.dtype, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)

        L = _hip.lib()
        # with room for every sample's t (N * max_steps floats, up to 256 MiB) the second pass does not march again
        full = L.ngp_march_rays_train_workspace_full(N, max_steps)
        ws = _hip.workspace(full if full <= (256 << 20) else L.ngp_march_rays_train_workspace(N), dev)
        _hip.check(L.ngp_march_rays_train_filled(_hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(density_bitfield), bound, dt_gamma,
                                          max_steps, N, C, H, M, _hip.ptr(nears.contiguous()), _hip.ptr(fars.contiguous()),
                                          _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays),
                                          _hip.ptr(step_counter), int(perturb), _hip.ptr(ws), ws.numel(), _hip.stream()),
                   "march_rays_train")

        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    """reference: raymarching/raymarching.py:233-283 (grad_depth ignored: :270)"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        _hip.check(_hip.lib().ngp_composite_rays_train_forward(_hip.ptr(sigmas), _hip.ptr(rgbs), _hip.ptr(deltas), _hip.ptr(rays),
                                                               M, N, _hip.ptr(weights_sum), _hip.ptr(depth), _hip.ptr(image),
                                                               _hip.stream()), "composite_rays_train_forward")
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N]
        return weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        grad_weights_sum = grad_weights_sum.contiguous()
        grad_image = grad_image.contiguous()
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N = ctx.dims
        # (a caller that runs this body directly may hand in buffers it has already cleared: ngp/train.py's direct step clears them in its loss-head launch)
        grad_sigmas, grad_rgbs = getattr(ctx, "cleared_grads", None) or (torch.zeros_like(sigmas), torch.zeros_like(rgbs))
        _hip.check(_hip.lib().ngp_composite_rays_train_backward(_hip.ptr(grad_weights_sum), _hip.ptr(grad_image), _hip.ptr(sigmas),
                                                                _hip.ptr(rgbs), _hip.ptr(deltas), _hip.ptr(rays),
                                                                _hip.ptr(weights_sum), _hip.ptr(image), M, N,
                                                                _hip.ptr(grad_sigmas), _hip.ptr(grad_rgbs), _hip.stream()),
                   "composite_rays_train_backward")
        return grad_sigmas, grad_rgbs, None, None


composite_rays_train = _composite_rays_train.apply


class _march_rays(Function):
    """reference: raymarching/raymarching.py:292-335 (M padded past the next multiple of `align`)"""

    @staticmethod
    @_f32_forward
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        # the reference allocates with torch.zeros (three fill launches per loop iteration); here the march kernel writes the
        # zeros of the slots no ray reaches and of the alignment rows itself (ngp_march_rays_fill): same contents, one launch
        xyzs = torch.empty(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        dirs = torch.empty(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        deltas = torch.empty(M, 2, dtype=rays_o.dtype, device=rays_o.device)
        L = _hip.lib()
        ws = _hip.workspace(L.ngp_march_rays_workspace(C, H), rays_o.device)     # coarse occupancy map, rebuilt by the call
        with _hip.timed("march_rays"):
            _hip.check(L.ngp_march_rays_fill(n_alive, n_step, _hip.ptr(rays_alive), _hip.ptr(rays_t), _hip.ptr(rays_o),
                                             _hip.ptr(rays_d), bound, dt_gamma, max_steps, C, H, _hip.ptr(density_bitfield),
                                             _hip.ptr(near), _hip.ptr(far), _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas),
                                             M, int(perturb), _hip.ptr(ws), ws.numel(), _hip.stream()), "march_rays")
        return xyzs, dirs, deltas


march_rays = _public(_march_rays)


class _composite_rays(Function):
    """reference: raymarching/raymarching.py:340-359 (returns an empty tuple; mutates its arguments in place)"""

    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        # (custom_fwd(cast_inputs=float32) in the reference: under autocast half arguments are widened.  Half COLOURS -- what a field under autocast
        #  returns -- are widened by the kernel's own loads instead: the same values, one elementwise launch per iteration of the inference loop less)
        half_rgb = rgbs.dtype == torch.float16 and rgbs.is_cuda
        if torch.is_autocast_enabled("cuda"):
            sigmas, deltas = (a.float() if a.dtype in (torch.float16, torch.bfloat16) else a for a in (sigmas, deltas))
            if not half_rgb and rgbs.dtype in (torch.float16, torch.bfloat16):
                rgbs = rgbs.float()
        fn = _hip.lib().ngp_composite_rays_half if half_rgb else _hip.lib().ngp_composite_rays
        _hip.check(fn(n_alive, n_step, _hip.ptr(rays_alive), _hip.ptr(rays_t), _hip.ptr(sigmas.contiguous()), _hip.ptr(rgbs.contiguous()), _hip.ptr(deltas),
                      _hip.ptr(weights_sum), _hip.ptr(depth), _hip.ptr(image), _hip.stream()), "composite_rays")
        return tuple()


composite_rays = _public(_composite_rays)


_HOST_PAIRS = {}          # device index -> [ctypes int32 pair in pinned coherent memory, device address, last sequence number]


def _host_pair(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    entry = _HOST_PAIRS.get(key)
    if entry is None:
        p = ctypes.c_void_p()
        _hip.check(_hip.lib().ngp_host_words_alloc(2, ctypes.byref(p)), "host_words_alloc")
        entry = _HOST_PAIRS[key] = [(ctypes.c_int32 * 2).from_address(p.value), p, 0, threading.Lock()]
    return entry


def compact_alive(rays_alive, n_alive=None, count=False):
    """Stable compaction `rays_alive[rays_alive >= 0]` (nerf/renderer.py:365) done on the device.
    Returns (compacted [n_alive] int32, count [1] int32 device tensor); only the first count entries are valid.
    count=True: also the count as a Python int, third, WITHOUT a stream synchronisation -- the kernel stores it in two pinned words the host polls
    (ngp_compact_alive_publish); the reference's boolean-mask indexing pays a synchronisation + copy here, once per iteration of the inference loop."""
    n = rays_alive.shape[0] if n_alive is None else n_alive
    out = torch.empty(max(n, 1), dtype=torch.int32, device=rays_alive.device)
    cnt = torch.empty(1, dtype=torch.int32, device=rays_alive.device)
    L = _hip.lib()
    ws = _hip.workspace(L.ngp_compact_alive_workspace(n), rays_alive.device)
    if not count:
        _hip.check(L.ngp_compact_alive(_hip.ptr(rays_alive), n, _hip.ptr(out), _hip.ptr(cnt), _hip.ptr(ws), ws.numel(), _hip.stream()),
                   "compact_alive")
        return out, cnt
    entry = _host_pair(rays_alive.device)
    with entry[3]:                                  # one pair per device: a second thread's call waits for this one's count
        pair, seq = entry[0], (entry[2] % 0x7FFFFFF0) + 1
        entry[2] = seq
        _hip.check(L.ngp_compact_alive_publish(_hip.ptr(rays_alive), n, _hip.ptr(out), _hip.ptr(cnt), entry[1], seq, _hip.ptr(ws), ws.numel(), _hip.stream()),
                   "compact_alive_publish")
        spins, deadline = 0, None
        while pair[1] != seq:
            spins += 1
            if spins & 0xFFF == 0:                  # (a launch that never ran must not hang the caller: after 5 s fall back to the synchronising read)
                now = time.perf_counter()
                deadline = deadline or now + 5.0
                if now > deadline:
                    return out, cnt, int(cnt.item())
        return out, cnt, int(pair[0])
