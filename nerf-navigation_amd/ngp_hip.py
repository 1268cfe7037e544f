"""ctypes binding of libngp_hip.so (include/ngp_hip.h) for torch tensors.

This is plumbing only: torch supplies device memory and the current HIP stream, every kernel lives in the
shared library.  There is NO fallback: if the library is missing or a call fails this module raises, so a GPU
test can never pass on a silent CPU/eager path.
"""
import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NGP_HIP_LIB selects another build of the SAME library (A/B runs of kernel variants); it is never a fallback
LIB_PATH = os.environ.get("NGP_HIP_LIB") or os.path.join(_HERE, "lib", "libngp_hip.so")
CSRC = os.path.join(_HERE, "csrc")

F32, F16 = 0, 1

c_u32, c_f32, c_int, c_vp, c_sz = ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
c_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/ngp_hip.h one to one
_SIGNATURES = {
    "ngp_abi_version": (c_int, []),
    "ngp_last_error": (ctypes.c_char_p, []),
    "ngp_near_far_from_aabb": (c_int, [c_vp, c_vp, c_vp, c_u32, c_f32, c_vp, c_vp, c_vp]),
    "ngp_sph_from_ray": (c_int, [c_vp, c_vp, c_f32, c_u32, c_vp, c_vp]),
    "ngp_morton3D": (c_int, [c_vp, c_u32, c_vp, c_vp]),
    "ngp_morton3D_invert": (c_int, [c_vp, c_u32, c_vp, c_vp]),
    "ngp_packbits": (c_int, [c_vp, c_u32, c_f32, c_vp, c_vp]),
    "ngp_march_rays_train_workspace": (c_sz, [c_u32]),
    "ngp_march_rays_train_workspace_full": (c_sz, [c_u32, c_u32]),
    "ngp_march_rays_train": (c_int, [c_vp, c_vp, c_vp, c_f32, c_f32, c_u32, c_u32, c_u32, c_u32, c_u32, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_sz, c_vp]),
    "ngp_march_rays_train_filled": (c_int, [c_vp, c_vp, c_vp, c_f32, c_f32, c_u32, c_u32, c_u32, c_u32, c_u32, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_sz, c_vp]),
    "ngp_march_set_wave_per_ray": (c_int, [c_int]),
    "ngp_composite_set_scan": (c_int, [c_int]),
    "ngp_march_set_occupied_box": (c_int, [c_int]),
    "ngp_composite_rays_train_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp]),
    "ngp_composite_rays_train_backward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp]),
    "ngp_march_rays": (c_int, [c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp,
                               c_vp, c_vp, c_vp, c_u32, c_vp]),
    "ngp_march_rays_workspace": (c_sz, [c_u32, c_u32]),
    "ngp_march_rays_fill": (c_int, [c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp,
                                    c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_sz, c_vp]),
    "ngp_composite_rays": (c_int, [c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ngp_composite_rays_half": (c_int, [c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ngp_compact_alive_workspace": (c_sz, [c_u32]),
    "ngp_compact_alive": (c_int, [c_vp, c_u32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_compact_alive_publish": (c_int, [c_vp, c_u32, c_vp, c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "ngp_host_words_alloc": (c_int, [c_u32, c_vp]),
    "ngp_host_words_free": (c_int, [c_vp]),
    "ngp_density_grid_points": (c_u32, [c_u32, c_u32, c_int]),
    "ngp_density_grid_workspace": (c_sz, [c_u32, c_u32]),
    "ngp_density_grid_sample": (c_int, [c_vp, c_u32, c_u32, c_f32, c_int, c_u64, c_u64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_density_grid_update": (c_int, [c_vp, c_vp, c_u32, c_f32, c_f32, c_f32, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_mark_untrained_grid": (c_int, [c_vp, c_u32, c_f32, c_f32, c_f32, c_f32, c_u32, c_u32, c_f32, c_vp, c_vp]),
    "ngp_grid_encode_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_int, c_vp,
                                        c_u32, c_int, c_int, c_vp]),
    "ngp_grid_encode_forward_rows": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_u32, c_int, c_int, c_vp]),
    "ngp_grid_encode_backward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_int,
                                         c_vp, c_vp, c_u32, c_int, c_int, c_vp]),
    "ngp_grid_scatter_binned_workspace": (c_sz, [c_u32, c_u32]),
    "ngp_grid_scatter_binned": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_f32, c_u32, c_u32, c_u32, c_int, c_int, c_f32, c_vp, c_sz, c_vp]),
    "ngp_grid_scatter_binned_phase": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_u32, c_u32, c_int, c_int, c_f32, c_vp, c_sz, c_vp]),
    "ngp_grid_scatter_binned_listed": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_f32, c_u32, c_u32, c_u32, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_grid_scatter_binned_phase_listed": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_u32, c_u32, c_int, c_int, c_f32, c_vp, c_vp,
                                                     c_vp, c_sz, c_vp]),
    "ngp_adam_step": (c_int, [c_vp, c_u32, c_vp, c_vp, c_vp]),
    "ngp_train_mix_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_f32, c_u32, c_vp, c_vp, c_vp]),
    "ngp_train_mix_backward": (c_int, [c_vp, c_vp, c_u32, c_f32, c_u32, c_vp, c_vp]),
    "ngp_train_head_direct": (c_int, [c_vp, c_vp, c_vp, c_u32, c_f32, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_sz, c_vp]),
    "ngp_mse_head_workspace": (c_sz, []),
    "ngp_mse_head_forward": (c_int, [c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_mse_head_backward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_vp, c_vp]),
    "ngp_sh_encode_forward": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_int, c_vp, c_vp]),
    "ngp_sh_encode_backward": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp]),
    "ngp_freq_encode_forward": (c_int, [c_vp, c_u32, c_u32, c_u32, c_u32, c_vp, c_vp]),
    "ngp_freq_encode_backward": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_vp, c_vp]),
    "ngp_ffmlp_forward": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp]),
    "ngp_ffmlp_inference": (c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32, c_vp, c_vp, c_vp]),
    "ngp_ffmlp_backward_workspace": (c_sz, [c_u32, c_u32, c_u32, c_u32]),
    "ngp_ffmlp_backward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32, c_u32, c_int,
                                   c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_allocate_splitk": (c_int, [c_sz]),
    "ngp_free_splitk": (c_int, []),
    "ngp_field_forward": (c_int, [c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp]),
    "ngp_field_forward_half": (c_int, [c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp]),
    "ngp_field_density_workspace": (c_sz, [c_u32]),
    "ngp_field_density": (c_int, [c_vp, c_vp, c_u32, c_vp, c_vp, c_sz, c_vp]),
    "ngp_field_train_saved_bytes": (c_sz, [c_u32]),
    "ngp_field_train_workspace": (c_sz, [c_u32]),
    "ngp_field_train_forward": (c_int, [c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_field_train_backward": (c_int, [c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "ngp_field_train_live_list": (c_int, [c_vp, c_u32, c_vp, c_vp]),
    "ngp_render_frame_workspace": (c_sz, [c_u32]),
    "ngp_render_set_block_skip": (c_int, [c_int]),
    "ngp_render_set_occupied_box": (c_int, [c_int]),
    "ngp_field_train_set_two_pass": (c_int, [c_int]),
    "ngp_field_train_set_live_only": (c_int, [c_int]),
    "ngp_render_frame": (c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_f32, c_vp, c_u32, c_u32, c_f32, c_u32, c_vp,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_grid_encode_backward_inputs": (c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_f32, c_u32, c_vp, c_u32, c_int, c_int, c_vp]),
    "ngp_nav_field_workspace": (c_sz, []),
    "ngp_nav_field_prepare": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "ngp_nav_density_forward": (c_int, [c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp]),
    "ngp_nav_density_backward": (c_int, [c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp]),
    "ngp_nav_density_value_jac": (c_int, [c_vp, c_vp, c_vp, c_u32, c_vp, c_vp, c_vp, c_vp]),
    "ngp_nav_run_saved_bytes": (c_sz, [c_u32, c_u32]),
    "ngp_nav_run_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_nav_run_backward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "ngp_get_rays": (c_int, [c_vp, c_vp, c_u32, c_u32, c_vp, c_u32, c_vp, c_vp, c_vp]),
    "ngp_render_frame_camera": (c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_vp, c_f32, c_vp, c_u32, c_u32, c_f32, c_u32, c_vp,
                                        c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ngp_render_frames_workspace": (c_sz, [c_u32, c_u32]),
    "ngp_render_frames_camera": (c_int, [c_vp, c_vp, c_u32, c_vp, c_u32, c_u32, c_vp, c_f32, c_vp, c_u32, c_u32, c_f32, c_u32, c_vp,
                                         c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
}

EXPORTS = tuple(_SIGNATURES)


class ngp_field_t(ctypes.Structure):
    """include/ngp_hip.h: ngp_field_t"""
    _fields_ = [("embeddings", c_vp), ("offsets", c_vp), ("sigma_weights", c_vp), ("color_weights", c_vp),
                ("L", c_u32), ("H", c_u32), ("S", c_f32), ("bound", c_f32), ("density_scale", c_f32)]


class ngp_adam_tensor_t(ctypes.Structure):
    """include/ngp_hip.h: ngp_adam_tensor_t"""
    _fields_ = [("param", c_vp), ("grad", c_vp), ("exp_avg", c_vp), ("exp_avg_sq", c_vp), ("half_copy", c_vp), ("n", ctypes.c_uint64), ("lr", ctypes.c_double)]


class ngp_adam_hyper_t(ctypes.Structure):
    """include/ngp_hip.h: ngp_adam_hyper_t"""
    _fields_ = [("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("growth_factor", ctypes.c_double),
                ("backoff_factor", ctypes.c_double), ("growth_interval", ctypes.c_int32), ("scaler_enabled", ctypes.c_int32)]


ADAM_MAX_TENSORS = 16
ADAM_STATE_WORDS = 32
ADAM_STATE_SCALE, ADAM_STATE_GROWTH_TRACKER, ADAM_STATE_STEP, ADAM_STATE_FOUND_INF = 0, 1, 2, 4


class ngp_nav_field_t(ctypes.Structure):
    """include/ngp_hip.h: ngp_nav_field_t"""
    _fields_ = [("embeddings", c_vp), ("offsets_host", c_vp), ("sigma_w0", c_vp), ("sigma_w1", c_vp), ("color_w0", c_vp), ("color_w1", c_vp),
                ("color_w2", c_vp), ("L", c_u32), ("H", c_u32), ("S", c_f32), ("bound", c_f32), ("density_scale", c_f32)]


def build(verbose=False):
    """Compile libngp_hip.so for gfx950 with hipcc (csrc/Makefile).  Cross-compiles without a GPU."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j", "8"], stdout=out)
    return LIB_PATH


_lib = None
_ACTIVE = threading.local()


class _GuardedCall:
    """One entry point of the library behind a DEVICE GUARD (SURVEY 8b "Threading / streams": the reference launches on the legacy default
    stream of whatever device is current; the pybind shims guard with c10's OptionalHIPGuard, this is the ctypes side of the same rule).
    The device is that of the first tensor argument (`ptr(t)` objects carry their tensor).  When it is not the current device the call runs
    inside `torch.cuda.device(that device)`, and the `stream()` argument -- resolved only when ctypes converts it, i.e. inside the guard --
    is torch's current stream OF THAT DEVICE.  Tensors on two different devices in one call raise instead of launching."""
    __slots__ = ("fn", "name")

    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *args):
        dev = None
        for a in args:
            if type(a) is _DevPtr:
                d = a.tensor.device
                if dev is None:
                    dev = d
                elif d != dev:
                    raise RuntimeError(f"libngp_hip {self.name}: tensors on {dev} and {d} in one call")
        if dev is None or dev.type != "cuda":
            return self.fn(*args)
        _ACTIVE.device = dev                                 # `stream()` arguments resolve to this device's current stream
        try:
            if dev.index == torch.cuda.current_device():
                return self.fn(*args)
            with torch.cuda.device(dev):
                return self.fn(*args)
        finally:
            _ACTIVE.device = None


class _GuardedLib:
    """attribute access returns the guarded entry points; `raw` is the ctypes handle"""

    def __init__(self, handle):
        self.raw = handle

    def __getattr__(self, name):
        call = _GuardedCall(getattr(self.raw, name), name)
        self.__dict__[name] = call
        return call


def lib():
    """The loaded library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = _GuardedLib(handle)
        if os.environ.get("NGP_FT_TWO_PASS") in ("0", "1"):              # A/B switch of the training forward (tools, bench.py --mode train); default: two passes
            handle.ngp_field_train_set_two_pass(int(os.environ["NGP_FT_TWO_PASS"]))
        if os.environ.get("NGP_OCC_BOX") in ("0", "1"):                  # ... of the march limit at the occupied box (frame kernel and per-op march kernels)
            handle.ngp_render_set_occupied_box(int(os.environ["NGP_OCC_BOX"]))
            handle.ngp_march_set_occupied_box(int(os.environ["NGP_OCC_BOX"]))
        if os.environ.get("NGP_FT_LIVE_ONLY") in ("0", "1"):             # ... of the training backward (live samples only | all samples)
            handle.ngp_field_train_set_live_only(int(os.environ["NGP_FT_LIVE_ONLY"]))
    return _lib


# Optional live timing of individual native calls (bench.py's roofline of a kernel inside a multi-kernel step): set
# `ngp_hip.TIMERS = {}` and every `with timed(name):` block records a HIP event pair on the current stream; None = no overhead.
TIMERS = None


class timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TIMERS is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if TIMERS is not None:
            self.ev[1].record()
            TIMERS.setdefault(self.name, []).append(self.ev)
        return False


def timer_ms(name):
    """mean / count of the recorded intervals of `name` (synchronises)"""
    ev = (TIMERS or {}).get(name, [])
    if not ev:
        return None, 0
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    return sum(ms) / len(ms), len(ms)


def check(rc, what=""):
    if rc != 0:
        msg = lib().ngp_last_error().decode(errors="replace")
        raise RuntimeError(f"libngp_hip {what} failed ({rc}): {msg}")


class _DevPtr:
    """A device pointer that keeps its tensor alive: ctypes reads `_as_parameter_`, and the argument tuple holds this
    object until the call returns, so `ptr(x.contiguous())` can never hand the kernel freed memory."""
    __slots__ = ("tensor", "_as_parameter_")

    def __init__(self, tensor):
        self.tensor = tensor
        self._as_parameter_ = ctypes.c_void_p(tensor.data_ptr())


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return _DevPtr(t)


def _raw_stream(index):
    try:
        return torch._C._cuda_getCurrentRawStream(index)
    except AttributeError:                                   # a torch without the raw accessor
        return torch.cuda.current_stream(index).cuda_stream


class _CurrentStream:
    """The HIP stream torch is enqueueing on FOR THE CURRENT DEVICE AT THE MOMENT OF THE CALL: ctypes reads `_as_parameter_` while it converts
    the arguments, which happens inside `_GuardedCall`'s device guard, so a call on a cuda:1 tensor made while cuda:0 is current gets cuda:1's
    stream (the reference used the legacy default stream of the current device)."""
    __slots__ = ()

    @property
    def _as_parameter_(self):
        dev = getattr(_ACTIVE, "device", None)
        # the raw handle of torch's current stream on that device (what torch.cuda.current_stream(dev).cuda_stream returns, without building the
        # Stream object: 7 us -> 0.5 us per launch, and a training step makes a dozen)
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if dev is None or dev.index is None else dev.index))


_STREAM = _CurrentStream()


def stream():
    return _STREAM


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libngp_hip: expected a tensor on the GPU (there is no CPU path)")


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.float16:
        return F16
    raise RuntimeError(f"libngp_hip: unsupported dtype {dt} (float32 / float16 only)")


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def camera_args(pose, intrinsics):
    """host arguments of ngp_get_rays / ngp_render_frame_camera: pose [4,4] or [3,4] (tensor, array or nested list) as 16
    row-major floats (last row ignored by the library), intrinsics (fx, fy, cx, cy) as 4 floats"""
    if isinstance(pose, torch.Tensor):
        pose = pose.detach().cpu().tolist()
    rows = [[float(v) for v in r] for r in pose]
    if len(rows) not in (3, 4) or any(len(r) != 4 for r in rows):
        raise ValueError("pose must be [4,4] or [3,4]")
    flat = [v for r in rows[:3] for v in r] + [0.0, 0.0, 0.0, 1.0]
    return (ctypes.c_float * 16)(*flat), (ctypes.c_float * 4)(*[float(v) for v in intrinsics])
