"""ctypes/numpy binding of oracle/ngp_oracle.c.

TEST INFRASTRUCTURE ONLY (see the header of ngp_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
package under nerf-navigation_amd/.

Every function takes and returns numpy arrays and mirrors the argument order of
the reference's native entry points (raymarching/src/raymarching.h:7-18,
gridencoder/src/gridencoder.h:12-13, ffmlp/src/ffmlp.h:8-14).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ngp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_threads(n):
    """team size of the C loops' OpenMP regions; returns the previous maximum"""
    return int(lib().o_set_threads(ctypes.c_int(int(n))))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


c_u32, c_f32, c_i32, c_u64 = ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_uint64

# ----------------------------------------------------------------------------
# scalars / helpers
# ----------------------------------------------------------------------------


def expf(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().o_expf_array(_p(x), _p(y), c_u32(x.size))
    return y


def sinf(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().o_sinf_array(_p(x), _p(y), c_u32(x.size))
    return y


def freq_encode_forward(inputs, degree):
    """freqencoder.cu:27-60.  inputs [B,D] f32 -> [B, D + 2 D degree] f32."""
    x = _f32(inputs)
    B, D = x.shape
    C = D + 2 * D * degree
    out = np.empty((B, C), np.float32)
    lib().o_freq_encode_forward(_p(x), c_u32(B), c_u32(D), c_u32(degree), c_u32(C), _p(out))
    return out


def freq_encode_backward(grad, outputs, input_dim, degree):
    """freqencoder.cu:65-92.  grad, outputs [B,C] -> grad_inputs [B,D]."""
    g, o = _f32(grad), _f32(outputs)
    B, C = g.shape
    gi = np.empty((B, input_dim), np.float32)
    lib().o_freq_encode_backward(_p(g), _p(o), c_u32(B), c_u32(input_dim), c_u32(degree), c_u32(C), _p(gi))
    return gi


def f32_to_f16_bits(x):
    x = _f32(x)
    y = np.empty(x.shape, dtype=np.uint16)
    lib().o_f32_to_f16(_p(x), _p(y), c_u64(x.size))
    return y


def f16_bits_to_f32(h):
    h = np.ascontiguousarray(h, dtype=np.uint16)
    y = np.empty(h.shape, dtype=np.float32)
    lib().o_f16_to_f32(_p(h), _p(y), c_u64(h.size))
    return y


def pcg32_kat(seed, seq, advances):
    adv = np.ascontiguousarray(advances, dtype=np.uint64)
    u = np.empty(adv.size, dtype=np.uint32)
    f = np.empty(adv.size, dtype=np.float32)
    lib().o_pcg32_kat(c_u64(seed), c_u64(seq), _p(adv), c_u32(adv.size), _p(u), _p(f))
    return u, f


def pcg32_stream(seed, seq, n):
    u = np.empty(n, dtype=np.uint32)
    lib().o_pcg32_stream(c_u64(seed), c_u64(seq), c_u32(n), _p(u))
    return u


# ----------------------------------------------------------------------------
# raymarching
# ----------------------------------------------------------------------------


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().o_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), c_u32(N), c_f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().o_sph_from_ray(_p(rays_o), _p(rays_d), c_f32(radius), c_u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords).reshape(-1, 3)
    out = np.empty(coords.shape[0], np.int32)
    lib().o_morton3D(_p(coords), c_u32(coords.shape[0]), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices).reshape(-1)
    out = np.empty((indices.shape[0], 3), np.int32)
    lib().o_morton3D_invert(_p(indices), c_u32(indices.shape[0]), _p(out))
    return out


def packbits(grid, thresh, bitfield=None):
    grid = _f32(grid)
    N = grid.size // 8
    if bitfield is None:
        bitfield = np.empty(N, np.uint8)
    lib().o_packbits(_p(grid), c_u32(N), c_f32(thresh), _p(bitfield))
    return bitfield


def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, counter=None, mean_count=-1,
                     perturb=False, align=-1, force_all_rays=False, dt_gamma=0.0, max_steps=1024):
    """raymarching/raymarching.py:161-230 including the M / align / slicing logic."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    N = rays_o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs, dirs = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if counter is None:
        counter = np.zeros(2, np.int32)
    lib().o_march_rays_train(_p(rays_o), _p(rays_d), _p(bitfield), c_f32(bound), c_f32(dt_gamma), c_u32(max_steps),
                             c_u32(N), c_u32(C), c_u32(H), c_u32(M), _p(_f32(nears)), _p(_f32(fars)),
                             _p(xyzs), _p(dirs), _p(deltas), _p(rays), _p(counter), c_u32(int(perturb)))
    if force_all_rays or mean_count <= 0:
        m = int(counter[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


def composite_rays_train_forward(sigmas, rgbs, deltas, rays):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().o_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), c_u32(M), c_u32(N),
                                         _p(ws), _p(depth), _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().o_composite_rays_train_backward(_p(_f32(grad_ws)), _p(_f32(grad_image)), _p(sigmas), _p(rgbs), _p(deltas),
                                          _p(rays), _p(_f32(weights_sum)), _p(_f32(image)), c_u32(M), c_u32(N),
                                          _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars,
               align=-1, perturb=False, dt_gamma=0.0, max_steps=1024):
    """raymarching/raymarching.py:292-337 (M padded past the next multiple of align)."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    # the C restatement indexes like the reference's kernel, unchecked: a ray index past the arrays reads garbage and can march for ever
    alive = np.asarray(rays_alive).reshape(-1)[:n_alive]
    if alive.size < n_alive or (n_alive and (int(alive.max()) >= rays_o.shape[0] or int(alive.max()) >= np.asarray(rays_t).size)):
        raise ValueError("march_rays: rays_alive names rays the inputs do not hold")
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    lib().o_march_rays(c_u32(n_alive), c_u32(n_step), _p(_i32(rays_alive)), _p(_f32(rays_t)), _p(rays_o), _p(rays_d),
                       c_f32(bound), c_f32(dt_gamma), c_u32(max_steps), c_u32(C), c_u32(H),
                       _p(np.ascontiguousarray(bitfield, dtype=np.uint8)), _p(_f32(nears)), _p(_f32(fars)),
                       _p(xyzs), _p(dirs), _p(deltas), c_u32(int(perturb)))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """In place on rays_alive, rays_t, weights_sum, depth, image (all must be the right dtype, contiguous)."""
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32), (image, np.float32)):
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
    lib().o_composite_rays(c_u32(n_alive), c_u32(n_step), _p(rays_alive), _p(rays_t), _p(_f32(sigmas)), _p(_f32(rgbs)),
                           _p(_f32(deltas)), _p(weights_sum), _p(depth), _p(image))


# ----------------------------------------------------------------------------
# grid encoder
# ----------------------------------------------------------------------------


def grid_level_table(L, S, H):
    scale, reso = np.empty(L, np.float32), np.empty(L, np.uint32)
    lib().o_grid_level_table(c_u32(L), c_f32(S), c_u32(H), _p(scale), _p(reso))
    return scale, reso


def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, align_corners=False):
    """Level table of GridEncoder.__init__ (gridencoder/grid.py:97-123). Returns (offsets int32[L+1], per_level_scale)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params = int(np.ceil(params / 8) * 8)
        offsets.append(offset)
        offset += params
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                        gridtype=0, align_corners=False):
    """Returns (outputs [L,B,C], dy_dx [B, L*D*C] or None). embeddings float32 or float16 decides the arithmetic."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    offsets = _i32(offsets)
    L = offsets.shape[0] - 1
    C = embeddings.shape[1]
    is_half = embeddings.dtype == np.float16
    emb = np.ascontiguousarray(embeddings)
    S = np.float32(np.log2(per_level_scale))
    outputs = np.empty((L, B, C), emb.dtype)
    dy_dx = np.empty((B, L * D * C), emb.dtype) if calc_grad_inputs else None
    lib().o_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(outputs), c_u32(B), c_u32(D), c_u32(C), c_u32(L),
                                c_f32(S), c_u32(base_resolution), c_i32(int(calc_grad_inputs)), _p(dy_dx),
                                c_u32(gridtype), c_i32(int(align_corners)), c_i32(int(is_half)))
    return outputs, dy_dx


def grid_encode_backward(grad, inputs, embeddings, offsets, per_level_scale, base_resolution, dy_dx=None,
                         gridtype=0, align_corners=False):
    """grad [L,B,C] in the table dtype. Returns (grad_embeddings float64 [sO,C], grad_inputs [B,D] or None)."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    offsets = _i32(offsets)
    L = offsets.shape[0] - 1
    C = embeddings.shape[1]
    is_half = embeddings.dtype == np.float16
    grad = np.ascontiguousarray(grad, dtype=embeddings.dtype)
    S = np.float32(np.log2(per_level_scale))
    ge = np.zeros(embeddings.shape, np.float64)
    calc = dy_dx is not None
    gi = np.zeros((B, D), embeddings.dtype) if calc else None
    lib().o_grid_encode_backward(_p(grad), _p(inputs), _p(np.ascontiguousarray(embeddings)), _p(offsets), _p(ge),
                                 c_u32(B), c_u32(D), c_u32(C), c_u32(L), c_f32(S), c_u32(base_resolution),
                                 c_i32(int(calc)), _p(np.ascontiguousarray(dy_dx) if calc else None), _p(gi),
                                 c_u32(gridtype), c_i32(int(align_corners)), c_i32(int(is_half)))
    return ge, gi


# ----------------------------------------------------------------------------
# fully fused MLP
# ----------------------------------------------------------------------------


def ffmlp_num_params(input_dim, output_dim_padded, hidden_dim, num_layers):
    return hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + output_dim_padded)


def ffmlp_forward(inputs, weights, input_dim, output_dim, hidden_dim, num_layers, save=False, activation=0):
    """inputs [B,in] float16, weights flat float16. Returns (outputs [B,out] f16, forward_buffer or None).
    activation: the hidden activation code of ffmlp.py:89-96 (0 relu, 1 exponential, 2 sine, 3 sigmoid, 4 squareplus, 5 softplus, 6 none)."""
    x = np.ascontiguousarray(inputs, dtype=np.float16)
    w = np.ascontiguousarray(weights, dtype=np.float16)
    B = x.shape[0]
    out = np.empty((B, output_dim), np.float16)
    fb = np.empty((num_layers, B, hidden_dim), np.float16) if save else None
    lib().o_ffmlp_forward_act(_p(x), _p(w), c_u32(B), c_u32(input_dim), c_u32(output_dim), c_u32(hidden_dim),
                              c_u32(num_layers), c_u32(activation), _p(fb), _p(out))
    return out, fb


def ffmlp_backward(grad, inputs, weights, forward_buffer, input_dim, output_dim, hidden_dim, num_layers,
                   calc_grad_inputs=False, activation=0):
    """Returns (grad_weights float32 flat, grad_inputs float32 [B,in] or None, backward_buffer f16)."""
    g = np.ascontiguousarray(grad, dtype=np.float16)
    x = np.ascontiguousarray(inputs, dtype=np.float16)
    w = np.ascontiguousarray(weights, dtype=np.float16)
    fb = np.ascontiguousarray(forward_buffer, dtype=np.float16)
    B = x.shape[0]
    bb = np.zeros((num_layers, B, hidden_dim), np.float16)
    gi = np.zeros((B, input_dim), np.float32) if calc_grad_inputs else None
    gw = np.zeros(w.shape, np.float32)
    lib().o_ffmlp_backward_act(_p(g), _p(x), _p(w), _p(fb), c_u32(B), c_u32(input_dim), c_u32(output_dim), c_u32(hidden_dim),
                               c_u32(num_layers), c_u32(activation), c_i32(int(calc_grad_inputs)), _p(bb), _p(gi), _p(gw))
    return gw, gi, bb
