"""Synthetic stand-ins for the reference's assets (none of which exist offline: README.md:83).

Scene "S-ring" (BASELINE config 2): a Stonehenge-like ring of axis-aligned boxes in a bound-2 volume
-- 12 pillars (0.12 x 0.12 x 0.45) on a circle of radius 0.65, two lintels, a ground slab -- with analytic density
60 inside, 0 outside.  `make_model` builds a field of the reference's network_ff architecture whose hash table and
MLP weights are hand-set so that the field actually represents the scene (density ~60 inside the boxes, ~0.017
outside), so that rays terminate the way they do in a trained model; everything not needed for that is seeded
random so the gathers and GEMMs are full-entropy.  numpy only; the product and the oracle both consume its
outputs, neither is imported here.
"""
import math

import numpy as np

BOUND = 2.0
GRID = 128
SIGMA_IN = 60.0
P1, P2 = np.uint32(2654435761), np.uint32(805459861)


def church_boxes():
    """Scene "shell" of BASELINE config 5 ("the same generator with a larger shell scene, bound 2"): a hollow
    church -- nave walls, a stepped roof, a tower, a floor and two rows of interior columns -- filling most of the inner cascade
    and reaching into the outer one, so that rays cross several thin occupied shells and much more of the grid is occupied than
    in the ring scene (more samples per ray, more table rows touched)."""
    t = 0.07                                                   # wall thickness
    L, Wd, Hh = 1.5, 0.8, 0.9                                  # half length (x), half width (y), wall height (z)
    boxes = [((-L, -Wd, -0.05), (L, Wd, 0.0)),                 # floor
             ((-L, -Wd, 0.0), (L, -Wd + t, Hh)), ((-L, Wd - t, 0.0), (L, Wd, Hh)),          # long walls
             ((-L, -Wd, 0.0), (-L + t, Wd, Hh)), ((L - t, -Wd, 0.0), (L, Wd, Hh))]          # end walls
    for k in range(6):                                         # stepped gable roof: slabs narrowing towards the ridge
        w = Wd * (1 - k / 6)
        boxes.append(((-L, -w, Hh + 0.09 * k), (L, w, Hh + 0.09 * k + t)))
    boxes.append(((L - 0.1, -0.3, 0.0), (L + 0.5, 0.3, 1.85)))                               # tower (solid shell approximated by a block wall set)
    boxes.append(((L + 0.05, -0.2, 1.85), (L + 0.35, 0.2, 1.97)))
    for k in range(7):                                         # two rows of columns
        x = -1.2 + 0.4 * k
        for y in (-0.4, 0.4):
            boxes.append(((x - 0.04, y - 0.04, 0.0), (x + 0.04, y + 0.04, Hh)))
    return [(np.array(lo, np.float64), np.array(hi, np.float64)) for lo, hi in boxes]


def scene_boxes(scene="ring"):
    """[(lo[3], hi[3])] in world units.  scene: "ring" (BASELINE config 2-4) or "church" (config 5)."""
    if scene == "church":
        return church_boxes()
    assert scene == "ring", scene
    boxes = []
    for k in range(12):
        a = 2 * math.pi * k / 12
        cx, cy = 0.65 * math.cos(a), 0.65 * math.sin(a)
        boxes.append(((cx - 0.06, cy - 0.06, 0.0), (cx + 0.06, cy + 0.06, 0.45)))
    for k in (0, 6):                                         # lintels across pillars k and k+1
        a0, a1 = 2 * math.pi * k / 12, 2 * math.pi * (k + 1) / 12
        xs = sorted([0.65 * math.cos(a0), 0.65 * math.cos(a1)])
        ys = sorted([0.65 * math.sin(a0), 0.65 * math.sin(a1)])
        boxes.append(((xs[0] - 0.06, ys[0] - 0.06, 0.45), (xs[1] + 0.06, ys[1] + 0.06, 0.53)))
    boxes.append(((-1.0, -1.0, -0.05), (1.0, 1.0, 0.0)))      # ground slab
    return [(np.array(lo, np.float64), np.array(hi, np.float64)) for lo, hi in boxes]


def inside(points, boxes=None):
    """bool [N]: point inside any box."""
    boxes = boxes or scene_boxes()
    p = np.asarray(points, np.float64)
    m = np.zeros(p.shape[0], bool)
    for lo, hi in boxes:
        m |= np.all((p >= lo) & (p <= hi), axis=1)
    return m


def _spread3(v):
    v = np.asarray(v, np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3(x, y, z):
    return _spread3(x) | (_spread3(y) << np.uint32(1)) | (_spread3(z) << np.uint32(2))


def cascade_count(bound):
    return 1 + math.ceil(math.log2(bound))                   # nerf/renderer.py:73


def density_grid(bound=BOUND, H=GRID, boxes=None, scene="ring"):
    """Analytic density grid [cascade, H^3] float32 in Morton order: SIGMA_IN in every cell a box touches."""
    boxes = boxes or scene_boxes(scene)
    cas = cascade_count(bound)
    grid = np.zeros((cas, H ** 3), np.float32)
    for c in range(cas):
        b = min(2.0 ** c, bound)
        cell = 2 * b / H
        for lo, hi in boxes:
            i0 = np.clip(np.floor((lo + b) / cell).astype(int), 0, H - 1)
            i1 = np.clip(np.ceil((hi + b) / cell).astype(int) - 1, 0, H - 1)
            if np.any(hi < -b) or np.any(lo > b):
                continue
            xs, ys, zs = (np.arange(i0[d], i1[d] + 1, dtype=np.uint32) for d in range(3))
            X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
            grid[c, morton3(X.ravel(), Y.ravel(), Z.ravel())] = SIGMA_IN
    return grid


def bitfield_from_grid(grid, density_thresh=10.0):
    """update_extra_state's rule (nerf/renderer.py:526-531): thresh = min(mean(clamp(grid,0)), density_thresh);
    bit = grid > thresh, 8 cells per byte, LSB first."""
    mean_density = float(np.mean(np.clip(grid, 0, None)))
    thresh = min(mean_density, density_thresh)
    return np.packbits(grid.reshape(-1) > np.float32(thresh), bitorder="little"), thresh


def grid_offsets(bound=BOUND, num_levels=16, base_resolution=16, log2_hashmap_size=19):
    """GridEncoder level table for desired_resolution = 2048*bound (gridencoder/grid.py:97-123, nerf/network.py:31)."""
    pls = np.exp2(np.log2(2048 * bound / base_resolution) / (num_levels - 1))
    offsets, off = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * pls ** i))
        n = min(2 ** log2_hashmap_size, (res + 1) ** 3)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(off)
        off += n
    offsets.append(off)
    return np.array(offsets, np.int32), float(pls)


OCC_LEVEL = 6          # hashed level that carries the occupancy feature (cell 4/148 = 0.027 at bound 2)
ONE_LEVEL = 0          # dense level whose second feature is the constant 1


def make_model(seed=0, bound=BOUND, scene="ring"):
    """Returns dict(embeddings f32 [sO,2], offsets, per_level_scale, sigma_weights f32 [7168], color_weights f32 [11264])
    in the layouts of GridEncoder.embeddings / FFMLP.weights (gridencoder/grid.py:129, ffmlp/ffmlp.py:121-122)."""
    rng = np.random.default_rng(seed)
    offsets, pls = grid_offsets(bound)
    emb = (rng.uniform(-1, 1, size=(int(offsets[-1]), 2)) * 2.0 ** -4).astype(np.float32)

    # constant-one feature on the coarsest (dense) level: trilinear weights sum to 1, so it interpolates to 1 everywhere
    emb[offsets[ONE_LEVEL]:offsets[ONE_LEVEL + 1], 1] = 1.0

    # occupancy feature on a hashed level: 1 at every grid vertex inside a box (gridencoder.cu:125-138 vertex positions)
    S = np.float32(np.log2(pls))
    scale = np.float32(np.exp2(np.float32(OCC_LEVEL) * S) * 16 - 1)
    res = int(np.ceil(scale)) + 1
    size = int(offsets[OCC_LEVEL + 1] - offsets[OCC_LEVEL])
    emb[offsets[OCC_LEVEL]:offsets[OCC_LEVEL + 1], 0] = 0.0
    ii = np.arange(res + 1, dtype=np.float64)
    world = ((ii - 0.5) / float(scale)) * 2 * bound - bound          # vertex i sits at x01 = (i - 0.5) / scale
    for lo, hi in scene_boxes(scene):
        sel = [np.flatnonzero((world >= lo[d]) & (world <= hi[d])).astype(np.uint32) for d in range(3)]
        if min(len(s) for s in sel) == 0:
            continue
        X, Y, Z = np.meshgrid(*sel, indexing="ij")
        h = X.ravel() ^ (Y.ravel() * P1) ^ (Z.ravel() * P2)            # fast_hash (gridencoder.cu:35-51), uint32 wrap
        emb[offsets[OCC_LEVEL] + (h % np.uint32(size)).astype(np.int64), 0] = 1.0

    f_occ, f_one = 2 * OCC_LEVEL, 2 * ONE_LEVEL + 1
    # density net FFMLP(32 -> 64 -> 64 -> 16): rows 0/1 carry occupancy / one straight through, row 0 of the output
    # layer forms the logit 8.2*occ - 4.1 (sigma = e^4.1 ~ 60 inside, e^-4.1 ~ 0.017 outside); the rest is random.
    w1 = rng.uniform(-0.3, 0.3, size=(64, 32)); w2 = rng.uniform(-0.3, 0.3, size=(64, 64)); w3 = rng.uniform(-0.3, 0.3, size=(16, 64))
    w1[0] = 0; w1[1] = 0; w1[0, f_occ] = 1.0; w1[1, f_one] = 1.0
    w2[0] = 0; w2[1] = 0; w2[0, 0] = 1.0; w2[1, 1] = 1.0
    w3[0] = 0; w3[0, 0] = 8.2; w3[0, 1] = -4.1
    sigma_w = np.concatenate([w1.ravel(), w2.ravel(), w3.ravel()]).astype(np.float32)
    # colour net FFMLP(32 -> 64 -> 64 -> 64 -> 16): the reference's own init, U(+-sqrt(3/64)) (ffmlp/ffmlp.py:141-144)
    std = math.sqrt(3 / 64)
    color_w = rng.uniform(-std, std, size=64 * (32 + 128 + 16)).astype(np.float32)
    return dict(embeddings=emb, offsets=offsets, per_level_scale=pls, sigma_weights=sigma_w, color_weights=color_w, bound=bound)


def nav_weights(seed=0):
    """The DEFAULT field's five bias-free Linear matrices (nerf/network.py:33-68: 32-64-16 | 31-64-64-3) hand-set like `make_model`'s FFMLP so that,
    over `make_model(seed)["embeddings"]`, the field the nav loop queries in float32 (BASELINE config 4) is the same scene: sigma = exp(8.2 occ - 4.1),
    occ the trilinearly interpolated occupancy feature -- smooth across box faces, so d sigma / d x is informative for the planner.
    Returns (sigma_layers [64,32],[16,64]; colour_layers [64,31],[64,64],[3,64]) float32."""
    rng = np.random.default_rng(seed + 1000)
    f_occ, f_one = 2 * OCC_LEVEL, 2 * ONE_LEVEL + 1
    w1 = rng.uniform(-0.3, 0.3, size=(64, 32)); w2 = rng.uniform(-0.3, 0.3, size=(16, 64))
    w1[0] = 0; w1[1] = 0; w1[0, f_occ] = 1.0; w1[1, f_one] = 1.0
    w2[0] = 0; w2[0, 0] = 8.2; w2[0, 1] = -4.1
    std = math.sqrt(3 / 64)
    c = [rng.uniform(-std, std, size=shape) for shape in ((64, 31), (64, 64), (3, 64))]
    c[2] = c[2] * 6.0                                          # spread the colours over (0, 1) instead of hovering around 0.5
    return [w1.astype(np.float32), w2.astype(np.float32)], [m.astype(np.float32) for m in c]


def intrinsics(H, W, fovx=0.6911):
    """(fx, fy, cx, cy) of a square-pixel pinhole; fovx is nerf_synthetic's camera_angle_x."""
    f = 0.5 * W / math.tan(0.5 * fovx)
    return np.array([f, f, W / 2, H / 2], np.float32)


def scene_orbit(scene="ring"):
    """(radius, height) of the test-pose orbit: the ring scene is viewed from 1.6; the church is larger, 2.3 keeps the camera outside it"""
    return (2.3, 0.9) if scene == "church" else (1.6, 0.6)


def orbit_pose(k, n=8, radius=1.6, height=0.6):
    """Camera-to-world [4,4] float32: on a circle of `radius` at z = `height`, looking at the origin, z up.
    Camera axes follow get_rays (nerf/utils.py:103-108): +z forward, +x right, +y down."""
    a = 2 * math.pi * k / n
    eye = np.array([radius * math.cos(a), radius * math.sin(a), height])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0])); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, eye
    return pose.astype(np.float32)


def get_rays(pose, intr, H, W):
    """get_rays for a full image (nerf/utils.py:53-116, N = -1 branch), float32 numpy: rays_o, rays_d [H*W, 3]."""
    fx, fy, cx, cy = (np.float32(v) for v in intr)
    i = (np.arange(W, dtype=np.float32) + np.float32(0.5))[None, :].repeat(H, 0).reshape(-1)
    j = (np.arange(H, dtype=np.float32) + np.float32(0.5))[:, None].repeat(W, 1).reshape(-1)
    xs = (i - cx) / fx
    ys = (j - cy) / fy
    d = np.stack([xs, ys, np.ones_like(xs)], -1)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    rays_d = (d @ pose[:3, :3].T).astype(np.float32)
    rays_o = np.broadcast_to(pose[:3, 3], rays_d.shape).astype(np.float32).copy()
    return rays_o, rays_d
