"""GPU: the fused kernels on level tables other than the reference default.  The encoder picks straight-line code per
iteration from the level table (render_fused.hip rf_classify: dense / hashed / select / generic) and the frame kernel has a
compile-time instance for the default pattern only, so these exercise the run-time classes:
  log2_hashmap_size 15  -> levels 0..2 dense, 3.. hashed: iteration 0 is a dense/hashed mix ("select")
  log2_hashmap_size 22  -> levels 0..7 dense: iterations 0 and 1 all dense
  hand-made offsets     -> hashed levels whose row count is not a power of two: the generic class (index % size)
Reference for all: the oracle's encoder (index % hashmap_size for every level, gridencoder.cu:54-72) + MLP."""
import numpy as np
import pytest
import torch

from oracle import render_oracle as R

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _model(W, offsets, pls, seed=0):
    rng = np.random.default_rng(seed)
    base = W.make_model(0)
    emb = (rng.uniform(-1, 1, size=(int(offsets[-1]), 2)) * 0.25).astype(np.float32)
    return dict(base, embeddings=emb, offsets=np.asarray(offsets, np.int32), per_level_scale=pls)


def _field(dev, W, model, log2):
    from gridencoder import GridEncoder
    from ngp.field import NGPFieldFF
    field = NGPFieldFF(bound=W.BOUND)
    field.encoder = GridEncoder(desired_resolution=2048 * W.BOUND, log2_hashmap_size=log2)
    field = field.to(dev)
    if not np.array_equal(field.encoder.offsets.cpu().numpy(), model["offsets"]):      # hand-made table: install it
        field.encoder.offsets = t(model["offsets"], dev)
        field.encoder.embeddings = torch.nn.Parameter(torch.empty(int(model["offsets"][-1]), 2, device=dev))
    return field.load_arrays(model)


def _tables(W):
    out = {}
    for log2 in (15, 22):
        offsets, pls = W.grid_offsets(W.BOUND, log2_hashmap_size=log2)
        out[f"log2_{log2}"] = (offsets, pls, log2)
    offsets, pls = W.grid_offsets(W.BOUND, log2_hashmap_size=15)
    sizes = np.diff(offsets).astype(np.int64)
    sizes[sizes == 2 ** 15] -= 8 * np.arange(1, (sizes == 2 ** 15).sum() + 1)         # 32760, 32752, ...: not powers of two
    out["odd_sizes"] = (np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), pls, 15)
    return out


@pytest.mark.parametrize("name", ["log2_15", "log2_22", "odd_sizes"])
def test_field_forward_on_other_tables(name, dev):
    from ngp import workload as W
    offsets, pls, log2 = _tables(W)[name]
    model = _model(W, offsets, pls)
    field = _field(dev, W, model, log2)
    rng = np.random.default_rng(1)
    x = rng.uniform(-2, 2, size=(6000, 3)).astype(np.float32)
    x[:8] = np.array([[2, 2, 2], [-2, -2, -2], [2, -2, 0], [0, 0, 0], [1, 1, 1], [-1, 2, -2], [2, 0, 0], [0, 0, -2]], np.float32)
    d = rng.normal(size=(6000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sig_ref, rgb_ref = R.field_forward(model, x, d, 1.0)
    sig, rgb = field.forward_fused(t(x, dev), t(d, dev))
    sig, rgb = sig.cpu().numpy(), rgb.cpu().numpy()
    assert np.max(np.abs(sig - sig_ref) / np.maximum(sig_ref, 1e-6)) < 5e-3 and (sig == sig_ref).mean() > 0.95
    assert np.max(np.abs(rgb - rgb_ref)) < 4e-3 and (rgb == rgb_ref).mean() > 0.9
    # and the drop-in encoder on the same table: bit-exact features
    from oracle import ngp_oracle as O
    x01 = ((x + np.float32(2)) / np.float32(4)).astype(np.float32)
    feats_ref, _ = O.grid_encode_forward(x01, model["embeddings"].astype(np.float16), model["offsets"], pls, 16, False, 0, False)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        feats = field.encoder(t(x, dev), bound=W.BOUND)
    assert np.array_equal(feats.cpu().numpy().view(np.uint16), np.ascontiguousarray(feats_ref.transpose(1, 0, 2).reshape(-1, 32)).view(np.uint16))


@pytest.mark.parametrize("name", ["log2_15", "odd_sizes"])
def test_render_fused_on_other_tables(name, dev):
    """whole frames through the run-time-class instance of the frame kernel"""
    from ngp import workload as W
    from ngp.render import NGPRenderer
    offsets, pls, log2 = _tables(W)[name]
    model = _model(W, offsets, pls)
    field = _field(dev, W, model, log2)
    ren = NGPRenderer(field, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    grid = W.density_grid()
    bf, _ = W.bitfield_from_grid(grid)
    ren.load_density_grid(grid)
    o, d = W.get_rays(W.orbit_pose(2), W.intrinsics(40, 40), 40, 40)
    ref = R.render_single_march(lambda x, dd: R.field_forward(model, x, dd, 1.0), o, d, bf, W.BOUND, 2)
    out = ren.render_fused(t(o, dev)[None], t(d, dev)[None], bg_color=1, image_width=40)
    stats = out["stats"].cpu().numpy()
    img = out["image"][0].cpu().numpy()
    assert abs(int(stats[0]) - ref["samples"]) <= max(8, 2e-3 * ref["samples"]) and stats[1] == 0
    assert np.max(np.abs(img - ref["image"])) < 8e-3 and R.psnr(img, ref["image"]) > 50
