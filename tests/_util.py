"""Shared synthetic inputs for the tests (seeded, small enough for the CPU oracle)."""
import numpy as np


def blob_bitfield(oracle, cascade=2, H=128, seed=0, n_blobs=40, bound=2.0):
    """Occupancy bitfield [cascade*H^3/8] from random spheres; Morton-ordered cells like the reference's density grid."""
    rng = np.random.default_rng(seed)
    idx = np.arange(H ** 3, dtype=np.int32)
    coords = oracle.morton3D_invert(idx).astype(np.float32)           # cell (x,y,z) of each Morton index
    grid = np.zeros((cascade, H ** 3), np.float32)
    for cas in range(cascade):
        b = min(2.0 ** cas, bound)
        centres = ((coords + 0.5) / H * 2 - 1) * b
        c = rng.uniform(-0.8 * b, 0.8 * b, size=(n_blobs, 3)).astype(np.float32)
        r = rng.uniform(0.05 * b, 0.25 * b, size=n_blobs).astype(np.float32)
        for k in range(n_blobs):
            d2 = ((centres - c[k]) ** 2).sum(1)
            grid[cas, d2 < r[k] ** 2] = 1.0
    return oracle.packbits(grid, 0.5), grid


def camera_rays(n_side=32, radius=3.0, seed=0, jitter=True):
    """Pinhole rays from a camera on an orbit looking at the origin; a few degenerate directions appended."""
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, 2 * np.pi)
    eye = np.array([radius * np.cos(th), radius * np.sin(th), 0.7 * radius * 0.3], np.float32)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0, 0, 1], np.float32)
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    u, v = np.meshgrid(np.linspace(-0.6, 0.6, n_side), np.linspace(-0.6, 0.6, n_side))
    d = fwd[None] + u.reshape(-1, 1) * right[None] + v.reshape(-1, 1) * up[None]
    if jitter:
        d += rng.normal(scale=1e-3, size=d.shape)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = np.broadcast_to(eye, d.shape).astype(np.float32).copy()
    # axis-parallel rays (a zero component in d: 1/d = inf) and a ray that misses the box
    extra_o = np.array([[-3, 0.1, 0.2], [0.3, -3, 0.1], [0.2, 0.1, 3], [5, 5, 5]], np.float32)
    extra_d = np.array([[1, 0, 0], [0, 1, 0], [0, 0, -1], [1, 0, 0]], np.float32)
    return np.concatenate([o, extra_o]), np.concatenate([d, extra_d])


def oracle_field(field, dtype=None):
    """oracle.callers_oracle.DefaultField holding the parameters of a product field module (ngp.field.NGPField or NGPFieldFF):
    the CPU checker the GPU parity tests compare against."""
    import torch
    from oracle import callers_oracle as CO
    dtype = dtype or torch.float32
    emb = field.encoder.embeddings.detach().float().cpu().numpy()
    offsets = field.encoder.offsets.cpu().numpy()
    pls = float(field.encoder.per_level_scale)
    if hasattr(field.sigma_net, "weights"):                                   # FFMLP: flat [64,in] + k [64,64] + [16,64] (ffmlp/ffmlp.py:121-122)
        def split(net):
            w = net.weights.detach().float().cpu().numpy()
            shapes = [(net.hidden_dim, net.input_dim)] + [(net.hidden_dim, net.hidden_dim)] * (net.num_layers - 1) + [(net.padded_output_dim, net.hidden_dim)]
            out, off = [], 0
            for r, c in shapes:
                out.append(w[off:off + r * c].reshape(r, c)); off += r * c
            assert off == w.size
            return out
        return CO.DefaultField(emb, offsets, pls, split(field.sigma_net), split(field.color_net), field.bound, dtype=dtype, ff_layout=True)
    sw = [l.weight.detach().float().cpu().numpy() for l in field.sigma_net]
    cw = [l.weight.detach().float().cpu().numpy() for l in field.color_net]
    return CO.DefaultField(emb, offsets, pls, sw, cw, field.bound, dtype=dtype)


def ff_grads_as_matrices(net):
    """.grad of an FFMLP's flat weights as the list of per-layer matrices (same split as oracle_field)."""
    g = net.weights.grad.detach().float().cpu().numpy()
    shapes = [(net.hidden_dim, net.input_dim)] + [(net.hidden_dim, net.hidden_dim)] * (net.num_layers - 1) + [(net.padded_output_dim, net.hidden_dim)]
    out, off = [], 0
    for r, c in shapes:
        out.append(g[off:off + r * c].reshape(r, c)); off += r * c
    return out


def ff_model_matrices(model):
    """the flat FFMLP weights of a workload.make_model() dict as per-layer matrices (ffmlp/ffmlp.py:121-122: [64,32] + k [64,64] + [16,64]):
    (sigma net [64,32],[64,64],[16,64]; colour net [64,32],[64,64],[64,64],[16,64])"""
    def split(flat, n_hidden):
        shapes = [(64, 32)] + [(64, 64)] * (n_hidden - 1) + [(16, 64)]
        out, off = [], 0
        for r, c in shapes:
            out.append(np.asarray(flat[off:off + r * c], np.float32).reshape(r, c)); off += r * c
        assert off == flat.size
        return out
    return split(model["sigma_weights"], 2), split(model["color_weights"], 3)


def linear_field_from_model(model, dev):
    """ngp.field.NGPField (nn.Linear layers, float32) holding the numbers of a workload.make_model() dict: the FFMLP shapes 32-64-64-16 and
    32-64-64-64-16 as Linear stacks (the colour net's zero input column and its 13 unused output rows dropped, nerf/network_ff.py:67-74)."""
    import torch
    from ngp.field import NGPField
    sw, cw = ff_model_matrices(model)
    field = NGPField(bound=model["bound"], num_layers=3, num_layers_color=4).to(dev)
    cw = [cw[0][:, :31]] + cw[1:-1] + [cw[-1][:3]]
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.from_numpy(model["embeddings"]))
        for layer, w in zip(list(field.sigma_net) + list(field.color_net), sw + cw):
            assert tuple(layer.weight.shape) == w.shape, (layer.weight.shape, w.shape)
            layer.weight.copy_(torch.from_numpy(np.ascontiguousarray(w)))
    return field
