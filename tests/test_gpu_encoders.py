"""GPU parity for the grid and SH encoders through the drop-in packages against the CPU oracle.
Grid forward / dy_dx / input gradient: BIT-EXACT in float32 and in float16 (the reference's scalar_t arithmetic is
reproduced operation for operation).  Table gradient: atomics => order-dependent sums, compared with a tolerance.
SH: float32 recurrences vs float64 polynomials, absolute tolerance stated in the test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def same_bits(got, want, what):
    g = got.detach().cpu().numpy()
    assert g.dtype == want.dtype and g.shape == want.shape, f"{what}: {g.dtype}{g.shape} vs {want.dtype}{want.shape}"
    u = np.uint32 if g.dtype == np.float32 else np.uint16
    bad = np.flatnonzero(np.ascontiguousarray(g).view(u).reshape(-1) != np.ascontiguousarray(want).view(u).reshape(-1))
    assert bad.size == 0, f"{what}: {bad.size}/{g.size} differ, first {bad[:4]}: got {g.reshape(-1)[bad[:4]]} want {want.reshape(-1)[bad[:4]]}"


def make_points(B, D, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, size=(B, D)).astype(np.float32)
    x[0] = 0.0; x[1] = 1.0                      # the closed ends of the valid range
    x[2, 0] = -1e-3; x[3, 1] = 1.001            # out of range => zeros (gridencoder.cu:99-123)
    x[4] = 0.5
    return x


CASES = [  # (D, C, L, base, log2_hashmap, desired_resolution, gridtype, align_corners)
    (3, 2, 16, 16, 19, 4096, "hash", False),     # the Stonehenge configuration (bound 2)
    (3, 2, 16, 16, 19, 2048, "hash", False),     # bound 1
    (3, 2, 8, 16, 15, 512, "tiled", False),
    (3, 4, 6, 8, 14, 256, "hash", True),
    (3, 8, 4, 8, 12, 64, "hash", False),
    (3, 1, 4, 8, 12, 64, "hash", False),
    (2, 2, 4, 16, 19, 2048, "hash", False),      # the background model's 2-D grid (nerf/network.py:78)
    (4, 2, 3, 4, 12, 16, "hash", False),
    (5, 2, 2, 4, 12, 8, "hash", False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"D{c[0]}C{c[1]}L{c[2]}{c[6]}{'ac' if c[7] else ''}")
@pytest.mark.parametrize("dtype", [np.float32, np.float16], ids=["f32", "f16"])
def test_grid_forward_and_jacobian_bit_exact(oracle, dev, case, dtype):
    import ngp_hip
    D, C, L, base, log2T, res, gridtype, ac = case
    if dtype == np.float16 and C == 1:
        pytest.skip("the reference never runs a half table with odd C (grid.py:38)")
    offsets, pls = oracle.grid_offsets(D, L, C, 2, base, log2T, res, ac)
    rng = np.random.default_rng(1)
    emb = rng.uniform(-1, 1, size=(offsets[-1], C)).astype(np.float32)
    emb[::7] *= 1e-4                             # some entries down in the half-subnormal range (the 1e-4 init scale)
    emb = emb.astype(dtype)
    B = 3001
    x = make_points(B, D, 2)
    gid = 0 if gridtype == "hash" else 1
    out_ref, jac_ref = oracle.grid_encode_forward(x, emb, offsets, pls, base, True, gid, ac)

    tdt = torch.float32 if dtype == np.float32 else torch.float16
    out = torch.empty(L, B, C, dtype=tdt, device=dev)
    jac = torch.empty(B, L * D * C, dtype=tdt, device=dev)
    tx, te, to = t(x, dev), t(emb, dev), t(offsets, dev)
    ngp_hip.check(ngp_hip.lib().ngp_grid_encode_forward(ngp_hip.ptr(tx), ngp_hip.ptr(te), ngp_hip.ptr(to), ngp_hip.ptr(out), B, D, C, L,
                                                        float(np.log2(pls)), base, 1, ngp_hip.ptr(jac), gid, int(ac),
                                                        ngp_hip.dtype_code(tdt), ngp_hip.stream()))
    same_bits(out, out_ref, "outputs")
    same_bits(jac, jac_ref, "dy_dx")
    assert not out_ref[:, 2].any() and not out_ref[:, 3].any()      # the out-of-range rows are zero


@pytest.mark.parametrize("autocast", [False, True], ids=["fp32", "autocast"])
def test_grid_encoder_module_forward_backward(oracle, dev, autocast):
    """GridEncoder as nerf/network.py uses it, world coordinates in [-bound, bound], with autograd."""
    from gridencoder import GridEncoder
    bound = 2.0
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048 * bound).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-0.5, 0.5, generator=torch.Generator(device=dev).manual_seed(0)) if False else \
            enc.embeddings.copy_(torch.from_numpy(np.random.default_rng(0).uniform(-0.5, 0.5, size=tuple(enc.embeddings.shape)).astype(np.float32)))
    offsets = enc.offsets.cpu().numpy()
    assert offsets[-1] == 6328848                                     # SURVEY Appendix C, bound 2
    B = 2000
    rng = np.random.default_rng(3)
    xw = rng.uniform(-bound, bound, size=(B, 3)).astype(np.float32)
    xt = t(xw, dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
        y = enc(xt, bound=bound)
    np_dt = np.float16 if autocast else np.float32
    assert y.shape == (B, 32) and y.dtype == (torch.float16 if autocast else torch.float32)

    emb = enc.embeddings.detach().cpu().numpy().astype(np_dt)
    x01 = ((xt.detach() + bound) / (2 * bound)).cpu().numpy()          # same float32 ops as GridEncoder.forward
    out_ref, jac_ref = oracle.grid_encode_forward(x01, emb, offsets, enc.per_level_scale, 16, True, 0, False)
    same_bits(y, np.ascontiguousarray(out_ref.transpose(1, 0, 2).reshape(B, 32)), "GridEncoder output")

    g = rng.normal(size=(B, 32)).astype(np.float32)
    y.backward(t(g, dev).to(y.dtype))
    g_lbc = np.ascontiguousarray(g.astype(np_dt).reshape(B, 16, 2).transpose(1, 0, 2))
    ge_ref, gi_ref = oracle.grid_encode_backward(g_lbc, x01, emb, offsets, enc.per_level_scale, 16, jac_ref, 0, False)
    # input gradient: exact in the table dtype, then the chain rule through (x + bound) / (2 bound) in float32
    gi = (xt.grad * (2 * bound)).cpu().numpy()
    np.testing.assert_allclose(gi, gi_ref.astype(np.float32), rtol=2e-6, atol=1e-6)
    # table gradient: float atomics (order-dependent); half2 atomics round every partial sum to half
    ge = enc.embeddings.grad.cpu().numpy().astype(np.float64)
    assert enc.embeddings.grad.dtype == torch.float32 and ge.shape == ge_ref.shape
    scale = np.abs(ge_ref).max()
    tol = 3e-3 if autocast else 2e-6
    assert np.max(np.abs(ge - ge_ref)) <= tol * scale, np.max(np.abs(ge - ge_ref)) / scale
    assert (ge_ref != 0).sum() > 10000


def test_grid_backward_input_gradient_bit_exact(oracle, dev):
    import ngp_hip
    D, C, L, base = 3, 2, 16, 16
    offsets, pls = oracle.grid_offsets(D, L, C, 2, base, 19, 4096, False)
    rng = np.random.default_rng(5)
    B = 1500
    x = make_points(B, D, 6)
    for dtype, tdt in ((np.float32, torch.float32), (np.float16, torch.float16)):
        emb = rng.uniform(-1, 1, size=(offsets[-1], C)).astype(dtype)
        _, jac = oracle.grid_encode_forward(x, emb, offsets, pls, base, True, 0, False)
        grad = rng.normal(size=(L, B, C)).astype(dtype)
        _, gi_ref = oracle.grid_encode_backward(grad, x, emb, offsets, pls, base, jac, 0, False)
        ge = torch.zeros(offsets[-1], C, dtype=tdt, device=dev)
        gi = torch.zeros(B, D, dtype=tdt, device=dev)
        ngp_hip.check(ngp_hip.lib().ngp_grid_encode_backward(ngp_hip.ptr(t(grad, dev)), ngp_hip.ptr(t(x, dev)), ngp_hip.ptr(t(emb, dev)),
                                                             ngp_hip.ptr(t(offsets, dev)), ngp_hip.ptr(ge), B, D, C, L, float(np.log2(pls)),
                                                             base, 1, ngp_hip.ptr(t(jac, dev)), ngp_hip.ptr(gi), 0, 0,
                                                             ngp_hip.dtype_code(tdt), ngp_hip.stream()))
        same_bits(gi, gi_ref, f"grad_inputs {dtype.__name__}")


def test_grid_rejects_unsupported(dev):
    import ngp_hip
    z = torch.zeros(16, device=dev)
    rc = ngp_hip.lib().ngp_grid_encode_forward(ngp_hip.ptr(z), ngp_hip.ptr(z), ngp_hip.ptr(z), ngp_hip.ptr(z), 1, 3, 3, 1, 1.0, 16, 0, None,
                                               0, 0, 0, ngp_hip.stream())
    assert rc == -1 and b"C must be 1, 2, 4, or 8" in ngp_hip.lib().ngp_last_error()
    rc = ngp_hip.lib().ngp_grid_encode_forward(ngp_hip.ptr(z), ngp_hip.ptr(z), ngp_hip.ptr(z), ngp_hip.ptr(z), 1, 6, 2, 1, 1.0, 16, 0, None,
                                               0, 0, 0, ngp_hip.stream())
    assert rc == -1


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encoder_forward_backward(dev, degree):
    from oracle import sh_oracle
    from shencoder import SHEncoder
    rng = np.random.default_rng(degree)
    B = 5000
    v = rng.normal(size=(B, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[:50] *= rng.uniform(0.5, 1.5, size=(50, 1)).astype(np.float32)      # not exactly unit: the polynomials still apply
    ref, jac = sh_oracle.sh_encode(v, degree, True)
    enc = SHEncoder(degree=degree)
    tv = t(v, dev).requires_grad_(True)
    y = enc(tv)
    assert y.shape == (B, degree * degree) and y.dtype == torch.float32
    # float32 recurrences vs float64 polynomials: a few ulp of the largest term (|Y| <= ~3 at degree 8 on the unit sphere)
    atol = 2e-6 * max(1.0, float(np.abs(ref).max()))
    assert np.max(np.abs(y.detach().cpu().numpy() - ref)) < atol
    g = rng.normal(size=ref.shape).astype(np.float32)
    y.backward(t(g, dev))
    gi_ref = sh_oracle.sh_encode_backward(g, jac, degree)
    assert np.max(np.abs(tv.grad.cpu().numpy() - gi_ref)) < 3e-5 * max(1.0, float(np.abs(gi_ref).max()))
    # no gradient requested => backward returns None and dy_dx is never materialised
    y2 = enc(t(v, dev))
    assert not y2.requires_grad


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encoder_against_the_reference_polynomial_table(dev, degree):
    """Golden: the reference's own 64 + 192 polynomials (shencoder/src/shencoder.cu:50-355) evaluated in float64 from their text
    (tests/golden/make_sh_golden.py).  The HIP kernel uses float32 recurrences: tolerance 2e-6 of the largest term for the outputs,
    3e-5 of the largest input gradient for the backward (sum of up to 64 float32 products)."""
    import os
    from shencoder import SHEncoder
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sh_deg8.npz"))
    v, C2 = z["inputs"], degree * degree
    ref, jac = z["outputs"][:, :C2], z["dy_dx"][:, :, :C2]
    enc = SHEncoder(degree=degree)
    tv = t(v, dev).requires_grad_(True)
    y = enc(tv)
    assert np.max(np.abs(y.detach().cpu().numpy() - ref)) < 2e-6 * max(1.0, float(np.abs(ref).max()))
    rng = np.random.default_rng(degree)
    g = rng.normal(size=ref.shape).astype(np.float32)
    y.backward(t(g, dev))
    gi_ref = np.einsum("bc,bdc->bd", g.astype(np.float64), jac)              # shencoder.cu:359-383
    assert np.max(np.abs(tv.grad.cpu().numpy() - gi_ref)) < 3e-5 * max(1.0, float(np.abs(gi_ref).max()))


def test_trunc_exp_product_against_reference_golden(dev):
    """T1: ngp.field.trunc_exp (the product's activation, activation.py:5-18) forward and backward on the GPU against the vectors the
    reference's own module produced (tests/golden/trunc_exp.npz, incl. +-15.0001, 88, -104).  float32 exp of two libms: 2 ulp."""
    import os
    from ngp.field import trunc_exp
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "trunc_exp.npz"))
    x, g, y_ref, dx_ref = z["x"], z["g"], z["y"], z["dx"]
    xt = t(x, dev).requires_grad_(True)
    y = trunc_exp(xt)
    y.backward(t(g, dev))
    y, dx = y.detach().cpu().numpy(), xt.grad.cpu().numpy()
    fin = np.isfinite(y_ref)
    assert np.array_equal(np.isfinite(y), fin)                               # exp(88) finite, overflow to inf at the same inputs
    assert np.max(np.abs(y[fin] - y_ref[fin]) / np.maximum(y_ref[fin], 1e-45)) < 2.5e-7
    assert np.max(np.abs(dx - dx_ref) / np.maximum(np.abs(dx_ref), 1e-30)) < 5e-7
    # the clamp: beyond +-15 the gradient is g * exp(+-15) exactly as in the reference, not g * y
    far = np.abs(x) > 15
    assert far.sum() > 5 and np.allclose(dx[far], g[far] * np.exp(np.clip(x[far], -15, 15)), rtol=5e-7)
    # under autocast the input is cast to float32 first (custom_fwd(cast_inputs=float32), activation.py:7)
    with torch.autocast("cuda", dtype=torch.float16):
        yh = trunc_exp(t(x[:64], dev).half())
    assert yh.dtype == torch.float32


# ---------------------------------------------------------------------------------------------------------------------
# freqencoder (optional fifth module)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,degree", [(3, 6), (3, 4), (2, 10), (5, 1)])
def test_freq_encoder_bit_exact_and_gradient(oracle, dev, D, degree):
    from freqencoder import FreqEncoder, freq_encode
    rng = np.random.default_rng(D * 100 + degree)
    x = rng.uniform(-2, 2, size=(3001, D)).astype(np.float32)
    x[:3] = 0.0
    enc = FreqEncoder(input_dim=D, degree=degree)
    assert enc.output_dim == D + 2 * D * degree
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    y = enc(xt)
    ref = oracle.freq_encode_forward(x, degree)
    assert y.shape == ref.shape and np.array_equal(y.detach().cpu().numpy().view(np.uint32), ref.view(np.uint32))
    g = rng.normal(size=ref.shape).astype(np.float32)
    y.backward(torch.from_numpy(g).to(dev))
    dref = oracle.freq_encode_backward(g, ref, D, degree)
    assert np.array_equal(xt.grad.cpu().numpy().view(np.uint32), dref.view(np.uint32))
    # prefix shapes, autocast (inputs are cast to float32), empty batch
    with torch.autocast("cuda", dtype=torch.float16):
        y3 = enc(torch.from_numpy(x[:12]).to(dev).half().view(3, 4, D))
    assert y3.shape == (3, 4, enc.output_dim) and y3.dtype == torch.float32
    assert freq_encode(torch.zeros(0, D, device=dev), degree, enc.output_dim).shape == (0, enc.output_dim)


def test_get_encoder_frequency(dev):
    from ngp.field import get_encoder
    enc, dim = get_encoder("frequency", input_dim=3, multires=6)
    assert dim == 39 and enc(torch.zeros(5, 3, device=dev)).shape == (5, 39)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("level_dim,gridtype,align", [(2, "hash", False), (4, "tiled", True), (1, "hash", False)])
def test_grid_input_gradient_without_dy_dx_is_bit_identical(dev, half, level_dim, gridtype, align):
    """gridencoder/grid.py:45-48,80-84 saves dy_dx [B, L*D*C] for the input gradient; the default route here recomputes it
    in backward (ngp_grid_encode_backward_inputs).  Same bits as the saved-Jacobian route, with and without the table
    gradient, including points outside [0,1] (zero gradient)."""
    import gridencoder.grid as G
    torch.manual_seed(3)
    enc = G.GridEncoder(input_dim=3, num_levels=8, level_dim=level_dim, base_resolution=8, log2_hashmap_size=12,
                        desired_resolution=256, gridtype=gridtype, align_corners=align).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = torch.rand(5000, 3, device=dev) * 2.4 - 1.2                     # bound 1: about a third of the points are outside
    gout = torch.randn(5000, 8 * level_dim, device=dev)

    def run(recompute, frozen):
        G.RECOMPUTE_INPUT_GRAD, G.RECOMPUTE_MIN_POINTS = recompute, 0
        enc.embeddings.requires_grad_(not frozen)
        enc.embeddings.grad = None
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=half):
            y = enc(xi, bound=1)
        y.backward(gout.to(y.dtype))
        return y.detach(), xi.grad, None if frozen else enc.embeddings.grad.clone()

    try:
        for frozen in (True, False):
            y0, g0, t0 = run(False, frozen)
            y1, g1, t1 = run(True, frozen)
            assert torch.equal(y0, y1) and torch.equal(g0, g1) and g0.abs().max() > 0
            outside = ((x < -1) | (x > 1)).any(dim=-1)
            assert outside.sum() > 100 and (g1[outside] == 0).all()
            if not frozen:
                # the same scatter both times; atomics make the sum order-dependent (half2 atomics round every partial sum: ~36 adds per coarse row here,
                # each good to 2^-11 of the running sum; 5e-3 was exceeded once in about ten full runs of the suite, on C = 4 where runs are not pre-merged)
                assert (t0 - t1).abs().max() <= (1.5e-2 if half else 1e-5) * t0.abs().max()
    finally:
        G.RECOMPUTE_INPUT_GRAD, G.RECOMPUTE_MIN_POINTS = True, 32768
        enc.embeddings.requires_grad_(True)


@pytest.mark.parametrize("half", [False, True])
def test_grid_forward_rows_equals_level_major_plus_permute(dev, half):
    """ngp_grid_encode_forward_rows writes [B, L*C] directly; the reference's route is [L, B, C] + permute + copy (grid.py:42,52).
    Same bits for float32 and float16 tables, with out-of-range points (zero rows) and a batch that is not a multiple of 256."""
    import gridencoder.grid as G
    torch.manual_seed(7)
    enc = G.GridEncoder(desired_resolution=4096).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = torch.rand(70001, 3, device=dev) * 2.4 - 1.2
    outs = {}
    for rows in (True, False):
        G.ROWS_FORWARD = rows
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=half):
                outs[rows] = enc(x, bound=1)
        finally:
            G.ROWS_FORWARD = True
    assert outs[True].shape == (70001, 32) and outs[True].dtype == (torch.float16 if half else torch.float32)
    assert torch.equal(outs[True], outs[False])
    outside = ((x < -1) | (x > 1)).any(dim=-1)
    assert outside.sum() > 1000 and (outs[True][outside] == 0).all() and outs[True][~outside].abs().sum() > 0


def test_grid_32_level_float32_table_takes_the_level_major_route(dev, oracle):
    """a 32-level float32 table needs 256 x 33 x 8 B = 67,584 B of LDS in the row kernel, more than the 64 KiB a kernel gets by
    default: the entry point refuses it (status code, no launch) and the module falls back to [L, B, C] + permute.  Values against the oracle."""
    import gridencoder.grid as G
    import ngp_hip
    enc = G.GridEncoder(num_levels=32, log2_hashmap_size=12, base_resolution=4, desired_resolution=512).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = torch.rand(5000, 3, device=dev) * 2 - 1
    with torch.no_grad():
        out = enc(x, bound=1)
    assert out.shape == (5000, 64)
    x01 = ((x + 1) / 2).cpu().numpy()
    ref, _ = oracle.grid_encode_forward(x01, enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy(), enc.per_level_scale, 4, False, 0, False)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), np.ascontiguousarray(ref.transpose(1, 0, 2).reshape(5000, 64)).view(np.uint32))
    rows = torch.empty(5000, 64, device=dev)
    rc = ngp_hip.lib().ngp_grid_encode_forward_rows(ngp_hip.ptr(x), ngp_hip.ptr(enc.embeddings.detach()), ngp_hip.ptr(enc.offsets), ngp_hip.ptr(rows), 5000, 3, 2, 32,
                                                    float(np.log2(enc.per_level_scale)), 4, 0, 0, ngp_hip.F32, ngp_hip.stream())
    assert rc != 0 and b"LDS" in ngp_hip.lib().ngp_last_error()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):                  # the half table of the same encoder fits (33,792 B): row kernel
        assert enc(x, bound=1).shape == (5000, 64)
