"""CPU: pins the oracle for the encoders, the MLP and trunc_exp (SURVEY.md 8c relations 4-7).
  * grid encoder: dense levels == textbook trilinear interpolation of the vertex lattice; hashed levels == a
    brute-force python loop over the reference's index formula; dy_dx == finite differences; half == float within
    the half rounding bound;
  * SH: 13 of the reference's own polynomials (cited by line) + finite differences of the Jacobian;
  * MLP: torch.nn.functional.linear chain;  trunc_exp: golden vectors generated from the reference's activation.py."""
import os

import numpy as np
import torch

from oracle import sh_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_grid_level_table_matches_reference_appendix(oracle):
    # SURVEY Appendix C (from gridencoder/grid.py:113-123): bound 1 and bound 2 tables
    off1, pls1 = oracle.grid_offsets(3, 16, 2, 2, 16, 19, 2048, False)
    off2, pls2 = oracle.grid_offsets(3, 16, 2, 2, 16, 19, 4096, False)
    assert off1[-1] == 6119864 and off2[-1] == 6328848
    assert np.diff(off2)[:6].tolist() == [4920, 15632, 42880, 125000, 373248, 524288]
    assert np.diff(off1)[:6].tolist() == [4920, 13824, 32768, 85184, 216000, 524288]
    assert abs(pls1 - 1.381913) < 1e-6 and abs(pls2 - 1.447269) < 1e-6
    scale, reso = oracle.grid_level_table(16, np.float32(np.log2(pls2)), 16)
    assert reso[0] == 16 and reso[15] == 4096                       # the kernel's resolution (python's table says 4097)


def _trilinear_dense(x01, emb_level, scale, res):
    """textbook: value at vertex (i,j,k) = emb[i + j*(res+1) + k*(res+1)^2]; sample at x*scale + 0.5"""
    pos = x01.astype(np.float32) * np.float32(scale) + np.float32(0.5)          # float32, as the kernel positions the sample
    i0 = np.floor(pos).astype(np.int64)
    f = (pos - i0.astype(np.float32)).astype(np.float64)
    out = np.zeros((x01.shape[0], emb_level.shape[1]))
    s = res + 1
    for c in range(8):
        o = np.array([(c >> d) & 1 for d in range(3)])
        w = np.prod(np.where(o, f, 1 - f), axis=1)
        idx = (i0[:, 0] + o[0]) + (i0[:, 1] + o[1]) * s + (i0[:, 2] + o[2]) * s * s
        out += w[:, None] * emb_level[idx].astype(np.float64)
    return out


def test_grid_dense_levels_are_trilinear_and_hashed_levels_follow_the_hash(oracle):
    offsets, pls = oracle.grid_offsets(3, 16, 2, 2, 16, 19, 4096, False)
    rng = np.random.default_rng(0)
    emb = rng.uniform(-1, 1, size=(offsets[-1], 2)).astype(np.float32)
    x = rng.uniform(0, 1, size=(300, 3)).astype(np.float32)
    out, _ = oracle.grid_encode_forward(x, emb, offsets, pls, 16, False, 0, False)      # [L,B,C]
    scale, reso = oracle.grid_level_table(16, np.float32(np.log2(pls)), 16)
    for level in range(5):                                                                # dense: (res+1)^3 <= rows
        assert (reso[level] + 1) ** 3 <= offsets[level + 1] - offsets[level]
        want = _trilinear_dense(x, emb[offsets[level]:offsets[level + 1]], scale[level], int(reso[level]))
        assert np.max(np.abs(out[level] - want)) < 2e-6
    P = [1, 2654435761, 805459861]
    for level in (5, 9, 15):                                                              # hashed
        size = int(offsets[level + 1] - offsets[level])
        for b in range(0, 300, 37):
            pos = x[b] * np.float32(scale[level]) + np.float32(0.5)
            i0 = np.floor(pos).astype(np.int64); f = (pos - i0.astype(np.float32)).astype(np.float64)
            acc = np.zeros(2)
            for c in range(8):
                o = [(c >> d) & 1 for d in range(3)]
                w = np.prod([f[d] if o[d] else 1 - f[d] for d in range(3)])
                h = 0
                for d in range(3):
                    h ^= ((int(i0[d]) + o[d]) * P[d]) & 0xFFFFFFFF
                acc += w * emb[offsets[level] + h % size]
            assert np.max(np.abs(out[level, b] - acc)) < 2e-6


def test_grid_out_of_range_and_tiled(oracle):
    offsets, pls = oracle.grid_offsets(3, 4, 2, 2, 16, 19, 128, False)
    rng = np.random.default_rng(1)
    emb = rng.uniform(-1, 1, size=(offsets[-1], 2)).astype(np.float32)
    x = np.array([[0.5, 0.5, 0.5], [-0.01, 0.5, 0.5], [0.5, 1.01, 0.5], [0.0, 1.0, 0.0]], np.float32)
    out, jac = oracle.grid_encode_forward(x, emb, offsets, pls, 16, True, 0, False)
    assert not out[:, 1].any() and not out[:, 2].any() and out[:, 0].any() and out[:, 3].any()
    assert not jac[1].any() and not jac[2].any()
    out_t, _ = oracle.grid_encode_forward(x, emb, offsets, pls, 16, False, 1, False)      # tiled: same on dense levels
    assert np.array_equal(out[0], out_t[0])


def test_grid_jacobian_vs_finite_differences_and_input_gradient(oracle):
    offsets, pls = oracle.grid_offsets(3, 8, 2, 2, 16, 19, 512, False)
    rng = np.random.default_rng(2)
    emb = rng.uniform(-1, 1, size=(offsets[-1], 2)).astype(np.float32)
    x = rng.uniform(0.05, 0.95, size=(200, 3)).astype(np.float32)
    out, jac = oracle.grid_encode_forward(x, emb, offsets, pls, 16, True, 0, False)
    jac = jac.reshape(200, 8, 3, 2)
    scale, _ = oracle.grid_level_table(8, np.float32(np.log2(pls)), 16)
    checked = 0
    for d in range(3):
        for level in range(8):
            # stay inside one cell: step much smaller than the cell, skip samples within the step of a cell face
            eps = 1e-3 / float(scale[level])
            pos = x[:, d].astype(np.float64) * float(scale[level]) + 0.5
            fr = pos - np.floor(pos)
            ok = (fr > 0.05) & (fr < 0.95)
            xp, xm = x.astype(np.float64).copy(), x.astype(np.float64).copy()
            xp[:, d] += eps; xm[:, d] -= eps
            op, _ = oracle.grid_encode_forward(xp.astype(np.float32), emb, offsets, pls, 16, False, 0, False)
            om, _ = oracle.grid_encode_forward(xm.astype(np.float32), emb, offsets, pls, 16, False, 0, False)
            step = (xp.astype(np.float32)[:, d].astype(np.float64) - xm.astype(np.float32)[:, d].astype(np.float64))
            fd = (op[level].astype(np.float64) - om[level].astype(np.float64)) / step[:, None]
            err = np.abs(fd - jac[:, level, d])[ok]
            assert err.max() < 2e-2 * max(1.0, float(scale[level])), (d, level, err.max())
            checked += int(ok.sum())
    assert checked > 3000
    g = rng.normal(size=(8, 200, 2)).astype(np.float32)
    ge, gi = oracle.grid_encode_backward(g, x, emb, offsets, pls, 16, jac.reshape(200, -1), 0, False)
    want = np.einsum("lbc,bldc->bd", g.astype(np.float64), jac.astype(np.float64))
    assert np.max(np.abs(gi - want)) < 1e-3 * np.abs(want).max()
    # table gradient: sum of the weights scattered to a level == sum of the incoming gradient (weights sum to 1)
    for level in range(8):
        assert abs(ge[offsets[level]:offsets[level + 1]].sum() - g[level].astype(np.float64).sum()) < 1e-3


def test_grid_half_arithmetic_close_to_float(oracle):
    offsets, pls = oracle.grid_offsets(3, 16, 2, 2, 16, 19, 4096, False)
    rng = np.random.default_rng(3)
    emb = rng.uniform(-1, 1, size=(offsets[-1], 2)).astype(np.float32)
    x = rng.uniform(0, 1, size=(500, 3)).astype(np.float32)
    o32, _ = oracle.grid_encode_forward(x, emb.astype(np.float16).astype(np.float32), offsets, pls, 16, False, 0, False)
    o16, _ = oracle.grid_encode_forward(x, emb.astype(np.float16), offsets, pls, 16, False, 0, False)
    assert o16.dtype == np.float16
    # 16 roundings to half of values <= 1 in magnitude: <= 16 * 2^-11
    assert np.max(np.abs(o16.astype(np.float32) - o32)) < 16 * 2.0 ** -11


def test_sh_matches_reference_polynomials_and_finite_differences():
    rng = np.random.default_rng(0)
    v = rng.normal(size=(500, 3))
    out, jac = sh_oracle.sh_encode(v, 8, True)
    for idx, line, f in sh_oracle.REFERENCE_SPOT_TERMS:                                   # shencoder.cu:<line>
        ref = f(v[:, 0], v[:, 1], v[:, 2])
        assert np.max(np.abs(out[:, idx] - ref) / (1 + np.abs(ref))) < 1e-12, (idx, line)
    jac = jac.reshape(500, 3, 64)
    eps = 1e-6
    for d in range(3):
        vp, vm = v.copy(), v.copy()
        vp[:, d] += eps; vm[:, d] -= eps
        fd = (sh_oracle.sh_encode(vp, 8) - sh_oracle.sh_encode(vm, 8)) / (2 * eps)
        assert np.max(np.abs(fd - jac[:, d]) / (1 + np.abs(fd))) < 1e-6
    # orthonormality on the sphere (Monte Carlo): integral Y_i Y_j = delta_ij
    u = rng.normal(size=(200000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    Y = sh_oracle.sh_encode(u, 4)
    G = 4 * np.pi * (Y.T @ Y) / u.shape[0]
    assert np.max(np.abs(G - np.eye(16))) < 0.03


def test_sh_oracle_equals_all_256_reference_polynomials():
    """tests/golden/sh_deg8.npz = the 64 outputs and 192 partial derivatives of shencoder/src/shencoder.cu:50-355, read as text and
    evaluated in float64 by tests/golden/make_sh_golden.py: the oracle's Legendre-derived basis must BE that table."""
    z = np.load(os.path.join(GOLDEN, "sh_deg8.npz"))
    v = z["inputs"].astype(np.float64)
    for degree in range(1, 9):
        out, jac = sh_oracle.sh_encode(v, degree, True)
        C2 = degree * degree
        jac = jac.reshape(-1, 3, C2)
        assert np.max(np.abs(out - z["outputs"][:, :C2])) < 1e-12 * max(1.0, np.abs(z["outputs"][:, :C2]).max())
        assert np.max(np.abs(jac - z["dy_dx"][:, :, :C2])) < 1e-12 * max(1.0, np.abs(z["dy_dx"][:, :, :C2]).max())


def test_ffmlp_oracle_vs_torch_linear(oracle):
    rng = np.random.default_rng(0)
    for input_dim, num_layers in ((32, 2), (32, 3), (64, 2)):
        B = 64
        nw = oracle.ffmlp_num_params(input_dim, 16, 64, num_layers)
        assert nw == 64 * (input_dim + 64 * (num_layers - 1) + 16)                       # ffmlp.py:120
        w = rng.uniform(-0.2, 0.2, nw).astype(np.float16)
        x = rng.normal(size=(B, input_dim)).astype(np.float16)
        out, fb = oracle.ffmlp_forward(x, w, input_dim, 16, 64, num_layers, save=True)
        h = torch.from_numpy(x.astype(np.float32)).double()
        wt = torch.from_numpy(w.astype(np.float32)).double()
        off = 0
        dims = [input_dim] + [64] * num_layers + [16]
        for m in range(num_layers + 1):
            W = wt[off:off + dims[m + 1] * dims[m]].view(dims[m + 1], dims[m]); off += dims[m + 1] * dims[m]
            h = torch.nn.functional.linear(h, W)
            if m < num_layers:
                h = torch.relu(h)
            h = h.float().half().double()                                                # the layer output is stored as half
            if m < num_layers:
                assert np.array_equal(fb[m].view(np.uint16), h.half().numpy().view(np.uint16))
        assert np.array_equal(out.view(np.uint16), h.half().numpy().view(np.uint16))


def test_ffmlp_backward_oracle_vs_autograd(oracle):
    rng = np.random.default_rng(1)
    input_dim, num_layers, B = 32, 3, 48
    nw = oracle.ffmlp_num_params(input_dim, 16, 64, num_layers)
    w = rng.uniform(-0.2, 0.2, nw).astype(np.float16)
    x = rng.normal(size=(B, input_dim)).astype(np.float16)
    g = rng.normal(size=(B, 16)).astype(np.float16)
    out, fb = oracle.ffmlp_forward(x, w, input_dim, 16, 64, num_layers, save=True)
    gw, gi, bb = oracle.ffmlp_backward(g, x, w, fb, input_dim, 16, 64, num_layers, True)
    xt = torch.from_numpy(x.astype(np.float32)).double().requires_grad_(True)
    wt = torch.from_numpy(w.astype(np.float32)).double().requires_grad_(True)
    h, off, dims = xt, 0, [input_dim] + [64] * num_layers + [16]
    for m in range(num_layers + 1):
        W = wt[off:off + dims[m + 1] * dims[m]].view(dims[m + 1], dims[m]); off += dims[m + 1] * dims[m]
        h = torch.nn.functional.linear(h, W)
        if m < num_layers:
            h = torch.relu(h)
    h.backward(torch.from_numpy(g.astype(np.float32)).double())
    # the oracle rounds activations and back-propagated gradients to half per layer (as the reference stores them):
    # agreement to a few half ulps of the gradient scale
    assert np.max(np.abs(gw - wt.grad.numpy())) < 2e-2 * np.abs(wt.grad.numpy()).max()
    assert np.max(np.abs(gi - xt.grad.numpy())) < 2e-2 * np.abs(xt.grad.numpy()).max()
    assert bb.shape == (num_layers, B, 64)


def test_trunc_exp_golden_from_reference_activation(oracle):
    """activation.py:5-18 (vectors made by tests/golden/make_golden.py from the reference's own module)"""
    z = np.load(os.path.join(GOLDEN, "trunc_exp.npz"))
    x, g, y, dx = z["x"], z["g"], z["y"], z["dx"]
    with np.errstate(over="ignore"):
        assert np.allclose(y, np.exp(x.astype(np.float32)), rtol=2e-7, atol=0)
        assert np.allclose(dx, g * np.exp(np.clip(x, -15, 15)), rtol=2e-6, atol=0)
    # the deterministic exp the fused kernel uses for trunc_exp's forward agrees with it to 2 ulp over its range
    m = (x > -87) & (x < 88)
    assert np.max(np.abs(oracle.expf(x[m]) - y[m]) / y[m]) < 2.5e-7


# ---------------------------------------------------------------------------------------------------------------------
# freqencoder (the optional fifth module): oracle vs the reference's own pure-torch FreqEncoder and vs float64
# ---------------------------------------------------------------------------------------------------------------------
def test_sinf_stand_in_accuracy(oracle):
    x = np.concatenate([np.linspace(-70, 70, 200001), np.random.default_rng(0).uniform(-8192, 8192, 100000)]).astype(np.float32)
    y = oracle.sinf(x)
    ref = np.sin(x.astype(np.float64))
    assert np.max(np.abs(y - ref)[np.abs(x) <= 70]) < 1.3e-7                         # about one ulp of 1.0
    assert np.max(np.abs(y - ref)) < 2e-4                                            # three-part reduction degrades slowly
    assert oracle.sinf(np.float32([0.0]))[0] == 0.0 and oracle.sinf(np.float32([-0.0]))[0] == 0.0


def test_freq_encoder_matches_the_reference_torch_class(oracle):
    """tests/golden/freq_encoder.npz was produced by the reference's encoding.FreqEncoder (pure torch) and torch autograd."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "freq_encoder.npz"))
    deg = int(z["degree"])
    y = oracle.freq_encode_forward(z["x"], deg)
    assert y.shape == z["y"].shape == (z["x"].shape[0], 3 + 6 * deg)
    assert np.array_equal(y[:, :3], z["x"])                                          # the input copy is exact
    # torch: sin / cos of (x * 2^f) in float32 libm; ours: sin(ldexp(x, f) + phase) with the deterministic sinf.  The cosine
    # column adds a float32 pi/2 to the argument, which costs up to half an ulp of the argument (|arg| <= 64 + 1.6).
    assert np.max(np.abs(y - z["y"])) < 5e-6
    dx = oracle.freq_encode_backward(z["g"], y, 3, deg)
    assert np.max(np.abs(dx - z["dx"])) < 2e-3 * np.max(np.abs(z["dx"]))              # sums of 2^f-weighted terms, f up to 5
    # and against float64 directly
    xs = z["x"].astype(np.float64)
    want = np.concatenate([xs] + [fn(xs * 2.0 ** f) for f in range(deg) for fn in (np.sin, np.cos)], axis=1)
    assert np.max(np.abs(y - want)) < 5e-6
