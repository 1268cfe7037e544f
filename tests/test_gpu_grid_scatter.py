"""GPU: the binned table-gradient scatter (csrc/gridencoder.hip "Binned scatter": no global atomics, float32 sums in LDS) against the CPU
oracle's float64 scatter (oracle/ngp_oracle.c o_grid_encode_backward, gridencoder.cu:227-314) and against the atomic kernel it replaces."""
import importlib

import numpy as np
import pytest
import torch

importlib.import_module("nerf-navigation_amd")
pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def ray_points(n_rays, per_ray, seed):
    """march-like points: runs of consecutive samples along rays (so that the run aggregation has something to merge) + out-of-range points"""
    rng = np.random.default_rng(seed)
    o = rng.uniform(0.05, 0.95, (n_rays, 1, 3))
    d = rng.normal(size=(n_rays, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    s = (np.arange(per_ray)[None, :, None] * 0.0017)
    x = (o + d * s).reshape(-1, 3).astype(np.float32)
    x[::97] = rng.uniform(-0.2, 1.2, (len(x[::97]), 3)).astype(np.float32)      # some outside [0,1]: contribute nothing
    return x


@pytest.mark.parametrize("case", [dict(B=70001, levels=16, log2T=19, res=4096), dict(B=5000, levels=16, log2T=19, res=4096),
                                   dict(B=1, levels=16, log2T=19, res=4096), dict(B=30011, levels=5, log2T=14, res=256)],
                         ids=["70001", "5000", "one", "small_table"])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16], ids=["f32out", "f16out"])
def test_binned_scatter_against_oracle_and_atomic_kernel(oracle, dev, case, out_dtype):
    import ngp_hip
    from gridencoder import grid as G
    B, L = case["B"], case["levels"]
    offsets, pls = oracle.grid_offsets(3, L, 2, 2, 16, case["log2T"], case["res"], False)
    x = ray_points(max(B // 50, 1) + 1, 50, 3)[:B]
    rng = np.random.default_rng(4)
    grad = (rng.normal(size=(L, B, 2)) * 0.1).astype(np.float16)
    grad[:, ::13] = 0                                                   # samples behind a saturated ray carry exact zeros
    emb = np.zeros((int(offsets[-1]), 2), np.float16)
    ref, _ = oracle.grid_encode_backward(grad, x, emb, offsets, pls, 16)                # float64 sums of half(w * g) products
    scale = 0.5
    out = G.table_gradient_binned(t(grad, dev), t(x, dev), t(offsets, dev), B, L, np.log2(pls), 16, 0, False, out_dtype=out_dtype, out_scale=scale)
    assert out.dtype == out_dtype and tuple(out.shape) == ref.shape
    got = out.float().cpu().numpy() / scale
    big = np.abs(ref).max()
    # float32 sums of half-rounded (run-summed) products: 2e-3 of the largest entry covers the half rounding of a run's sum; half output adds its own
    tol = 2e-3 if out_dtype == torch.float32 else 4e-3
    assert np.max(np.abs(got - ref)) <= tol * big, (np.max(np.abs(got - ref)), big)
    if out_dtype == torch.float32:                                     # the same rows are touched (half output flushes the smallest sums to zero)
        assert np.count_nonzero((got != 0) != (ref != 0)) <= 1e-4 * np.count_nonzero(ref)
    # ... and closer to the float64 oracle than the half2-atomic kernel it replaces
    old = torch.zeros(int(offsets[-1]), 2, dtype=torch.float16, device=dev)
    dummy = torch.empty(1, dtype=torch.float16, device=dev)
    ngp_hip.check(ngp_hip.lib().ngp_grid_encode_backward(ngp_hip.ptr(t(grad, dev)), ngp_hip.ptr(t(x, dev)), ngp_hip.ptr(old), ngp_hip.ptr(t(offsets, dev)), ngp_hip.ptr(old),
                                                         B, 3, 2, L, float(np.log2(pls)), 16, 0, ngp_hip.ptr(dummy), ngp_hip.ptr(dummy), 0, 0, ngp_hip.F16, ngp_hip.stream()))
    err_old = np.linalg.norm(old.float().cpu().numpy() - ref)
    err_new = np.linalg.norm(got - ref)
    if out_dtype == torch.float32 and B > 1:
        assert err_new <= err_old * 1.05


def test_binned_scatter_multi_pass_and_empty(oracle, dev):
    """more than 2^22 samples run in passes that add into the output; B = 0 writes zeros"""
    from gridencoder import grid as G
    L = 4
    offsets, pls = oracle.grid_offsets(3, L, 2, 2, 16, 12, 64, False)
    B = (1 << 22) + 12345
    x = ray_points(B // 64 + 1, 64, 5)[:B]
    grad = np.full((L, B, 2), 0.001, np.float16)
    out = G.table_gradient_binned(t(grad, dev), t(x, dev), t(offsets, dev), B, L, np.log2(pls), 16, 0, False)
    ref, _ = oracle.grid_encode_backward(grad, x, np.zeros((int(offsets[-1]), 2), np.float16), offsets, pls, 16)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 2e-3 * np.abs(ref).max()
    z = G.table_gradient_binned(torch.empty(L, 0, 2, dtype=torch.float16, device=dev), torch.empty(0, 3, device=dev), t(offsets, dev), 0, L, np.log2(pls), 16, 0, False)
    assert z.shape[0] == int(offsets[-1]) and float(z.abs().max()) == 0.0


def test_grid_encoder_module_uses_the_binned_scatter_under_autocast(dev):
    """_grid_encode.backward under autocast: binned (float32 gradient straight into .grad) == the atomic route to half precision"""
    import gridencoder.grid as G
    torch.manual_seed(3)
    enc = G.GridEncoder(desired_resolution=4096).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = torch.from_numpy(ray_points(800, 50, 6)).to(dev) * 2 - 1
    w = torch.randn(x.shape[0], 32, device=dev)
    grads = {}
    for binned in (True, False):
        G.BINNED_SCATTER = binned
        try:
            enc.embeddings.grad = None
            with torch.autocast("cuda", dtype=torch.float16):
                (enc(x, bound=1).float() * w).sum().backward()
            grads[binned] = enc.embeddings.grad.clone()
        finally:
            G.BINNED_SCATTER = True
    assert grads[True].dtype == torch.float32
    assert float((grads[True] - grads[False]).abs().max()) <= 4e-3 * float(grads[False].abs().max())
    assert float(grads[True].abs().max()) > 0


def test_binned_scatter_in_level_groups_equals_one_call(oracle, dev):
    """ngp_grid_scatter_binned_phase: bin once, then sum the table one group of levels at a time (what the gradient exchange uses to overlap its
    all-reduces): the sums are exact and order-independent, so the result equals the single call BIT FOR BIT; the callback sees contiguous row
    ranges that cover the table once, finest levels first."""
    from gridencoder import grid as G
    L = 16
    offsets, pls = oracle.grid_offsets(3, L, 2, 2, 16, 19, 4096, False)
    B = 50003
    x = ray_points(B // 50 + 1, 50, 7)[:B]
    grad = (np.random.default_rng(8).normal(size=(L, B, 2)) * 0.05).astype(np.float16)
    tg, tx, to = t(grad, dev), t(x, dev), t(offsets, dev)
    one = G.table_gradient_binned(tg, tx, to, B, L, np.log2(pls), 16, 0, False, out_dtype=torch.float16, out_scale=0.25)
    seen = []
    grouped = G.table_gradient_binned(tg, tx, to, B, L, np.log2(pls), 16, 0, False, out_dtype=torch.float16, out_scale=0.25,
                                      on_group=lambda out, r0, r1: seen.append((r0, r1)))
    assert torch.equal(one, grouped)
    assert len(seen) == G.LEVEL_GROUPS and seen[0][1] == int(offsets[-1]) and seen[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(seen[:-1], seen[1:]))                      # contiguous, descending
    G.LEVEL_GROUPS, keep = 4, G.LEVEL_GROUPS
    try:
        seen4 = []
        four = G.table_gradient_binned(tg, tx, to, B, L, np.log2(pls), 16, 0, False, out_dtype=torch.float16, out_scale=0.25,
                                       on_group=lambda out, r0, r1: seen4.append((r0, r1)))
    finally:
        G.LEVEL_GROUPS = keep
    assert torch.equal(one, four) and len(seen4) == 4
    assert seen4[-1][1] - seen4[-1][0] < 0.05 * int(offsets[-1])                        # with four groups the last (exposed) one is levels 0-3: 3 % of the rows
    f32 = G.table_gradient_binned(tg, tx, to, B, L, np.log2(pls), 16, 0, False, on_group=lambda *a: None)
    assert torch.equal(f32, G.table_gradient_binned(tg, tx, to, B, L, np.log2(pls), 16, 0, False))


def test_binned_scatter_full_size_checksum_and_reproducibility(oracle, dev):
    """BASELINE config 3's size (4,096 rays x up to 512 samples = 2.1 M march-ordered points, 16 levels, 2^19-row hashed levels) through properties that do not
    need the oracle's minutes: (1) the eight trilinear weights of a sample sum to 1, so every level's column sums of the table gradient equal the column sums
    of that level's incoming gradient over the samples inside [0,1]^3 -- a checksum of checksums, to the half rounding of the products; (2) the sums are exact
    and order-independent: a second call gives the same bits; (3) linearity in the gradient for a power-of-two factor (to the half subnormals)."""
    from gridencoder import grid as G
    L = 16
    offsets, pls = oracle.grid_offsets(3, L, 2, 2, 16, 19, 4096, False)
    B = 4096 * 512
    x = ray_points(4096, 512, 11)
    rng = np.random.default_rng(12)
    grad = (rng.normal(size=(L, B, 2)) * 0.02).astype(np.float16)
    tg, tx, to = t(grad, dev), t(x, dev), t(offsets, dev)
    out = G.table_gradient_binned(tg, tx, to, B, L, np.log2(pls), 16, 0, False)
    assert torch.equal(out, G.table_gradient_binned(tg, tx, to, B, L, np.log2(pls), 16, 0, False))      # run-to-run identical
    inside = ((tx >= 0) & (tx <= 1)).all(dim=1)
    want = (tg.float() * inside[None, :, None]).double().sum(dim=1).cpu().numpy()                        # [L, 2]
    mass = (tg.float().abs() * inside[None, :, None]).double().sum(dim=1).cpu().numpy()
    off = offsets.astype(np.int64)
    got = np.stack([out[off[l]:off[l + 1]].double().sum(dim=0).cpu().numpy() for l in range(L)])
    # each of the 8 products of a sample is rounded to half (2^-11 relative); the rounding errors of 16 M products of random sign mostly cancel
    assert np.all(np.abs(got - want) <= 1e-4 * mass + 1e-6), (np.abs(got - want) / mass).max()
    assert (np.abs(want) > 1e-3 * mass).any()                                                            # the checksum is not trivially zero
    twice = G.table_gradient_binned((tg * 2).contiguous(), tx, to, B, L, np.log2(pls), 16, 0, False)
    # scaling by 2 commutes with every rounding in the path except where a product falls into the half subnormals (absolute spacing 2^-24)
    assert float((twice - out * 2).abs().max()) <= 1e-4 * float(out.abs().max())


@pytest.mark.parametrize("count", [0, 1, 1023, 1024, 20000, 70001])
def test_listed_scatter_equals_the_scatter_of_the_listed_samples(oracle, dev, count):
    """ngp_grid_scatter_binned_listed: gradient row i belongs to the sample at inputs[list[i]], i < *count (both on the device; the field's training
    backward lists the samples that got a gradient).  Scattering the first `count` list entries must give the table the unlisted entry point gives for
    those samples gathered by hand -- exactly when the list keeps neighbours together (the run sums are then the same), and rows from `count` on are
    neither read (they hold NaN here) nor binned; count = 0 writes a table of zeros."""
    import ctypes
    from gridencoder import grid as G
    B, L = 70001, 16
    offsets, pls = oracle.grid_offsets(3, L, 2, 2, 16, 19, 4096, False)
    x = ray_points(B // 50 + 1, 50, 5)[:B]
    rng = np.random.default_rng(6)
    order = np.sort(rng.choice(B, size=B, replace=False)[:max(count, 1)]).astype(np.int32)      # an increasing list, like the ordered compaction's
    order = np.concatenate([order, np.zeros(B - len(order), np.int32)])
    grad = (rng.normal(size=(L, B, 2)) * 0.1).astype(np.float16)
    grad[:, count:] = np.nan
    lst, cnt = t(order, dev), torch.tensor([count], dtype=torch.int32, device=dev)
    listed = (ctypes.c_void_p(lst.data_ptr()), ctypes.c_void_p(cnt.data_ptr()))
    got = G.table_gradient_binned(t(grad, dev), t(x, dev), t(offsets, dev), B, L, np.log2(pls), 16, 0, False, listed=listed)
    assert bool(torch.isfinite(got).all())
    if count == 0:
        assert not bool(got.any())
        return
    sub_x = np.ascontiguousarray(x[order[:count]])
    sub_g = np.ascontiguousarray(grad[:, :count])
    want = G.table_gradient_binned(t(sub_g, dev), t(sub_x, dev), t(offsets, dev), count, L, np.log2(pls), 16, 0, False)
    assert torch.equal(got, want)
