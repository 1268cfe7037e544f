"""GPU: the field's training step in native launches (ngp/field.py `_field_train`; csrc/field_train.hip) against the
op-by-op autograd graph of the same module (GridEncoder -> FFMLP -> trunc_exp ; SH ++ geo -> FFMLP -> sigmoid; nerf/network_ff.py:51-77),
which tests/test_gpu_callers_parity.py in turn pins against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _field(dev):
    from ngp import workload as W
    from ngp.field import NGPFieldFF
    return NGPFieldFF(bound=W.BOUND).to(dev).load_arrays(W.make_model(0)), W


def _points(W, M, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * W.BOUND
    x[::97] *= 1.3                                                     # a few samples outside the box: they encode to zeros
    d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
    return x.to(dev), d.to(dev), torch.randn(M, generator=g).to(dev), torch.randn(M, 3, generator=g).to(dev)


def _step(field, fused, x, d, gs, gc, scale):
    field.fused_training = fused
    for p in field.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.float16):
        sig, rgb = field(x, d)
        loss = ((sig * gs).sum() + (rgb.float() * gc).sum()) * scale
    loss.backward()
    return (sig.detach().float(), rgb.detach().float(), field.encoder.embeddings.grad.clone(), field.sigma_net.weights.grad.clone(),
            field.color_net.weights.grad.clone())


@pytest.mark.parametrize("M", [4096, 5000, 37])
def test_fused_training_step_equals_the_op_graph(dev, M):
    field, W = _field(dev)
    field.train()
    x, d, gs, gc = _points(W, M, dev, seed=M)
    ref = _step(field, False, x, d, gs, gc, 1.0)            # (a larger loss scale overflows the half gradients of BOTH paths on this model)
    got = _step(field, True, x, d, gs, gc, 1.0)
    # forward: the same half logits; exp / sigmoid are the library's own float32 routines (not torch's): a couple of ulps, and at most
    # one half ulp after the sigmoid's rounding
    assert float(((got[0] - ref[0]).abs() / ref[0]).max()) < 1e-6, "sigma"
    assert float((got[1] - ref[1]).abs().max()) <= 2.0 ** -11 and float((got[1] != ref[1]).float().mean()) < 0.01, "rgb"
    names = ["table", "density-net weights", "colour-net weights"]
    for name, a, b in zip(names, got[2:], ref[2:]):
        assert a.shape == b.shape and a.dtype == b.dtype
        scale = float(b.abs().max())
        assert scale > 0 and bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all()), name
        # same half roundings along the chain; the sums differ in order only (float atomics / one f32 sum rounded to half once)
        err = float((a - b).abs().max())
        assert err <= 1e-3 * scale, (name, err, scale)
        assert float((a - b).abs().mean()) <= 1e-3 * float(b.abs().mean()) + 1e-9, name                     # (a half ulp is 5e-4 relative)


@pytest.mark.parametrize("M", [5000, 150001])
def test_fused_training_step_with_samples_that_get_no_gradient(dev, M):
    """the backward works on the samples with a non-zero incoming gradient only (k_ft_live_count / k_ft_live_write: behind the compositor's early exit
    half of a converged batch gets none): runs of dead samples of every length and alignment, a dead tail, one live sample among dead ones"""
    field, W = _field(dev)
    field.train()
    x, d, gs, gc = _points(W, M, dev, seed=M)
    rng = np.random.default_rng(M)
    dead = np.zeros(M, bool)
    i = 0
    while i < M:
        run = int(rng.integers(1, 200))
        if rng.random() < 0.5:
            dead[i:i + run] = True
        i += run
    dead[-M // 5:] = True
    dead[-7] = False
    dead_t = torch.from_numpy(dead).to(dev)
    gs, gc = gs.masked_fill(dead_t, 0.0), gc.masked_fill(dead_t[:, None], 0.0)
    assert 0.5 < dead.mean() < 0.8
    import ngp_hip
    ref = _step(field, False, x, d, gs, gc, 1.0)
    got = _step(field, True, x, d, gs, gc, 1.0)
    previous = ngp_hip.lib().ngp_field_train_set_live_only(0)
    try:
        every = _step(field, True, x, d, gs, gc, 1.0)             # the same kernels over every sample
    finally:
        ngp_hip.lib().ngp_field_train_set_live_only(previous)
    # (a sample's feature gradient does not depend on which samples share its tile and the table sums are exact, but the scatter first adds up each run of
    # consecutive samples in one cell in float32 and rounds it to half, and runs end at wave boundaries -- which fall elsewhere in the list: half ulps)
    for name, a, b, c in zip(["table", "density-net weights", "colour-net weights"], got[2:], ref[2:], every[2:]):
        scale = float(b.abs().max())
        assert scale > 0 and bool(torch.isfinite(a).all()), name
        assert float((a - c).abs().max()) <= 5e-4 * scale, name      # the weight sums in another order, rounded to half once: at most a half ulp apart
        # against the op graph, as above (its own sums run in yet another order and its partial sums are rounded more often: 2e-3 at 150,000 samples)
        assert float((a - b).abs().max()) <= 2e-3 * scale, name
        assert float((a - b).abs().mean()) <= 1e-3 * float(b.abs().mean()) + 1e-9, name


@pytest.mark.parametrize("M", [1, 63, 4096, 70001])
def test_fused_training_step_when_no_sample_gets_a_gradient(dev, M):
    """an empty live list (every incoming gradient +0, e.g. a batch of rays that hit nothing): zero gradients everywhere, written (not left as they were)"""
    field, W = _field(dev)
    field.train()
    x, d, gs, gc = _points(W, M, dev, seed=M)
    got = _step(field, True, x, d, torch.zeros_like(gs), torch.zeros_like(gc), 1.0)
    for g in got[2:]:
        assert bool(torch.isfinite(g).all()) and not bool(g.any())
    # ... and ONE live sample among M: only its 16 x 8 table rows (at most) and the weights get a gradient
    one = torch.zeros_like(gs)
    one[M // 2] = 1.0
    got = _step(field, True, x, d, one, torch.zeros_like(gc), 1.0)
    ref = _step(field, False, x, d, one, torch.zeros_like(gc), 1.0)
    assert 0 < int((got[2].abs().sum(-1) > 0).sum()) <= 128
    for a, b in zip(got[2:4], ref[2:4]):
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-12


def test_fused_training_is_used_only_where_it_applies(dev):
    field, W = _field(dev)
    x, d, gs, gc = _points(W, 256, dev, seed=1)
    field.train()
    assert field._fused_training_applies(x, d) is False                # no autocast: the reference's FFMLP refuses float32 anyway
    with torch.autocast("cuda", dtype=torch.float16):
        assert field._fused_training_applies(x, d)
        with torch.no_grad():
            assert not field._fused_training_applies(x, d)             # density-grid refresh, evaluation
        assert not field._fused_training_applies(x.clone().requires_grad_(True), d)    # gradients to the points: the op graph
        field.encoder.embeddings.requires_grad_(False)
        assert not field._fused_training_applies(x, d)                 # frozen table


def test_fused_training_with_an_empty_batch_and_a_non_power_of_two_bound(dev):
    """M = 0 (a batch whose rays all miss the occupied cells) gives empty outputs and zero gradients; bound 1.5 takes the division path of the
    normalisation (2 * bound is not a power of two) and must still agree with the op graph."""
    from ngp.field import NGPFieldFF
    field, W = _field(dev)
    field.train()
    x, d = torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev)
    with torch.autocast("cuda", dtype=torch.float16):
        sig, rgb = field(x, d)
        assert sig.shape == (0,) and rgb.shape == (0, 3)
        (sig.sum() + rgb.float().sum()).backward()
    for p in (field.encoder.embeddings, field.sigma_net.weights, field.color_net.weights):
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0

    torch.manual_seed(0)
    odd = NGPFieldFF(bound=1.5).to(dev).train()                          # (its table has another size than the workload's: seeded random values)
    with torch.no_grad():
        odd.encoder.embeddings.uniform_(-0.5, 0.5)
    xs, ds, gs, gc = _points(W, 2048, dev, seed=3)
    xs = xs * (1.5 / W.BOUND)
    ref = _step(odd, False, xs, ds, gs, gc, 1.0)
    got = _step(odd, True, xs, ds, gs, gc, 1.0)
    # (the encoded features of the two paths are the same bits at every bound; the first layer of the fused density net sums its 32 products in the
    #  encoder's level-interleaved order, the op graph in natural order: with this seeded-random table (|features| up to 0.5, not the workload's small
    #  ones) a few pre-activations land on the other side of a half rounding step)
    assert float(((got[0] - ref[0]).abs() / ref[0]).max()) < 2e-3 and float((got[0] != ref[0]).float().mean()) < 0.05
    for a, b in zip(got[2:], ref[2:]):
        assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max())


@pytest.mark.parametrize("M", [70000, 200001])
def test_two_pass_forward_equals_the_one_launch_forward_bit_for_bit(dev, M):
    """the training forward encodes level by level (k_ft_encode_levels: one level's table live in L2 at a time) and runs the networks
    in a second pass.  Same arithmetic per (sample, level) as the fused gather: sigma, rgb, the kept features (through the gradients they produce) and
    every gradient are the same BITS as with the single launch -- and the whole step is bitwise reproducible (no atomics: exact table sums, fixed-order
    weight-gradient sums)."""
    import ngp_hip
    field, W = _field(dev)
    field.train()
    x, d, gs, gc = _points(W, M, dev, seed=M)
    x[::3] *= 0.2                                                      # a third of the points near the centre: the dense and the hashed levels see runs and collisions
    L = ngp_hip.lib()
    try:
        assert L.ngp_field_train_set_two_pass(0) == 1                  # (two passes are the default)
        one = _step(field, True, x, d, gs, gc, 1.0)
        L.ngp_field_train_set_two_pass(1)
        two = _step(field, True, x, d, gs, gc, 1.0)
        again = _step(field, True, x, d, gs, gc, 1.0)
    finally:
        L.ngp_field_train_set_two_pass(1)
    for name, a, b, c in zip(["sigma", "rgb", "table gradient", "density-net weight gradient", "colour-net weight gradient"], one, two, again):
        assert torch.equal(a, b), name
        assert torch.equal(b, c), name + " (second run)"
    assert float(two[2].abs().max()) > 0 and float(two[3].abs().max()) > 0
