"""GPU parity for the fused MLP (v_mfma_f32_16x16x32_f16) against the CPU oracle.
 * exact-integer data proves the MFMA fragment maps and the permuted-k weight layout (every product and sum is exactly
   representable, so any misplaced element changes the result);
 * random half data: tolerance.  The reference accumulates in half inside WMMA, the oracle in double, the kernel in
   binary32 inside the MFMA, each rounding the layer output to half once: results may differ by one half ulp after a
   layer, which later layers amplify -- bound stated below."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def run_hip(dev, x, w, input_dim, num_layers, save):
    import ngp_hip
    B = x.shape[0]
    out = torch.empty(B, 16, dtype=torch.float16, device=dev)
    fb = torch.empty(num_layers, B, 64, dtype=torch.float16, device=dev) if save else None
    L = ngp_hip.lib()
    fn = L.ngp_ffmlp_forward if save else L.ngp_ffmlp_inference
    ngp_hip.check(fn(ngp_hip.ptr(t(x, dev)), ngp_hip.ptr(t(w, dev)), B, input_dim, 16, 64, num_layers, 0, 6, ngp_hip.ptr(fb),
                     ngp_hip.ptr(out), ngp_hip.stream()))
    return out.cpu().numpy(), (fb.cpu().numpy() if save else None)


@pytest.mark.parametrize("input_dim,num_layers", [(32, 2), (32, 3), (64, 2), (16, 2), (48, 3), (32, 4)])
def test_ffmlp_exact_integer_data(oracle, dev, input_dim, num_layers):
    rng = np.random.default_rng(input_dim * 10 + num_layers)
    B = 16 * 37
    nw = oracle.ffmlp_num_params(input_dim, 16, 64, num_layers)
    # weights in {-1, 0, 1} (sparse), inputs small non-negative integers: all activations stay small integers, exact in half
    w = rng.choice([-1.0, 0, 0, 0, 0, 0, 0, 1.0], size=nw).astype(np.float16)
    x = rng.integers(0, 3, size=(B, input_dim)).astype(np.float16)
    ref, fb_ref = oracle.ffmlp_forward(x, w, input_dim, 16, 64, num_layers, save=True)
    assert np.abs(ref.astype(np.float32)).max() < 2048 and np.abs(ref.astype(np.float32)).max() > 3
    got, fb = run_hip(dev, x, w, input_dim, num_layers, True)
    assert np.array_equal(fb.view(np.uint16), fb_ref.view(np.uint16)), "forward_buffer"
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), "outputs"
    got_inf, _ = run_hip(dev, x, w, input_dim, num_layers, False)
    assert np.array_equal(got_inf.view(np.uint16), ref.view(np.uint16)), "inference outputs"


@pytest.mark.parametrize("input_dim,num_layers", [(32, 2), (32, 3)])
def test_ffmlp_random_data_tolerance(oracle, dev, input_dim, num_layers):
    rng = np.random.default_rng(1)
    B = 128 * 9
    nw = oracle.ffmlp_num_params(input_dim, 16, 64, num_layers)
    std = np.sqrt(3 / 64)
    w = rng.uniform(-std, std, size=nw).astype(np.float16)
    x = rng.normal(size=(B, input_dim)).astype(np.float16)
    ref, fb_ref = oracle.ffmlp_forward(x, w, input_dim, 16, 64, num_layers, save=True)
    got, fb = run_hip(dev, x, w, input_dim, num_layers, True)
    # first hidden layer: same exact products, sums differ only in accumulation order => at most 1 half ulp
    a, b = fb[0].astype(np.float32), fb_ref[0].astype(np.float32)
    assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 2.0 ** -10)) <= 2.0 ** -10
    # outputs: a half ulp per layer amplified by |W| <= sqrt(3/64) rows of 64
    scale = np.abs(ref.astype(np.float32)).max()
    assert np.max(np.abs(got.astype(np.float32) - ref.astype(np.float32))) <= 4e-3 * scale
    # most outputs agree to the bit
    assert (got.view(np.uint16) == ref.view(np.uint16)).mean() > 0.9


def test_ffmlp_module_pads_like_reference(oracle, dev):
    from ffmlp import FFMLP
    net = FFMLP(32, 16, 64, 2).to(dev).eval()
    w = net.weights.detach().cpu().numpy().astype(np.float16)
    for B in (100, 128, 1000):
        x = np.random.default_rng(B).normal(size=(B, 32)).astype(np.float32)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            y = net(t(x, dev))
        assert y.shape == (B, 16) and y.dtype == torch.float16
        ref, _ = oracle.ffmlp_forward(x.astype(np.float16), w, 32, 16, 64, 2)
        assert np.max(np.abs(y.cpu().numpy().astype(np.float32) - ref.astype(np.float32))) <= 4e-3 * np.abs(ref.astype(np.float32)).max()
    with pytest.raises(RuntimeError):                      # without autocast the inputs are float: rejected like CHECK_IS_HALF
        net(t(x, dev))


def test_ffmlp_rejects_unsupported(dev):
    """what the reference's module refuses too (ffmlp.cu:659, ffmlp.py:110-113): a width outside [16, 32, 64, 128, 256], an output activation"""
    import ngp_hip
    z = torch.zeros(1 << 16, dtype=torch.float16, device=dev)
    rc = ngp_hip.lib().ngp_ffmlp_inference(ngp_hip.ptr(z), ngp_hip.ptr(z), 16, 32, 16, 96, 2, 0, 6, ngp_hip.ptr(z), ngp_hip.ptr(z), ngp_hip.stream())
    assert rc == -1 and b"hidden_dim" in ngp_hip.lib().ngp_last_error()
    rc = ngp_hip.lib().ngp_ffmlp_inference(ngp_hip.ptr(z), ngp_hip.ptr(z), 16, 32, 16, 64, 2, 0, 3, ngp_hip.ptr(z), ngp_hip.ptr(z), ngp_hip.stream())
    assert rc == -1 and b"output activation" in ngp_hip.lib().ngp_last_error()


def run_hip_backward(dev, g, x, w, fb, input_dim, num_layers, calc):
    import ngp_hip
    B = x.shape[0]
    L = ngp_hip.lib()
    tg, tx, tw, tfb = t(g, dev), t(x, dev), t(w, dev), t(fb, dev)
    bb = torch.zeros(num_layers, B, 64, dtype=torch.float16, device=dev)
    gi = torch.zeros(B, input_dim, dtype=torch.float16, device=dev)
    gw = torch.zeros(w.shape[0], dtype=torch.float16, device=dev)
    ws = ngp_hip.workspace(L.ngp_ffmlp_backward_workspace(input_dim, 16, 64, num_layers), dev)
    ngp_hip.check(L.ngp_ffmlp_backward(ngp_hip.ptr(tg), ngp_hip.ptr(tx), ngp_hip.ptr(tw), ngp_hip.ptr(tfb), B, input_dim, 16, 64, num_layers,
                                       0, 6, int(calc), ngp_hip.ptr(bb), ngp_hip.ptr(gi), ngp_hip.ptr(gw), ngp_hip.ptr(ws), ws.numel(),
                                       ngp_hip.stream()))
    return gw.cpu().numpy(), gi.cpu().numpy(), bb.cpu().numpy()


@pytest.mark.parametrize("input_dim,num_layers", [(32, 2), (32, 3), (64, 2), (16, 2), (48, 4)])
def test_ffmlp_backward_exact_integer_data(oracle, dev, input_dim, num_layers):
    """small-integer data: every product and partial sum is exact in half/f32, so the transposed weight fragments, the
    ReLU masks, the LDS transposes of the weight-gradient GEMMs and the atomics must reproduce the oracle exactly"""
    rng = np.random.default_rng(100 + input_dim + num_layers)
    B = 128 if num_layers < 4 else 32                          # keep every gradient below 2048 (exact in half)
    nw = oracle.ffmlp_num_params(input_dim, 16, 64, num_layers)
    w = rng.choice([-1.0, 0, 0, 0, 0, 0, 0, 1.0], size=nw).astype(np.float16)
    x = rng.integers(0, 2, size=(B, input_dim)).astype(np.float16)
    g = rng.choice([-1.0, 0, 0, 1.0], size=(B, 16)).astype(np.float16)
    _, fb = oracle.ffmlp_forward(x, w, input_dim, 16, 64, num_layers, save=True)
    gw_ref, gi_ref, bb_ref = oracle.ffmlp_backward(g, x, w, fb, input_dim, 16, 64, num_layers, True)
    assert np.abs(gw_ref).max() < 2048 and np.abs(gi_ref).max() < 2048 and np.abs(bb_ref.astype(np.float32)).max() < 2048
    assert np.abs(gw_ref).max() > 8 and (bb_ref != 0).mean() > 0.05
    gw, gi, bb = run_hip_backward(dev, g, x, w, fb, input_dim, num_layers, True)
    assert np.array_equal(bb.view(np.uint16), bb_ref.view(np.uint16)), "backward_buffer"
    assert np.array_equal(gi.astype(np.float32), gi_ref), "grad_inputs"
    assert np.array_equal(gw.astype(np.float32), gw_ref), "grad_weights"


def test_ffmlp_backward_random_data_tolerance(oracle, dev):
    rng = np.random.default_rng(7)
    input_dim, num_layers, B = 32, 3, 128 * 40
    nw = oracle.ffmlp_num_params(input_dim, 16, 64, num_layers)
    w = rng.uniform(-np.sqrt(3 / 64), np.sqrt(3 / 64), size=nw).astype(np.float16)
    x = rng.normal(size=(B, input_dim)).astype(np.float16)
    g = (rng.normal(size=(B, 16)) * 1e-2).astype(np.float16)
    _, fb = oracle.ffmlp_forward(x, w, input_dim, 16, 64, num_layers, save=True)
    gw_ref, gi_ref, bb_ref = oracle.ffmlp_backward(g, x, w, fb, input_dim, 16, 64, num_layers, True)
    gw, gi, bb = run_hip_backward(dev, g, x, w, fb, input_dim, num_layers, True)
    # activation gradients are halves: one half ulp per layer, amplified by the next layers
    s = np.abs(bb_ref.astype(np.float32)).max()
    assert np.max(np.abs(bb.astype(np.float32) - bb_ref.astype(np.float32))) < 4e-3 * s
    assert np.max(np.abs(gi.astype(np.float32) - gi_ref)) < 6e-3 * np.abs(gi_ref).max()
    # weight gradients: f32 MFMA accumulation over 5120 samples, rounded to half once (the reference accumulates in half)
    assert np.max(np.abs(gw.astype(np.float32) - gw_ref)) < 4e-3 * np.abs(gw_ref).max()


def test_ffmlp_module_autograd(oracle, dev):
    from ffmlp import FFMLP
    net = FFMLP(32, 16, 64, 2).to(dev).train()
    w = net.weights.detach().cpu().numpy().astype(np.float16)
    rng = np.random.default_rng(3)
    B = 300                                                    # padded to 320 inside FFMLP.forward (zero rows change nothing)
    x = rng.normal(size=(B, 32)).astype(np.float32)
    g = (rng.normal(size=(B, 16)) * 1e-2).astype(np.float32)
    xt = t(x, dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(xt)
    y.backward(t(g, dev).half())
    assert net.weights.grad.dtype == torch.float32 and xt.grad.shape == (B, 32)
    xp = np.concatenate([x.astype(np.float16), np.zeros((20, 32), np.float16)])
    gp = np.concatenate([g.astype(np.float16), np.zeros((20, 16), np.float16)])
    _, fb = oracle.ffmlp_forward(xp, w, 32, 16, 64, 2, save=True)
    gw_ref, gi_ref, _ = oracle.ffmlp_backward(gp, xp, w, fb, 32, 16, 64, 2, True)
    assert np.max(np.abs(net.weights.grad.cpu().numpy() - gw_ref)) < 4e-3 * np.abs(gw_ref).max()
    assert np.max(np.abs(xt.grad.cpu().numpy() - gi_ref[:B])) < 6e-3 * np.abs(gi_ref).max()
