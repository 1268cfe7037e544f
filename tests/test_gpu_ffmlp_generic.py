"""GPU parity of the layer-by-layer FFMLP (csrc/ffmlp_generic.hip) -- every width, depth and activation the reference's module accepts
(ffmlp/ffmlp.py:100-117, ffmlp/src/ffmlp.cu:652-659, ffmlp/src/utils.h:423-590) beyond the 64-wide ReLU networks of its models -- against the
CPU oracle (oracle/ngp_oracle.c: o_ffmlp_forward_act / o_ffmlp_backward_act).
 * ReLU / none on small-integer data: exact (every product and sum is representable, so a misplaced fragment element changes the result);
 * the transcendental activations on random data: one half ulp per layer (expf / logf / sinf of the device against libm), amplified by later layers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ACT = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5, "none": 6}


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def hip_forward(dev, x, w, input_dim, hidden, num_layers, act, save):
    import ngp_hip
    B = x.shape[0]
    out = torch.empty(B, 16, dtype=torch.float16, device=dev)
    fb = torch.empty(num_layers, B, hidden, dtype=torch.float16, device=dev)
    L = ngp_hip.lib()
    fn = L.ngp_ffmlp_forward if save else L.ngp_ffmlp_inference
    ngp_hip.check(fn(ngp_hip.ptr(t(x, dev)), ngp_hip.ptr(t(w, dev)), B, input_dim, 16, hidden, num_layers, act, 6, ngp_hip.ptr(fb),
                     ngp_hip.ptr(out), ngp_hip.stream()))
    return out.cpu().numpy(), fb.cpu().numpy()


def hip_backward(dev, g, x, w, fb, input_dim, hidden, num_layers, act, calc=True):
    import ngp_hip
    B = x.shape[0]
    L = ngp_hip.lib()
    tg, tx, tw, tfb = t(g, dev), t(x, dev), t(w, dev), t(fb, dev)
    bb = torch.zeros(num_layers, B, hidden, dtype=torch.float16, device=dev)
    gi = torch.zeros(B, input_dim, dtype=torch.float16, device=dev)
    gw = torch.zeros(w.shape[0], dtype=torch.float16, device=dev)
    ws = ngp_hip.workspace(L.ngp_ffmlp_backward_workspace(input_dim, 16, hidden, num_layers), dev)
    rc = L.ngp_ffmlp_backward(ngp_hip.ptr(tg), ngp_hip.ptr(tx), ngp_hip.ptr(tw), ngp_hip.ptr(tfb), B, input_dim, 16, hidden, num_layers,
                              act, 6, int(calc), ngp_hip.ptr(bb), ngp_hip.ptr(gi), ngp_hip.ptr(gw), ngp_hip.ptr(ws), ws.numel(), ngp_hip.stream())
    return rc, gw.cpu().numpy(), gi.cpu().numpy(), bb.cpu().numpy()


SHAPES = [(16, 16, 2), (32, 32, 3), (48, 128, 2), (96, 64, 5), (32, 256, 2), (256, 128, 3), (80, 32, 4), (64, 64, 6)]     # (input_dim, hidden, num_layers)


@pytest.mark.parametrize("act", ["relu", "none"])
@pytest.mark.parametrize("input_dim,hidden,num_layers", SHAPES)
def test_generic_ffmlp_exact_integer_data(oracle, dev, input_dim, hidden, num_layers, act):
    if act == "relu" and hidden == 64 and input_dim <= 64 and num_layers <= 4:
        pytest.skip("the register-resident path (tests/test_gpu_ffmlp.py)")
    rng = np.random.default_rng(hidden + input_dim + num_layers)
    B = 16 * 11 + 5                                                       # not a multiple of 16: the kernels guard the last tile
    nw = oracle.ffmlp_num_params(input_dim, 16, hidden, num_layers)
    x = rng.integers(0, 2, size=(B, input_dim)).astype(np.float16)
    g = rng.choice([-1.0, 0, 0, 0, 1.0], size=(B, 16)).astype(np.float16)
    # weights in {-1, 0, 1}, sparse enough that every activation, gradient and weight gradient stays a small integer (exact in half and binary32)
    for zeros in (6, 14, 30, 62, 126, 254, 510):
        w = rng.choice([-1.0] + [0.0] * zeros + [1.0], size=nw).astype(np.float16)
        ref, fb_ref = oracle.ffmlp_forward(x, w, input_dim, 16, hidden, num_layers, save=True, activation=ACT[act])
        gw_ref, gi_ref, bb_ref = oracle.ffmlp_backward(g, x, w, fb_ref, input_dim, 16, hidden, num_layers, True, activation=ACT[act])
        if max(np.abs(fb_ref.astype(np.float32)).max(), np.abs(gw_ref).max(), np.abs(gi_ref).max(), np.abs(bb_ref.astype(np.float32)).max()) < 2048:
            break
    assert np.abs(ref.astype(np.float32)).max() > 0 and np.abs(gw_ref).max() >= 2 and (bb_ref != 0).any()
    for save in (True, False):
        got, fb = hip_forward(dev, x, w, input_dim, hidden, num_layers, ACT[act], save)
        assert np.array_equal(fb.astype(np.float32), fb_ref.astype(np.float32)), "forward / inference buffer"
        assert np.array_equal(got.astype(np.float32), ref.astype(np.float32)), "outputs"
    rc, gw, gi, bb = hip_backward(dev, g, x, w, fb_ref, input_dim, hidden, num_layers, ACT[act])
    assert rc == 0
    assert np.array_equal(bb.astype(np.float32), bb_ref.astype(np.float32)), "backward_buffer"
    assert np.array_equal(gi.astype(np.float32), gi_ref), "grad_inputs"
    assert np.array_equal(gw.astype(np.float32), gw_ref), "grad_weights"


@pytest.mark.parametrize("act", ["exponential", "sigmoid", "squareplus", "softplus", "sine", "relu"])
@pytest.mark.parametrize("input_dim,hidden,num_layers", [(32, 128, 2), (48, 32, 3), (16, 256, 2), (32, 64, 3)])
def test_generic_ffmlp_activations_random_data(oracle, dev, input_dim, hidden, num_layers, act):
    if act == "relu" and hidden == 64:
        pytest.skip("the register-resident path")
    rng = np.random.default_rng(3)
    B = 128 * 5
    nw = oracle.ffmlp_num_params(input_dim, 16, hidden, num_layers)
    std = np.sqrt(3 / hidden) * (0.5 if act == "exponential" else 1.0)
    w = rng.uniform(-std, std, size=nw).astype(np.float16)
    x = rng.normal(size=(B, input_dim)).astype(np.float16) * np.float16(0.5)
    ref, fb_ref = oracle.ffmlp_forward(x, w, input_dim, 16, hidden, num_layers, save=True, activation=ACT[act])
    got, fb = hip_forward(dev, x, w, input_dim, hidden, num_layers, ACT[act], True)
    a, b = fb[0].astype(np.float32), fb_ref[0].astype(np.float32)
    assert np.isfinite(b).all()
    # first hidden layer: same exact products; the pre-activation may differ by one half ulp (accumulation order), the activation is 1-Lipschitz-ish
    assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 2.0 ** -8)) <= 2.0 ** -8, "first hidden layer"
    scale = np.abs(ref.astype(np.float32)).max()
    assert np.max(np.abs(got.astype(np.float32) - ref.astype(np.float32))) <= 8e-3 * scale, "outputs"
    assert (got.view(np.uint16) == ref.view(np.uint16)).mean() > 0.7

    g = (rng.normal(size=(B, 16)) * 1e-2).astype(np.float16)
    rc, gw, gi, bb = hip_backward(dev, g, x, w, fb_ref, input_dim, hidden, num_layers, ACT[act])
    if act == "sine":                                                     # no backward in the reference either (utils.h:552-556)
        import ngp_hip
        assert rc != 0 and b"Sine" in ngp_hip.lib().ngp_last_error()
        return
    assert rc == 0
    gw_ref, gi_ref, bb_ref = oracle.ffmlp_backward(g, x, w, fb_ref, input_dim, 16, hidden, num_layers, True, activation=ACT[act])
    s = np.abs(bb_ref.astype(np.float32)).max()
    assert np.max(np.abs(bb.astype(np.float32) - bb_ref.astype(np.float32))) < 8e-3 * s, "backward_buffer"
    assert np.max(np.abs(gi.astype(np.float32) - gi_ref)) < 1e-2 * np.abs(gi_ref).max(), "grad_inputs"
    assert np.max(np.abs(gw.astype(np.float32) - gw_ref)) < 8e-3 * np.abs(gw_ref).max(), "grad_weights"


def test_generic_ffmlp_module_trains(dev):
    """the drop-in module with a shape outside the reference's models: FFMLP(48, 3, 128, 3, 'softplus') under autocast, forward + backward + a descent step"""
    from ffmlp import FFMLP
    net = FFMLP(48, 3, 128, 3, activation="softplus").to(dev)
    x = torch.randn(1000, 48, device=dev, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
        assert y.shape == (1000, 3) and y.dtype == torch.float16
        loss = (y.float() ** 2).mean()
    loss.backward()
    assert net.weights.grad is not None and torch.isfinite(net.weights.grad).all() and float(net.weights.grad.abs().max()) > 0
    assert x.grad is not None and x.grad.shape == (1000, 48) and torch.isfinite(x.grad).all()
    before = float(loss.detach())
    with torch.no_grad():                                                # a short step against the gradient must lower the loss
        net.weights -= 0.05 * net.weights.grad / net.weights.grad.norm()
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        after = float((net(x.detach()).float() ** 2).mean())
    assert after < before
