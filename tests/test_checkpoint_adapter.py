"""Checkpoint compatibility: nn.Linear field <-> flat FFMLP weights, reference state-dict key layout."""
import importlib
import io

import numpy as np
import pytest
import torch
import torch.nn.functional as F

importlib.import_module("nerf-navigation_amd")
from ngp import checkpoint as CK  # noqa: E402


def _ffmlp_reference(x, flat, input_dim, hidden, num_layers, out_pad=16):
    """What FFMLP computes from its flat weights (ffmlp.cu:631-634 layout): num_layers + 1 matmuls, ReLU between them."""
    off = 0
    w = flat[off:off + hidden * input_dim].view(hidden, input_dim); off += hidden * input_dim
    h = F.relu(x @ w.t())
    for _ in range(num_layers - 1):
        w = flat[off:off + hidden * hidden].view(hidden, hidden); off += hidden * hidden
        h = F.relu(h @ w.t())
    w = flat[off:off + out_pad * hidden].view(out_pad, hidden); off += out_pad * hidden
    assert off == flat.numel()
    return h @ w.t()


def test_linear_to_ffmlp_is_the_same_function():
    g = torch.Generator().manual_seed(0)
    W = [torch.randn(64, 31, generator=g, dtype=torch.float64), torch.randn(64, 64, generator=g, dtype=torch.float64),
         torch.randn(3, 64, generator=g, dtype=torch.float64)]
    flat, n = CK.linear_to_ffmlp([w.float() for w in W], 32)
    assert n == 3 and flat.numel() == 64 * (32 + 64 * 2 + 16)                       # the colour FFMLP of network_ff.py:42-49
    x31 = torch.randn(257, 31, generator=g, dtype=torch.float64)
    want = F.relu(F.relu(x31 @ W[0].float().double().t()) @ W[1].float().double().t()) @ W[2].float().double().t()
    x32 = torch.cat([x31, torch.randn(257, 1, generator=g, dtype=torch.float64)], 1)  # whatever sits in the padded column is ignored
    got = _ffmlp_reference(x32, flat.double(), 32, 64, n)
    # the identity layer and the zero padding add exact zeros; only the BLAS summation order over K = 31 vs 32 may differ
    assert torch.allclose(got[:, :3], want, rtol=1e-12, atol=1e-10) and float(got[:, 3:].abs().max()) == 0.0
    flat2, n2 = CK.linear_to_ffmlp([torch.randn(64, 32, generator=g), torch.randn(16, 64, generator=g)], 32)
    assert n2 == 2 and flat2.numel() == 64 * (32 + 64 + 16)                          # the density FFMLP of network_ff.py:31-38


def test_reference_state_dict_split_and_file_round_trip(tmp_path):
    g = torch.Generator().manual_seed(1)
    sd = {
        "aabb_train": torch.tensor([-2., -2, -2, 2, 2, 2]), "aabb_infer": torch.tensor([-2., -2, -2, 2, 2, 2]),
        "density_grid": torch.rand(2, 8, generator=g), "density_bitfield": torch.randint(0, 255, (2,), dtype=torch.uint8, generator=g),
        "step_counter": torch.zeros(16, 2, dtype=torch.int32),
        "encoder.embeddings": torch.rand(64, 2, generator=g), "encoder.offsets": torch.arange(17, dtype=torch.int32),
        "sigma_net.0.weight": torch.randn(64, 32, generator=g), "sigma_net.1.weight": torch.randn(16, 64, generator=g),
        "color_net.0.weight": torch.randn(64, 31, generator=g), "color_net.1.weight": torch.randn(64, 64, generator=g),
        "color_net.2.weight": torch.randn(3, 64, generator=g),
    }
    path = tmp_path / "ngp_ep0001.pth"
    torch.save({"model": sd, "epoch": 1}, path)                                       # the layout Trainer.save_checkpoint writes
    back = CK.read_checkpoint(str(path))
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    field, ren = CK.split_state_dict(back)
    assert set(ren) == set(CK.RENDERER_KEYS) and not CK.is_ff_state_dict(field)
    ff = CK.ff_state_dict_from_linear(field)
    assert CK.is_ff_state_dict(ff) and ff["sigma_net.weights"].numel() == 64 * 112 and ff["color_net.weights"].numel() == 64 * 176
    assert torch.equal(ff["encoder.embeddings"], sd["encoder.embeddings"])
    # density net: [W0 | I | W1]
    s = ff["sigma_net.weights"]
    assert torch.equal(s[:2048].view(64, 32), sd["sigma_net.0.weight"]) and torch.equal(s[2048:2048 + 4096].view(64, 64), torch.eye(64))
    assert torch.equal(s[6144:].view(16, 64), sd["sigma_net.1.weight"])
    # a 'best' checkpoint (nerf/utils.py:984-986 drops density_grid) with the scalars load_checkpoint restores (:1026-1030)
    best = {k: v for k, v in sd.items() if k != "density_grid"}
    torch.save({"model": best, "epoch": 3, "global_step": 900, "mean_count": 73211, "mean_density": 0.0625}, tmp_path / "best.pth")
    model, extra = CK.read_checkpoint_full(str(tmp_path / "best.pth"))
    assert "density_grid" not in model and extra == {"mean_count": 73211, "mean_density": 0.0625, "epoch": 3, "global_step": 900}

    class Ren:                                                                        # the attributes load_renderer_buffers touches
        pass
    r = Ren()
    r.density_grid, r.density_bitfield = torch.zeros(2, 8), torch.zeros(2, dtype=torch.uint8)
    r.aabb_train, r.aabb_infer, r.step_counter = torch.zeros(6), torch.zeros(6), torch.zeros(16, 2, dtype=torch.int32)
    r.mean_count, r.mean_density = 0, 0
    CK.load_renderer_buffers(r, CK.split_state_dict(model)[1], extra)
    assert r.mean_count == 73211 and r.mean_density == 0.0625 and torch.equal(r.density_bitfield, sd["density_bitfield"])
    assert CK.read_checkpoint_full(str(path))[1] == {"epoch": 1}
    # a full checkpoint also carries torch_ema's state (nerf/utils.py:955-958): it loads into WeightEMA as it is
    from ngp.train import WeightEMA
    params = [torch.nn.Parameter(sd["encoder.embeddings"].clone()), torch.nn.Parameter(sd["sigma_net.0.weight"].clone())]
    shadow = [p.detach() * 0.5 for p in params]
    torch.save({"model": sd, "epoch": 7, "ema": {"decay": 0.95, "num_updates": 12, "shadow_params": shadow, "collected_params": None}}, tmp_path / "full.pth")
    _, extra = CK.read_checkpoint_full(str(tmp_path / "full.pth"))
    ema = WeightEMA(params, decay=0.5)
    ema.load_state_dict(extra["ema"])
    assert ema.decay == 0.95 and ema.num_updates == 12 and all(torch.equal(a, b) for a, b in zip(ema.shadow, shadow))


@pytest.mark.gpu
def test_linear_checkpoint_renders_on_the_fused_path(dev):
    """A default (nn.Linear) model saved in the reference's layout, loaded back as an FFMLP field: the same sigma / rgb as
    the nn.Linear model under autocast, through the drop-in ops and through the one-launch kernel."""
    from ngp import workload as W
    from ngp.field import NGPField
    from ngp.render import NGPRenderer
    torch.manual_seed(3)
    lin = NGPField(bound=W.BOUND).to(dev)
    with torch.no_grad():
        lin.encoder.embeddings.uniform_(-0.5, 0.5)
        for layer in list(lin.sigma_net) + list(lin.color_net):
            layer.weight.uniform_(-0.3, 0.3)
    ren = NGPRenderer(lin, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    ren.load_density_grid(W.density_grid())
    buf = io.BytesIO()
    torch.save({"model": CK.to_reference_state_dict(ren)}, buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu", weights_only=True)["model"]
    field_sd, ren_sd = CK.split_state_dict(sd)
    assert "sigma_net.0.weight" in field_sd and "density_bitfield" in ren_sd
    ff = CK.field_from_state_dict(field_sd, bound=W.BOUND, fused=True).to(dev).eval()
    ren2 = NGPRenderer(ff, bound=W.BOUND, cuda_ray=True, density_thresh=10.0).to(dev).eval()
    CK.load_renderer_buffers(ren2, ren_sd)
    assert torch.equal(ren2.density_bitfield, ren.density_bitfield) and torch.equal(ren2.density_grid, ren.density_grid)

    g = torch.Generator(device=dev).manual_seed(4)
    x = (torch.rand(4096, 3, device=dev, generator=g) * 2 - 1) * W.BOUND
    d = F.normalize(torch.randn(4096, 3, device=dev, generator=g), dim=-1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s_lin, c_lin = lin(x, d)
        s_ff, c_ff = ff(x, d)
    s_fu, c_fu = ff.forward_fused(x, d)
    # same function; the layers accumulate in a different order (hipBLASLt vs MFMA tiles): half-precision noise only
    scale = float(s_lin.abs().max())
    assert float((s_ff.float() - s_lin.float()).abs().max()) < 1e-2 * scale
    assert float((s_fu - s_lin.float()).abs().max()) < 1e-2 * scale
    assert float((c_ff.float() - c_lin.float()).abs().max()) < 4e-3 and float((c_fu - c_lin.float()).abs().max()) < 4e-3
    assert float((s_fu - s_ff.float()).abs().max()) < 2e-3 * scale                   # fused vs drop-in FFMLP: the same kernels' arithmetic
