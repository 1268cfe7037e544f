"""The Level-2 native surface: pybind11 modules `_raymarching`, `_gridencoder`, `_shencoder`, `_ffmlp`, `_freqencoder`
built from nerf-navigation_amd/bindings/*.cpp (plain C++ over the C ABI; nothing hipified) -- what the reference's OWN wrappers bind
with `import _raymarching as _backend` (raymarching/raymarching.py:9-12).

CPU: every module imports and its exported names, arities and argument kinds equal the reference's bindings.cpp / headers
(fixture tests/golden/native_surface.json, made from the reference's text by tests/golden/make_native_surface.py; re-parsed live when
/root/reference is present).  GPU: calls through the modules give the same bits as the ctypes path the packages use."""
import importlib
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

pkg = importlib.import_module("nerf-navigation_amd")
LIBDIR = os.path.join(pkg.ROOT, "lib")
HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "native_surface.json")))
KIND = {"torch.Tensor": "tensor", "typing.SupportsInt": "int", "typing.SupportsFloat": "float", "bool": "bool", "int": "int", "float": "float"}
REF_KIND = {"tensor": "tensor", "uint32_t": "int", "size_t": "int", "float": "float", "bool": "bool"}


def load(name):
    if LIBDIR not in sys.path:
        sys.path.insert(0, LIBDIR)
    try:
        return importlib.import_module(name)
    except ImportError as e:                                   # not built: build() makes them (make -C nerf-navigation_amd/bindings)
        pytest.fail(f"{name} is not built: {e}")


def signature(fn):
    first = fn.__doc__.strip().splitlines()[0]
    args = re.match(r"\w+\((.*)\) -> None", first).group(1)
    return [KIND[a.split(": ")[1]] for a in args.split(", ")] if args else []


@pytest.mark.parametrize("module", sorted(SURFACE))
def test_module_surface_equals_the_reference(module):
    m = load(module)
    exported = sorted(k for k in dir(m) if not k.startswith("_"))
    assert exported == sorted(SURFACE[module]), module
    for name, kinds in SURFACE[module].items():
        assert signature(getattr(m, name)) == [REF_KIND[k] for k in kinds], (module, name)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the authoring container only")
def test_fixture_is_current_with_the_reference_text(tmp_path):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_native_surface as gen
    gen.HERE = str(tmp_path)
    gen.main()
    assert json.load(open(tmp_path / "native_surface.json")) == SURFACE


def test_shims_reject_cpu_tensors_loudly():
    m = load("_raymarching")
    z = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        m.near_far_from_aabb(z, z, torch.zeros(6), 4, 0.2, torch.zeros(4), torch.zeros(4))
    g = load("_gridencoder")
    with pytest.raises(RuntimeError, match="GPU tensor"):
        g.grid_encode_forward(z, z, torch.zeros(3, dtype=torch.int32), z, 4, 3, 2, 2, 1.0, 16, False, z, 0, False)
    load("_ffmlp").allocate_splitk(3)                          # accepted no-ops (ffmlp.cu:721-740 created streams here)
    load("_ffmlp").free_splitk()


@pytest.mark.gpu
def test_shims_compute_the_same_bits_as_the_ctypes_path(dev, oracle):
    """the reference wrappers' own call shapes (raymarching/raymarching.py:37,326; gridencoder/grid.py:50; shencoder/sphere_harmonics.py:32;
    ffmlp/ffmlp.py:40) through the shim modules"""
    import raymarching
    from _util import blob_bitfield, camera_rays
    rm, ge, sh, ff = load("_raymarching"), load("_gridencoder"), load("_shencoder"), load("_ffmlp")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)                  # noqa: E731
    bound, cas, H = 2.0, 2, 128
    bitfield, _ = blob_bitfield(oracle, cas, H, seed=1, bound=bound)
    o, d = camera_rays(24, radius=3.2, seed=2)
    N = o.shape[0]
    to, td, aabb = t(o), t(d), t(np.array([-2, -2, -2, 2, 2, 2], np.float32))
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    rm.near_far_from_aabb(to, td, aabb, N, 0.2, nears, fars)
    n2, f2 = raymarching.near_far_from_aabb(to, td, aabb, 0.2)
    assert torch.equal(nears, n2) and torch.equal(fars, f2)
    n_step, M = 4, N * 4 + 128 - (N * 4) % 128
    alive = torch.arange(N, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rm.march_rays(N, n_step, alive, nears.clone(), to, td, bound, 0.0, 1024, cas, H, t(bitfield), nears, fars, xyzs, dirs, deltas, 0)
    x2, d2, l2 = raymarching.march_rays(N, n_step, alive, nears.clone(), to, td, bound, t(bitfield), cas, H, nears, fars, 128, False, 0.0, 1024)
    assert torch.equal(xyzs, x2) and torch.equal(deltas, l2) and (deltas[:, 0] > 0).sum() > 100
    idx = torch.empty(N, dtype=torch.int32, device=dev)
    coords = torch.randint(0, 128, (N, 3), dtype=torch.int32, device=dev)
    rm.morton3D(coords, N, idx)
    assert torch.equal(idx, raymarching.morton3D(coords))
    # grid encoder: float32 and float16 tables
    from gridencoder import GridEncoder
    enc = GridEncoder(num_levels=8, log2_hashmap_size=14, desired_resolution=256).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x01 = torch.rand(1000, 3, device=dev)
    for table in (enc.embeddings.detach(), enc.embeddings.detach().half()):
        out = torch.empty(8, 1000, 2, device=dev, dtype=table.dtype)
        dy = torch.empty(1000, 8 * 3 * 2, device=dev, dtype=table.dtype)
        ge.grid_encode_forward(x01, table, enc.offsets, out, 1000, 3, 2, 8, float(np.log2(enc.per_level_scale)), 16, True, dy, 0, False)
        ref, _ = oracle.grid_encode_forward(x01.cpu().numpy(), table.cpu().numpy(), enc.offsets.cpu().numpy(), enc.per_level_scale, 16, True, 0, False)
        assert np.array_equal(out.cpu().numpy().view(np.uint16 if table.dtype == torch.half else np.uint32),
                              ref.view(np.uint16 if table.dtype == torch.half else np.uint32))
    # SH and the fused MLP
    from shencoder import SHEncoder
    v = torch.nn.functional.normalize(torch.randn(500, 3, device=dev), dim=1)
    so = torch.empty(500, 16, device=dev)
    sh.sh_encode_forward(v, so, 500, 3, 4, False, torch.empty(1, device=dev))
    assert torch.equal(so, SHEncoder(degree=4)(v))
    from ffmlp import FFMLP
    net = FFMLP(32, 16, 64, 2).to(dev).eval()
    xin = torch.randn(256, 32, device=dev).half()
    outm = torch.empty(256, 16, device=dev, dtype=torch.half)
    ff.ffmlp_inference(xin, net.weights.detach().half(), 256, 32, 16, 64, 2, 0, 6, torch.empty(1, device=dev, dtype=torch.half), outm)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert torch.equal(outm, net(xin))


@pytest.mark.gpu
def test_raymarching_shim_refuses_wrong_dtypes_and_short_tensors(dev):
    """the `_raymarching` module checks scalar types and sizes (the reference's entry points check nothing and dispatch on the dtype; these
    kernels are float32 / int32 / uint8 only): a half, double or int64 tensor, or one that is too short, is a RuntimeError, not reinterpreted memory."""
    rm = load("_raymarching")
    N = 64
    o, d = torch.rand(N, 3, device=dev), torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=1)
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2], device=dev)
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    rm.near_far_from_aabb(o, d, aabb, N, 0.2, nears, fars)
    with pytest.raises(RuntimeError, match="rays_d must be Float"):
        rm.near_far_from_aabb(o, d.double(), aabb, N, 0.2, nears, fars)
    with pytest.raises(RuntimeError, match="nears has 32 elements"):
        rm.near_far_from_aabb(o, d, aabb, N, 0.2, nears[:32].contiguous(), fars)
    idx = torch.empty(N, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="coords must be Int"):
        rm.morton3D(torch.zeros(N, 3, dtype=torch.int64, device=dev), N, idx)
    M = N * 2 + 128
    sig, rgb, dl = torch.rand(M, device=dev), torch.rand(M, 3, device=dev), torch.rand(M, 2, device=dev)
    alive, rt = torch.arange(N, dtype=torch.int32, device=dev), nears.clone()
    ws, dep, img = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
    rm.composite_rays(N, 2, alive, rt, sig, rgb, dl, ws, dep, img)
    with pytest.raises(RuntimeError, match="rgbs must be Float"):                       # what the autocast field returns: the wrapper casts, a direct caller must
        rm.composite_rays(N, 2, alive, rt, sig, rgb.half(), dl, ws, dep, img)
    with pytest.raises(RuntimeError, match="sigmas has"):
        rm.composite_rays(N, 2, alive, rt, sig[:N].contiguous(), rgb, dl, ws, dep, img)
    bitfield = torch.full((2 * 128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    with pytest.raises(RuntimeError, match="xyzs must be"):
        rm.march_rays(N, 8, alive, rt, o, d, 2.0, 0.0, 1024, 2, 128, bitfield, nears, fars, xyzs, dirs, deltas, 0)      # M = 256 < n_alive * n_step = 512
    with pytest.raises(RuntimeError, match="grid must be Byte"):
        rm.march_rays(N, 2, alive, rt, o, d, 2.0, 0.0, 1024, 2, 128, bitfield.float(), nears, fars, xyzs, dirs, deltas, 0)
