"""The ctypes path's device guard (SURVEY 8b "Threading / streams"; VERDICT r2 missing 5): every entry point of libngp_hip.so runs under
`torch.cuda.device(<device of its tensors>)` and gets the current stream OF THAT DEVICE.  One GPU (or none) is enough to test the rule: the
current device is mocked."""
import contextlib
import importlib
import types

import numpy as np
import pytest
import torch

importlib.import_module("nerf-navigation_amd")
import ngp_hip as hip  # noqa: E402


class _FakeTensor:
    def __init__(self, index):
        self.device = torch.device("cuda", index)

    def data_ptr(self):
        return 0x1000


def _mock_devices(monkeypatch, current=0):
    state = {"current": current, "entered": []}

    @contextlib.contextmanager
    def device(dev):
        state["entered"].append(torch.device(dev).index)
        prev, state["current"] = state["current"], torch.device(dev).index
        try:
            yield
        finally:
            state["current"] = prev
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["current"])
    monkeypatch.setattr(torch.cuda, "device", device)
    monkeypatch.setattr(torch.cuda, "current_stream",
                        lambda d=None: types.SimpleNamespace(cuda_stream=0xA000 + (state["current"] if d is None else torch.device(d).index)))
    monkeypatch.setattr(hip, "_raw_stream", lambda index: 0xA000 + int(index))    # the raw handle of that device's current stream (ngp_hip._raw_stream)
    return state


def test_guard_enters_the_tensors_device_and_resolves_its_stream(monkeypatch):
    state = _mock_devices(monkeypatch, current=0)
    seen = {}

    def entry(a, n, s):
        seen["stream"] = s._as_parameter_.value                      # what ctypes reads while converting the arguments
        seen["current"] = state["current"]
        return 0
    call = hip._GuardedCall(entry, "fake")
    assert call(hip.ptr(_FakeTensor(1)), 3, hip.stream()) == 0
    assert state["entered"] == [1] and seen == {"stream": 0xA001, "current": 1}
    assert call(hip.ptr(_FakeTensor(0)), 3, hip.stream()) == 0       # already current: no guard, its own stream
    assert state["entered"] == [1] and seen == {"stream": 0xA000, "current": 0}
    assert call(7, 8, hip.stream()) == 0                              # no tensor argument (size queries): called as is
    with pytest.raises(RuntimeError, match="cuda:0 and cuda:1|cuda:1 and cuda:0"):
        call(hip.ptr(_FakeTensor(0)), hip.ptr(_FakeTensor(1)), hip.stream())


def test_every_export_is_guarded():
    L = hip.lib()
    for name in hip.EXPORTS:
        assert isinstance(getattr(L, name), hip._GuardedCall), name
    assert L.ngp_abi_version() >= 1


@pytest.mark.gpu
def test_call_on_cuda0_while_another_device_is_current(dev, monkeypatch, oracle):
    """a drop-in op on cuda:0 tensors while the process believes cuda:1 is current: the guard is entered with cuda:0 and the result is right"""
    import raymarching
    from _util import camera_rays
    o, d = camera_rays(8, radius=3.2, seed=2)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    to, td, ta = (torch.from_numpy(a).to(dev) for a in (o, d, aabb))
    entered = []
    real_device = torch.cuda.device

    class recording(real_device):                                     # a subclass: torch itself does isinstance(x, torch.cuda.device)
        def __init__(self, devarg):
            entered.append(torch.device(devarg).index)
            super().__init__(devarg)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    monkeypatch.setattr(torch.cuda, "device", recording)
    nears, fars = raymarching.near_far_from_aabb(to, td, ta, 0.2)
    monkeypatch.undo()
    assert 0 in entered
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    assert np.array_equal(nears.cpu().numpy().view(np.uint32), n_ref.view(np.uint32))
    assert np.array_equal(fars.cpu().numpy().view(np.uint32), f_ref.view(np.uint32))
