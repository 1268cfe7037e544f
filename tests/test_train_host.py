"""Host logic of ngp/train.py that needs neither a GPU nor a process group: the weight average (WeightEMA) against the closed form of the
published torch_ema recurrence (nerf/utils.py:324-325, :814-815, :851-853; torch_ema is absent from this image: parity unpinned by the package
itself), and NGPTrainer's epoch / evaluation plumbing on a stand-in renderer."""
import importlib

import torch

importlib.import_module("nerf-navigation_amd")
from ngp.train import NGPTrainer, WeightEMA  # noqa: E402

from test_distributed_cpu import _MockRenderer  # noqa: E402


def test_weight_ema_follows_the_published_recurrence():
    torch.manual_seed(0)
    p = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2), requires_grad=False)]
    ema = WeightEMA(p, decay=0.95)
    assert len(ema.shadow) == 2                                          # frozen parameters are not averaged
    want = [q.detach().clone().double() for q in p[:2]]
    for k in range(1, 40):
        with torch.no_grad():
            for q in p[:2]:
                q.add_(0.1 * torch.randn_like(q))
        ema.update()
        d = min(0.95, (1 + k) / (10 + k))                                # torch_ema: warm-up of the decay by the number of updates
        want = [d * w + (1 - d) * q.detach().double() for w, q in zip(want, p[:2])]
        for s, w in zip(ema.shadow, want):
            assert torch.allclose(s.double(), w, rtol=0, atol=2e-6)
    assert ema.num_updates == 39
    live = [q.detach().clone() for q in p[:2]]
    ema.store(); ema.copy_to()
    assert all(torch.equal(q.detach(), s) for q, s in zip(p[:2], ema.shadow))
    ema.restore()
    assert all(torch.equal(q.detach(), v) for q, v in zip(p[:2], live))
    fixed = WeightEMA(p, decay=0.5, use_num_updates=False)
    with torch.no_grad():
        p[1].add_(2.0)
    fixed.update()
    assert torch.allclose(fixed.shadow[1], p[1].detach() - 1.0)


def test_trainer_epochs_update_the_average_and_eval_uses_it():
    ren = _MockRenderer()
    tr = NGPTrainer(ren, lr=1e-2, iters=100, fp16=False, update_extra_interval=4, seed=1, ema_decay=0.95, steps_per_epoch=5)
    g = torch.Generator().manual_seed(3)
    for _ in range(12):
        tr.step(torch.rand(1, 64, 3, generator=g), torch.randn(1, 64, 3, generator=g), torch.rand(1, 64, 3, generator=g))
    assert tr.ema.num_updates == 2                                       # steps 5 and 10
    live = ren.field.w.detach().clone()
    assert not torch.equal(tr.ema.shadow[1], live)
    epoch = ren.field._epoch if hasattr(ren.field, "_epoch") else None
    with tr.eval_weights():
        assert torch.equal(ren.field.w.detach(), tr.ema.shadow[1])       # evaluation sees the averaged weights ...
    assert torch.equal(ren.field.w.detach(), live) and epoch is None     # ... and training continues from the live ones
    off = NGPTrainer(_MockRenderer(), fp16=False, ema_decay=None)
    assert off.ema is None
    with off.eval_weights():
        pass


def test_cpu_trainer_uses_torch_classes_and_its_state_round_trips():
    """a CPU model never gets the native optimiser (there is no CPU path in ngp/optim.py); the trainer's state has the reference's checkpoint keys
    (nerf/utils.py:944-958) and a second trainer resumes from it"""
    import pytest
    from ngp.optim import NativeAdam
    ren = _MockRenderer()
    tr = NGPTrainer(ren, lr=1e-2, iters=50, fp16=False, update_extra_interval=4, seed=1, ema_decay=0.95, steps_per_epoch=3)
    assert not tr.native_adam and type(tr.opt) is torch.optim.Adam and isinstance(tr.scaler, torch.amp.GradScaler)
    with pytest.raises(ValueError, match="no CPU path"):
        NativeAdam(ren.field.parameters())
    g = torch.Generator().manual_seed(3)
    batch = lambda: (torch.rand(1, 64, 3, generator=g), torch.randn(1, 64, 3, generator=g), torch.rand(1, 64, 3, generator=g))   # noqa: E731
    for _ in range(7):
        tr.step(*batch())
    import copy
    state = copy.deepcopy(tr.state_dict())           # what a torch.save / torch.load round trip gives: own tensors (load_state_dict adopts the ones it is handed)
    assert set(state) == {"global_step", "optimizer", "lr_scheduler", "scaler", "ema"} and state["global_step"] == 7
    ren2 = _MockRenderer()
    ren2.load_state_dict(ren.state_dict())
    tr2 = NGPTrainer(ren2, lr=1e-2, iters=50, fp16=False, update_extra_interval=4, seed=1, ema_decay=0.95, steps_per_epoch=3)
    tr2.load_state_dict(state)
    assert tr2.global_step == 7 and tr2.ema.num_updates == tr.ema.num_updates == 2
    assert tr2.sched.get_last_lr() == tr.sched.get_last_lr()
    b = batch()
    l1, l2 = tr.step(*b), tr2.step(*b)
    assert torch.equal(l1, l2)
    for p, q in zip(ren.field.parameters(), ren2.field.parameters()):
        assert torch.equal(p, q)
