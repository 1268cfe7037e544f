"""Host logic of the cached parameter copies (gridencoder/grid.py `_half_table`, ngp/field.py `_ParamEpoch`): an optimiser that updates
parameters without bumping `_version` (torch's fused Adam) must not leave a stale half table / packed field behind."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("nerf-navigation_amd")


def _fused_adam_step(p):
    opt = torch.optim.Adam([p], lr=0.1, fused=True)
    p.grad = torch.ones_like(p)
    v = p._version
    opt.step()
    return v == p._version


def test_half_table_is_not_reused_across_a_versionless_update():
    from gridencoder import grid as G
    emb = torch.nn.Parameter(torch.linspace(-1, 1, 64).reshape(32, 2).clone())
    frozen = G._half_table(emb, False)
    assert G._half_table(emb, False) is frozen                                   # a frozen model reuses the copy
    train = G._half_table(emb, True)                                             # a differentiated forward: fresh copy, cache dropped
    assert train is not frozen and torch.equal(train, frozen)
    silent = _fused_adam_step(emb)                                               # the update; on this torch it does not touch _version
    after = G._half_table(emb, False)
    assert torch.equal(after, emb.detach().half()) and not torch.equal(after, frozen)
    if not silent:                                                               # (a torch that does bump the version is fine too)
        assert after is not frozen


def test_param_epoch_advances_on_training_forwards_only():
    from ngp.field import _ParamEpoch

    class F(_ParamEpoch):
        class encoder:
            embeddings = torch.nn.Parameter(torch.zeros(4, 2))

    f = F()
    with torch.no_grad():
        f._training_forward()
    assert f._param_epoch == 0
    f._training_forward()
    assert f._param_epoch == 1
    F.encoder.embeddings.requires_grad_(False)                                   # a frozen model queried with input gradients (nav/)
    f._training_forward()
    assert f._param_epoch == 1
    f.mark_updated()
    assert f._param_epoch == 2


def test_fused_state_is_rebuilt_after_training_forward_backward_and_versionless_step():
    """the training forward itself fills the cache (`_field_train.forward` -> `fused_state`) under the epoch it has just
    advanced; a fused Adam step then changes the parameters without touching `_version`.  The post-accumulate-grad hooks advance the epoch
    when the backward writes `.grad`, so the first frozen-model call after the step rebuilds.  Host logic only (CPU tensors, no launch)."""
    from ngp.field import NGPFieldFF
    f = NGPFieldFF(bound=2)
    with torch.no_grad():
        f.encoder.embeddings.uniform_(-1, 1)
    params = [f.encoder.embeddings, f.sigma_net.weights, f.color_net.weights]
    f._training_forward()                                                        # what forward() does first ...
    f.fused_state(1.0)                                                           # ... and what _field_train.forward does next
    stale = f._fused["tensors"][0]
    assert f.fused_state(1.0) is f.fused_state(1.0) and f._fused["tensors"][0] is stale
    epoch = f._param_epoch
    sum(p.sum() for p in params).backward()                                      # any backward that reaches the parameters
    assert f._param_epoch >= epoch + len(params)
    opt = torch.optim.Adam(params, lr=0.1, fused=True)
    versions = [p._version for p in params]
    opt.step()
    silent = versions == [p._version for p in params]
    f.fused_state(1.0)                                                           # a frozen-model call (render_fused / forward_fused / nav struct)
    fresh = f._fused["tensors"]
    assert fresh[0] is not stale
    for half, p in zip(fresh, params):
        assert torch.equal(half, p.detach().half())
    assert not torch.equal(fresh[0], stale)
    assert silent or True                                                        # (holds whether or not this torch bumps _version)


def test_half_table_cache_serves_eval_no_grad_and_dies_with_any_backward():
    """the copy lives on the parameter; `model.eval()` + `no_grad()` reuses it without freezing the parameter; a backward that
    writes the table's .grad by ANY route (here: plain autograd, not _grid_encode) drops it -- the fused training route never enters _grid_encode."""
    from gridencoder import GridEncoder
    from gridencoder import grid as G
    enc = GridEncoder(num_levels=2, log2_hashmap_size=8, desired_resolution=32)
    emb = enc.embeddings
    assert emb.requires_grad
    with torch.no_grad():
        training = torch.is_grad_enabled() and emb.requires_grad                 # the rule _grid_encode.forward applies
        assert not training
        a = G._half_table(emb, training)
        assert G._half_table(emb, training) is a
    (emb * 2).sum().backward()                                                   # e.g. _field_train's scatter accumulating into .grad
    assert emb._ngp_half is None
    with torch.no_grad():
        emb.add_(1.0)                                                            # bumps _version; a fused Adam step would not
    b = G._half_table(emb, False)
    assert b is not a and torch.equal(b, emb.detach().half())
    moved = enc.to(torch.float64).to(torch.float32)                              # storage replaced: key mismatch -> rebuilt
    assert torch.equal(G._half_table(moved.embeddings, False), moved.embeddings.detach().half())
