"""Checkpoint compatibility: torch-ngp `ngp_ep*.pth` state dicts <-> this package's modules, and the
adapter that lets a model trained with the default nn.Linear field (nerf/network.py:10-206, what the published
Stonehenge checkpoint is, README.md:83) run on the matrix-core FFMLP / one-launch fused path.

The reference's model IS its renderer (class NeRFNetwork(NeRFRenderer)), so one state dict holds both:
  renderer buffers  aabb_train, aabb_infer, density_grid, density_bitfield, step_counter   (nerf/renderer.py:60-88)
  field             encoder.embeddings, encoder.offsets, sigma_net.{i}.weight, color_net.{i}.weight   (nn.Linear field)
                or  encoder.embeddings, encoder.offsets, sigma_net.weights, color_net.weights         (FFMLP field)
Here the renderer owns a `.field`, so the keys are split.  Files are read with torch.load(weights_only=True) only.

nn.Linear -> FFMLP.  FFMLP(num_layers = n) is n + 1 matmuls (ffmlp.cu:371-407) while the default field has n Linear
layers, so the flat FFMLP weights get one extra hidden layer: the identity.  relu(I relu(h)) == relu(h) exactly (a product
with 1 and a sum with zeros are exact in any precision), so the converted model computes the same function:
  density net   Linear(32,64), Linear(64,16)            -> [W0 | I64 | W1]
  colour net    Linear(31,64), Linear(64,64), Linear(64,3) -> [V0 with a zero 32nd input column | V1 | I64 | V2 in rows 0..2 of 16]
(the FF field feeds cat(SH16, geo15, 0): network_ff.py:67-68; the default field feeds cat(SH16, geo15): network.py:104).
"""
import torch

RENDERER_KEYS = ("aabb_train", "aabb_infer", "density_grid", "density_bitfield", "step_counter")


def read_checkpoint(path):
    """-> the model state dict of a torch-ngp checkpoint file ({'model': state_dict, ...}, nerf/utils.py:938-1000) or of a
    bare state dict file.  weights_only=True: nothing in the file is executed."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    return blob["model"] if isinstance(blob, dict) and "model" in blob else blob


def read_checkpoint_full(path):
    """-> (model state dict, extra): `extra` holds the scalars the reference's Trainer.save_checkpoint stores beside the model
    (nerf/utils.py:944-953) and load_checkpoint restores into it (:1018-1022): mean_count, mean_density, epoch, global_step, and -- in a full
    checkpoint (:955-958, :1024-1025) -- `ema`, torch_ema's state dict {decay, num_updates, shadow_params, collected_params}, whose parameter
    order is the model's; `ngp.train.WeightEMA.load_state_dict` takes it as it is."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(blob, dict) and "model" in blob:
        return blob["model"], {k: blob[k] for k in ("mean_count", "mean_density", "epoch", "global_step", "ema", "optimizer", "lr_scheduler", "scaler", "stats")
                               if k in blob}
    return blob, {}


def split_state_dict(sd):
    """-> (field state dict, renderer buffers) of a reference model state dict."""
    field, ren = {}, {}
    for k, v in sd.items():
        (ren if k in RENDERER_KEYS else field)[k] = v
    return field, ren


def is_ff_state_dict(field_sd):
    return "sigma_net.weights" in field_sd


def _linear_weights(field_sd, prefix):
    out, i = [], 0
    while f"{prefix}.{i}.weight" in field_sd:
        out.append(field_sd[f"{prefix}.{i}.weight"].float())
        i += 1
    return out


def linear_to_ffmlp(weights, input_dim, hidden_dim=64, padded_output_dim=16):
    """[W0 [hidden, in'], hidden layers [hidden, hidden]..., W_last [out', hidden]] (nn.Linear weights, in' <= input_dim,
    out' <= padded_output_dim) -> (flat FFMLP weights, num_layers) with one identity hidden layer appended after the
    last real hidden layer."""
    assert len(weights) >= 2, "an MLP has at least an input and an output layer"
    first, last, hidden = weights[0], weights[-1], list(weights[1:-1])
    assert first.shape[0] == hidden_dim and last.shape[1] == hidden_dim and first.shape[1] <= input_dim
    assert last.shape[0] <= padded_output_dim and all(tuple(h.shape) == (hidden_dim, hidden_dim) for h in hidden)
    w0 = torch.zeros(hidden_dim, input_dim)
    w0[:, :first.shape[1]] = first
    wl = torch.zeros(padded_output_dim, hidden_dim)
    wl[:last.shape[0]] = last
    parts = [w0] + hidden + [torch.eye(hidden_dim), wl]
    num_layers = len(parts) - 1                        # FFMLP counts matmuls - 1
    return torch.cat([p.reshape(-1) for p in parts]), num_layers


def ff_state_dict_from_linear(field_sd, hidden_dim=64):
    """nn.Linear field state dict -> FFMLP field state dict (same encoder), see the module docstring."""
    sig, _ = linear_to_ffmlp(_linear_weights(field_sd, "sigma_net"), 32, hidden_dim)
    col, _ = linear_to_ffmlp(_linear_weights(field_sd, "color_net"), 32, hidden_dim)
    out = {k: v for k, v in field_sd.items() if not (k.startswith("sigma_net.") or k.startswith("color_net."))}
    out["sigma_net.weights"] = sig
    out["color_net.weights"] = col
    return out


def field_from_state_dict(field_sd, bound, density_scale=1, fused=True):
    """Build the field a reference state dict describes.  fused=True returns an NGPFieldFF (nn.Linear checkpoints are
    converted), which is what `NGPRenderer.render_fused` needs; fused=False keeps nn.Linear checkpoints as NGPField."""
    from .field import NGPField, NGPFieldFF
    n_sigma = len(_linear_weights(field_sd, "sigma_net"))
    n_color = len(_linear_weights(field_sd, "color_net"))
    if is_ff_state_dict(field_sd):
        field = NGPFieldFF(bound=bound, density_scale=density_scale)
    elif fused:
        assert n_sigma == 2 and n_color == 3, "the fused path is built for the reference's default depths (2 and 3 Linear layers)"
        field_sd = ff_state_dict_from_linear(field_sd)
        field = NGPFieldFF(bound=bound, density_scale=density_scale)
    else:
        field = NGPField(bound=bound, num_layers=n_sigma, num_layers_color=n_color, density_scale=density_scale)
    missing, unexpected = field.load_state_dict(field_sd, strict=False)
    assert not unexpected, f"unexpected keys {unexpected}"
    assert all(k.endswith("offsets") for k in missing), f"missing keys {missing}"
    if hasattr(field, "_fused"):
        field._fused = None
    return field


def load_renderer_buffers(renderer, ren_sd, extra=None):
    """density grid / bitfield / aabbs / step counter of a reference checkpoint into an NGPRenderer.  `extra` (read_checkpoint_full)
    restores mean_count / mean_density as the reference's load_checkpoint does: 'best' checkpoints drop density_grid, and without the
    stored scalars a resumed run would allocate N * max_steps samples for 16 steps and threshold its first grid refresh at 0."""
    with torch.no_grad():
        for k, v in ren_sd.items():
            if hasattr(renderer, k) and getattr(renderer, k) is not None and torch.is_tensor(getattr(renderer, k)):
                getattr(renderer, k).copy_(v.to(getattr(renderer, k).dtype))
    if "density_grid" in ren_sd:
        renderer.mean_density = float(torch.mean(renderer.density_grid.clamp(min=0)))
    if extra:
        if "mean_count" in extra:
            renderer.mean_count = int(extra["mean_count"])
        if "mean_density" in extra:
            renderer.mean_density = float(extra["mean_density"])


def to_reference_state_dict(renderer):
    """The inverse: one flat state dict with the reference's key names (what its Trainer.save_checkpoint writes)."""
    sd = {k: v.detach().clone() for k, v in renderer.field.state_dict().items()}
    for k in RENDERER_KEYS:
        if hasattr(renderer, k) and torch.is_tensor(getattr(renderer, k)):
            sd[k] = getattr(renderer, k).detach().clone()
    return sd


def write_checkpoint(path, renderer, trainer=None, epoch=0, stats=None, full=True):
    """Trainer.save_checkpoint (nerf/utils.py:938-986) for an NGPRenderer and (full=True) the optimisation state of an ngp.train.NGPTrainer: one file with
    the reference's keys -- epoch, global_step, stats, mean_count, mean_density, [optimizer, lr_scheduler, scaler, ema,] model -- each part in the layout of
    the class the reference uses for it (torch.optim.Adam, LambdaLR, GradScaler, torch_ema), so the reference's load_checkpoint reads it and
    `resume_trainer` reads the reference's.  Written with torch.save; every value is a tensor, a number, a string or a container of those, so
    torch.load(weights_only=True) reads it back."""
    state = {"epoch": int(epoch), "global_step": int(trainer.global_step) if trainer is not None else 0, "stats": stats if stats is not None else {}}
    if getattr(renderer, "cuda_ray", False):
        state["mean_count"], state["mean_density"] = int(renderer.mean_count), float(renderer.mean_density)
    if full and trainer is not None:
        t = trainer.state_dict()
        state["optimizer"], state["lr_scheduler"], state["scaler"] = t["optimizer"], t["lr_scheduler"], t["scaler"]
        if "ema" in t:
            state["ema"] = t["ema"]
    state["model"] = to_reference_state_dict(renderer)
    torch.save(state, path)
    return state


def resume_trainer(path, renderer, trainer):
    """Trainer.load_checkpoint (nerf/utils.py:1002-1060) for a file of write_checkpoint or of the reference (same field type as `renderer.field`): model
    and renderer buffers, mean_count / mean_density, and whatever optimisation state the file holds.  Returns the epoch stored in it."""
    field_sd_all, extra = read_checkpoint_full(path)
    field_sd, ren_sd = split_state_dict(field_sd_all)
    renderer.field.load_state_dict(field_sd)
    if hasattr(renderer.field, "mark_updated"):
        renderer.field.mark_updated()
    load_renderer_buffers(renderer, ren_sd, extra)
    trainer.load_state_dict({k: extra[k] for k in ("global_step", "optimizer", "lr_scheduler", "scaler", "ema") if k in extra})
    return int(extra.get("epoch", 0))
