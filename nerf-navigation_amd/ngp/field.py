"""The reference's field models on top of the drop-in ops.

`NGPFieldFF` mirrors nerf/network_ff.py:11-148 (hash grid -> FFMLP -> trunc_exp ; SH ++ geo ++ 0 -> FFMLP -> sigmoid),
`NGPField` mirrors nerf/network.py:10-206 (the same field with bias-free nn.Linear layers).  Both keep the reference's
method names (forward / density / color / get_params) and tensor contracts so that a torch-ngp style renderer or the
nav/ lambdas (simulate.py:343-347) can call them unchanged.  `fused_state()` packs the FF model for the one-launch
kernels of csrc/render_fused.hip.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import ngp_hip as _hip
from ffmlp import FFMLP
from gridencoder import GridEncoder
from shencoder import SHEncoder


class _trunc_exp(Function):
    """activation.py:5-18 : exp forward in float32, gradient through exp(clamp(x, -15, 15))."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply


def get_encoder(encoding, input_dim=3, degree=4, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                desired_resolution=2048, align_corners=False, **kwargs):
    """encoding.py:45-77 for the encoders on the hot path."""
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency":
        from freqencoder import FreqEncoder
        enc = FreqEncoder(input_dim=input_dim, degree=kwargs.get("multires", 6))
    elif encoding == "sphere_harmonics":
        enc = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        enc = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                          log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                          gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners)
    else:
        raise NotImplementedError("Unknown encoding mode, choose from [None, frequency, sphere_harmonics, hashgrid, tiledgrid]")
    return enc, enc.output_dim


class _field_train(Function):
    """NGPFieldFF.forward and its backward in native launches (csrc/field_train.hip): forward = one launch
    that keeps 64 B per sample; backward = both networks recomputed, activation and weight gradients on the matrix cores (two launches),
    then the table scatter.  Replaces ~40 op launches and ~1.5 KB per sample of saved / copied activations of the op-by-op graph.
    Values follow the op-by-op path under autocast (same half roundings; weight gradients are summed in float32 and rounded once)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, d, embeddings, w_sigma, w_color, field):
        x, d = x.contiguous(), d.contiguous()
        M = x.shape[0]
        L = _hip.lib()
        f = field.fused_state(1.0)                       # the renderer applies density_scale itself (nerf/renderer.py:392)
        sig = torch.empty(M, dtype=torch.float32, device=x.device)
        rgb = torch.empty(M, 3, dtype=torch.float32, device=x.device)
        saved = _hip.workspace(L.ngp_field_train_saved_bytes(M), x.device)
        with _hip.timed("field_train_forward"):
            _hip.check(L.ngp_field_train_forward(ctypes.byref(f), _hip.ptr(x), _hip.ptr(d), M, _hip.ptr(sig), _hip.ptr(rgb), _hip.ptr(saved),
                                                 saved.numel(), _hip.stream()), "field_train_forward")
        ctx.save_for_backward(x, d, saved, *field._fused["tensors"])
        ctx.fstruct, ctx.field = f, field
        return sig, rgb

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_sig, g_rgb):
        x, d, saved, emb_half, ws_half, wc_half = ctx.saved_tensors
        field, enc = ctx.field, ctx.field.encoder
        M = x.shape[0]
        L = _hip.lib()
        g_sig = torch.zeros(M, dtype=torch.float32, device=x.device) if g_sig is None else g_sig.contiguous().float()
        g_rgb = torch.zeros(M, 3, dtype=torch.float32, device=x.device) if g_rgb is None else g_rgb.contiguous().float()
        grad_enc = torch.empty(enc.num_levels, M, enc.level_dim, dtype=torch.float16, device=x.device)
        g_ws = torch.empty(ws_half.numel(), dtype=torch.float32, device=x.device)
        g_wc = torch.empty(wc_half.numel(), dtype=torch.float32, device=x.device)
        work = _hip.workspace(L.ngp_field_train_workspace(M), x.device)      # per-workgroup partial sums of the weight gradients, the live list: nothing to clear
        from gridencoder import grid as G
        binned = bool(ctx.needs_input_grad[2]) and G.BINNED_SCATTER and G.offsets_max_rows(enc.offsets) <= (1 << 19)
        # only the samples that got a gradient (behind the compositor's early exit half of a converged batch gets none): the kernels list them and the
        # binned scatter takes the list along; the atomic scatter (tables beyond 2^19 rows per level) wants every row, in sample order
        live_only = binned and 0 < M <= G.PHASE_MAX_POINTS
        with _hip.timed("field_train_backward"):
            _hip.check(L.ngp_field_train_backward(ctypes.byref(ctx.fstruct), _hip.ptr(saved), _hip.ptr(d), M, _hip.ptr(g_sig), _hip.ptr(g_rgb),
                                                  _hip.ptr(grad_enc), _hip.ptr(g_ws), _hip.ptr(g_wc), _hip.ptr(work), work.numel(), int(live_only),
                                                  _hip.stream()), "field_train_backward")
        listed = None
        if live_only:
            p_list, p_count = ctypes.c_void_p(), ctypes.c_void_p()
            _hip.check(L.ngp_field_train_live_list(_hip.ptr(work), M, ctypes.byref(p_list), ctypes.byref(p_count)), "field_train_live_list")
            listed = (p_list, p_count)
        # data-parallel training: the exchange (ngp/train.py GradExchange) takes the gradients as they appear -- the weight bucket now, so that its
        # all-reduce runs underneath the table scatter; the table gradient as the half tensor the scatter writes, pre-divided by the world size
        sink = getattr(field, "grad_sink", None)
        if sink is not None:
            inv = 1.0 / sink.world_size()
            sink.deliver([field.sigma_net.weights, field.color_net.weights], torch.cat([g_ws, g_wc]).mul_(inv))
        grad_emb = None
        if ctx.needs_input_grad[2]:
            inputs = ((x + field.bound) / (2 * field.bound)).contiguous()          # GridEncoder.forward (gridencoder/grid.py:144)
            S = float(np.log2(enc.per_level_scale))
            if binned:
                # summed on chip (exactly, 64-bit fixed point) and written once: as the float32 gradient of the float32 parameter (no zero fill, no
                # atomics, no widening), or, for the exchange, as a half tensor scaled by 1 / world size (half the bytes on the links)
                grad_emb = G.table_gradient_binned(grad_enc, inputs, enc.offsets, M, enc.num_levels, S, enc.base_resolution, enc.gridtype_id,
                                                   enc.align_corners, out_dtype=torch.float16 if sink is not None else torch.float32,
                                                   out_scale=(1.0 / sink.world_size()) if sink is not None else 1.0,
                                                   on_group=(lambda out, r0, r1: sink.deliver_rows(enc.embeddings, out, r0, r1)) if sink is not None else None,
                                                   groups=getattr(sink, "level_groups", None), listed=listed)
                delivered_table = sink is not None       # group by group, finest levels first: their all-reduce runs while the coarser ones are summed
            else:
                grad_emb = torch.zeros_like(emb_half)
                dummy = torch.empty(1, dtype=torch.float16, device=x.device)
                with _hip.timed("grid_encode_backward"):
                    _hip.check(L.ngp_grid_encode_backward(_hip.ptr(grad_enc), _hip.ptr(inputs), _hip.ptr(emb_half), _hip.ptr(enc.offsets), _hip.ptr(grad_emb),
                                                          M, 3, enc.level_dim, enc.num_levels, S, enc.base_resolution, 0,
                                                          _hip.ptr(dummy), _hip.ptr(dummy), enc.gridtype_id, int(enc.align_corners), _hip.F16, _hip.stream()),
                               "grid_encode_backward")
                delivered_table = False
                if sink is not None:
                    grad_emb.mul_(1.0 / sink.world_size())
            if sink is not None and not delivered_table:
                sink.deliver(enc.embeddings, grad_emb)
        if sink is not None:
            return None, None, None, None, None, None
        return None, None, grad_emb, g_ws, g_wc, None


class _ParamEpoch:
    """Cached derived copies of the parameters (half table, packed weights, the nav kernels' transposes) are keyed on the tensors'
    `_version` counters AND on this epoch.  Not every in-place update bumps `_version` (torch's fused Adam updates parameters without
    it), so the epoch advances
      * whenever a backward writes the `.grad` of one of the field's parameters (post-accumulate-grad hooks, `_watch_parameters`): an
        optimiser step can only come after that, and it comes AFTER the training forward -- which itself fills the cache with the
        pre-step copies (`_field_train.forward` calls `fused_state`), so advancing at the forward alone would leave those copies valid
        for the first frozen-model call after the step;
      * at every forward that is differentiated with respect to the parameters (kept: it costs nothing and covers a caller that swaps
        parameter storage between steps).
    `mark_updated()` is for callers that write parameters by other means (load_state_dict does bump `_version`; `p.data = ...` does not).

    An optimiser that REFRESHES the half copies itself (ngp/optim.py NativeAdam writes them in its update launch) says so with
    `mirrors_are_current()` (NGPFieldFF): the copies then stay valid through the next training forward, which would otherwise convert
    the 12.7 M-entry table again on every step.  `_foreign_writes` counts the `mark_updated()` calls, the writes this class cannot see."""
    _param_epoch = 0
    _foreign_writes = 0
    _mirror_ok = False

    def _training_forward(self):
        if torch.is_grad_enabled() and self.encoder.embeddings.requires_grad and not self._mirror_ok:
            self._param_epoch += 1

    def mark_updated(self):
        self._param_epoch += 1
        self._foreign_writes += 1
        self._mirror_ok = False
        emb = getattr(getattr(self, "encoder", None), "embeddings", None)
        if emb is not None:
            emb._ngp_half = None                     # the op-by-op encoder's own half copy (gridencoder/grid.py) is keyed on `_version` alone

    def _watch_parameters(self):
        """call at the end of __init__: every parameter's gradient accumulation advances the epoch (the hook holds the module weakly)"""
        import weakref
        ref = weakref.ref(self)

        def bump(_param):
            me = ref()
            if me is not None:
                me._param_epoch += 1
                me._mirror_ok = False
        for p in self.parameters():
            p.register_post_accumulate_grad_hook(bump)


class NGPFieldFF(_ParamEpoch, nn.Module):
    """nerf/network_ff.py: density net FFMLP(32,16,64,num_layers=2), colour net FFMLP(32,3,64,num_layers=3)."""

    def __init__(self, bound=1, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64,
                 density_scale=1):
        super().__init__()
        self.bound = bound
        self.density_scale = density_scale
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder("hashgrid", desired_resolution=2048 * bound)
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + geo_feat_dim, hidden_dim=hidden_dim, num_layers=num_layers)
        self.encoder_dir, self.in_dim_color = get_encoder("sphere_harmonics")
        self.in_dim_color += geo_feat_dim + 1                      # padded to 32 (network_ff.py:42)
        self.color_net = FFMLP(input_dim=self.in_dim_color, output_dim=3, hidden_dim=hidden_dim_color, num_layers=num_layers_color)
        self._fused = None
        self.fused_training = True           # training forwards under autocast go through _field_train when the field has the default shape
        self.fused_inference = True          # forwards without gradients under autocast go through forward_fused (one launch), likewise
        self.grad_sink = None                # ngp/train.py GradExchange while a data-parallel step runs: the native backward delivers its gradients to it
        self._watch_parameters()

    def _fused_training_applies(self, x, d):
        p = self.encoder.embeddings
        return (self.fused_training and x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled() and p.requires_grad
                and p.dtype == torch.float32 and not x.requires_grad and not d.requires_grad and x.dim() == 2 and self._fused_shape_ok())

    def _fused_inference_applies(self, x, d):
        return (self.fused_inference and x.is_cuda and not torch.is_grad_enabled() and torch.is_autocast_enabled() and x.dim() == 2
                and self.encoder.embeddings.dtype == torch.float32 and self._fused_shape_ok())

    def forward(self, x, d):
        """sigma [M] float32, rgb [M,3] (half under autocast) as nerf/network_ff.py:51-77 returns them.  Under autocast the default field takes
        one native launch when nothing is differentiated (`forward_fused`: what an unmodified renderer's inference loop gets per iteration
        instead of ~25 small launches) and three when the parameters are (`_field_train`); every other case runs op by op."""
        self._training_forward()
        if self._fused_training_applies(x, d):
            return _field_train.apply(x, d, self.encoder.embeddings, self.sigma_net.weights, self.color_net.weights, self)
        if self._fused_inference_applies(x, d):
            return self.forward_fused(x, d, density_scale=1.0, rgb_dtype=torch.float16)     # rgb half, like torch.sigmoid of FFMLP's half output
        x = self.encoder(x, bound=self.bound)
        h = self.sigma_net(x)
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        d = self.encoder_dir(d)
        h = self.color_net(self._color_input(d, geo_feat))
        return sigma, torch.sigmoid(h)

    def density(self, x):
        self._training_forward()
        x = self.encoder(x, bound=self.bound)
        h = self.sigma_net(x)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    @torch.no_grad()
    def density_sigma(self, x):
        """`density(x)['sigma']` for a caller that wants nothing else and no gradient -- the occupancy-grid refresh (nerf/renderer.py:478-486,511-517), millions
        of random points every 16 training steps.  Default field under autocast: two native launches (ngp_field_density: the level-by-level encoder of
        the training forward + the density net on the matrix cores; the logits of `forward_fused` bit for bit) instead of encoder + permute + FFMLP + exp;
        anything else: the op chain."""
        if not (x.is_cuda and torch.is_autocast_enabled() and x.dim() == 2 and self.encoder.embeddings.dtype == torch.float32 and self._fused_shape_ok()):
            return self.density(x)["sigma"].reshape(-1).float()
        x = x.contiguous().float()
        M = x.shape[0]
        L = _hip.lib()
        sig = torch.empty(M, dtype=torch.float32, device=x.device)
        ws = _hip.workspace(L.ngp_field_density_workspace(M), x.device)
        f = self.fused_state(1.0)
        _hip.check(L.ngp_field_density(ctypes.byref(f), _hip.ptr(x), M, _hip.ptr(sig), _hip.ptr(ws), ws.numel(), _hip.stream()), "field_density")
        return sig

    @staticmethod
    def _color_input(d, geo_feat):
        """cat(SH16, geo15, one zero column) (network_ff.py:67-68).  The SH features are float32 and geo_feat is half under autocast: the
        reference's cat promotes all 32 columns to float32 and FFMLP's cast_inputs=half rounds them back -- two full-width copies per
        call (282 MB each at 2.2 M training points).  Rounding the 16 SH columns first and concatenating halves gives the same 32 halves."""
        if geo_feat.dtype == torch.float16 and torch.is_autocast_enabled():
            d = d.to(torch.float16)
        p = torch.zeros_like(geo_feat[..., :1])
        return torch.cat([d, geo_feat, p], dim=-1)

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        self._training_forward()
        if mask is not None:
            # network.py:139-146 gathers with `t[mask]` and scatters with `rgbs[mask] = h`; the row indices of a boolean mask
            # are unique, so index_select / index_copy give the same values and the same gradients without autograd's
            # sorting index_put(accumulate=True) backward (0.7 ms of a 4 ms pose-filter iteration)
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            rows = mask.nonzero(as_tuple=True)[0]
            if rows.numel() == 0:
                return rgbs
            d, geo_feat = d.index_select(0, rows), geo_feat.index_select(0, rows)
        d = self.encoder_dir(d)
        h = torch.sigmoid(self.color_net(self._color_input(d, geo_feat)))
        if mask is not None:
            return rgbs.index_copy(0, rows, h.to(rgbs.dtype))
        return h

    def get_params(self, lr):
        return [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.sigma_net.parameters(), "lr": lr},
                {"params": self.encoder_dir.parameters(), "lr": lr}, {"params": self.color_net.parameters(), "lr": lr}]

    # ---- fused path -------------------------------------------------------------------------------------------
    def load_arrays(self, model):
        """Load a workload.make_model() dict (float32 arrays in the module's own layouts)."""
        with torch.no_grad():
            self.encoder.embeddings.copy_(torch.from_numpy(model["embeddings"]))
            self.sigma_net.weights.copy_(torch.from_numpy(model["sigma_weights"]))
            self.color_net.weights.copy_(torch.from_numpy(model["color_weights"]))
        self._fused = None
        return self

    def _apply(self, fn, *args, **kwargs):
        # .to() / .half() / .float() / .cuda(i) replace parameter storage without bumping _version: drop the fused half copies
        self._fused = None
        return super()._apply(fn, *args, **kwargs)

    def _check_fused_shape(self):
        """The one-launch kernels (csrc/render_fused.hip) are specialised for the reference's default field: 16 levels x 2 features of a
        3-D hash grid, density net 32-64-64-16, colour net 32-64-64-64-16.  Any other shape would be misread silently: refuse it."""
        if not self._fused_shape_ok():
            raise RuntimeError("the fused path supports the default field only (hash grid 16 x 2, FFMLP 32-64-64-16 and 32-64-64-64-16); "
                               "use the per-op path (run_cuda / forward) for this configuration")

    def _fused_shape_ok(self):
        e, sn, cn = self.encoder, self.sigma_net, self.color_net
        return (e.num_levels == 16 and e.level_dim == 2 and e.input_dim == 3 and e.gridtype == "hash" and not e.align_corners
              and sn.hidden_dim == 64 and cn.hidden_dim == 64 and sn.num_layers == 2 and cn.num_layers == 3
              and sn.input_dim == 32 and cn.input_dim == 32 and self.geo_feat_dim == 15
              and sn.weights.numel() == 7168 and cn.weights.numel() == 11264)

    def fused_state(self, density_scale=None):
        """Half copies of table and weights plus the ngp_field_t describing them (include/ngp_hip.h).  The reference
        converts the table to half on EVERY forward under autocast (gridencoder/grid.py:38-39: a 50 MB read + 25 MB
        write per call); the fused path keeps the half table resident and rebuilds it only when parameters change.
        density_scale: the renderer's (nerf/renderer.py:64 owns the one density_scale); default the field's own."""
        self._check_fused_shape()
        scale = float(self.density_scale if density_scale is None else density_scale)
        emb_p = self.encoder.embeddings
        key = self._fused_key()
        if self._fused is None or self._fused["key"] != key:
            emb = emb_p.detach().to(torch.half).contiguous()
            ws = self.sigma_net.weights.detach().to(torch.half).contiguous()
            wc = self.color_net.weights.detach().to(torch.half).contiguous()
            self._fused = {"key": key, "tensors": (emb, ws, wc), "structs": {}, "storage": self._storage_key()}
        st = self._fused["structs"]
        if scale not in st:
            emb, ws, wc = self._fused["tensors"]
            st[scale] = _hip.ngp_field_t(emb.data_ptr(), self.encoder.offsets.data_ptr(), ws.data_ptr(), wc.data_ptr(),
                                         self.encoder.num_levels, self.encoder.base_resolution,
                                         float(np.log2(self.encoder.per_level_scale)), float(self.bound), scale)
        return st[scale]

    def _storage_key(self):
        """what identifies the parameter VALUES as far as torch and `mark_updated()` can tell (gradient accumulation does not change them)"""
        emb_p, ws, wc = self.encoder.embeddings, self.sigma_net.weights, self.color_net.weights
        return (self._foreign_writes, emb_p._version, ws._version, wc._version, emb_p.data_ptr(), ws.data_ptr(), wc.data_ptr(), str(emb_p.device))

    def _fused_key(self):
        return (self._param_epoch,) + self._storage_key() + (self.encoder.offsets.data_ptr(), float(self.bound))

    @torch.no_grad()
    def half_mirrors(self):
        """{parameter: the float16 copy the fused kernels read} for an optimiser that writes the copies itself (ngp/optim.py NativeAdam.half_mirrors).
        The buffers of the last forward are handed out as they are when the parameters have not been written since (a backward in between only
        advanced the epoch); otherwise they are rebuilt from the parameters first, so that a skipped update leaves valid copies behind."""
        self._check_fused_shape()
        if self._fused is None or self._fused.get("storage") != self._storage_key():
            self._fused = None
            self.fused_state()
        emb, ws, wc = self._fused["tensors"]
        return {self.encoder.embeddings: emb, self.sigma_net.weights: ws, self.color_net.weights: wc}

    def mirrors_are_current(self):
        """the optimiser has just updated the parameters AND the buffers `half_mirrors()` handed out (or skipped both): keep them for the next forward"""
        emb_p = self.encoder.embeddings
        if self._fused is not None:
            self._fused["key"] = self._fused_key()
            self._fused["storage"] = self._storage_key()
            self._mirror_ok = True
            emb_p._ngp_half = (emb_p._version, emb_p.data_ptr(), self._fused["tensors"][0])      # the op-by-op encoder reads the same copy
        else:
            emb_p._ngp_half = None

    @torch.no_grad()
    def forward_fused(self, x, d, density_scale=None, rgb_dtype=torch.float32):
        """sigma (already times density_scale) and rgb for [M,3] points / directions in one launch (sigma float32; rgb float32, or float16 written by
        the kernel itself: the values are halves either way)."""
        x, d = x.contiguous().float(), d.contiguous().float()
        M = x.shape[0]
        sig = torch.empty(M, dtype=torch.float32, device=x.device)
        rgb = torch.empty(M, 3, dtype=rgb_dtype, device=x.device)
        f = self.fused_state(density_scale)
        L = _hip.lib()
        entry = L.ngp_field_forward_half if rgb_dtype == torch.float16 else L.ngp_field_forward
        _hip.check(entry(ctypes.byref(f), _hip.ptr(x), _hip.ptr(d), M, _hip.ptr(sig), _hip.ptr(rgb), _hip.stream()), "field_forward")
        return sig, rgb


class NGPField(_ParamEpoch, nn.Module):
    """nerf/network.py: the same field with nn.Linear(bias=False) layers (32->64->16 ; 31->64->64->3)."""

    def __init__(self, bound=1, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64,
                 density_scale=1, bg_radius=-1, num_layers_bg=2, hidden_dim_bg=64):
        super().__init__()
        self.bound = bound
        self.density_scale = density_scale
        self.num_layers, self.num_layers_color, self.geo_feat_dim = num_layers, num_layers_color, geo_feat_dim
        self.encoder, self.in_dim = get_encoder("hashgrid", desired_resolution=2048 * bound)
        dims = [self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim]
        self.sigma_net = nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers)])
        self.encoder_dir, self.in_dim_dir = get_encoder("sphere_harmonics")
        dims = [self.in_dim_dir + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [3]
        self.color_net = nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers_color)])
        # background model on a sphere of radius bg_radius (nerf/network.py:69-92): a small 2-D hash grid of the sphere coordinates
        # ++ SH of the direction -> MLP -> sigmoid.  Only the default network has one (main_nerf.py:73: "not implemented for --ff").
        self.bg_radius = bg_radius
        self.bg_net = None
        if bg_radius > 0:
            self.encoder_bg, self.in_dim_bg = get_encoder("hashgrid", input_dim=2, num_levels=4, log2_hashmap_size=19, desired_resolution=2048)
            dims = [self.in_dim_bg + self.in_dim_dir] + [hidden_dim_bg] * (num_layers_bg - 1) + [3]
            self.bg_net = nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers_bg)])
        self._watch_parameters()

    @staticmethod
    def _mlp(layers, h):
        for i, layer in enumerate(layers):
            h = layer(h)
            if i != len(layers) - 1:
                h = F.relu(h, inplace=True)
        return h

    def forward(self, x, d):
        self._training_forward()
        h = self._mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        h = self._mlp(self.color_net, torch.cat([self.encoder_dir(d), geo_feat], dim=-1))
        return sigma, torch.sigmoid(h)

    def density(self, x):
        self._training_forward()
        h = self._mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        self._training_forward()
        if mask is not None:
            # network.py:139-146 gathers with `t[mask]` and scatters with `rgbs[mask] = h`; the row indices of a boolean mask
            # are unique, so index_select / index_copy give the same values and the same gradients without autograd's
            # sorting index_put(accumulate=True) backward (0.7 ms of a 4 ms pose-filter iteration)
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            rows = mask.nonzero(as_tuple=True)[0]
            if rows.numel() == 0:
                return rgbs
            d, geo_feat = d.index_select(0, rows), geo_feat.index_select(0, rows)
        h = torch.sigmoid(self._mlp(self.color_net, torch.cat([self.encoder_dir(d), geo_feat], dim=-1)))
        if mask is not None:
            return rgbs.index_copy(0, rows, h.to(rgbs.dtype))
        return h

    def background(self, x, d):
        """nerf/network.py:145-161: x [N,2] sphere coordinates in [-1,1] (raymarching.sph_from_ray), d [N,3] -> rgb [N,3]"""
        self._training_forward()
        h = torch.cat([self.encoder_dir(d), self.encoder_bg(x)], dim=-1)
        return torch.sigmoid(self._mlp(self.bg_net, h))

    def get_params(self, lr):
        params = [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.sigma_net.parameters(), "lr": lr},
                  {"params": self.encoder_dir.parameters(), "lr": lr}, {"params": self.color_net.parameters(), "lr": lr}]
        if self.bg_radius > 0:
            params += [{"params": self.encoder_bg.parameters(), "lr": lr}, {"params": self.bg_net.parameters(), "lr": lr}]
        return params
