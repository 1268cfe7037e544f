"""Drop-in `gridencoder` package: `_grid_encode` / `GridEncoder` of the reference's gridencoder/grid.py, backed by
libngp_hip.so (csrc/gridencoder.hip).

Same surface as the reference: argument order and defaults, the autocast rule (half table only when autocast is on
and C is even, grid.py:36-39), level-major kernel output permuted back to [B, L*C] (grid.py:42,52), dy_dx saved only
when the inputs require grad, a dense zero-initialised grad_embeddings per backward (grid.py:74), state_dict keys
`embeddings` and `offsets`.  As in the reference, `backward` calls a non-differentiable native op, so the gradient
w.r.t. the inputs is a constant under create_graph=True (the pose filter's Hessian relies on this).
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import ngp_hip as _hip

_gridtype_to_id = {"hash": 0, "tiled": 1}

# True: the gradient w.r.t. the inputs is computed in backward from the table itself (ngp_grid_encode_backward_inputs) instead
# of from a saved dy_dx [B, L*D*C] (grid.py:45-48): same bits, no 96-values-per-point tensor.  False: the reference's route.
RECOMPUTE_INPUT_GRAD = True
# ... from this many points on: the recomputing kernel walks the levels of a point in one lane (that keeps the reference's
# summation order), so a small batch (the planner's 10,000 body points) does not fill the chip and is faster with dy_dx
RECOMPUTE_MIN_POINTS = 32768
# True: the table gradient of a half D = 3, C = 2 grid (the reference's hash grid under autocast) is summed on chip by the binned scatter
# (ngp_grid_scatter_binned: no global atomics, float32 sums, float32 output = the parameter's dtype, no zero fill, no widening copy).
# False: the reference's route, half2 atomics into a zero-filled half table (ngp_grid_encode_backward).
BINNED_SCATTER = True
# True: D = 3, C = 2 forwards write [B, L*C] directly (ngp_grid_encode_forward_rows) instead of [L, B, C] + permute copy.  Same bits.
ROWS_FORWARD = True
# ... except where the caller says its points are incoherent (`with level_major_forward():`, the occupancy-grid refresh's random points): then the
# reference's level-major kernel + permute copy is faster, because one level's table (2 MB) at a time stays in an XCD's 4 MiB L2 while the row kernel
# has all 16 levels (25 MB) live at once (tools/time_grid_level_major.py: 2.1 M random points 0.97 against 1.26 ms; 640 k coherent points 0.14 against 0.10).
_LEVEL_MAJOR_DEPTH = 0


class level_major_forward:
    """context manager: D = 3, C = 2 forwards inside take the level-major kernel + permute (same bits as the row kernel)"""

    def __enter__(self):
        global _LEVEL_MAJOR_DEPTH
        _LEVEL_MAJOR_DEPTH += 1

    def __exit__(self, *exc):
        global _LEVEL_MAJOR_DEPTH
        _LEVEL_MAJOR_DEPTH -= 1
        return False


# The reference converts the float32 table to half on EVERY autocast forward (grid.py:38-39: 50 MB read + 25 MB written per call, 64
# times per rendered frame).  The half copy is kept ON THE PARAMETER (`embeddings._ngp_half`) while the parameter has the same version
# counter and storage, and it is dropped
#   * by every forward that will be differentiated with respect to the table (grad mode on and requires_grad: an optimiser step follows), and
#   * by a post-accumulate-grad hook GridEncoder registers on its table: whenever a backward -- through this op, through the field's native
#     training launches (ngp/field.py: _field_train, which never enters this function) or through anything else -- writes `.grad`, the copy
#     goes, because not every optimiser announces its in-place update through `_version` (torch's fused Adam does not).
def drop_half_table(embeddings):
    embeddings._ngp_half = None


def _half_table(embeddings, training):
    """The float32 table rounded to half, kept between forwards of a model that is not being trained (the drop-in inference loop encodes 64
    times per frame: 2.1 ms of conversions; `model.eval()` + `torch.no_grad()` is enough, the parameter need not be frozen)."""
    if training:
        embeddings._ngp_half = None
        return embeddings.detach().to(torch.half)
    c = getattr(embeddings, "_ngp_half", None)
    if c is not None and c[0] == embeddings._version and c[1] == embeddings.data_ptr() and c[2].device == embeddings.device:
        return c[2]
    half = embeddings.detach().to(torch.half)
    embeddings._ngp_half = (embeddings._version, embeddings.data_ptr(), half)
    return half


_MAX_ROWS = {}


def offsets_info(offsets):
    """(largest number of rows of any level, rows of the whole table, the offsets as a list): one host read per offsets tensor, cached on its
    storage pointer"""
    key = (offsets.data_ptr(), offsets.numel(), str(offsets.device))
    if key not in _MAX_ROWS:
        o = offsets.detach().cpu().to(torch.int64)
        _MAX_ROWS[key] = (int((o[1:] - o[:-1]).max()), int(o[-1]), [int(v) for v in o])
    return _MAX_ROWS[key]


def offsets_max_rows(offsets):
    return offsets_info(offsets)[0]


import os as _os

# table_gradient_binned(on_group=...): the table is summed in this many groups of levels, finest first (1 = one launch, the whole table handed over at
# the end).  Every extra group costs a launch of the summing kernel with its own tail (measured on one MI355X, forced single-rank RCCL, ms per steady /
# early step: 1 group 1.64 / 3.64, 2 groups 1.66 / 3.69, 4 groups 1.78 / 3.90) and hides that group's share of the all-reduce: two groups (levels 8-15,
# then 0-7) cost 0.02 ms and leave 34 % of the table (8.6 MB of halves) exposed.  This is only the DEFAULT of a caller that names no count: the
# data-parallel trainer passes its own, rank-checked value (ngp/train.py GradExchange.level_groups) -- a collective schedule must not depend on a
# per-process environment variable.
LEVEL_GROUPS = int(_os.environ.get("NGP_LEVEL_GROUPS", "2"))
PHASE_MAX_POINTS = 1 << 22             # ngp_grid_scatter_binned_phase takes one pass; larger batches go through the multi-pass entry point


def level_group_rows(offsets, L, groups):
    """[(row_lo, row_hi)] of the level groups in the order they are handed over (finest levels first).  A function of the level table and the group
    count ONLY: every rank of a data-parallel job posts the same all-reduces whatever its own batch looked like."""
    bounds = offsets_info(offsets)[2]
    groups = max(1, min(int(groups), L))
    cuts = [round(L * g / groups) for g in range(groups + 1)]
    return [(cuts[g], cuts[g + 1], int(bounds[cuts[g]]), int(bounds[cuts[g + 1]])) for g in range(groups - 1, -1, -1)]


def table_gradient_binned(grad, inputs, offsets, B, L, S, H, gridtype, align_corners, out_dtype=torch.float32, out_scale=1.0, on_group=None, groups=None,
                          listed=None):
    """grad [L,B,2] half (level-major), inputs [B,3] float32 in [0,1] -> the table gradient [sO,2] in `out_dtype`, summed on chip
    (ngp_grid_scatter_binned).  Shared by _grid_encode.backward and the field's native training step (ngp/field.py).
    listed = (list, count) device addresses: gradient row i belongs to the sample at inputs[list[i]], i < *count (the field's backward lists the samples
    that got a gradient, ngp_field_train_live_list); B <= PHASE_MAX_POINTS.
    on_group(out, row_lo, row_hi): called after the rows [row_lo, row_hi) of `out` (a group of levels, finest levels first, the few rows of the
    coarsest levels last) have been queued -- the data-parallel gradient exchange starts their all-reduce while the next group is summed.
    The SEQUENCE of on_group calls (row ranges, order) depends on (offsets, L, groups) only, never on B: B is a rank's own sample count, and ranks whose
    batches fall on different sides of a size limit must still post matching collectives (an empty batch hands over zeros; a batch beyond the one-pass
    limit is summed by the multi-pass entry point first and handed over in the same pieces)."""
    lib = _hip.lib()
    rows, n, bounds = offsets_info(offsets)
    out = torch.empty(n, 2, dtype=out_dtype, device=inputs.device)
    ws = _hip.workspace(lib.ngp_grid_scatter_binned_workspace(B, L), inputs.device)
    pieces = level_group_rows(offsets, L, LEVEL_GROUPS if groups is None else groups) if on_group is not None else None
    if listed is not None and not 0 < B <= PHASE_MAX_POINTS:
        raise ValueError("a listed scatter takes between 1 and 2^22 samples")
    who = (listed,) if listed is not None else ()              # the listed entry points take (list, count) in front of the workspace
    phase_fn = lib.ngp_grid_scatter_binned_phase_listed if listed is not None else lib.ngp_grid_scatter_binned_phase
    whole_fn = lib.ngp_grid_scatter_binned_listed if listed is not None else lib.ngp_grid_scatter_binned
    tail = (float(S), H, rows, gridtype, int(align_corners), _hip.dtype_code(out_dtype), float(out_scale), *(who[0] if who else ()), _hip.ptr(ws), ws.numel(),
            _hip.stream())
    if pieces is not None and len(pieces) > 1 and 0 < B <= PHASE_MAX_POINTS:
        args = (B, L)
        with _hip.timed("grid_encode_backward"):
            _hip.check(phase_fn(1, _hip.ptr(grad), _hip.ptr(inputs), _hip.ptr(offsets), _hip.ptr(out), *args, 0, L, *tail), "grid_scatter_binned_phase")
            for lo, hi, row_lo, row_hi in pieces:
                _hip.check(phase_fn(2, _hip.ptr(grad), _hip.ptr(inputs), _hip.ptr(offsets), _hip.ptr(out), *args, lo, hi, *tail), "grid_scatter_binned_phase")
                on_group(out, row_lo, row_hi)
        return out
    with _hip.timed("grid_encode_backward"):
        _hip.check(whole_fn(_hip.ptr(grad), _hip.ptr(inputs), _hip.ptr(offsets), _hip.ptr(out), B, L, *tail), "grid_scatter_binned")
    for lo, hi, row_lo, row_hi in pieces or ():
        on_group(out, row_lo, row_hi)
    return out


class _grid_encode(Function):
    """reference: gridencoder/grid.py:19-87"""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False):
        _hip.require_cuda(inputs, embeddings, offsets)
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)       # float64 here, narrowed to float at the C boundary like the reference
        H = base_resolution

        if torch.is_autocast_enabled() and C % 2 == 0:
            # "training" = this forward is differentiated w.r.t. the table; ctx.needs_input_grad[1] is True under no_grad as well
            training = torch.is_grad_enabled() and embeddings.requires_grad
            embeddings = _half_table(embeddings, training) if embeddings.dtype == torch.float32 else embeddings.to(torch.half)
        embeddings = embeddings.contiguous()
        if inputs.dtype != torch.float32:
            raise RuntimeError("inputs must be a float32 tensor")

        recompute = bool(calc_grad_inputs) and RECOMPUTE_INPUT_GRAD and B >= RECOMPUTE_MIN_POINTS and D in (2, 3) and C in (1, 2, 4, 8)
        if calc_grad_inputs and not recompute:
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype)
        else:
            dy_dx = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)

        # the row kernel stages 256 x (L + 1) feature pairs in LDS: it takes tables whose tile fits the default 64 KiB (csrc/gridencoder.hip)
        rows_fit = 256 * (L + 1) * 2 * embeddings.element_size() <= 65536
        if ROWS_FORWARD and _LEVEL_MAJOR_DEPTH == 0 and D == 3 and C == 2 and rows_fit and not (calc_grad_inputs and not recompute):
            # the reference's kernel writes [L, B, C] and grid.py:42,52 permutes + copies to [B, L*C]; this kernel writes the rows directly
            # (same bits): one launch and 2 x B x L x C values of traffic less per call
            outputs = torch.empty(B, L * C, device=inputs.device, dtype=embeddings.dtype)
            _hip.check(_hip.lib().ngp_grid_encode_forward_rows(_hip.ptr(inputs), _hip.ptr(embeddings), _hip.ptr(offsets), _hip.ptr(outputs),
                                                               B, D, C, L, float(S), H, gridtype, int(align_corners),
                                                               _hip.dtype_code(embeddings.dtype), _hip.stream()), "grid_encode_forward_rows")
        else:
            outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
            _hip.check(_hip.lib().ngp_grid_encode_forward(_hip.ptr(inputs), _hip.ptr(embeddings), _hip.ptr(offsets), _hip.ptr(outputs),
                                                          B, D, C, L, float(S), H, int(calc_grad_inputs and not recompute), _hip.ptr(dy_dx),
                                                          gridtype, int(align_corners), _hip.dtype_code(embeddings.dtype),
                                                          _hip.stream()), "grid_encode_forward")
            outputs = outputs.permute(1, 0, 2).reshape(B, L * C)

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype]
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.recompute = recompute
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        calc_grad_inputs = ctx.calc_grad_inputs

        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous().to(embeddings.dtype)
        # The reference always scatters into a dense zeros_like(embeddings) (grid.py:74).  A frozen model (the nav loop only
        # differentiates w.r.t. poses: ngp/nav.py) does not need it: skip the 25-50 MB fill and the 128 atomics per sample.
        need_table = ctx.needs_input_grad[1]
        if not need_table and not calc_grad_inputs:
            return None, None, None, None, None, None, None, None
        binned = (BINNED_SCATTER and need_table and D == 3 and C == 2 and embeddings.dtype == torch.float16
                  and int(offsets_max_rows(offsets)) <= (1 << 19))
        grad_embeddings = None
        if need_table:
            grad_embeddings = table_gradient_binned(grad, inputs, offsets, B, L, S, H, gridtype, ctx.align_corners) if binned else torch.zeros_like(embeddings)
        if calc_grad_inputs:
            grad_inputs = (torch.empty_like if ctx.recompute else torch.zeros_like)(inputs, dtype=embeddings.dtype)
        else:
            grad_inputs = torch.zeros(1, device=inputs.device, dtype=embeddings.dtype)

        if (need_table and not binned) or (calc_grad_inputs and not ctx.recompute):
            with _hip.timed("grid_encode_backward"):
                _hip.check(_hip.lib().ngp_grid_encode_backward(_hip.ptr(grad), _hip.ptr(inputs), _hip.ptr(embeddings), _hip.ptr(offsets),
                                                               _hip.ptr(grad_embeddings) if (need_table and not binned) else None, B, D, C, L, float(S), H,
                                                               int(calc_grad_inputs and not ctx.recompute),
                                                               _hip.ptr(dy_dx), _hip.ptr(grad_inputs), gridtype, int(ctx.align_corners),
                                                               _hip.dtype_code(embeddings.dtype), _hip.stream()), "grid_encode_backward")
        if ctx.recompute:
            _hip.check(_hip.lib().ngp_grid_encode_backward_inputs(_hip.ptr(grad), _hip.ptr(inputs), _hip.ptr(embeddings), _hip.ptr(offsets),
                                                                  B, D, C, L, float(S), H, _hip.ptr(grad_inputs), gridtype,
                                                                  int(ctx.align_corners), _hip.dtype_code(embeddings.dtype),
                                                                  _hip.stream()), "grid_encode_backward_inputs")

        if calc_grad_inputs:
            return grad_inputs.to(inputs.dtype), grad_embeddings, None, None, None, None, None, None
        return None, grad_embeddings, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    """reference: gridencoder/grid.py:93-156"""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))

        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.align_corners = align_corners

        # level table: rows per level = min(2^log2_hashmap_size, (res [+1])^D) rounded up to 8 (grid.py:113-123)
        offsets, offset = [], 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim

        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()
        # any backward that writes embeddings.grad is followed by an optimiser step: cached half copies of the table die here (see _half_table)
        self.embeddings.register_post_accumulate_grad_hook(drop_half_table)

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners}")

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)       # [-bound, bound] -> [0, 1]
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners)
        return outputs.view(prefix_shape + [self.output_dim])
