"""Drop-in `ffmlp` package: `_ffmlp_forward` / `FFMLP` of the reference's ffmlp/ffmlp.py, backed by libngp_hip.so
(csrc/ffmlp.hip: v_mfma_f32_16x16x32_f16, weights in registers, f32 accumulation).

Same surface: inputs/weights cast to half (custom_fwd cast_inputs), batch padded up past the next multiple of 128,
output padded to 16 columns and sliced back, flat `weights` parameter laid out as the concatenated nn.Linear weights
(ffmlp.cu:631-634), U(+-sqrt(3/hidden)) init under torch.manual_seed(42) (ffmlp.py:141-144).
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import ngp_hip as _hip


class _ffmlp_forward(Function):
    """reference: ffmlp/ffmlp.py:15-83"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False):
        _hip.require_cuda(inputs, weights)
        if inputs.dtype != torch.half or weights.dtype != torch.half:
            raise RuntimeError("inputs must be a half tensor (run FFMLP under autocast, as the reference requires)")
        B = inputs.shape[0]
        inputs = inputs.contiguous()
        weights = weights.contiguous()
        outputs = torch.empty(B, output_dim, device=inputs.device, dtype=inputs.dtype)
        L = _hip.lib()
        if not inference:
            forward_buffer = torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
            _hip.check(L.ngp_ffmlp_forward(_hip.ptr(inputs), _hip.ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                           activation, output_activation, _hip.ptr(forward_buffer), _hip.ptr(outputs),
                                           _hip.stream()), "ffmlp_forward")
            ctx.save_for_backward(inputs, weights, outputs, forward_buffer)
            ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
        else:
            # the register-resident kernels (width 64, ReLU, 2-4 layers, input_dim <= 64) keep their activations in registers; every other shape
            # runs layer by layer and needs the reference's inference buffer (ffmlp.py:40-41)
            fast = hidden_dim == 64 and activation == 0 and output_activation == 6 and 2 <= num_layers <= 4 and input_dim <= 64
            buf = None if fast else torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
            _hip.check(L.ngp_ffmlp_inference(_hip.ptr(inputs), _hip.ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                             activation, output_activation, _hip.ptr(buf), _hip.ptr(outputs), _hip.stream()),
                       "ffmlp_inference")
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        B = grad.shape[0]
        grad = grad.contiguous()
        inputs, weights, outputs, forward_buffer = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.dims
        # The reference zero-fills grad_inputs and the [num_layers, B, hidden] backward_buffer on every call (ffmlp.py:66-73:
        # ~1.6 GB of writes per training step here).  The kernel writes every element of both, so they are only allocated.
        if calc_grad_inputs:
            grad_inputs = torch.empty_like(inputs)
        else:
            grad_inputs = torch.zeros(1, device=grad.device, dtype=grad.dtype)
        grad_weights = torch.zeros_like(weights)
        backward_buffer = torch.empty(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
        L = _hip.lib()
        ws = _hip.workspace(L.ngp_ffmlp_backward_workspace(input_dim, output_dim, hidden_dim, num_layers), grad.device)
        _hip.check(L.ngp_ffmlp_backward(_hip.ptr(grad), _hip.ptr(inputs), _hip.ptr(weights), _hip.ptr(forward_buffer), B, input_dim,
                                        output_dim, hidden_dim, num_layers, activation, output_activation, int(calc_grad_inputs),
                                        _hip.ptr(backward_buffer), _hip.ptr(grad_inputs), _hip.ptr(grad_weights),
                                        _hip.ptr(ws), ws.numel(), _hip.stream()), "ffmlp_backward")
        if calc_grad_inputs:
            return grad_inputs, grad_weights, None, None, None, None, None, None, None, None
        return None, grad_weights, None, None, None, None, None, None, None, None


ffmlp_forward = _ffmlp_forward.apply


def convert_activation(act):
    """reference: ffmlp/ffmlp.py:89-96"""
    return {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}.get(act, 6)


class FFMLP(nn.Module):
    """reference: ffmlp/ffmlp.py:99-168"""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16

        # the reference's own limits (ffmlp.py:110-113); libngp_hip runs width 64 / ReLU / 2-4 layers / input_dim <= 64 (every model of the
        # reference) on register-resident kernels and everything else layer by layer (csrc/ffmlp_generic.hip)
        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert input_dim <= 256, f"FFMLP (gfx950) input_dim is limited to 256, but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"
        assert num_layers <= 16, f"FFMLP (gfx950) num_layers is limited to 16, but got {num_layers}"

        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()
        _hip.check(_hip.lib().ngp_allocate_splitk(self.num_layers + 1), "allocate_splitk")

    def cleanup(self):
        _hip.check(_hip.lib().ngp_free_splitk(), "free_splitk")

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)
        std = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-std, std)

    def forward(self, inputs):
        B, C = inputs.shape
        # The reference pads the batch past the next multiple of 128 with a torch.cat (ffmlp.py:157-159; a full 128 rows when B % 128 == 0)
        # because its kernel works on 128-row blocks.  These kernels need a multiple of 32 only (forward 16, backward 32): batches that
        # already are (every batch the march produces: align = 128) go in as they are -- the rows returned are the same, without
        # copying the whole input.
        pad = (32 - (B % 32)) % 32
        if pad > 0:
            inputs = torch.cat([inputs, torch.zeros(pad, C, dtype=inputs.dtype, device=inputs.device)], dim=0)
        outputs = ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                                self.activation, self.output_activation, not self.training, inputs.requires_grad)
        if B != outputs.shape[0] or self.padded_output_dim != self.output_dim:
            outputs = outputs[:B, :self.output_dim]
        return outputs
