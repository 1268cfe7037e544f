"""Drop-in `shencoder` package: `_sh_encoder` / `SHEncoder` of the reference's shencoder/sphere_harmonics.py, backed by
libngp_hip.so (csrc/shencoder.hip).  Inputs are forced to float32 like the reference (custom_fwd cast)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import ngp_hip as _hip


class _sh_encoder(Function):
    """reference: shencoder/sphere_harmonics.py:14-54"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        _hip.require_cuda(inputs)
        inputs = inputs.contiguous()
        B, input_dim = inputs.shape
        output_dim = degree ** 2
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        if calc_grad_inputs:
            dy_dx = torch.empty(B, input_dim * output_dim, dtype=inputs.dtype, device=inputs.device)
        else:
            dy_dx = torch.empty(1, dtype=inputs.dtype, device=inputs.device)
        _hip.check(_hip.lib().ngp_sh_encode_forward(_hip.ptr(inputs), _hip.ptr(outputs), B, input_dim, degree,
                                                    int(calc_grad_inputs), _hip.ptr(dy_dx), _hip.stream()), "sh_encode_forward")
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, degree]
        ctx.calc_grad_inputs = calc_grad_inputs
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        if not ctx.calc_grad_inputs:
            return None, None, None
        grad = grad.contiguous()
        inputs, dy_dx = ctx.saved_tensors
        B, input_dim, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _hip.check(_hip.lib().ngp_sh_encode_backward(_hip.ptr(grad), _hip.ptr(inputs), B, input_dim, degree, _hip.ptr(dy_dx),
                                                     _hip.ptr(grad_inputs), _hip.stream()), "sh_encode_backward")
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    """reference: shencoder/sphere_harmonics.py:61-86"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
