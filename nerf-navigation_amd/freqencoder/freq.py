"""Drop-in `freqencoder` package: `_freq_encoder` / `FreqEncoder` of the reference's freqencoder/freq.py, backed by
libngp_hip.so (csrc/freqencoder.hip).  Same surface: inputs cast to float32 (custom_fwd cast_inputs), output
[..., input_dim + 2 * input_dim * degree] laid out as [x | sin(2^f x), cos(2^f x) per frequency], analytic backward from the
saved outputs."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import ngp_hip as _hip


class _freq_encoder(Function):
    """reference: freqencoder/freq.py:15-52"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        _hip.require_cuda(inputs)
        inputs = inputs.contiguous()
        B, input_dim = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        _hip.check(_hip.lib().ngp_freq_encode_forward(_hip.ptr(inputs), B, input_dim, degree, output_dim, _hip.ptr(outputs), _hip.stream()),
                   "freq_encode_forward")
        ctx.save_for_backward(inputs, outputs)
        ctx.dims = [B, input_dim, degree, output_dim]
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        grad = grad.contiguous()
        inputs, outputs = ctx.saved_tensors
        B, input_dim, degree, output_dim = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _hip.check(_hip.lib().ngp_freq_encode_backward(_hip.ptr(grad), _hip.ptr(outputs), B, input_dim, degree, output_dim,
                                                       _hip.ptr(grad_inputs), _hip.stream()), "freq_encode_backward")
        return grad_inputs, None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    """reference: freqencoder/freq.py:58-83"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = freq_encode(inputs, self.degree, self.output_dim)
        return outputs.reshape(prefix_shape + [self.output_dim])
