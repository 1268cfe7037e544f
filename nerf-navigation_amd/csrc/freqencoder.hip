// freqencoder.hip -- frequency (positional) encoding, the reference's optional fifth native module
// (freqencoder/src/freqencoder.cu:27-140; reached through encoding.get_encoder('frequency'), encoding.py:56-58).
//   outputs[b] = [ x (D) | for f < deg: sin(2^f x) (D), sin(2^f x + pi/2) (D) ]          C = D + 2 D deg
//   grad_inputs[b,d] = grad[d] + sum_f 2^f (grad_sin * out_cos - grad_cos * out_sin)
// One lane per output element (forward) / per input element (backward), float32 as the reference forces
// (freq.py:17 custom_fwd(cast_inputs=float32)).  __sinf is a CUDA fast-math intrinsic; both this kernel and the oracle use
// the same deterministic ngp_sinf, so outputs are bit-exact against the oracle.
#include "ngp_device.h"

__global__ __launch_bounds__(256) void k_freq_forward(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                                      float* __restrict__ outputs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)B * C) return;
    const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
    const float* in = inputs + (uint64_t)b * D;
    if (c < D) { outputs[t] = in[c]; return; }
    const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
    const float phase_shift = (float)(col % 2) * (3.141592653589793f / 2);
    outputs[t] = ngp_sinf(__builtin_ldexpf(in[d], (int)freq) + phase_shift);
}

__global__ __launch_bounds__(256) void k_freq_backward(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B, uint32_t D,
                                                       uint32_t deg, uint32_t C, float* __restrict__ grad_inputs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)B * D) return;
    const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
    const float* g = grad + (uint64_t)b * C;
    const float* o = outputs + (uint64_t)b * C;
    float result = g[d];
    g += D; o += D;
    for (uint32_t f = 0; f < deg; f++) {
        result += __builtin_ldexpf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
        g += 2 * D; o += 2 * D;
    }
    grad_inputs[t] = result;
}

extern "C" int ngp_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs, void* stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && outputs, "freq_encode_forward: null pointer");
    NGP_REQUIRE(D >= 1 && C == D + 2 * D * deg, "freq_encode_forward: output_dim must be input_dim + 2 * input_dim * degree");
    NGP_REQUIRE(deg <= 24, "freq_encode_forward: degree must be <= 24");
    const uint64_t n = (uint64_t)B * C;
    hipLaunchKernelGGL(k_freq_forward, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inputs, B, D, deg, C, outputs);
    NGP_CHECK_LAUNCH("freq_encode_forward");
    return NGP_OK;
}

extern "C" int ngp_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                        float* grad_inputs, void* stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && outputs && grad_inputs, "freq_encode_backward: null pointer");
    NGP_REQUIRE(D >= 1 && C == D + 2 * D * deg, "freq_encode_backward: output_dim must be input_dim + 2 * input_dim * degree");
    const uint64_t n = (uint64_t)B * D;
    hipLaunchKernelGGL(k_freq_backward, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad, outputs, B, D, deg, C,
                       grad_inputs);
    NGP_CHECK_LAUNCH("freq_encode_backward");
    return NGP_OK;
}
