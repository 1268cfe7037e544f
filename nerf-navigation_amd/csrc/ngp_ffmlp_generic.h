// ngp_ffmlp_generic.h -- the layer-by-layer FFMLP for every shape and activation the reference's module accepts (ffmlp/ffmlp.py:100-117,
// ffmlp/src/ffmlp.cu:652-659): hidden_dim 16 / 32 / 64 / 128 / 256, input_dim any multiple of 16 (up to 256), any num_layers >= 2, the seven
// activations of ffmlp/src/utils.h:29-37.  ffmlp.hip / ffmlp_backward.hip use their register-resident kernels for the shapes the reference's models
// have (width 64, ReLU, 2-4 layers) and these for everything else.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

bool ffmlp_fast_shape(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation);
int ffmlp_generic_check(const char* who, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                        uint32_t activation, uint32_t output_activation);
// buffer: [num_layers][B][hidden] halves (the reference's forward / inference buffer: written in both modes)
int ffmlp_generic_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                          uint32_t num_layers, uint32_t activation, void* buffer, void* outputs, hipStream_t s);
size_t ffmlp_generic_backward_workspace(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers);
int ffmlp_generic_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B, uint32_t input_dim,
                           uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, int calc_grad_inputs,
                           void* backward_buffer, void* grad_inputs, void* grad_weights, void* workspace, size_t workspace_bytes, hipStream_t s);

// Weight gradients without float atomics (bitwise reproducible): workgroup column x of a weight-gradient launch stores its partial sums as row x of a
// [rows][nw] f32 buffer at the head of the workspace; ffmlp_sum_partials adds the rows in a fixed order and rounds to half.
uint32_t ffmlp_partial_rows(uint32_t nw);                                  // rows the workspace holds (<= 256, <= 64 MiB of partials)
size_t ffmlp_partial_bytes(uint32_t nw);                                   // the buffer's size, 256-byte aligned
void ffmlp_sum_partials(const float* partials, uint32_t rows, uint32_t nw, void* grad_weights_half, hipStream_t s);
