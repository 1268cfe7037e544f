// _ffmlp : ffmlp/src/ffmlp.h:8-14, bindings.cpp:6-10
#include "shim_common.h"
using namespace shim;

static void check_half(const at::Tensor& t, const char* name) {                                   // ffmlp.cu:636-642
    on_gpu(t, name);
    TORCH_CHECK(t.scalar_type() == at::kHalf, name, " must be float16");
}

void ffmlp_forward(const at::Tensor inputs, const at::Tensor weights, const uint32_t B, const uint32_t input_dim, const uint32_t output_dim,
                   const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation_, const uint32_t output_activation_,
                   at::Tensor forward_buffer, at::Tensor outputs) {
    check_half(inputs, "inputs"); check_half(weights, "weights"); check_half(forward_buffer, "forward_buffer"); check_half(outputs, "outputs");
    device_guard g(inputs.device());
    ok(ngp_ffmlp_forward(inputs.data_ptr(), weights.data_ptr(), B, input_dim, output_dim, hidden_dim, num_layers, activation_, output_activation_,
                         forward_buffer.data_ptr(), outputs.data_ptr(), stream_of(inputs)), "ffmlp_forward");
}

void ffmlp_inference(const at::Tensor inputs, const at::Tensor weights, const uint32_t B, const uint32_t input_dim, const uint32_t output_dim,
                     const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation_, const uint32_t output_activation_,
                     at::Tensor inference_buffer, at::Tensor outputs) {
    check_half(inputs, "inputs"); check_half(weights, "weights"); check_half(outputs, "outputs");
    device_guard g(inputs.device());
    ok(ngp_ffmlp_inference(inputs.data_ptr(), weights.data_ptr(), B, input_dim, output_dim, hidden_dim, num_layers, activation_, output_activation_,
                           inference_buffer.defined() ? inference_buffer.data_ptr() : nullptr, outputs.data_ptr(), stream_of(inputs)), "ffmlp_inference");
}

void ffmlp_backward(const at::Tensor grad, const at::Tensor inputs, const at::Tensor weights, const at::Tensor forward_buffer, const uint32_t B,
                    const uint32_t input_dim, const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers, const uint32_t activation,
                    const uint32_t output_activation, const bool calc_grad_inputs, at::Tensor backward_buffer, at::Tensor grad_inputs,
                    at::Tensor grad_weights) {
    check_half(grad, "grad"); check_half(inputs, "inputs"); check_half(weights, "weights"); check_half(forward_buffer, "forward_buffer");
    check_half(backward_buffer, "backward_buffer"); check_half(grad_weights, "grad_weights");
    device_guard g(inputs.device());
    at::Tensor ws = bytes_like(inputs, ngp_ffmlp_backward_workspace(input_dim, output_dim, hidden_dim, num_layers));   // the reference's GPUMemory scratch
    ok(ngp_ffmlp_backward(grad.data_ptr(), inputs.data_ptr(), weights.data_ptr(), forward_buffer.data_ptr(), B, input_dim, output_dim, hidden_dim, num_layers,
                          activation, output_activation, calc_grad_inputs ? 1 : 0, backward_buffer.data_ptr(),
                          calc_grad_inputs ? grad_inputs.data_ptr() : nullptr, grad_weights.data_ptr(), ws.data_ptr(), (size_t)ws.numel(),
                          stream_of(inputs)), "ffmlp_backward");
}

void allocate_splitk(size_t size) { ok(ngp_allocate_splitk(size), "allocate_splitk"); }
void free_splitk() { ok(ngp_free_splitk(), "free_splitk"); }

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("ffmlp_forward", &ffmlp_forward, "ffmlp_forward (gfx950)");
    m.def("ffmlp_inference", &ffmlp_inference, "ffmlp_inference (gfx950)");
    m.def("ffmlp_backward", &ffmlp_backward, "ffmlp_backward (gfx950)");
    m.def("allocate_splitk", &allocate_splitk, "allocate_splitk (no-op)");
    m.def("free_splitk", &free_splitk, "free_splitk (no-op)");
}
