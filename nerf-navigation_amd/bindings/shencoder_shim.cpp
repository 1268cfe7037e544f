// _shencoder : shencoder/src/shencoder.h:10,13, bindings.cpp:6-7
#include "shim_common.h"
using namespace shim;

void sh_encode_forward(at::Tensor inputs, at::Tensor outputs, const uint32_t B, const uint32_t D, const uint32_t C, const bool calc_grad_inputs,
                       at::Tensor dy_dx) {
    on_gpu(inputs, "inputs"); on_gpu(outputs, "outputs");                                         // shencoder.cu:403-413
    TORCH_CHECK(inputs.scalar_type() == at::kFloat && outputs.scalar_type() == at::kFloat, "inputs / outputs must be float32");
    device_guard g(inputs.device());
    ok(ngp_sh_encode_forward(ptr<float>(inputs), ptr<float>(outputs), B, D, C, calc_grad_inputs ? 1 : 0, calc_grad_inputs ? ptr<float>(dy_dx) : nullptr,
                             stream_of(inputs)), "sh_encode_forward");
}

void sh_encode_backward(at::Tensor grad, at::Tensor inputs, const uint32_t B, const uint32_t D, const uint32_t C, at::Tensor dy_dx, at::Tensor grad_inputs) {
    on_gpu(grad, "grad"); on_gpu(inputs, "inputs"); on_gpu(dy_dx, "dy_dx"); on_gpu(grad_inputs, "grad_inputs");
    device_guard g(inputs.device());
    ok(ngp_sh_encode_backward(ptr<float>(grad), ptr<float>(inputs), B, D, C, ptr<float>(dy_dx), ptr<float>(grad_inputs), stream_of(inputs)),
       "sh_encode_backward");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("sh_encode_forward", &sh_encode_forward, "SH encode forward (gfx950)");
    m.def("sh_encode_backward", &sh_encode_backward, "SH encode backward (gfx950)");
}
