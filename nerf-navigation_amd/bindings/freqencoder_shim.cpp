// _freqencoder : freqencoder/src/freqencoder.h:7,10, bindings.cpp:6-7
#include "shim_common.h"
using namespace shim;

void freq_encode_forward(at::Tensor inputs, const uint32_t B, const uint32_t D, const uint32_t deg, const uint32_t C, at::Tensor outputs) {
    on_gpu(inputs, "inputs"); on_gpu(outputs, "outputs");
    TORCH_CHECK(inputs.scalar_type() == at::kFloat && outputs.scalar_type() == at::kFloat, "inputs / outputs must be float32");
    device_guard g(inputs.device());
    ok(ngp_freq_encode_forward(ptr<float>(inputs), B, D, deg, C, ptr<float>(outputs), stream_of(inputs)), "freq_encode_forward");
}

void freq_encode_backward(at::Tensor grad, at::Tensor outputs, const uint32_t B, const uint32_t D, const uint32_t deg, const uint32_t C,
                          at::Tensor grad_inputs) {
    on_gpu(grad, "grad"); on_gpu(outputs, "outputs"); on_gpu(grad_inputs, "grad_inputs");
    device_guard g(grad.device());
    ok(ngp_freq_encode_backward(ptr<float>(grad), ptr<float>(outputs), B, D, deg, C, ptr<float>(grad_inputs), stream_of(grad)), "freq_encode_backward");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("freq_encode_forward", &freq_encode_forward, "freq encode forward (gfx950)");
    m.def("freq_encode_backward", &freq_encode_backward, "freq encode backward (gfx950)");
}
