// _gridencoder : gridencoder/src/gridencoder.h:12-13, bindings.cpp:6-7
#include "shim_common.h"
using namespace shim;

static int table_dtype(const at::Tensor& embeddings) {
    TORCH_CHECK(embeddings.scalar_type() == at::kFloat || embeddings.scalar_type() == at::kHalf, "embeddings must be float32 or float16");
    return embeddings.scalar_type() == at::kHalf ? NGP_F16 : NGP_F32;
}

void grid_encode_forward(const at::Tensor inputs, const at::Tensor embeddings, const at::Tensor offsets, at::Tensor outputs, const uint32_t B,
                         const uint32_t D, const uint32_t C, const uint32_t L, const float S, const uint32_t H, const bool calc_grad_inputs,
                         at::Tensor dy_dx, const uint32_t gridtype, const bool align_corners) {
    on_gpu(inputs, "inputs"); on_gpu(embeddings, "embeddings"); on_gpu(offsets, "offsets"); on_gpu(outputs, "outputs");       // gridencoder.cu:424-440
    TORCH_CHECK(inputs.scalar_type() == at::kFloat, "inputs must be float32");
    TORCH_CHECK(offsets.scalar_type() == at::kInt, "offsets must be int32");
    TORCH_CHECK(outputs.scalar_type() == embeddings.scalar_type(), "outputs must have the table's dtype");
    device_guard g(inputs.device());
    ok(ngp_grid_encode_forward(ptr<float>(inputs), embeddings.data_ptr(), ptr<int32_t>(offsets), outputs.data_ptr(), B, D, C, L, S, H,
                               calc_grad_inputs ? 1 : 0, calc_grad_inputs ? dy_dx.data_ptr() : nullptr, gridtype, align_corners ? 1 : 0,
                               table_dtype(embeddings), stream_of(inputs)), "grid_encode_forward");
}

void grid_encode_backward(const at::Tensor grad, const at::Tensor inputs, const at::Tensor embeddings, const at::Tensor offsets,
                          at::Tensor grad_embeddings, const uint32_t B, const uint32_t D, const uint32_t C, const uint32_t L, const float S,
                          const uint32_t H, const bool calc_grad_inputs, const at::Tensor dy_dx, at::Tensor grad_inputs, const uint32_t gridtype,
                          const bool align_corners) {
    on_gpu(grad, "grad"); on_gpu(inputs, "inputs"); on_gpu(embeddings, "embeddings"); on_gpu(offsets, "offsets"); on_gpu(grad_embeddings, "grad_embeddings");
    TORCH_CHECK(grad.scalar_type() == embeddings.scalar_type() && grad_embeddings.scalar_type() == embeddings.scalar_type(),
                "grad and grad_embeddings must have the table's dtype");
    device_guard g(inputs.device());
    ok(ngp_grid_encode_backward(grad.data_ptr(), ptr<float>(inputs), embeddings.data_ptr(), ptr<int32_t>(offsets), grad_embeddings.data_ptr(), B, D, C, L,
                                S, H, calc_grad_inputs ? 1 : 0, calc_grad_inputs ? dy_dx.data_ptr() : nullptr,
                                calc_grad_inputs ? grad_inputs.data_ptr() : nullptr, gridtype, align_corners ? 1 : 0, table_dtype(embeddings),
                                stream_of(inputs)), "grid_encode_backward");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("grid_encode_forward", &grid_encode_forward, "grid_encode_forward (gfx950)");
    m.def("grid_encode_backward", &grid_encode_backward, "grid_encode_backward (gfx950)");
}
