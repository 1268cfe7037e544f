// shim_common.h -- shared by the Level-2 native modules (_raymarching, _gridencoder, _shencoder, _ffmlp, _freqencoder):
// pybind11 modules with the reference's function names and argument lists (raymarching/src/bindings.cpp:7-18 and the same
// file in every package) that forward at::Tensor arguments, as device pointers, to the C ABI of include/ngp_hip.h.
// Plain C++ (no .cu / .hip source, nothing hipified): every kernel lives in libngp_hip.so.
#pragma once
#include <torch/extension.h>
// PyTorch-ROCm presents its HIP devices under the device type "cuda"; the guard / stream classes for that are the *MasqueradingAsCUDA ones
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "../../include/ngp_hip.h"

namespace shim {

// torch's current stream on the tensor's device (the reference launched on the legacy default stream: SURVEY 8b)
inline void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

using device_guard = c10::hip::HIPGuardMasqueradingAsCUDA;

inline void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", ngp_last_error()); }

inline void on_gpu(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor (libngp_hip has no CPU path)");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

// dtype + size validation: the reference's raymarching entry points validate nothing and dispatch on the scalar type
// (AT_DISPATCH_FLOATING_TYPES_AND_HALF, raymarching.cu:152-158 ...); these kernels are float32 / int32 / uint8 only, so anything else must be an
// error here rather than reinterpreted memory
inline void need(const at::Tensor& t, const char* name, at::ScalarType st, int64_t min_numel) {
    on_gpu(t, name);
    TORCH_CHECK(t.scalar_type() == st, name, " must be ", st, " (got ", t.scalar_type(), "); cast it, as the Python wrappers' custom_fwd(cast_inputs=float32) does");
    TORCH_CHECK(t.numel() >= min_numel, name, " has ", t.numel(), " elements, this call reads or writes ", min_numel);
}
inline void need_f32(const at::Tensor& t, const char* name, int64_t n) { need(t, name, at::kFloat, n); }
inline void need_i32(const at::Tensor& t, const char* name, int64_t n) { need(t, name, at::kInt, n); }
inline void need_u8(const at::Tensor& t, const char* name, int64_t n) { need(t, name, at::kByte, n); }

template <typename T> inline T* ptr(const at::Tensor& t) { return t.defined() && t.numel() ? (T*)t.data_ptr() : (T*)nullptr; }

inline at::Tensor bytes_like(const at::Tensor& t, size_t n) {
    return at::empty({(int64_t)(n ? n : 16)}, t.options().dtype(at::kByte));
}

}  // namespace shim
