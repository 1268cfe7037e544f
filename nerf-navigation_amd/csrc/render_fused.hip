// render_fused.hip -- placeholder translation unit; replaced by the fused renderer.
#include "ngp_device.h"
extern "C" int ngp_field_forward(const ngp_field_t*, const float*, const float*, uint32_t, float*, float*, void*) {
    return ngp_fail(NGP_EINVAL, "field_forward: not built");
}
extern "C" size_t ngp_render_frame_workspace(uint32_t) { return 16; }
extern "C" int ngp_render_frame(const ngp_field_t*, const float*, const float*, uint32_t, const float*, float, const uint8_t*, uint32_t,
                                uint32_t, float, uint32_t, const float*, float*, float*, float*, uint32_t*, void*, size_t, void*) {
    return ngp_fail(NGP_EINVAL, "render_frame: not built");
}
extern "C" size_t ngp_ffmlp_backward_workspace(uint32_t, uint32_t, uint32_t, uint32_t) { return 16; }
extern "C" int ngp_ffmlp_backward(const void*, const void*, const void*, const void*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                                  uint32_t, uint32_t, int, void*, void*, void*, void*, size_t, void*) {
    return ngp_fail(NGP_EINVAL, "ffmlp_backward: not built");
}
