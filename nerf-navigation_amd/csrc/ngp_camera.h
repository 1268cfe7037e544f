// Pinhole camera -> ray (reference: get_rays, nerf/utils.py:53-116, the arithmetic of :98-108), shared by k_get_rays and
// the fused frame kernel's camera mode.  Operation order is part of the contract (the oracle restates it in binary32):
//   xs = ((col + 0.5) - cx) / fx            ys = ((row + 0.5) - cy) / fy             zs = 1
//   n  = sqrt((xs*xs + ys*ys) + 1)          dir = (xs / n, ys / n, 1 / n)
//   rays_d[k] = (dir.x * R[k][0] + dir.y * R[k][1]) + dir.z * R[k][2]               rays_o = T
// (torch evaluates the same formula; its norm and matmul may associate differently: equal to ~1 ulp, not bit for bit.)
#pragma once
#include <stdint.h>

struct ngp_camera {
    float r[9];                      // rotation, row-major: poses[:3, :3]
    float t[3];                      // translation: poses[:3, 3]
    float fx, fy, cx, cy;
    uint32_t W, H;
};

__device__ __forceinline__ void ngp_camera_ray(const ngp_camera& c, uint32_t pixel, float d[3]) {
    const uint32_t row = pixel / c.W, col = pixel - row * c.W;
    const float xs = (((float)col + 0.5f) - c.cx) / c.fx;
    const float ys = (((float)row + 0.5f) - c.cy) / c.fy;
    const float n = sqrtf((xs * xs + ys * ys) + 1.0f);
    const float ux = xs / n, uy = ys / n, uz = 1.0f / n;
    #pragma unroll
    for (int k = 0; k < 3; k++) d[k] = (ux * c.r[3 * k] + uy * c.r[3 * k + 1]) + uz * c.r[3 * k + 2];
}
