// ngp_device.h -- device/host helpers shared by the gfx950 kernels of libngp_hip.so.
//
// gfx950 only: wave = 64 lanes, no portability macros.  Everything here is
// compiled with -ffp-contract=off: the march/composite/encoder arithmetic must
// round exactly as written (DESIGN.md "Numerics").
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ngp_hip.h"

// ---------------------------------------------------------------------------
// host: error reporting
// ---------------------------------------------------------------------------

extern thread_local char ngp_err_buf[512];

static inline int ngp_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ngp_err_buf, sizeof(ngp_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

#define NGP_REQUIRE(cond, ...) \
    do { if (!(cond)) return ngp_fail(NGP_EINVAL, __VA_ARGS__); } while (0)

#define NGP_CHECK_LAUNCH(name) \
    do { hipError_t e_ = hipGetLastError(); \
         if (e_ != hipSuccess) return ngp_fail(NGP_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

static inline uint32_t ngp_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ---------------------------------------------------------------------------
// device: scalar helpers
// ---------------------------------------------------------------------------

// binary32 -> binary16 of a value that may be the result of a multiplication.  Left alone, the compiler merges
// `(_Float16)(a * b)` into v_fma_mixlo/mixhi_f16, and that instruction rounds the exact product ONCE to binary16
// (measured on gfx950: different halves in ~1e-6 of the products, tools/cmp_frames.py), while the reference arithmetic
// (a binary32 product stored to at::Half) rounds twice.  Pinning the binary32 value in a register keeps the two roundings.
__device__ __forceinline__ _Float16 ngp_f2h(float v) {
    asm("" : "+v"(v));
    return (_Float16)v;
}

__device__ __forceinline__ float ngp_clampf(float x, float lo, float hi) {
    return fminf(hi, fmaxf(lo, x));                 // reference clamp(): raymarching.cu:36-38
}

// Deterministic exp standing in for the reference's __expf (raymarching.cu:547,650,869): identical, operation
// for operation, to o_expf() of the oracle.  Two-term Cody-Waite reduction and a degree-7 Horner chain, all fmaf.
__device__ __forceinline__ float ngp_expf(float x) {
    if (!(x >= -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.0f) return __builtin_inff();
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723e-06f, r);
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return __int_as_float(__float_as_int(p) + (((int)n) << 23));
}

// Deterministic sine standing in for the reference's __sinf (freqencoder.cu:57), identical, operation for operation, to
// o_sinf() of the oracle: octant reduction with a three-part pi/4 (Cody-Waite), then the degree-7 sine or degree-8 cosine
// minimax polynomial on [-pi/4, pi/4]; plain multiplies and adds (the build never contracts them).  |x| < 8192.
__device__ __forceinline__ float ngp_sinf(float x) {
    float ax = __builtin_fabsf(x);
    const bool neg = x < 0.0f;
    float y = __builtin_floorf(ax * 1.27323954473516f);          // 4 / pi
    int j = (int)y;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    bool flip = neg;
    if (j > 3) { flip = !flip; j -= 4; }
    const float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    float v;
    if (j == 1 || j == 2)
        v = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    else
        v = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    return flip ? -v : v;
}

// 10-bit-per-axis Morton interleave (reference: raymarching.cu:58-83)
__device__ __forceinline__ uint32_t ngp_spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t ngp_morton3(uint32_t x, uint32_t y, uint32_t z) {
    return ngp_spread3(x) | (ngp_spread3(y) << 1) | (ngp_spread3(z) << 2);
}
__device__ __forceinline__ uint32_t ngp_compact3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// PCG-XSH-RR 64/32 with Brown's jump-ahead (reference: raymarching/src/pcg32.h:44-205)
struct ngp_pcg32 {
    uint64_t state, inc;
    static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;
    __host__ __device__ uint32_t next_uint() {
        const uint64_t s = state;
        state = s * MULT + inc;
        const uint32_t xs = (uint32_t)(((s >> 18u) ^ s) >> 27u);
        const uint32_t rot = (uint32_t)(s >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
    }
    __host__ __device__ void seed(uint64_t initstate, uint64_t initseq = 1u) {
        state = 0u;
        inc = (initseq << 1u) | 1u;
        next_uint();
        state += initstate;
        next_uint();
    }
    __host__ __device__ void advance(uint64_t delta) {
        uint64_t cm = MULT, cp = inc, am = 1u, ap = 0u;
        while (delta > 0) {
            if (delta & 1u) { am *= cm; ap = ap * cm + cp; }
            cp = (cm + 1u) * cp;
            cm *= cm;
            delta >>= 1;
        }
        state = am * state + ap;
    }
    __device__ float next_float() {
        return __uint_as_float((next_uint() >> 9) | 0x3f800000u) - 1.0f;
    }
};

// ray / AABB slab test (reference: raymarching.cu:109-146); miss => FLT_MAX for both
__device__ __forceinline__ void ngp_near_far_inline(const float* o, const float* d, const float* aabb, float min_near,
                                                    float& near, float& far) {
    const float rdx = 1.0f / d[0], rdy = 1.0f / d[1], rdz = 1.0f / d[2];
    float tn = (aabb[0] - o[0]) * rdx, tf = (aabb[3] - o[0]) * rdx;
    if (tn > tf) { const float s = tn; tn = tf; tf = s; }
    float yn = (aabb[1] - o[1]) * rdy, yf = (aabb[4] - o[1]) * rdy;
    if (yn > yf) { const float s = yn; yn = yf; yf = s; }
    if (tn > yf || yn > tf) { near = far = 3.402823466e+38f; return; }
    if (yn > tn) tn = yn;
    if (yf < tf) tf = yf;
    float zn = (aabb[2] - o[2]) * rdz, zf = (aabb[5] - o[2]) * rdz;
    if (zn > zf) { const float s = zn; zn = zf; zf = s; }
    if (tn > zf || zn > tf) { near = far = 3.402823466e+38f; return; }
    if (zn > tn) tn = zn;
    if (zf < tf) tf = zf;
    if (tn < min_near) tn = min_near;
    near = tn; far = tf;
}

// ---------------------------------------------------------------------------
// device: the occupancy march step shared by march_rays_train, march_rays and the fused renderer
// (reference: raymarching.cu:363-404, 431-483, 759-813)
// ---------------------------------------------------------------------------

#define NGP_SKIP_GUARD 65536   // bound on the empty-space substep loop (the reference's is unbounded: :400-402)

// (ngp_march_t, the per-ray march state of the per-op kernels, lives in ngp_march.h)
