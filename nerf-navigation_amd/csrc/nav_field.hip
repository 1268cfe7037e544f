// nav_field.hip -- the navigation loop's queries (BASELINE config 4) as fused float32 gfx950 kernels.
//
// The reference's planner and pose filter (simulate.py:340-347, nav/quad_plot.py:224-250, nav/estimator_helpers.py:293-327) call the
// DEFAULT field (nerf/network.py: hash grid -> Linear(32,64) -> Linear(64,16) ; SH(16) ++ geo(15) -> Linear(31,64) -> Linear(64,64) ->
// Linear(64,3), no biases) in float32, without autocast, through torch autograd:
//     density_fn(x)          = exp(h0(x))                                   with d sigma / d x
//     render_fn(rays)        = NeRFRenderer.run(num_steps = 512, upsample_steps = 0)      with d image / d rays_o, d rays_d
// Per filter iteration that is ~130 launches (encoder, 5 rocBLAS GEMMs and their 8 backward GEMMs, ~60 elementwise kernels over
// [1024, 512] tensors).  Here: one launch forward, one launch backward.
//
//   k_nav_density_fwd / _bwd   one lane per point: 16 x 8 float2 gathers, the 32-64-16 MLP on the VALU with the weights read through
//                              the scalar cache (every lane of a wave multiplies by the same weight: an SGPR operand, no LDS, no VGPR)
//   k_nav_run_fwd / _bwd       one 256-thread workgroup per ray, samples in chunks of 256: positions, density, the alpha / cumprod
//                              weights of nerf/renderer.py:206-210 by a block scan, colour only where weights > 1e-4 (:217), the
//                              image / depth / weights_sum reductions; the backward recomputes the forward, derives d L / d sigma_i
//                              analytically (prefix product + suffix sum), and runs colour-net and density-net backward per sample down
//                              to d L / d xyz, reduced into d L / d rays_o and d L / d rays_d.
// float32 FMA throughput is the same on the VALU and on the f32 MFMA of this chip, and the batch is small (10 k - 524 k points), so
// the matrix cores would buy nothing here; what the fusion removes is launches and the [M, 64] round trips through HBM.
//
// Arithmetic: float32 like the reference path, but sums are fused multiply-adds in a fixed order of their own (rocBLAS has its own):
// results agree with the torch path to float32 rounding (tests/test_gpu_nav_native.py states the tolerances against the CPU oracle).
#include <atomic>
#include "ngp_sh.h"

#ifndef NV_RUN_U
#define NV_RUN_U 1                         // levels of the hash grid the run kernels may overlap (gathers in flight per lane: 8 x U); A/B: profiles/HISTORY.md 4.5
#endif
static constexpr int NV_L = 16;            // levels
static constexpr int NV_H = 64;            // hidden width
static constexpr int NV_GEO = 15;
static constexpr int NV_CIN = 31;          // 16 SH + 15 geo
static constexpr uint32_t NV_BLOCK = 256;
#define NV_FENCE() __builtin_amdgcn_sched_barrier(0)

struct nav_levels {
    float scale[NV_L];
    uint32_t base[NV_L], size[NV_L], mask[NV_L], s1[NV_L], s2[NV_L];     // rows; mask = size - 1 when size is 2^k else 0; s1 == 0: hashed
};

// Weights are read through the CONSTANT address space: a load from it is invariant by definition, so a wave-uniform address always
// becomes a scalar load (s_load_dwordx8/x16 into SGPRs) -- also behind the barriers of the run kernels, where the compiler must assume
// that ordinary global memory may have changed and would otherwise fetch every weight per lane into a VGPR.
typedef const float __attribute__((address_space(4)))* nv_wptr;

// Constant-space loads are speculatable, so left alone the optimiser hoists EVERY weight load of a kernel to its entry (out of the
// `weights > 1e-4` branch, out of the chunk loops): ~9,000 live SGPRs, spilled lane by lane into VGPRs (15,000 v_readlane in
// k_nav_run_bwd).  Passing the pointer through an empty volatile asm where a layer starts pins that layer's loads inside it.
__device__ __forceinline__ nv_wptr nv_hide(nv_wptr p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (nv_wptr)v;
}

struct nav_params {
    const float2* table;                   // [sO] rows of 2 float32 features
    nv_wptr w1t;                           // [32][64]  sigma_net.0.weight transposed
    nv_wptr w2, w2t;                       // [16][64]  sigma_net.1.weight, [64][16] its transpose
    nv_wptr v0, v0t;                       // [64][31]  color_net.0.weight, [31][64]
    nv_wptr v1, v1t;                       // [64][64]  color_net.1.weight and its transpose
    nv_wptr v2, v2t;                       // [3][64]   color_net.2.weight, [64][3]
    nav_levels lv;
    sh_norm shn;
    float bound, r2b, density_scale;
};

// ---------------------------------------------------------------------------
// hash-grid encoding of one level (gridencoder.cu:125-170, float32), optionally with the Jacobian contraction
// ---------------------------------------------------------------------------

__device__ __forceinline__ uint32_t nv_row(const nav_params& P, int l, uint32_t x, uint32_t y, uint32_t z) {
    if (P.lv.s1[l]) return P.lv.base[l] + x + y * P.lv.s1[l] + z * P.lv.s2[l];                    // dense level: always < size
    const uint32_t h = x ^ (y * 2654435761u) ^ (z * 805459861u);                                   // fast_hash (gridencoder.cu:35-51)
    return P.lv.base[l] + (P.lv.mask[l] ? (h & P.lv.mask[l]) : (h % P.lv.size[l]));
}

struct nv_cell { uint32_t gx, gy, gz; float fx, fy, fz; };

__device__ __forceinline__ nv_cell nv_locate(float scale, float x0, float x1, float x2) {
    nv_cell c;
    const float px = x0 * scale + 0.5f, py = x1 * scale + 0.5f, pz = x2 * scale + 0.5f;
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    c.gx = (uint32_t)flx; c.gy = (uint32_t)fly; c.gz = (uint32_t)flz;
    c.fx = px - flx; c.fy = py - fly; c.fz = pz - flz;
    return c;
}

// the two features of level l at a normalised position in [0,1]^3: corners in the reference's order, w = ((wx) wy) wz
__device__ __forceinline__ void nv_level(const nav_params& P, int l, float x0, float x1, float x2, float& e0, float& e1) {
    const nv_cell c = nv_locate(P.lv.scale[l], x0, x1, x2);
    float2 v[8];
    #pragma unroll
    for (int k = 0; k < 8; k++) v[k] = P.table[nv_row(P, l, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + (k >> 2))];
    e0 = 0.0f; e1 = 0.0f;
    #pragma unroll
    for (int k = 0; k < 8; k++) {
        const float w = (((k & 1) ? c.fx : 1.0f - c.fx) * ((k & 2) ? c.fy : 1.0f - c.fy)) * ((k & 4) ? c.fz : 1.0f - c.fz);
        e0 = __builtin_fmaf(w, v[k].x, e0);
        e1 = __builtin_fmaf(w, v[k].y, e1);
    }
}

// g0 * d e0 / d x01 + g1 * d e1 / d x01 for level l (the dy_dx of gridencoder.cu:173-222 contracted with the gradient)
__device__ __forceinline__ void nv_level_grad(const nav_params& P, int l, float x0, float x1, float x2, float g0, float g1,
                                              float& ax, float& ay, float& az) {
    const float scale = P.lv.scale[l];
    const nv_cell c = nv_locate(scale, x0, x1, x2);
    float s[8];                                                          // g . value at each corner
    #pragma unroll
    for (int k = 0; k < 8; k++) {
        const float2 v = P.table[nv_row(P, l, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + (k >> 2))];
        s[k] = __builtin_fmaf(g0, v.x, g1 * v.y);
    }
    const float wx0 = 1.0f - c.fx, wy0 = 1.0f - c.fy, wz0 = 1.0f - c.fz;
    // d/dx: sum over (y,z) of wy wz (right - left), etc.
    const float dx = (wy0 * wz0) * (s[1] - s[0]) + (c.fy * wz0) * (s[3] - s[2]) + (wy0 * c.fz) * (s[5] - s[4]) + (c.fy * c.fz) * (s[7] - s[6]);
    const float dy = (wx0 * wz0) * (s[2] - s[0]) + (c.fx * wz0) * (s[3] - s[1]) + (wx0 * c.fz) * (s[6] - s[4]) + (c.fx * c.fz) * (s[7] - s[5]);
    const float dz = (wx0 * wy0) * (s[4] - s[0]) + (c.fx * wy0) * (s[5] - s[1]) + (wx0 * c.fy) * (s[6] - s[2]) + (c.fx * c.fy) * (s[7] - s[3]);
    ax = __builtin_fmaf(scale, dx, ax);
    ay = __builtin_fmaf(scale, dy, ay);
    az = __builtin_fmaf(scale, dz, az);
}

// ---------------------------------------------------------------------------
// density net: hash grid -> Linear(32,64) -> ReLU -> Linear(64,16)      (nerf/network.py:95-111)
// ---------------------------------------------------------------------------

// world position -> normalised position (grid.py:144); `inside` false = outside [0,1]^3 (or NaN): the encoder returns zeros (gridencoder.cu:99-123)
__device__ __forceinline__ bool nv_normalise(const nav_params& P, float wx, float wy, float wz, float& x0, float& x1, float& x2) {
    // torch on the GPU divides by a host scalar by multiplying with its binary32 reciprocal (ATen BinaryDivTrueKernel.cu): r2b = fl(1 / (2 bound))
    x0 = (wx + P.bound) * P.r2b; x1 = (wy + P.bound) * P.r2b; x2 = (wz + P.bound) * P.r2b;
    const bool inside = (x0 >= 0.0f && x0 <= 1.0f) && (x1 >= 0.0f && x1 <= 1.0f) && (x2 >= 0.0f && x2 <= 1.0f);
    if (!inside) { x0 = 0.0f; x1 = 0.0f; x2 = 0.0f; }                     // gather somewhere valid; the features are discarded
    return inside;
}

// y[j] = sum_k W[k][j] x[k] for j < NOUT: a rolled loop over the reduction index k, x[k] read from this lane's LDS column, the NOUT
// accumulators in registers with compile-time indices, row k of W (NOUT contiguous floats) through the scalar cache.  One row is at
// most 64 SGPRs, so nothing spills; the loop stays rolled so that no more than one row is ever in flight.
template <int NIN, int NOUT>
__device__ __forceinline__ void nv_matvec(nv_wptr W, const float* __restrict__ col, float (&y)[NOUT]) {
    #pragma unroll
    for (int j = 0; j < NOUT; j++) y[j] = 0.0f;
    #pragma unroll 1
    for (int k = 0; k < NIN; k++) {
        const float a = col[k * NV_BLOCK];
        const nv_wptr w = W + k * NOUT;
        #pragma unroll
        for (int j = 0; j < NOUT; j++) y[j] = __builtin_fmaf(w[j], a, y[j]);
    }
}

template <int N>
__device__ __forceinline__ void nv_park(const float (&v)[N], float* __restrict__ col, int at = 0) {
    #pragma unroll
    for (int k = 0; k < N; k++) col[(at + k) * NV_BLOCK] = v[k];
}

template <int N>
__device__ __forceinline__ uint64_t nv_park_relu(const float (&v)[N], float* __restrict__ col) {       // relu(v) into the column; returns the mask
    uint64_t m = 0;
    #pragma unroll
    for (int k = 0; k < N; k++) { col[k * NV_BLOCK] = fmaxf(v[k], 0.0f); m |= (uint64_t)(v[k] > 0.0f) << k; }
    return m;
}

// hidden pre-activations h[64] = W1 . enc(x): the first layer is accumulated level by level, so the 32 features never exist together
// (U = how many levels the compiler may overlap: 1 inside the run kernels, whose eight waves per CU hide the gather latency by themselves;
//  4 in the point kernels, where a planner-sized batch is 40 workgroups on 256 CUs and the 16 dependent round trips are the run time)
template <int U>
__device__ __forceinline__ void nv_hidden(const nav_params& P, bool inside, float x0, float x1, float x2, float (&h)[NV_H]) {
    #pragma unroll
    for (int j = 0; j < NV_H; j++) h[j] = 0.0f;
    #pragma unroll U
    for (int l = 0; l < NV_L; l++) {
        float e0, e1;
        nv_level(P, l, x0, x1, x2, e0, e1);
        if (!inside) { e0 = 0.0f; e1 = 0.0f; }
        const nv_wptr wa = P.w1t + (2 * l) * NV_H;                       // wave-uniform addresses: scalar loads
        const nv_wptr wb = wa + NV_H;
        #pragma unroll
        for (int j = 0; j < NV_H; j++) h[j] = __builtin_fmaf(wa[j], e0, h[j]);
        #pragma unroll
        for (int j = 0; j < NV_H; j++) h[j] = __builtin_fmaf(wb[j], e1, h[j]);
    }
}

// density net forward for one point: out[16] = W2 . relu(W1 . enc(x)); returns the ReLU mask of the hidden layer
template <int U = 1>
__device__ __forceinline__ uint64_t nv_density_forward(const nav_params& P, bool inside, float x0, float x1, float x2, float* __restrict__ col,
                                                       float (&out)[16]) {
    float h[NV_H];
    nv_hidden<U>(P, inside, x0, x1, x2, h);
    const uint64_t relu = nv_park_relu(h, col);
    nv_matvec<NV_H, 16>(P.w2t, col, out);
    return relu;
}

// backward of the density net for one point: gout[16] = d L / d out, `relu` = which hidden units were active  ->  d L / d (world position)
template <int U = 1>
__device__ __forceinline__ void nv_density_backward(const nav_params& P, bool inside, float x0, float x1, float x2, uint64_t relu,
                                                    const float (&gout)[16], float* __restrict__ col, float& gx, float& gy, float& gz) {
    float gh[NV_H];                                                      // d L / d h (ReLU mask applied)
    nv_park(gout, col);
    nv_matvec<16, NV_H>(P.w2, col, gh);
    #pragma unroll
    for (int j = 0; j < NV_H; j++) gh[j] = ((relu >> j) & 1ull) ? gh[j] : 0.0f;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    #pragma unroll U
    for (int l = 0; l < NV_L; l++) {
        const nv_wptr wa = P.w1t + (2 * l) * NV_H;
        const nv_wptr wb = wa + NV_H;
        float g0 = 0.0f, g1 = 0.0f;                                      // d L / d enc[2l], enc[2l+1]
        #pragma unroll
        for (int j = 0; j < NV_H; j++) g0 = __builtin_fmaf(wa[j], gh[j], g0);
        #pragma unroll
        for (int j = 0; j < NV_H; j++) g1 = __builtin_fmaf(wb[j], gh[j], g1);
        nv_level_grad(P, l, x0, x1, x2, g0, g1, ax, ay, az);
    }
    const float k = inside ? P.r2b : 0.0f;                                                 // d x01 / d world; zero outside the box
    gx = ax * k; gy = ay * k; gz = az * k;
}

__device__ __forceinline__ float nv_exp_clamped(float x) { return expf(fminf(fmaxf(x, -15.0f), 15.0f)); }   // trunc_exp backward (activation.py:16-18)

// ---------------------------------------------------------------------------
// colour net: cat(SH16(dir), geo15) -> Linear(31,64) -> ReLU -> Linear(64,64) -> ReLU -> Linear(64,3) -> sigmoid   (network.py:113-123)
// ---------------------------------------------------------------------------

// SH(4) of a direction and, on request, the contraction of its Jacobian with a gradient vector (shencoder.cu:50-355 through the
// recurrences of ngp_sh.h)
__device__ __forceinline__ void nv_sh(const nav_params& P, float x, float y, float z, float (&sh)[16]) { sh_eval<4>(x, y, z, P.shn, sh); }

__device__ __forceinline__ void nv_sh_backward(const nav_params& P, float x, float y, float z, const float (&g)[16], float& dx, float& dy, float& dz) {
    sh_tables<4> t;
    t.build(x, y, z);
    dx = 0.0f; dy = 0.0f; dz = 0.0f;
    #pragma unroll
    for (uint32_t l = 0; l < 4; l++) {
        #pragma unroll
        for (uint32_t m = 0; m <= l; m++) {
            const float nq = P.shn.n[l][m] * t.Q[l][m], nqz = P.shn.n[l][m] * t.Q[l][m + 1];
            const uint32_t ip = l * l + l + m, in = l * l + l - m;
            dz = __builtin_fmaf(g[ip], nqz * t.A[m], dz);
            if (m > 0) {
                const float fm = (float)m;
                dz = __builtin_fmaf(g[in], nqz * t.B[m], dz);
                dx = __builtin_fmaf(g[ip], nq * (fm * t.A[m - 1]), dx);
                dy = __builtin_fmaf(g[ip], nq * (-fm * t.B[m - 1]), dy);
                dx = __builtin_fmaf(g[in], nq * (fm * t.B[m - 1]), dx);
                dy = __builtin_fmaf(g[in], nq * (fm * t.A[m - 1]), dy);
            }
        }
    }
}

// colour forward for one sample; returns the ReLU masks of the two hidden layers and the pre-sigmoid... no: rgb after the sigmoid
__device__ __forceinline__ void nv_color_forward(const nav_params& P, float dxr, float dyr, float dzr, const float (&geo)[NV_GEO],
                                                 float* __restrict__ col, float (&rgb)[3], uint64_t& mask_a, uint64_t& mask_b) {
    {
        float sh[16];
        nv_sh(P, dxr, dyr, dzr, sh);
        nv_park(sh, col);
        nv_park(geo, col, 16);
    }
    float hid[NV_H];
    nv_matvec<NV_CIN, NV_H>(P.v0t, col, hid);
    mask_a = nv_park_relu(hid, col);
    nv_matvec<NV_H, NV_H>(P.v1t, col, hid);
    mask_b = nv_park_relu(hid, col);
    float o[3];
    nv_matvec<NV_H, 3>(P.v2t, col, o);
    #pragma unroll
    for (int c = 0; c < 3; c++) rgb[c] = 1.0f / (1.0f + expf(-o[c]));
}

// ---------------------------------------------------------------------------
// density queries on explicit points (planner: nav/quad_plot.py:224-250)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(NV_BLOCK) void k_nav_density_fwd(nav_params P, const float* __restrict__ xyz, uint32_t M, float* __restrict__ sigma,
                                                              float* __restrict__ geo) {
    extern __shared__ float nv_smem[];
    float* col = nv_smem + 16 + threadIdx.x;
    const uint32_t i = blockIdx.x * NV_BLOCK + threadIdx.x;
    const uint32_t ic = i < M ? i : M - 1;
    float x0, x1, x2;
    const bool inside = nv_normalise(P, xyz[3ull * ic], xyz[3ull * ic + 1], xyz[3ull * ic + 2], x0, x1, x2);
    float out[16];
    nv_density_forward<4>(P, inside, x0, x1, x2, col, out);
    if (i >= M) return;
    if (geo) {
        #pragma unroll
        for (int m = 0; m < NV_GEO; m++) geo[(uint64_t)i * NV_GEO + m] = out[1 + m];
    }
    sigma[i] = expf(out[0]);                                           // trunc_exp forward (activation.py:9-10)
}

__global__ __launch_bounds__(NV_BLOCK) void k_nav_density_bwd(nav_params P, const float* __restrict__ xyz, uint32_t M, const float* __restrict__ gsigma,
                                                              const float* __restrict__ ggeo, float* __restrict__ gxyz) {
    extern __shared__ float nv_smem[];
    float* col = nv_smem + 16 + threadIdx.x;
    const uint32_t i = blockIdx.x * NV_BLOCK + threadIdx.x;
    const uint32_t ic = i < M ? i : M - 1;
    float x0, x1, x2;
    const bool inside = nv_normalise(P, xyz[3ull * ic], xyz[3ull * ic + 1], xyz[3ull * ic + 2], x0, x1, x2);
    float out[16], gout[16];
    const uint64_t relu = nv_density_forward<4>(P, inside, x0, x1, x2, col, out);
    gout[0] = gsigma[ic] * nv_exp_clamped(out[0]);
    #pragma unroll
    for (int m = 0; m < NV_GEO; m++) gout[1 + m] = ggeo ? ggeo[(uint64_t)ic * NV_GEO + m] : 0.0f;
    float gx, gy, gz;
    nv_density_backward<4>(P, inside, x0, x1, x2, relu, gout, col, gx, gy, gz);
    if (i >= M) return;
    gxyz[3ull * i] = gx; gxyz[3ull * i + 1] = gy; gxyz[3ull * i + 2] = gz;
}

// ---------------------------------------------------------------------------
// The planner's query (nav/quad_plot.py:224-250 through simulate.py:340-343): sigma AND d sigma / d x for a FEW THOUSAND points, in one launch.
//
// k_nav_density_fwd / _bwd walk a point's 16 levels in one lane: for 10,000 body points that is 40 workgroups on 256 CUs, each waiting for 16 dependent
// gather round trips and a 6,000-FMA matrix-vector chain on one wave -- slower than the level-parallel op chain (0.21 against 0.15 ms).  Here a
// point's levels are split over the FOUR WAVES of its workgroup (64 points per workgroup, lane = point, wave w = levels 4w..4w+3): all 32 gathers of a
// lane are issued before the first use (one round trip), each wave accumulates its share of the first layer, the shares meet in LDS, and since only
// sigma = exp(out[0]) is asked for, the second layer is ONE row: out0 = W2[0] . relu(h) and d out0 / d h = W2[0] masked by the ReLU -- no matrix-vector
// product on the way back.  The Jacobian reuses the corner values the forward gathered (nv_level_grad would fetch them again).
// `rot`: simulate.py:340's axis change x @ rot folded in (a 3x3 row-major matrix or null): p = x rot on the way in, d sigma / d x = rot (d sigma / d p) on the
// way out.  jac already contains trunc_exp's backward factor exp(clamp(out0, -15, 15)) (activation.py:16-18): grad_x = grad_sigma * jac.
// ---------------------------------------------------------------------------
struct nav_rot { float m[9]; uint32_t on; };
static constexpr uint32_t NV_VJ_POINTS = 64;
static constexpr size_t NV_VJ_LDS = sizeof(float) * (4 * NV_H * 64 + NV_H * 64 + 4 * 64 + 4 * 3 * 64);      // h shares | d out0 / d h | out0 shares | gradient shares

__global__ __launch_bounds__(256) void k_nav_density_vj(nav_params P, nav_rot R, const float* __restrict__ xyz, uint32_t M, float* __restrict__ sigma,
                                                        float* __restrict__ jac) {
    extern __shared__ float nv_smem[];
    float* lds_h = nv_smem;                                              // [4 waves][64 j][64 lanes]
    float* lds_gh = lds_h + 4 * NV_H * 64;                               // [64 j][64 lanes]
    float* lds_o = lds_gh + NV_H * 64;                                   // [4][64]
    float* lds_a = lds_o + 4 * 64;                                       // [4][3][64]
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t i = blockIdx.x * NV_VJ_POINTS + lane;
    const uint32_t ic = i < M ? i : M - 1;
    const float wx = xyz[3ull * ic], wy = xyz[3ull * ic + 1], wz = xyz[3ull * ic + 2];
    float px = wx, py = wy, pz = wz;
    if (R.on) {                                                          // p = x @ rot (row vector times matrix)
        px = (wx * R.m[0] + wy * R.m[3]) + wz * R.m[6];
        py = (wx * R.m[1] + wy * R.m[4]) + wz * R.m[7];
        pz = (wx * R.m[2] + wy * R.m[5]) + wz * R.m[8];
    }
    float x0, x1, x2;
    const bool inside = nv_normalise(P, px, py, pz, x0, x1, x2);

    // ---- this wave's four levels: cells, all 32 corner rows in flight at once ----
    nv_cell c[4];
    float2 v[4][8];
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const int l = (int)(4u * wave) + k;
        c[k] = nv_locate(P.lv.scale[l], x0, x1, x2);
        #pragma unroll
        for (int q = 0; q < 8; q++) v[k][q] = P.table[nv_row(P, l, c[k].gx + (q & 1), c[k].gy + ((q >> 1) & 1), c[k].gz + (q >> 2))];
    }
    // ---- its share of the hidden pre-activations: h += W1t[2l] e0 + W1t[2l+1] e1 (nv_hidden's order inside a level) ----
    {
        float h[NV_H];
        #pragma unroll
        for (int j = 0; j < NV_H; j++) h[j] = 0.0f;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            float e0 = 0.0f, e1 = 0.0f;
            #pragma unroll
            for (int q = 0; q < 8; q++) {
                const float w = (((q & 1) ? c[k].fx : 1.0f - c[k].fx) * ((q & 2) ? c[k].fy : 1.0f - c[k].fy)) * ((q & 4) ? c[k].fz : 1.0f - c[k].fz);
                e0 = __builtin_fmaf(w, v[k][q].x, e0);
                e1 = __builtin_fmaf(w, v[k][q].y, e1);
            }
            if (!inside) { e0 = 0.0f; e1 = 0.0f; }
            const nv_wptr wa = nv_hide(P.w1t) + (2 * (4 * wave + k)) * NV_H;
            const nv_wptr wb = wa + NV_H;
            #pragma unroll
            for (int j = 0; j < NV_H; j++) h[j] = __builtin_fmaf(wa[j], e0, h[j]);
            #pragma unroll
            for (int j = 0; j < NV_H; j++) h[j] = __builtin_fmaf(wb[j], e1, h[j]);
        }
        #pragma unroll
        for (int j = 0; j < NV_H; j++) lds_h[(wave * NV_H + j) * 64 + lane] = h[j];
    }
    __syncthreads();
    // ---- wave w owns hidden units 16w .. 16w+15: sum the four shares (fixed order), ReLU, its part of out0 = W2[0] . relu(h) and of d out0 / d h ----
    {
        const nv_wptr w2 = nv_hide(P.w2);                                // row 0 of sigma_net.1.weight [16][64]
        float o = 0.0f;
        #pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int j = (int)(16u * wave) + jj;
            const float hj = ((lds_h[(0 * NV_H + j) * 64 + lane] + lds_h[(1 * NV_H + j) * 64 + lane]) + lds_h[(2 * NV_H + j) * 64 + lane]) + lds_h[(3 * NV_H + j) * 64 + lane];
            const float wj = w2[j];
            o = __builtin_fmaf(wj, fmaxf(hj, 0.0f), o);
            lds_gh[j * 64 + lane] = hj > 0.0f ? wj : 0.0f;
        }
        lds_o[wave * 64 + lane] = o;
    }
    __syncthreads();
    const float out0 = ((lds_o[lane] + lds_o[64 + lane]) + lds_o[128 + lane]) + lds_o[192 + lane];
    // ---- back through the first layer for this wave's levels, contracted with the trilinear Jacobian of the corners it still holds ----
    {
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const nv_wptr wa = nv_hide(P.w1t) + (2 * (4 * wave + k)) * NV_H;
            const nv_wptr wb = wa + NV_H;
            float g0 = 0.0f, g1 = 0.0f;
            #pragma unroll
            for (int j = 0; j < NV_H; j++) { const float gj = lds_gh[j * 64 + lane]; g0 = __builtin_fmaf(wa[j], gj, g0); g1 = __builtin_fmaf(wb[j], gj, g1); }
            float sv[8];
            #pragma unroll
            for (int q = 0; q < 8; q++) sv[q] = __builtin_fmaf(g0, v[k][q].x, g1 * v[k][q].y);
            const float fx = c[k].fx, fy = c[k].fy, fz = c[k].fz, wx0 = 1.0f - fx, wy0 = 1.0f - fy, wz0 = 1.0f - fz;
            const float dx = (wy0 * wz0) * (sv[1] - sv[0]) + (fy * wz0) * (sv[3] - sv[2]) + (wy0 * fz) * (sv[5] - sv[4]) + (fy * fz) * (sv[7] - sv[6]);
            const float dy = (wx0 * wz0) * (sv[2] - sv[0]) + (fx * wz0) * (sv[3] - sv[1]) + (wx0 * fz) * (sv[6] - sv[4]) + (fx * fz) * (sv[7] - sv[5]);
            const float dz = (wx0 * wy0) * (sv[4] - sv[0]) + (fx * wy0) * (sv[5] - sv[1]) + (wx0 * fy) * (sv[6] - sv[2]) + (fx * fy) * (sv[7] - sv[3]);
            const float scale = P.lv.scale[4 * wave + k];
            ax = __builtin_fmaf(scale, dx, ax); ay = __builtin_fmaf(scale, dy, ay); az = __builtin_fmaf(scale, dz, az);
        }
        lds_a[(wave * 3 + 0) * 64 + lane] = ax; lds_a[(wave * 3 + 1) * 64 + lane] = ay; lds_a[(wave * 3 + 2) * 64 + lane] = az;
    }
    __syncthreads();
    if (wave != 0 || i >= M) return;
    float a[3];
    #pragma unroll
    for (int d = 0; d < 3; d++) a[d] = ((lds_a[(0 * 3 + d) * 64 + lane] + lds_a[(1 * 3 + d) * 64 + lane]) + lds_a[(2 * 3 + d) * 64 + lane]) + lds_a[(3 * 3 + d) * 64 + lane];
    const float k = (inside ? P.r2b : 0.0f) * nv_exp_clamped(out0);      // d x01 / d p, times trunc_exp's backward factor
    const float gpx = a[0] * k, gpy = a[1] * k, gpz = a[2] * k;
    float gx = gpx, gy = gpy, gz = gpz;
    if (R.on) {                                                          // d sigma / d x = rot . (d sigma / d p)
        gx = (R.m[0] * gpx + R.m[1] * gpy) + R.m[2] * gpz;
        gy = (R.m[3] * gpx + R.m[4] * gpy) + R.m[5] * gpz;
        gz = (R.m[6] * gpx + R.m[7] * gpy) + R.m[8] * gpz;
    }
    sigma[i] = expf(out0);                                               // trunc_exp forward (activation.py:9-10)
    jac[3ull * i] = gx; jac[3ull * i + 1] = gy; jac[3ull * i + 2] = gz;
}

// ---------------------------------------------------------------------------
// NeRFRenderer.run (nerf/renderer.py:125-254) with upsample_steps = 0, perturb = False
// ---------------------------------------------------------------------------

struct nav_run {
    const float* rays_o; const float* rays_d; const float* nears; const float* fars;
    uint32_t N, T;
    float aabb[6];
    float bg[3];
};

// torch.linspace(0, 1, T)[i] in float32 as ATen's GPU kernel produces it (RangeFactories.cu: start + step * i for the first half,
// end - step * (T - 1 - i) for the second, compiled with contraction: ONE rounding).  Matching its bits matters more than it looks: a
// sample one ulp away can sit in the neighbouring cell of a fine level, where the derivative of the trilinear interpolation differs.
__device__ __forceinline__ float nv_lin(uint32_t i, uint32_t T) {
    if (T < 2) return 0.0f;
    const float step = 1.0f / (float)(T - 1);
    return i < T / 2 ? step * (float)i : __builtin_fmaf(-step, (float)(T - 1 - i), 1.0f);
}

struct nv_sample { float z, znext_minus_z, px, py, pz, lin, bx, by, bz; };       // b*: how much of the gradient the clip of that coordinate passes

// d min(max(x, lo), hi) / d x as torch.max / torch.min differentiate it (:159): 1 inside, 0 outside, and HALF on a tie -- which is the
// common case for the first sample of a ray, whose position is the point where the ray enters the box
__device__ __forceinline__ float nv_clip_pass(float x, float lo, float hi) {
    const float a = x > lo ? 1.0f : (x == lo ? 0.5f : 0.0f);
    const float m = fmaxf(x, lo);
    const float b = m < hi ? 1.0f : (m == hi ? 0.5f : 0.0f);
    return a * b;
}

__device__ __forceinline__ nv_sample nv_sample_at(const nav_run& R, uint32_t i, float near, float far, float ox, float oy, float oz,
                                                  float dx, float dy, float dz) {
    nv_sample s;
    const float span = far - near;
    s.lin = nv_lin(i, R.T);
    s.z = near + span * s.lin;                                           // :147-148
    const float zn = (i + 1 < R.T) ? near + span * nv_lin(i + 1, R.T) : 0.0f;
    s.znext_minus_z = (i + 1 < R.T) ? zn - s.z : span / (float)R.T;      // deltas, last = sample_dist (:206-207, :151)
    const float x = ox + dx * s.z, y = oy + dy * s.z, z = oz + dz * s.z; // :158
    s.px = fminf(fmaxf(x, R.aabb[0]), R.aabb[3]);                        // :159
    s.py = fminf(fmaxf(y, R.aabb[1]), R.aabb[4]);
    s.pz = fminf(fmaxf(z, R.aabb[2]), R.aabb[5]);
    s.bx = nv_clip_pass(x, R.aabb[0], R.aabb[3]);
    s.by = nv_clip_pass(y, R.aabb[1], R.aabb[4]);
    s.bz = nv_clip_pass(z, R.aabb[2], R.aabb[5]);
    return s;
}

// exclusive prefix product of one value per thread over the block (256 threads = 4 waves); `total` = the product of all of them
__device__ __forceinline__ float nv_block_excl_product(float v, float* __restrict__ lds4, float& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float incl = v;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const float up = __shfl_up(incl, off, 64); if (lane >= off) incl = up * incl; }
    __syncthreads();
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    float before = __shfl_up(incl, 1, 64);
    if (lane == 0) before = 1.0f;
    float base = 1.0f, tot = 1.0f;
    #pragma unroll
    for (int w = 0; w < (int)(NV_BLOCK / 64); w++) { const float t = lds4[w]; if (w < wave) base *= t; tot *= t; }
    total = tot;
    return base * before;
}

__device__ __forceinline__ float nv_block_sum(float v, float* __restrict__ lds4) {
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// alpha and the transmittance factor of one sample (:208-209)
__device__ __forceinline__ void nv_alpha(float delta, float scale, float sigma, float& alpha, float& keep) {
    alpha = 1.0f - expf(-delta * scale * sigma);
    keep = 1.0f - alpha + 1e-15f;
}

// What the forward leaves for the backward, one record per sample, stored word-major ([word][N*T]: the lanes of a wave are consecutive
// samples, so every word is a coalesced 256-byte store): the density net's 16 outputs, its ReLU mask, alpha and the transmittance in
// front of the sample, the colour and the colour net's two ReLU masks.  108 bytes per sample (57 MB for 1,024 rays x 512 steps)
// instead of recomputing two encoder passes and both networks in the backward.
static constexpr uint32_t NV_REC_WORDS = 27;
enum { NV_R_OUT = 0, NV_R_RELU = 16, NV_R_ALPHA = 18, NV_R_TR = 19, NV_R_RGB = 20, NV_R_MA = 23, NV_R_MB = 25 };

__global__ __launch_bounds__(NV_BLOCK) void k_nav_run_fwd(nav_params P, nav_run R, float* __restrict__ image, float* __restrict__ depth,
                                                          float* __restrict__ weights_sum, uint32_t* __restrict__ save) {
    extern __shared__ float nv_smem[];
    float* lds4 = nv_smem;                                              // 16 floats of scan scratch
    float* col = nv_smem + 16 + threadIdx.x;                            // this lane's column of the [64][256] scratch
    const uint32_t n = blockIdx.x;
    const size_t NT = (size_t)R.N * R.T;
    const float ox = R.rays_o[3ull * n], oy = R.rays_o[3ull * n + 1], oz = R.rays_o[3ull * n + 2];
    const float dx = R.rays_d[3ull * n], dy = R.rays_d[3ull * n + 1], dz = R.rays_d[3ull * n + 2];
    const float near = R.nears[n], far = R.fars[n], span = far - near;
    float carry = 1.0f, acc_w = 0.0f, acc_d = 0.0f, acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
    for (uint32_t c0 = 0; c0 < R.T; c0 += NV_BLOCK) {
        const uint32_t i = c0 + threadIdx.x;
        const bool live = i < R.T;
        const nv_sample s = nv_sample_at(R, live ? i : R.T - 1, near, far, ox, oy, oz, dx, dy, dz);
        float x0, x1, x2;
        const bool inside = nv_normalise(P, s.px, s.py, s.pz, x0, x1, x2);
        float out[16];
        const uint64_t relu = nv_density_forward<NV_RUN_U>(P, inside, x0, x1, x2, col, out);
        float alpha, keep;
        nv_alpha(s.znext_minus_z, P.density_scale, expf(out[0]), alpha, keep);
        if (!live) { alpha = 0.0f; keep = 1.0f; }
        float total;
        const float Tr = carry * nv_block_excl_product(keep, lds4, total);
        const float w = alpha * Tr;
        uint32_t* q = save + (size_t)n * R.T + i;
        if (save && live) {                                              // before the colour net, so that out[] dies here
            #pragma unroll
            for (int m = 0; m < 16; m++) q[(NV_R_OUT + m) * NT] = __float_as_uint(out[m]);
            q[NV_R_RELU * NT] = (uint32_t)relu; q[(NV_R_RELU + 1) * NT] = (uint32_t)(relu >> 32);
            q[NV_R_ALPHA * NT] = __float_as_uint(alpha); q[NV_R_TR * NT] = __float_as_uint(Tr);
        }
        float rgb[3] = {0.0f, 0.0f, 0.0f};
        uint64_t ma = 0, mb = 0;
        if (live && w > 1e-4f) {                                         // :217 colour only where the weight matters
            float geo[NV_GEO];
            #pragma unroll
            for (int m = 0; m < NV_GEO; m++) geo[m] = out[1 + m];
            nv_color_forward(P, dx, dy, dz, geo, col, rgb, ma, mb);
        }
        if (save && live) {
            q[NV_R_RGB * NT] = __float_as_uint(rgb[0]); q[(NV_R_RGB + 1) * NT] = __float_as_uint(rgb[1]); q[(NV_R_RGB + 2) * NT] = __float_as_uint(rgb[2]);
            q[NV_R_MA * NT] = (uint32_t)ma; q[(NV_R_MA + 1) * NT] = (uint32_t)(ma >> 32);
            q[NV_R_MB * NT] = (uint32_t)mb; q[(NV_R_MB + 1) * NT] = (uint32_t)(mb >> 32);
        }
        const float oz01 = fminf(fmaxf((s.z - near) / span, 0.0f), 1.0f); // :225
        acc_w += w; acc_d += w * oz01; acc_r += w * rgb[0]; acc_g += w * rgb[1]; acc_b += w * rgb[2];
        carry *= total;
    }
    const float ws = nv_block_sum(acc_w, lds4), dsum = nv_block_sum(acc_d, lds4);
    const float r = nv_block_sum(acc_r, lds4), g = nv_block_sum(acc_g, lds4), b = nv_block_sum(acc_b, lds4);
    if (threadIdx.x == 0) {
        weights_sum[n] = ws;
        depth[n] = dsum;
        image[3ull * n] = r + (1.0f - ws) * R.bg[0];                    // :241
        image[3ull * n + 1] = g + (1.0f - ws) * R.bg[1];
        image[3ull * n + 2] = b + (1.0f - ws) * R.bg[2];
    }
}

// colour net backward from the saved forward: grgb[3] = d L / d rgb  ->  ggeo[15], d L / d dir (added)
__device__ __forceinline__ void nv_color_backward_saved(const nav_params& P, float dxr, float dyr, float dzr, const float (&rgb)[3], uint64_t mask_a,
                                                        uint64_t mask_b, const float (&grgb)[3], float* __restrict__ col, float (&ggeo)[NV_GEO],
                                                        float& gdx, float& gdy, float& gdz) {
    float go[3];
    #pragma unroll
    for (int c = 0; c < 3; c++) go[c] = grgb[c] * rgb[c] * (1.0f - rgb[c]);
    float g[NV_H];
    nv_park(go, col);
    nv_matvec<3, NV_H>(P.v2, col, g);                                    // d L / d hb
    #pragma unroll
    for (int j = 0; j < NV_H; j++) col[j * NV_BLOCK] = ((mask_b >> j) & 1ull) ? g[j] : 0.0f;
    nv_matvec<NV_H, NV_H>(P.v1, col, g);                                 // d L / d ha
    #pragma unroll
    for (int k = 0; k < NV_H; k++) col[k * NV_BLOCK] = ((mask_a >> k) & 1ull) ? g[k] : 0.0f;
    float gin[NV_CIN];
    nv_matvec<NV_H, NV_CIN>(P.v0, col, gin);                             // d L / d cat(SH, geo)
    float gsh[16];
    #pragma unroll
    for (int i = 0; i < 16; i++) gsh[i] = gin[i];
    #pragma unroll
    for (int i = 0; i < NV_GEO; i++) ggeo[i] = gin[16 + i];
    float sx, sy, sz;
    nv_sh_backward(P, dxr, dyr, dzr, gsh, sx, sy, sz);
    gdx += sx; gdy += sy; gdz += sz;
}

__global__ __launch_bounds__(NV_BLOCK) void k_nav_run_bwd(nav_params P, nav_run R, const float* __restrict__ gimage, const float* __restrict__ gdepth,
                                                          const float* __restrict__ gws, const uint32_t* __restrict__ save,
                                                          float* __restrict__ grays_o, float* __restrict__ grays_d) {
    extern __shared__ float nv_smem[];
    float* lds4 = nv_smem;
    float* col = nv_smem + 16 + threadIdx.x;
    const uint32_t n = blockIdx.x;
    const size_t NT = (size_t)R.N * R.T;
    const float ox = R.rays_o[3ull * n], oy = R.rays_o[3ull * n + 1], oz = R.rays_o[3ull * n + 2];
    const float dx = R.rays_d[3ull * n], dy = R.rays_d[3ull * n + 1], dz = R.rays_d[3ull * n + 2];
    const float near = R.nears[n], far = R.fars[n], span = far - near;
    const float gi0 = gimage[3ull * n], gi1 = gimage[3ull * n + 1], gi2 = gimage[3ull * n + 2];
    const float gd = gdepth ? gdepth[n] : 0.0f, gw = gws ? gws[n] : 0.0f;
    const uint32_t nchunks = (R.T + NV_BLOCK - 1) / NV_BLOCK;

    // back to front: d L / d sigma_i needs the sum over the samples BEHIND i, so the chunks run in reverse and carry that sum
    float suffix = 0.0f;
    float go0 = 0.0f, go1 = 0.0f, go2 = 0.0f, gdx = 0.0f, gdy = 0.0f, gdz = 0.0f;
    for (uint32_t cc = nchunks; cc-- > 0;) {
        const uint32_t i = cc * NV_BLOCK + threadIdx.x;
        const bool live = i < R.T;
        const nv_sample s = nv_sample_at(R, live ? i : R.T - 1, near, far, ox, oy, oz, dx, dy, dz);
        const uint32_t* q = save + (size_t)n * R.T + (live ? i : R.T - 1);
        const float alpha = live ? __uint_as_float(q[NV_R_ALPHA * NT]) : 0.0f, Tr = live ? __uint_as_float(q[NV_R_TR * NT]) : 0.0f;
        const float rgb[3] = {__uint_as_float(q[NV_R_RGB * NT]), __uint_as_float(q[(NV_R_RGB + 1) * NT]), __uint_as_float(q[(NV_R_RGB + 2) * NT])};
        const float w = alpha * Tr;
        const bool masked = live && w > 1e-4f;
        const float oz01 = fminf(fmaxf((s.z - near) / span, 0.0f), 1.0f);
        // A_i = d L / d w_i : image (colour minus background), depth, weights_sum
        const float A = live ? (gi0 * (rgb[0] - R.bg[0]) + gi1 * (rgb[1] - R.bg[1]) + gi2 * (rgb[2] - R.bg[2]) + gd * oz01 + gw) : 0.0f;
        const float Aw = A * w;
        // EXCLUSIVE by construction (scan of the sequence shifted by one lane): an inclusive scan minus the own term would cancel --
        // an opaque sample's own A w is many orders of magnitude above the sum of everything behind it, and that small sum, divided
        // by the sample's equally small (1 - alpha), is a first-order term of d L / d alpha
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float rs = __shfl_down(Aw, 1, 64);
        if (lane == 63) rs = 0.0f;
        #pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const float dn = __shfl_down(rs, off, 64); if (lane + off < 64) rs += dn; }
        __syncthreads();
        if (lane == 0) lds4[wave] = rs + Aw;                             // the wave's total
        __syncthreads();
        float behind = suffix, chunk_total = 0.0f;
        #pragma unroll
        for (int wv = (int)(NV_BLOCK / 64) - 1; wv >= 0; wv--) { const float t = lds4[wv]; if (wv > wave) behind += t; chunk_total += t; }
        const float S = behind + rs;                                     // sum over k > i of A_k w_k
        suffix += chunk_total;
        // w_i = alpha_i T_i, T_k (k > i) carries the factor (1 - alpha_i + eps):
        //   d L / d alpha_i = A_i T_i - S_i / (1 - alpha_i + eps);   d alpha_i / d sigma_i = delta_i scale (1 - alpha_i)
        const float keep = 1.0f - alpha + 1e-15f;
        const float gsigma = live ? s.znext_minus_z * P.density_scale * (1.0f - alpha) * (A * Tr - S / keep) : 0.0f;

        float gout[16];
        #pragma unroll
        for (int m = 0; m < 16; m++) gout[m] = 0.0f;
        if (masked) {
            float ggeo[NV_GEO];
            const uint64_t ma = (uint64_t)q[NV_R_MA * NT] | ((uint64_t)q[(NV_R_MA + 1) * NT] << 32);
            const uint64_t mb = (uint64_t)q[NV_R_MB * NT] | ((uint64_t)q[(NV_R_MB + 1) * NT] << 32);
            const float grgb[3] = {gi0 * w, gi1 * w, gi2 * w};
            nv_color_backward_saved(P, dx, dy, dz, rgb, ma, mb, grgb, col, ggeo, gdx, gdy, gdz);
            #pragma unroll
            for (int m = 0; m < NV_GEO; m++) gout[1 + m] = ggeo[m];
        }
        gout[0] = gsigma * nv_exp_clamped(__uint_as_float(q[NV_R_OUT * NT]));
        const uint64_t relu = (uint64_t)q[NV_R_RELU * NT] | ((uint64_t)q[(NV_R_RELU + 1) * NT] << 32);
        float x0, x1, x2;
        const bool inside = nv_normalise(P, s.px, s.py, s.pz, x0, x1, x2);
        float gx, gy, gz;
        nv_density_backward<NV_RUN_U>(P, inside, x0, x1, x2, relu, gout, col, gx, gy, gz);
        if (!live) { gx = 0.0f; gy = 0.0f; gz = 0.0f; }
        gx *= s.bx; gy *= s.by; gz *= s.bz;                              // clipped coordinates pass no (or half the) gradient (:159)
        go0 += gx; go1 += gy; go2 += gz;
        gdx = __builtin_fmaf(s.z, gx, gdx); gdy = __builtin_fmaf(s.z, gy, gdy); gdz = __builtin_fmaf(s.z, gz, gdz);
    }
    const float a0 = nv_block_sum(go0, lds4), a1 = nv_block_sum(go1, lds4), a2 = nv_block_sum(go2, lds4);
    const float b0 = nv_block_sum(gdx, lds4), b1 = nv_block_sum(gdy, lds4), b2 = nv_block_sum(gdz, lds4);
    if (threadIdx.x == 0) {
        grays_o[3ull * n] = a0; grays_o[3ull * n + 1] = a1; grays_o[3ull * n + 2] = a2;
        grays_d[3ull * n] = b0; grays_d[3ull * n + 1] = b1; grays_d[3ull * n + 2] = b2;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_nav_transpose(const float* __restrict__ src, uint32_t rows, uint32_t cols, float* __restrict__ dst) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * cols) return;
    const uint32_t r = t / cols, c = t - r * cols;
    dst[c * rows + r] = src[t];
}

static constexpr size_t NV_PREP_W1T = 0, NV_PREP_W2T = 32 * 64, NV_PREP_V0T = NV_PREP_W2T + 64 * 16, NV_PREP_V1T = NV_PREP_V0T + 31 * 64,
                        NV_PREP_V2T = NV_PREP_V1T + 64 * 64, NV_PREP_FLOATS = NV_PREP_V2T + 64 * 3;

extern "C" size_t ngp_nav_field_workspace(void) { return sizeof(float) * NV_PREP_FLOATS; }

// transposed copies of the five weight matrices into `workspace`; call again whenever the weights change
extern "C" int ngp_nav_field_prepare(const ngp_nav_field_t* f, void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(f && f->sigma_w0 && f->color_w1 && workspace && workspace_bytes >= ngp_nav_field_workspace(), "nav_field_prepare: bad argument");
    NGP_REQUIRE(f->sigma_w1 && f->color_w0 && f->color_w2, "nav_field_prepare: null weight pointer");
    float* w = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_nav_transpose, dim3(8), dim3(256), 0, st, f->sigma_w0, 64u, 32u, w + NV_PREP_W1T);
    hipLaunchKernelGGL(k_nav_transpose, dim3(4), dim3(256), 0, st, f->sigma_w1, 16u, 64u, w + NV_PREP_W2T);
    hipLaunchKernelGGL(k_nav_transpose, dim3(8), dim3(256), 0, st, f->color_w0, 64u, 31u, w + NV_PREP_V0T);
    hipLaunchKernelGGL(k_nav_transpose, dim3(16), dim3(256), 0, st, f->color_w1, 64u, 64u, w + NV_PREP_V1T);
    hipLaunchKernelGGL(k_nav_transpose, dim3(1), dim3(256), 0, st, f->color_w2, 3u, 64u, w + NV_PREP_V2T);
    NGP_CHECK_LAUNCH("nav_field_prepare");
    return NGP_OK;
}

// the kernels use slightly more than the default 64 KiB of dynamic LDS; the raised limit is a per-device function attribute
static int nav_allow_big_lds() {
    static std::atomic<unsigned long long> devices{0};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0) return ngp_fail(NGP_ELAUNCH, "nav: no current device");
    const unsigned long long bit = 1ull << (device & 63);
    if (device < 64 && (devices.load(std::memory_order_acquire) & bit)) return NGP_OK;
    const void* kernels[5] = {(const void*)k_nav_density_fwd, (const void*)k_nav_density_bwd, (const void*)k_nav_run_fwd, (const void*)k_nav_run_bwd,
                              (const void*)k_nav_density_vj};
    for (const void* k : kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return ngp_fail(NGP_ELAUNCH, "nav: cannot raise the dynamic LDS limit");
    devices.fetch_or(bit, std::memory_order_release);
    return NGP_OK;
}

static int nav_fill(const char* who, const ngp_nav_field_t* f, const void* prepared, nav_params& P) {
    NGP_REQUIRE(f && f->embeddings && f->offsets_host && f->sigma_w0 && f->sigma_w1 && f->color_w0 && f->color_w1 && f->color_w2 && prepared,
                "%s: null field pointer", who);
    NGP_REQUIRE(f->L == NV_L && f->bound > 0.0f, "%s: the fused nav queries are built for the default field (16 levels x 2 features, 32-64-16 | 31-64-64-3)", who);
    P.table = (const float2*)f->embeddings;
    const float* prep = (const float*)prepared;
    P.w1t = (nv_wptr)(uintptr_t)(prep + NV_PREP_W1T);
    P.w2t = (nv_wptr)(uintptr_t)(prep + NV_PREP_W2T);
    P.v0t = (nv_wptr)(uintptr_t)(prep + NV_PREP_V0T);
    P.v1t = (nv_wptr)(uintptr_t)(prep + NV_PREP_V1T);
    P.v2t = (nv_wptr)(uintptr_t)(prep + NV_PREP_V2T);
    P.w2 = (nv_wptr)(uintptr_t)f->sigma_w1; P.v0 = (nv_wptr)(uintptr_t)f->color_w0;
    P.v1 = (nv_wptr)(uintptr_t)f->color_w1; P.v2 = (nv_wptr)(uintptr_t)f->color_w2;
    for (int l = 0; l < NV_L; l++) {
        const float sc = exp2f((float)l * f->S) * (float)f->H - 1.0f;     // gridencoder.cu:126 on the host, as everywhere else
        const uint32_t rs = (uint32_t)ceilf(sc) + 1u;
        const uint32_t o0 = (uint32_t)f->offsets_host[l], size = (uint32_t)f->offsets_host[l + 1] - o0;
        uint32_t stride = 1, s1 = 0, s2 = 0;
        bool dense = true;
        for (int d = 0; d < 3; d++) {                                    // get_grid_index (gridencoder.cu:54-72)
            if (stride <= size) { if (d == 1) s1 = stride; if (d == 2) s2 = stride; stride *= rs + 1; }
            else dense = false;
        }
        if (stride > size) dense = false;
        P.lv.scale[l] = sc; P.lv.base[l] = o0; P.lv.size[l] = size;
        P.lv.mask[l] = (size & (size - 1)) == 0 ? size - 1 : 0u;
        P.lv.s1[l] = dense ? s1 : 0u; P.lv.s2[l] = dense ? s2 : 0u;
        NGP_REQUIRE(size > 0, "%s: empty level %d", who, l);
    }
    sh_fill_norm(P.shn);
    P.bound = f->bound;
    P.r2b = 1.0f / (2.0f * f->bound);
    P.density_scale = f->density_scale;
    return NGP_OK;
}

extern "C" int ngp_nav_density_forward(const ngp_nav_field_t* f, const void* prepared, const float* xyz, uint32_t M, float* sigma, float* geo,
                                       void* stream) {
    nav_params P;
    const int rc = nav_fill("nav_density_forward", f, prepared, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyz && sigma, "nav_density_forward: null pointer");
    { const int rc_lds = nav_allow_big_lds(); if (rc_lds != NGP_OK) return rc_lds; }
    hipLaunchKernelGGL(k_nav_density_fwd, dim3(ngp_div_up(M, NV_BLOCK)), dim3(NV_BLOCK), sizeof(float) * (16 + NV_H * NV_BLOCK), (hipStream_t)stream, P, xyz, M, sigma, geo);
    NGP_CHECK_LAUNCH("nav_density_forward");
    return NGP_OK;
}

extern "C" int ngp_nav_density_backward(const ngp_nav_field_t* f, const void* prepared, const float* xyz, uint32_t M, const float* grad_sigma,
                                        const float* grad_geo, float* grad_xyz, void* stream) {
    nav_params P;
    const int rc = nav_fill("nav_density_backward", f, prepared, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyz && grad_sigma && grad_xyz, "nav_density_backward: null pointer");
    { const int rc_lds = nav_allow_big_lds(); if (rc_lds != NGP_OK) return rc_lds; }
    hipLaunchKernelGGL(k_nav_density_bwd, dim3(ngp_div_up(M, NV_BLOCK)), dim3(NV_BLOCK), sizeof(float) * (16 + NV_H * NV_BLOCK), (hipStream_t)stream, P, xyz, M, grad_sigma, grad_geo, grad_xyz);
    NGP_CHECK_LAUNCH("nav_density_backward");
    return NGP_OK;
}

extern "C" int ngp_nav_density_value_jac(const ngp_nav_field_t* f, const void* prepared, const float* xyz, uint32_t M, const float* rot9_host, float* sigma,
                                         float* jac, void* stream) {
    nav_params P;
    const int rc = nav_fill("nav_density_value_jac", f, prepared, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyz && sigma && jac, "nav_density_value_jac: null pointer");
    nav_rot R;
    R.on = rot9_host ? 1u : 0u;
    for (int k = 0; k < 9; k++) R.m[k] = rot9_host ? rot9_host[k] : (k % 4 == 0 ? 1.0f : 0.0f);
    { const int rc_lds = nav_allow_big_lds(); if (rc_lds != NGP_OK) return rc_lds; }
    hipLaunchKernelGGL(k_nav_density_vj, dim3(ngp_div_up(M, NV_VJ_POINTS)), dim3(256), NV_VJ_LDS, (hipStream_t)stream, P, R, xyz, M, sigma, jac);
    NGP_CHECK_LAUNCH("nav_density_value_jac");
    return NGP_OK;
}

extern "C" size_t ngp_nav_run_saved_bytes(uint32_t N, uint32_t num_steps) { return sizeof(uint32_t) * NV_REC_WORDS * (size_t)N * num_steps; }

static int nav_run_args(const char* who, const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N, uint32_t T,
                        const float* aabb_host, const float* bg_host, nav_run& R) {
    NGP_REQUIRE(rays_o && rays_d && nears && fars && aabb_host, "%s: null pointer", who);
    NGP_REQUIRE(T >= 1 && T <= 4096, "%s: num_steps must be in [1, 4096]", who);
    R.rays_o = rays_o; R.rays_d = rays_d; R.nears = nears; R.fars = fars; R.N = N; R.T = T;
    for (int i = 0; i < 6; i++) R.aabb[i] = aabb_host[i];
    for (int i = 0; i < 3; i++) R.bg[i] = bg_host ? bg_host[i] : 1.0f;
    return NGP_OK;
}

extern "C" int ngp_nav_run_forward(const ngp_nav_field_t* f, const void* prepared, const float* rays_o, const float* rays_d, const float* nears,
                                   const float* fars, uint32_t N, uint32_t num_steps, const float* aabb_host, const float* bg_color3_host,
                                   float* image, float* depth, float* weights_sum, void* saved, size_t saved_bytes, void* stream) {
    nav_params P;
    int rc = nav_fill("nav_run_forward", f, prepared, P);
    if (rc != NGP_OK) return rc;
    if (N == 0) return NGP_OK;
    nav_run R;
    rc = nav_run_args("nav_run_forward", rays_o, rays_d, nears, fars, N, num_steps, aabb_host, bg_color3_host, R);
    if (rc != NGP_OK) return rc;
    NGP_REQUIRE(image && depth && weights_sum, "nav_run_forward: null output");
    NGP_REQUIRE(!saved || saved_bytes >= ngp_nav_run_saved_bytes(N, num_steps), "nav_run_forward: `saved` is smaller than ngp_nav_run_saved_bytes(N, num_steps)");
    const size_t lds = sizeof(float) * (16 + NV_H * NV_BLOCK);
    { const int rc_lds = nav_allow_big_lds(); if (rc_lds != NGP_OK) return rc_lds; }
    hipLaunchKernelGGL(k_nav_run_fwd, dim3(N), dim3(NV_BLOCK), lds, (hipStream_t)stream, P, R, image, depth, weights_sum, (uint32_t*)saved);
    NGP_CHECK_LAUNCH("nav_run_forward");
    return NGP_OK;
}

extern "C" int ngp_nav_run_backward(const ngp_nav_field_t* f, const void* prepared, const float* rays_o, const float* rays_d, const float* nears,
                                    const float* fars, uint32_t N, uint32_t num_steps, const float* aabb_host, const float* bg_color3_host,
                                    const float* grad_image, const float* grad_depth, const float* grad_weights_sum, const void* saved,
                                    size_t saved_bytes, float* grad_rays_o, float* grad_rays_d, void* stream) {
    nav_params P;
    int rc = nav_fill("nav_run_backward", f, prepared, P);
    if (rc != NGP_OK) return rc;
    if (N == 0) return NGP_OK;
    nav_run R;
    rc = nav_run_args("nav_run_backward", rays_o, rays_d, nears, fars, N, num_steps, aabb_host, bg_color3_host, R);
    if (rc != NGP_OK) return rc;
    NGP_REQUIRE(grad_image && grad_rays_o && grad_rays_d, "nav_run_backward: null pointer");
    NGP_REQUIRE(saved && saved_bytes >= ngp_nav_run_saved_bytes(N, num_steps), "nav_run_backward: needs the `saved` buffer its forward filled");
    const size_t lds = sizeof(float) * (16 + NV_H * NV_BLOCK);
    { const int rc_lds = nav_allow_big_lds(); if (rc_lds != NGP_OK) return rc_lds; }
    hipLaunchKernelGGL(k_nav_run_bwd, dim3(N), dim3(NV_BLOCK), lds, (hipStream_t)stream, P, R, grad_image, grad_depth, grad_weights_sum,
                       (const uint32_t*)saved, grad_rays_o, grad_rays_d);
    NGP_CHECK_LAUNCH("nav_run_backward");
    return NGP_OK;
}
