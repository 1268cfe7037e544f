// shencoder.hip -- gfx950 kernels behind the `_shencoder` native surface of the reference
// (shencoder/src/shencoder.h:10,13): real spherical-harmonics direction encoding of degree 1..8 and its
// analytic Jacobian.
//
// The reference hard-codes one polynomial per output and per partial derivative (shencoder.cu:50-355).  Here
// the same polynomials are produced by recurrences, which is shorter, covers every degree uniformly and lets
// the Jacobian fall out of the same tables:
//
//   Y[l*l + l + m] = N(l,|m|) * Q_l^|m|(z) * { A_m  (m >= 0) ;  B_|m|  (m < 0) }
//     A_m + i B_m = (x + i y)^m                        (harmonic part, recurrence in m)
//     Q_l^m(z)    = d^m/dz^m P_l(z)                    (Legendre derivative polynomials, recurrence in l)
//     N(l,m)      = (-1)^m sqrt((2 - [m==0]) (2l+1)/(4 pi) (l-m)!/(l+m)!)
//   dY/dz = N * Q_l^(m+1) * {A,B} ;  dA_m/dx = m A_(m-1), dA_m/dy = -m B_(m-1), dB_m/dx = m B_(m-1), dB_m/dy = m A_(m-1)
//
// These are the reference's polynomials exactly (it also treats x, y, z as independent and uses the unit-sphere
// form in z), evaluated in a different order, so results agree to binary32 rounding (tests: 2e-6 abs at degree 4).
// Bound: 12 B in, 4*C^2 (+ 12*C^2) B out per sample -- pure HBM streaming, negligible next to the grid encoder.
#include "ngp_sh.h"

template <uint32_t C, bool GRAD>
__global__ __launch_bounds__(256) void k_sh_forward(const float* __restrict__ inputs, float* __restrict__ outputs,
                                                    uint32_t B, uint32_t D, sh_norm nrm, float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float x = inputs[(uint64_t)b * D], y = inputs[(uint64_t)b * D + 1], z = inputs[(uint64_t)b * D + 2];
    sh_tables<C> tb;
    tb.build(x, y, z);
    const float (&A)[C] = tb.A;
    const float (&Bm)[C] = tb.B;
    const float (&Q)[C][C + 1] = tb.Q;

    float* out = outputs + (uint64_t)b * C * C;
    float* jx = dy_dx + (uint64_t)b * D * C * C;
    float* jy = jx + C * C;
    float* jz = jy + C * C;
    #pragma unroll
    for (uint32_t l = 0; l < C; l++) {
        #pragma unroll
        for (uint32_t m = 0; m <= l; m++) {
            const float nq = nrm.n[l][m] * Q[l][m];
            const uint32_t ip = l * l + l + m, in = l * l + l - m;
            out[ip] = nq * A[m];
            if (m > 0) out[in] = nq * Bm[m];
            if (GRAD) {
                const float nqz = nrm.n[l][m] * Q[l][m + 1];
                jz[ip] = nqz * A[m];
                if (m > 0) {
                    const float fm = (float)m;
                    jz[in] = nqz * Bm[m];
                    jx[ip] = nq * (fm * A[m - 1]);
                    jy[ip] = nq * (-fm * Bm[m - 1]);
                    jx[in] = nq * (fm * Bm[m - 1]);
                    jy[in] = nq * (fm * A[m - 1]);
                } else {
                    jx[ip] = 0.0f;
                    jy[ip] = 0.0f;
                }
            }
        }
    }
}

// grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]   (reference: shencoder.cu:359-383, sequential in ch)
__global__ __launch_bounds__(256) void k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                                                     const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const uint32_t d = t - b * D;
    const float* g = grad + (uint64_t)b * C2;
    const float* j = dy_dx + ((uint64_t)b * D + d) * C2;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) acc += g[ch] * j[ch];
    grad_inputs[t] = acc;
}

template <uint32_t C>
static void sh_launch(const float* inputs, float* outputs, uint32_t B, uint32_t D, const sh_norm& nrm, bool calc, float* dy_dx, hipStream_t s) {
    if (calc) hipLaunchKernelGGL((k_sh_forward<C, true>), dim3(ngp_div_up(B, 256)), dim3(256), 0, s, inputs, outputs, B, D, nrm, dy_dx);
    else hipLaunchKernelGGL((k_sh_forward<C, false>), dim3(ngp_div_up(B, 256)), dim3(256), 0, s, inputs, outputs, B, D, nrm, dy_dx);
}

extern "C" int ngp_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                                     int calc_grad_inputs, float* dy_dx, void* stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && outputs, "sh_encode_forward: null pointer");
    NGP_REQUIRE(D == 3, "SH encoder only support input dim == 3");
    NGP_REQUIRE(C >= 1 && C <= SH_MAX, "SH encoder only supports degree in [1, 8]");
    NGP_REQUIRE(!calc_grad_inputs || dy_dx, "sh_encode_forward: calc_grad_inputs needs dy_dx");
    sh_norm nrm;
    sh_fill_norm(nrm);
    hipStream_t s = (hipStream_t)stream;
    const bool calc = calc_grad_inputs != 0;
    switch (C) {
        case 1: sh_launch<1>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
        case 2: sh_launch<2>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
        case 3: sh_launch<3>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
        case 4: sh_launch<4>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
        case 5: sh_launch<5>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
        case 6: sh_launch<6>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
        case 7: sh_launch<7>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
        default: sh_launch<8>(inputs, outputs, B, D, nrm, calc, dy_dx, s); break;
    }
    NGP_CHECK_LAUNCH("sh_encode_forward");
    return NGP_OK;
}

extern "C" int ngp_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                                      const float* dy_dx, float* grad_inputs, void* stream) {
    (void)inputs;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: null pointer");
    NGP_REQUIRE(D == 3 && C >= 1 && C <= SH_MAX, "sh_encode_backward: D must be 3 and degree in [1, 8]");
    hipLaunchKernelGGL(k_sh_backward, dim3(ngp_div_up((uint64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad, B, D, C * C, dy_dx, grad_inputs);
    NGP_CHECK_LAUNCH("sh_encode_backward");
    return NGP_OK;
}
