// ngp_sh.h -- real spherical harmonics by recurrence, shared by shencoder.hip and render_fused.hip so that both
// evaluate the identical operation sequence (see shencoder.hip for the derivation and the reference lines).
#pragma once
#include "ngp_device.h"
#include <math.h>

static constexpr uint32_t SH_MAX = 8;

struct sh_norm { float n[SH_MAX][SH_MAX]; };      // n[l][m], m <= l : (-1)^m sqrt((2-[m==0]) (2l+1)/(4 pi) (l-m)!/(l+m)!)

static inline void sh_fill_norm(sh_norm& t) {
    for (uint32_t l = 0; l < SH_MAX; l++)
        for (uint32_t m = 0; m <= l; m++) {
            double r = 1.0;                        // (l-m)! / (l+m)!
            for (uint32_t k = l - m + 1; k <= l + m; k++) r /= (double)k;
            const double v = sqrt((m == 0 ? 1.0 : 2.0) * (2.0 * l + 1.0) / (4.0 * M_PI) * r);
            t.n[l][m] = (float)((m & 1u) ? -v : v);
        }
}

// harmonic part A_m + i B_m = (x + i y)^m and Legendre-derivative table Q[l][m] = d^m/dz^m P_l(z), m <= l (Q[l][l+1] = 0)
template <uint32_t C>
struct sh_tables {
    float A[C], B[C], Q[C][C + 1];
    __device__ __forceinline__ void build(float x, float y, float z) {
        A[0] = 1.0f; B[0] = 0.0f;
        #pragma unroll
        for (uint32_t m = 1; m < C; m++) {
            A[m] = x * A[m - 1] - y * B[m - 1];
            B[m] = x * B[m - 1] + y * A[m - 1];
        }
        #pragma unroll
        for (uint32_t m = 0; m < C; m++) {
            float dfact = 1.0f;                     // (2m-1)!!
            #pragma unroll
            for (uint32_t k = 1; k <= m; k++) dfact *= (float)(2 * k - 1);
            Q[m][m] = dfact;
            Q[m][m + 1] = 0.0f;
            if (m + 1 < C) Q[m + 1][m] = (float)(2 * m + 1) * z * dfact;
            #pragma unroll
            for (uint32_t l = m + 2; l < C; l++) {
                const float a = (float)(2 * l - 1) / (float)(l - m);
                const float c = (float)(l + m - 1) / (float)(l - m);
                Q[l][m] = a * z * Q[l - 1][m] - c * Q[l - 2][m];
            }
        }
    }
};

// out[l*l + l +- m] for all l < C
template <uint32_t C>
__device__ __forceinline__ void sh_eval(float x, float y, float z, const sh_norm& nrm, float (&out)[C * C]) {
    sh_tables<C> t;
    t.build(x, y, z);
    #pragma unroll
    for (uint32_t l = 0; l < C; l++) {
        #pragma unroll
        for (uint32_t m = 0; m <= l; m++) {
            const float nq = nrm.n[l][m] * t.Q[l][m];
            out[l * l + l + m] = nq * t.A[m];
            if (m > 0) out[l * l + l - m] = nq * t.B[m];
        }
    }
}
