// ngp_mlp.h -- the 64-wide fully fused MLP on gfx950 matrix cores, shared by ffmlp.hip (the drop-in op) and
// render_fused.hip (the fused renderer).
//
// Orientation.  Everything is computed transposed, H'^T = W . H^T, with v_mfma_f32_16x16x32_f16:
//     A = W tile      16 (out features) x 32 (k)        lane l: row l&15,  k = 8*(l>>4) + j   (j = 0..7)
//     B = H^T tile    32 (k) x 16 (samples)             lane l: col l&15,  k = 8*(l>>4) + j
//     D = H'^T tile   16 (out features) x 16 (samples)  lane l: col l&15,  rows 4*(l>>4) + r  (r = 0..3)
// so a SAMPLE stays on the same lane column through the whole network: the D registers of one layer, after
// ReLU and a round to half, ARE the B operand of the next layer -- no LDS round trip and no shuffles between
// layers.  The price is a permuted k order inside each 32-deep step, which is free because the weights (A) are
// loaded once per wave in that same permuted order:
//     D tiles 2c and 2c+1 of lane group g hold features 32c + 4g + r and 32c + 16 + 4g + r, so element j of the
//     next B fragment of k-step c is feature  k(c,g,j) = 32c + 16*(j>>2) + 4g + (j&3).
// The first layer reads its B fragments straight from the row-major input (natural k order, one 16-byte load).
//
// Weights stay in VGPRs for the life of the wave (14 fragments = 56 VGPRs for the density net, 22 = 88 for the
// colour net); accumulation is binary32 inside the MFMA, activations are rounded to half once per layer, exactly
// where the reference stores them to shared memory as __half (ffmlp.cu:118).
#pragma once
#include "ngp_device.h"

typedef _Float16 ngp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ngp_h4 __attribute__((ext_vector_type(4)));
typedef float ngp_f4 __attribute__((ext_vector_type(4)));

static constexpr int MLP_W = 64;          // hidden width (the only one the reference's models use)
static constexpr int MLP_MT = MLP_W / 16; // 4 output-feature tiles per hidden layer

__device__ __forceinline__ ngp_f4 ngp_mfma(ngp_h8 a, ngp_h8 b, ngp_f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// A fragment of W[out][in] (row-major, leading dimension ld) for output tile t, k-step c, natural k order.
// Elements with k >= in_dim are zero (input widths that are not a multiple of 32).
__device__ __forceinline__ ngp_h8 mlp_load_a_natural(const _Float16* __restrict__ W, int ld, int in_dim, int t, int c, int lane) {
    const int row = 16 * t + (lane & 15), k0 = 32 * c + 8 * (lane >> 4);
    ngp_h8 a;
    if (k0 + 8 <= in_dim) {
        a = *reinterpret_cast<const ngp_h8*>(W + row * ld + k0);
    } else {
        #pragma unroll
        for (int j = 0; j < 8; j++) a[j] = (k0 + j < in_dim) ? W[row * ld + k0 + j] : (_Float16)0.0f;
    }
    return a;
}

// A fragment in the permuted k order k(c,g,j) used when the B operand is a previous layer's D registers.
__device__ __forceinline__ ngp_h8 mlp_load_a_permuted(const _Float16* __restrict__ W, int ld, int t, int c, int lane) {
    const int row = 16 * t + (lane & 15), g = lane >> 4;
    const ngp_h4 lo = *reinterpret_cast<const ngp_h4*>(W + row * ld + 32 * c + 4 * g);
    const ngp_h4 hi = *reinterpret_cast<const ngp_h4*>(W + row * ld + 32 * c + 16 + 4 * g);
    ngp_h8 a;
    a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
    a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
    return a;
}

// ReLU + round-to-half of two D tiles -> one B fragment of the next layer (k-step c = tiles 2c, 2c+1)
// (round first, then ReLU on packed halves: rounding is monotone and sign-preserving, so half(max(x,0)) == max(half(x),0)
//  bit for bit, and it is 4 v_cvt_pk + 4 v_pk_max instead of 8 v_max + 4 v_cvt_pk)
__device__ __forceinline__ ngp_h8 mlp_pack_relu(ngp_f4 d0, ngp_f4 d1) {
    ngp_h8 b;
    #pragma unroll
    for (int r = 0; r < 4; r++) {
        b[r] = (_Float16)d0[r];
        b[4 + r] = (_Float16)d1[r];
    }
    const ngp_h8 zero = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    return __builtin_elementwise_max(b, zero);
}

// The weights of one network held in registers.  NHID = number of hidden (64x64) matmuls = num_layers - 1,
// INC = ceil(input_dim / 32) k-steps of the first layer.
template <int NHID, int INC>
struct mlp_weights {
    ngp_h8 w_in[MLP_MT][INC];
    ngp_h8 w_hid[NHID][MLP_MT][2];
    ngp_h8 w_out[2];

    __device__ __forceinline__ void load(const _Float16* __restrict__ W, int input_dim, int lane) {
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++)
            #pragma unroll
            for (int c = 0; c < INC; c++) w_in[t][c] = mlp_load_a_natural(W, input_dim, input_dim, t, c, lane);
        const _Float16* Wh = W + MLP_W * input_dim;
        #pragma unroll
        for (int h = 0; h < NHID; h++)
            #pragma unroll
            for (int t = 0; t < MLP_MT; t++)
                #pragma unroll
                for (int c = 0; c < 2; c++) w_hid[h][t][c] = mlp_load_a_permuted(Wh + h * MLP_W * MLP_W, MLP_W, t, c, lane);
        const _Float16* Wo = Wh + NHID * MLP_W * MLP_W;
        #pragma unroll
        for (int c = 0; c < 2; c++) w_out[c] = mlp_load_a_permuted(Wo, MLP_W, 0, c, lane);
    }
};

// One 16-sample column tile through the whole network.
//   x[c]      : B fragments of the input (natural k order), c < INC
//   on_hidden : callback(layer m, tile t, packed 4 halves of features 16t+4g..+3) for saving activations
// returns the 16 x 16 output tile (binary32; rows = output features 4g + r of this lane's sample)
template <int NHID, int INC, typename F>
__device__ __forceinline__ ngp_f4 mlp_forward_tile(const mlp_weights<NHID, INC>& w, const ngp_h8 (&x)[INC], F&& on_hidden) {
    ngp_h8 act[2];
    {
        ngp_f4 d[MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            d[t] = ngp_f4{0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int c = 0; c < INC; c++) d[t] = ngp_mfma(w.w_in[t][c], x[c], d[t]);
        }
        act[0] = mlp_pack_relu(d[0], d[1]);
        act[1] = mlp_pack_relu(d[2], d[3]);
        on_hidden(0, act);
    }
    #pragma unroll
    for (int h = 0; h < NHID; h++) {
        ngp_f4 d[MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            d[t] = ngp_mfma(w.w_hid[h][t][0], act[0], ngp_f4{0.f, 0.f, 0.f, 0.f});
            d[t] = ngp_mfma(w.w_hid[h][t][1], act[1], d[t]);
        }
        act[0] = mlp_pack_relu(d[0], d[1]);
        act[1] = mlp_pack_relu(d[2], d[3]);
        on_hidden(h + 1, act);
    }
    ngp_f4 o = ngp_mfma(w.w_out[0], act[0], ngp_f4{0.f, 0.f, 0.f, 0.f});
    o = ngp_mfma(w.w_out[1], act[1], o);
    return o;
}
