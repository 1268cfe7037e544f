// ffmlp_generic.hip -- see ngp_ffmlp_generic.h.  One launch per layer: Y = epilogue(X . W^T) with the matrix cores, a wave per 16 samples.
//
//   forward   Y = act(half(X W^T))                                        (ffmlp.cu:331-407; utils.h:423-470: the activation acts on the HALF pre-activation)
//   backward  G' = half(G W) * act'(forward output)                       (ffmlp.cu:410-518; utils.h:536-590: derivatives from the saved post-activations)
//             the same kernel on the transposed weights (transposed once per call into the workspace)
//   weights   dW = G^T A in 64 x 64 blocks, samples transposed through LDS, f32 atomics into the workspace, rounded to half at the end
// Orientation as in ngp_mlp.h: D[feature][sample] = sum_k W[feature][k] X[sample][k]; both operands are 16-byte row loads.
#include "ngp_mlp.h"
#include "ngp_ffmlp_generic.h"

enum : uint32_t { FG_RELU = 0, FG_EXP = 1, FG_SINE = 2, FG_SIGMOID = 3, FG_SQUAREPLUS = 4, FG_SOFTPLUS = 5, FG_NONE = 6 };
static constexpr float FG_K_ACT = 10.0f;              // utils.h:41

__device__ __forceinline__ float fg_h(float v) { return (float)ngp_f2h(v); }

// utils.h:423-470 on the half pre-activation x
__device__ __forceinline__ float fg_act(uint32_t act, float x) {
    switch (act) {
        case FG_RELU: return x > 0.0f ? x : 0.0f;
        case FG_EXP: return expf(x);
        case FG_SINE: return sinf(x);
        case FG_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case FG_SQUAREPLUS: { const float y = x * FG_K_ACT; return 0.5f * (y + sqrtf(y * y + 4.0f)) / FG_K_ACT; }
        case FG_SOFTPLUS: return logf(expf(x * FG_K_ACT) + 1.0f) / FG_K_ACT;
        default: return x;
    }
}

// utils.h:536-590: gradient g (half) times the derivative expressed through the forward OUTPUT y (half); every product is a half product there
__device__ __forceinline__ float fg_act_backward(uint32_t act, float g, float y) {
    switch (act) {
        case FG_RELU: return y > 0.0f ? g : 0.0f;
        case FG_EXP: return fg_h(g * y);
        case FG_SIGMOID: return fg_h(g * fg_h(y * fg_h(1.0f - y)));
        case FG_SQUAREPLUS: { const float t = y * FG_K_ACT; return fg_h(g * fg_h(t * t / (t * t + 1.0f))); }
        case FG_SOFTPLUS: return fg_h(g * fg_h(1.0f - expf(-y * FG_K_ACT)));
        default: return g;                            // None (Sine has no backward in the reference: refused on the host)
    }
}

// MODE 0: Y = act(X W^T)   MODE 1: Y = act_backward(X W^T, F)   MODE 2: Y = X W^T        X [B][K], W [N][K], Y [B][N], F [B][N] halves, row-major
template <int MODE>
__global__ __launch_bounds__(256) void k_fg_layer(const _Float16* __restrict__ X, const _Float16* __restrict__ W, _Float16* __restrict__ Y,
                                                  const _Float16* __restrict__ F, uint32_t B, int K, int N, uint32_t act) {
    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
    const int kc = (K + 31) >> 5;                                        // k-steps of 32 (the last one half empty when K % 32 == 16)
    for (uint32_t tile = wave; tile < (B + 15) / 16; tile += nwaves) {
        const uint32_t m = tile * 16 + s;
        ngp_h8 xb[8];                                                    // K <= 256
        #pragma unroll
        for (int c = 0; c < 8; c++) {
            const int k0 = 32 * c + 8 * g;
            ngp_h8 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (c < kc && k0 < K && m < B) v = *reinterpret_cast<const ngp_h8*>(X + (uint64_t)m * K + k0);
            xb[c] = v;
        }
        for (int t = 0; t < N / 16; t++) {
            ngp_f4 acc = {0.f, 0.f, 0.f, 0.f};
            const _Float16* wr = W + (uint64_t)(16 * t + s) * K + 8 * g;  // A fragment: row = feature 16 t + (lane & 15)
            #pragma unroll
            for (int c = 0; c < 8; c++) {
                if (c < kc) {
                    ngp_h8 a = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    if (32 * c + 8 * g < K) a = *reinterpret_cast<const ngp_h8*>(wr + 32 * c);
                    acc = ngp_mfma(a, xb[c], acc);
                }
            }
            if (m < B) {
                ngp_h4 out;
                ngp_h4 f = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                if (MODE == 1) f = *reinterpret_cast<const ngp_h4*>(F + (uint64_t)m * N + 16 * t + 4 * g);
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = fg_h(acc[r]);                        // the half the reference's WMMA accumulator holds
                    out[r] = MODE == 0 ? ngp_f2h(fg_act(act, v)) : MODE == 1 ? ngp_f2h(fg_act_backward(act, v, (float)f[r])) : ngp_f2h(v);
                }
                *reinterpret_cast<ngp_h4*>(Y + (uint64_t)m * N + 16 * t + 4 * g) = out;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_fg_transpose(const _Float16* __restrict__ src, int rows, int cols, _Float16* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;                        // dst [cols][rows]
    if (e < rows * cols) { const int c = e / rows, r = e - c * rows; dst[e] = src[r * cols + c]; }
}

// dW[go][ai] (f32, row stride ai) += G[B][go]^T A[B][ai]; blockIdx.y = 64 x 64 block of dW, blockIdx.x strides over 32-sample steps
static constexpr int FGW_S = 32, FGW_LD = FGW_S + 8;
__global__ __launch_bounds__(256) void k_fg_wgrad(const _Float16* __restrict__ G, int go, const _Float16* __restrict__ A, int ai, float* __restrict__ out, uint32_t B,
                                                  uint32_t row_stride /* floats between the partial-sum rows of consecutive blockIdx.x */) {
    __shared__ __attribute__((aligned(16))) _Float16 Gt[64 * FGW_LD];
    __shared__ __attribute__((aligned(16))) _Float16 At[64 * FGW_LD];
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15, wave = threadIdx.x >> 6;
    const int nbi = (ai + 63) / 64, bo = blockIdx.y / nbi, bi = blockIdx.y - bo * nbi;
    const int o0 = 64 * bo, i0 = 64 * bi, no = go - o0 < 64 ? go - o0 : 64, ni = ai - i0 < 64 ? ai - i0 : 64;
    const int nt = no >> 4, nu = ni >> 4;
    ngp_f4 acc[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = ngp_f4{0.f, 0.f, 0.f, 0.f};
    const uint32_t nsteps = (B + FGW_S - 1) / FGW_S;
    for (uint32_t step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const uint64_t s0 = (uint64_t)step * FGW_S;
        for (int e = threadIdx.x; e < FGW_S * (no >> 3); e += 256) {
            const int row = e / (no >> 3), f0 = (e % (no >> 3)) * 8;
            ngp_h8 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (s0 + row < B) v = *reinterpret_cast<const ngp_h8*>(G + (s0 + row) * go + o0 + f0);
            #pragma unroll
            for (int j = 0; j < 8; j++) Gt[(f0 + j) * FGW_LD + row] = v[j];
        }
        for (int e = threadIdx.x; e < FGW_S * (ni >> 3); e += 256) {
            const int row = e / (ni >> 3), f0 = (e % (ni >> 3)) * 8;
            ngp_h8 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (s0 + row < B) v = *reinterpret_cast<const ngp_h8*>(A + (s0 + row) * ai + i0 + f0);
            #pragma unroll
            for (int j = 0; j < 8; j++) At[(f0 + j) * FGW_LD + row] = v[j];
        }
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const int tid = q * 4 + wave;
            if (tid < nt * nu) {
                const int t = tid / nu, u = tid - t * nu;
                const ngp_h8 a = *reinterpret_cast<const ngp_h8*>(&Gt[(16 * t + r) * FGW_LD + 8 * g]);
                const ngp_h8 b = *reinterpret_cast<const ngp_h8*>(&At[(16 * u + r) * FGW_LD + 8 * g]);
                acc[q] = ngp_mfma(a, b, acc[q]);
            }
        }
        __syncthreads();
    }
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        const int tid = q * 4 + wave;
        if (tid < nt * nu) {
            const int t = tid / nu, u = tid - t * nu;
            #pragma unroll
            for (int rr = 0; rr < 4; rr++)                              // this workgroup column's row of the partial sums: plain stores, every element
                out[(uint64_t)blockIdx.x * row_stride + (uint64_t)(o0 + 16 * t + 4 * g + rr) * ai + i0 + 16 * u + r] = acc[q][rr];
        }
    }
}

// [rows][n] f32 partial sums -> half gradients: 16 chunks of consecutive rows, each summed front to back by one thread, then the chunk sums front to
// back -- a fixed order, whatever order the workgroups that wrote the rows finished in
static constexpr uint32_t FG_SUM_COLS = 64, FG_SUM_CHUNKS = 16;
__global__ __launch_bounds__(FG_SUM_COLS * FG_SUM_CHUNKS) void k_fg_sum_partials(const float* __restrict__ ws, uint32_t rows, uint32_t n, _Float16* __restrict__ gw) {
    __shared__ float part[FG_SUM_CHUNKS][FG_SUM_COLS];
    const uint32_t col = threadIdx.x % FG_SUM_COLS, chunk = threadIdx.x / FG_SUM_COLS;
    const uint32_t i = blockIdx.x * FG_SUM_COLS + col;
    const uint32_t per = (rows + FG_SUM_CHUNKS - 1) / FG_SUM_CHUNKS;
    const uint32_t r0 = chunk * per, r1 = r0 + per < rows ? r0 + per : rows;
    float sum = 0.0f;
    if (i < n)
        for (uint32_t r = r0; r < r1; r++) sum += ws[(size_t)r * n + i];
    part[chunk][col] = sum;
    __syncthreads();
    if (chunk == 0 && i < n) {
        float v = part[0][col];
        #pragma unroll
        for (uint32_t c = 1; c < FG_SUM_CHUNKS; c++) v += part[c][col];
        gw[i] = (_Float16)v;
    }
}

uint32_t ffmlp_partial_rows(uint32_t nw) {
    const uint64_t fit = (64ull << 20) / (4ull * (nw ? nw : 1u));
    return (uint32_t)(fit > 256 ? 256 : (fit ? fit : 1));
}
size_t ffmlp_partial_bytes(uint32_t nw) { return ((size_t)ffmlp_partial_rows(nw) * nw * sizeof(float) + 255) & ~(size_t)255; }
void ffmlp_sum_partials(const float* partials, uint32_t rows, uint32_t nw, void* grad_weights_half, hipStream_t s) {
    hipLaunchKernelGGL(k_fg_sum_partials, dim3(ngp_div_up(nw, FG_SUM_COLS)), dim3(FG_SUM_COLS * FG_SUM_CHUNKS), 0, s, partials, rows, nw, (_Float16*)grad_weights_half);
}

// ---------------------------------------------------------------------------

bool ffmlp_fast_shape(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation) {
    return hidden_dim == 64 && output_dim == 16 && input_dim > 0 && input_dim % 16 == 0 && input_dim <= 64 && num_layers >= 2 && num_layers <= 4 &&
           activation == FG_RELU && output_activation == FG_NONE;
}

int ffmlp_generic_check(const char* who, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                        uint32_t activation, uint32_t output_activation) {
    (void)B;
    NGP_REQUIRE(hidden_dim == 16 || hidden_dim == 32 || hidden_dim == 64 || hidden_dim == 128 || hidden_dim == 256,
                "%s: hidden_dim should be in [16, 32, 64, 128, 256] (ffmlp.cu:659)", who);
    NGP_REQUIRE(output_dim == 16, "%s: output_dim must be the padded width 16 (ffmlp.py:112,117)", who);
    NGP_REQUIRE(input_dim > 0 && input_dim % 16 == 0 && input_dim <= 256, "%s: input_dim must be a multiple of 16 (ffmlp.py:111), at most 256 here", who);
    NGP_REQUIRE(num_layers >= 2 && num_layers <= 16, "%s: num_layers must be in 2..16", who);
    NGP_REQUIRE(activation <= FG_NONE, "%s: unknown activation %u", who, activation);
    NGP_REQUIRE(output_activation == FG_NONE, "%s: output activation is not supported (ffmlp.py:108)", who);
    return NGP_OK;
}

static uint32_t fg_blocks(uint32_t B) {
    uint32_t b = ngp_div_up(ngp_div_up(B, 16), 4);
    return b > 2048 ? 2048 : (b ? b : 1);
}

int ffmlp_generic_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                          uint32_t num_layers, uint32_t activation, void* buffer, void* outputs, hipStream_t s) {
    NGP_REQUIRE(buffer, "ffmlp (layer-by-layer path): the forward / inference buffer is needed");
    const _Float16* W = (const _Float16*)weights;
    _Float16* fb = (_Float16*)buffer;
    const uint64_t BW = (uint64_t)B * hidden_dim;
    const int H = (int)hidden_dim;
    hipLaunchKernelGGL(k_fg_layer<0>, dim3(fg_blocks(B)), dim3(256), 0, s, (const _Float16*)inputs, W, fb, (const _Float16*)nullptr, B, (int)input_dim, H, activation);
    W += (uint64_t)H * input_dim;
    for (uint32_t l = 1; l < num_layers; l++) {
        hipLaunchKernelGGL(k_fg_layer<0>, dim3(fg_blocks(B)), dim3(256), 0, s, fb + (l - 1) * BW, W, fb + l * BW, (const _Float16*)nullptr, B, H, H, activation);
        W += (uint64_t)H * H;
    }
    hipLaunchKernelGGL(k_fg_layer<2>, dim3(fg_blocks(B)), dim3(256), 0, s, fb + (num_layers - 1) * BW, W, (_Float16*)outputs, (const _Float16*)nullptr, B, H, (int)output_dim, (uint32_t)FG_NONE);
    return NGP_OK;
}

static uint32_t fg_nparams(uint32_t in, uint32_t out, uint32_t hid, uint32_t nl) { return hid * (in + hid * (nl - 1) + out); }

size_t ffmlp_generic_backward_workspace(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers) {
    const uint32_t nw = fg_nparams(input_dim, output_dim, hidden_dim, num_layers);
    return ffmlp_partial_bytes(nw) + sizeof(_Float16) * (size_t)nw + 256;       // [partial sums of the weight gradients | transposed weights]
}

int ffmlp_generic_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B, uint32_t input_dim,
                           uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, int calc_grad_inputs,
                           void* backward_buffer, void* grad_inputs, void* grad_weights, void* workspace, size_t workspace_bytes, hipStream_t s) {
    NGP_REQUIRE(activation != FG_SINE, "ffmlp_backward: the Sine activation has no backward (the reference keeps post-activations only, utils.h:552-556)");
    const uint32_t nw = fg_nparams(input_dim, output_dim, hidden_dim, num_layers);
    NGP_REQUIRE(grad_weights && workspace && workspace_bytes >= ffmlp_generic_backward_workspace(input_dim, output_dim, hidden_dim, num_layers),
                "ffmlp_backward: grad_weights / workspace missing or too small");
    float* ws = (float*)workspace;
    _Float16* wt = (_Float16*)((unsigned char*)workspace + ffmlp_partial_bytes(nw));
    uint32_t gx = 0;                                                        // rows of partial sums written (0: an empty batch, all-zero gradients)
    if (B > 0) {
        NGP_REQUIRE(grad && inputs && weights && forward_buffer && backward_buffer, "ffmlp_backward: null pointer");
        NGP_REQUIRE(!calc_grad_inputs || grad_inputs, "ffmlp_backward: calc_grad_inputs needs grad_inputs");
        const int H = (int)hidden_dim;
        const _Float16* W = (const _Float16*)weights;
        const _Float16* fb = (const _Float16*)forward_buffer;
        _Float16* bb = (_Float16*)backward_buffer;
        const uint64_t BW = (uint64_t)B * hidden_dim;
        const uint32_t off_hid = hidden_dim * input_dim, off_last = off_hid + (num_layers - 1) * hidden_dim * hidden_dim;
        // transposed copies: W_last^T [H][16], hidden^T [H][H] each, W_in^T [in][H]
        hipLaunchKernelGGL(k_fg_transpose, dim3(ngp_div_up(H * output_dim, 256)), dim3(256), 0, s, W + off_last, (int)output_dim, H, wt + off_last);
        for (uint32_t l = 0; l + 1 < num_layers; l++)
            hipLaunchKernelGGL(k_fg_transpose, dim3(ngp_div_up(H * H, 256)), dim3(256), 0, s, W + off_hid + l * H * H, H, H, wt + off_hid + l * H * H);
        if (calc_grad_inputs)
            hipLaunchKernelGGL(k_fg_transpose, dim3(ngp_div_up(H * input_dim, 256)), dim3(256), 0, s, W, H, (int)input_dim, wt);
        // activation gradients (ffmlp.cu:410-518): bb[0] from the output gradient, bb[k+1] from bb[k]
        hipLaunchKernelGGL(k_fg_layer<1>, dim3(fg_blocks(B)), dim3(256), 0, s, (const _Float16*)grad, wt + off_last, bb, fb + (num_layers - 1) * BW, B, (int)output_dim, H, activation);
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            const uint32_t mi = num_layers - 2 - k;
            hipLaunchKernelGGL(k_fg_layer<1>, dim3(fg_blocks(B)), dim3(256), 0, s, bb + k * BW, wt + off_hid + mi * H * H, bb + (k + 1) * BW, fb + mi * BW, B, H, H, activation);
        }
        if (calc_grad_inputs)
            hipLaunchKernelGGL(k_fg_layer<2>, dim3(fg_blocks(B)), dim3(256), 0, s, bb + (num_layers - 1) * BW, wt, (_Float16*)grad_inputs, (const _Float16*)nullptr, B, H, (int)input_dim, (uint32_t)FG_NONE);
        // weight gradients (ffmlp.cu:804-810, 851-857, 869-875)
        gx = ngp_div_up(ngp_div_up(B, FGW_S), 4);
        const uint32_t rows = ffmlp_partial_rows(nw);
        gx = gx > rows ? rows : (gx ? gx : 1);
        const int nbh = (H + 63) / 64, nbin = ((int)input_dim + 63) / 64;
        hipLaunchKernelGGL(k_fg_wgrad, dim3(gx, 1 * nbh), dim3(256), 0, s, (const _Float16*)grad, (int)output_dim, fb + (num_layers - 1) * BW, H, ws + off_last, B, nw);
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            const uint32_t mi = num_layers - 2 - k;
            hipLaunchKernelGGL(k_fg_wgrad, dim3(gx, nbh * nbh), dim3(256), 0, s, bb + k * BW, H, fb + mi * BW, H, ws + off_hid + mi * H * H, B, nw);
        }
        hipLaunchKernelGGL(k_fg_wgrad, dim3(gx, nbh * nbin), dim3(256), 0, s, bb + (num_layers - 1) * BW, H, (const _Float16*)inputs, (int)input_dim, ws, B, nw);
    }
    ffmlp_sum_partials(ws, gx, nw, grad_weights, s);
    return NGP_OK;
}
