// ffmlp.hip -- gfx950 kernels behind the `_ffmlp` native surface of the reference (ffmlp/src/ffmlp.h:8-14):
// the tiny-cuda-nn style fully fused MLP (bias-free linear layers, ReLU hidden activations, half precision).
//
// The reference runs WMMA 16x16x16 with half accumulation and keeps activations in shared memory
// (ffmlp.cu:47-129).  Here the network runs on v_mfma_f32_16x16x32_f16 in the transposed orientation of
// ngp_mlp.h: weights live in VGPRs, activations never leave registers, accumulation is binary32.
// Roofline: MFMA-shaped work, but at 64 wide the op is bound by streaming its operands:
//   forward   : input_dim*2 B in + 32 B out per sample (+ num_layers*128 B when activations are saved)
//   inference : same without the activation stores
// FLOPs per sample = 2 * (64*in + 64*64*(num_layers-1) + 64*16).
#include "ngp_mlp.h"
#include "ngp_ffmlp_generic.h"

template <int NHID, int INC, bool SAVE>
__global__ __launch_bounds__(256) void k_ffmlp_forward(const _Float16* __restrict__ X, const _Float16* __restrict__ W,
                                                       uint32_t B, int input_dim, _Float16* __restrict__ fb,
                                                       _Float16* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
    mlp_weights<NHID, INC> w;
    w.load(W, input_dim, lane);
    const uint32_t ntiles = B >> 4;
    for (uint32_t tile = wave; tile < ntiles; tile += nwaves) {
        const uint64_t s = (uint64_t)tile * 16 + (lane & 15);
        ngp_h8 x[INC];
        #pragma unroll
        for (int c = 0; c < INC; c++) {
            const int k0 = 32 * c + 8 * g;
            if (k0 + 8 <= input_dim) {
                x[c] = *reinterpret_cast<const ngp_h8*>(X + s * input_dim + k0);
            } else {
                #pragma unroll
                for (int j = 0; j < 8; j++) x[c][j] = (k0 + j < input_dim) ? X[s * input_dim + k0 + j] : (_Float16)0.0f;
            }
        }
        const ngp_f4 o = mlp_forward_tile<NHID, INC>(w, x, [&](int m, const ngp_h8 (&act)[2]) {
            if (SAVE) {
                _Float16* row = fb + ((uint64_t)m * B + s) * MLP_W;
                #pragma unroll
                for (int c = 0; c < 2; c++) {
                    ngp_h4 lo, hi;
                    #pragma unroll
                    for (int r = 0; r < 4; r++) { lo[r] = act[c][r]; hi[r] = act[c][4 + r]; }
                    *reinterpret_cast<ngp_h4*>(row + 32 * c + 4 * g) = lo;
                    *reinterpret_cast<ngp_h4*>(row + 32 * c + 16 + 4 * g) = hi;
                }
            }
        });
        ngp_h4 oh;
        #pragma unroll
        for (int r = 0; r < 4; r++) oh[r] = (_Float16)o[r];
        *reinterpret_cast<ngp_h4*>(out + s * 16 + 4 * g) = oh;
    }
}

template <int NHID, int INC>
static void ffmlp_launch(bool save, const void* X, const void* W, uint32_t B, uint32_t input_dim, void* fb, void* out, hipStream_t s) {
    const uint32_t ntiles = B >> 4;
    uint32_t blocks = ngp_div_up(ntiles, 4 * 8);      // >= 8 tiles per wave to amortise the weight load
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    if (save) hipLaunchKernelGGL((k_ffmlp_forward<NHID, INC, true>), dim3(blocks), dim3(256), 0, s,
                                 (const _Float16*)X, (const _Float16*)W, B, (int)input_dim, (_Float16*)fb, (_Float16*)out);
    else hipLaunchKernelGGL((k_ffmlp_forward<NHID, INC, false>), dim3(blocks), dim3(256), 0, s,
                            (const _Float16*)X, (const _Float16*)W, B, (int)input_dim, (_Float16*)fb, (_Float16*)out);
}

static int ffmlp_check(const char* who, const void* inputs, const void* weights, const void* outputs, uint32_t B, uint32_t input_dim,
                       uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation) {
    NGP_REQUIRE(B == 0 || (inputs && weights && outputs), "%s: null pointer", who);
    NGP_REQUIRE(hidden_dim == 64, "%s: hidden_dim must be 64 (the width every reference model uses)", who);
    NGP_REQUIRE(output_dim == 16, "%s: output_dim must be the padded width 16 (FFMLP pads, ffmlp.py:117)", who);
    NGP_REQUIRE(input_dim > 0 && input_dim % 16 == 0 && input_dim <= 64, "%s: input_dim must be 16, 32, 48 or 64", who);
    NGP_REQUIRE(num_layers >= 2 && num_layers <= 4, "%s: num_layers must be 2, 3 or 4", who);
    NGP_REQUIRE(activation == 0 && output_activation == 6, "%s: only ReLU hidden / no output activation (ffmlp.py:107-108)", who);
    NGP_REQUIRE(B % 16 == 0, "%s: batch must be a multiple of 16 (the wrapper pads to 128)", who);
    return NGP_OK;
}

static int ffmlp_dispatch(bool save, const void* X, const void* W, uint32_t B, uint32_t input_dim, uint32_t num_layers,
                          void* fb, void* out, hipStream_t s) {
    const int inc = (int)((input_dim + 31) / 32);
    const int nhid = (int)num_layers - 1;
    #define FF_CASE(NH, IC) if (nhid == NH && inc == IC) { ffmlp_launch<NH, IC>(save, X, W, B, input_dim, fb, out, s); return NGP_OK; }
    FF_CASE(1, 1) FF_CASE(1, 2) FF_CASE(2, 1) FF_CASE(2, 2) FF_CASE(3, 1) FF_CASE(3, 2)
    #undef FF_CASE
    return ngp_fail(NGP_EINVAL, "ffmlp: unsupported (num_layers=%u, input_dim=%u)", num_layers, input_dim);
}

extern "C" int ngp_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                 uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                 void* forward_buffer, void* outputs, void* stream) {
    if (!ffmlp_fast_shape(input_dim, output_dim, hidden_dim, num_layers, activation, output_activation)) {
        // every other shape / activation of the reference's module: layer by layer (ffmlp_generic.hip)
        int rg = ffmlp_generic_check("ffmlp_forward", B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
        if (rg != NGP_OK) return rg;
        if (B == 0) return NGP_OK;
        NGP_REQUIRE(inputs && weights && outputs && forward_buffer, "ffmlp_forward: null pointer");
        rg = ffmlp_generic_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, forward_buffer, outputs, (hipStream_t)stream);
        if (rg != NGP_OK) return rg;
        NGP_CHECK_LAUNCH("ffmlp_forward");
        return NGP_OK;
    }
    int rc = ffmlp_check("ffmlp_forward", inputs, weights, outputs, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
    if (rc != NGP_OK) return rc;
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(forward_buffer, "ffmlp_forward: null forward_buffer");
    rc = ffmlp_dispatch(true, inputs, weights, B, input_dim, num_layers, forward_buffer, outputs, (hipStream_t)stream);
    if (rc != NGP_OK) return rc;
    NGP_CHECK_LAUNCH("ffmlp_forward");
    return NGP_OK;
}

extern "C" int ngp_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                   uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                   void* inference_buffer, void* outputs, void* stream) {
    if (!ffmlp_fast_shape(input_dim, output_dim, hidden_dim, num_layers, activation, output_activation)) {
        int rg = ffmlp_generic_check("ffmlp_inference", B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
        if (rg != NGP_OK) return rg;
        if (B == 0) return NGP_OK;
        NGP_REQUIRE(inputs && weights && outputs && inference_buffer, "ffmlp_inference: null pointer (the layer-by-layer path needs the inference buffer)");
        rg = ffmlp_generic_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, inference_buffer, outputs, (hipStream_t)stream);
        if (rg != NGP_OK) return rg;
        NGP_CHECK_LAUNCH("ffmlp_inference");
        return NGP_OK;
    }
    (void)inference_buffer;                            // activations never leave registers
    int rc = ffmlp_check("ffmlp_inference", inputs, weights, outputs, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
    if (rc != NGP_OK) return rc;
    if (B == 0) return NGP_OK;
    rc = ffmlp_dispatch(false, inputs, weights, B, input_dim, num_layers, nullptr, outputs, (hipStream_t)stream);
    if (rc != NGP_OK) return rc;
    NGP_CHECK_LAUNCH("ffmlp_inference");
    return NGP_OK;
}

extern "C" int ngp_allocate_splitk(size_t size) { (void)size; return NGP_OK; }
extern "C" int ngp_free_splitk(void) { return NGP_OK; }
