// ffmlp_backward.hip -- backward of the fused MLP (reference: ffmlp/src/ffmlp.cu:410-518 for the activation
// gradients, :749-895 for the weight-gradient GEMMs that the reference hands to CUTLASS split-K on side streams).
//
//   k_ffmlp_bwd_act    per 16-sample tile, transposed orientation (ngp_mlp.h): G'^T = W^T . G^T with the ReLU mask of
//                      the saved forward activations; the D registers of one step are the B fragment of the next.
//                      Weight fragments are W^T in the permuted k order, gathered once per wave into VGPRs.
//                      Writes backward_buffer[k] (half, as the reference stores it) and, on request, grad_inputs.
//   k_ffmlp_bwd_wgrad  dW[o][i] = sum_s G[s][o] * A[s][i] for every layer in one launch (blockIdx.y = layer).  The
//                      contraction runs over samples, so both operands are transposed through LDS (each thread loads
//                      16 B of a row and scatters 8 halves to [feature][sample]); fragments are then one ds_read_b128.
//                      f32 accumulation in the MFMA, one f32 atomic per output element per workgroup into a workspace
//                      (18 K elements x 256 workgroups = 19 MB of atomics, ~15 us at the chip's 1.3 TB/s atomic rate),
//   k_ffmlp_bwd_cast   workspace f32 -> grad_weights f16 (the reference's dtype).
// The reference accumulates these GEMMs in half (cutlass_matmul.h:467-468); f32 accumulation is strictly more accurate.
#include "ngp_mlp.h"
#include "ngp_ffmlp_generic.h"

// A fragment of W^T for output tile t (rows = input index i), k-step c (k = output index o), permuted k order.
// W is row-major [n_out][ld]; rows o >= n_out read as zero (the 16-wide last layer padded to K = 32).
__device__ __forceinline__ ngp_h8 bwd_load_at_permuted(const _Float16* __restrict__ W, int ld, int n_out, int t, int c, int lane) {
    const int i = 16 * t + (lane & 15), g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        const int o = 32 * c + 16 * (j >> 2) + 4 * g + (j & 3);
        a[j] = (o < n_out) ? W[o * ld + i] : (_Float16)0.0f;
    }
    return a;
}
// same, natural k order (k = 8g + j): used for the first step, whose B fragment is loaded from the row-major grad
__device__ __forceinline__ ngp_h8 bwd_load_at_natural(const _Float16* __restrict__ W, int ld, int n_out, int t, int lane) {
    const int i = 16 * t + (lane & 15), g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        const int o = 8 * g + j;
        a[j] = (o < n_out) ? W[o * ld + i] : (_Float16)0.0f;
    }
    return a;
}

// ReLU backward from the saved post-activation (ffmlp/src/utils.h:537-582: pass the gradient where forward > 0),
// round to half, pack two D tiles into one B fragment, and store them to backward_buffer
__device__ __forceinline__ ngp_h8 bwd_mask_pack_store(ngp_f4 d0, ngp_f4 d1, const _Float16* __restrict__ fwd_row, _Float16* __restrict__ out_row,
                                                      int c, int g) {
    const ngp_h4 f0 = *reinterpret_cast<const ngp_h4*>(fwd_row + 32 * c + 4 * g);
    const ngp_h4 f1 = *reinterpret_cast<const ngp_h4*>(fwd_row + 32 * c + 16 + 4 * g);
    ngp_h8 b;
    ngp_h4 lo, hi;
    #pragma unroll
    for (int r = 0; r < 4; r++) {
        lo[r] = ((float)f0[r] > 0.0f) ? (_Float16)d0[r] : (_Float16)0.0f;
        hi[r] = ((float)f1[r] > 0.0f) ? (_Float16)d1[r] : (_Float16)0.0f;
        b[r] = lo[r]; b[4 + r] = hi[r];
    }
    *reinterpret_cast<ngp_h4*>(out_row + 32 * c + 4 * g) = lo;
    *reinterpret_cast<ngp_h4*>(out_row + 32 * c + 16 + 4 * g) = hi;
    return b;
}

template <int NHID, int INT, bool CALC_IN>   // INT = input_dim / 16 output tiles of the input-gradient step
__global__ __launch_bounds__(256) void k_ffmlp_bwd_act(const _Float16* __restrict__ grad, const _Float16* __restrict__ W,
                                                       const _Float16* __restrict__ fwd, uint32_t B, int input_dim,
                                                       _Float16* __restrict__ bwd, _Float16* __restrict__ grad_inputs) {
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
    const _Float16* W_hid = W + MLP_W * input_dim;
    const _Float16* W_last = W_hid + NHID * MLP_W * MLP_W;

    ngp_h8 wl[MLP_MT];                       // W_last^T  [64 x 16(+16 zero)]
    ngp_h8 wh[NHID][MLP_MT][2];              // W_hid[n-1-k]^T
    ngp_h8 wi[CALC_IN ? INT : 1][2];         // W_in^T    [in x 64]
    #pragma unroll
    for (int t = 0; t < MLP_MT; t++) wl[t] = bwd_load_at_natural(W_last, MLP_W, 16, t, lane);
    #pragma unroll
    for (int k = 0; k < NHID; k++)
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++)
            #pragma unroll
            for (int c = 0; c < 2; c++) wh[k][t][c] = bwd_load_at_permuted(W_hid + (NHID - 1 - k) * MLP_W * MLP_W, MLP_W, MLP_W, t, c, lane);
    if (CALC_IN) {
        #pragma unroll
        for (int t = 0; t < INT; t++)
            #pragma unroll
            for (int c = 0; c < 2; c++) wi[t][c] = bwd_load_at_permuted(W, input_dim, MLP_W, t, c, lane);
    }

    const ngp_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const uint32_t ntiles = B >> 4;
    for (uint32_t tile = wave; tile < ntiles; tile += nwaves) {
        const uint64_t s = (uint64_t)tile * 16 + (lane & 15);
        // B fragment of the incoming gradient: k = output index 8g + j, only 16 wide
        ngp_h8 gb;
        if (g < 2) gb = *reinterpret_cast<const ngp_h8*>(grad + s * 16 + 8 * g);
        else {
            #pragma unroll
            for (int j = 0; j < 8; j++) gb[j] = (_Float16)0.0f;
        }
        ngp_h8 act[2];
        {
            ngp_f4 d[MLP_MT];
            #pragma unroll
            for (int t = 0; t < MLP_MT; t++) d[t] = ngp_mfma(wl[t], gb, zero);
            const _Float16* frow = fwd + ((uint64_t)NHID * B + s) * MLP_W;          // forward_buffer[num_layers-1]
            _Float16* brow = bwd + s * MLP_W;                                        // backward_buffer[0]
            act[0] = bwd_mask_pack_store(d[0], d[1], frow, brow, 0, g);
            act[1] = bwd_mask_pack_store(d[2], d[3], frow, brow, 1, g);
        }
        #pragma unroll
        for (int k = 0; k < NHID; k++) {
            ngp_f4 d[MLP_MT];
            #pragma unroll
            for (int t = 0; t < MLP_MT; t++) {
                d[t] = ngp_mfma(wh[k][t][0], act[0], zero);
                d[t] = ngp_mfma(wh[k][t][1], act[1], d[t]);
            }
            const _Float16* frow = fwd + ((uint64_t)(NHID - 1 - k) * B + s) * MLP_W;
            _Float16* brow = bwd + ((uint64_t)(k + 1) * B + s) * MLP_W;
            act[0] = bwd_mask_pack_store(d[0], d[1], frow, brow, 0, g);
            act[1] = bwd_mask_pack_store(d[2], d[3], frow, brow, 1, g);
        }
        if (CALC_IN) {
            #pragma unroll
            for (int t = 0; t < INT; t++) {
                ngp_f4 d = ngp_mfma(wi[t][0], act[0], zero);
                d = ngp_mfma(wi[t][1], act[1], d);
                ngp_h4 o;
                #pragma unroll
                for (int r = 0; r < 4; r++) o[r] = (_Float16)d[r];
                *reinterpret_cast<ngp_h4*>(grad_inputs + s * input_dim + 16 * t + 4 * g) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------

struct wgrad_job { const _Float16* G; const _Float16* A; float* out; int go, ai; };   // dW[go][ai] = G[B][go]^T . A[B][ai], as one partial sum per workgroup column
struct wgrad_jobs { wgrad_job j[5]; int n; uint32_t row_stride; };                    // row_stride: floats between the partial-sum rows of consecutive blockIdx.x

static constexpr int WG_S = 32;                 // samples per step
static constexpr int WG_LD = WG_S + 8;          // LDS row stride in halves (80 B: 16-byte aligned rows, staggered banks)

__global__ __launch_bounds__(256) void k_ffmlp_bwd_wgrad(wgrad_jobs jobs, uint32_t B) {
    __shared__ __attribute__((aligned(16))) _Float16 Gt[64 * WG_LD];   // [feature][sample]
    __shared__ __attribute__((aligned(16))) _Float16 At[64 * WG_LD];
    const wgrad_job job = jobs.j[blockIdx.y];
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15, wave = threadIdx.x >> 6;
    const int nt = job.go >> 4, nu = job.ai >> 4;                       // output tiles: nt x nu (<= 4 x 4)
    ngp_f4 acc[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = ngp_f4{0.f, 0.f, 0.f, 0.f};

    const uint32_t nsteps = B / WG_S;                                   // B % 32 == 0 is checked on the host
    for (uint32_t step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const uint64_t s0 = (uint64_t)step * WG_S;
        // stage: thread th loads 8 halves of one row and scatters them to [feature][sample]
        for (int e = threadIdx.x; e < WG_S * (job.go >> 3); e += 256) {
            const int row = e / (job.go >> 3), f0 = (e % (job.go >> 3)) * 8;
            const ngp_h8 v = *reinterpret_cast<const ngp_h8*>(job.G + (s0 + row) * job.go + f0);
            #pragma unroll
            for (int j = 0; j < 8; j++) Gt[(f0 + j) * WG_LD + row] = v[j];
        }
        for (int e = threadIdx.x; e < WG_S * (job.ai >> 3); e += 256) {
            const int row = e / (job.ai >> 3), f0 = (e % (job.ai >> 3)) * 8;
            const ngp_h8 v = *reinterpret_cast<const ngp_h8*>(job.A + (s0 + row) * job.ai + f0);
            #pragma unroll
            for (int j = 0; j < 8; j++) At[(f0 + j) * WG_LD + row] = v[j];
        }
        __syncthreads();
        // wave w owns output tiles q*4 + w (tile id = t*nu + u)
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const int tid = q * 4 + wave;
            if (tid < nt * nu) {
                const int t = tid / nu, u = tid - t * nu;
                const ngp_h8 a = *reinterpret_cast<const ngp_h8*>(&Gt[(16 * t + r) * WG_LD + 8 * g]);   // A[row o][k = sample 8g+j]
                const ngp_h8 b = *reinterpret_cast<const ngp_h8*>(&At[(16 * u + r) * WG_LD + 8 * g]);   // B[k = sample][col i]
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
            for (int rr = 0; rr < 4; rr++)                              // D: row o = 16t + 4g + rr, col i = 16u + r; plain stores into this column's row
                job.out[(uint64_t)blockIdx.x * jobs.row_stride + (16 * t + 4 * g + rr) * job.ai + 16 * u + r] = acc[q][rr];
        }
    }
}

// ---------------------------------------------------------------------------

static uint32_t ffmlp_nparams(uint32_t in, uint32_t out, uint32_t hid, uint32_t nl) { return hid * (in + hid * (nl - 1) + out); }

extern "C" size_t ngp_ffmlp_backward_workspace(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers) {
    // (the layer-by-layer path also keeps transposed copies of the weights there)
    return ffmlp_generic_backward_workspace(input_dim, output_dim, hidden_dim, num_layers);
}

template <int NHID, int INT>
static void bwd_act_launch(bool calc, const void* grad, const void* W, const void* fwd, uint32_t B, uint32_t input_dim, void* bwd, void* gi, hipStream_t s) {
    const uint32_t ntiles = B >> 4;
    uint32_t blocks = ngp_div_up(ntiles, 4 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    if (calc) hipLaunchKernelGGL((k_ffmlp_bwd_act<NHID, INT, true>), dim3(blocks), dim3(256), 0, s, (const _Float16*)grad, (const _Float16*)W,
                                 (const _Float16*)fwd, B, (int)input_dim, (_Float16*)bwd, (_Float16*)gi);
    else hipLaunchKernelGGL((k_ffmlp_bwd_act<NHID, INT, false>), dim3(blocks), dim3(256), 0, s, (const _Float16*)grad, (const _Float16*)W,
                            (const _Float16*)fwd, B, (int)input_dim, (_Float16*)bwd, (_Float16*)gi);
}

extern "C" int ngp_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer,
                                  uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                                  uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,
                                  void* grad_inputs, void* grad_weights, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ffmlp_fast_shape(input_dim, output_dim, hidden_dim, num_layers, activation, output_activation)) {
        int rg = ffmlp_generic_check("ffmlp_backward", B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
        if (rg != NGP_OK) return rg;
        rg = ffmlp_generic_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, calc_grad_inputs,
                                    backward_buffer, grad_inputs, grad_weights, workspace, workspace_bytes, (hipStream_t)stream);
        if (rg != NGP_OK) return rg;
        NGP_CHECK_LAUNCH("ffmlp_backward");
        return NGP_OK;
    }
    NGP_REQUIRE(hidden_dim == 64 && output_dim == 16, "ffmlp_backward: hidden_dim must be 64 and output_dim the padded 16");
    NGP_REQUIRE(input_dim > 0 && input_dim % 16 == 0 && input_dim <= 64, "ffmlp_backward: input_dim must be 16, 32, 48 or 64");
    NGP_REQUIRE(num_layers >= 2 && num_layers <= 4, "ffmlp_backward: num_layers must be 2, 3 or 4");
    NGP_REQUIRE(activation == 0 && output_activation == 6, "ffmlp_backward: only ReLU hidden / no output activation");
    NGP_REQUIRE(B % 32 == 0, "ffmlp_backward: batch must be a multiple of 32 (the wrapper pads to 128)");
    const uint32_t nw = ffmlp_nparams(input_dim, output_dim, hidden_dim, num_layers);
    NGP_REQUIRE(grad_weights && workspace && workspace_bytes >= ffmlp_partial_bytes(nw), "ffmlp_backward: grad_weights / workspace missing or too small");
    hipStream_t s = (hipStream_t)stream;
    uint32_t gx = 0;                                                    // rows of partial sums written (0: an empty batch, all-zero gradients)
    if (B > 0) {
        NGP_REQUIRE(grad && inputs && weights && forward_buffer && backward_buffer, "ffmlp_backward: null pointer");
        NGP_REQUIRE(!calc_grad_inputs || grad_inputs, "ffmlp_backward: calc_grad_inputs needs grad_inputs");
        const int nhid = (int)num_layers - 1, it = (int)input_dim / 16;
        const bool calc = calc_grad_inputs != 0;
        bool launched = false;
        #define BW_CASE(NH, IT) if (nhid == NH && it == IT) { bwd_act_launch<NH, IT>(calc, grad, weights, forward_buffer, B, input_dim, backward_buffer, grad_inputs, s); launched = true; }
        BW_CASE(1, 1) BW_CASE(1, 2) BW_CASE(1, 3) BW_CASE(1, 4) BW_CASE(2, 1) BW_CASE(2, 2) BW_CASE(2, 3) BW_CASE(2, 4) BW_CASE(3, 1) BW_CASE(3, 2) BW_CASE(3, 3) BW_CASE(3, 4)
        #undef BW_CASE
        if (!launched) return ngp_fail(NGP_EINVAL, "ffmlp_backward: unsupported shape");

        // weight-gradient jobs (reference: ffmlp.cu:804-810, 851-857, 869-875)
        const _Float16* fb = (const _Float16*)forward_buffer;
        const _Float16* bb = (const _Float16*)backward_buffer;
        float* ws = (float*)workspace;
        const uint64_t BW = (uint64_t)B * 64;
        const uint32_t off_hid = 64 * input_dim, off_last = off_hid + (num_layers - 1) * 64 * 64;
        wgrad_jobs jobs;
        jobs.n = 0;
        jobs.j[jobs.n++] = wgrad_job{(const _Float16*)grad, fb + (uint64_t)(num_layers - 1) * BW, ws + off_last, 16, 64};
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            const uint32_t mi = num_layers - 2 - k;
            jobs.j[jobs.n++] = wgrad_job{bb + (uint64_t)k * BW, fb + (uint64_t)mi * BW, ws + off_hid + mi * 64 * 64, 64, 64};
        }
        jobs.j[jobs.n++] = wgrad_job{bb + (uint64_t)(num_layers - 1) * BW, (const _Float16*)inputs, ws, 64, (int)input_dim};
        gx = ngp_div_up(B / WG_S, 4);
        const uint32_t rows = ffmlp_partial_rows(nw);
        if (gx > rows) gx = rows;
        if (gx == 0) gx = 1;
        jobs.row_stride = nw;
        // no atomics: each workgroup column stores its partial sums as a row, added in a fixed order below (the reference's split-K CUTLASS GEMMs
        // accumulate in half in an order its streams decide, ffmlp.cu:804-875; float atomics here made two identical runs differ in the last bits)
        hipLaunchKernelGGL(k_ffmlp_bwd_wgrad, dim3(gx, jobs.n), dim3(256), 0, s, jobs, B);
    }
    ffmlp_sum_partials((const float*)workspace, gx, nw, grad_weights, s);
    NGP_CHECK_LAUNCH("ffmlp_backward");
    return NGP_OK;
}
