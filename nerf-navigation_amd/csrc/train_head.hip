// train_head.hip -- the few lines of torch arithmetic between the compositor and the loss of a training step, as four small launches.
//
// Reference (Python, elementwise torch ops on [N] / [N,3] tensors of a 4,096-ray batch):
//     nerf/renderer.py:318-319   image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
//                                depth = torch.clamp(depth - nears, min=0) / (fars - nears)
//     nerf/utils.py:450,480      loss = criterion(pred_rgb, gt_rgb).mean(-1) ... loss.mean()         (MSELoss: the mean of the squared differences)
//     nerf/utils.py:789          scaler.scale(loss).backward()
// Under autograd that is 10 launches forward and 10 backward, ~5 us each on a GPU that has nothing else queued: 0.10 ms of a 1.3 ms step, the same as the
// compositor's two kernels together.  Here: mix forward (1), loss forward (1: the loss, the scaled loss and d loss / d pred in one pass), loss backward
// (1: d loss / d pred times the incoming scalar), mix backward (1: grad_weights_sum; grad_image IS the incoming gradient).
// Arithmetic: the reference's operations in the reference's order in float32 (this library is built with -ffp-contract=off); the only liberty is the
// order in which the mean adds its terms (torch's reduce kernel has its own tree; here a fixed one, so the loss is run-to-run identical).
#include "ngp_device.h"

#define TH_BLOCK 256

// bg_rows: 0 = bg_value for every ray and channel (bg_color = 1, the trainer's case), 1 = bg[3] for every ray, N = bg[N,3]
__device__ __forceinline__ float th_bg(const float* __restrict__ bg, uint32_t bg_rows, float bg_value, uint32_t n, uint32_t c) {
    return bg_rows == 0 ? bg_value : (bg_rows == 1 ? bg[c] : bg[3ull * n + c]);
}

__global__ __launch_bounds__(TH_BLOCK) void k_train_mix_fwd(const float* __restrict__ weights_sum, const float* __restrict__ depth, const float* __restrict__ image,
                                                         const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ bg,
                                                         uint32_t bg_rows, float bg_value, uint32_t N, float* __restrict__ out_image, float* __restrict__ out_depth) {
    const uint32_t n = blockIdx.x * TH_BLOCK + threadIdx.x;
    if (n >= N) return;
    const float rest = 1.0f - weights_sum[n];
    #pragma unroll
    for (uint32_t c = 0; c < 3; c++) out_image[3ull * n + c] = image[3ull * n + c] + rest * th_bg(bg, bg_rows, bg_value, n, c);
    if (out_depth) {
        const float d = depth[n] - nears[n];
        out_depth[n] = (d < 0.0f ? 0.0f : d) / (fars[n] - nears[n]);          // torch.clamp(min = 0) keeps a NaN, fmaxf would not
    }
}

// grad_weights_sum = -(g . bg): autograd's path is (g * bg).sum(-1) through the unsqueeze, then the minus of `1 - ws`
__global__ __launch_bounds__(TH_BLOCK) void k_train_mix_bwd(const float* __restrict__ grad_out_image, const float* __restrict__ bg, uint32_t bg_rows, float bg_value,
                                                         uint32_t N, float* __restrict__ grad_weights_sum) {
    const uint32_t n = blockIdx.x * TH_BLOCK + threadIdx.x;
    if (n >= N) return;
    float s = grad_out_image[3ull * n] * th_bg(bg, bg_rows, bg_value, n, 0);
    s = s + grad_out_image[3ull * n + 1] * th_bg(bg, bg_rows, bg_value, n, 1);
    s = s + grad_out_image[3ull * n + 2] * th_bg(bg, bg_rows, bg_value, n, 2);
    grad_weights_sum[n] = -s;
}

// ---- mean squared error: loss, loss * scale and the gradient with respect to pred for a unit incoming gradient -----------------------------------------
// Workgroups sum their 4,096-element slices (fixed tree), the last one to finish adds the partial sums in index order: one launch, deterministic.
// workspace: [0] ticket (uint32, zero before the first call; the last workgroup puts the zero back), [1..] one partial sum per workgroup.
#define TH_SLICE (TH_BLOCK * 16)
#define TH_MAX_BLOCKS 1024

__device__ __forceinline__ float th_block_sum(float v, float* lds) {
    #pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
    if (threadIdx.x == 0) {
        #pragma unroll
        for (int w = 0; w < TH_BLOCK / 64; w++) s += lds[w];
    }
    return s;                                                                 // valid in thread 0
}

__global__ __launch_bounds__(TH_BLOCK) void k_mse_head_fwd(const float* __restrict__ pred, const float* __restrict__ target, uint32_t numel, uint32_t per_block,
                                                        const float* __restrict__ scale, float* __restrict__ loss, float* __restrict__ grad_unit,
                                                        uint32_t* ticket, float* partial) {
    __shared__ float lds[TH_BLOCK / 64];
    __shared__ uint32_t s_last;
    const uint32_t lo = blockIdx.x * per_block;
    const uint32_t hi = lo + per_block < numel ? lo + per_block : numel;
    const float unit = 2.0f / (float)numel;                                   // mse_loss_backward: 2 / numel * (input - target) * grad
    float acc = 0.0f;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += TH_BLOCK) {
        const float d = pred[i] - target[i];
        acc += d * d;
        grad_unit[i] = unit * d;
    }
    const float total = th_block_sum(acc, lds);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = total;
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u || threadIdx.x != 0) return;
    __threadfence();
    float s = 0.0f;
    for (uint32_t b = 0; b < gridDim.x; b++) s += partial[b];
    const float mean = s / (float)numel;
    loss[0] = mean;
    loss[1] = scale ? mean * scale[0] : mean;
    *ticket = 0u;
}

// grad_pred = grad_unit * (g_loss + g_scaled * scale): whichever of the two outputs the caller differentiated
__global__ __launch_bounds__(TH_BLOCK) void k_mse_head_bwd(const float* __restrict__ grad_unit, const float* __restrict__ g_loss, const float* __restrict__ g_scaled,
                                                        const float* __restrict__ scale, uint32_t numel, float* __restrict__ grad_pred) {
    const uint32_t i = blockIdx.x * TH_BLOCK + threadIdx.x;
    if (i >= numel) return;
    float g = g_loss ? g_loss[0] : 0.0f;
    if (g_scaled) g = g_loss ? g + g_scaled[0] * (scale ? scale[0] : 1.0f) : g_scaled[0] * (scale ? scale[0] : 1.0f);
    grad_pred[i] = grad_unit[i] * g;
}

// ---- the four launches above as ONE, for a caller that runs forward and backward back to back (ngp/train.py's direct step) ----------------------------------
// Element i = channel i % 3 of ray i / 3: mixed = image + (1 - weights_sum) * bg; d = mixed - target; the loss sums d * d with k_mse_head_fwd's partition and
// tree (the same bits); grad_image = (2 / numel * d) * scale -- k_mse_head_fwd's grad_unit times k_mse_head_bwd's factor for an incoming gradient of one on
// the scaled loss --; grad_weights_sum = -(g . bg) by the lane that holds a ray's first channel.  Every workgroup also clears its share of up to three
// buffers the backward launches that follow expect zeroed (the compositor's two gradient arrays, the header of the field backward's workspace).
struct th_zero { void* p[3]; uint64_t bytes[3]; };

__global__ __launch_bounds__(TH_BLOCK) void k_train_head_direct(const float* __restrict__ weights_sum, const float* __restrict__ image, const float* __restrict__ bg,
                                                             uint32_t bg_rows, float bg_value, const float* __restrict__ target, const float* __restrict__ scale,
                                                             uint32_t N, uint32_t loss_blocks, uint32_t per_block, float* __restrict__ out_image,
                                                             float* __restrict__ loss, float* __restrict__ grad_image, float* __restrict__ grad_weights_sum,
                                                             uint32_t* ticket, float* partial, th_zero Z) {
    __shared__ float lds[TH_BLOCK / 64];
    __shared__ uint32_t s_last;
    #pragma unroll
    for (int z = 0; z < 3; z++) {
        uint4* q = reinterpret_cast<uint4*>(Z.p[z]);
        const uint64_t n16 = Z.bytes[z] / 16;
        for (uint64_t i = (uint64_t)blockIdx.x * TH_BLOCK + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * TH_BLOCK) q[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (blockIdx.x >= loss_blocks) return;
    const uint32_t numel = 3u * N;
    const uint32_t lo = blockIdx.x * per_block;
    const uint32_t hi = lo + per_block < numel ? lo + per_block : numel;
    const float unit = 2.0f / (float)numel;
    const float sc = scale ? scale[0] : 1.0f;
    float acc = 0.0f;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += TH_BLOCK) {
        const uint32_t n = i / 3u, c = i - 3u * n;
        const float rest = 1.0f - weights_sum[n];
        const float mixed = image[i] + rest * th_bg(bg, bg_rows, bg_value, n, c);
        out_image[i] = mixed;
        const float d = mixed - target[i];
        acc += d * d;
        const float g = (unit * d) * sc;
        grad_image[i] = g;
        if (c == 0) {                                                       // this lane also forms the ray's grad_weights_sum: the other two channels again
            const float g1 = (unit * ((image[i + 1] + rest * th_bg(bg, bg_rows, bg_value, n, 1)) - target[i + 1])) * sc;
            const float g2 = (unit * ((image[i + 2] + rest * th_bg(bg, bg_rows, bg_value, n, 2)) - target[i + 2])) * sc;
            float s = g * th_bg(bg, bg_rows, bg_value, n, 0);
            s = s + g1 * th_bg(bg, bg_rows, bg_value, n, 1);
            s = s + g2 * th_bg(bg, bg_rows, bg_value, n, 2);
            grad_weights_sum[n] = -s;
        }
    }
    const float total = th_block_sum(acc, lds);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = total;
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == loss_blocks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u || threadIdx.x != 0) return;
    __threadfence();
    float s = 0.0f;
    for (uint32_t b = 0; b < loss_blocks; b++) s += partial[b];
    const float mean = s / (float)numel;
    loss[0] = mean;
    loss[1] = scale ? mean * scale[0] : mean;
    *ticket = 0u;
}

extern "C" int ngp_train_head_direct(const float* weights_sum, const float* image, const float* bg, uint32_t bg_rows, float bg_value, const float* target,
                                     const float* scale, uint32_t N, float* out_image, float* loss, float* grad_image, float* grad_weights_sum,
                                     void* const* zero_ptrs, const uint64_t* zero_bytes, uint32_t zero_count, void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(N >= 1 && N <= 0x55555555u / 1u, "train_head_direct: 1 .. 2^32 / 3 rays");
    NGP_REQUIRE(weights_sum && image && target && out_image && loss && grad_image && grad_weights_sum && workspace, "train_head_direct: null pointer");
    NGP_REQUIRE(bg_rows == 0 || ((bg_rows == 1 || bg_rows == N) && bg), "train_head_direct: bg_rows must be 0 (bg_value), 1 (bg[3]) or N (bg[N,3])");
    NGP_REQUIRE(zero_count <= 3 && (zero_count == 0 || (zero_ptrs && zero_bytes)), "train_head_direct: at most three buffers to clear");
    if (workspace_bytes < (size_t)(1 + TH_MAX_BLOCKS) * 4) return ngp_fail(NGP_EWORKSPACE, "train_head_direct: workspace of %zu bytes, %zu needed", workspace_bytes, (size_t)(1 + TH_MAX_BLOCKS) * 4);
    th_zero Z{{nullptr, nullptr, nullptr}, {0, 0, 0}};
    uint64_t most = 0;
    for (uint32_t z = 0; z < zero_count; z++) {
        NGP_REQUIRE(zero_bytes[z] == 0 || (zero_ptrs[z] && ((uintptr_t)zero_ptrs[z] % 16) == 0 && zero_bytes[z] % 16 == 0), "train_head_direct: buffers to clear must be 16-byte aligned multiples of 16 bytes");
        Z.p[z] = zero_ptrs[z]; Z.bytes[z] = zero_bytes[z];
        if (zero_bytes[z] > most) most = zero_bytes[z];
    }
    const uint32_t numel = 3u * N;
    uint32_t loss_blocks = ngp_div_up(numel, TH_SLICE);                    // k_mse_head_fwd's partition: the same sums in the same order
    if (loss_blocks > TH_MAX_BLOCKS) loss_blocks = TH_MAX_BLOCKS;
    const uint32_t per_block = ngp_div_up(numel, loss_blocks);
    uint32_t blocks = ngp_div_up(most, (uint64_t)TH_BLOCK * 16 * 8);       // ~8 16-byte stores per thread of the clearing loop
    if (blocks > 2048u) blocks = 2048u;
    if (blocks < loss_blocks) blocks = loss_blocks;
    uint32_t* ticket = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_train_head_direct, dim3(blocks), dim3(TH_BLOCK), 0, (hipStream_t)stream, weights_sum, image, bg, bg_rows, bg_value, target, scale, N,
                       loss_blocks, per_block, out_image, loss, grad_image, grad_weights_sum, ticket, (float*)(ticket + 1), Z);
    NGP_CHECK_LAUNCH("train_head_direct");
    return NGP_OK;
}

extern "C" int ngp_train_mix_forward(const float* weights_sum, const float* depth, const float* image, const float* nears, const float* fars,
                                     const float* bg, uint32_t bg_rows, float bg_value, uint32_t N, float* out_image, float* out_depth, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(weights_sum && image && out_image, "train_mix_forward: null pointer");
    NGP_REQUIRE(!out_depth || (depth && nears && fars), "train_mix_forward: the normalised depth needs depth, nears and fars");
    NGP_REQUIRE(bg_rows == 0 || ((bg_rows == 1 || bg_rows == N) && bg), "train_mix_forward: bg_rows must be 0 (bg_value), 1 (bg[3]) or N (bg[N,3])");
    hipLaunchKernelGGL(k_train_mix_fwd, dim3(ngp_div_up(N, TH_BLOCK)), dim3(TH_BLOCK), 0, (hipStream_t)stream, weights_sum, depth, image, nears, fars, bg, bg_rows,
                       bg_value, N, out_image, out_depth);
    NGP_CHECK_LAUNCH("train_mix_forward");
    return NGP_OK;
}

extern "C" int ngp_train_mix_backward(const float* grad_out_image, const float* bg, uint32_t bg_rows, float bg_value, uint32_t N, float* grad_weights_sum,
                                      void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grad_out_image && grad_weights_sum, "train_mix_backward: null pointer");
    NGP_REQUIRE(bg_rows == 0 || ((bg_rows == 1 || bg_rows == N) && bg), "train_mix_backward: bg_rows must be 0 (bg_value), 1 (bg[3]) or N (bg[N,3])");
    hipLaunchKernelGGL(k_train_mix_bwd, dim3(ngp_div_up(N, TH_BLOCK)), dim3(TH_BLOCK), 0, (hipStream_t)stream, grad_out_image, bg, bg_rows, bg_value, N,
                       grad_weights_sum);
    NGP_CHECK_LAUNCH("train_mix_backward");
    return NGP_OK;
}

extern "C" size_t ngp_mse_head_workspace(void) { return (size_t)(1 + TH_MAX_BLOCKS) * 4; }

extern "C" int ngp_mse_head_forward(const float* pred, const float* target, uint32_t numel, const float* scale, float* loss, float* grad_unit,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(numel >= 1, "mse_head_forward: empty input (torch's mean of nothing is NaN; callers decide)");
    NGP_REQUIRE(pred && target && loss && grad_unit && workspace, "mse_head_forward: null pointer");
    if (workspace_bytes < ngp_mse_head_workspace()) return ngp_fail(NGP_EWORKSPACE, "mse_head_forward: workspace of %zu bytes, %zu needed", workspace_bytes, ngp_mse_head_workspace());
    uint32_t blocks = ngp_div_up(numel, TH_SLICE);
    if (blocks > TH_MAX_BLOCKS) blocks = TH_MAX_BLOCKS;
    const uint32_t per_block = ngp_div_up(numel, blocks);
    uint32_t* ticket = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_mse_head_fwd, dim3(blocks), dim3(TH_BLOCK), 0, (hipStream_t)stream, pred, target, numel, per_block, scale, loss, grad_unit, ticket,
                       (float*)(ticket + 1));
    NGP_CHECK_LAUNCH("mse_head_forward");
    return NGP_OK;
}

extern "C" int ngp_mse_head_backward(const float* grad_unit, const float* grad_loss, const float* grad_scaled, const float* scale, uint32_t numel,
                                     float* grad_pred, void* stream) {
    if (numel == 0) return NGP_OK;
    NGP_REQUIRE(grad_unit && grad_pred && (grad_loss || grad_scaled), "mse_head_backward: null pointer");
    hipLaunchKernelGGL(k_mse_head_bwd, dim3(ngp_div_up(numel, TH_BLOCK)), dim3(TH_BLOCK), 0, (hipStream_t)stream, grad_unit, grad_loss, grad_scaled, scale, numel,
                       grad_pred);
    NGP_CHECK_LAUNCH("mse_head_backward");
    return NGP_OK;
}
