// adam.hip -- the optimiser half of the training step: torch.optim.Adam + torch.cuda.amp.GradScaler as the reference drives them
// (main_nerf.py:126 `Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)`; nerf/utils.py:329, :789-791 `scaler.scale(loss).backward();
// scaler.step(optimizer); scaler.update()`), for all parameters of the field in three launches.
//
// What torch runs for those three lines on 12.7 M float32 parameters: `_amp_foreach_non_finite_check_and_unscale_` (reads and rewrites every gradient),
// Adam through multi_tensor_apply (65,536-element chunks: 194 workgroups for the table, on a 256-CU chip), `_amp_update_scale_`, and -- because the next
// forward runs under autocast -- a float32 -> float16 copy of the table and of both weight vectors: 0.20 + 0.02 ms of a 1.55 ms step, ~0.53 GB moved.
// Here:   k_adam_check    reads the gradients once (51 MB) and raises found_inf;
//         k_adam_prepare  one thread: the scalars of this step (1/scale, the bias corrections of step t in double, per-tensor step sizes) and the
//                         GradScaler's scale / growth-tracker recurrence (`_amp_update_scale_`, AmpKernels.cu of torch 2.x);
//         k_adam_update   p, m, v <- Adam(p, m, v, g / scale) and, when asked, the float16 copy of p the next forward reads: 28 + 2 B per parameter,
//                         3,090 workgroups for the table.  Skipped as a whole when found_inf is set, as GradScaler.step skips optimizer.step().
// Elementwise arithmetic = torch's `_multi_tensor_adam` / `_single_tensor_adam` (torch/optim/adam.py), one float32 operation per torch operation, with
// the fused multiply-adds nvcc forms inside lerp / addcmul / addcdiv written out (this library is built with -ffp-contract=off):
//     g   = grad * inv_scale                                  (_amp_foreach_non_finite_check_and_unscale_)
//     m   = fma(1 - beta1, g - m, m)                          (exp_avg.lerp_(grad, 1 - beta1))
//     v   = fma((1 - beta2) * g, g, v * beta2)                (exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2))
//     den = sqrt(v) / sqrt(1 - beta2^t) + eps                 (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//     p   = fma(-(lr / (1 - beta1^t)), m / den, p)            (param.addcdiv_(exp_avg, denom, value = -step_size))
// t, lr / (1 - beta1^t) and sqrt(1 - beta2^t) are formed in double and narrowed to float once, as torch's Python side does.
// Parity: torch is importable here, so tests/test_gpu_adam.py steps torch.optim.Adam + GradScaler beside this on the same gradients (incl. an
// overflow step and scale growth) -- values to 2 ulp, the skip / scale / step-count decisions exactly.
#include "ngp_device.h"

#define AD_THREADS 256
#define AD_UNROLL 4
#define AD_BLOCK_ELEMS (AD_THREADS * 4 * AD_UNROLL)

// device-side state words (a 32-word float32/int32 buffer owned by the caller; ngp_hip.h documents the public ones)
enum { AD_SCALE = NGP_ADAM_STATE_SCALE, AD_TRACKER = NGP_ADAM_STATE_GROWTH_TRACKER, AD_STEP = NGP_ADAM_STATE_STEP, AD_FOUND = 3, AD_LAST_FOUND = NGP_ADAM_STATE_FOUND_INF, AD_SKIP = 5, AD_INV_SCALE = 6, AD_BC2_SQRT = 7, AD_NEG_STEP = 8 };

struct ad_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    _Float16* h;
    uint32_t n;
    uint32_t block0;         // first workgroup of this tensor
    double lr;
};
struct ad_args {
    ad_tensor t[NGP_ADAM_MAX_TENSORS];
    uint32_t count;
};

__device__ __forceinline__ int ad_find(const ad_args& A, uint32_t block) {
    int i = 0;
    #pragma unroll
    for (int k = 1; k < NGP_ADAM_MAX_TENSORS; k++)
        if (k < (int)A.count && block >= A.t[k].block0) i = k;
    return i;
}

__device__ __forceinline__ uint32_t ad_nonfinite(float x) { return (__builtin_bit_cast(uint32_t, x) & 0x7f800000u) == 0x7f800000u ? 1u : 0u; }

// A workgroup owns AD_BLOCK_ELEMS consecutive elements of one tensor.  Full, 16-byte aligned blocks (all but the last of a tensor) take the straight-line
// vector path: every lane's AD_UNROLL loads per stream are in flight together; the rest goes element by element.
__global__ __launch_bounds__(AD_THREADS) void k_adam_check(ad_args A, float* __restrict__ state) {
    const int i = ad_find(A, blockIdx.x);
    const float* __restrict__ g = A.t[i].g;
    const uint32_t n = A.t[i].n;
    const uint32_t base = (blockIdx.x - A.t[i].block0) * AD_BLOCK_ELEMS;
    uint32_t bad = 0;                                  // integer ORs: no short circuit
    if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0 && base + AD_BLOCK_ELEMS <= n) {
        float4 q[AD_UNROLL];
        #pragma unroll
        for (int u = 0; u < AD_UNROLL; u++) q[u] = *reinterpret_cast<const float4*>(g + base + (u * AD_THREADS + threadIdx.x) * 4);
        #pragma unroll
        for (int u = 0; u < AD_UNROLL; u++) bad |= ad_nonfinite(q[u].x) | ad_nonfinite(q[u].y) | ad_nonfinite(q[u].z) | ad_nonfinite(q[u].w);
    } else {
        for (uint32_t k = base + threadIdx.x; k < base + AD_BLOCK_ELEMS && k < n; k += AD_THREADS) bad |= ad_nonfinite(g[k]);
    }
    if (__ballot(bad != 0u) != 0ull && (threadIdx.x & 63u) == 0) state[AD_FOUND] = 1.0f;      // every writer stores the same value
}

__global__ void k_adam_prepare(ad_args A, ngp_adam_hyper_t H, float* __restrict__ state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int* istate = reinterpret_cast<int*>(state);
    const bool scaled = H.scaler_enabled != 0;
    const bool found = scaled && state[AD_FOUND] != 0.0f;
    state[AD_LAST_FOUND] = found ? 1.0f : 0.0f;
    state[AD_FOUND] = 0.0f;
    istate[AD_SKIP] = found ? 1 : 0;
    // GradScaler.step: inv_scale = scale.double().reciprocal().float()
    state[AD_INV_SCALE] = scaled ? (float)(1.0 / (double)state[AD_SCALE]) : 1.0f;
    if (!found) {
        const int t = istate[AD_STEP] + 1;
        istate[AD_STEP] = t;
        const double bc1 = 1.0 - pow(H.beta1, (double)t);
        const double bc2 = 1.0 - pow(H.beta2, (double)t);
        state[AD_BC2_SQRT] = (float)sqrt(bc2);
        for (uint32_t i = 0; i < A.count; i++) state[AD_NEG_STEP + i] = (float)(-(A.t[i].lr / bc1));
    }
    if (scaled) {                                                      // _amp_update_scale_ (growth_factor and backoff_factor are doubles there)
        if (found) {
            state[AD_SCALE] = (float)((double)state[AD_SCALE] * H.backoff_factor);
            istate[AD_TRACKER] = 0;
        } else {
            const int ok = istate[AD_TRACKER] + 1;
            if (ok == H.growth_interval) {
                const float grown = (float)((double)state[AD_SCALE] * H.growth_factor);
                if (ad_nonfinite(grown) == 0u) state[AD_SCALE] = grown;
                istate[AD_TRACKER] = 0;
            } else {
                istate[AD_TRACKER] = ok;
            }
        }
    }
}

__device__ __forceinline__ void ad_one(float& p, float g, float& m, float& v, float inv, float w1, float beta2, float w2, float bc2s, float eps, float neg_step) {
    g = g * inv;
    m = __builtin_fmaf(w1, g - m, m);
    v = __builtin_fmaf(w2 * g, g, v * beta2);
    const float den = sqrtf(v) / bc2s + eps;
    p = __builtin_fmaf(neg_step, m / den, p);
}

__global__ __launch_bounds__(AD_THREADS) void k_adam_update(ad_args A, ngp_adam_hyper_t H, const float* __restrict__ state) {
    if (reinterpret_cast<const int*>(state)[AD_SKIP] != 0) return;
    const int i = ad_find(A, blockIdx.x);
    float* __restrict__ p = A.t[i].p;
    const float* __restrict__ g = A.t[i].g;
    float* __restrict__ m = A.t[i].m;
    float* __restrict__ v = A.t[i].v;
    _Float16* __restrict__ h = A.t[i].h;
    const uint32_t n = A.t[i].n;
    const uint32_t base = (blockIdx.x - A.t[i].block0) * AD_BLOCK_ELEMS;
    const float inv = state[AD_INV_SCALE], bc2s = state[AD_BC2_SQRT], neg_step = state[AD_NEG_STEP + i];
    // torch forms 1 - beta as Python doubles and narrows them at the kernel boundary: (float)(1.0 - 0.9) is 0.1f, 1.0f - 0.9f is not
    const float w1 = (float)(1.0 - H.beta1), w2 = (float)(1.0 - H.beta2), beta2 = (float)H.beta2, eps = (float)H.eps;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0 &&
                     (reinterpret_cast<uintptr_t>(h) & 7u) == 0;
    if (vec && base + AD_BLOCK_ELEMS <= n) {
        float4 P[AD_UNROLL], G[AD_UNROLL], M[AD_UNROLL], V[AD_UNROLL];
        #pragma unroll
        for (int u = 0; u < AD_UNROLL; u++) {
            const uint32_t e = base + (u * AD_THREADS + threadIdx.x) * 4;
            P[u] = *reinterpret_cast<const float4*>(p + e);
            G[u] = *reinterpret_cast<const float4*>(g + e);
            M[u] = *reinterpret_cast<const float4*>(m + e);
            V[u] = *reinterpret_cast<const float4*>(v + e);
        }
        #pragma unroll
        for (int u = 0; u < AD_UNROLL; u++) {
            const uint32_t e = base + (u * AD_THREADS + threadIdx.x) * 4;
            ad_one(P[u].x, G[u].x, M[u].x, V[u].x, inv, w1, beta2, w2, bc2s, eps, neg_step);
            ad_one(P[u].y, G[u].y, M[u].y, V[u].y, inv, w1, beta2, w2, bc2s, eps, neg_step);
            ad_one(P[u].z, G[u].z, M[u].z, V[u].z, inv, w1, beta2, w2, bc2s, eps, neg_step);
            ad_one(P[u].w, G[u].w, M[u].w, V[u].w, inv, w1, beta2, w2, bc2s, eps, neg_step);
            *reinterpret_cast<float4*>(p + e) = P[u];
            *reinterpret_cast<float4*>(m + e) = M[u];
            *reinterpret_cast<float4*>(v + e) = V[u];
            if (h) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 q = {(_Float16)P[u].x, (_Float16)P[u].y, (_Float16)P[u].z, (_Float16)P[u].w};
                *reinterpret_cast<h4*>(h + e) = q;
            }
        }
    } else {
        for (uint32_t k = base + threadIdx.x; k < base + AD_BLOCK_ELEMS && k < n; k += AD_THREADS) {
            float pk = p[k], mk = m[k], vk = v[k];
            ad_one(pk, g[k], mk, vk, inv, w1, beta2, w2, bc2s, eps, neg_step);
            p[k] = pk; m[k] = mk; v[k] = vk;
            if (h) h[k] = (_Float16)pk;
        }
    }
}

extern "C" int ngp_adam_step(const ngp_adam_tensor_t* tensors, uint32_t count, const ngp_adam_hyper_t* hyper, float* state, void* stream) {
    NGP_REQUIRE(tensors && hyper && state, "adam_step: null pointer");
    NGP_REQUIRE(count >= 1 && count <= NGP_ADAM_MAX_TENSORS, "adam_step: 1..%d tensors per call", NGP_ADAM_MAX_TENSORS);
    NGP_REQUIRE(hyper->beta1 >= 0.0 && hyper->beta1 < 1.0 && hyper->beta2 >= 0.0 && hyper->beta2 < 1.0 && hyper->eps >= 0.0, "adam_step: betas must lie in [0, 1), eps >= 0");
    NGP_REQUIRE(!hyper->scaler_enabled || (hyper->growth_interval >= 1 && hyper->growth_factor > 1.0 && hyper->backoff_factor > 0.0 && hyper->backoff_factor < 1.0),
                "adam_step: GradScaler needs growth_factor > 1, 0 < backoff_factor < 1, growth_interval >= 1");
    ad_args A;
    uint64_t blocks = 0;
    for (uint32_t i = 0; i < count; i++) {
        const ngp_adam_tensor_t& t = tensors[i];
        NGP_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, "adam_step: tensor %u: null pointer", i);
        NGP_REQUIRE(t.n >= 1 && t.n <= 0xffff0000ull, "adam_step: tensor %u: 1 .. 2^32 - 65536 elements", i);
        NGP_REQUIRE(((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) % 4 == 0 && (uintptr_t)t.half_copy % 2 == 0,
                    "adam_step: tensor %u: misaligned pointer", i);
        NGP_REQUIRE(t.lr >= 0.0, "adam_step: tensor %u: negative learning rate", i);
        A.t[i] = {(float*)t.param, (const float*)t.grad, (float*)t.exp_avg, (float*)t.exp_avg_sq, (_Float16*)t.half_copy, (uint32_t)t.n, (uint32_t)blocks, t.lr};
        blocks += ngp_div_up(t.n, AD_BLOCK_ELEMS);
    }
    for (uint32_t i = count; i < NGP_ADAM_MAX_TENSORS; i++) A.t[i] = {nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0xffffffffu, 0.0};
    A.count = count;
    NGP_REQUIRE(blocks <= 0x7fffffffull, "adam_step: too many elements for one launch");
    hipStream_t s = (hipStream_t)stream;
    if (hyper->scaler_enabled) hipLaunchKernelGGL(k_adam_check, dim3((uint32_t)blocks), dim3(AD_THREADS), 0, s, A, state);
    hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, s, A, *hyper, state);
    hipLaunchKernelGGL(k_adam_update, dim3((uint32_t)blocks), dim3(AD_THREADS), 0, s, A, *hyper, (const float*)state);
    NGP_CHECK_LAUNCH("adam_step");
    return NGP_OK;
}
