// gridencoder.hip -- gfx950 kernels behind the `_gridencoder` native surface of the reference
// (gridencoder/src/gridencoder.h:12-13): multi-resolution hash / tiled grid encoding, forward with optional
// dy_dx, scatter-add backward into the table, and the input gradient from dy_dx.
//
// Layouts are the reference's: inputs [B,D] f32, table [sO,C], outputs [L,B,C] (level-major), dy_dx [B,L,D,C].
// Roofline: HBM / Infinity-Cache gather bound -- 2^D * C * sizeof(T) table bytes per (sample, level), every one a
// scattered access; all 2^D corner loads of a lane are issued before the first is consumed.
//
// Arithmetic is the reference's, operation for operation: positions and weights in binary32, accumulation in the
// table dtype (c10::Half arithmetic = compute in float, round to half after every operation).
#include "ngp_device.h"
#include <atomic>

static constexpr uint32_t GE_MAX_LEVELS = 32;

struct ge_levels {
    float scale[GE_MAX_LEVELS];          // exp2f(level * S) * H - 1, evaluated on the host (gridencoder.cu:126)
    uint32_t resolution[GE_MAX_LEVELS];  // ceil(scale) + 1                                 (gridencoder.cu:127)
};

static void ge_fill_levels(ge_levels& lv, uint32_t L, float S, uint32_t H) {
    for (uint32_t l = 0; l < L; l++) {
        lv.scale[l] = exp2f((float)l * S) * (float)H - 1.0f;
        lv.resolution[l] = (uint32_t)ceilf(lv.scale[l]) + 1u;
    }
}

// scalar_t arithmetic of the reference: float for float, round-to-half after every op for half
template <typename T> struct ge_num;
template <> struct ge_num<float> {
    static __device__ __forceinline__ float rnd(float v) { return v; }
    static __device__ __forceinline__ float mul_rnd(float a, float b) { return a * b; }
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ge_num<_Float16> {
    static __device__ __forceinline__ float rnd(float v) { return (float)ngp_f2h(v); }   // two roundings, see ngp_f2h
    static __device__ __forceinline__ float mul_rnd(float a, float b) { return (float)ngp_f2h(a * b); }
    static __device__ __forceinline__ float load(const _Float16* p) { return (float)*p; }
    static __device__ __forceinline__ void store(_Float16* p, float v) { *p = ngp_f2h(v); }
};

template <typename T, uint32_t C> struct alignas(sizeof(T) * C) ge_vec { T v[C]; };

static __device__ __constant__ const uint32_t GE_PRIMES[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};

template <uint32_t D>
__device__ __forceinline__ uint32_t ge_index(uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                             uint32_t resolution, const uint32_t (&pg)[D]) {
    // reference get_grid_index / fast_hash: gridencoder.cu:35-72 (returns the row, not row*C)
    uint32_t stride = 1, index = 0;
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pg[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t h = 0;
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) h ^= pg[d] * GE_PRIMES[d];
        index = h;
    }
    return index % hashmap_size;
}

template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_forward(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                      const int* __restrict__ offsets, T* __restrict__ outputs,
                                                      uint32_t B, uint32_t L, ge_levels lv, bool calc_grad_inputs,
                                                      T* __restrict__ dy_dx, uint32_t gridtype, bool align_corners) {
    using num = ge_num<T>;
    using vec = ge_vec<T, C>;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;

    const vec* tab = reinterpret_cast<const vec*>(grid) + (uint32_t)offsets[level];
    const float* in = inputs + (uint64_t)b * D;
    vec* out = reinterpret_cast<vec*>(outputs) + ((uint64_t)level * B + b);
    T* dydx = dy_dx + ((uint64_t)b * L + level) * D * C;

    float xin[D];
    bool oob = false;
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) { xin[d] = in[d]; oob |= (xin[d] < 0 || xin[d] > 1); }
    if (oob) {                                        // gridencoder.cu:99-123
        vec z;
        #pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) z.v[ch] = (T)0.0f;
        *out = z;
        if (calc_grad_inputs) {
            #pragma unroll
            for (uint32_t i = 0; i < D * C; i++) dydx[i] = (T)0.0f;
        }
        return;
    }

    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];

    float pos[D];
    uint32_t pg[D];
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = xin[d] * scale + (align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }

    // issue all 2^D corner gathers, then blend in the reference's corner order
    vec corner[1u << D];
    float w[1u << D];
    #pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float wi = 1;
        uint32_t pl[D];
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { wi *= 1 - pos[d]; pl[d] = pg[d]; }
            else { wi *= pos[d]; pl[d] = pg[d] + 1; }
        }
        w[idx] = wi;
        corner[idx] = tab[ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pl)];
    }
    float res[C];
    #pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) res[ch] = 0.0f;
    #pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        #pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) {
            // scalar_t += float: the product narrows to scalar_t, then the sum rounds to scalar_t (gridencoder.cu:166)
            const float prod = num::rnd(w[idx] * (float)corner[idx].v[ch]);
            res[ch] = num::rnd(res[ch] + prod);
        }
    }
    vec o;
    #pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) o.v[ch] = (T)res[ch];
    *out = o;

    if (calc_grad_inputs) {                           // gridencoder.cu:180-223
        #pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
            #pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0.0f;
            #pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float wi = scale;
                uint32_t pl[D];
                #pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { wi *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { wi *= pos[d]; pl[d] = pg[d] + 1; }
                }
                pl[gd] = pg[gd];
                const vec left = tab[ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pl)];
                pl[gd] = pg[gd] + 1;
                const vec right = tab[ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pl)];
                #pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float diff = num::rnd((float)right.v[ch] - (float)left.v[ch]);
                    const float prod = num::rnd(wi * diff);
                    rg[ch] = num::rnd(rg[ch] + prod);
                }
            }
            #pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) dydx[gd * C + ch] = (T)rg[ch];
        }
    }
}

// scatter: one lane per (sample, pair of features, level)   (reference: gridencoder.cu:227-314)
//
// Wave-aggregated when a lane carries all C features of its sample (C <= 2, the reference's hash grid): consecutive
// samples of a ray sit in the same cell of a coarse level for dozens of steps, so the 64 lanes of a wave are a few runs of
// lanes with the same cell and therefore the same 8 rows.  Each run is summed across its lanes in binary32 (segmented
// shuffle scan: 6 steps) and its last lane issues ONE atomic per corner.  On the training workload that is 3x fewer
// atomics overall and ~60x fewer on the coarsest levels, whose few thousand rows otherwise serialise in L2 (32 ms -> see
// DESIGN.md).  Sums are at least as accurate as the reference's (it rounds every product to half and adds in half, in
// arbitrary order, gridencoder.cu:302-308); the parity tests' tolerance is unchanged.
template <typename T, uint32_t D, uint32_t C, uint32_t N_C>
__global__ __launch_bounds__(256) void k_grid_backward(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                       const int* __restrict__ offsets, T* __restrict__ grad_grid,
                                                       uint32_t B, uint32_t L, ge_levels lv, uint32_t gridtype, bool align_corners) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b = (uint32_t)(((uint64_t)tid * N_C) / C);
    constexpr bool AGG = (C == N_C);                   // one lane = one sample: runs of lanes are runs of samples
    if (!AGG && b >= B) return;
    bool valid = b < B;
    const uint32_t level = blockIdx.y;
    const uint32_t ch = valid ? tid * N_C - b * C : 0u;

    T* gg = grad_grid + (uint64_t)(uint32_t)offsets[level] * C;
    const float* in = inputs + (uint64_t)(valid ? b : 0u) * D;
    const T* g = grad + ((uint64_t)level * B + (valid ? b : 0u)) * C + ch;

    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];

    float pos[D];
    uint32_t pg[D];
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float x = in[d];
        if (x < 0 || x > 1) {                          // out of range: contributes nothing
            if (!AGG) return;
            valid = false;
        }
        pos[d] = x * scale + (align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    float gc[N_C];
    #pragma unroll
    for (uint32_t c = 0; c < N_C; c++) gc[c] = valid ? (float)g[c] : 0.0f;

    // runs of consecutive lanes in the same cell
    const int lane = (int)(threadIdx.x & 63u);
    int start = lane;
    bool tail = true;
    if constexpr (AGG) {
        bool same = valid && lane > 0 && __shfl_up((int)valid, 1, 64) != 0;
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) same = (__shfl_up(pg[d], 1, 64) == pg[d]) && same;
        const unsigned long long heads = __ballot(!same);                      // bit l: lane l starts a run
        start = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));     // first lane of this lane's run
        tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    }

    uint32_t pair_row[2] = {0u, 0u}, pair_val[2] = {0u, 0u};
    int pair_act[2] = {0, 0};
    #pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        bool send = true;
        float wi = 1;
        uint32_t pl[D];
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { wi *= 1 - pos[d]; pl[d] = pg[d]; }
            else { wi *= pos[d]; pl[d] = pg[d] + 1; }
        }
        float v[N_C];
        #pragma unroll
        for (uint32_t c = 0; c < N_C; c++) v[c] = wi * gc[c];
        if constexpr (AGG) {
            #pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                float o[N_C];
                #pragma unroll
                for (uint32_t c = 0; c < N_C; c++) o[c] = __shfl_up(v[c], off, 64);
                if (lane - off >= start) {
                    #pragma unroll
                    for (uint32_t c = 0; c < N_C; c++) v[c] += o[c];
                }
            }
            bool zero = true;                          // padding rows and samples behind a saturated ray carry exact zeros
            #pragma unroll
            for (uint32_t c = 0; c < N_C; c++) zero = zero && v[c] == 0.0f;
            send = tail && valid && !zero;
            if (sizeof(T) != 2 && !send) continue;     // (the half path exchanges with the neighbouring lane first)
        }
        const uint64_t row = (uint64_t)ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pl) * C + ch;
        if constexpr (sizeof(T) == 2 && AGG) {
            // Float atomics execute at the memory side as 64-byte requests (about 20 G requests/s chip-wide, which is what
            // this kernel runs at): lanes of ONE instruction that hit the same 64-byte line share a request.  The two
            // x-neighbours of a sample are rows idx and idx ^ (x ^ (x+1)): the same line 15 times out of 16, but as two
            // instructions they are two requests.  So neighbouring lanes swap work: one instruction carries both x-corners
            // of the even lane's sample (even lane: x, odd lane: x+1), the next both of the odd lane's.
            static_assert(N_C == 2, "half scatter needs feature pairs");
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 hv;
            hv.x = ngp_f2h(v[0]);                      // (__half)(w * grad) : gridencoder.cu:302
            hv.y = ngp_f2h(v[1]);
            pair_row[idx & 1u] = (uint32_t)row;
            pair_val[idx & 1u] = __builtin_bit_cast(uint32_t, hv);
            pair_act[idx & 1u] = send ? 1 : 0;
            if (idx & 1u) {
                const bool odd = (lane & 1) != 0;
                // quad_perm [0,0,2,2] = 0xA0: odd lanes read their even neighbour;  [1,1,3,3] = 0xF5: even lanes read their odd neighbour
                // (all six exchanges are evaluated by every lane BEFORE any selection: inside a conditional expression only the
                //  lanes taking that side would be active, and a DPP read of a switched-off lane returns 0)
                const uint32_t row1_even = (uint32_t)__builtin_amdgcn_mov_dpp((int)pair_row[1], 0xA0, 0xf, 0xf, true);
                const uint32_t val1_even = (uint32_t)__builtin_amdgcn_mov_dpp((int)pair_val[1], 0xA0, 0xf, 0xf, true);
                const int act1_even = __builtin_amdgcn_mov_dpp(pair_act[1], 0xA0, 0xf, 0xf, true);
                const uint32_t row0_odd = (uint32_t)__builtin_amdgcn_mov_dpp((int)pair_row[0], 0xF5, 0xf, 0xf, true);
                const uint32_t val0_odd = (uint32_t)__builtin_amdgcn_mov_dpp((int)pair_val[0], 0xF5, 0xf, 0xf, true);
                const int act0_odd = __builtin_amdgcn_mov_dpp(pair_act[0], 0xF5, 0xf, 0xf, true);
                const uint32_t rowA = odd ? row1_even : pair_row[0], valA = odd ? val1_even : pair_val[0];
                const uint32_t rowB = odd ? pair_row[1] : row0_odd, valB = odd ? pair_val[1] : val0_odd;
                const int actA = odd ? act1_even : pair_act[0], actB = odd ? pair_act[1] : act0_odd;
                if (actA) __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(gg + rowA), __builtin_bit_cast(h2, valA));
                if (actB) __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(gg + rowB), __builtin_bit_cast(h2, valB));
            }
        } else if constexpr (sizeof(T) == 2) {
            static_assert(sizeof(T) != 2 || N_C == 2, "half scatter needs feature pairs");
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 hv;
            hv.x = ngp_f2h(v[0]);                      // (__half)(w * grad) : gridencoder.cu:302
            hv.y = ngp_f2h(v[1]);
            __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(gg + row), hv);
        } else {
            #pragma unroll
            for (uint32_t c = 0; c < N_C; c++) unsafeAtomicAdd((float*)gg + row + c, v[c]);
        }
    }
}

// grad_inputs[b,d] = sum_{l,ch} grad[l,b,ch] * dy_dx[b,l,d,ch]   (reference: gridencoder.cu:317-343)
template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_input_backward(const T* __restrict__ grad, const T* __restrict__ dy_dx,
                                                             T* __restrict__ grad_inputs, uint32_t B, uint32_t L) {
    using num = ge_num<T>;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* jac = dy_dx + (uint64_t)b * L * D * C;
    float result = 0;
    for (uint32_t l = 0; l < L; l++) {
        #pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) {
            const float prod = num::rnd((float)grad[((uint64_t)l * B + b) * C + ch] * (float)jac[(l * D + d) * C + ch]);
            result = num::rnd(result + prod);
        }
    }
    grad_inputs[t] = (T)result;
}

// The same input gradient without the [B, L*D*C] Jacobian in memory: one lane per point walks the levels, gathers the
// 2^D corners once per level, forms dy_dx for the D directions exactly as k_grid_forward does (gridencoder.cu:180-223)
// and adds grad * dy_dx in k_grid_input_backward's order (l outer, channel inner, rounded to T at every step), so the
// result is bit-identical to the two-kernel route.  Saves writing and re-reading 96 values per point and the 24
// second-time corner gathers of the forward's dy_dx branch.
template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_input_backward_recompute(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                                       const T* __restrict__ grid, const int* __restrict__ offsets,
                                                                       T* __restrict__ grad_inputs, uint32_t B, uint32_t L, ge_levels lv,
                                                                       uint32_t gridtype, bool align_corners) {
    using num = ge_num<T>;
    using vec = ge_vec<T, C>;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* in = inputs + (uint64_t)b * D;
    float xin[D];
    bool oob = false;
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) { xin[d] = in[d]; oob |= (xin[d] < 0 || xin[d] > 1); }
    float result[D];
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) result[d] = 0.0f;
    if (!oob) {                                       // out of range: dy_dx is zero (gridencoder.cu:99-123)
        for (uint32_t level = 0; level < L; level++) {
            const vec* tab = reinterpret_cast<const vec*>(grid) + (uint32_t)offsets[level];
            const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
            const float scale = lv.scale[level];
            const uint32_t resolution = lv.resolution[level];
            float pos[D];
            uint32_t pg[D];
            #pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = xin[d] * scale + (align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            vec corner[1u << D];
            #pragma unroll
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                uint32_t pl[D];
                #pragma unroll
                for (uint32_t d = 0; d < D; d++) pl[d] = pg[d] + ((idx >> d) & 1u);
                corner[idx] = tab[ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pl)];
            }
            vec g = reinterpret_cast<const vec*>(grad)[(uint64_t)level * B + b];
            #pragma unroll
            for (uint32_t gd = 0; gd < D; gd++) {
                float rg[C];
                #pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0.0f;
                #pragma unroll
                for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                    float wi = scale;
                    uint32_t cl = 0;                  // corner number of the "left" neighbour: bit d set = upper cell in d
                    #pragma unroll
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                        if ((idx & (1u << nd)) == 0) { wi *= 1 - pos[d]; }
                        else { wi *= pos[d]; cl |= 1u << d; }
                    }
                    const vec left = corner[cl], right = corner[cl | (1u << gd)];
                    #pragma unroll
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float diff = num::rnd((float)right.v[ch] - (float)left.v[ch]);
                        const float prod = num::mul_rnd(wi, diff);
                        rg[ch] = num::rnd(rg[ch] + prod);
                    }
                }
                #pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float prod = num::rnd((float)g.v[ch] * rg[ch]);
                    result[gd] = num::rnd(result[gd] + prod);
                }
            }
        }
    }
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) grad_inputs[(uint64_t)b * D + d] = (T)result[d];
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------

template <typename T, uint32_t D, uint32_t C>
static void ge_launch_forward(const float* inputs, const void* emb, const int* offsets, void* outputs, uint32_t B, uint32_t L,
                              const ge_levels& lv, bool calc, void* dy_dx, uint32_t gridtype, bool ac, hipStream_t s) {
    hipLaunchKernelGGL((k_grid_forward<T, D, C>), dim3(ngp_div_up(B, 256), L), dim3(256), 0, s,
                       inputs, (const T*)emb, offsets, (T*)outputs, B, L, lv, calc, (T*)dy_dx, gridtype, ac);
}

template <typename T, uint32_t D>
static int ge_forward_c(uint32_t C, const float* inputs, const void* emb, const int* offsets, void* outputs, uint32_t B, uint32_t L,
                        const ge_levels& lv, bool calc, void* dy_dx, uint32_t gridtype, bool ac, hipStream_t s) {
    switch (C) {
        case 1: ge_launch_forward<T, D, 1>(inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s); return NGP_OK;
        case 2: ge_launch_forward<T, D, 2>(inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s); return NGP_OK;
        case 4: ge_launch_forward<T, D, 4>(inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s); return NGP_OK;
        case 8: ge_launch_forward<T, D, 8>(inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s); return NGP_OK;
        default: return ngp_fail(NGP_EINVAL, "GridEncoding: C must be 1, 2, 4, or 8.");
    }
}

template <typename T>
static int ge_forward_d(uint32_t D, uint32_t C, const float* inputs, const void* emb, const int* offsets, void* outputs, uint32_t B,
                        uint32_t L, const ge_levels& lv, bool calc, void* dy_dx, uint32_t gridtype, bool ac, hipStream_t s) {
    switch (D) {
        case 2: return ge_forward_c<T, 2>(C, inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s);
        case 3: return ge_forward_c<T, 3>(C, inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s);
        case 4: return ge_forward_c<T, 4>(C, inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s);
        case 5: return ge_forward_c<T, 5>(C, inputs, emb, offsets, outputs, B, L, lv, calc, dy_dx, gridtype, ac, s);
        default: return ngp_fail(NGP_EINVAL, "GridEncoding: D must be 2, 3, 4, or 5.");
    }
}

// ---------------------------------------------------------------------------
// forward straight into the layout the caller wants, [B, L*C]
//
// The reference's kernel writes [L, B, C] (a level per blockIdx.y) and its wrapper then permutes and copies to [B, L*C] (grid.py:42,52): 41 MB read
// and written again per call at 640 k points, 64 times per rendered frame.  Here a workgroup owns 256 samples, walks the L levels itself, parks the
// L*C values of every sample in LDS (rows padded to an odd number of words: conflict-free column writes) and writes whole rows, coalesced.
// Arithmetic per (sample, level): k_grid_forward's, operation for operation, so the values are the same bits.  D = 3, C = 2 (the reference's grid).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_grid_forward_rows(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                           const int* __restrict__ offsets, T* __restrict__ outputs,
                                                           uint32_t B, uint32_t L, ge_levels lv, uint32_t gridtype, bool align_corners) {
    constexpr uint32_t D = 3, C = 2;
    using num = ge_num<T>;
    using vec = ge_vec<T, C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char ge_smem[];
    vec* tile = reinterpret_cast<vec*>(ge_smem);                       // [256][L + 1] feature pairs
    const uint32_t stride = L + 1;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const bool live = b < B;
    float xin[D] = {0.0f, 0.0f, 0.0f};
    bool oob = !live;
    if (live) {
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) { xin[d] = inputs[(uint64_t)b * D + d]; oob |= (xin[d] < 0 || xin[d] > 1); }
    }
    for (uint32_t level = 0; level < L; level++) {
        vec o;
        o.v[0] = (T)0.0f; o.v[1] = (T)0.0f;
        if (!oob) {
            const vec* tab = reinterpret_cast<const vec*>(grid) + (uint32_t)offsets[level];
            const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
            const float scale = lv.scale[level];
            const uint32_t resolution = lv.resolution[level];
            float pos[D];
            uint32_t pg[D];
            #pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = xin[d] * scale + (align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            vec corner[8];
            float w[8];
            #pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                float wi = 1;
                uint32_t pl[D];
                #pragma unroll
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { wi *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { wi *= pos[d]; pl[d] = pg[d] + 1; }
                }
                w[idx] = wi;
                corner[idx] = tab[ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pl)];
            }
            float res[C] = {0.0f, 0.0f};
            #pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                #pragma unroll
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float prod = num::rnd(w[idx] * (float)corner[idx].v[ch]);
                    res[ch] = num::rnd(res[ch] + prod);
                }
            }
            o.v[0] = (T)res[0]; o.v[1] = (T)res[1];
        }
        tile[threadIdx.x * stride + level] = o;
    }
    __syncthreads();
    // rows out: consecutive lanes write consecutive feature pairs of a row, 256 / L rows per pass
    const uint32_t rows_here = (B - blockIdx.x * 256u) < 256u ? (B - blockIdx.x * 256u) : 256u;
    vec* out = reinterpret_cast<vec*>(outputs) + (uint64_t)blockIdx.x * 256 * L;
    for (uint32_t i = threadIdx.x; i < rows_here * L; i += 256) {
        const uint32_t r = i / L, c = i - r * L;
        out[i] = tile[r * stride + c];
    }
}

extern "C" int ngp_grid_encode_forward_rows(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                            uint32_t gridtype, int align_corners, int dtype, void* stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward_rows: null pointer");
    NGP_REQUIRE(D == 3 && C == 2, "grid_encode_forward_rows: D must be 3 and C 2 (use grid_encode_forward + a permute otherwise)");
    NGP_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, "grid_encode_forward_rows: L must be in 1..32");
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, "grid_encode_forward_rows: dtype must be f32 or f16");
    ge_levels lv;
    ge_fill_levels(lv, L, S, H);
    const dim3 grid_dim(ngp_div_up(B, 256)), block(256);
    const size_t lds = 256 * (size_t)(L + 1) * 2 * (dtype == NGP_F32 ? 4 : 2);
    // 64 KiB is what a kernel gets without hipFuncSetAttribute(MaxDynamicSharedMemorySize): f32 tables up to 31 levels, f16 up to 32.
    // gridencoder/grid.py sends larger ones to ngp_grid_encode_forward + permute (ngp_grid_encode_forward_rows_fits below is its test)
    NGP_REQUIRE(lds <= 65536, "grid_encode_forward_rows: the row tile does not fit 64 KiB of LDS (use grid_encode_forward + a permute)");
    if (dtype == NGP_F32)
        hipLaunchKernelGGL(k_grid_forward_rows<float>, grid_dim, block, lds, (hipStream_t)stream, inputs, (const float*)embeddings, offsets, (float*)outputs,
                           B, L, lv, gridtype, align_corners != 0);
    else
        hipLaunchKernelGGL(k_grid_forward_rows<_Float16>, grid_dim, block, lds, (hipStream_t)stream, inputs, (const _Float16*)embeddings, offsets,
                           (_Float16*)outputs, B, L, lv, gridtype, align_corners != 0);
    NGP_CHECK_LAUNCH("grid_encode_forward_rows");
    return NGP_OK;
}

extern "C" int ngp_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                       int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners,
                                       int dtype, void* stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: null pointer");
    NGP_REQUIRE(!calc_grad_inputs || dy_dx, "grid_encode_forward: calc_grad_inputs needs dy_dx");
    NGP_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, "grid_encode_forward: L must be in 1..32");
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, "grid_encode_forward: dtype must be f32 or f16");
    NGP_REQUIRE((uint64_t)B * L * D * C < (1ull << 40), "grid_encode_forward: size overflow");
    if (B == 0) return NGP_OK;
    ge_levels lv;
    ge_fill_levels(lv, L, S, H);
    int rc = dtype == NGP_F32
        ? ge_forward_d<float>(D, C, inputs, embeddings, offsets, outputs, B, L, lv, calc_grad_inputs != 0, dy_dx, gridtype, align_corners != 0, (hipStream_t)stream)
        : ge_forward_d<_Float16>(D, C, inputs, embeddings, offsets, outputs, B, L, lv, calc_grad_inputs != 0, dy_dx, gridtype, align_corners != 0, (hipStream_t)stream);
    if (rc != NGP_OK) return rc;
    NGP_CHECK_LAUNCH("grid_encode_forward");
    return NGP_OK;
}

template <typename T, uint32_t D, uint32_t C>
static void ge_launch_backward(const void* grad, const float* inputs, const int* offsets, void* gg, uint32_t B, uint32_t L,
                               const ge_levels& lv, bool calc, const void* dy_dx, void* gi, uint32_t gridtype, bool ac, hipStream_t s) {
    constexpr uint32_t N_C = C < 2 ? C : 2;           // features per lane (gridencoder.cu:380)
    const uint32_t nthreads = (uint32_t)(((uint64_t)B * C) / N_C);
    if (gg)                                            // null: the caller does not need the table gradient (frozen model)
        hipLaunchKernelGGL((k_grid_backward<T, D, C, N_C>), dim3(ngp_div_up(nthreads, 256), L), dim3(256), 0, s,
                           (const T*)grad, inputs, offsets, (T*)gg, B, L, lv, gridtype, ac);
    if (calc)
        hipLaunchKernelGGL((k_grid_input_backward<T, D, C>), dim3(ngp_div_up((uint64_t)B * D, 256)), dim3(256), 0, s,
                           (const T*)grad, (const T*)dy_dx, (T*)gi, B, L);
}

template <typename T, uint32_t D>
static int ge_backward_c(uint32_t C, const void* grad, const float* inputs, const int* offsets, void* gg, uint32_t B, uint32_t L,
                         const ge_levels& lv, bool calc, const void* dy_dx, void* gi, uint32_t gridtype, bool ac, hipStream_t s) {
    switch (C) {
        case 1:
            if constexpr (sizeof(T) == 2) return ngp_fail(NGP_EINVAL, "grid_encode_backward: f16 needs an even C (the reference forces f32 for odd C, grid.py:36-39)");
            else { ge_launch_backward<T, D, 1>(grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s); return NGP_OK; }
        case 2: ge_launch_backward<T, D, 2>(grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s); return NGP_OK;
        case 4: ge_launch_backward<T, D, 4>(grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s); return NGP_OK;
        case 8: ge_launch_backward<T, D, 8>(grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s); return NGP_OK;
        default: return ngp_fail(NGP_EINVAL, "GridEncoding: C must be 1, 2, 4, or 8.");
    }
}

template <typename T>
static int ge_backward_d(uint32_t D, uint32_t C, const void* grad, const float* inputs, const int* offsets, void* gg, uint32_t B, uint32_t L,
                         const ge_levels& lv, bool calc, const void* dy_dx, void* gi, uint32_t gridtype, bool ac, hipStream_t s) {
    switch (D) {
        case 2: return ge_backward_c<T, 2>(C, grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s);
        case 3: return ge_backward_c<T, 3>(C, grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s);
        case 4: return ge_backward_c<T, 4>(C, grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s);
        case 5: return ge_backward_c<T, 5>(C, grad, inputs, offsets, gg, B, L, lv, calc, dy_dx, gi, gridtype, ac, s);
        default: return ngp_fail(NGP_EINVAL, "GridEncoding: D must be 2, 3, 4, or 5.");
    }
}

extern "C" int ngp_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                        void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                                        int align_corners, int dtype, void* stream) {
    (void)embeddings;                                  // kept for signature parity; the scatter never reads the table
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && inputs && offsets, "grid_encode_backward: null pointer");
    NGP_REQUIRE(grad_embeddings || calc_grad_inputs, "grid_encode_backward: nothing to compute (no table gradient, no input gradient)");
    NGP_REQUIRE(!calc_grad_inputs || (dy_dx && grad_inputs), "grid_encode_backward: calc_grad_inputs needs dy_dx and grad_inputs");
    NGP_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, "grid_encode_backward: L must be in 1..32");
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, "grid_encode_backward: dtype must be f32 or f16");
    if (B == 0) return NGP_OK;
    ge_levels lv;
    ge_fill_levels(lv, L, S, H);
    int rc = dtype == NGP_F32
        ? ge_backward_d<float>(D, C, grad, inputs, offsets, grad_embeddings, B, L, lv, calc_grad_inputs != 0, dy_dx, grad_inputs, gridtype, align_corners != 0, (hipStream_t)stream)
        : ge_backward_d<_Float16>(D, C, grad, inputs, offsets, grad_embeddings, B, L, lv, calc_grad_inputs != 0, dy_dx, grad_inputs, gridtype, align_corners != 0, (hipStream_t)stream);
    if (rc != NGP_OK) return rc;
    NGP_CHECK_LAUNCH("grid_encode_backward");
    return NGP_OK;
}

// grad_inputs [B,D] (dtype) from grad [L,B,C] without dy_dx: see k_grid_input_backward_recompute.
template <typename T, uint32_t D>
static int ge_backward_inputs_c(uint32_t C, const void* grad, const float* inputs, const void* emb, const int* offsets, void* gi,
                                uint32_t B, uint32_t L, const ge_levels& lv, uint32_t gridtype, bool ac, hipStream_t s) {
#define GE_LAUNCH_BI(CC) hipLaunchKernelGGL((k_grid_input_backward_recompute<T, D, CC>), dim3(ngp_div_up(B, 256u)), dim3(256), 0, s, \
                                            (const T*)grad, inputs, (const T*)emb, offsets, (T*)gi, B, L, lv, gridtype, ac); return NGP_OK;
    switch (C) {
        case 1: GE_LAUNCH_BI(1)
        case 2: GE_LAUNCH_BI(2)
        case 4: GE_LAUNCH_BI(4)
        case 8: GE_LAUNCH_BI(8)
    }
#undef GE_LAUNCH_BI
    return ngp_fail(NGP_EINVAL, "grid_encode_backward_inputs: C must be 1, 2, 4 or 8");
}

extern "C" int ngp_grid_encode_backward_inputs(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                               void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, void* stream) {
    if (B == 0) return NGP_OK;
    NGP_REQUIRE(grad && inputs && embeddings && offsets && grad_inputs, "grid_encode_backward_inputs: null pointer");
    NGP_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, "grid_encode_backward_inputs: L must be in 1..32");
    NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, "grid_encode_backward_inputs: dtype must be f32 or f16");
    ge_levels lv;
    ge_fill_levels(lv, L, S, H);
    hipStream_t s = (hipStream_t)stream;
    const bool ac = align_corners != 0;
    int rc = NGP_EINVAL;
    if (dtype == NGP_F32) {
        switch (D) {
            case 2: rc = ge_backward_inputs_c<float, 2>(C, grad, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, s); break;
            case 3: rc = ge_backward_inputs_c<float, 3>(C, grad, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, s); break;
            default: return ngp_fail(NGP_EINVAL, "grid_encode_backward_inputs: D must be 2 or 3");
        }
    } else {
        switch (D) {
            case 2: rc = ge_backward_inputs_c<_Float16, 2>(C, grad, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, s); break;
            case 3: rc = ge_backward_inputs_c<_Float16, 3>(C, grad, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, s); break;
            default: return ngp_fail(NGP_EINVAL, "grid_encode_backward_inputs: D must be 2 or 3");
        }
    }
    if (rc != NGP_OK) return rc;
    NGP_CHECK_LAUNCH("grid_encode_backward_inputs");
    return NGP_OK;
}

// =====================================================================================================================================
// Binned scatter: the table gradient WITHOUT global atomics (D = 3, C = 2, half gradients -- the reference's hash grid under autocast).
//
// k_grid_backward issues one memory-side float atomic per (run of samples, corner): ~29 M 64-byte requests per 2 M-point step, executed
// at the fabric at ~11 G requests/s whatever their locality (MI355X_MICROARCH.md "Global float atomics": nothing stays in L2) -- 2.7 ms,
// half of a training step.  Here every contribution is summed ON CHIP instead:
//   k_gs_bin         one 1,024-thread workgroup per (1,024 consecutive samples, level): the same arithmetic and the same wave-level run
//                    aggregation as k_grid_backward, but a contribution becomes a 6-byte entry (row inside its slice: u16, value: half2)
//                    that is counting-sorted by SLICE (about 64 per level: 8,192 table rows of a hashed level down to 512 of the coarsest) in LDS and written out as one contiguous,
//                    slice-ordered region plus a directory of slice offsets.  Plain coalesced stores, no atomics, no capacity guess:
//                    a region has room for all 8,192 possible entries.
//   k_gs_accumulate  persistent workgroups draw slices from a ticket counter; a slice's accumulators (<= 128 KiB) live in LDS as 64-BIT
//                    FIXED POINT in units of 2^-24, in which every finite half is an integer: the sum of a slice's entries is EXACT and
//                    independent of the order of the adds (bitwise reproducible), and the adds are integer LDS atomics -- ds_add_u64 runs
//                    at ~15 cycles per 64 lanes on gfx950 while ds_add_f32 / ds_pk_add_f16 serialise at ~390 / ~195 (measured:
//                    tools/micro/lds_atomics.hip; a float32 version of this kernel spent 3.9 of its 4.4 ms in them).  The workgroup walks
//                    every region's piece for its slice (16 lanes per piece, 64 pieces in flight), and writes the finished rows ONCE,
//                    coalesced, as float32 or half -- the whole table is written, so the destination needs no zero fill, and float32
//                    output needs no widening copy afterwards.  An inf / NaN entry (loss-scale overflow) poisons its slice with NaN.
// Arithmetic: products w * grad summed over a run in binary32 and rounded to half (as k_grid_backward, gridencoder.cu:302); the sum over
// entries is exact (the reference and k_grid_backward add in half, in arbitrary order), rounded once on output.
// Algorithmic bytes per point: 512 B of contributions (16 levels x 8 corners x 2 features x 2 B); traffic: 6 B per entry written and
// read once (~58 entries per point on a training batch: runs merge on the coarse levels only) + 76 B per sample read + the table written once.
// Measured (profiles/r15_train_summary.md, 1.80 M points): k_gs_bin 0.43 + k_gs_accumulate 0.39 ms per step against 2.68 ms for k_grid_backward.
// =====================================================================================================================================
static constexpr uint32_t GS_CHUNK = 1024;            // samples per region = threads per k_gs_bin workgroup
static constexpr uint32_t GS_REGION = GS_CHUNK * 8;   // entries a region can hold (every corner of every sample)
static constexpr uint32_t GS_MAX_SLICES = 128;        // slices per level (directory rows: GS_MAX_SLICES + 1)
static constexpr uint32_t GS_SLICE_ROWS = 8192;       // rows per slice of a large level (2 x 64-bit fixed-point accumulators per row = 128 KiB of LDS)
static constexpr uint32_t GS_PASS_SAMPLES = 1u << 22; // samples per pass: one pass for any 4,096-ray x 1,024-step training batch (workspace <= 3.2 GB;
                                                       // more samples run in further passes that add into the output)

// rows per slice as a shift: a level is cut into about 64 slices -- 8,192 rows for a hashed level of 2^19 rows (the most the LDS holds: 128 KiB of accumulators),
// down to 512 rows for the coarsest levels.  The few rows of a coarse level collect as many entries as a whole fine level's slice does, and a slice is summed by
// ONE workgroup: with 4,096-row slices level 0 was a single slice of 1.26 M entries and the launch waited for it (1.22 ms on 4.19 M points, see HISTORY).
__device__ __forceinline__ uint32_t gs_shift(uint32_t level_rows) {
    const uint32_t per = (level_rows + 63u) >> 6;                       // rows per slice for 64 slices
    uint32_t sh = per <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(per - 1u);   // ceil(log2(per))
    return sh < 9u ? 9u : (sh > 13u ? 13u : sh);
}

// A half as a 64-bit fixed-point number in units of 2^-24 (the smallest half subnormal): EXACT for every finite half (|q| < 2^40), so
// the sum of up to 2^22 entries is exact in 64 bits and does not depend on the order of the adds -- the LDS adds are integer adds
// (ds_add_u64: ~15 cycles per 64 lanes on gfx950, measured; ds_add_f32 / ds_pk_add_f16 serialise at ~195-390: tools/micro/lds_atomics.hip).
// Returns false for inf / NaN (a GradScaler overflow): the caller poisons the slice instead.
__device__ __forceinline__ bool gs_half_to_fixed(uint32_t bits, long long& q) {
    const uint32_t e = (bits >> 10) & 31u, m = bits & 1023u;
    const unsigned long long mag = e ? ((unsigned long long)(1024u | m) << (e - 1u)) : (unsigned long long)m;
    q = (bits & 0x8000u) ? -(long long)mag : (long long)mag;
    return e != 31u;
}

struct gs_ws {                                         // workspace layout (byte offsets), computed on the host
    size_t vals, rows, dir, ticket, total;
    uint32_t nchunks;
};

static gs_ws gs_layout(uint32_t B, uint32_t L) {
    gs_ws w;
    const uint32_t b = B < GS_PASS_SAMPLES ? B : GS_PASS_SAMPLES;
    w.nchunks = ngp_div_up(b, GS_CHUNK);
    const size_t regions = (size_t)L * w.nchunks;
    w.ticket = 0;                                      // [1] u32 (+ padding to 256 B)
    w.dir = 256;                                       // [L][GS_MAX_SLICES + 1][nchunks] u16
    w.vals = w.dir + (((size_t)L * (GS_MAX_SLICES + 1) * w.nchunks * 2 + 255) & ~(size_t)255);   // [L][nchunks][GS_REGION] u32
    w.rows = w.vals + regions * GS_REGION * 4;                                                // [L][nchunks][GS_REGION] u16
    w.total = w.rows + regions * GS_REGION * 2;
    return w;
}

__global__ __launch_bounds__(1024) void k_gs_bin(const _Float16* __restrict__ grad, const float* __restrict__ inputs, const int* __restrict__ offsets,
                                                 uint32_t* __restrict__ g_vals, uint16_t* __restrict__ g_rows, uint16_t* __restrict__ g_dir,
                                                 uint32_t B, uint32_t first, uint32_t count, uint32_t nchunks, ge_levels lv, uint32_t gridtype,
                                                 bool align_corners, uint32_t* __restrict__ ticket, const uint32_t* __restrict__ list,
                                                 const uint32_t* __restrict__ list_count) {
    if (ticket && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) *ticket = 0u;       // for the k_gs_accumulate that follows on the stream (saves a memset launch)
    // listed samples (the field's training backward: only the samples that got a gradient): gradient row i belongs to the sample at inputs[list[i]], i < *list_count
    if (list_count) {
        const uint32_t n = *list_count;
        count = n > first ? (n - first < count ? n - first : count) : 0u;
        if (blockIdx.x * GS_CHUNK >= count) {              // nothing listed for this workgroup: an empty region
            if (threadIdx.x <= GS_MAX_SLICES) g_dir[((size_t)blockIdx.y * (GS_MAX_SLICES + 1) + threadIdx.x) * nchunks + blockIdx.x] = 0;
            return;
        }
    }
    __shared__ uint32_t s_vals[GS_REGION];             // 32 KiB: the region, slice-sorted
    __shared__ uint16_t s_rows[GS_REGION];             // 16 KiB
    __shared__ uint32_t s_hist[GS_MAX_SLICES + 1];     // entries per slice, then (after the scan) first entry of each slice
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    constexpr uint32_t D = 3;
    const uint32_t tid = threadIdx.x, chunk = blockIdx.x, level = blockIdx.y;
    const uint32_t i = chunk * GS_CHUNK + tid;         // sample inside this pass
    bool valid = i < count;
    const uint32_t b = first + (valid ? i : 0u);           // the gradient's row
    const uint32_t bx = list ? list[b] : b;                // the sample's position
    const uint32_t level_rows = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t shift = gs_shift(level_rows);
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    if (tid <= GS_MAX_SLICES) s_hist[tid] = 0u;

    float pos[D];
    uint32_t pg[D];
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float x = inputs[(uint64_t)bx * D + d];
        if (x < 0 || x > 1) valid = false;             // out of range: contributes nothing (gridencoder.cu:251-258)
        pos[d] = x * scale + (align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    const h2 gin = *reinterpret_cast<const h2*>(grad + ((uint64_t)level * B + b) * 2);
    const float g0 = valid ? (float)gin.x : 0.0f, g1 = valid ? (float)gin.y : 0.0f;

    // runs of consecutive lanes in the same cell (consecutive samples of a ray): summed across the run, the last lane emits
    const int lane = (int)(tid & 63u);
    // (run DETECTION looks one lane down with __shfl_up = ds_bpermute, four per wave: DPP wave_shr:1 does not carry across the rows of 16 lanes on this
    // chip and cut the runs at lanes 16 / 32 / 48.  The run SUMS below are DPP only.)
    bool same = valid && lane > 0 && __shfl_up((int)valid, 1, 64) != 0;
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) same = (__shfl_up(pg[d], 1, 64) == pg[d]) && same;
    const unsigned long long heads = __ballot(!same);
    const int start = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    const bool merge = heads != ~0ull;                 // wave-uniform: false on the fine levels, where every sample sits in a cell of its own
    float take[6];                                     // see the run sums below
    #pragma unroll
    for (int k = 0; k < 4; k++) take[k] = (lane - (1 << k) >= start) ? 1.0f : 0.0f;        // row_shr:2^k (a source outside the row reads 0)
    take[4] = (start < (lane & ~15)) ? 1.0f : 0.0f;                                        // the run reaches into the row below
    take[5] = (lane >= 32 && start < 32) ? 1.0f : 0.0f;                                    // ... and into the lower half of the wave
    // how this level is indexed (uniform over the workgroup): the strides of get_grid_index, and whether they all fit (dense) or the level is hashed
    const uint32_t side = align_corners ? resolution : resolution + 1u;
    const uint32_t s1 = side, s2 = side * side;
    const bool idx_dense = (uint64_t)side * side * side <= (uint64_t)level_rows;           // every stride fits: x + y s1 + z s2 < rows, no modulo
    const bool idx_pow2 = !idx_dense && gridtype == 0 && (level_rows & (level_rows - 1u)) == 0;   // hashed, 2^k rows: the modulo is a mask
    __syncthreads();                                   // s_hist is zero

    uint32_t e_val[8], e_key[8], e_rank[8];
    #pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float wi = 1;
        uint32_t pl[D];
        #pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { wi *= 1 - pos[d]; pl[d] = pg[d]; }
            else { wi *= pos[d]; pl[d] = pg[d] + 1; }
        }
        float v0 = wi * g0, v1 = wi * g1;
        if (merge) {
            // segmented inclusive sum over the run, on the VALU alone (DPP): four steps inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the carry
            // from the row below (row_bcast:15 into rows 1 and 3) and from the lower half (row_bcast:31 into rows 2 and 3).  Whether a lane may take a
            // step's value is the same for all eight corners: 1.0 / 0.0 multipliers computed once (take[]), a step is one fused multiply-add per value
            // (fma(o, 1, v) = o + v exactly, fma(o, 0, v) = v; an inf / NaN contribution -- a loss-scale overflow -- may spread to a neighbouring run
            // through 0 * inf: the step is skipped either way).  The 96 ds_bpermute of a shuffle-based scan kept the CU's LDS pipe busy for ~11 us per
            // workgroup: THAT, not the VALU, bounded the kernel on the levels where runs exist.
            #define GS_DPP(v, ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rows, 0xF, true))
            #define GS_STEP(ctrl, rows, k) { const float o0 = GS_DPP(v0, ctrl, rows), o1 = GS_DPP(v1, ctrl, rows); \
                                             v0 = __builtin_fmaf(o0, take[k], v0); v1 = __builtin_fmaf(o1, take[k], v1); }
            GS_STEP(0x111, 0xF, 0) GS_STEP(0x112, 0xF, 1) GS_STEP(0x114, 0xF, 2) GS_STEP(0x118, 0xF, 3)
            GS_STEP(0x142, 0xA, 4) GS_STEP(0x143, 0xC, 5)
            #undef GS_STEP
            #undef GS_DPP
        }
        const bool send = tail && valid && !(v0 == 0.0f && v1 == 0.0f);
        // get_grid_index (gridencoder.cu:54-72) without its `% hashmap_size` where that is the identity or a mask -- a 32-bit division by a
        // run-time divisor is ~20 VALU instructions, eight times per lane, in a kernel that is VALU-bound (profiles/r15_train): a level is either
        // dense (its last stride fits: the index is below the row count by construction) or hashed with 2^k rows (mask); wave-uniform choice
        uint32_t row;
        if (idx_dense) row = pl[0] + pl[1] * s1 + pl[2] * s2;
        else if (idx_pow2) row = (pl[0] ^ (pl[1] * 2654435761u) ^ (pl[2] * 805459861u)) & (level_rows - 1u);
        else row = ge_index<D>(gridtype, align_corners, level_rows, resolution, pl);
        h2 hv;
        hv.x = ngp_f2h(v0);
        hv.y = ngp_f2h(v1);
        e_val[idx] = __builtin_bit_cast(uint32_t, hv);
        uint32_t slice = row >> shift;
        if (slice >= GS_MAX_SLICES) slice = GS_MAX_SLICES - 1;                              // (never for levels of <= 2^19 rows; the host checks)
        e_key[idx] = send ? ((slice << 16) | (row - (slice << shift))) : 0xFFFFFFFFu;
        e_rank[idx] = send ? atomicAdd(&s_hist[slice], 1u) : 0u;                            // ds_add_rtn_u32: the entry's rank inside its slice
    }
    __syncthreads();
    if (tid < 64) {                                    // exclusive scan of the 128 slice counts by one wave (two per lane)
        const uint32_t c0 = s_hist[2 * tid], c1 = s_hist[2 * tid + 1];
        uint32_t incl = c0 + c1;
        #pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if ((int)tid >= off) incl += o;
        }
        s_hist[2 * tid] = incl - c0 - c1;
        s_hist[2 * tid + 1] = incl - c1;
        if (tid == 63) s_hist[GS_MAX_SLICES] = incl;
    }
    __syncthreads();
    #pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        if (e_key[idx] != 0xFFFFFFFFu) {
            const uint32_t p = s_hist[e_key[idx] >> 16] + e_rank[idx];
            s_vals[p] = e_val[idx];
            s_rows[p] = (uint16_t)(e_key[idx] & 0xFFFFu);
        }
    }
    __syncthreads();
    const uint32_t total = s_hist[GS_MAX_SLICES];
    const size_t region = ((size_t)level * nchunks + chunk) * GS_REGION;
    for (uint32_t k = tid; k < total; k += GS_CHUNK) g_vals[region + k] = s_vals[k];
    uint32_t* rows32 = reinterpret_cast<uint32_t*>(g_rows + region);
    const uint32_t* s_rows32 = reinterpret_cast<const uint32_t*>(s_rows);
    for (uint32_t k = tid; k < (total + 1) / 2; k += GS_CHUNK) rows32[k] = s_rows32[k];
    if (tid <= GS_MAX_SLICES) g_dir[((size_t)level * (GS_MAX_SLICES + 1) + tid) * nchunks + chunk] = (uint16_t)s_hist[tid];
}

// the (level, slice) of ticket t, levels from the finest down so that the large hashed levels start first.  Evaluated by one wave: lane l
// holds level L-1-l (L <= 32), an inclusive scan of the slice counts finds the level in one step instead of a 16-deep chain of loads.
__device__ __forceinline__ uint32_t gs_decode_wave(const int* __restrict__ offsets, uint32_t level_lo, uint32_t L, uint32_t t, uint32_t lane) {
    uint32_t n = 0;
    const int l = (int)L - 1 - (int)lane;              // levels [level_lo, L): the caller may sum the table one group of levels at a time
    if (l >= (int)level_lo) {
        const uint32_t rows = (uint32_t)(offsets[l + 1] - offsets[l]);
        const uint32_t sh = gs_shift(rows);
        n = (rows + (1u << sh) - 1) >> sh;
        if (n > GS_MAX_SLICES) n = GS_MAX_SLICES;
    }
    uint32_t incl = n;
    #pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += o;
    }
    const unsigned long long hit = __ballot(l >= (int)level_lo && t < incl);     // first lane whose inclusive count exceeds t
    if (hit == 0ull) return 0xFFFFFFFFu;
    const int first = __ffsll((long long)hit) - 1;
    const uint32_t before = __shfl(incl - n, first, 64);
    return ((uint32_t)((int)L - 1 - first) << 16) | (t - before);
}

template <typename OUT_T>
__global__ __launch_bounds__(1024) void k_gs_accumulate(const uint32_t* __restrict__ g_vals, const uint16_t* __restrict__ g_rows,
                                                        const uint16_t* __restrict__ g_dir, const int* __restrict__ offsets, uint32_t* ticket,
                                                        OUT_T* __restrict__ out, uint32_t level_lo, uint32_t L, uint32_t nchunks, float out_scale,
                                                        bool add_to_out) {
    extern __shared__ unsigned long long s_acc[];      // [GS_SLICE_ROWS][2] 64-bit fixed point (units of 2^-24)
    __shared__ uint32_t s_ticket, s_poison;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, sub = lane >> 4, l16 = lane & 15u;
    for (;;) {
        __syncthreads();                               // the previous slice is written out; s_ticket may be overwritten
        if (wave == 0) {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(ticket, 1u);
            t = __shfl(t, 0, 64);
            const uint32_t code = gs_decode_wave(offsets, level_lo, L, t, lane);
            if (lane == 0) { s_ticket = code; s_poison = 0u; }
        }
        __syncthreads();
        if (s_ticket == 0xFFFFFFFFu) return;
        const uint32_t level = s_ticket >> 16, slice = s_ticket & 0xFFFFu;
        const uint32_t level_rows = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t shift = gs_shift(level_rows);
        const uint32_t row0 = slice << shift;
        uint32_t nrows = level_rows - row0;
        if (nrows > (1u << shift) && slice + 1 < GS_MAX_SLICES) nrows = 1u << shift;
        if (nrows > GS_SLICE_ROWS) nrows = GS_SLICE_ROWS;                                   // (rows beyond were never binned here: host check)
        for (uint32_t k = tid; k < nrows * 2; k += 1024) s_acc[k] = 0ull;
        __syncthreads();
        const uint16_t* dir0 = g_dir + ((size_t)level * (GS_MAX_SLICES + 1) + slice) * nchunks;
        const uint16_t* dir1 = dir0 + nchunks;
        // a wave-iteration = 4 regions, 16 lanes per region's piece; iterations are dealt round-robin to the 16 waves (every wave busy for any
        // number of regions); the directory entry of the NEXT iteration is loaded before this one's entries are consumed
        const uint32_t n_it = (nchunks + 3) / 4;
        uint32_t off = 0, cnt = 0;
        if (wave < n_it) { const uint32_t r = wave * 4 + sub; if (r < nchunks) { off = dir0[r]; cnt = (uint32_t)dir1[r] - off; } }
        for (uint32_t it = wave; it < n_it; it += 16) {
            const uint32_t o = off, c = cnt, r = it * 4 + sub;
            off = 0; cnt = 0;
            if (it + 16 < n_it) { const uint32_t rn = (it + 16) * 4 + sub; if (rn < nchunks) { off = dir0[rn]; cnt = (uint32_t)dir1[rn] - off; } }
            const size_t base = ((size_t)level * nchunks + r) * GS_REGION + o;
            // GS_PER_LANE entries per lane per trip, all their loads issued before the first add: the kernel is a chain of load -> LDS round trips, so
            // what counts is bytes in flight (2 per lane: 1.11 ms, 4: 0.93, 8: 0.88 on 4.19 M points; issuing the next iteration's first trip before
            // this one's adds on top of that: 0.90)
            constexpr uint32_t GS_PER_LANE = 8;
            for (uint32_t k = l16; k < c; k += 16 * GS_PER_LANE) {
                uint32_t v[GS_PER_LANE], rw[GS_PER_LANE];
                #pragma unroll
                for (uint32_t j = 0; j < GS_PER_LANE; j++) {
                    const bool in = k + 16 * j < c;
                    v[j] = in ? g_vals[base + k + 16 * j] : 0u;
                    rw[j] = in ? (uint32_t)g_rows[base + k + 16 * j] : 0xFFFFu;
                }
                #pragma unroll
                for (uint32_t j = 0; j < GS_PER_LANE; j++) {
                    long long q0, q1;
                    const bool f0 = gs_half_to_fixed(v[j] & 0xFFFFu, q0), f1 = gs_half_to_fixed(v[j] >> 16, q1);
                    if (!(f0 && f1)) s_poison = 1u;
                    else if (rw[j] < nrows) {
                        atomicAdd(&s_acc[2 * rw[j]], (unsigned long long)q0);               // ds_add_u64, no return
                        atomicAdd(&s_acc[2 * rw[j] + 1], (unsigned long long)q1);
                    }
                }
            }
        }
        __syncthreads();
        OUT_T* dst = out + ((size_t)(uint32_t)offsets[level] + row0) * 2;
        const bool poison = s_poison != 0u;            // an inf / NaN contribution (loss-scale overflow): the slice reports NaN, the step is skipped
        for (uint32_t k = tid; k < nrows * 2; k += 1024) {
            double v = (double)(long long)s_acc[k] * (1.0 / 16777216.0) * (double)out_scale;   // exact sum ...
            if (add_to_out) v += (double)(float)dst[k];
            dst[k] = poison ? (OUT_T)__builtin_nanf("") : (OUT_T)v;                            // ... ONE rounding on output: double -> float, or double -> half directly
        }
    }
}

extern "C" size_t ngp_grid_scatter_binned_workspace(uint32_t B, uint32_t L) {
    return gs_layout(B, L).total;
}

static int gs_run(const char* who, const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t L, float S, uint32_t H,
                  uint32_t max_level_rows, uint32_t gridtype, int align_corners, int out_dtype, float out_scale, bool do_bin, bool do_sum, uint32_t level_lo,
                  uint32_t level_hi, const uint32_t* list, const uint32_t* list_count, void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE((list == nullptr) == (list_count == nullptr), "%s: a sample list needs its count (and the other way round)", who);
    NGP_REQUIRE(!list || B <= GS_PASS_SAMPLES, "%s: a listed scatter takes one pass (B <= 2^22 samples)", who);
    NGP_REQUIRE(offsets && (!do_sum || grad_embeddings) && (B == 0 || !do_bin || (grad && inputs)), "%s: null pointer", who);
    NGP_REQUIRE(L >= 1 && L <= GE_MAX_LEVELS, "%s: L must be in 1..32", who);
    NGP_REQUIRE(level_lo < level_hi && level_hi <= L, "%s: bad level range", who);
    NGP_REQUIRE(out_dtype == NGP_F32 || out_dtype == NGP_F16, "%s: out_dtype must be f32 or f16", who);
    NGP_REQUIRE(max_level_rows >= 1 && max_level_rows <= 64u * GS_SLICE_ROWS,
                "%s: a level may have at most 2^19 rows (use grid_encode_backward for larger tables)", who);
    const bool split = !(do_bin && do_sum && level_lo == 0 && level_hi == L);
    NGP_REQUIRE(!split || B <= GS_PASS_SAMPLES, "%s: binning and summing in separate calls needs B <= 2^22 samples (one pass)", who);
    const gs_ws w = gs_layout(B, L);
    NGP_REQUIRE(workspace && workspace_bytes >= w.total, "%s: workspace too small (see ngp_grid_scatter_binned_workspace)", who);
    hipStream_t s = (hipStream_t)stream;
    ge_levels lv;
    ge_fill_levels(lv, L, S, H);
    char* base = (char*)workspace;
    const size_t lds = (size_t)GS_SLICE_ROWS * 2 * sizeof(unsigned long long);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return ngp_fail(NGP_ELAUNCH, "%s: no current device", who);
    {   // the raised dynamic-LDS limit is a per-DEVICE function attribute: set once on every device this process scatters on (one process may drive several)
        static std::atomic<unsigned long long> lds_devices{0};
        const unsigned long long bit = 1ull << (dev & 63);
        if (dev >= 64 || !(lds_devices.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)k_gs_accumulate<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_gs_accumulate<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return ngp_fail(NGP_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", who, lds);
            lds_devices.fetch_or(bit, std::memory_order_release);
        }
    }
    int cus = 256;
    {   // (hipGetDeviceProperties fills a 1.5 KB structure through the driver on every call: asked once per device)
        static std::atomic<int> cu_count[64];
        if (dev < 64) {
            int c = cu_count[dev].load(std::memory_order_relaxed);
            if (c == 0) {
                hipDeviceProp_t p;
                c = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
                cu_count[dev].store(c, std::memory_order_relaxed);
            }
            cus = c;
        }
    }
    uint32_t first = 0;
    bool add = false;
    do {                                               // B == 0: one pass that writes zeros (the whole table is always written)
        const uint32_t count = B - first < GS_PASS_SAMPLES ? B - first : GS_PASS_SAMPLES;
        const uint32_t nchunks = count ? ngp_div_up(count, GS_CHUNK) : 0u;
        if (do_bin && nchunks)
            hipLaunchKernelGGL(k_gs_bin, dim3(nchunks, L), dim3(GS_CHUNK), 0, s, (const _Float16*)grad, inputs, offsets, (uint32_t*)(base + w.vals),
                               (uint16_t*)(base + w.rows), (uint16_t*)(base + w.dir), B, first, count, nchunks, lv, gridtype, align_corners != 0,
                               do_sum ? (uint32_t*)(base + w.ticket) : (uint32_t*)nullptr, list, list_count);
        if (do_sum) {
            // the ticket starts at zero: the bin launch just before wrote it, otherwise (no bin launch in this call) a memset does
            if (!(do_bin && nchunks) && hipMemsetAsync(base + w.ticket, 0, 4, s) != hipSuccess) return ngp_fail(NGP_ELAUNCH, "%s: memset failed", who);
            if (out_dtype == NGP_F32)
                hipLaunchKernelGGL(k_gs_accumulate<float>, dim3(cus), dim3(1024), lds, s, (const uint32_t*)(base + w.vals), (const uint16_t*)(base + w.rows),
                                   (const uint16_t*)(base + w.dir), offsets, (uint32_t*)(base + w.ticket), (float*)grad_embeddings, level_lo, level_hi, nchunks,
                                   out_scale, add);
            else
                hipLaunchKernelGGL(k_gs_accumulate<_Float16>, dim3(cus), dim3(1024), lds, s, (const uint32_t*)(base + w.vals), (const uint16_t*)(base + w.rows),
                                   (const uint16_t*)(base + w.dir), offsets, (uint32_t*)(base + w.ticket), (_Float16*)grad_embeddings, level_lo, level_hi, nchunks,
                                   out_scale, add);
        }
        first += count;
        add = true;
    } while (first < B);
    NGP_CHECK_LAUNCH(who);
    return NGP_OK;
}

extern "C" int ngp_grid_scatter_binned(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                       uint32_t B, uint32_t L, float S, uint32_t H, uint32_t max_level_rows, uint32_t gridtype, int align_corners,
                                       int out_dtype, float out_scale, void* workspace, size_t workspace_bytes, void* stream) {
    return gs_run("grid_scatter_binned", grad, inputs, offsets, grad_embeddings, B, L, S, H, max_level_rows, gridtype, align_corners, out_dtype, out_scale,
                  true, true, 0, L, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// The same for a LISTED batch: gradient row i (of every level, rows B apart as above) belongs to the sample at inputs[list[i]], i < *list_count <= B; both on the
// device, read by the kernels (the host never learns the count: no synchronisation).  The field's native training backward hands over the samples that got a
// gradient this way (ngp_field_train_live_list): the rest of the batch -- half of it on a converged scene -- is neither binned nor read.  B <= 2^22.
extern "C" int ngp_grid_scatter_binned_listed(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                              uint32_t B, uint32_t L, float S, uint32_t H, uint32_t max_level_rows, uint32_t gridtype, int align_corners,
                                              int out_dtype, float out_scale, const uint32_t* list, const uint32_t* list_count,
                                              void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(list && list_count, "grid_scatter_binned_listed: null list");
    return gs_run("grid_scatter_binned_listed", grad, inputs, offsets, grad_embeddings, B, L, S, H, max_level_rows, gridtype, align_corners, out_dtype, out_scale,
                  true, true, 0, L, list, list_count, workspace, workspace_bytes, stream);
}

// The same in two steps, for a caller that wants the table one GROUP OF LEVELS at a time (the data-parallel gradient exchange starts the all-reduce
// of a group's rows while the next group is summed): phase 1 bins all levels (grad_embeddings unused), phase 2 sums levels [level_lo, level_hi) into
// their rows of grad_embeddings; any number of phase-2 calls over one phase-1 call, same workspace, same stream order.  B <= 2^22.
extern "C" int ngp_grid_scatter_binned_phase(int phase, const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                             uint32_t B, uint32_t L, uint32_t level_lo, uint32_t level_hi, float S, uint32_t H, uint32_t max_level_rows,
                                             uint32_t gridtype, int align_corners, int out_dtype, float out_scale, void* workspace, size_t workspace_bytes,
                                             void* stream) {
    NGP_REQUIRE(phase == 1 || phase == 2, "grid_scatter_binned_phase: phase must be 1 (bin) or 2 (sum a group of levels)");
    return gs_run("grid_scatter_binned_phase", grad, inputs, offsets, grad_embeddings, B, L, S, H, max_level_rows, gridtype, align_corners, out_dtype, out_scale,
                  phase == 1, phase == 2, phase == 1 ? 0u : level_lo, phase == 1 ? L : level_hi, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// ... and for a listed batch (list, list_count as in ngp_grid_scatter_binned_listed; phase 2 does not read them)
extern "C" int ngp_grid_scatter_binned_phase_listed(int phase, const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                                    uint32_t B, uint32_t L, uint32_t level_lo, uint32_t level_hi, float S, uint32_t H, uint32_t max_level_rows,
                                                    uint32_t gridtype, int align_corners, int out_dtype, float out_scale, const uint32_t* list,
                                                    const uint32_t* list_count, void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(phase == 1 || phase == 2, "grid_scatter_binned_phase_listed: phase must be 1 (bin) or 2 (sum a group of levels)");
    NGP_REQUIRE(list && list_count, "grid_scatter_binned_phase_listed: null list");
    return gs_run("grid_scatter_binned_phase_listed", grad, inputs, offsets, grad_embeddings, B, L, S, H, max_level_rows, gridtype, align_corners, out_dtype, out_scale,
                  phase == 1, phase == 2, phase == 1 ? 0u : level_lo, phase == 1 ? L : level_hi, list, list_count, workspace, workspace_bytes, stream);
}
