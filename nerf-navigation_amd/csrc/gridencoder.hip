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
