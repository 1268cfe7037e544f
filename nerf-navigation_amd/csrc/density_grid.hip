// density_grid.hip -- occupancy-grid maintenance as native gfx950 ops.
//
// The reference keeps its density grid current with Python loops over torch ops (nerf/renderer.py:446-537
// update_extra_state, :381-442 mark_untrained_grid): meshgrid -> morton3D -> jitter -> density query -> indexed scatter ->
// masked EMA/max -> mean -> packbits, ~40 small launches and a 16.8 MB temporary per sweep.  Here the sweep is
//   ngp_density_grid_sample   1 launch (full sweep) / 4 launches (partial: compaction of the occupied cells + sampling)
//   <caller evaluates the density at the sample points: any field>
//   ngp_density_grid_update   3-5 launches: [fill, scatter-max,] EMA + per-block partial means, mean + threshold, packbits
// Layout: the grid is [cascade][H^3] f32 in Morton order (as the reference); the FULL sweep enumerates its points in grid order
// (point e = cell e), so the density query writes straight into the grid's order and no scatter exists at all.
//
// Random numbers: the reference draws torch.rand_like / torch.randint from the global generator.  Here every sweep is one
// pcg32 stream `pcg32(seed, seq = iteration)` (raymarching/src/pcg32.h) and sample e owns draws [16 e, 16 e + 16), reached
// with the O(log n) advance: deterministic, identical on every rank and restated by the oracle
// (oracle/callers_oracle.py: grid_update_randoms).
//
// All position arithmetic is binary32 with one rounding per written operation (-ffp-contract=off), in the order of
// nerf/renderer.py:473-483; per-cascade constants that the reference computes as Python floats (double) and then narrows are
// computed in double on the host.
#include "ngp_device.h"

static constexpr uint32_t DG_BLOCK = 256;
static constexpr uint32_t DG_MAX_CAS = 8;
static constexpr uint32_t DG_RNG_STRIDE = 16;

struct dg_cascades {
    float span[DG_MAX_CAS];      // (float)(b - b / H), b = min(2^cas, bound)      (nerf/renderer.py:476-478)
    float half[DG_MAX_CAS];      // (float)(b / H)
};

static inline dg_cascades dg_make_cascades(uint32_t cascade, uint32_t H, float bound) {
    dg_cascades c;
    for (uint32_t k = 0; k < DG_MAX_CAS; k++) {
        const double b = (k < cascade) ? ((double)(1u << k) < (double)bound ? (double)(1u << k) : (double)bound) : 0.0;
        c.half[k] = (float)(b / (double)H);
        c.span[k] = (float)(b - b / (double)H);
    }
    return c;
}

// xyz of cell (cx, cy, cz) of a cascade with jitter r in [0,1)^3:  (2 c / (H-1) - 1) * span + (2 r - 1) * half
__device__ __forceinline__ void dg_position(uint32_t cx, uint32_t cy, uint32_t cz, float rx, float ry, float rz, float Hm1, float span,
                                            float half, float* out) {
    const float ux = 2.0f * (float)cx / Hm1 - 1.0f, uy = 2.0f * (float)cy / Hm1 - 1.0f, uz = 2.0f * (float)cz / Hm1 - 1.0f;
    out[0] = ux * span + (rx * 2.0f - 1.0f) * half;
    out[1] = uy * span + (ry * 2.0f - 1.0f) * half;
    out[2] = uz * span + (rz * 2.0f - 1.0f) * half;
}

// ---------------------------------------------------------------------------
// sampling
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(DG_BLOCK) void k_dg_sample_full(uint32_t n, uint32_t H3, uint32_t H, dg_cascades cc, uint64_t seed,
                                                             uint64_t iteration, float* __restrict__ xyzs) {
    const uint32_t e = blockIdx.x * DG_BLOCK + threadIdx.x;
    if (e >= n) return;
    const uint32_t cas = e / H3, m = e - cas * H3;
    ngp_pcg32 rng; rng.seed(seed, iteration);
    rng.advance((uint64_t)e * DG_RNG_STRIDE);
    const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
    float p[3];
    dg_position(ngp_compact3(m), ngp_compact3(m >> 1), ngp_compact3(m >> 2), rx, ry, rz, (float)(H - 1), cc.span[cas], cc.half[cas], p);
    xyzs[3ull * e] = p[0]; xyzs[3ull * e + 1] = p[1]; xyzs[3ull * e + 2] = p[2];
}

// occupied cells (density_grid > 0) of every cascade, ascending, as torch.nonzero lists them (nerf/renderer.py:493)
__global__ __launch_bounds__(DG_BLOCK) void k_dg_occ_count(const float* __restrict__ grid, uint32_t n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds4[DG_BLOCK / 64];
    const uint32_t e = blockIdx.x * DG_BLOCK + threadIdx.x;
    const bool keep = (e < n) && (grid[e] > 0.0f);
    const unsigned long long mask = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// exclusive scan of the block sums, restarted at every cascade (blocks_per_cas blocks each); n_occ[cas] = the cascade's total
__global__ __launch_bounds__(DG_BLOCK) void k_dg_occ_scan(uint32_t* __restrict__ block_sums, uint32_t blocks_per_cas, uint32_t cascade,
                                                          uint32_t* __restrict__ n_occ) {
    __shared__ uint32_t lds4[DG_BLOCK / 64];
    __shared__ uint32_t carry;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t cas = 0; cas < cascade; cas++) {
        uint32_t* sums = block_sums + (size_t)cas * blocks_per_cas;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < blocks_per_cas; i0 += DG_BLOCK) {
            const uint32_t i = i0 + threadIdx.x;
            const uint32_t v = (i < blocks_per_cas) ? sums[i] : 0u;
            uint32_t s = v;
            #pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(s, off, 64);
                if ((int)lane >= off) s += up;
            }
            if (lane == 63u) lds4[wave] = s;
            __syncthreads();
            uint32_t base = 0, total = 0;
            #pragma unroll
            for (uint32_t w = 0; w < DG_BLOCK / 64; w++) { const uint32_t t = lds4[w]; if (w < wave) base += t; total += t; }
            const uint32_t c = carry;
            if (i < blocks_per_cas) sums[i] = c + base + s - v;
            __syncthreads();
            if (threadIdx.x == 0) carry = c + total;
            __syncthreads();
        }
        if (threadIdx.x == 0) n_occ[cas] = carry;
        __syncthreads();
    }
}

__global__ __launch_bounds__(DG_BLOCK) void k_dg_occ_write(const float* __restrict__ grid, uint32_t n, uint32_t H3,
                                                           const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ occ) {
    __shared__ uint32_t lds4[DG_BLOCK / 64];
    const uint32_t e = blockIdx.x * DG_BLOCK + threadIdx.x;
    const bool keep = (e < n) && (grid[e] > 0.0f);
    const unsigned long long mask = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) lds4[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = block_sums[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++) base += lds4[w];
    if (keep) {
        const uint32_t cas = e / H3;                              // H3 is a multiple of DG_BLOCK: a block never straddles cascades
        occ[(size_t)cas * H3 + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = e - cas * H3;
    }
}

// partial sweep: per cascade N = H^3/4 random cells, then N picks among the occupied cells (nerf/renderer.py:486-511)
__global__ __launch_bounds__(DG_BLOCK) void k_dg_sample_partial(uint32_t cascade, uint32_t N, uint32_t H3, uint32_t H, dg_cascades cc,
                                                                uint64_t seed, uint64_t iteration, const uint32_t* __restrict__ occ,
                                                                const uint32_t* __restrict__ n_occ, float* __restrict__ xyzs,
                                                                int* __restrict__ cells) {
    const uint32_t s = blockIdx.x * DG_BLOCK + threadIdx.x;
    if (s >= cascade * N) return;
    const uint32_t cas = s / N, i = s - cas * N;
    ngp_pcg32 rng; rng.seed(seed, iteration);
    rng.advance((uint64_t)s * DG_RNG_STRIDE);
    const uint32_t ux = rng.next_uint(), uy = rng.next_uint(), uz = rng.next_uint(), up = rng.next_uint();
    const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float();
    const float q0 = rng.next_float(), q1 = rng.next_float(), q2 = rng.next_float();
    const float Hm1 = (float)(H - 1), span = cc.span[cas], half = cc.half[cas];
    const uint32_t cx = (uint32_t)(((uint64_t)ux * H) >> 32), cy = (uint32_t)(((uint64_t)uy * H) >> 32), cz = (uint32_t)(((uint64_t)uz * H) >> 32);
    const size_t e0 = (size_t)cas * 2u * N + i, e1 = e0 + N;
    float p[3];
    dg_position(cx, cy, cz, r0, r1, r2, Hm1, span, half, p);
    xyzs[3 * e0] = p[0]; xyzs[3 * e0 + 1] = p[1]; xyzs[3 * e0 + 2] = p[2];
    cells[e0] = (int)(cas * H3 + ngp_morton3(cx, cy, cz));
    const uint32_t no = n_occ[cas];
    if (no == 0) {                                                // the reference would raise (randint(0, 0)); here: no sample
        xyzs[3 * e1] = 0.0f; xyzs[3 * e1 + 1] = 0.0f; xyzs[3 * e1 + 2] = 0.0f;
        cells[e1] = -1;
        return;
    }
    const uint32_t m = occ[(size_t)cas * H3 + (uint32_t)(((uint64_t)up * no) >> 32)];
    dg_position(ngp_compact3(m), ngp_compact3(m >> 1), ngp_compact3(m >> 2), q0, q1, q2, Hm1, span, half, p);
    xyzs[3 * e1] = p[0]; xyzs[3 * e1 + 1] = p[1]; xyzs[3 * e1 + 2] = p[2];
    cells[e1] = (int)(cas * H3 + m);
}

// ---------------------------------------------------------------------------
// update
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(DG_BLOCK) void k_dg_fill(float* __restrict__ tmp, uint32_t n4) {
    const uint32_t i = blockIdx.x * DG_BLOCK + threadIdx.x;
    if (i < n4) reinterpret_cast<float4*>(tmp)[i] = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
}

// tmp_grid[cell] = sigma * density_scale; several samples of one cell keep the largest (any one of them is a valid outcome of the
// reference's indexed assignment, :511).  Non-negative floats order like their bit patterns as signed ints and -1.0f is negative.
__global__ __launch_bounds__(DG_BLOCK) void k_dg_scatter_max(const float* __restrict__ sigmas, const int* __restrict__ cells, uint32_t n,
                                                             float density_scale, float* __restrict__ tmp) {
    const uint32_t e = blockIdx.x * DG_BLOCK + threadIdx.x;
    if (e >= n) return;
    const int c = cells[e];
    const float v = sigmas[e] * density_scale;
    if (c >= 0 && v >= 0.0f) atomicMax(reinterpret_cast<int*>(tmp) + c, __float_as_int(v));
}

// valid = grid >= 0 & tmp >= 0 ; grid = max(grid * decay, tmp)   (:523-524) ; per-block sums of clamp(grid, 0) in double
template <bool DIRECT>
__global__ __launch_bounds__(DG_BLOCK) void k_dg_ema(const float* __restrict__ src, float density_scale, float decay, uint32_t n,
                                                     float* __restrict__ grid, double* __restrict__ partial) {
    __shared__ double lds[DG_BLOCK / 64];
    const uint32_t i0 = (blockIdx.x * DG_BLOCK + threadIdx.x) * 4u;
    double acc = 0.0;
    if (i0 < n) {                                                 // n is a multiple of 4 (H^3 is)
        float4 g = *reinterpret_cast<const float4*>(grid + i0);
        const float4 s = *reinterpret_cast<const float4*>(src + i0);
        float* gv = reinterpret_cast<float*>(&g);
        const float* sv = reinterpret_cast<const float*>(&s);
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const float t = DIRECT ? sv[k] * density_scale : sv[k];
            if (gv[k] >= 0.0f && t >= 0.0f) gv[k] = fmaxf(gv[k] * decay, t);
            acc += (double)fmaxf(gv[k], 0.0f);
        }
        *reinterpret_cast<float4*>(grid + i0) = g;
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// mean_density = mean(clamp(grid, 0)) (:525) from the block partials in a fixed order; thresh = min(mean, density_thresh) (:529)
__global__ __launch_bounds__(DG_BLOCK) void k_dg_mean(const double* __restrict__ partial, uint32_t nblocks, uint32_t n, float density_thresh,
                                                      float* __restrict__ mean_density, float* __restrict__ thresh) {
    __shared__ double lds[DG_BLOCK / 64];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < nblocks; i += DG_BLOCK) acc += partial[i];
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mean = (float)(((lds[0] + lds[1]) + (lds[2] + lds[3])) / (double)n);
        mean_density[0] = mean;
        thresh[0] = fminf(mean, density_thresh);
    }
}

__global__ __launch_bounds__(DG_BLOCK) void k_dg_packbits(const float* __restrict__ grid, uint32_t nbytes, const float* __restrict__ thresh_p,
                                                          uint8_t* __restrict__ bitfield) {
    const uint32_t b = blockIdx.x * DG_BLOCK + threadIdx.x;
    if (b >= nbytes) return;
    const float thresh = thresh_p[0];
    const float4 lo = reinterpret_cast<const float4*>(grid)[2ull * b], hi = reinterpret_cast<const float4*>(grid)[2ull * b + 1];
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t bits = 0;
    #pragma unroll
    for (int k = 0; k < 8; k++) bits |= (v[k] > thresh) ? (1u << k) : 0u;             // strict >, bit k = cell 8 b + k (raymarching.cu:270-291)
    bitfield[b] = (uint8_t)bits;
}

// ---------------------------------------------------------------------------
// mark_untrained_grid
// ---------------------------------------------------------------------------

// one lane per cell: is the cell centre inside the frustum of ANY camera (+ 2 half cells of slack)?  poses [B,4,4] row-major c2w.
__global__ __launch_bounds__(DG_BLOCK) void k_dg_mark_untrained(const float* __restrict__ poses, uint32_t B, float tan_x, float tan_y,
                                                                uint32_t n, uint32_t H3, uint32_t H, dg_cascades cc, float* __restrict__ grid) {
    __shared__ float P[64][12];
    const uint32_t e = blockIdx.x * DG_BLOCK + threadIdx.x;
    const uint32_t ec = e < n ? e : n - 1;
    const uint32_t cas = ec / H3, m = ec - cas * H3;
    const float Hm1 = (float)(H - 1), span = cc.span[cas], slack = cc.half[cas] * 2.0f;
    const float px = (2.0f * (float)ngp_compact3(m) / Hm1 - 1.0f) * span;
    const float py = (2.0f * (float)ngp_compact3(m >> 1) / Hm1 - 1.0f) * span;
    const float pz = (2.0f * (float)ngp_compact3(m >> 2) / Hm1 - 1.0f) * span;
    bool seen = false;
    for (uint32_t b0 = 0; b0 < B; b0 += 64) {
        const uint32_t nb = (B - b0) < 64u ? (B - b0) : 64u;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nb * 12; k += DG_BLOCK) P[k / 12][k % 12] = poses[(size_t)(b0 + k / 12) * 16 + (k % 12)];
        __syncthreads();
        for (uint32_t b = 0; b < nb; b++) {
            const float* T = P[b];                                // rows 0..2 of the 4x4: T[4 r + c]
            const float qx = px - T[3], qy = py - T[7], qz = pz - T[11];
            const float camx = (qx * T[0] + qy * T[4]) + qz * T[8];                    // (p - t) @ R  (nerf/renderer.py:425-426)
            const float camy = (qx * T[1] + qy * T[5]) + qz * T[9];
            const float camz = (qx * T[2] + qy * T[6]) + qz * T[10];
            seen |= (camz > 0.0f) && (fabsf(camx) < tan_x * camz + slack) && (fabsf(camy) < tan_y * camz + slack);
        }
    }
    if (e < n && !seen) grid[e] = -1.0f;
}

// ---------------------------------------------------------------------------
// entry points
// ---------------------------------------------------------------------------

static inline bool dg_shape_ok(uint32_t cascade, uint32_t H) {
    return cascade >= 1 && cascade <= DG_MAX_CAS && H >= 8 && H <= 1024 && (H & (H - 1)) == 0 && (uint64_t)cascade * H * H * H <= (1ull << 31);
}

extern "C" uint32_t ngp_density_grid_points(uint32_t cascade, uint32_t H, int partial) {
    const uint64_t H3 = (uint64_t)H * H * H;
    return (uint32_t)(partial ? cascade * 2u * (H3 / 4u) : cascade * H3);
}

// workspace layout: tmp grid [cascade H^3] f32 | occupied-cell lists [cascade H^3] u32 | block sums / partial means | n_occ, thresh
extern "C" size_t ngp_density_grid_workspace(uint32_t cascade, uint32_t H) {
    const size_t n = (size_t)cascade * H * H * H;
    const size_t nblocks = (n + DG_BLOCK - 1) / DG_BLOCK;
    return n * 4 + n * 4 + nblocks * 8 + 256;
}

struct dg_ws {
    float* tmp; uint32_t* occ; double* partial; uint32_t* block_sums; uint32_t* n_occ; float* thresh;
};
static inline dg_ws dg_carve(void* workspace, uint32_t cascade, uint32_t H) {
    const size_t n = (size_t)cascade * H * H * H;
    const size_t nblocks = (n + DG_BLOCK - 1) / DG_BLOCK;
    char* p = (char*)workspace;
    dg_ws w;
    w.tmp = (float*)p; p += n * 4;
    w.occ = (uint32_t*)p; p += n * 4;
    w.partial = (double*)p; w.block_sums = (uint32_t*)p; p += nblocks * 8;         // never live at the same time
    w.n_occ = (uint32_t*)p; p += 64;
    w.thresh = (float*)p;
    return w;
}

extern "C" int ngp_density_grid_sample(const float* density_grid, uint32_t cascade, uint32_t H, float bound, int partial, uint64_t seed,
                                       uint64_t iteration, float* xyzs, int32_t* cells, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    NGP_REQUIRE(dg_shape_ok(cascade, H), "density_grid_sample: cascade must be 1..8 and H a power of two in [8, 1024]");
    NGP_REQUIRE(xyzs && bound > 0.0f, "density_grid_sample: null pointer / bad bound");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t H3 = H * H * H, n = cascade * H3;
    const dg_cascades cc = dg_make_cascades(cascade, H, bound);
    if (!partial) {
        hipLaunchKernelGGL(k_dg_sample_full, dim3(ngp_div_up(n, DG_BLOCK)), dim3(DG_BLOCK), 0, s, n, H3, H, cc, seed, iteration, xyzs);
        NGP_CHECK_LAUNCH("density_grid_sample");
        return NGP_OK;
    }
    NGP_REQUIRE(density_grid && cells, "density_grid_sample: the partial sweep needs density_grid and cells");
    NGP_REQUIRE(workspace && workspace_bytes >= ngp_density_grid_workspace(cascade, H), "density_grid_sample: workspace too small");
    const dg_ws w = dg_carve(workspace, cascade, H);
    const uint32_t nblocks = n / DG_BLOCK, N = H3 / 4;
    hipLaunchKernelGGL(k_dg_occ_count, dim3(nblocks), dim3(DG_BLOCK), 0, s, density_grid, n, w.block_sums);
    hipLaunchKernelGGL(k_dg_occ_scan, dim3(1), dim3(DG_BLOCK), 0, s, w.block_sums, H3 / DG_BLOCK, cascade, w.n_occ);
    hipLaunchKernelGGL(k_dg_occ_write, dim3(nblocks), dim3(DG_BLOCK), 0, s, density_grid, n, H3, w.block_sums, w.occ);
    hipLaunchKernelGGL(k_dg_sample_partial, dim3(ngp_div_up(cascade * N, DG_BLOCK)), dim3(DG_BLOCK), 0, s, cascade, N, H3, H, cc, seed, iteration,
                       w.occ, w.n_occ, xyzs, cells);
    NGP_CHECK_LAUNCH("density_grid_sample");
    return NGP_OK;
}

extern "C" int ngp_density_grid_update(const float* sigmas, const int32_t* cells, uint32_t n_points, float density_scale, float decay,
                                       float density_thresh, uint32_t cascade, uint32_t H, float* density_grid, uint8_t* bitfield,
                                       float* mean_density, void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(dg_shape_ok(cascade, H), "density_grid_update: cascade must be 1..8 and H a power of two in [8, 1024]");
    NGP_REQUIRE(sigmas && density_grid && bitfield && mean_density, "density_grid_update: null pointer");
    NGP_REQUIRE(workspace && workspace_bytes >= ngp_density_grid_workspace(cascade, H), "density_grid_update: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n = cascade * H * H * H;
    const dg_ws w = dg_carve(workspace, cascade, H);
    const uint32_t nb4 = ngp_div_up(n / 4, DG_BLOCK);
    if (cells) {
        hipLaunchKernelGGL(k_dg_fill, dim3(nb4), dim3(DG_BLOCK), 0, s, w.tmp, n / 4);
        if (n_points)
            hipLaunchKernelGGL(k_dg_scatter_max, dim3(ngp_div_up(n_points, DG_BLOCK)), dim3(DG_BLOCK), 0, s, sigmas, cells, n_points, density_scale, w.tmp);
        hipLaunchKernelGGL(k_dg_ema<false>, dim3(nb4), dim3(DG_BLOCK), 0, s, w.tmp, density_scale, decay, n, density_grid, w.partial);
    } else {
        NGP_REQUIRE(n_points == n, "density_grid_update: a full sweep (cells == NULL) carries one sigma per cell");
        hipLaunchKernelGGL(k_dg_ema<true>, dim3(nb4), dim3(DG_BLOCK), 0, s, sigmas, density_scale, decay, n, density_grid, w.partial);
    }
    hipLaunchKernelGGL(k_dg_mean, dim3(1), dim3(DG_BLOCK), 0, s, w.partial, nb4, n, density_thresh, mean_density, w.thresh);
    hipLaunchKernelGGL(k_dg_packbits, dim3(ngp_div_up(n / 8, DG_BLOCK)), dim3(DG_BLOCK), 0, s, density_grid, n / 8, w.thresh, bitfield);
    NGP_CHECK_LAUNCH("density_grid_update");
    return NGP_OK;
}

extern "C" int ngp_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t cascade, uint32_t H,
                                       float bound, float* density_grid, void* stream) {
    NGP_REQUIRE(dg_shape_ok(cascade, H), "mark_untrained_grid: cascade must be 1..8 and H a power of two in [8, 1024]");
    NGP_REQUIRE(poses && density_grid && B > 0 && fx != 0.0f && fy != 0.0f && bound > 0.0f, "mark_untrained_grid: bad argument");
    const uint32_t H3 = H * H * H, n = cascade * H3;
    const dg_cascades cc = dg_make_cascades(cascade, H, bound);
    const float tan_x = (float)((double)cx / (double)fx), tan_y = (float)((double)cy / (double)fy);     // Python floats in the reference (:432-433)
    hipLaunchKernelGGL(k_dg_mark_untrained, dim3(ngp_div_up(n, DG_BLOCK)), dim3(DG_BLOCK), 0, (hipStream_t)stream, poses, B, tan_x, tan_y, n, H3, H,
                       cc, density_grid);
    NGP_CHECK_LAUNCH("mark_untrained_grid");
    return NGP_OK;
}
