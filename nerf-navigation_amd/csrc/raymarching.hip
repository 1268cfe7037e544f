// raymarching.hip -- gfx950 kernels behind the `_raymarching` native surface of the reference
// (raymarching/src/raymarching.h:7-18): AABB slab test, Morton codes, bit packing, the occupancy-grid
// ray marcher (training and inference form) and the alpha compositors.
//
// One lane per ray.  A wave holds 64 rays; blocks are 256 threads (4 waves) so that a launch of N rays needs
// N/256 workgroups -- >= 2500 for an 800x800 frame, enough to fill 256 CUs several times over.  The occupancy
// bitfield (cascade * 128^3 / 8 = 0.5 MB at bound 2) is L2-resident; the kernels are bound by the divergent
// per-ray loops, not by HBM.
#include <atomic>
#include <cstring>
#include "ngp_march.h"

thread_local char ngp_err_buf[512] = {0};

extern "C" int ngp_abi_version(void) { return 1; }
extern "C" const char* ngp_last_error(void) { return ngp_err_buf; }

static constexpr uint32_t RM_BLOCK = 256;
// Kernels with one lane per RAY run long sequential loops over few rays (4,096 per training step; the alive set of the
// inference loop shrinks to a few thousand): 64-thread workgroups spread those waves over 4x as many CUs, each with its own
// L1 / texture path, instead of packing four per CU and leaving most of the chip idle.
static constexpr uint32_t RM_RAY_BLOCK = 64;
static constexpr uint32_t RM_CHUNK = 8;         // samples fetched ahead by the per-ray composite loops

// ---------------------------------------------------------------------------
// near / far
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(RM_BLOCK) void k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ aabb, uint32_t N, float min_near,
                                                       float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (n >= N) return;
    float bb[6];
    #pragma unroll
    for (int i = 0; i < 6; i++) bb[i] = aabb[i];
    float near, far;
    ngp_near_far_inline(rays_o + 3ull * n, rays_d + 3ull * n, bb, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

extern "C" int ngp_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                      float min_near, float* nears, float* fars, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null pointer");
    if (N == 0) return NGP_OK;
    hipLaunchKernelGGL(k_near_far, dim3(ngp_div_up(N, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream,
                       rays_o, rays_d, aabb, N, min_near, nears, fars);
    NGP_CHECK_LAUNCH("near_far_from_aabb");
    return NGP_OK;
}

// ---------------------------------------------------------------------------
// background sphere coordinates (cold path: only with bg_radius > 0)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(RM_BLOCK) void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                           float radius, uint32_t N, float* __restrict__ coords) {
    // reference: raymarching.cu:165-200
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[3ull * n], oy = rays_o[3ull * n + 1], oz = rays_o[3ull * n + 2];
    const float dx = rays_d[3ull * n], dy = rays_d[3ull * n + 1], dz = rays_d[3ull * n + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float C = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    const float RPI = 0.3183098861837907f;
    coords[2ull * n] = 2 * theta * RPI - 1;
    coords[2ull * n + 1] = phi * RPI;
}

extern "C" int ngp_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays_o && rays_d && coords, "sph_from_ray: null pointer");
    if (N == 0) return NGP_OK;
    hipLaunchKernelGGL(k_sph_from_ray, dim3(ngp_div_up(N, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream,
                       rays_o, rays_d, radius, N, coords);
    NGP_CHECK_LAUNCH("sph_from_ray");
    return NGP_OK;
}

// ---------------------------------------------------------------------------
// Morton codes, bit packing
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(RM_BLOCK) void k_morton3D(const int* __restrict__ coords, uint32_t N, int* __restrict__ indices) {
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int)ngp_morton3((uint32_t)coords[3ull * n], (uint32_t)coords[3ull * n + 1], (uint32_t)coords[3ull * n + 2]);
}

__global__ __launch_bounds__(RM_BLOCK) void k_morton3D_invert(const int* __restrict__ indices, uint32_t N, int* __restrict__ coords) {
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (n >= N) return;
    const int ind = indices[n];                      // signed shifts, as the reference (raymarching.cu:251-255)
    coords[3ull * n + 0] = (int)ngp_compact3((uint32_t)(ind >> 0));
    coords[3ull * n + 1] = (int)ngp_compact3((uint32_t)(ind >> 1));
    coords[3ull * n + 2] = (int)ngp_compact3((uint32_t)(ind >> 2));
}

// One lane packs one byte from 8 consecutive floats: two 16-byte loads per lane, fully coalesced across the wave.
__global__ __launch_bounds__(RM_BLOCK) void k_packbits(const float* __restrict__ grid, uint32_t N, float thresh,
                                                       uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[2ull * n];
    const float4 b = reinterpret_cast<const float4*>(grid)[2ull * n + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

extern "C" int ngp_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, "morton3D: null pointer");
    if (N == 0) return NGP_OK;
    hipLaunchKernelGGL(k_morton3D, dim3(ngp_div_up(N, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream, coords, N, indices);
    NGP_CHECK_LAUNCH("morton3D");
    return NGP_OK;
}

extern "C" int ngp_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(coords && indices, "morton3D_invert: null pointer");
    if (N == 0) return NGP_OK;
    hipLaunchKernelGGL(k_morton3D_invert, dim3(ngp_div_up(N, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream, indices, N, coords);
    NGP_CHECK_LAUNCH("morton3D_invert");
    return NGP_OK;
}

extern "C" int ngp_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grid && bitfield, "packbits: null pointer");
    NGP_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 15u) == 0, "packbits: grid must be 16-byte aligned");
    if (N == 0) return NGP_OK;
    hipLaunchKernelGGL(k_packbits, dim3(ngp_div_up(N, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream, grid, N, density_thresh, bitfield);
    NGP_CHECK_LAUNCH("packbits");
    return NGP_OK;
}

// ---------------------------------------------------------------------------
// training march: count -> scan -> write  (reference does both passes in one kernel with atomics,
// raymarching.cu:314-484; the prefix sum makes slot order deterministic: ray_index = n)
// ---------------------------------------------------------------------------

struct march_args {
    const float* rays_o; const float* rays_d; const uint8_t* grid;
    const float* nears; const float* fars;
    float bound, dt_gamma;
    uint32_t max_steps, N, C, H, M, perturb;
    const uint32_t* coarse = nullptr;   // optional coarse occupancy map (global; the kernels stage it in LDS), one bit per 4^3 block
    uint32_t coarse_words = 0;         // 32-bit words per cascade level (H^3 / 64 / 32)
    const float* occ = nullptr;        // optional: the box of everything occupied (k_rm_occupied_box: lo[3], hi[3], pad), where a ray's march may stop
};

// coarse bit b = any cell of Morton block b (64 cells = one aligned 64-bit word of the bitfield) is occupied.  One lane per block word, the
// wave's ballot is 64 coarse bits (two words): 65,536 lanes for a bound-2 grid instead of 2,048 lanes reading 256 bytes each (8 us -> ~2 us)
__global__ __launch_bounds__(RM_BLOCK) void k_rm_build_coarse(const uint8_t* __restrict__ bitfield, uint32_t n_blocks_total,
                                                              uint32_t* __restrict__ coarse) {
    const uint32_t blk = blockIdx.x * RM_BLOCK + threadIdx.x;
    const bool any = blk < n_blocks_total && reinterpret_cast<const uint64_t*>(bitfield)[blk] != 0ull;
    const unsigned long long bits = __ballot(any);
    if ((threadIdx.x & 63u) == 0 && blk < n_blocks_total) {             // n_blocks_total is a multiple of 64 (H >= 16)
        coarse[blk >> 5] = (uint32_t)bits;
        coarse[(blk >> 5) + 1] = (uint32_t)(bits >> 32);
    }
}

static constexpr size_t RM_COARSE_MAX = 48 * 1024;                     // largest map the march kernels stage in LDS (H = 128: 4 KiB per cascade)
static constexpr size_t RM_OCC_BYTES = 64;                             // behind the map in the workspace: the occupied box, 7 floats
// validation switch (process-wide, default 1): the per-op march kernels stop a ray where it leaves the box of everything occupied; 0 = at its own far
static std::atomic<int> rm_occ_box_enabled{1};
extern "C" int ngp_march_set_occupied_box(int enabled) { return rm_occ_box_enabled.exchange(enabled ? 1 : 0, std::memory_order_relaxed); }

// bytes of the coarse map for this grid, or 0 when it is not used (cell index not exact in binary32, H^3 / 64 not a multiple of 32,
// misaligned bitfield, or too large for LDS): exactly the condition under which ngp_march_t caches 64-bit block words
static size_t rm_coarse_bytes(const uint8_t* grid, uint32_t C, uint32_t H) {
    const bool exact = (H & (H - 1u)) == 0u && H >= 16u && (uint64_t)C * H * H * H <= (1ull << 24) && (reinterpret_cast<uintptr_t>(grid) & 15u) == 0u;
    if (!exact) return 0;
    const size_t bytes = (size_t)C * ((size_t)H * H * H / 64) / 8;
    return bytes <= RM_COARSE_MAX ? bytes : 0;
}

// The box of everything occupied, from the coarse map, once per march call (one workgroup): the extent of the set blocks of all cascades on one integer
// lattice (ngp_march.h: ngp_occ_extent_word), reduced with LDS atomics, written in world coordinates one unit wider on every side, + box[6] = how far behind the box a
// march may still look (two of the largest steps).  Beyond that box every cell is empty: the reference tests those cells and finds nothing, so a march that
// stops there produces the same samples.  Needs the cascades nested in powers of two: the condition of the verified block skips (ngp_skip_allowed).
__global__ __launch_bounds__(1024) void k_rm_occupied_box(const uint32_t* __restrict__ coarse, uint32_t coarse_words, uint32_t C, uint32_t H, float top, float unit,
                                                          float* __restrict__ box) {
    __shared__ uint32_t s_ext[6];
    if (threadIdx.x < 6) s_ext[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t nw = C * coarse_words, blocks_per_level = coarse_words * 32u;
    uint32_t nb = 1;
    while (nb * nb * nb < blocks_per_level) nb <<= 1;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = threadIdx.x; i < nw; i += 1024u) {
        const uint32_t level = i / coarse_words;
        ngp_occ_extent_word(coarse[i], i - level * coarse_words, level, C, nb, lo, hi);
    }
    if (hi[0]) {
        #pragma unroll
        for (int k = 0; k < 3; k++) { atomicMax(s_ext + k, (1u << 20) - lo[k]); atomicMax(s_ext + 3 + k, hi[k]); }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const uint32_t k = threadIdx.x;
        const bool any = s_ext[3] != 0u;                               // nothing occupied: an empty box far away, every ray misses it
        box[k] = any ? -top + ((float)((1u << 20) - s_ext[k]) - 1.0f) * unit : 3.0e38f;
        box[3 + k] = any ? -top + ((float)s_ext[3 + k] + 1.0f) * unit : 3.1e38f;
        if (k == 0) box[6] = 2.0f * (((2.0f * 1.7320508075688772f) * (float)(1u << (C - 1u))) / (float)H);
    }
}

// builds the map into `dst` (device) and points the march arguments at it
static void rm_attach_coarse(march_args& a, void* dst, hipStream_t s, bool with_box) {
    const uint32_t n_blocks_total = a.C * (a.H * a.H * a.H / 64);
    hipLaunchKernelGGL(k_rm_build_coarse, dim3(ngp_div_up(n_blocks_total, RM_BLOCK)), dim3(RM_BLOCK), 0, s, a.grid, n_blocks_total, (uint32_t*)dst);
    a.coarse = (const uint32_t*)dst;
    a.coarse_words = a.H * a.H * a.H / 64 / 32;
    // the occupied box (behind the map, RM_OCC_BYTES): one more small launch per call
    if (with_box && rm_occ_box_enabled.load(std::memory_order_relaxed) && ngp_skip_allowed(a.C, a.H, a.bound)) {
        float* box = reinterpret_cast<float*>(static_cast<unsigned char*>(dst) + (size_t)a.C * a.coarse_words * 4);
        hipLaunchKernelGGL(k_rm_occupied_box, dim3(1), dim3(1024), 0, s, a.coarse, a.coarse_words, a.C, a.H, a.C == 1u ? a.bound : (float)(1u << (a.C - 1u)),
                           2.0f * (a.C == 1u ? a.bound : 1.0f) / (float)(a.H / 4u), box);
        a.occ = box;
    }
}

// every thread of a RM_RAY_BLOCK workgroup calls this before marching: copies the map into dynamic LDS (or returns nullptr without one)
// where the march of ray (o, d) may stop: its own far, or a little behind the occupied box (-inf when it misses the box: nothing to march)
__device__ __forceinline__ float rm_march_far(const float* box, const float* o, const float* d, float far) {
    if (!box) return far;
    float n2, f2;
    ngp_near_far_inline(o, d, box, 0.0f, n2, f2);
    return f2 == 3.402823466e+38f ? -__builtin_inff() : fminf(far, f2 + box[6]);
}

template <uint32_t BLOCK = RM_RAY_BLOCK>
__device__ __forceinline__ const uint32_t* rm_stage_coarse(const march_args& a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t rm_lds_coarse[];
    if (!a.coarse) return nullptr;
    const uint32_t nw = a.C * a.coarse_words;
    for (uint32_t i = threadIdx.x * 4; i < nw; i += BLOCK * 4)
        *reinterpret_cast<uint4*>(rm_lds_coarse + i) = *reinterpret_cast<const uint4*>(a.coarse + i);
    __syncthreads();
    return rm_lds_coarse;
}

__device__ __forceinline__ float train_t0(const ngp_march_t& m, float near, uint32_t n, uint32_t perturb) {
    if (!perturb) return near;
    ngp_pcg32 rng; rng.seed(42u);                    // hard-coded seed of the reference (raymarching.cu:489)
    rng.advance((uint64_t)n);
    return near + m.dt_min * rng.next_float();
}

// block-wide inclusive scan of one uint per thread (NW waves): wave scan by DPP-free shuffles, then wave totals in LDS
template <uint32_t NW>
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* lds4, uint32_t& block_total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
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
    for (uint32_t w = 0; w < NW; w++) {
        const uint32_t t = lds4[w];
        if (w < wave) base += t;
        total += t;
    }
    __syncthreads();
    block_total = total;
    return s + base;
}

__global__ __launch_bounds__(RM_RAY_BLOCK) void k_march_train_count(march_args a, int* __restrict__ rays,
                                                                const int* __restrict__ counter,
                                                                uint32_t* __restrict__ block_sums, float* __restrict__ tbuf) {
    __shared__ uint32_t lds4[RM_RAY_BLOCK / 64];
    const uint32_t* lds_coarse = rm_stage_coarse(a);
    const float* occ = a.occ;
    const uint32_t n = blockIdx.x * RM_RAY_BLOCK + threadIdx.x;
    uint32_t num_steps = 0;
    if (n < a.N) {
        ngp_march_t m;
        m.setup(a.rays_o + 3ull * n, a.rays_d + 3ull * n, a.bound, a.dt_gamma, a.max_steps, a.C, a.H, a.grid);
        m.use_coarse(lds_coarse, a.coarse_words);
        const float far = rm_march_far(occ, a.rays_o + 3ull * n, a.rays_d + 3ull * n, a.fars[n]);
        m.allow_skip(a.C, a.H, occ ? far + 4.0f * a.bound : far);
        float t = train_t0(m, a.nears[n], n, a.perturb), x, y, z, dt;
        float* tb = tbuf ? tbuf + (size_t)n * a.max_steps : nullptr;
        while (t < far && num_steps < a.max_steps) {
            if (m.probe(t, x, y, z, dt)) {
                if (tb) tb[num_steps] = t;                            // the sample's parameter: all the fill pass needs
                num_steps++; t += dt;
            }
        }
        const uint32_t slot = (uint32_t)counter[1] + n;
        if (slot < a.N) rays[3ull * slot + 2] = (int)num_steps;      // stash; the write pass completes the record
    }
    uint32_t total;
    (void)block_inclusive_scan<RM_RAY_BLOCK / 64>(num_steps, lds4, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------
// The count pass with ONE WAVE PER RAY (dt_gamma == 0, grids whose cell index is exact: the usual case).
//
// One lane per ray walks ~600 dependent probes while 4,096 rays fill 64 waves of a 1,024-SIMD chip (0.88 ms per training step, all of it
// latency).  With a constant step the points the reference can ever test form a fixed lattice t_0 = start, t_{k+1} = fl(t_k + dt), and inside
// one binade fl(t + dt) = t + du with the same du for every t (ngp_lattice_jump, ngp_march.h): 64 consecutive lattice points are
// t_cur + k du, exactly.  So 64 lanes evaluate 64 lattice points at once (cell, occupancy bit, the cell's exit parameter), and the wave then
// replays the reference's control flow over them -- which points it TESTS -- run by run instead of point by point:
//     an occupied tested point is a sample, and so is every following lattice point of the same cell (after a sample the reference steps
//     by dt and tests again);
//     an empty tested point p sends the reference to the first lattice point >= cell_exit(p) (raymarching.cu:392-402), evaluated here
//     with p's own arithmetic, so that the rounding of the exit parameter decides exactly as it does there.
// A window has ~8 runs, so the replay is ~8 uniform iterations of a ballot and a readlane.  Same samples, same order, same count as
// k_march_train_count, bit for bit (tests/test_gpu_raymarching.py compares both with the oracle).
// ---------------------------------------------------------------------------
// The window march shared by the wave-per-ray kernels: from lattice point t_start, find up to max_new samples of ONE ray (all 64 lanes hold
// the same ray in `m`) and hand each window's samples to emit(found_so_far, smask, tk): smask = the lanes whose lattice point tk is a sample,
// in order (du = the lattice spacing of that window).  Returns the number of samples found.  Requires dt_gamma == 0 and an exact grid (m.blocks_per_level != 0).
template <class Emit>
__device__ __forceinline__ uint32_t rm_wave_march(const ngp_march_t& m, const uint32_t* __restrict__ lds_coarse, uint32_t coarse_words,
                                                  float t_start, float far, uint32_t max_new, bool replay_only, Emit&& emit) {
    const int lane = (int)(threadIdx.x & 63u);
    const float dtc = ngp_clampf(0.0f, m.dt_min, m.dt_max);            // dt(t) for dt_gamma == 0
    float t_cur = t_start;
    uint32_t count = 0;
    float t_skip = -__builtin_inff();                                  // lattice points below it are not tested (an empty cell is being left)
    int guard = 0;
    while (t_cur < far && count < max_new && ++guard < NGP_SKIP_GUARD) {
        // ---- the window's lattice points: t_cur + k du, exact for k <= kmax ----
        const float t1 = t_cur + dtc;                                  // the reference's own step
        const float du = t1 - t_cur;                                   // exact
        if (!(du > 0.0f)) break;                                       // no progress: the reference would spin (its guard ends it)
        int e;
        (void)frexpf(t_cur, &e);
        const float end = __builtin_ldexpf(1.0f, e);
        const float err = dtc - du;
        int kmax = 1;
        if (t1 < end && fabsf(err) != __builtin_ldexpf(1.0f, e - 25)) {
            const float J = floorf((end - t1) * __builtin_amdgcn_rcpf(du)) - 1.0f;     // t1 + j du < end for j <= J whatever the rounding here
            if (J > 0.0f) kmax = 1 + (int)fminf(J, 62.0f);
        }
        const float tk = t_cur + (float)lane * du;                     // exact for lane <= kmax
        const bool valid = lane <= kmax && tk < far;
        // ---- every lane: its point's cell, occupancy and exit parameter (the arithmetic of ngp_march_t::probe) ----
        ngp_point<ngp_march_t> r;
        r.at(m, valid ? tk : t_cur);
        const uint32_t mort = ngp_morton3((uint32_t)r.nx, (uint32_t)r.ny, (uint32_t)r.nz);
        bool occ = false;
        {
            const uint32_t blk = mort >> 6;
            const bool maybe = !lds_coarse || ((lds_coarse[(uint32_t)r.level * coarse_words + (blk >> 5)] >> (blk & 31u)) & 1u);
            if (maybe) {
                const uint2 w = reinterpret_cast<const uint2*>(m.grid)[(uint32_t)r.level * m.blocks_per_level + blk];
                occ = (((mort & 32u) ? w.y : w.x) >> (mort & 31u)) & 1u;
            }
        }
        const float tt = r.cell_exit(m, valid ? tk : t_cur);
        const uint32_t cell = ((uint32_t)r.level << 30) | mort;        // level < 4: the host uses these kernels for C <= 4 only
        const uint32_t prev_cell = __shfl_up(cell, 1, 64);
        const unsigned long long valid_mask = __ballot(valid);
        const unsigned long long occ_mask = __ballot(valid && occ);
        const unsigned long long heads = __ballot(lane == 0 || cell != prev_cell);       // bit k: lane k starts a run of one cell
        // ---- the reference's control flow over the window ----
        unsigned long long smask = 0ull;                               // lattice points that are samples
        uint32_t room = max_new - count;
        // Almost always the tested points are exactly: the first lattice point of every cell's run, and every point of an occupied run.  Each lane
        // that heads an empty run checks that guess against the comparisons the replay below would make (the next point at or beyond its exit
        // parameter is the next run's head, the point before that is not); one ballot accepts the whole window, anything else is replayed.
        const unsigned long long cand0 = valid_mask & __ballot(tk >= t_skip);
        bool replay = cand0 != 0ull && room > 0;
        if (replay && !replay_only) {
            const int j0 = __builtin_ctzll(cand0);                     // first tested point of the window
            const int nv = 64 - __builtin_clzll(valid_mask);           // valid lanes are 0 .. nv - 1 (tk is monotone)
            const unsigned long long H = (heads | (1ull << j0)) & cand0;
            const unsigned long long S = cand0 & occ_mask;
            const unsigned long long later = lane >= 63 ? 0ull : (H & (~0ull << (lane + 1)));
            const int nh = later ? __builtin_ctzll(later) : nv;        // the next run's head (nv: none in this window)
            const float t_nh = t_cur + (float)nh * du, t_before = t_cur + (float)(nh - 1) * du;      // those lanes' own tk
            const bool bad = (((H & ~occ_mask) >> lane) & 1ull) && ((nh < nv && !(t_nh >= tt)) || (nh - 1 > lane && t_before >= tt));
            if (__ballot(bad) == 0ull && (uint32_t)__popcll(S) <= room && valid_mask == (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull))) {
                smask = S;
                const int hl = 63 - __builtin_clzll(H);                // the last run decides what the next window inherits
                t_skip = ((occ_mask >> hl) & 1ull) ? -__builtin_inff() : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), hl));
                replay = false;
            }
        }
        int after = -1;                                                // candidates are lanes > after
        for (int it = 0; replay && it < 66 && room > 0; it++) {
            const unsigned long long low = after < 0 ? 0ull : (after >= 63 ? ~0ull : ((2ull << after) - 1ull));
            const unsigned long long cand = valid_mask & __ballot(tk >= t_skip) & ~low;
            if (!cand) break;
            const int j = __builtin_ctzll(cand);
            if ((occ_mask >> j) & 1ull) {
                // the rest of this cell's run is tested point by point and every point is a sample
                const unsigned long long later_heads = heads & ~((j >= 63) ? ~0ull : ((2ull << j) - 1ull));
                int run_end = later_heads ? __builtin_ctzll(later_heads) - 1 : 63;
                const int last_valid = 63 - __builtin_clzll(valid_mask);
                if (run_end > last_valid) run_end = last_valid;
                if ((uint32_t)(run_end - j + 1) > room) run_end = j + (int)room - 1;
                const unsigned long long upto = run_end >= 63 ? ~0ull : ((2ull << run_end) - 1ull);
                const unsigned long long from = j == 0 ? ~0ull : ~((1ull << j) - 1ull);
                smask |= upto & from;
                room -= (uint32_t)(run_end - j + 1);
                after = run_end;
                t_skip = -__builtin_inff();                            // after a sample the next lattice point is tested whatever came before
            } else {
                t_skip = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), j));
                after = j;                                             // "do t += dt while (t < tt)": at least one step
            }
        }
        if (smask) emit(count, smask, tk, du);
        count += (uint32_t)__popcll(smask);
        // next window starts one real step behind this window's last exact point
        const int klast = kmax < 63 ? kmax : 63;
        t_cur = (t_cur + (float)klast * du) + dtc;
    }
    return count;
}

__global__ __launch_bounds__(RM_RAY_BLOCK) void k_march_train_count_wave(march_args a, int* __restrict__ rays, const int* __restrict__ counter,
                                                                     float* __restrict__ tbuf, uint32_t replay_only) {
    const uint32_t* lds_coarse = rm_stage_coarse(a);
    const float* occ = a.occ;
    const uint32_t n = blockIdx.x;
    const int lane = (int)threadIdx.x;
    ngp_march_t m;
    m.setup(a.rays_o + 3ull * n, a.rays_d + 3ull * n, a.bound, a.dt_gamma, a.max_steps, a.C, a.H, a.grid);
    float* tb = tbuf + (size_t)n * a.max_steps;
    const float far = rm_march_far(occ, a.rays_o + 3ull * n, a.rays_d + 3ull * n, a.fars[n]);     // (windows behind the occupied box hold no sample)
    const uint32_t count = rm_wave_march(m, lds_coarse, a.coarse_words, train_t0(m, a.nears[n], n, a.perturb), far, a.max_steps, replay_only != 0u,
                                         [&](uint32_t found, unsigned long long smask, float tk, float) {
        if ((smask >> lane) & 1ull) tb[found + (uint32_t)__popcll(smask & ((1ull << lane) - 1ull))] = tk;    // the samples' parameters, in order
    });
    if (lane == 0) {
        const uint32_t slot = (uint32_t)counter[1] + n;
        if (slot < a.N) rays[3ull * slot + 2] = (int)count;
    }
}

// per-64-ray totals of the counts the wave kernel left in rays[.][2] (what k_march_train_count computes with its block scan)
__global__ __launch_bounds__(RM_BLOCK) void k_march_train_sum64(const int* __restrict__ rays, const int* __restrict__ counter, uint32_t N,
                                                                uint32_t nblocks, uint32_t* __restrict__ block_sums) {
    const uint32_t b = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t ray_base = (uint32_t)counter[1];
    uint32_t total = 0;
    for (uint32_t k = 0; k < RM_RAY_BLOCK; k++) {
        const uint32_t n = b * RM_RAY_BLOCK + k, slot = ray_base + n;
        if (n < N && slot < N) total += (uint32_t)rays[3ull * slot + 2];
    }
    block_sums[b] = total;
}

// single workgroup: exclusive scan of the per-block totals, then bump the two counters
__global__ __launch_bounds__(RM_BLOCK) void k_march_train_scan(uint32_t* __restrict__ block_sums, uint32_t nblocks,
                                                               int* __restrict__ counter, uint32_t N,
                                                               uint32_t* __restrict__ bases /* [2]: point base, ray base */) {
    __shared__ uint32_t lds4[RM_BLOCK / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = (uint32_t)counter[0];
    __syncthreads();
    const uint32_t start = carry;
    for (uint32_t i0 = 0; i0 < nblocks; i0 += RM_BLOCK) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t v = (i < nblocks) ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t inc = block_inclusive_scan<RM_BLOCK / 64>(v, lds4, total);
        const uint32_t c = carry;
        if (i < nblocks) block_sums[i] = c + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bases[0] = start;
        bases[1] = (uint32_t)counter[1];
        bases[2] = carry;                            // first slot no ray of this call fills; k_march_train_write lowers it to the first dropped ray's
        counter[0] = (int)carry;
        counter[1] = (int)((uint32_t)counter[1] + N);
    }
}

// The three launches above (per-64-ray sums, their scan, the rays' offsets) as ONE single-workgroup launch, for the wave-per-ray count pass on a
// training-sized batch (a 4,096-ray step: 4 rounds of 1,024 rays; 19 us of launches -> one of ~6): the same slots, offsets, counters and bases.
static constexpr uint32_t RM_OFFSETS_BLOCK = 1024, RM_OFFSETS_MAX_RAYS = 1u << 16;
__global__ __launch_bounds__(RM_OFFSETS_BLOCK) void k_march_train_offsets(int* __restrict__ rays, int* __restrict__ counter, uint32_t N, uint32_t M,
                                                                      uint32_t* __restrict__ bases) {
    __shared__ uint32_t lds4[RM_OFFSETS_BLOCK / 64];
    __shared__ uint32_t carry, s_dropped;
    const uint32_t start = (uint32_t)counter[0], ray_base = (uint32_t)counter[1];
    if (threadIdx.x == 0) { carry = start; s_dropped = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t dropped = 0xFFFFFFFFu;                   // point index of the first ray whose samples do not fit (at most one ray qualifies)
    for (uint32_t n0 = 0; n0 < N; n0 += RM_OFFSETS_BLOCK) {
        const uint32_t n = n0 + threadIdx.x, slot = ray_base + n;
        const bool live = n < N && slot < N;
        const uint32_t num_steps = live ? (uint32_t)rays[3ull * slot + 2] : 0u;
        uint32_t total;
        const uint32_t inc = block_inclusive_scan<RM_OFFSETS_BLOCK / 64>(num_steps, lds4, total);
        const uint32_t c = carry;
        if (live) {
            const uint32_t point_index = c + inc - num_steps;
            rays[3ull * slot] = (int)n;
            rays[3ull * slot + 1] = (int)point_index;
            if (num_steps != 0 && point_index < M && point_index + num_steps >= M) dropped = point_index;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (dropped != 0xFFFFFFFFu) s_dropped = dropped;  // (one thread at most)
    __syncthreads();
    if (threadIdx.x == 0) {
        bases[0] = start;
        bases[1] = ray_base;
        bases[2] = s_dropped != 0xFFFFFFFFu ? s_dropped : carry;      // every ray fits: the unfilled tail starts behind the last one
        counter[0] = (int)carry;
        counter[1] = (int)(ray_base + N);
    }
}

// The slots no ray of the call fills, zeroed like the torch.zeros buffers the reference's wrapper hands in (raymarching.py:205-208; a zero delta marks
// a sample that does not exist, raymarching.cu:548): the tail [bases[2], M) -- ray slots are prefix sums, so from the first ray that does not fit on
// nothing is written -- and the head [0, bases[0]) below the point base the caller's counter[0] held at entry (0 for every caller in this repository;
// a caller that hands in a pre-advanced step_counter gets zeros there as from the reference).
__global__ __launch_bounds__(RM_BLOCK) void k_march_train_zero_tail(const uint32_t* __restrict__ bases, uint32_t M, float* __restrict__ xyzs,
                                                                    float* __restrict__ dirs, float* __restrict__ deltas) {
    const uint32_t lo = bases[2] < M ? bases[2] : M;
    for (uint64_t i = 3ull * lo + blockIdx.x * RM_BLOCK + threadIdx.x; i < 3ull * M; i += (uint64_t)gridDim.x * RM_BLOCK) { xyzs[i] = 0.0f; dirs[i] = 0.0f; }
    for (uint64_t i = 2ull * lo + blockIdx.x * RM_BLOCK + threadIdx.x; i < 2ull * M; i += (uint64_t)gridDim.x * RM_BLOCK) deltas[i] = 0.0f;
    const uint32_t head = bases[0] < M ? bases[0] : M;
    for (uint64_t i = blockIdx.x * RM_BLOCK + threadIdx.x; i < 3ull * head; i += (uint64_t)gridDim.x * RM_BLOCK) { xyzs[i] = 0.0f; dirs[i] = 0.0f; }
    for (uint64_t i = blockIdx.x * RM_BLOCK + threadIdx.x; i < 2ull * head; i += (uint64_t)gridDim.x * RM_BLOCK) deltas[i] = 0.0f;
}

__global__ __launch_bounds__(RM_RAY_BLOCK) void k_march_train_write(march_args a, int* __restrict__ rays,
                                                                const uint32_t* __restrict__ block_sums,
                                                                uint32_t* bases /* [0], [1] read; [2] lowered by the first ray that does not fit */,
                                                                float* __restrict__ xyzs, float* __restrict__ dirs,
                                                                float* __restrict__ deltas, int offsets_only) {
    __shared__ uint32_t lds4[RM_RAY_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_RAY_BLOCK + threadIdx.x;
    const uint32_t ray_base = bases[1];
    const uint32_t slot = ray_base + n;
    const bool live = (n < a.N) && (slot < a.N);
    const uint32_t num_steps = live ? (uint32_t)rays[3ull * slot + 2] : 0u;
    uint32_t total;
    const uint32_t inc = block_inclusive_scan<RM_RAY_BLOCK / 64>(num_steps, lds4, total);
    if (!live) return;
    const uint32_t point_index = block_sums[blockIdx.x] + inc - num_steps;
    rays[3ull * slot] = (int)n;
    rays[3ull * slot + 1] = (int)point_index;
    // the first ray whose samples do not fit (there is at most one with point_index < M <= point_index + num_steps): the unfilled tail starts at its slot
    if (num_steps != 0 && point_index < a.M && point_index + num_steps >= a.M) bases[2] = point_index;
    if (offsets_only) return;                        // k_march_train_fill writes the samples from the recorded parameters
    if (num_steps == 0) return;
    if (point_index + num_steps >= a.M) return;      // dropped ray (reference :420)

    ngp_march_t m;
    m.setup(a.rays_o + 3ull * n, a.rays_d + 3ull * n, a.bound, a.dt_gamma, a.max_steps, a.C, a.H, a.grid);
    const float far = a.fars[n];
    m.allow_skip(a.C, a.H, far);
    float t = train_t0(m, a.nears[n], n, a.perturb), x, y, z, dt;
    float last_t = t;
    uint32_t step = 0;
    float* px = xyzs + 3ull * point_index;
    float* pd = dirs + 3ull * point_index;
    float* pl = deltas + 2ull * point_index;
    while (t < far && step < num_steps) {
        if (m.probe(t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2; step++;
        }
    }
}

// Second pass when the count pass recorded every sample's t (workspace permitting): one 64-thread workgroup per ray, one
// lane per sample.  Everything the march wrote follows from t alone with the march's own operations
// (raymarching.cu:365-367,386-391): x = clamp(o + t d), dt = clamp(t dt_gamma, dt_min, dt_max), deltas = (dt, (t + dt) - last_t)
// with last_t the previous sample's t + dt (the ray's start for the first).  No second march.
__global__ __launch_bounds__(RM_RAY_BLOCK) void k_march_train_fill(march_args a, const int* __restrict__ rays, const uint32_t* __restrict__ bases,
                                                               const float* __restrict__ tbuf, float* __restrict__ xyzs,
                                                               float* __restrict__ dirs, float* __restrict__ deltas) {
    const uint32_t n = blockIdx.x;
    const uint32_t slot = bases[1] + n;
    if (slot >= a.N) return;
    const uint32_t point_index = (uint32_t)rays[3ull * slot + 1], num_steps = (uint32_t)rays[3ull * slot + 2];
    if (num_steps == 0 || point_index + num_steps >= a.M) return;     // nothing to write / dropped ray (reference :420)
    const float* o = a.rays_o + 3ull * n;
    const float* d = a.rays_d + 3ull * n;
    const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
    const float dt_min = (2.0f * 1.7320508075688772f) / (float)a.max_steps;
    const float dt_max = ((2.0f * 1.7320508075688772f) * (float)(1 << (a.C - 1))) / (float)a.H;
    float t_start = a.nears[n];
    if (a.perturb) {
        ngp_pcg32 rng; rng.seed(42u);
        rng.advance((uint64_t)n);
        t_start = t_start + dt_min * rng.next_float();
    }
    const float* tb = tbuf + (size_t)n * a.max_steps;
    for (uint32_t k = threadIdx.x; k < num_steps; k += RM_RAY_BLOCK) {
        const float t = tb[k];
        const float dt = ngp_clampf(t * a.dt_gamma, dt_min, dt_max);
        float last_t = t_start;
        if (k > 0) { const float tp = tb[k - 1]; last_t = tp + ngp_clampf(tp * a.dt_gamma, dt_min, dt_max); }
        const uint64_t p = (uint64_t)point_index + k;
        xyzs[3 * p] = ngp_clampf(ox + t * dx, -a.bound, a.bound);
        xyzs[3 * p + 1] = ngp_clampf(oy + t * dy, -a.bound, a.bound);
        xyzs[3 * p + 2] = ngp_clampf(oz + t * dz, -a.bound, a.bound);
        dirs[3 * p] = dx; dirs[3 * p + 1] = dy; dirs[3 * p + 2] = dz;
        deltas[2 * p] = dt;
        deltas[2 * p + 1] = (t + dt) - last_t;
    }
}

// validation switch: 0 = always march one lane per ray (tests compare the two count passes)
static std::atomic<int> rm_wave_march_enabled{1};
static constexpr uint32_t RM_WAVE_PER_RAY_MAX = 1u << 17;      // composite: from this many rays on, one lane per ray fills the chip by itself
// 0: one lane per ray; 1: one wave per ray; 2: one wave per ray, every window through the serial replay (the path the one-ballot acceptance of
// rm_wave_march falls back to: selectable so that tests exercise it on whole scenes)
extern "C" int ngp_march_set_wave_per_ray(int enabled) { return rm_wave_march_enabled.exchange(enabled < 0 || enabled > 2 ? 1 : enabled, std::memory_order_relaxed); }

static size_t rm_train_ws_base(uint32_t N) { return (sizeof(uint32_t) * ((size_t)ngp_div_up(N ? N : 1, RM_RAY_BLOCK) + 4) + 255) & ~(size_t)255; }

// block sums and bases | room for a coarse occupancy map (built per call: the bitfield changes every 16 steps)
extern "C" size_t ngp_march_rays_train_workspace(uint32_t N) { return rm_train_ws_base(N) + RM_COARSE_MAX + RM_OCC_BYTES; }

// The same plus room for every sample's t (N * max_steps floats): with it the second pass does not march again.
extern "C" size_t ngp_march_rays_train_workspace_full(uint32_t N, uint32_t max_steps) {
    return ngp_march_rays_train_workspace(N) + sizeof(float) * (size_t)N * max_steps;
}

static int rm_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                               uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                               const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                               int32_t* rays, int32_t* counter, uint32_t perturb,
                               void* workspace, size_t workspace_bytes, void* stream, bool zero_tail) {
    if (N == 0) {
        if (zero_tail && M) {
            NGP_REQUIRE(xyzs && dirs && deltas, "march_rays_train: null pointer");
            const hipError_t e0 = hipMemsetAsync(xyzs, 0, 12ull * M, (hipStream_t)stream), e1 = hipMemsetAsync(dirs, 0, 12ull * M, (hipStream_t)stream),
                             e2 = hipMemsetAsync(deltas, 0, 8ull * M, (hipStream_t)stream);
            if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) return ngp_fail(NGP_ELAUNCH, "march_rays_train: hipMemsetAsync failed");
        }
        return NGP_OK;
    }
    NGP_REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter, "march_rays_train: null pointer");
    NGP_REQUIRE(C >= 1 && C <= 16 && H >= 1 && H <= 1024 && max_steps >= 1, "march_rays_train: bad C/H/max_steps");
    NGP_REQUIRE(workspace && workspace_bytes >= ngp_march_rays_train_workspace(N), "march_rays_train: workspace too small");
    if (N == 0) return NGP_OK;
    const uint32_t nblocks = ngp_div_up(N, RM_RAY_BLOCK);
    uint32_t* block_sums = (uint32_t*)workspace;
    uint32_t* bases = block_sums + nblocks;
    march_args a{rays_o, rays_d, grid, nears, fars, bound, dt_gamma, max_steps, N, C, H, M, perturb};
    hipStream_t s = (hipStream_t)stream;
    float* tbuf = nullptr;
    if (workspace_bytes >= ngp_march_rays_train_workspace_full(N, max_steps))
        tbuf = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + ngp_march_rays_train_workspace(N));
    const size_t cbytes = rm_coarse_bytes(grid, C, H);
    // (no occupied box for the training march: a training batch's rays mostly end inside the learned grid's box, the wave-per-ray count pass crosses what is
    //  left in 64-point windows, and the extra launch and per-ray slab test cost more than they save: 0.872 against 0.846 ms per step, measured)
    if (cbytes) rm_attach_coarse(a, reinterpret_cast<unsigned char*>(workspace) + rm_train_ws_base(N), s, false);
    // one wave per ray when the lattice is closed-form (constant step) and the sample parameters can be recorded; else one lane per ray
    const int wave_mode = rm_wave_march_enabled.load(std::memory_order_relaxed);
    const bool wave_per_ray = tbuf && cbytes && dt_gamma == 0.0f && C <= 4 && wave_mode != 0;
    if (wave_per_ray && N <= RM_OFFSETS_MAX_RAYS) {
        hipLaunchKernelGGL(k_march_train_count_wave, dim3(N), dim3(RM_RAY_BLOCK), cbytes, s, a, rays, counter, tbuf, wave_mode == 2 ? 1u : 0u);
        hipLaunchKernelGGL(k_march_train_offsets, dim3(1), dim3(RM_OFFSETS_BLOCK), 0, s, rays, counter, N, M, bases);
    } else {
        if (wave_per_ray) {
            hipLaunchKernelGGL(k_march_train_count_wave, dim3(N), dim3(RM_RAY_BLOCK), cbytes, s, a, rays, counter, tbuf, wave_mode == 2 ? 1u : 0u);
            hipLaunchKernelGGL(k_march_train_sum64, dim3(ngp_div_up(nblocks, RM_BLOCK)), dim3(RM_BLOCK), 0, s, rays, counter, N, nblocks, block_sums);
        } else {
            hipLaunchKernelGGL(k_march_train_count, dim3(nblocks), dim3(RM_RAY_BLOCK), cbytes, s, a, rays, counter, block_sums, tbuf);
        }
        hipLaunchKernelGGL(k_march_train_scan, dim3(1), dim3(RM_BLOCK), 0, s, block_sums, nblocks, counter, N, bases);
        hipLaunchKernelGGL(k_march_train_write, dim3(nblocks), dim3(RM_RAY_BLOCK), 0, s, a, rays, block_sums, bases, xyzs, dirs, deltas, tbuf ? 1 : 0);
    }
    if (tbuf) hipLaunchKernelGGL(k_march_train_fill, dim3(N), dim3(RM_RAY_BLOCK), 0, s, a, rays, bases, tbuf, xyzs, dirs, deltas);
    if (zero_tail && M) {
        const uint32_t zb = ngp_div_up(3ull * M, RM_BLOCK * 8ull);
        hipLaunchKernelGGL(k_march_train_zero_tail, dim3(zb < 2048u ? (zb ? zb : 1u) : 2048u), dim3(RM_BLOCK), 0, s, bases, M, xyzs, dirs, deltas);
    }
    NGP_CHECK_LAUNCH("march_rays_train");
    return NGP_OK;
}

extern "C" int ngp_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                    const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                    int32_t* rays, int32_t* counter, uint32_t perturb,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return rm_march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, perturb, workspace,
                               workspace_bytes, stream, false);
}

// The same, for buffers that were NOT zeroed: the slots no ray fills are zeroed by the call (one small launch instead of three whole-buffer fills).
extern "C" int ngp_march_rays_train_filled(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                           uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                           const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                           int32_t* rays, int32_t* counter, uint32_t perturb,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    return rm_march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, perturb, workspace,
                               workspace_bytes, stream, true);
}

// ---------------------------------------------------------------------------
// training composite
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(RM_RAY_BLOCK) void k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                  const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                  uint32_t M, uint32_t N, float* __restrict__ weights_sum,
                                                                  float* __restrict__ depth, float* __restrict__ image) {
    // reference: raymarching.cu:506-582
    const uint32_t n = blockIdx.x * RM_RAY_BLOCK + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps >= M) {
        weights_sum[index] = 0; depth[index] = 0;
        image[3ull * index] = 0; image[3ull * index + 1] = 0; image[3ull * index + 2] = 0;
        return;
    }
    const float* s = sigmas + offset;
    const float* c = rgbs + 3ull * offset;
    const float* dl = deltas + 2ull * offset;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    // One ray per lane and only a few thousand rays per step: the loop is bound by the latency of its six loads per sample.
    // Samples are therefore fetched RM_CHUNK at a time (all loads of a chunk in flight together) and then composited in the
    // reference's order with the reference's arithmetic: same results, a fraction of the exposed latency.
    bool done = false;
    for (uint32_t k0 = 0; k0 < num_steps && !done; k0 += RM_CHUNK) {
        float sv[RM_CHUNK], d0[RM_CHUNK], d1[RM_CHUNK], c0[RM_CHUNK], c1[RM_CHUNK], c2[RM_CHUNK];
        #pragma unroll
        for (uint32_t j = 0; j < RM_CHUNK; j++) {
            const uint32_t k = (k0 + j < num_steps) ? k0 + j : num_steps - 1;     // clamp: the tail re-reads the last sample
            sv[j] = s[k]; d0[j] = dl[2 * k]; d1[j] = dl[2 * k + 1];
            c0[j] = c[3 * k]; c1[j] = c[3 * k + 1]; c2[j] = c[3 * k + 2];
        }
        #pragma unroll
        for (uint32_t j = 0; j < RM_CHUNK; j++) {
            if (done || k0 + j >= num_steps) { done = true; continue; }
            const float alpha = 1.0f - ngp_expf(-sv[j] * d0[j]);
            const float w = alpha * T;
            r += w * c0[j]; g += w * c1[j]; b += w * c2[j];
            t += d1[j];
            d += w * t;
            ws += w;
            T *= 1.0f - alpha;
            if (T < 1e-4f) done = true;
        }
    }
    weights_sum[index] = ws; depth[index] = d;
    image[3ull * index] = r; image[3ull * index + 1] = g; image[3ull * index + 2] = b;
}

__global__ __launch_bounds__(RM_RAY_BLOCK) void k_composite_train_bwd(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                                  const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                  const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                  const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                  uint32_t M, uint32_t N, float* __restrict__ grad_sigmas,
                                                                  float* __restrict__ grad_rgbs) {
    // reference: raymarching.cu:607-688
    const uint32_t n = blockIdx.x * RM_RAY_BLOCK + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps >= M) return;
    const float gws = grad_weights_sum[index];
    const float g0 = grad_image[3ull * index], g1 = grad_image[3ull * index + 1], g2 = grad_image[3ull * index + 2];
    const float rf = image[3ull * index], gf = image[3ull * index + 1], bf = image[3ull * index + 2], wsf = weights_sum[index];
    const float* s = sigmas + offset;
    const float* c = rgbs + 3ull * offset;
    const float* dl = deltas + 2ull * offset;
    float* gs = grad_sigmas + offset;
    float* gc = grad_rgbs + 3ull * offset;
    float T = 1.0f, r = 0, g = 0, b = 0;
    bool done = false;                                 // chunked like the forward pass (see there)
    for (uint32_t k0 = 0; k0 < num_steps && !done; k0 += RM_CHUNK) {
        float sv[RM_CHUNK], d0[RM_CHUNK], c0[RM_CHUNK], c1[RM_CHUNK], c2[RM_CHUNK];
        #pragma unroll
        for (uint32_t j = 0; j < RM_CHUNK; j++) {
            const uint32_t k = (k0 + j < num_steps) ? k0 + j : num_steps - 1;
            sv[j] = s[k]; d0[j] = dl[2 * k];
            c0[j] = c[3 * k]; c1[j] = c[3 * k + 1]; c2[j] = c[3 * k + 2];
        }
        #pragma unroll
        for (uint32_t j = 0; j < RM_CHUNK; j++) {
            if (done || k0 + j >= num_steps) { done = true; continue; }
            const uint32_t k = k0 + j;
            const float alpha = 1.0f - ngp_expf(-sv[j] * d0[j]);
            const float w = alpha * T;
            r += w * c0[j]; g += w * c1[j]; b += w * c2[j];
            T *= 1.0f - alpha;
            if (T < 1e-4f) { done = true; continue; }
            gc[3 * k] = g0 * w; gc[3 * k + 1] = g1 * w; gc[3 * k + 2] = g2 * w;
            gs[k] = d0[j] * (g0 * (T * c0[j] - (rf - r)) +
                             g1 * (T * c1[j] - (gf - g)) +
                             g2 * (T * c2[j] - (bf - b)) +
                             gws * (1.0f - wsf));
        }
    }
}

// One WAVE per ray (training batches are a few thousand rays: one lane per ray leaves 94 % of the SIMDs idle and every lane waits for its own
// loads).  The 64 lanes fetch 64 consecutive samples of the ray with coalesced loads and evaluate alpha = 1 - exp(-sigma delta) together;
// the recurrences T, r, g, b, ... then run in the reference's order with the reference's arithmetic, every lane computing the same chain on
// values broadcast with v_readlane: the results are the lane-per-ray kernel's, bit for bit (tests/test_gpu_raymarching.py compares them).
static constexpr uint32_t RM_WAVES_PER_BLOCK = 4;
__device__ __forceinline__ float rm_bcast(float v, uint32_t lane_uniform) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)lane_uniform));
}

__global__ __launch_bounds__(64 * RM_WAVES_PER_BLOCK) void k_composite_train_fwd_wave(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                                    const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                                    uint32_t M, uint32_t N, float* __restrict__ weights_sum,
                                                                                    float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * RM_WAVES_PER_BLOCK + (threadIdx.x >> 6)));
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps >= M) {
        if (lane == 0) {
            weights_sum[index] = 0; depth[index] = 0;
            image[3ull * index] = 0; image[3ull * index + 1] = 0; image[3ull * index + 2] = 0;
        }
        return;
    }
    const float* s = sigmas + offset;
    const float* c = rgbs + 3ull * offset;
    const float* dl = deltas + 2ull * offset;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    bool done = false;
    for (uint32_t k0 = 0; k0 < num_steps && !done; k0 += 64) {
        const uint32_t k = (k0 + lane < num_steps) ? k0 + lane : num_steps - 1;
        const float sv = s[k], d0 = dl[2 * k], d1 = dl[2 * k + 1], c0 = c[3 * k], c1 = c[3 * k + 1], c2 = c[3 * k + 2];
        const float alpha = 1.0f - ngp_expf(-sv * d0);
        const uint32_t cnt = num_steps - k0 < 64u ? num_steps - k0 : 64u;
        for (uint32_t j = 0; j < cnt; j++) {
            const float a = rm_bcast(alpha, j);
            const float w = a * T;
            r += w * rm_bcast(c0, j); g += w * rm_bcast(c1, j); b += w * rm_bcast(c2, j);
            t += rm_bcast(d1, j);
            d += w * t;
            ws += w;
            T *= 1.0f - a;
            if (T < 1e-4f) { done = true; break; }
        }
    }
    if (lane == 0) {
        weights_sum[index] = ws; depth[index] = d;
        image[3ull * index] = r; image[3ull * index + 1] = g; image[3ull * index + 2] = b;
    }
}

__global__ __launch_bounds__(64 * RM_WAVES_PER_BLOCK) void k_composite_train_bwd_wave(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                                                    const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                                    const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                                    const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                                    uint32_t M, uint32_t N, float* __restrict__ grad_sigmas,
                                                                                    float* __restrict__ grad_rgbs) {
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * RM_WAVES_PER_BLOCK + (threadIdx.x >> 6)));
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps >= M) return;
    const float gws = grad_weights_sum[index];
    const float g0 = grad_image[3ull * index], g1 = grad_image[3ull * index + 1], g2 = grad_image[3ull * index + 2];
    const float rf = image[3ull * index], gf = image[3ull * index + 1], bf = image[3ull * index + 2], wsf = weights_sum[index];
    const float* s = sigmas + offset;
    const float* c = rgbs + 3ull * offset;
    const float* dl = deltas + 2ull * offset;
    float* gs = grad_sigmas + offset;
    float* gc = grad_rgbs + 3ull * offset;
    float T = 1.0f, r = 0, g = 0, b = 0;
    bool done = false;
    for (uint32_t k0 = 0; k0 < num_steps && !done; k0 += 64) {
        const uint32_t k = (k0 + lane < num_steps) ? k0 + lane : num_steps - 1;
        const float sv = s[k], d0 = dl[2 * k], c0 = c[3 * k], c1 = c[3 * k + 1], c2 = c[3 * k + 2];
        const float alpha = 1.0f - ngp_expf(-sv * d0);
        const uint32_t cnt = num_steps - k0 < 64u ? num_steps - k0 : 64u;
        float mw = 0, mT = 0, mr = 0, mg = 0, mb = 0;          // the chain's values after this lane's sample
        bool mine = false;                                     // (the sample on which T drops below 1e-4 gets no gradient, like every later one)
        for (uint32_t j = 0; j < cnt; j++) {
            const float a = rm_bcast(alpha, j);
            const float w = a * T;
            r += w * rm_bcast(c0, j); g += w * rm_bcast(c1, j); b += w * rm_bcast(c2, j);
            T *= 1.0f - a;
            if (T < 1e-4f) { done = true; break; }
            if (lane == j) { mw = w; mT = T; mr = r; mg = g; mb = b; mine = true; }
        }
        if (mine) {
            gc[3 * k] = g0 * mw; gc[3 * k + 1] = g1 * mw; gc[3 * k + 2] = g2 * mw;
            gs[k] = d0 * (g0 * (mT * c0 - (rf - mr)) +
                          g1 * (mT * c1 - (gf - mg)) +
                          g2 * (mT * c2 - (bf - mb)) +
                          gws * (1.0f - wsf));
        }
    }
}

// ---------------------------------------------------------------------------
// The wave-per-ray compositors again, with the serial part cut to what IS serial (round 4).
//
// Above, every lane of the wave executes the whole recurrence on broadcast values: ~20 wave-instructions per sample, and a training batch's launch lasts as
// long as its longest rays' chains (up to 1,024 samples on the early grid) times the four waves that share a SIMD.  But only two things are serial, and
// bit-exactness only asks that each quantity be formed by the SAME operations in the SAME order as kernel_composite_rays_train_* (raymarching.cu:506-688):
//   * T_j = ((T_in (1 - a_0)) (1 - a_1)) ... : one multiply per sample.  Lane j gets its T_j from a 64-step chain in which step j multiplies the lanes > j
//     by (1 - a_j) -- the exec mask shifts left by one lane per step (1 SALU), the factor comes from v_readlane: 2 VALU per sample (rm_chain64; t, the running
//     sum of deltas[.][1], rides the same loop for 2 more);
//   * the sums ws, depth, r, g, b are sequential ADDITIONS of per-sample terms w_j, w_j t_j, w_j c_j that the 64 lanes form in parallel: each sum is one
//     lane's chain of 64 adds over a row the wave parks in LDS (five lanes work at once); the backward needs the running sums r_j, g_j, b_j themselves:
//     three lanes form them in place and hand them back through LDS.
// Terms behind the sample on which T falls below 1e-4 are +0 (x + 0 = x exactly), so the early exit needs no branch inside a chunk.  ~6 wave-instructions
// per sample instead of ~20; same bits (tests/test_gpu_raymarching.py compares every variant with the oracle).
// ---------------------------------------------------------------------------
#define RM_REP8(M, b) M(b##0) M(b##1) M(b##2) M(b##3) M(b##4) M(b##5) M(b##6) M(b##7)
#define RM_REP64(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) RM_REP8(M, 1) M(18) M(19) RM_REP8(M, 2) M(28) M(29) RM_REP8(M, 3) M(38) M(39) \
                    RM_REP8(M, 4) M(48) M(49) RM_REP8(M, 5) M(58) M(59) M(60) M(61) M(62) M(63)
#define RM_CHAIN_T_STEP(j) "v_readlane_b32 s90, %[om], " #j "\n s_lshl_b64 exec, exec, 1\n v_mul_f32 %[T], s90, %[T]\n"
#define RM_CHAIN_TT_STEP(j) "v_readlane_b32 s90, %[om], " #j "\n v_readlane_b32 s91, %[d1], " #j "\n v_add_f32 %[t], s91, %[t]\n s_lshl_b64 exec, exec, 1\n v_mul_f32 %[T], s90, %[T]\n"

// in: T, t = the chains' values before this chunk (the same in every lane), om = 1 - alpha and d1 of this lane's sample (1 and 0 on lanes without one).
// out: lane L holds T before its sample = ((T om_0) om_1) ... om_{L-1}, and t including its sample = ((t + d1_0) + d1_1) ... + d1_L.
// Call with all 64 lanes active.  (Inline assembly is opaque to the compiler's hazard recogniser: the s_nop covers "VALU writes a VGPR, v_readlane
// reads it" for whatever instruction precedes the block; inside it no v_readlane source is written and SALU writes of exec need no wait before a VALU.)
__device__ __forceinline__ void rm_chain64(float om, float d1, float& T, float& t) {
    asm volatile("s_mov_b64 s[92:93], exec\n s_nop 1\n" RM_REP64(RM_CHAIN_TT_STEP) "s_mov_b64 exec, s[92:93]\n"
                 : [T] "+v"(T), [t] "+v"(t) : [om] "v"(om), [d1] "v"(d1) : "s90", "s91", "s92", "s93", "scc");
}
__device__ __forceinline__ void rm_chain64(float om, float& T) {
    asm volatile("s_mov_b64 s[92:93], exec\n s_nop 1\n" RM_REP64(RM_CHAIN_T_STEP) "s_mov_b64 exec, s[92:93]\n"
                 : [T] "+v"(T) : [om] "v"(om) : "s90", "s92", "s93", "scc");
}
__device__ __forceinline__ void rm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
static constexpr uint32_t RM_SCAN_ROW = 68;                              // floats per LDS row: 16-byte aligned, rows on different banks

__global__ __launch_bounds__(64 * RM_WAVES_PER_BLOCK) void k_composite_train_fwd_scan(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                                    const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                                    uint32_t M, uint32_t N, float* __restrict__ weights_sum,
                                                                                    float* __restrict__ depth, float* __restrict__ image) {
    __shared__ __attribute__((aligned(16))) float lds_rows[RM_WAVES_PER_BLOCK][5][RM_SCAN_ROW];
    const uint32_t wib = threadIdx.x >> 6;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * RM_WAVES_PER_BLOCK + wib));
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps >= M) {
        if (lane == 0) {
            weights_sum[index] = 0; depth[index] = 0;
            image[3ull * index] = 0; image[3ull * index + 1] = 0; image[3ull * index + 2] = 0;
        }
        return;
    }
    const float* s = sigmas + offset;
    const float* c = rgbs + 3ull * offset;
    const float* dl = deltas + 2ull * offset;
    float (*rows)[RM_SCAN_ROW] = lds_rows[wib];
    const uint32_t q = lane < 5u ? lane : 0u;                            // lanes 0..4 own the sums ws, depth, r, g, b
    float T = 1.0f, t = 0.0f, acc = 0.0f;
    bool done = false;
    for (uint32_t k0 = 0; k0 < num_steps && !done; k0 += 64) {
        const bool valid = k0 + lane < num_steps;
        const uint32_t k = valid ? k0 + lane : num_steps - 1;
        const float sv = s[k], d0 = dl[2 * k], d1 = dl[2 * k + 1], c0 = c[3 * k], c1 = c[3 * k + 1], c2 = c[3 * k + 2];
        const float alpha = 1.0f - ngp_expf(-sv * d0);
        const float om = valid ? 1.0f - alpha : 1.0f;
        float Tl = T, tl = t;
        rm_chain64(om, valid ? d1 : 0.0f, Tl, tl);
        const float Tafter = Tl * om;                                    // T after this lane's sample
        const unsigned long long brk = __ballot(valid && Tafter < 1e-4f);
        const bool inc = valid && (brk == 0ull || lane <= (uint32_t)__builtin_ctzll(brk));   // the sample on which T drops below 1e-4 still counts
        const float w = alpha * Tl;
        rows[0][lane] = inc ? w : 0.0f;
        rows[1][lane] = inc ? w * tl : 0.0f;
        rows[2][lane] = inc ? w * c0 : 0.0f;
        rows[3][lane] = inc ? w * c1 : 0.0f;
        rows[4][lane] = inc ? w * c2 : 0.0f;
        rm_wave_sync();
        {
            const float4* row = reinterpret_cast<const float4*>(rows[q]);
            float x[64];
            #pragma unroll
            for (int i = 0; i < 16; i++) { const float4 v = row[i]; x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w; }
            #pragma unroll
            for (int j = 0; j < 64; j++) acc += x[j];
        }
        rm_wave_sync();                                                  // the rows are rewritten by the next chunk
        done = brk != 0ull;
        T = rm_bcast(Tafter, 63);
        t = rm_bcast(tl, 63);
    }
    const float ws = rm_bcast(acc, 0), d = rm_bcast(acc, 1), r = rm_bcast(acc, 2), g = rm_bcast(acc, 3), b = rm_bcast(acc, 4);
    if (lane == 0) {
        weights_sum[index] = ws; depth[index] = d;
        image[3ull * index] = r; image[3ull * index + 1] = g; image[3ull * index + 2] = b;
    }
}

__global__ __launch_bounds__(64 * RM_WAVES_PER_BLOCK) void k_composite_train_bwd_scan(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                                                    const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                                    const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                                    const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                                    uint32_t M, uint32_t N, float* __restrict__ grad_sigmas,
                                                                                    float* __restrict__ grad_rgbs) {
    __shared__ __attribute__((aligned(16))) float lds_rows[RM_WAVES_PER_BLOCK][3][RM_SCAN_ROW];
    const uint32_t wib = threadIdx.x >> 6;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * RM_WAVES_PER_BLOCK + wib));
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps >= M) return;
    const float gws = grad_weights_sum[index];
    const float g0 = grad_image[3ull * index], g1 = grad_image[3ull * index + 1], g2 = grad_image[3ull * index + 2];
    const float rf = image[3ull * index], gf = image[3ull * index + 1], bf = image[3ull * index + 2], wsf = weights_sum[index];
    const float* s = sigmas + offset;
    const float* c = rgbs + 3ull * offset;
    const float* dl = deltas + 2ull * offset;
    float* gs = grad_sigmas + offset;
    float* gc = grad_rgbs + 3ull * offset;
    float (*rows)[RM_SCAN_ROW] = lds_rows[wib];
    const uint32_t q = lane < 3u ? lane : 0u;                            // lanes 0..2 own the running sums r, g, b
    float T = 1.0f, acc = 0.0f;
    bool done = false;
    for (uint32_t k0 = 0; k0 < num_steps && !done; k0 += 64) {
        const bool valid = k0 + lane < num_steps;
        const uint32_t k = valid ? k0 + lane : num_steps - 1;
        const float sv = s[k], d0 = dl[2 * k], c0 = c[3 * k], c1 = c[3 * k + 1], c2 = c[3 * k + 2];
        const float alpha = 1.0f - ngp_expf(-sv * d0);
        const float om = valid ? 1.0f - alpha : 1.0f;
        float Tl = T;
        rm_chain64(om, Tl);
        const float mT = Tl * om;                                        // T after this lane's sample
        const unsigned long long brk = __ballot(valid && mT < 1e-4f);
        const uint32_t jstar = brk ? (uint32_t)__builtin_ctzll(brk) : 64u;
        const bool inc = valid && lane <= jstar;                        // the running sums include the sample on which T drops below 1e-4 ...
        const bool mine = valid && lane < jstar;                        // ... but it gets no gradient, like every later one (raymarching.cu:661)
        const float mw = alpha * Tl;
        rows[0][lane] = inc ? mw * c0 : 0.0f;
        rows[1][lane] = inc ? mw * c1 : 0.0f;
        rows[2][lane] = inc ? mw * c2 : 0.0f;
        rm_wave_sync();
        {
            float4* row = reinterpret_cast<float4*>(rows[q]);
            float x[64];
            #pragma unroll
            for (int i = 0; i < 16; i++) { const float4 v = row[i]; x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w; }
            x[0] = acc + x[0];
            #pragma unroll
            for (int j = 1; j < 64; j++) x[j] = x[j - 1] + x[j];
            acc = x[63];
            rm_wave_sync();                                              // every lane has read its row before lanes 0..2 overwrite theirs
            if (lane < 3u) {
                #pragma unroll
                for (int i = 0; i < 16; i++) row[i] = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
            }
        }
        rm_wave_sync();
        const float mr = rows[0][lane], mg = rows[1][lane], mb = rows[2][lane];
        if (mine) {
            gc[3 * k] = g0 * mw; gc[3 * k + 1] = g1 * mw; gc[3 * k + 2] = g2 * mw;
            gs[k] = d0 * (g0 * (mT * c0 - (rf - mr)) +
                          g1 * (mT * c1 - (gf - mg)) +
                          g2 * (mT * c2 - (bf - mb)) +
                          gws * (1.0f - wsf));
        }
        rm_wave_sync();                                                  // the rows are rewritten by the next chunk
        done = brk != 0ull;
        T = rm_bcast(mT, 63);
    }
}

// process-wide switch between the two wave-per-ray compositors (tests compare both with the oracle; A/B timing)
static std::atomic<int> rm_scan_composite_enabled{1};
extern "C" int ngp_composite_set_scan(int enabled) { return rm_scan_composite_enabled.exchange(enabled ? 1 : 0, std::memory_order_relaxed); }

extern "C" int ngp_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                                uint32_t M, uint32_t N, float* weights_sum, float* depth, float* image, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(rays && weights_sum && depth && image, "composite_rays_train_forward: null pointer");
    NGP_REQUIRE(M == 0 || (sigmas && rgbs && deltas), "composite_rays_train_forward: null sample pointer");
    if (N == 0) return NGP_OK;
    if (N < RM_WAVE_PER_RAY_MAX && rm_wave_march_enabled.load(std::memory_order_relaxed) && rm_scan_composite_enabled.load(std::memory_order_relaxed))
        hipLaunchKernelGGL(k_composite_train_fwd_scan, dim3(ngp_div_up(N, RM_WAVES_PER_BLOCK)), dim3(64 * RM_WAVES_PER_BLOCK), 0, (hipStream_t)stream,
                           sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image);
    else if (N < RM_WAVE_PER_RAY_MAX && rm_wave_march_enabled.load(std::memory_order_relaxed))
        hipLaunchKernelGGL(k_composite_train_fwd_wave, dim3(ngp_div_up(N, RM_WAVES_PER_BLOCK)), dim3(64 * RM_WAVES_PER_BLOCK), 0, (hipStream_t)stream,
                           sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image);
    else
        hipLaunchKernelGGL(k_composite_train_fwd, dim3(ngp_div_up(N, RM_RAY_BLOCK)), dim3(RM_RAY_BLOCK), 0, (hipStream_t)stream,
                           sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image);
    NGP_CHECK_LAUNCH("composite_rays_train_forward");
    return NGP_OK;
}

extern "C" int ngp_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                                 const float* rgbs, const float* deltas, const int32_t* rays,
                                                 const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                                 float* grad_sigmas, float* grad_rgbs, void* stream) {
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(grad_weights_sum && grad_image && rays && weights_sum && image, "composite_rays_train_backward: null pointer");
    NGP_REQUIRE(M == 0 || (sigmas && rgbs && deltas && grad_sigmas && grad_rgbs), "composite_rays_train_backward: null sample pointer");
    if (N == 0) return NGP_OK;
    if (N < RM_WAVE_PER_RAY_MAX && rm_wave_march_enabled.load(std::memory_order_relaxed) && rm_scan_composite_enabled.load(std::memory_order_relaxed))
        hipLaunchKernelGGL(k_composite_train_bwd_scan, dim3(ngp_div_up(N, RM_WAVES_PER_BLOCK)), dim3(64 * RM_WAVES_PER_BLOCK), 0, (hipStream_t)stream,
                           grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs);
    else if (N < RM_WAVE_PER_RAY_MAX && rm_wave_march_enabled.load(std::memory_order_relaxed))
        hipLaunchKernelGGL(k_composite_train_bwd_wave, dim3(ngp_div_up(N, RM_WAVES_PER_BLOCK)), dim3(64 * RM_WAVES_PER_BLOCK), 0, (hipStream_t)stream,
                           grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs);
    else
        hipLaunchKernelGGL(k_composite_train_bwd, dim3(ngp_div_up(N, RM_RAY_BLOCK)), dim3(RM_RAY_BLOCK), 0, (hipStream_t)stream,
                           grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs);
    NGP_CHECK_LAUNCH("composite_rays_train_backward");
    return NGP_OK;
}

// ---------------------------------------------------------------------------
// inference march / composite
// ---------------------------------------------------------------------------

// FILL: the kernel also writes the zeros the reference gets from torch.zeros (raymarching.py:327-329): the slots a ray does not
// reach (zero delta = terminated, raymarching.cu:867) and the alignment rows [n_alive * n_step, M); the buffers need no pre-zeroing.
// BLOCK: 64 threads when few rays are alive (their waves spread over 4x as many CUs), 256 when the launch fills the chip anyway (a quarter of
// the workgroups stage the coarse map: 80 MB -> 20 MB of L2 reads per 640 k-ray launch)
template <bool FILL, uint32_t BLOCK>
__global__ __launch_bounds__(BLOCK) void k_march_rays(uint32_t n_alive, uint32_t n_step, const int* __restrict__ rays_alive,
                                                         const float* __restrict__ rays_t, march_args a,
                                                         float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas) {
    // reference: raymarching.cu:707-814
    const uint32_t* lds_coarse = rm_stage_coarse<BLOCK>(a);
    const float* occ = a.occ;
    const uint32_t n = blockIdx.x * BLOCK + threadIdx.x;
    if (FILL) {
        // a.M rows in all; rows past the last ray's slots are spread over the launch's lanes
        const uint64_t used = (uint64_t)n_alive * n_step;
        for (uint64_t r = used + n; r < a.M; r += (uint64_t)gridDim.x * BLOCK) {
            xyzs[3 * r] = 0.f; xyzs[3 * r + 1] = 0.f; xyzs[3 * r + 2] = 0.f;
            dirs[3 * r] = 0.f; dirs[3 * r + 1] = 0.f; dirs[3 * r + 2] = 0.f;
            deltas[2 * r] = 0.f; deltas[2 * r + 1] = 0.f;
        }
    }
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    ngp_march_t m;
    m.setup(a.rays_o + 3ll * index, a.rays_d + 3ll * index, a.bound, a.dt_gamma, a.max_steps, a.C, a.H, a.grid);
    m.use_coarse(lds_coarse, a.coarse_words);
    float* px = xyzs + 3ull * n * n_step;
    float* pd = dirs + 3ull * n * n_step;
    float* pl = deltas + 2ull * n * n_step;
    float t = rays_t[index];
    // (a ray that has left the box of everything occupied has no sample left: it ends here, not after walking to its far through empty cells -- the walk that
    //  set the duration of every late call of the inference loop, profiles/HISTORY.md 4.4)
    const float far = rm_march_far(occ, a.rays_o + 3ll * index, a.rays_d + 3ll * index, a.fars[index]);
    m.allow_skip(a.C, a.H, occ ? far + 4.0f * a.bound : far);
    if (a.perturb) {                                  // seed = perturb, jump = alive SLOT (reference :752-755,819)
        ngp_pcg32 rng; rng.seed((uint64_t)a.perturb);
        rng.advance((uint64_t)n);
        t += m.dt_min * rng.next_float();
    }
    float last_t = t, x, y, z, dt;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        if (m.probe(t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2; step++;
        }
    }
    if (FILL) {
        for (; step < n_step; step++) {
            px[0] = 0.f; px[1] = 0.f; px[2] = 0.f;
            pd[0] = 0.f; pd[1] = 0.f; pd[2] = 0.f;
            pl[0] = 0.f; pl[1] = 0.f;
            px += 3; pd += 3; pl += 2;
        }
    }
}

// RGB_T: float (the reference's kernel) or _Float16 -- the colours of a field under autocast, which the reference's wrapper widens to float32 first
// (raymarching.py:343 custom_fwd(cast_inputs=float32): an elementwise launch per iteration of the inference loop); widening in the load is the same value
template <class RGB_T>
__global__ __launch_bounds__(RM_RAY_BLOCK) void k_composite_rays(uint32_t n_alive, uint32_t n_step, int* __restrict__ rays_alive,
                                                             float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                                             const RGB_T* __restrict__ rgbs, const float* __restrict__ deltas,
                                                             float* __restrict__ weights_sum, float* __restrict__ depth,
                                                             float* __restrict__ image) {
    // reference: raymarching.cu:829-913
    const uint32_t n = blockIdx.x * RM_RAY_BLOCK + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float* s = sigmas + (uint64_t)n * n_step;
    const RGB_T* c = rgbs + 3ull * n * n_step;
    const float* dl = deltas + 2ull * n * n_step;
    float t = rays_t[index];
    float ws = weights_sum[index], d = depth[index];
    float r = image[3ll * index], g = image[3ll * index + 1], b = image[3ll * index + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;
        const float alpha = 1.0f - ngp_expf(-s[0] * dl[0]);
        const float T = 1 - ws;
        const float w = alpha * T;
        ws += w;
        t += dl[1];
        d += w * t;
        r += w * (float)c[0]; g += w * (float)c[1]; b += w * (float)c[2];
        if ((double)T < 1e-4) break;                  // double literal in the reference (:890)
        s++; c += 3; dl += 2; step++;
    }
    if (step < n_step) rays_alive[n] = -1;
    else rays_t[index] = t;
    weights_sum[index] = ws; depth[index] = d;
    image[3ll * index] = r; image[3ll * index + 1] = g; image[3ll * index + 2] = b;
}

static int march_rays_launch(bool fill, uint32_t M, uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                             const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                             uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                             float* xyzs, float* dirs, float* deltas, uint32_t perturb, void* workspace, size_t workspace_bytes, void* stream) {
    if ((n_alive == 0 || n_step == 0) && !(fill && M)) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas, "march_rays: null pointer");
    NGP_REQUIRE(C >= 1 && C <= 16 && H >= 1 && H <= 1024 && max_steps >= 1, "march_rays: bad C/H/max_steps");
    NGP_REQUIRE(!fill || (uint64_t)M >= (uint64_t)n_alive * n_step, "march_rays: M is smaller than n_alive * n_step");
    march_args a{rays_o, rays_d, grid, nears, fars, bound, dt_gamma, max_steps, 0u, C, H, M, perturb};
    // the coarse map pays when rays cross empty space; it is rebuilt on every call (2 us: the bitfield may have changed, and a cache
    // keyed on a pointer could go stale silently)
    const size_t cbytes = (workspace && n_alive) ? rm_coarse_bytes(grid, C, H) : 0;
    const size_t lds = (cbytes && workspace_bytes >= cbytes + RM_OCC_BYTES) ? cbytes : 0;
    if (lds) rm_attach_coarse(a, workspace, (hipStream_t)stream, true);
    const bool big = n_alive >= 65536u;          // (measured: the frame takes the same time with 64-thread workgroups throughout; 256 stage the map 4x less often)
    const uint32_t bs = big ? RM_BLOCK : RM_RAY_BLOCK;
    const dim3 grid_dim(ngp_div_up(n_alive ? n_alive : 1, bs)), block(bs);
    hipStream_t st = (hipStream_t)stream;
    if (fill && big) hipLaunchKernelGGL((k_march_rays<true, RM_BLOCK>), grid_dim, block, lds, st, n_alive, n_step, rays_alive, rays_t, a, xyzs, dirs, deltas);
    else if (fill) hipLaunchKernelGGL((k_march_rays<true, RM_RAY_BLOCK>), grid_dim, block, lds, st, n_alive, n_step, rays_alive, rays_t, a, xyzs, dirs, deltas);
    else if (big) hipLaunchKernelGGL((k_march_rays<false, RM_BLOCK>), grid_dim, block, lds, st, n_alive, n_step, rays_alive, rays_t, a, xyzs, dirs, deltas);
    else hipLaunchKernelGGL((k_march_rays<false, RM_RAY_BLOCK>), grid_dim, block, lds, st, n_alive, n_step, rays_alive, rays_t, a, xyzs, dirs, deltas);
    NGP_CHECK_LAUNCH("march_rays");
    return NGP_OK;
}

extern "C" int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                              const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                              uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                              float* xyzs, float* dirs, float* deltas, uint32_t perturb, void* stream) {
    return march_rays_launch(false, 0, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars,
                             xyzs, dirs, deltas, perturb, nullptr, 0, stream);
}

extern "C" int ngp_march_rays_fill(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                                   float* xyzs, float* dirs, float* deltas, uint32_t M, uint32_t perturb,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return march_rays_launch(true, M, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars,
                             xyzs, dirs, deltas, perturb, workspace, workspace_bytes, stream);
}

extern "C" size_t ngp_march_rays_workspace(uint32_t C, uint32_t H) { (void)C; (void)H; return RM_COARSE_MAX + RM_OCC_BYTES; }

extern "C" int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t, const float* sigmas,
                                  const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image, void* stream) {
    if (n_alive == 0) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, "composite_rays: null pointer");
    hipLaunchKernelGGL(k_composite_rays<float>, dim3(ngp_div_up(n_alive, RM_RAY_BLOCK)), dim3(RM_RAY_BLOCK), 0, (hipStream_t)stream,
                       n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    NGP_CHECK_LAUNCH("composite_rays");
    return NGP_OK;
}

// rgbs as halves [n_alive * n_step, 3] (what a field under autocast returns): the same values without the widening copy
extern "C" int ngp_composite_rays_half(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t, const float* sigmas,
                                       const void* rgbs_half, const float* deltas, float* weights_sum, float* depth, float* image, void* stream) {
    if (n_alive == 0) return NGP_OK;
    NGP_REQUIRE(rays_alive && rays_t && sigmas && rgbs_half && deltas && weights_sum && depth && image, "composite_rays_half: null pointer");
    hipLaunchKernelGGL(k_composite_rays<_Float16>, dim3(ngp_div_up(n_alive, RM_RAY_BLOCK)), dim3(RM_RAY_BLOCK), 0, (hipStream_t)stream,
                       n_alive, n_step, rays_alive, rays_t, sigmas, (const _Float16*)rgbs_half, deltas, weights_sum, depth, image);
    NGP_CHECK_LAUNCH("composite_rays_half");
    return NGP_OK;
}

// ---------------------------------------------------------------------------
// stable compaction of the alive list (rays_alive[rays_alive >= 0], nerf/renderer.py:365) without a host sync:
// wave ballot + popcount for the in-wave rank, block scan for the block rank, a scanned table of block totals
// for the global rank.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(RM_BLOCK) void k_compact_count(const int* __restrict__ rays_alive, uint32_t n_alive,
                                                            uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds4[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    const bool keep = (n < n_alive) && (rays_alive[n] >= 0);
    const unsigned long long mask = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// Stable compaction in TWO launches (count, then write): every workgroup of the write pass adds up the counts of the workgroups before it itself
// (<= 2,500 L2-resident words for an 800 x 800 frame, 256 lanes at a time) instead of waiting for a one-workgroup scan launch in between -- the loop
// runs this once per iteration, ~66 times per frame, and every launch is ~5 us on its critical path.  The last workgroup also publishes the total.
__global__ __launch_bounds__(RM_BLOCK) void k_compact_write(const int* __restrict__ rays_alive, uint32_t n_alive,
                                                            const uint32_t* __restrict__ block_sums, int* __restrict__ out, int* __restrict__ n_out,
                                                            int* host_pair, int seq) {
    __shared__ uint32_t lds4[RM_BLOCK / 64], lds_base[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    const int v = (n < n_alive) ? rays_alive[n] : -1;
    const bool keep = v >= 0;
    const unsigned long long mask = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t before = 0;                                                   // kept rays in the workgroups before this one
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += RM_BLOCK) before += block_sums[i];
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
    if (lane == 0) { lds4[wave] = (uint32_t)__popcll(mask); lds_base[wave] = before; }
    __syncthreads();
    uint32_t base = 0;
    #pragma unroll
    for (uint32_t w = 0; w < RM_BLOCK / 64; w++) base += lds_base[w];
    uint32_t mine = 0;
    #pragma unroll
    for (uint32_t w = 0; w < RM_BLOCK / 64; w++) { if (w < wave) base += lds4[w]; mine += lds4[w]; }
    if (keep) out[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        uint32_t total = 0;
        #pragma unroll
        for (uint32_t w = 0; w < RM_BLOCK / 64; w++) total += lds_base[w];
        n_out[0] = (int)(total + mine);
        if (host_pair) {                                                   // the count straight into the caller's pinned words, then the sequence number it waits for
            __hip_atomic_store(host_pair, (int)(total + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_pair + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

extern "C" size_t ngp_compact_alive_workspace(uint32_t n_alive) {
    return sizeof(uint32_t) * ((size_t)ngp_div_up(n_alive ? n_alive : 1, RM_BLOCK) + 4);
}

static int rm_compact_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out, int32_t* host_pair, int32_t seq,
                            void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(rays_alive && out && n_out, "compact_alive: null pointer");
    NGP_REQUIRE(workspace && workspace_bytes >= ngp_compact_alive_workspace(n_alive), "compact_alive: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nblocks = ngp_div_up(n_alive ? n_alive : 1, RM_BLOCK);
    uint32_t* block_sums = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_compact_count, dim3(nblocks), dim3(RM_BLOCK), 0, s, rays_alive, n_alive, block_sums);
    hipLaunchKernelGGL(k_compact_write, dim3(nblocks), dim3(RM_BLOCK), 0, s, rays_alive, n_alive, block_sums, out, n_out, host_pair, seq);
    NGP_CHECK_LAUNCH("compact_alive");
    return NGP_OK;
}

extern "C" int ngp_compact_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    return rm_compact_alive(rays_alive, n_alive, out, n_out, nullptr, 0, workspace, workspace_bytes, stream);
}

// The same, and the count also goes to the HOST without a stream synchronisation: host_pair = two int32 of pinned, device-visible, coherent memory
// (ngp_host_words_alloc); the kernel stores the count in [0] and then `seq` in [1] (system scope, release): a caller that needs the number to shape its
// next tensors (the inference loop, nerf/renderer.py:365, once per iteration) polls [1] for its sequence number instead of paying an interrupt-driven
// hipStreamSynchronize + a 4-byte copy per iteration.  What is launched next on the stream is ordered behind the compaction as always.
extern "C" int ngp_compact_alive_publish(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out, int32_t* host_pair, int32_t seq,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(host_pair, "compact_alive_publish: null host words");
    return rm_compact_alive(rays_alive, n_alive, out, n_out, host_pair, seq, workspace, workspace_bytes, stream);
}

extern "C" int ngp_host_words_alloc(uint32_t n_words, void** host_ptr) {
    NGP_REQUIRE(host_ptr && n_words >= 1 && n_words <= 4096, "host_words_alloc: 1 .. 4096 words");
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)n_words * 4, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !p)
        return ngp_fail(NGP_ELAUNCH, "host_words_alloc: hipHostMalloc failed");
    memset(p, 0, (size_t)n_words * 4);
    *host_ptr = p;
    return NGP_OK;
}
extern "C" int ngp_host_words_free(void* host_ptr) {
    if (host_ptr && hipHostFree(host_ptr) != hipSuccess) return ngp_fail(NGP_ELAUNCH, "host_words_free: hipHostFree failed");
    return NGP_OK;
}
