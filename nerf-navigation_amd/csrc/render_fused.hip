// render_fused.hip -- the fused inference path: field evaluation (hash grid -> density MLP -> SH -> colour MLP)
// and a whole frame of NeRFRenderer.run_cuda's inference branch (nerf/renderer.py:325-374) in ONE launch.
//
// Why: the reference's loop is ~15 launches and one host sync per iteration, O(100) iterations per frame, with
// every intermediate (xyzs, dirs, deltas, features, sigmas, rgbs: ~224 B per sample) making a round trip through
// HBM.  Here a persistent grid of waves pulls rays from a queue; each lane owns one ray,
// marches it to its next occupied sample, the wave evaluates the field for its 64 samples on the matrix cores,
// each lane composites its own sample, and finished lanes are refilled from the queue.  Per-sample HBM/L2 traffic
// is the hash-table gather only (512 B algorithmic); rays in and pixels out are amortised over the ray's samples.
//
// Work mapping inside a wave (wave64):
//   march / composite : lane l <-> ray l                               (64 rays in flight per wave)
//   field evaluation  : 4 passes; pass p evaluates the samples of lanes 16p..16p+15 as one 16-column MFMA tile.
//                       In a pass, lane (g = l>>4, s = l&15) gathers levels 4g..4g+3 of column s's sample, which is
//                       exactly its slice of the first layer's B fragment (features 8g..8g+7): the gather lands in
//                       matrix-operand layout with no transpose.
//   density MLP       : ngp_mlp.h (weights in VGPRs, activations never leave registers)
//   colour MLP input  : the reference concatenates [SH(16), geo(15), 0] (nerf/network_ff.py:67-68).  A lane already
//                       holds h[4g..4g+3] of the density output and computes SH[4g..4g+3] itself, so its B fragment
//                       is {h[4g..4g+3], SH[4g..4g+3]} and the colour net's first-layer weights are loaded in that
//                       same k order (column of h0 zeroed: it is the density logit, not a colour input).
//
// Semantics vs the reference loop (DESIGN.md "Fused path"): each ray is marched by a single resumable march from
// `near` (the reference re-enters march_rays every n_step samples); a ray consumes at most max_steps samples (the
// reference offers between max_steps and max_steps+7 depending on the schedule; such rays are counted in stats[1]).
#include <atomic>
#include "ngp_mlp.h"
#include "ngp_sh.h"
#include "ngp_field.h"
#include "ngp_march.h"
#include "ngp_camera.h"

__global__ void k_field_forward_lds(rf_params P, const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                                    float* __restrict__ sigmas, void* __restrict__ rgbs, bool rgb_half);

static int rf_field_forward(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M, float* sigmas, void* rgbs, bool rgb_half, void* stream) {
    rf_params P;
    int rc = rf_fill_params("field_forward", field_host, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && dirs && sigmas && rgbs, "field_forward: null pointer");
    const uint32_t npairs = (M + 31) >> 5;
    uint32_t blocks = ngp_div_up(npairs, RF_BLOCK / 64);
    if (blocks > 256 * RF_FIELD_WG_PER_CU) blocks = 256 * RF_FIELD_WG_PER_CU;
    hipLaunchKernelGGL(k_field_forward_lds, dim3(blocks), dim3(RF_BLOCK), 36 * 1024, (hipStream_t)stream, P, xyzs, dirs, M, sigmas, rgbs, rgb_half);
    NGP_CHECK_LAUNCH("field_forward");
    return NGP_OK;
}

extern "C" int ngp_field_forward(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                                 float* sigmas, float* rgbs, void* stream) {
    return rf_field_forward(field_host, xyzs, dirs, M, sigmas, rgbs, false, stream);
}

// The same with rgbs [M,3] written as HALF -- the dtype nerf/network_ff.py:51-77 returns under autocast (torch.sigmoid of FFMLP's half output); the values are
// halves either way.  Saves the caller's float32 -> float16 copy launch per loop iteration (profiles/r16_dropin_summary.md).
extern "C" int ngp_field_forward_half(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                                      float* sigmas, void* rgbs_half, void* stream) {
    return rf_field_forward(field_host, xyzs, dirs, M, sigmas, rgbs_half, true, stream);
}

// ---------------------------------------------------------------------------
// render_frame: one persistent 1024-thread workgroup per CU (16 waves, 1024 rays in flight), one ray per lane,
// ray queue in global memory.  LDS per workgroup:
//   [0, 36 KiB)        the 36 MFMA weight fragments of both networks, fragment-major (lane l reads 16 B at 16*l:
//                      conflict-free ds_read_b128), so no weight occupies a VGPR and 4 waves fit per SIMD;
//   [36 KiB, +C*4 KiB) coarse occupancy: one bit per 4x4x4 block of density-grid cells.  In Morton order such a
//                      block is 64 consecutive bits = 8 consecutive bytes of the bitfield, so the coarse bit is
//                      just "those 8 bytes != 0".  A probe whose block is empty never touches global memory:
//                      the dependent global load leaves the march's critical path in empty space, and the decision
//                      is identical to reading the fine bit (empty block => empty cell);
//   [.., +16*2 KiB)    the 16 SH coefficients (half) of each lane's current ray, written once per ray.
// ---------------------------------------------------------------------------

// Tuning knobs, each A/B-measured on MI355X with tools/ab_variants.sh (800x800 S-ring frame, ms per frame):
//   256 thr x 2/CU (2 waves/SIMD) 16.8 | 256 x 3 (3 waves/SIMD, 164 VGPR) 12.2 | + paired loads (136 VGPR) 10.5
//   | 512 x 2 (4 waves/SIMD, 128 VGPR, 24 B/lane scratch) 9.2 | + 4x4 patches 9.1
//   | packed-half ReLU, activations once per round, exact reciprocals 8.2 | RV_S = 4 samples per ray per round
//   (1024 x 1, 156 KiB LDS) 6.85   (100 timed frames each; short runs scatter by +-10 %)
// Once the frame had become VALU-bound what counts is how full a round is, i.e. how many samples per lane it
// holds, and LDS bounds that: 1024 threads x 5 samples 4.33 | 768 x 7 4.24 | 512 x 12 (2 waves/SIMD, 256 VGPRs, no scratch;
// march budget 512) 4.05 | 512 x 10 4.57 | 512 x 8 4.25.
#ifndef RV_S
#define RV_S 12                        // samples each lane may march per round (k_render_frame_multi); 1 = k_render_frame
#endif
#ifndef RV_SORT_COLUMNS
#define RV_SORT_COLUMNS 1      // 1: tile columns taken in order of decreasing sample count (fewer dummy columns), 0: lane order
#endif
#ifndef RV_XCD_QUEUES
#define RV_XCD_QUEUES 1        // 1: eight ray queues, one per XCD (image bands), with stealing; 0: one global queue
#endif
// Coherence knobs (round 2; same images, tools/ab_variants.sh and tools/ab_trained.sh: ms per 800x800 frame of the hand-set model | G ray-samples/s on
// a model fitted for 2,000 and for 8,000 steps).  Lane l of a wave gathers for the samples of ray l; what the 64 addresses of one gather instruction
// share (cells of the coarse and middle levels, i.e. cache lines) decides how fast a CU's texture path and L1 turn a tile around.
#ifndef RV_BANDS
#define RV_BANDS 32u              // image bands (ray queues): a multiple of 8, at most 32 (queue words 32..63 of the workspace header).  XCD x works through bands
                                  // x, x + 8, x + 16, x + 24: with 8 thick bands the XCDs whose band was sky or ground finished early and queued behind one another
                                  // on their neighbour's; thin interleaved bands give every XCD a fair sample of the image and keep its rays together.
                                  // 8: 3.47-3.56 | 7.8-8.0    16: 3.40 | 7.4    32: 3.35-3.38 | 7.7    one global queue: 3.29-3.42 | 7.6-7.9, with 2x the fabric traffic
#endif
#ifndef RV_REFILL_MIN
#define RV_REFILL_MIN 64       // a wave draws new rays only when this many of its lanes are free.  64 = a whole 8x8 pixel tile at a time: the rays of a wave stay
#endif                         // neighbours for life (1 = any free lane takes the next ray of the queue at once; after a few rounds a wave holds rays of many
                               // tiles at unrelated depths).   1: 3.78 | 3.73, 4.36    16: 3.84 | 4.43    32: - | 4.88    48: 3.60 | -, 4.95    64: 3.45-3.49 | 5.3-5.5, 5.1-5.3
#ifndef RV_SLAB_STEPS
#define RV_SLAB_STEPS 16       // a round's samples lie within this many steps behind the wave's FRONT = the nearest next sample of its 64 rays (march phase of
#endif                         // k_render_frame_multi): neighbouring rays are sampled at the same depth together, whatever empty space each crossed before.
                               // 0 (off): 3.49 | 5.3, 5.3    8: 3.53 | 8.1, 6.7    12: 3.51 | 7.8    16: 3.46 | 8.1, 6.9    24: 3.45 | 7.7, 6.6    32: 3.46 | 7.3, 6.5    48: 3.49 | 6.7
                               // (limiting every lane to (smallest t of the wave) + 24 steps instead -- the laggard crawls through its empty space 24 steps a
                               //  round -- gave 7.4-7.7 on the fitted model but 4.06 ms on the hand-set one: rounds of 87 samples instead of 636)
#ifndef RV_CU_CHUNKS
#define RV_CU_CHUNKS 0         // 1: the waves of a workgroup (= one CU) draw their 8x8 tiles from a CU-LOCAL chunk of 8 tiles = a block of 4 x 2 adjacent tiles
#endif                         // (32 x 16 pixels), fetched from the band queue with one atomic: neighbouring tiles share an L1.  A/B on MI355X: see DESIGN 3.2.
#ifndef RV_BLOCK_THREADS
#define RV_BLOCK_THREADS 512
#endif
#ifndef RV_BLOCKS_PER_CU
#define RV_BLOCKS_PER_CU (RV_S > 1 ? 1 : 2)   // the sample slots (96 KiB at 512 x 12) only fit beside ONE copy of the weights per CU
#endif
#ifndef RV_TILE_PAIRS
#define RV_TILE_PAIRS 1                // evaluate tiles k and k+1 of a lane group together: two MFMA chains per pass
#endif
#ifndef RV_PIPELINE
#define RV_PIPELINE 0                  // 1: issue the next tile's gathers before the current tile's MLP (software pipeline across
#endif                                 // tiles).  A/B on MI355X: at 4 waves/SIMD (hashed half prefetched) 5.0-5.2 vs 4.8-5.0 ms; at
                                       // 2 waves/SIMD (whole tile prefetched, 208 VGPRs) 4.17-4.30 vs 4.12-4.36 ms: the frame is bound
                                       // by VALU throughput, not by the latency a prefetch would hide.  Off.
#ifndef RV_TILE_ORDER
#define RV_TILE_ORDER 0                // 1: hand out the 8x8 pixel tiles most expensive first (k_tile_estimate / k_tile_order).
#endif                                 // A/B on MI355X: 5.25-5.35 ms with, 5.0-5.2 ms without: the frame is bound by L1 tag and
                                       // VALU throughput, not by its tail, and sorted tiles lose spatial locality.  Kept for scenes
                                       // with a heavier tail; off by default.
#ifndef RV_BLOCK_SKIP
#define RV_BLOCK_SKIP 1                // verified skips through empty 4^3 / 16^3 blocks of the occupancy grid (rv_probe)
#endif
#ifndef RV_PATCH_4X4
#define RV_PATCH_4X4 1                 // each 16-lane column group covers a 4x4 pixel patch
#endif
static constexpr uint32_t RV_BLOCK = RV_BLOCK_THREADS;
static constexpr int RV_WAVES_PER_SIMD = (RV_BLOCK_THREADS / 256) * RV_BLOCKS_PER_CU;
static constexpr int RV_WAVES = RV_BLOCK / 64;
static constexpr uint32_t RV_LDS_W = RV_NFRAG * 1024;                  // weight fragments
static constexpr uint32_t RV_LDS_SH = RV_WAVES * 64 * 32;              // 16 halves per lane
static constexpr uint32_t RV_LDS_LV = 4 * 96;                          // rf_lane_levels of the 4 lane groups
[[maybe_unused]] static constexpr uint32_t RV_LDS_CHUNK = RV_CU_CHUNKS ? 16 : 0;        // the CU's current tile chunk (one 64-bit word), k_render_frame_multi only

struct rf_frame {
    const float* rays_o; const float* rays_d; uint32_t N;   // rays_o == null: the rays are those of `cam` (pixel = ray id)
    ngp_camera cam;
    const ngp_camera* cams; uint32_t frame_rays;             // several frames in one launch (ngp_render_frames_camera): ray r belongs to pixel
                                                             // r % frame_rays of camera cams[r / frame_rays]; null = one frame, `cam`
    float aabb[6]; float min_near;
    const uint8_t* bitfield; uint32_t C, H;
    float dt_gamma; uint32_t max_steps;
    float bg[3];
    float* image; float* depth; float* weights_sum;
    uint32_t* stats; uint32_t* queue;
#ifdef RV_COUNTERS
    uint32_t* hist;                                    // debug timeline (4 x 512 bins) in the tile-order area of the workspace
#endif
    const uint32_t* coarse;          // [C * (H/4)^3 / 32] words, or null when H is not a power of two >= 4
    uint32_t coarse_words;           // words per cascade level
    uint32_t tile_w;                 // image width in pixels when the rays are a row-major image (8x8 tile order), else 0
    uint32_t skip;                   // 1: empty 4^3 / 16^3 blocks may be skipped (rv_probe); decided on the host from H, C, bound
    const uint32_t* occ_ext;         // the extent of the occupied blocks (k_build_coarse), or null: a ray marches no further than where it leaves that box
    float occ_unit, occ_top;         // world size of one unit of that lattice, and the half-width of the outermost cascade (its origin is -occ_top)
    const uint32_t* tile_order;      // [N/64] 8x8 tiles, most expensive first (k_tile_order), or null
};

// coarse[level][m] = any fine bit set in Morton block m (64 bits = 8 bytes of the bitfield)
// coarse[level][m] = any fine bit set in Morton block m (64 bits = 8 bytes of the bitfield).
// ext (optional, 6 words, zeroed by the caller): the extent of the occupied blocks of ALL cascades on one integer lattice (ngp_march.h: ngp_occ_extent_word)
// as [2^20 - lo] x 3 (atomicMax of an inverted minimum) and [hi] x 3; hi == 0: nothing is occupied.
__global__ __launch_bounds__(256) void k_build_coarse(const uint8_t* __restrict__ bitfield, uint32_t n_blocks_total,
                                                      uint32_t* __restrict__ coarse, uint32_t blocks_per_level, uint32_t C, uint32_t* __restrict__ ext) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;                 // one 32-bit word of coarse bits per thread
    if (w * 32 >= n_blocks_total) return;
    const uint64_t* b64 = reinterpret_cast<const uint64_t*>(bitfield);
    uint32_t bits = 0;
    #pragma unroll 8
    for (uint32_t i = 0; i < 32; i++) {
        const uint32_t blk = w * 32 + i;
        if (blk < n_blocks_total && b64[blk] != 0ull) bits |= 1u << i;
    }
    coarse[w] = bits;
    if (ext && bits) {
        uint32_t nb = 1;                                               // blocks per axis
        while (nb * nb * nb < blocks_per_level) nb <<= 1;
        uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
        const uint32_t words_per_level = blocks_per_level / 32u, level = w / words_per_level;
        ngp_occ_extent_word(bits, w - level * words_per_level, level, C, nb, lo, hi);
        #pragma unroll
        for (int k = 0; k < 3; k++) { atomicMax(ext + k, (1u << 20) - lo[k]); atomicMax(ext + 3 + k, hi[k]); }
    }
}

// The march state of ngp_march_t split in two so that only what differs per ray occupies VGPRs:
struct rv_ray { float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz; };
struct rv_consts {                                     // wave-uniform (SGPRs), same formulas as ngp_march_t::setup
    float bound, rbound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Cf, Hm1;
    const uint8_t* grid;
    __device__ __forceinline__ int mip(int e) const { return (int)fminf(Cf - 1.0f, fmaxf(0.0f, (float)e)); }
};
__device__ __forceinline__ float rv_uniform(float v) {     // a wave-uniform value computed by the vector ALU -> SGPR
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
struct rv_view : rv_ray, rv_consts {                   // what rv_probe reads: per-ray VGPRs + uniform SGPRs
    __device__ __forceinline__ rv_view(const rv_ray& r, const rv_consts& k) : rv_ray(r), rv_consts(k) {}
};

typedef ngp_point<rv_view> rv_point;             // one march sample point (ngp_march.h)

// The 64 cells of a 4^3 block are one aligned 64-bit word of the bitfield (Morton order).  A ray tests dozens of lattice
// points per block, so the word it last loaded stays in registers: one 8-byte load per block entered instead of one byte
// load per test.  Under load a dependent global load costs the march thousands of cycles (it queues behind the gathers of
// the 15 other waves of the CU); this takes almost all of them off the march's critical path.
struct rv_block_cache { uint32_t blk = 0xffffffffu, lo = 0, hi = 0; };

// The occupancy test of one lattice point (coarse map, then the cached 64-bit word of the block): everything the reference
// derives from t, and whether the cell is occupied.  `st` keeps what a following rv_leave needs.
struct rv_tested { rv_point r; bool maybe; uint32_t sw; };

__device__ __forceinline__ bool rv_test(const rv_view& m, const uint32_t* __restrict__ lds_coarse, uint32_t coarse_words, rv_block_cache& bc,
                                        float tc, rv_tested& st
#ifdef RV_COUNTERS
                                        , int* rv_dbg_ptr
#endif
                                        ) {
    rv_point& r = st.r;
    r.at(m, tc);
    const uint32_t mort = ngp_morton3((uint32_t)r.nx, (uint32_t)r.ny, (uint32_t)r.nz);
    st.maybe = true;
    st.sw = 0;
    if (lds_coarse) {
        const uint32_t blk = mort >> 6;
        st.maybe = (lds_coarse[(uint32_t)r.level * coarse_words + (blk >> 5)] >> (blk & 31u)) & 1u;
        st.sw = (uint32_t)r.level * coarse_words + ((blk >> 6) << 1);    // the 64 coarse bits of the 16^3 block: 2 words
#ifdef RV_COUNTERS
        *rv_dbg_ptr = st.maybe ? 0 : ((lds_coarse[st.sw] | lds_coarse[st.sw + 1]) == 0u ? 2 : 1);
#endif
    }
    if (!st.maybe) return false;
    if (lds_coarse) {
        // with a coarse map the cell index level * H^3 + morton is exact in binary32 (host: C * H^3 <= 2^24), so the
        // reference's bit (raymarching.cu:783-784) is bit (morton & 63) of word level * H^3 / 64 + (morton >> 6)
        const uint32_t gblk = (uint32_t)r.level * (coarse_words << 5) + (mort >> 6);
        if (gblk != bc.blk) {
            const uint2 w = reinterpret_cast<const uint2*>(m.grid)[gblk];
            bc.blk = gblk; bc.lo = w.x; bc.hi = w.y;
        }
        return (((mort & 32u) ? bc.hi : bc.lo) >> (mort & 31u)) & 1u;
    }
    const uint32_t index = (uint32_t)((float)r.level * m.H3 + (float)mort);
    return (m.grid[index >> 3] >> (index & 7u)) & 1u;
}

// Leave the empty cell tested at tc (state `st`): a verified block skip when the block is empty (ngp_march.h), else the
// reference's step.
template <bool SKIP>
__device__ __forceinline__ float rv_leave(const rv_view& m, const uint32_t* __restrict__ lds_coarse, float M, const rv_tested& st, float tc) {
    const rv_point& r = st.r;
    const float tt = r.cell_exit(m, tc);
    if (SKIP && lds_coarse && !st.maybe) {
        const int sh = ((lds_coarse[st.sw] | lds_coarse[st.sw + 1]) == 0u) ? 4 : 2;
        const float ta = ngp_try_skip(m, r, tc, tt, sh, M);
        if (ta >= 0.0f) return ta;
    }
    float tp;
    return ngp_advance(m, tc, tt, tp);
}

template <bool SKIP>
__device__ __forceinline__ bool rv_probe(const rv_ray& ray, const rv_consts& k, const uint32_t* __restrict__ lds_coarse, uint32_t coarse_words,
                                         float M, rv_block_cache& bc, float& t, float& x, float& y, float& z, float& dt
#ifdef RV_COUNTERS
                                         , int* rv_dbg_ptr
#endif
                                         ) {
    const rv_view m(ray, k);
    rv_tested st;
#ifdef RV_COUNTERS
    const bool occ = rv_test(m, lds_coarse, coarse_words, bc, t, st, rv_dbg_ptr);
#else
    const bool occ = rv_test(m, lds_coarse, coarse_words, bc, t, st);
#endif
    x = st.r.x; y = st.r.y; z = st.r.z; dt = st.r.dt;
    if (occ) return true;
    t = rv_leave<SKIP>(m, lds_coarse, M, st, t);
    return false;
}


// ---------------------------------------------------------------------------
// field_forward for explicit points (NeRFNetwork.forward in one launch): the frame kernel's field step on its own.
// 256-thread workgroups, weights in LDS (36 KiB each, RF_FIELD_WG_PER_CU workgroups share a CU), two 16-point tiles per pass.
// ---------------------------------------------------------------------------
template <bool FIXED>
__device__ __forceinline__ void rf_points_loop(const rf_params& P, const rf_iter_class cls_rt, const rf_lane_levels& lv, const ngp_h8* __restrict__ lds_w,
                                               const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                                               float* __restrict__ sigmas, void* __restrict__ rgbs_v, bool rgb_half) {
    const rf_iter_class cls = FIXED ? rf_iter_class{1u, 12u, 2u} : cls_rt;
    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
    const uint32_t wave = (blockIdx.x * RF_BLOCK + threadIdx.x) >> 6, nwaves = gridDim.x * (RF_BLOCK / 64);
    const uint32_t npairs = (M + 31) >> 5;
    for (uint32_t pair = wave; pair < npairs; pair += nwaves) {
        ngp_h8 x[2];
        ngp_h4 shq[2];
        uint32_t m[2];
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            m[n] = pair * 32 + 16 * n + s;
            const uint64_t mm = m[n] < M ? m[n] : 0;
            const float px = xyzs[3 * mm], py = xyzs[3 * mm + 1], pz = xyzs[3 * mm + 2];
            float sh[16];
            sh_eval<4>(dirs[3 * mm], dirs[3 * mm + 1], dirs[3 * mm + 2], P.shn, sh);
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = sh[j];
                if (g == 1) v = sh[4 + j];
                if (g == 2) v = sh[8 + j];
                if (g == 3) v = sh[12 + j];
                shq[n][j] = ngp_f2h(v);               // cat(...) enters FFMLP through cast_inputs=half
            }
            x[n] = rf_encode<false>(P, lv, cls, px, py, pz);
        }
        float sg[2], cr[2], cg[2], cb[2];
        rv_mlp_tiles<2>(lds_w, lane, x, shq, sg, cr, cg, cb);
        #pragma unroll
        for (int n = 0; n < 2; n++)
            if (g == 0 && m[n] < M) {
                rv_activate(P, sg[n], cr[n], cg[n], cb[n]);
                sigmas[m[n]] = sg[n];
                if (rgb_half) {                                // (wave-uniform) the colours ARE halves: rv_activate rounded them
                    _Float16* rgbs = static_cast<_Float16*>(rgbs_v);
                    rgbs[3ull * m[n]] = (_Float16)cr[n]; rgbs[3ull * m[n] + 1] = (_Float16)cg[n]; rgbs[3ull * m[n] + 2] = (_Float16)cb[n];
                } else {
                    float* rgbs = static_cast<float*>(rgbs_v);
                    rgbs[3ull * m[n]] = cr[n]; rgbs[3ull * m[n] + 1] = cg[n]; rgbs[3ull * m[n] + 2] = cb[n];
                }
            }
    }
}

__global__ __launch_bounds__(RF_BLOCK, RF_FIELD_WG_PER_CU) void k_field_forward_lds(rf_params P, const float* __restrict__ xyzs, const float* __restrict__ dirs,
                                                                    uint32_t M, float* __restrict__ sigmas, void* __restrict__ rgbs, bool rgb_half) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rf_smem);
    const int lane = threadIdx.x & 63, g = lane >> 4, wave = threadIdx.x >> 6;
    rv_stage_weights(P, lds_w, wave, RF_BLOCK / 64, lane);
    rf_lane_levels lv;
    rf_setup_levels(P, g, lv);
    __syncthreads();
    const rf_iter_class cls = rf_classify(lv);
    if (cls.dense == 1u && cls.select == 2u && cls.hashed == 12u) rf_points_loop<true>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs, rgb_half);
    else rf_points_loop<false>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs, rgb_half);
}

// queue index -> ray id.  With tile_w set (rays are a row-major image whose width and height are multiples of 8)
// consecutive queue indices walk 8x8 pixel tiles, so the 64 lanes of a wave start on a compact patch of the image
// and their gathers share cache lines; otherwise the identity.
__device__ __forceinline__ uint32_t rv_ray_of(uint32_t idx, uint32_t tile_w, const uint32_t* __restrict__ tile_order, uint32_t n_rays = 0) {
    if (tile_w == 0) return idx;
    uint32_t tile = idx >> 6;
    const uint32_t in = idx & 63u, tiles_x = tile_w >> 3;
    if (tile_order) tile = tile_order[tile];
#if RV_CU_CHUNKS
    // eight consecutive tiles form a block of 4 x 2 tiles (the chunk a CU draws at once) when the image divides into such blocks (else row-major as before)
    uint32_t ty, tx;
    if ((tiles_x & 3u) == 0 && ((n_rays / tile_w) & 15u) == 0) {      // whole 4 x 2 blocks only: the image is a multiple of 32 x 16 pixels
        const uint32_t c = tile >> 3, i = tile & 7u, bpr = tiles_x >> 2;
        const uint32_t by = c / bpr, bx = c - by * bpr;
        ty = by * 2 + (i >> 2); tx = bx * 4 + (i & 3u);
    } else { ty = tile / tiles_x; tx = tile - ty * tiles_x; }
#else
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
#endif
#if RV_PATCH_4X4
    // lanes 16p..16p+15 (one MFMA column tile, one gather instruction group) cover a compact 4x4 pixel patch
    const uint32_t p = in >> 4, s = in & 15u;
    const uint32_t px = (p & 1u) * 4 + (s & 3u), py = (p >> 1) * 4 + (s >> 2);
    return (ty * 8 + py) * tile_w + tx * 8 + px;
#else
    return (ty * 8 + (in >> 3)) * tile_w + tx * 8 + (in & 7u);
#endif
}

#if RV_S == 1
__global__ __launch_bounds__(RV_BLOCK, RV_WAVES_PER_SIMD) void k_render_frame(rf_params P, rf_frame F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rv_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rv_smem);
    _Float16* lds_sh = reinterpret_cast<_Float16*>(rv_smem + RV_LDS_W);
    uint32_t* lds_coarse = F.coarse ? reinterpret_cast<uint32_t*>(rv_smem + RV_LDS_W + RV_LDS_SH + RV_LDS_LV) : nullptr;

    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15, wave = threadIdx.x >> 6;

    // ---- stage the weight fragments (each in the k order its consumer expects) and the coarse map ----
    rv_stage_weights(P, lds_w, wave, RV_WAVES, lane);
    if (lds_coarse) {
        const uint32_t nw = F.coarse_words * F.C;
        for (uint32_t i = threadIdx.x; i < nw; i += RV_BLOCK) lds_coarse[i] = F.coarse[i];
    }
    // per-lane-group level constants live in LDS (re-read in each pass) instead of 24 VGPRs across the march loop
    rf_lane_levels* lds_lv = reinterpret_cast<rf_lane_levels*>(rv_smem + RV_LDS_W + RV_LDS_SH);
    if (wave == 0 && s == 0) {
        rf_lane_levels tmp;
        rf_setup_levels(P, g, tmp);
        lds_lv[g] = tmp;
    }
    __syncthreads();
    const rf_iter_class cls = rf_classify(lds_lv[g]);

    _Float16* my_sh = lds_sh + (wave * 64 + lane) * 16;                // this lane's ray
    const _Float16* wave_sh = lds_sh + wave * 64 * 16;

    // wave-uniform march constants (ngp_march_t::setup's formulas), kept out of the per-ray state
    // float conversions and divisions run on the vector ALU even for uniform inputs; rv_uniform moves the results to SGPRs
    rv_consts K;
    K.bound = P.bound; K.rbound = rv_uniform(1.0f / P.bound); K.dt_gamma = F.dt_gamma;
    K.Hf = rv_uniform((float)F.H); K.Cf = rv_uniform((float)F.C); K.Hm1 = rv_uniform((float)(F.H - 1));
    K.rH = rv_uniform(1.0f / K.Hf);
    K.H3 = rv_uniform((float)(F.H * F.H * F.H));
    K.dt_min = rv_uniform((2.0f * 1.7320508075688772f) / (float)F.max_steps);
    K.dt_max = rv_uniform(((2.0f * 1.7320508075688772f) * (float)(1 << (F.C - 1))) / K.Hf);
    K.grid = F.bitfield;

    bool active = false;
    uint32_t ray = 0, nsamp = 0;
    rv_ray m;
    float t = 0, last_t = 0, near = 0, far = 0;
    float ws = 0, dacc = 0, cr = 0, cg = 0, cb = 0, tcomp = 0;
    bool exhausted = false;
    rv_block_cache bc;
    uint32_t n_samples_local = 0;

    for (;;) {
        // ---- refill finished lanes from the queue (one atomic per wave) ----
        if (!exhausted) {
            const unsigned long long need = __ballot(!active);
            if (need) {
                const uint32_t cnt = (uint32_t)__popcll(need);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(F.queue, cnt);
                base = __shfl(base, 0, 64);
                if (!active) {
                    const uint32_t idx = base + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                    if (idx < F.N) {
                        ray = rv_ray_of(idx, F.tile_w, F.tile_order);
                        const float* o = F.rays_o + 3ull * ray;
                        const float* d = F.rays_d + 3ull * ray;
                        ngp_near_far_inline(o, d, F.aabb, F.min_near, near, far);
                        m.ox = o[0]; m.oy = o[1]; m.oz = o[2];
                        m.dx = d[0]; m.dy = d[1]; m.dz = d[2];
                        m.rdx = 1.0f / m.dx; m.rdy = 1.0f / m.dy; m.rdz = 1.0f / m.dz;
                        t = near; last_t = near; tcomp = near;
                        ws = 0; dacc = 0; cr = 0; cg = 0; cb = 0; nsamp = 0;
                        float sh[16];
                        sh_eval<4>(m.dx, m.dy, m.dz, P.shn, sh);      // the ray's direction encoding, once per ray
                        #pragma unroll
                        for (int j = 0; j < 16; j++) my_sh[j] = ngp_f2h(sh[j]);
                        active = true;
                    }
                }
                if (base + cnt >= F.N) exhausted = true;
            }
        }
        if (__ballot(active) == 0ull) break;            // queue drained and every ray of this wave is finished

        // ---- march each active lane towards its next occupied sample (bounded probes per round) ----
        bool has = false, ended = false;
        float x = 0, y = 0, z = 0, dt = 0, d1 = 0;
        if (active) {
            int probes = 0;
            for (;;) {
                if (!(t < far && nsamp < F.max_steps)) { ended = true; break; }
                if (rv_probe<false>(m, K, lds_coarse, F.coarse_words, 0.0f, bc, t, x, y, z, dt)) { has = true; break; }
                if (++probes >= RF_PROBES_PER_ROUND) break;
            }
            if (has) {
                t += dt;
                d1 = t - last_t;
                last_t = t;
                nsamp++;
            }
        }

        // ---- field evaluation: 4 passes of 16 columns ----
        float sig = 0, sr = 0, sg = 0, sb = 0;
        #pragma unroll 1
        for (int p = 0; p < 4; p++) {
            const int src = 16 * p + s;
            const bool v = __shfl((int)has, src, 64) != 0;
            if (__ballot(v) == 0ull) continue;           // wave-uniform: nothing to evaluate in this pass
#ifdef RV_EXPERIMENT_SAMEPOS       // timing-only build: columns share positions in groups of RV_EXPERIMENT_SAMEPOS
            const int psrc = 16 * p + (s & ~(RV_EXPERIMENT_SAMEPOS - 1));
            const float qx = __shfl(x, psrc, 64), qy = __shfl(y, psrc, 64), qz = __shfl(z, psrc, 64);
#else
            const float qx = __shfl(x, src, 64), qy = __shfl(y, src, 64), qz = __shfl(z, src, 64);
#endif
            const ngp_h4 shq = *reinterpret_cast<const ngp_h4*>(wave_sh + src * 16 + 4 * g);
            float a, b, c, d;
            const rf_lane_levels lv = lds_lv[g];
            rv_field_tile(P, lv, cls, lds_w, lane, qx, qy, qz, shq, a, b, c, d);
            const float ra = __shfl(a, s, 64), rb = __shfl(b, s, 64), rc = __shfl(c, s, 64), rd = __shfl(d, s, 64);
            if (g == p) { sig = ra; sr = rb; sg = rc; sb = rd; }
        }

        // ---- composite (kernel_composite_rays arithmetic, raymarching.cu:865-896) ----
        bool done = ended;
        if (has) {
            rv_activate(P, sig, sr, sg, sb);
            n_samples_local++;
            const float alpha = 1.0f - ngp_expf(-sig * dt);
            const float T = 1 - ws;
            const float w = alpha * T;
            ws += w;
            tcomp += d1;
            dacc += w * tcomp;
            cr += w * sr; cg += w * sg; cb += w * sb;
            if ((double)T < 1e-4) done = true;
        }
        if (done) {
            F.image[3ull * ray] = cr + (1 - ws) * F.bg[0];           // nerf/renderer.py:371-372
            F.image[3ull * ray + 1] = cg + (1 - ws) * F.bg[1];
            F.image[3ull * ray + 2] = cb + (1 - ws) * F.bg[2];
            F.depth[ray] = fmaxf(dacc - near, 0.0f) / (far - near);
            F.weights_sum[ray] = ws;
            if (nsamp >= F.max_steps && t < far) atomicAdd(F.stats + 1, 1u);
            if (nsamp > 0) atomicAdd(F.stats + 2, 1u);
            active = false;
        }
    }
    uint32_t tot = n_samples_local;
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if (lane == 0 && tot) atomicAdd(F.stats, tot);
}

#endif  // RV_S == 1

// ---------------------------------------------------------------------------
// k_render_frame_multi: the same frame kernel with RV_S samples per ray per round.
//
// The march is a lock-step loop: as long as one lane of the wave is still crossing empty space the other 63 wait, and in
// steady state some lane always is.  Here a round's march loop lets every lane collect up to RV_S samples (lanes in
// dense space fill up in RV_S iterations, lanes in empty space use the whole probe budget), so its cost is shared by up
// to RV_S samples per lane instead of one.  Samples wait in an LDS slot array [lane][RV_S] = (x, y, z, dt) + d1; the field
// is evaluated tile by tile, a tile being sample k of the 16 rays of one column group; the half-precision network outputs
// overwrite the first 8 bytes of the slot; each lane then composites its samples in order.  A ray that saturates at
// sample j < count has marched count-1-j samples too many: they are discarded (they were never composited, so images,
// counts and statistics are identical to the one-sample kernel); the price is their field evaluation.
// ---------------------------------------------------------------------------
#if RV_S > 1
static constexpr uint32_t RV_LDS_SMP = RV_WAVES * 64 * RV_S * 16;      // per sample: float4 (x, y, z, t before the step)

// ---------------------------------------------------------------------------
// Tile order.  A frame's rays differ a lot in cost (0 to a few hundred samples) and a lane only ever sees two or three
// of them, so the order in which the queue hands them out decides how long the last waves run alone.  Longest first:
// k_tile_estimate marches the centre ray of every 8x8 pixel tile through the occupancy grid and counts its samples (no
// field evaluation); k_tile_order sorts the tiles by that count, descending (counting sort, one workgroup).  The frame
// kernel maps queue position -> tile through the table.  Results do not depend on the order (tests: order invariance).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_estimate(rf_frame F, float bound, uint32_t n_tiles, uint32_t* __restrict__ est) {
    const uint32_t tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t tiles_x = F.tile_w >> 3;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t ray = (ty * 8 + 4) * F.tile_w + tx * 8 + 4;
    const float* o = F.rays_o + 3ull * ray;
    const float* d = F.rays_d + 3ull * ray;
    float near, far;
    ngp_near_far_inline(o, d, F.aabb, F.min_near, near, far);
    rv_ray m;
    m.ox = o[0]; m.oy = o[1]; m.oz = o[2];
    m.dx = d[0]; m.dy = d[1]; m.dz = d[2];
    m.rdx = 1.0f / m.dx; m.rdy = 1.0f / m.dy; m.rdz = 1.0f / m.dz;
    rv_consts K;
    K.bound = bound; K.rbound = 1.0f / bound; K.dt_gamma = F.dt_gamma;
    K.Hf = (float)F.H; K.Cf = (float)F.C; K.Hm1 = (float)(F.H - 1);
    K.rH = 1.0f / K.Hf;
    K.H3 = (float)(F.H * F.H * F.H);
    K.dt_min = (2.0f * 1.7320508075688772f) / (float)F.max_steps;
    K.dt_max = ((2.0f * 1.7320508075688772f) * (float)(1 << (F.C - 1))) / K.Hf;
    K.grid = F.bitfield;
    const float M = F.skip ? ngp_skip_margin(m, bound, far) : __builtin_inff();
    float t = near;
    uint32_t n = 0;
    int probes = 0;
    rv_block_cache bc;
    while (t < far && n < 1023u && probes < 4096) {
        float x, y, z, dt;
#ifdef RV_COUNTERS
        int pc;
        if (rv_probe<true>(m, K, F.coarse, F.coarse_words, M, bc, t, x, y, z, dt, &pc)) { t += dt; n++; }
#else
        if (rv_probe<true>(m, K, F.coarse, F.coarse_words, M, bc, t, x, y, z, dt)) { t += dt; n++; }
#endif
        probes++;
    }
    est[tile] = n;
}

__global__ __launch_bounds__(1024) void k_tile_order(const uint32_t* __restrict__ est, uint32_t n_tiles, uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n_tiles; i += 1024) atomicAdd(&hist[1023u - est[i]], 1u);      // bin 0 = most samples
    __syncthreads();
    // exclusive prefix sum over the 1024 bins: wave scan, then the 16 wave totals
    const uint32_t v = hist[tid];
    uint32_t inc = v;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(inc, off, 64);
        if ((int)(tid & 63u) >= off) inc += up;
    }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) base += wsum[w];
    __syncthreads();
    hist[tid] = base + inc - v;
    __syncthreads();
    for (uint32_t i = tid; i < n_tiles; i += 1024) order[atomicAdd(&hist[1023u - est[i]], 1u)] = i;
}

// The persistent loop of k_render_frame_multi.  FIXED selects compile-time iteration classes for the reference's grid
// (iteration 0 dense, 1 mixed, 2 and 3 hashed: 16 levels from 16^3 at 2^19 rows per level): with the classes constant the
// encoder is straight-line code and all 32 gathers of a tile are in flight together; with run-time classes the compiler
// chains the alternatives and waits for one iteration's loads before issuing the next.
template <bool FIXED>
__device__ __forceinline__ void rv_frame_loop(const rf_params& P, const rf_frame& F, const rf_iter_class cls_rt,
                                              const ngp_h8* __restrict__ lds_w, _Float16* lds_sh, const rf_lane_levels* lds_lv,
                                              float4* lds_smp, const uint32_t* lds_coarse, unsigned long long* lds_chunk, const float* lds_occ) {
    const rf_iter_class cls = FIXED ? rf_iter_class{1u, 12u, 2u} : cls_rt;
    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15, wave = threadIdx.x >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);          // the same, known to be uniform

    _Float16* my_sh = lds_sh + (wave * 64 + lane) * 16;
    const _Float16* wave_sh = lds_sh + wave * 64 * 16;
    float4* my_smp = lds_smp + (wave * 64 + lane) * RV_S;               // this lane's sample slots
    float4* wave_smp = lds_smp + wave * 64 * RV_S;

    // float conversions and divisions run on the vector ALU even for uniform inputs; rv_uniform moves the results to SGPRs
    rv_consts K;
    K.bound = P.bound; K.rbound = rv_uniform(1.0f / P.bound); K.dt_gamma = F.dt_gamma;
    K.Hf = rv_uniform((float)F.H); K.Cf = rv_uniform((float)F.C); K.Hm1 = rv_uniform((float)(F.H - 1));
    K.rH = rv_uniform(1.0f / K.Hf);
    K.H3 = rv_uniform((float)(F.H * F.H * F.H));
    K.dt_min = rv_uniform((2.0f * 1.7320508075688772f) / (float)F.max_steps);
    K.dt_max = rv_uniform(((2.0f * 1.7320508075688772f) * (float)(1 << (F.C - 1))) / K.Hf);
    K.grid = F.bitfield;

    bool active = false;
    uint32_t ray = 0, nsamp = 0;
    rv_ray m;
    float t = 0, last_t = 0, near = 0, far = 0;
    float ws = 0, dacc = 0, cr = 0, cg = 0, cb = 0, tcomp = 0;
    bool exhausted = false;
#if RV_XCD_QUEUES
    uint32_t rv_q = blockIdx.x & 7u, rv_q_seen = 0;        // blockIdx % 8 labels the workgroups that share an XCD
#endif
    (void)lds_chunk;
    rv_block_cache bc;                                 // bitfield word of the block the ray last tested (block ids are global: stays valid across rays)
    uint32_t n_samples_local = 0, n_tiles = 0, n_capped_local = 0, n_hit_local = 0;
#ifdef RV_COUNTERS
    uint32_t n_rounds = 0, n_trips = 0, n_probe[4] = {0, 0, 0, 0};   // [3]: samples evaluated behind a ray's last one
    unsigned long long c_refill = 0, c_march = 0, c_tiles = 0, c_comp = 0;
    const unsigned long long c_start = __builtin_readcyclecounter();
    unsigned long long c_last = c_start;
#define RV_TICK(acc) { const unsigned long long c_now = __builtin_readcyclecounter(); acc += c_now - c_last; c_last = c_now; }
    uint32_t* rv_hist = F.hist;
    unsigned long long rv_t0 = 0;
    if (rv_hist) {
        unsigned long long* p0 = reinterpret_cast<unsigned long long*>(F.queue + 24);
        const unsigned long long now = wall_clock64();
        const unsigned long long old = atomicCAS(p0, 0ull, now);
        rv_t0 = __shfl(old ? old : now, 0, 64);
    }
#else
#define RV_TICK(acc)
#endif

    for (;;) {
        if (!exhausted) {
            const unsigned long long need = __ballot(!active);
            if (need && (uint32_t)__popcll(need) >= RV_REFILL_MIN) {
                const uint32_t cnt = (uint32_t)__popcll(need);
                uint32_t base = 0;
#if RV_XCD_QUEUES && RV_CU_CHUNKS && RV_REFILL_MIN == 64
                // One 8x8 tile for this wave, out of the CU's current chunk of 8 adjacent tiles (LDS word, compare-and-swap); the wave that finds the chunk
                // used up fetches the next one from ITS band's queue with one atomic and publishes it while the others wait.  Bands start at multiples of 8 tiles.
                const uint32_t n_tiles64 = (F.N + 63u) >> 6;
                uint32_t q_hi = 0;
                {
                    uint32_t tile = 0xFFFFFFFFu, band = rv_q;
                    if (lane == 0) {
                        for (;;) {
                            const unsigned long long d = __hip_atomic_load(lds_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const uint32_t taken = (uint32_t)(d >> 32) & 0xFFu;
                            unsigned long long want = d;
                            if (taken < 8u) {                  // a tile is left: take it
                                if (__hip_atomic_compare_exchange_strong(lds_chunk, &want, d + (1ull << 32), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                    tile = (uint32_t)d + taken; band = (uint32_t)(d >> 40); break;
                                }
                            } else if (taken == 8u) {          // used up: ONE wave fetches the next chunk (state 0xFF while it does), the others wait for it
                                if (__hip_atomic_compare_exchange_strong(lds_chunk, &want, d | (0xFFull << 32), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                    const uint32_t b_lo = (uint32_t)(((unsigned long long)n_tiles64 * rv_q) / RV_BANDS) & ~7u;
                                    const uint32_t b_hi = rv_q + 1u == RV_BANDS ? n_tiles64 : ((uint32_t)(((unsigned long long)n_tiles64 * (rv_q + 1u)) / RV_BANDS) & ~7u);
                                    const uint32_t got = atomicAdd(F.queue + 32 + rv_q, 8u) + b_lo;
                                    if (got >= b_hi) {         // this wave's band is dry: hand the slot back, move on to the next band
                                        __hip_atomic_store(lds_chunk, 8ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        tile = 0xFFFFFFFEu; break;
                                    }
                                    __hip_atomic_store(lds_chunk, (unsigned long long)got | ((unsigned long long)rv_q << 40) | (1ull << 32), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                                    tile = got; break;
                                }
                            } else {
                                __builtin_amdgcn_s_sleep(4);   // a fetch is in flight (one global atomic: ~2 us)
                            }
                        }
                    }
                    tile = __shfl(tile, 0, 64); band = __shfl(band, 0, 64);
                    if (tile == 0xFFFFFFFEu) {             // band dry: move on (same walk as the plain queues), try again next round
                        rv_q += 8u;
                        if (rv_q >= RV_BANDS) rv_q = (rv_q + 1u) & 7u;
                        if (++rv_q_seen == RV_BANDS) exhausted = true;
                        continue;
                    }
                    const uint32_t t_hi = band + 1u == RV_BANDS ? n_tiles64 : ((uint32_t)(((unsigned long long)n_tiles64 * (band + 1u)) / RV_BANDS) & ~7u);
                    base = tile << 6;
                    q_hi = tile < t_hi ? ((tile + 1u) << 6) : 0u;                                // a chunk may reach past its band's end: those tiles are nobody's
                    q_hi = q_hi < F.N ? q_hi : F.N;
                }
#elif RV_XCD_QUEUES
                // Queue q holds the rays of image band q (tiles [T q / RV_BANDS, T (q + 1) / RV_BANDS) of 64 rays): the workgroups of one XCD
                // work on the same band, so that the table lines neighbouring rays share are fetched into ONE L2 and
                // not into eight.  A wave that finds its queue empty moves on to its XCD's next band, then to the other XCDs'.
                const uint32_t n_tiles64 = (F.N + 63u) >> 6;
                const uint32_t q_lo = (uint32_t)(((unsigned long long)n_tiles64 * rv_q) / RV_BANDS) << 6;
                uint32_t q_hi = (uint32_t)(((unsigned long long)n_tiles64 * (rv_q + 1u)) / RV_BANDS) << 6;
                q_hi = q_hi < F.N ? q_hi : F.N;
                if (lane == 0) base = atomicAdd(F.queue + 32 + rv_q, cnt);
                base = __shfl(base, 0, 64) + q_lo;
#else
                if (lane == 0) base = atomicAdd(F.queue, cnt);
                base = __shfl(base, 0, 64);
#endif
                if (!active) {
                    const uint32_t idx = base + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
#if RV_XCD_QUEUES
                    if (idx < q_hi) {
#else
                    if (idx < F.N) {
#endif
                        ray = rv_ray_of(idx, F.tile_w, F.tile_order, F.N);
                        float o[3], d[3];
                        if (F.rays_o) {
                            #pragma unroll
                            for (int k = 0; k < 3; k++) { o[k] = F.rays_o[3ull * ray + k]; d[k] = F.rays_d[3ull * ray + k]; }
                        } else if (F.cams) {               // camera mode, several frames per launch: the frame's camera from memory (a tile lies in one frame)
                            const uint32_t f = ray / F.frame_rays;
                            const ngp_camera c = F.cams[f];
                            o[0] = c.t[0]; o[1] = c.t[1]; o[2] = c.t[2];
                            ngp_camera_ray(c, ray - f * F.frame_rays, d);
                        } else {                           // camera mode: get_rays fused into the refill (nerf/utils.py:98-108)
                            o[0] = F.cam.t[0]; o[1] = F.cam.t[1]; o[2] = F.cam.t[2];
                            ngp_camera_ray(F.cam, ray, d);
                        }
                        ngp_near_far_inline(o, d, F.aabb, F.min_near, near, far);
                        if (lds_occ) {
                            // `far` is from here on where the MARCH stops: no later than a little behind the box of everything occupied (beyond it every
                            // cell is empty: the reference would test and find nothing).  The ray's own far is formed again when the ray retires.
                            float n2, f2;
                            ngp_near_far_inline(o, d, lds_occ, 0.0f, n2, f2);
                            far = f2 == 3.402823466e+38f ? near : fminf(far, f2 + lds_occ[6]);
                        }
                        m.ox = o[0]; m.oy = o[1]; m.oz = o[2];
                        m.dx = d[0]; m.dy = d[1]; m.dz = d[2];
                        m.rdx = 1.0f / m.dx; m.rdy = 1.0f / m.dy; m.rdz = 1.0f / m.dz;
                        t = near; last_t = near; tcomp = near;
                        ws = 0; dacc = 0; cr = 0; cg = 0; cb = 0; nsamp = 0;
                        float sh[16];
                        sh_eval<4>(m.dx, m.dy, m.dz, P.shn, sh);
                        #pragma unroll
                        for (int j = 0; j < 16; j++) my_sh[j] = ngp_f2h(sh[j]);
                        active = true;
                    }
                }
#if RV_XCD_QUEUES && RV_CU_CHUNKS && RV_REFILL_MIN == 64
                (void)cnt;
#elif RV_XCD_QUEUES
                if (base + cnt >= q_hi) {                  // this queue has run dry: move on, until all eight have been seen
                    // XCD x owns bands x, x + 8, x + 16, ...: thin bands spread over the whole image, so that every XCD gets a fair sample of cheap and
                    // expensive rows; when its own are done it goes on with the next XCD's
                    rv_q += 8u;
                    if (rv_q >= RV_BANDS) rv_q = (rv_q + 1u) & 7u;
                    if (++rv_q_seen == RV_BANDS) exhausted = true;
                }
#else
                if (base + cnt >= F.N) exhausted = true;
#endif
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;                                  // the queue ran dry under this wave: draw from the next one
        }
        RV_TICK(c_refill)

        // ---- march: up to RV_S samples per lane within one shared probe budget ----
        int cnt = 0;
        bool ended = false;
#ifdef RV_COUNTERS
        n_rounds++;
        const unsigned long long rv_live = __ballot(active);
#endif
        {
            // The ray's constants live across the field evaluation, where every register is taken, so the allocator keeps them
            // in scratch; without this copy it reloads them at each use inside the probe loop (10 scratch loads per probe on
            // the march's critical path).  The copy is defined here and dies with the loop: one reload per round.
            rv_ray mr = m;
            float far_r = far;
            asm("" : "+v"(mr.ox), "+v"(mr.oy), "+v"(mr.oz), "+v"(mr.dx), "+v"(mr.dy), "+v"(mr.dz),
                     "+v"(mr.rdx), "+v"(mr.rdy), "+v"(mr.rdz), "+v"(far_r));
            // the same for the march constants: uniform, but the scalar registers are all taken, so they live in (spilled)
            // vector registers; a copy per round keeps every memory access out of the probe loop
            rv_consts Kr = K;
            asm("" : "+v"(Kr.Hf), "+v"(Kr.Hm1), "+v"(Kr.Cf), "+v"(Kr.rH), "+v"(Kr.dt_min), "+v"(Kr.dt_max), "+v"(Kr.rbound), "+v"(Kr.H3));
            // (the margin bounds rounding errors by the largest ray parameter in play: the march limit may have been pulled in to the occupied box, the
            //  points a skip evaluates lie at most one cascade further)
            const float M = (active && F.skip) ? ngp_skip_margin(mr, K.bound, lds_occ ? far_r + 4.0f * K.bound : far_r) : __builtin_inff();
            // this lane's slots, recomputed from the lane id (2 VALU) rather than kept across the field evaluation in scratch
            const uint32_t slot0 = ((uint32_t)wave_s * 64u + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) * RV_S;
            float4* const smp_w = lds_smp + slot0;
            int probes = 0;
            float t_lim = __builtin_inff();
            // the probe loop: collects samples while cnt < upto and t < t_lim
            auto march = [&](int upto) {
                for (;;) {
#ifdef RV_COUNTERS
                    n_trips = __builtin_amdgcn_readfirstlane(n_trips) + 1;
#endif
                    if (!(t < far_r && nsamp < F.max_steps)) { ended = true; break; }
                    if (RV_SLAB_STEPS > 0 && !(t < t_lim)) break;
                    float x, y, z, dt;
#ifdef RV_COUNTERS
                    int pc = 0;
                    const bool hit = rv_probe<RV_BLOCK_SKIP != 0>(mr, Kr, lds_coarse, F.coarse_words, M, bc, t, x, y, z, dt, &pc);
                    n_probe[pc]++;
                    if (hit) {
#else
                    if (rv_probe<RV_BLOCK_SKIP != 0>(mr, Kr, lds_coarse, F.coarse_words, M, bc, t, x, y, z, dt)) {
#endif
                        smp_w[cnt] = make_float4(x, y, z, t);     // the compositor re-derives dt and t - last_t from t (same operations)
                        t += dt;
                        nsamp++;
                        if (++cnt >= upto) break;
                    }
                    if (++probes >= RF_PROBES_PER_ROUND) break;
                }
            };
            if (RV_SLAB_STEPS > 0) {
                // Depth slab: every lane first finds its NEXT sample (crossing whatever empty space lies before it), the wave takes the nearest of
                // those as its front, and only lanes whose next sample lies within RV_SLAB_STEPS steps of the front sample this round; the others
                // put theirs back (t returns to it: one probe finds it again next round) and wait for the front to reach them.
                if (active) march(1);
                const float t_first = cnt ? smp_w[0].w : __builtin_inff();
                float tf = t_first;
                #pragma unroll
                for (int off = 1; off < 64; off <<= 1) tf = fminf(tf, __shfl_xor(tf, off, 64));
                t_lim = tf + (float)RV_SLAB_STEPS * K.dt_min;
                if (cnt && !(t_first < t_lim)) { t = t_first; nsamp--; cnt = 0; ended = false; }
                if (active && cnt && !ended) march(RV_S);
            } else if (active) {
                march(RV_S);
            }
        }

        // lanes hand samples to other lanes of the SAME wave through LDS: the LDS queue of a wave is in order, so only the
        // compiler has to be kept from reordering the accesses
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        RV_TICK(c_march)
        // ---- field evaluation: for each group of 16 rays, tile k = their k-th samples ----
#if RV_PIPELINE
        {
            // Software pipeline across tiles: ALL gathers of the next tile are issued before the MLP of the current one, so a
            // wave keeps the texture path busy while it sits in its 36-MFMA chain (two waves per SIMD do not overlap the
            // phases by themselves).  Costs 47 registers across the MLP (32 rows, 12 fractions, the position).
            // kmax[p] = most samples any ray of lane group p holds: tile (p, k) exists for k < kmax[p]
            int km = cnt;
            #pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int o = __shfl_xor(km, off, 64);
                km = o > km ? o : km;
            }
            int kmax[4];
            #pragma unroll
            for (int p = 0; p < 4; p++) kmax[p] = __builtin_amdgcn_readlane(km, 16 * p);
            auto first_group = [&](int p) { while (p < 4 && kmax[p < 4 ? p : 3] == 0) p++; return p; };
            const rf_lane_levels lv = lds_lv[g];
            rf_pair na, nb;                            // rows of the tile in flight
            int p = first_group(0), k = 0;
            auto issue = [&](int pp, int kk) {
                const int src = 16 * pp + s;
                float4 q = wave_smp[src * RV_S + kk];
                if (!(__shfl(cnt, src, 64) > kk)) q = make_float4(0.f, 0.f, 0.f, 0.f);
                float x0, x1, x2;
                rf_normalise(P, q.x, q.y, q.z, x0, x1, x2);
                rf_gather_pair<0>(P, lv, cls, x0, x1, x2, na);
                rf_gather_pair<1>(P, lv, cls, x0, x1, x2, nb);
            };
            if (p < 4) issue(p, 0);
            #pragma unroll 1
            while (p < 4) {
                const int src = 16 * p + s;
                const bool valid = __shfl(cnt, src, 64) > k;
                const ngp_h4 shq = *reinterpret_cast<const ngp_h4*>(wave_sh + src * 16 + 4 * g);
                n_tiles++;
                ngp_h8 x;
                rf_blend_pair(na, 0, x);
                rf_blend_pair(nb, 1, x);
                int p1 = p, k1 = k + 1;                // the next tile
                if (k1 >= kmax[p]) { p1 = first_group(p + 1); k1 = 0; }
                if (p1 < 4) issue(p1, k1);
                __builtin_amdgcn_sched_barrier(0);       // the prefetch stays above the MLP
                float a, b, c, d;
                rv_mlp_tile(lds_w, lane, x, shq, a, b, c, d);
                if (g == 0 && valid) {                   // the half-precision network outputs replace (x, y) of the slot
                    ngp_h4 r;
                    r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d;
                    *reinterpret_cast<ngp_h4*>(&wave_smp[src * RV_S + k]) = r;
                }
                p = p1; k = k1;
            }
        }
#else
#if RV_SORT_COLUMNS
        // Tile columns in order of decreasing sample count: lane group p evaluates the rays ranked 16p .. 16p+15, so the rays
        // of a group hold about the same number of samples and few columns of its tiles are dummies (a group runs
        // max-count tiles).  Counting sort over the 0..RV_S possible counts with ballots; `col_src` = the lane whose samples
        // column r evaluates.  Any assignment of rays to columns gives the same values (columns are independent).
        int col_src;
        {
            uint32_t rank = 0, above = 0;
            const uint32_t below_me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // lane id
            #pragma unroll
            for (int c = RV_S; c >= 0; c--) {
                const unsigned long long mc = __ballot(cnt == c);
                if (cnt == c) rank = above + (uint32_t)__popcll(mc & ((1ull << below_me) - 1ull));
                above += (uint32_t)__popcll(mc);
            }
            col_src = __builtin_amdgcn_ds_permute((int)(rank << 2), (int)below_me);
        }
#endif
        #pragma unroll 1
        for (int p = 0; p < 4; p++) {
#if RV_SORT_COLUMNS
            const int src = __shfl(col_src, 16 * p + s, 64);
#else
            const int src = 16 * p + s;
#endif
            const int ccol = __shfl(cnt, src, 64);
            if (__ballot(ccol > 0) == 0ull) continue;
            const ngp_h4 shq = *reinterpret_cast<const ngp_h4*>(wave_sh + src * 16 + 4 * g);
            const rf_lane_levels lv = lds_lv[g];
#if RV_TILE_PAIRS
            // tiles k and k+1 of the group together (the same 16 rays, consecutive samples): two MFMA chains per pass
            #pragma unroll 1
            for (int k = 0; k < RV_S; k += 2) {
                if (__ballot(ccol > k) == 0ull) break;   // counts only shrink with k
                if (k + 1 < RV_S && __ballot(ccol > k + 1) != 0ull) {
                    ngp_h8 xs[2];
                    #pragma unroll
                    for (int n = 0; n < 2; n++) {
                        float4 q = wave_smp[src * RV_S + k + n];
                        if (!(ccol > k + n)) q = make_float4(0.f, 0.f, 0.f, 0.f);
                        xs[n] = rf_encode<true>(P, lv, cls, q.x, q.y, q.z);
                    }
                    n_tiles += 2;
                    const ngp_h4 ss[2] = {shq, shq};
                    float a[2], b[2], c[2], d[2];
                    rv_mlp_tiles<2>(lds_w, lane, xs, ss, a, b, c, d);
                    #pragma unroll
                    for (int n = 0; n < 2; n++)
                        if (g == 0 && ccol > k + n) {
                            ngp_h4 r;
                            r[0] = (_Float16)a[n]; r[1] = (_Float16)b[n]; r[2] = (_Float16)c[n]; r[3] = (_Float16)d[n];
                            *reinterpret_cast<ngp_h4*>(&wave_smp[src * RV_S + k + n]) = r;
                        }
                } else {
                    float4 q = wave_smp[src * RV_S + k];
                    if (!(ccol > k)) q = make_float4(0.f, 0.f, 0.f, 0.f);
                    n_tiles++;
                    float a, b, c, d;
                    rv_field_tile(P, lv, cls, lds_w, lane, q.x, q.y, q.z, shq, a, b, c, d);
                    if (g == 0 && ccol > k) {
                        ngp_h4 r;
                        r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d;
                        *reinterpret_cast<ngp_h4*>(&wave_smp[src * RV_S + k]) = r;
                    }
                }
            }
#else
            #pragma unroll 1
            for (int k = 0; k < RV_S; k++) {
                if (__ballot(ccol > k) == 0ull) break;   // counts only shrink with k
                float4 q = wave_smp[src * RV_S + k];
                if (!(ccol > k)) q = make_float4(0.f, 0.f, 0.f, 0.f);   // column without a k-th sample: harmless dummy
                n_tiles++;
                float a, b, c, d;
                rv_field_tile(P, lv, cls, lds_w, lane, q.x, q.y, q.z, shq, a, b, c, d);
                if (g == 0 && ccol > k) {                // the half-precision network outputs replace (x, y) of the slot
                    ngp_h4 r;
                    r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d;
                    *reinterpret_cast<ngp_h4*>(&wave_smp[src * RV_S + k]) = r;
                }
            }
#endif
        }
#endif

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        RV_TICK(c_tiles)
        // ---- composite this lane's samples in order (kernel_composite_rays arithmetic, raymarching.cu:865-896) ----
        bool done = false;
        for (int k = 0; k < cnt; k++) {
            const ngp_h4 r = *reinterpret_cast<const ngp_h4*>(&my_smp[k]);
            // what the march computed for this sample (rv_point::at and kernel_march_rays, raymarching.cu:786-790), re-derived
            // from the sample's t instead of being stored: 16-byte slots leave room for a fifth sample per lane
            const float ts = my_smp[k].w;
            const float dt = ngp_clampf(ts * K.dt_gamma, K.dt_min, K.dt_max);
            const float ta = ts + dt;
            const float d1 = ta - last_t;
            last_t = ta;
            float sig = (float)r[0], sr = (float)r[1], sg = (float)r[2], sb = (float)r[3];
            rv_activate(P, sig, sr, sg, sb);
            n_samples_local++;
            const float alpha = 1.0f - ngp_expf(-sig * dt);
            const float T = 1 - ws;
            const float w = alpha * T;
            ws += w;
            tcomp += d1;
            dacc += w * tcomp;
            cr += w * sr; cg += w * sg; cb += w * sb;
            if ((double)T < 1e-4) {                          // samples marched beyond this one are discarded
                done = true;
#ifdef RV_COUNTERS
                n_probe[3] += (uint32_t)(cnt - (k + 1));
#endif
                break;
            }
        }
        float far0 = far;                                   // the ray's own far (the depth map's scale, the cap statistic): formed again, once per ray
        if (lds_occ && (done || ended)) {
            const float o3[3] = {m.ox, m.oy, m.oz}, d3[3] = {m.dx, m.dy, m.dz};
            float n0;
            ngp_near_far_inline(o3, d3, F.aabb, F.min_near, n0, far0);
        }
        const bool capped = !done && ended && nsamp >= F.max_steps && t < far0;
        if (ended) done = true;
        if (done) {
            F.image[3ull * ray] = cr + (1 - ws) * F.bg[0];
            F.image[3ull * ray + 1] = cg + (1 - ws) * F.bg[1];
            F.image[3ull * ray + 2] = cb + (1 - ws) * F.bg[2];
            F.depth[ray] = fmaxf(dacc - near, 0.0f) / (far0 - near);
            F.weights_sum[ray] = ws;
            n_capped_local += capped ? 1u : 0u;        // counted per lane, added to F.stats once when the wave retires
            n_hit_local += nsamp > 0 ? 1u : 0u;
            active = false;
        }
        RV_TICK(c_comp)
#ifdef RV_COUNTERS
        if (rv_hist) {                                  // debug timeline: per 20 us bin, samples / wave-rounds / live lanes / field-phase cycles
            const unsigned long long now = wall_clock64();
            const uint32_t bin = now > rv_t0 ? (uint32_t)((now - rv_t0) / 2000ull) : 0u;
            uint32_t rs = (uint32_t)cnt;
            #pragma unroll
            for (int off = 32; off > 0; off >>= 1) rs += __shfl_down(rs, off, 64);
            if (lane == 0 && bin < 512u) {
                atomicAdd(rv_hist + bin, rs); atomicAdd(rv_hist + 512 + bin, 1u);
                atomicAdd(rv_hist + 1024 + bin, (uint32_t)__popcll(rv_live));
            }
        }
#endif
    }
#ifdef RV_COUNTERS
    if (rv_hist && lane == 0) {
        const unsigned long long now = wall_clock64();
        const uint32_t bin = now > rv_t0 ? (uint32_t)((now - rv_t0) / 2000ull) : 0u;
        if (bin < 512u) atomicAdd(rv_hist + 1536 + bin, 1u);
    }
#endif
    uint32_t tot = n_samples_local, tot_capped = n_capped_local, tot_hit = n_hit_local;
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tot += __shfl_down(tot, off, 64);
        tot_capped += __shfl_down(tot_capped, off, 64);
        tot_hit += __shfl_down(tot_hit, off, 64);
    }
    if (lane == 0 && tot) atomicAdd(F.stats, tot);
    if (lane == 0 && tot_capped) atomicAdd(F.stats + 1, tot_capped);
    if (lane == 0 && tot_hit) atomicAdd(F.stats + 2, tot_hit);
    if (lane == 0 && n_tiles) atomicAdd(F.stats + 3, n_tiles);
#ifdef RV_COUNTERS
    if (lane == 0) { atomicAdd(F.queue + 1, n_rounds); atomicAdd(F.queue + 2, n_trips); }
    atomicAdd(F.queue + 3, n_probe[0]); atomicAdd(F.queue + 4, n_probe[1]); atomicAdd(F.queue + 5, n_probe[2]); atomicAdd(F.queue + 6, n_probe[3]);
    if (lane == 0) {                                   // cycle counters: u64 at bytes 32.. of the workspace header
        unsigned long long* c = reinterpret_cast<unsigned long long*>(F.queue + 8);
        const unsigned long long c_total = __builtin_readcyclecounter() - c_start;
        atomicAdd(c + 0, c_refill); atomicAdd(c + 1, c_march); atomicAdd(c + 2, c_tiles); atomicAdd(c + 3, c_comp);
        atomicAdd(c + 4, c_total); atomicMax(c + 5, c_total); atomicMax(c + 6, ~c_total);
    }
#endif
}

__global__ __launch_bounds__(RV_BLOCK, RV_WAVES_PER_SIMD) void k_render_frame_multi(rf_params P, rf_frame F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rv_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rv_smem);
    _Float16* lds_sh = reinterpret_cast<_Float16*>(rv_smem + RV_LDS_W);
    rf_lane_levels* lds_lv = reinterpret_cast<rf_lane_levels*>(rv_smem + RV_LDS_W + RV_LDS_SH);
    float4* lds_smp = reinterpret_cast<float4*>(rv_smem + RV_LDS_W + RV_LDS_SH + RV_LDS_LV + RV_LDS_CHUNK);
    uint32_t* lds_coarse = F.coarse ? reinterpret_cast<uint32_t*>(rv_smem + RV_LDS_W + RV_LDS_SH + RV_LDS_LV + RV_LDS_CHUNK + RV_LDS_SMP) : nullptr;
    unsigned long long* rv_chunk_p = reinterpret_cast<unsigned long long*>(rv_smem + RV_LDS_W + RV_LDS_SH + RV_LDS_LV);   // (RV_CU_CHUNKS only)

    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15, wave = threadIdx.x >> 6;
#if RV_CU_CHUNKS
    if (threadIdx.x == 0) *rv_chunk_p = 8ull << 32;    // {first tile of the CU's current chunk : 32 | tiles taken : 8 | band : 24}; "all 8 taken": the first wave to ask fetches one
#endif

    rv_stage_weights(P, lds_w, wave, RV_WAVES, lane);
    if (lds_coarse) {
        const uint32_t nw = F.coarse_words * F.C;
        for (uint32_t i = threadIdx.x; i < nw; i += RV_BLOCK) lds_coarse[i] = F.coarse[i];
    }
    // the box of everything occupied, in world units, one unit (a 4^3 block of cascade 0) wider on every side than the blocks say, + how far behind it a march may
    // still look (two of the largest steps): 7 floats behind the coarse map
    float* lds_occ = (lds_coarse && F.occ_ext) ? reinterpret_cast<float*>(lds_coarse + F.coarse_words * F.C) : nullptr;
    if (lds_occ && threadIdx.x < 3) {
        const uint32_t k = threadIdx.x;
        const uint32_t hi = F.occ_ext[3 + k], lo = (1u << 20) - F.occ_ext[k];
        const bool any = F.occ_ext[3] != 0u;                       // nothing occupied at all: an empty box far away, every ray misses it
        lds_occ[k] = any ? -F.occ_top + ((float)lo - 1.0f) * F.occ_unit : 3.0e38f;
        lds_occ[3 + k] = any ? -F.occ_top + ((float)hi + 1.0f) * F.occ_unit : 3.1e38f;
        if (k == 0) lds_occ[6] = 2.0f * (((2.0f * 1.7320508075688772f) * (float)(1 << (F.C - 1))) / (float)F.H);
    }
    if (wave == 0 && s == 0) {
        rf_lane_levels tmp;
        rf_setup_levels(P, g, tmp);
        lds_lv[g] = tmp;
    }
    __syncthreads();
    const rf_iter_class cls = rf_classify(lds_lv[g]);
    if (cls.dense == 1u && cls.select == 2u && cls.hashed == 12u)
        rv_frame_loop<true>(P, F, cls, lds_w, lds_sh, lds_lv, lds_smp, lds_coarse, rv_chunk_p, lds_occ);
    else
        rv_frame_loop<false>(P, F, cls, lds_w, lds_sh, lds_lv, lds_smp, lds_coarse, rv_chunk_p, lds_occ);
}
#endif  // RV_S > 1

static inline bool rv_pow2(uint32_t v) { return v && !(v & (v - 1)); }

// process-wide validation switch; atomic because callers may render from several host threads (one stream each)
static std::atomic<int> rv_occ_box_enabled{1};
// validation switch: 0 = every ray marches to its own far (as before round 4); results are identical either way
extern "C" int ngp_render_set_occupied_box(int enabled) { return rv_occ_box_enabled.exchange(enabled ? 1 : 0, std::memory_order_relaxed); }
static std::atomic<int> rv_block_skip_enabled{1};
extern "C" int ngp_render_set_block_skip(int enabled) { return rv_block_skip_enabled.exchange(enabled ? 1 : 0, std::memory_order_relaxed); }

static constexpr size_t RV_WS_COARSE = 256, RV_WS_TILES = 256 + 48 * 1024;   // header: global queue + debug words | 8 band queues
extern "C" size_t ngp_render_frame_workspace(uint32_t N) {
    // ray queue | coarse occupancy map (<= 48 KiB) | per-tile estimates and tile order (one u32 each per 64 rays)
    return RV_WS_TILES + 2 * sizeof(uint32_t) * (size_t)ngp_div_up(N, 64u);
}

static int rv_fill_camera(const char* who, const float* pose_host, const float* intrinsics_host, uint32_t H, uint32_t W, ngp_camera& cam) {
    NGP_REQUIRE(pose_host && intrinsics_host, "camera: pose / intrinsics are host pointers and must not be null");
    NGP_REQUIRE(H >= 1 && W >= 1 && (uint64_t)H * W <= 0xFFFFFFFFull, "camera: bad image size");
    NGP_REQUIRE(intrinsics_host[0] != 0.0f && intrinsics_host[1] != 0.0f, "camera: zero focal length");
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) cam.r[3 * r + c] = pose_host[4 * r + c];
        cam.t[r] = pose_host[4 * r + 3];
    }
    cam.fx = intrinsics_host[0]; cam.fy = intrinsics_host[1]; cam.cx = intrinsics_host[2]; cam.cy = intrinsics_host[3];
    cam.W = W; cam.H = H;
    (void)who;
    return NGP_OK;
}

static int rv_render_frame(const ngp_field_t* field_host, const float* rays_o, const float* rays_d, const ngp_camera* cam, uint32_t n_cams, uint32_t N,
                           uint32_t image_width, const float* aabb_host, float min_near, const uint8_t* bitfield, uint32_t C, uint32_t Hgrid,
                           float dt_gamma, uint32_t max_steps, const float* bg_color3_host,
                           float* image, float* depth, float* weights_sum, uint32_t* stats,
                           void* workspace, size_t workspace_bytes, void* stream) {
    rf_params P;
    int rc = rf_fill_params("render_frame", field_host, P);
    if (rc != NGP_OK) return rc;
    NGP_REQUIRE(stats && workspace && workspace_bytes >= RV_WS_COARSE, "render_frame: stats / workspace missing");
    NGP_REQUIRE(aabb_host && bg_color3_host, "render_frame: aabb / bg_color are host pointers and must not be null");
    NGP_REQUIRE(C >= 1 && C <= 16 && Hgrid >= 1 && Hgrid <= 1024 && max_steps >= 1, "render_frame: bad C/H/max_steps");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(stats, 0, 4 * sizeof(uint32_t), s) != hipSuccess || hipMemsetAsync(workspace, 0, RV_WS_COARSE, s) != hipSuccess)
        return ngp_fail(NGP_ELAUNCH, "render_frame: memset failed");
    if (N == 0) return NGP_OK;
    NGP_REQUIRE((cam || (rays_o && rays_d)) && bitfield && image && depth && weights_sum, "render_frame: null pointer");
    rf_frame F;
    F.rays_o = cam ? nullptr : rays_o; F.rays_d = cam ? nullptr : rays_d; F.N = N;
    F.cam = cam ? *cam : ngp_camera{};
    F.cams = nullptr; F.frame_rays = N;
    if (cam && n_cams > 1) {
        // the cameras travel in the workspace, behind the coarse occupancy map and the tile area (pageable host memory: the copy is staged before the call returns)
        const size_t at = (ngp_render_frame_workspace(N) + 255) & ~(size_t)255;
        NGP_REQUIRE(workspace_bytes >= at + n_cams * sizeof(ngp_camera), "render_frames_camera: workspace too small (ngp_render_frames_workspace)");
        ngp_camera* dst = reinterpret_cast<ngp_camera*>(reinterpret_cast<unsigned char*>(workspace) + at);
        if (hipMemcpyAsync(dst, cam, n_cams * sizeof(ngp_camera), hipMemcpyHostToDevice, s) != hipSuccess) return ngp_fail(NGP_ELAUNCH, "render_frames_camera: camera copy failed");
        F.cams = dst; F.frame_rays = N / n_cams;
    }
    for (int i = 0; i < 6; i++) F.aabb[i] = aabb_host[i];
    F.min_near = min_near; F.bitfield = bitfield; F.C = C; F.H = Hgrid;
    F.dt_gamma = dt_gamma; F.max_steps = max_steps;
    for (int i = 0; i < 3; i++) F.bg[i] = bg_color3_host[i];
    F.image = image; F.depth = depth; F.weights_sum = weights_sum;
    F.stats = stats; F.queue = (uint32_t*)workspace;
#ifdef RV_COUNTERS
    F.hist = nullptr;
#ifdef RV_TIMELINE                                      // the timeline's atomics perturb the cycle counters: a build of its own
    if (workspace_bytes >= RV_WS_TILES + 2048 * sizeof(uint32_t)) {
        F.hist = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(workspace) + RV_WS_TILES);
        if (hipMemsetAsync(F.hist, 0, 2048 * sizeof(uint32_t), s) != hipSuccess) return ngp_fail(NGP_ELAUNCH, "render_frame: memset failed");
    }
#endif
#endif
    F.tile_w = 0;
    const uint32_t hint = image_width;
    if (hint >= 8 && hint % 8 == 0 && N % hint == 0 && (N / hint) % 8 == 0) F.tile_w = hint;

    // coarse occupancy (needs Morton blocks: H a power of two >= 4) in the workspace, then in LDS
    F.coarse = nullptr; F.coarse_words = 0; F.skip = 0; F.occ_ext = nullptr; F.occ_unit = 0.0f; F.occ_top = 0.0f;
    static_assert(sizeof(rf_lane_levels) * 4 == RV_LDS_LV, "LDS carve of the level table");
    size_t lds = RV_LDS_W + RV_LDS_SH + RV_LDS_LV;
#if RV_S > 1
    lds += RV_LDS_SMP + RV_LDS_CHUNK;
    const void* kernel = reinterpret_cast<const void*>(k_render_frame_multi);
#else
    const void* kernel = reinterpret_cast<const void*>(k_render_frame);
#endif
    const uint64_t blocks_per_level = (uint64_t)Hgrid * Hgrid * Hgrid / 64;
    const uint64_t coarse_bytes = (uint64_t)C * blocks_per_level / 8;
    // (C * H^3 <= 2^24: the reference forms the cell index in binary32, raymarching.cu:783; beyond that it rounds)
    // and the map has to fit in the LDS left beside the weights, SH and sample slots (8 KiB for 2 cascades of 128^3; from 3
    // cascades on it does not: such a frame is marched without the map, i.e. cell by cell through the bitfield itself)
    if (rv_pow2(Hgrid) && Hgrid >= 8 && blocks_per_level % 32 == 0 && lds + coarse_bytes <= 160 * 1024 && (uint64_t)C * Hgrid * Hgrid * Hgrid <= (1ull << 24) &&
        workspace_bytes >= RV_WS_COARSE + coarse_bytes && (reinterpret_cast<uintptr_t>(bitfield) & 7u) == 0) {
        uint32_t* coarse = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(workspace) + RV_WS_COARSE);
        const uint32_t n_blocks_total = (uint32_t)(C * blocks_per_level);
        F.coarse = coarse;
        F.coarse_words = (uint32_t)(blocks_per_level / 32);
        lds += coarse_bytes;
        // block skipping and the occupied box need the cascades nested in powers of two (see below)
        uint32_t* ext = nullptr;
#if RV_S > 1
        int e2;
        const bool nested = Hgrid >= 64 && (C == 1 || frexpf(field_host->bound, &e2) == 0.5f);
        if (nested && rv_occ_box_enabled.load(std::memory_order_relaxed) && lds + 32 <= 160 * 1024) {
            ext = reinterpret_cast<uint32_t*>(workspace) + 26;          // header words 26..31 (zeroed above)
            F.occ_ext = ext;
            F.occ_top = C == 1 ? field_host->bound : (float)(1u << (C - 1));
            F.occ_unit = 2.0f * (C == 1 ? field_host->bound : 1.0f) / (float)(Hgrid / 4);
            lds += 32;
        }
#endif
        hipLaunchKernelGGL(k_build_coarse, dim3(ngp_div_up(n_blocks_total / 32, 256)), dim3(256), 0, s, bitfield, n_blocks_total, coarse,
                           (uint32_t)blocks_per_level, C, ext);
        // block skipping needs the 16^3 blocks aligned with the cascade boundaries (cells H/4 and 3H/4 of the next level) and
        // every level's half-width a power of two: H a power of two >= 64, and bound a power of two unless there is one cascade
        int e;
        F.skip = (rv_block_skip_enabled.load(std::memory_order_relaxed) && Hgrid >= 64 && (C == 1 || frexpf(field_host->bound, &e) == 0.5f)) ? 1u : 0u;
    }
    NGP_REQUIRE(lds <= 160 * 1024, "render_frame: LDS carve exceeds 160 KiB");
    F.tile_order = nullptr;
#if RV_TILE_ORDER
    if (F.tile_w && F.coarse && workspace_bytes >= ngp_render_frame_workspace(N)) {
        const uint32_t n_tiles = N / 64;
        uint32_t* est = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(workspace) + RV_WS_TILES);
        uint32_t* order = est + n_tiles;
        hipLaunchKernelGGL(k_tile_estimate, dim3(ngp_div_up(n_tiles, 256u)), dim3(256), 0, s, F, P.bound, n_tiles, est);
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, est, n_tiles, order);
        F.tile_order = order;
    }
#endif
    // the raised dynamic-LDS limit is a per-device function attribute: set it once on every device this process renders on
    static std::atomic<unsigned long long> attr_devices{0};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0) return ngp_fail(NGP_ELAUNCH, "render_frame: no current device");
    const unsigned long long device_bit = 1ull << (device & 63);
    if (device >= 64 || !(attr_devices.load(std::memory_order_acquire) & device_bit)) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return ngp_fail(NGP_ELAUNCH, "render_frame: cannot raise the dynamic LDS limit");
        attr_devices.fetch_or(device_bit, std::memory_order_release);
    }
    // persistent grid: RV_BLOCKS_PER_CU workgroups per CU, fewer when the frame is small
    uint32_t blocks = 256 * RV_BLOCKS_PER_CU;
    const uint32_t need = ngp_div_up(N, RV_BLOCK);
    if (blocks > need) blocks = need;
#if RV_S > 1
    hipLaunchKernelGGL(k_render_frame_multi, dim3(blocks), dim3(RV_BLOCK), lds, s, P, F);
#else
    hipLaunchKernelGGL(k_render_frame, dim3(blocks), dim3(RV_BLOCK), lds, s, P, F);
#endif
    NGP_CHECK_LAUNCH("render_frame");
    return NGP_OK;
}

extern "C" int ngp_render_frame(const ngp_field_t* field_host, const float* rays_o, const float* rays_d, uint32_t N,
                                uint32_t image_width, const float* aabb_host, float min_near, const uint8_t* bitfield, uint32_t C, uint32_t Hgrid,
                                float dt_gamma, uint32_t max_steps, const float* bg_color3_host,
                                float* image, float* depth, float* weights_sum, uint32_t* stats,
                                void* workspace, size_t workspace_bytes, void* stream) {
    return rv_render_frame(field_host, rays_o, rays_d, nullptr, 0, N, image_width, aabb_host, min_near, bitfield, C, Hgrid, dt_gamma, max_steps,
                           bg_color3_host, image, depth, weights_sum, stats, workspace, workspace_bytes, stream);
}

// The same frame from a camera instead of ray arrays: get_rays (nerf/utils.py:53-116, full-image branch) runs inside the
// kernel's refill, one ray per pixel in row-major order (ngp_camera.h); bit-identical to ngp_get_rays + ngp_render_frame.
extern "C" int ngp_render_frame_camera(const ngp_field_t* field_host, const float* pose_host, const float* intrinsics_host, uint32_t H, uint32_t W,
                                       const float* aabb_host, float min_near, const uint8_t* bitfield, uint32_t C, uint32_t Hgrid,
                                       float dt_gamma, uint32_t max_steps, const float* bg_color3_host,
                                       float* image, float* depth, float* weights_sum, uint32_t* stats,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    ngp_camera cam;
    const int rc = rv_fill_camera("render_frame_camera", pose_host, intrinsics_host, H, W, cam);
    if (rc != NGP_OK) return rc;
    return rv_render_frame(field_host, nullptr, nullptr, &cam, 1, H * W, W, aabb_host, min_near, bitfield, C, Hgrid, dt_gamma, max_steps,
                           bg_color3_host, image, depth, weights_sum, stats, workspace, workspace_bytes, stream);
}

// P frames in ONE launch: poses_host [P][16] (row-major 4x4, cam2world), one set of intrinsics; outputs [P][H*W] contiguous.  The frame kernel's
// ramp (256 workgroups staging 36 KiB of weights, the first tiles marching) and drain (the last waves finishing alone) are paid once per
// launch instead of once per frame: the same pixels, bit for bit, as P calls of ngp_render_frame_camera.  H and W must be multiples of 8
// (an 8x8 tile then never straddles two frames).  Workspace: ngp_render_frames_workspace(P, H * W).
extern "C" size_t ngp_render_frames_workspace(uint32_t P, uint32_t rays_per_frame) {
    return ((ngp_render_frame_workspace(P * rays_per_frame) + 255) & ~(size_t)255) + (size_t)P * sizeof(ngp_camera);
}

extern "C" int ngp_render_frames_camera(const ngp_field_t* field_host, const float* poses_host, uint32_t P, const float* intrinsics_host, uint32_t H, uint32_t W,
                                        const float* aabb_host, float min_near, const uint8_t* bitfield, uint32_t C, uint32_t Hgrid,
                                        float dt_gamma, uint32_t max_steps, const float* bg_color3_host,
                                        float* image, float* depth, float* weights_sum, uint32_t* stats,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    NGP_REQUIRE(P >= 1 && P <= 64, "render_frames_camera: 1..64 frames per launch");
    NGP_REQUIRE(H % 8 == 0 && W % 8 == 0, "render_frames_camera: H and W must be multiples of 8");
    NGP_REQUIRE((uint64_t)P * H * W <= 0xFFFFFFFFull, "render_frames_camera: too many rays");
    NGP_REQUIRE(poses_host, "render_frames_camera: poses is a host pointer and must not be null");
    ngp_camera cams[64];
    for (uint32_t p = 0; p < P; p++) {
        const int rc = rv_fill_camera("render_frames_camera", poses_host + 16 * p, intrinsics_host, H, W, cams[p]);
        if (rc != NGP_OK) return rc;
    }
    return rv_render_frame(field_host, nullptr, nullptr, cams, P, P * H * W, W, aabb_host, min_near, bitfield, C, Hgrid, dt_gamma, max_steps,
                           bg_color3_host, image, depth, weights_sum, stats, workspace, workspace_bytes, stream);
}

// get_rays as an op of its own (nerf/utils.py:53-116): ray k belongs to pixel inds[k] (row-major, inds on the device) or to
// pixel k when inds is null (then N must be H * W).  rays_o, rays_d: [N, 3].
__global__ __launch_bounds__(256) void k_get_rays(ngp_camera cam, const int64_t* __restrict__ inds, uint32_t N,
                                                  float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    uint32_t pixel = k;
    if (inds) {
        const int64_t p = inds[k];
        const int64_t last = (int64_t)cam.W * cam.H - 1;
        pixel = (uint32_t)(p < 0 ? 0 : p > last ? last : p);      // torch.gather would raise; out-of-range indices are clamped
    }
    float d[3];
    ngp_camera_ray(cam, pixel, d);
    #pragma unroll
    for (int c = 0; c < 3; c++) { rays_o[3ull * k + c] = cam.t[c]; rays_d[3ull * k + c] = d[c]; }
}

extern "C" int ngp_get_rays(const float* pose_host, const float* intrinsics_host, uint32_t H, uint32_t W,
                            const int64_t* inds, uint32_t N, float* rays_o, float* rays_d, void* stream) {
    ngp_camera cam;
    const int rc = rv_fill_camera("get_rays", pose_host, intrinsics_host, H, W, cam);
    if (rc != NGP_OK) return rc;
    if (N == 0) return NGP_OK;
    NGP_REQUIRE(inds || (uint64_t)N == (uint64_t)H * W, "get_rays: N must be H * W when inds is null");
    NGP_REQUIRE(rays_o && rays_d, "get_rays: null pointer");
    hipLaunchKernelGGL(k_get_rays, dim3(ngp_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, cam, inds, N, rays_o, rays_d);
    NGP_CHECK_LAUNCH("get_rays");
    return NGP_OK;
}
