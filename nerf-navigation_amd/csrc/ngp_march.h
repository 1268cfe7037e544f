// ngp_march.h -- the march arithmetic shared by the per-op kernels (raymarching.hip, through ngp_march_t in ngp_device.h's
// successor below) and the fused frame kernel (render_fused.hip): one sample point, the step out of an empty cell, and the
// verified skip through an empty block.  Templates over the "view" type V, which provides the ray (ox..dz, rdx..rdz) and the
// march constants (bound, rbound, dt_gamma, dt_min, dt_max, rH, Hf, Hm1, mip()).
#pragma once
#include "ngp_device.h"

#ifndef NGP_SKIP_WALK
#define NGP_SKIP_WALK 32               // dt_gamma > 0: steps a skip may walk (a varying step has no closed form)
#endif

// One march sample point: everything kernel_march_rays derives from t (raymarching.cu:748-781), same arithmetic.
template <class V>
struct ngp_point {
    float x, y, z, dt, mip_bound;
    int level, nx, ny, nz;
    __device__ __forceinline__ void at(const V& m, float tc) {
        x = ngp_clampf(m.ox + tc * m.dx, -m.bound, m.bound);
        y = ngp_clampf(m.oy + tc * m.dy, -m.bound, m.bound);
        z = ngp_clampf(m.oz + tc * m.dz, -m.bound, m.bound);
        dt = ngp_clampf(tc * m.dt_gamma, m.dt_min, m.dt_max);
        int e_pos, e_dt;
        (void)frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e_pos);
        (void)frexpf((dt * m.Hf) * 0.5f, &e_dt);
        const int lp = m.mip(e_pos), ld = m.mip(e_dt);
        level = lp > ld ? lp : ld;
        const float p2 = (float)(1 << level);
        mip_bound = fminf(p2, m.bound);
        // 1 / mip_bound without a division per probe: 2^-level is exact and 1 / bound is the same IEEE quotient, computed once
        const float mip_rbound = (p2 <= m.bound) ? __builtin_ldexpf(1.0f, -level) : m.rbound;
        nx = (int)ngp_clampf(((x * mip_rbound + 1.0f) * 0.5f) * m.Hf, 0.0f, m.Hm1);
        ny = (int)ngp_clampf(((y * mip_rbound + 1.0f) * 0.5f) * m.Hf, 0.0f, m.Hm1);
        nz = (int)ngp_clampf(((z * mip_rbound + 1.0f) * 0.5f) * m.Hf, 0.0f, m.Hm1);
    }
    // ray parameter at which the ray leaves this point's cell (raymarching.cu:792-797)
    __device__ __forceinline__ float cell_exit(const V& m, float tc) const {
        const float tx = (((((float)nx + 0.5f + 0.5f * copysignf(1.0f, m.dx)) * m.rH) * 2.0f - 1.0f) * mip_bound - x) * m.rdx;
        const float ty = (((((float)ny + 0.5f + 0.5f * copysignf(1.0f, m.dy)) * m.rH) * 2.0f - 1.0f) * mip_bound - y) * m.rdy;
        const float tz = (((((float)nz + 0.5f + 0.5f * copysignf(1.0f, m.dz)) * m.rH) * 2.0f - 1.0f) * mip_bound - z) * m.rdz;
        return tc + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    // the same for the aligned block of 2^sh cells per axis around the cell (an estimate: only used to choose a skip target)
    __device__ __forceinline__ float block_exit(const V& m, float tc, int sh) const {
        const float bs = (float)(1 << sh);
        const float tx = ((((((float)(nx >> sh) + 0.5f + 0.5f * copysignf(1.0f, m.dx)) * bs) * m.rH) * 2.0f - 1.0f) * mip_bound - x) * m.rdx;
        const float ty = ((((((float)(ny >> sh) + 0.5f + 0.5f * copysignf(1.0f, m.dy)) * bs) * m.rH) * 2.0f - 1.0f) * mip_bound - y) * m.rdy;
        const float tz = ((((((float)(nz >> sh) + 0.5f + 0.5f * copysignf(1.0f, m.dz)) * bs) * m.rH) * 2.0f - 1.0f) * mip_bound - z) * m.rdz;
        return tc + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
};


// Verified skips through empty blocks of the occupancy grid (ngp_try_skip).
//
// The reference leaves an empty cell by stepping t += dt until t passes the cell's exit (raymarching.cu:792-801).  The
// step depends on t alone, so the points t_0 = near, t_{k+1} = t_k + dt(t_k) form a fixed lattice per ray; the reference
// tests the first lattice point in every cell that holds one, and every lattice point of an occupied cell is a sample.
// When the whole 4^3 (or 16^3) block around an empty cell is empty, those tests can only fail, so the march may jump over
// them -- provided it rejoins the reference's sequence of tested points exactly.  It does so as follows (r = current
// point, all in the reference's own arithmetic):
//   1. move along the lattice to a point s short of the block's exit by a guard distance (constant step: in closed form,
//      ngp_lattice_jump; dt_gamma > 0: a bounded walk);
//   2. evaluate s as the reference would (level, cell): s must lie in the same block at the same level.  Cell indices are
//      monotone in t on every axis, so every lattice point between r and s then lies in that block as well, and block
//      alignment with the cascade boundaries (H a power of two >= 64, bound a power of two or a single cascade: decided
//      on the host, F.skip) keeps those points on the same level: whatever the reference tested in between was empty;
//   3. take the reference's step from s to the first lattice point a beyond the exit e(s) of s's cell.  The reference
//      arrives in that cell at some lattice point p (p <= s, or the point after s when s sits on the cell's entry face)
//      and steps from there beyond e(p).  e(p) and e(s) measure the same face; they differ by rounding only, by less than
//      the per-ray bound M (see ngp_skip_margin).  If no lattice point lies within M of e(s), both steps end on the same
//      point a, which is therefore a point the reference tests: the skip is accepted and the march continues from a;
//   4. in every other case the probe takes the reference's ordinary one-cell step from r.
// Pinned by: the per-op kernels (which use it too) against the CPU oracle, which marches cell by cell, bit for bit
// (tests/test_gpu_raymarching.py); the fused kernel with the switch on against off on 4 x 640,000 rays
// (tests/test_gpu_fullsize.py).
// Lattice points of a CONSTANT step inside one binade are equally spaced.  With u the binade's ulp, t = T u and
// dt = (c + f) u, |f| < 1/2, the sum t + dt rounds to (T + c) u whatever T is, as long as it stays below the binade's end:
// k steps from t land exactly on t + k (c u), and c u is what one real step adds.  (|f| = 1/2 would round to even and
// alternate; it is detected from the step's rounding error and not used.)  Returns the lattice point reached from t by
// one real step plus as many whole steps as stay below min(lim, end of t's binade): every operation here is exact.
__device__ __forceinline__ float ngp_lattice_jump(float t, float dtc, float lim) {
    const float t1 = t + dtc;                          // the reference's own step
    const float du = t1 - t;                           // exact (Sterbenz): what that step added
    int e;
    (void)frexpf(t, &e);                               // t in [2^(e-1), 2^e), ulp 2^(e-24)
    const float end = fminf(lim, __builtin_ldexpf(1.0f, e));
    const float err = dtc - du;                        // exact: rounding error of t + dtc
    if (!(t1 < end) || !(du > 0.0f) || fabsf(err) == __builtin_ldexpf(1.0f, e - 25)) return t1;
    // j <= (end - t1) / du - 1 keeps t1 + j du below `end` whatever the rounding of the estimate (relative error ~1e-6)
    const float j = floorf((end - t1) * __builtin_amdgcn_rcpf(du)) - 1.0f;
    return j > 0.0f ? t1 + j * du : t1;                // j du < 2^(e-1) is a multiple of u: both operations exact
}

// The reference's step out of an empty cell (raymarching.cu:798-801): do { t += dt(t); } while (t < tt).  Returns the first
// lattice point >= tt after at least one step, and in `prev` the lattice point before it.  With a constant step most of the
// way is one exact multiply-add (ngp_lattice_jump lands strictly below tt, or on the single step t + dt); the last steps
// are real steps, so the result is the reference's bit for bit.
template <class V>
__device__ __forceinline__ float ngp_advance(const V& k, float t, float tt, float& prev) {
    float tn;
    int guard = 0;
    if (k.dt_gamma == 0.0f) {
        const float dtc = ngp_clampf(0.0f, k.dt_min, k.dt_max);
        prev = t;
        tn = ngp_lattice_jump(t, dtc, tt);
        while (tn < tt && ++guard < NGP_SKIP_GUARD) { prev = tn; tn += dtc; }
    } else {
        tn = t;
        do {
            prev = tn;
            tn += ngp_clampf(tn * k.dt_gamma, k.dt_min, k.dt_max);
        } while (tn < tt && ++guard < NGP_SKIP_GUARD);
    }
    return tn;
}

// The skip attempt for the empty cell of point r (tested at tc, ordinary exit tt) inside an empty block of 2^sh cells per
// axis.  Returns the lattice point to continue from, or a negative number when the probe must take the ordinary step.
template <class V>
__device__ __forceinline__ float ngp_try_skip(const V& m, const ngp_point<V>& r, float tc, float tt, int sh, float M) {
    const float tb = r.block_exit(m, tc, sh);
    const float target = tb - (2.0f * ngp_clampf(tb * m.dt_gamma, m.dt_min, m.dt_max) + M);
    if (!(target > tt)) return -1.0f;                  // also when M is inf or NaN: such rays never skip
    // step 1: s only has to be a lattice point inside the block, the further the better
    float ts = tc;
    if (m.dt_gamma == 0.0f) ts = ngp_lattice_jump(tc, ngp_clampf(0.0f, m.dt_min, m.dt_max), target);
    else {
        int guard = 0;
        do {
            ts += ngp_clampf(ts * m.dt_gamma, m.dt_min, m.dt_max);
        } while (ts < target && ++guard < NGP_SKIP_WALK);
    }
    ngp_point<V> q;
    q.at(m, ts);
    if (q.level == r.level && (q.nx >> sh) == (r.nx >> sh) && (q.ny >> sh) == (r.ny >> sh) && (q.nz >> sh) == (r.nz >> sh)) {
        const float te = q.cell_exit(m, ts);
        float tp;
        const float ta = ngp_advance(m, ts, te, tp);
        if ((te - tp) > M && (ta - te) > M) return ta;
    }
    return -1.0f;
}

// Bound on how far two evaluations of one cell face's ray parameter (ngp_point::cell_exit from two points of the cell)
// can differ.  cell_exit = t + (plane - x(t)) * (1/d): the plane is exact (H and the cascade bound are powers of two),
// x(t) = o + t d carries two roundings of magnitude <= 2^-24 (|o| + 2 |x|) <= 2^-24 (|o| + 2 bound), the difference one more,
// the product with 1/d two more, the final sum one of 2^-24 t.  The bound below is 8x that estimate; a ray with a
// vanishing direction component gets M = inf and never skips.
template <class V>
__device__ __forceinline__ float ngp_skip_margin(const V& r, float bound, float far) {
    const float pos = fmaxf(fabsf(r.ox), fmaxf(fabsf(r.oy), fabsf(r.oz))) + 4.0f * bound;
    const float rd = fmaxf(fabsf(r.rdx), fmaxf(fabsf(r.rdy), fabsf(r.rdz)));
    return 4.8e-7f * (pos * rd + fabsf(far));           // 8 * 2^-24 = 4.8e-7
}

// ---------------------------------------------------------------------------
// The box of everything occupied (round 4).  Beyond it every cell of the grid is empty: the reference tests those cells and finds nothing, so a march that
// stops a little behind the box produces the same samples -- and a ray does not walk out of the scene through dozens of empty probes (46 empty probes per ray
// in the 800x800 frame, most of them behind the last sample: profiles/HISTORY.md 4.1).  The extent is kept on one integer lattice for all cascades: unit = a
// 4^3 block of cascade 0, origin = the low corner of the outermost cascade; cascade l: a block is 2^l units and its box starts (2^(C-1) - 2^l) nb / 2 units in
// (nb = blocks per axis) -- exact when the cascades nest in powers of two, the condition of the verified skips (ngp_skip_allowed).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ngp_unmorton(uint32_t v) {          // every third bit of a Morton index, packed
    v &= 0x09249249u;
    v = (v | (v >> 2)) & 0x030C30C3u;
    v = (v | (v >> 4)) & 0x0300F00Fu;
    v = (v | (v >> 8)) & 0x030000FFu;
    v = (v | (v >> 16)) & 0x000003FFu;
    return v;
}

// word = 32 coarse bits = the Morton blocks [32 wi, 32 wi + 32) of cascade `level`: an aligned group of 4 x 4 x 2 blocks (bit i: x = bits 0, 3 of i, y = bits 1, 4,
// z = bit 2).  Widens lo / hi (lattice units) by the set blocks of the word; no loop over the bits.
__device__ __forceinline__ void ngp_occ_extent_word(uint32_t word, uint32_t wi, uint32_t level, uint32_t C, uint32_t nb, uint32_t (&lo)[3], uint32_t (&hi)[3]) {
    if (!word) return;
    const uint32_t m = wi * 32u;
    const uint32_t base[3] = {ngp_unmorton(m), ngp_unmorton(m >> 1), ngp_unmorton(m >> 2)};
    const uint32_t X[4] = {0x00550055u, 0x00AA00AAu, 0x55005500u, 0xAA00AA00u}, Y[4] = {0x00003333u, 0x0000CCCCu, 0x33330000u, 0xCCCC0000u};
    uint32_t mn[3], mx[3];
    mn[0] = (word & X[0]) ? 0u : (word & X[1]) ? 1u : (word & X[2]) ? 2u : 3u;
    mx[0] = (word & X[3]) ? 3u : (word & X[2]) ? 2u : (word & X[1]) ? 1u : 0u;
    mn[1] = (word & Y[0]) ? 0u : (word & Y[1]) ? 1u : (word & Y[2]) ? 2u : 3u;
    mx[1] = (word & Y[3]) ? 3u : (word & Y[2]) ? 2u : (word & Y[1]) ? 1u : 0u;
    mn[2] = (word & 0x0F0F0F0Fu) ? 0u : 1u;
    mx[2] = (word & 0xF0F0F0F0u) ? 1u : 0u;
    const uint32_t off = (((1u << (C - 1u)) - (1u << level)) * nb) >> 1;
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t p = off + ((base[k] + mn[k]) << level), q = off + ((base[k] + mx[k] + 1u) << level);
        lo[k] = p < lo[k] ? p : lo[k];
        hi[k] = q > hi[k] ? q : hi[k];
    }
}

// Whether block skipping is exact for a grid: blocks must align with the cascade boundaries (cells H/4 and 3H/4 of the next
// level) and every level's half-width must be a power of two; the reference's binary32 cell index must be exact.
__host__ __device__ inline bool ngp_skip_allowed(uint32_t C, uint32_t H, float bound) {
    int e;
    return (H & (H - 1u)) == 0u && H >= 64u && (unsigned long long)C * H * H * H <= (1ull << 24) &&
           (C == 1u || frexpf(bound, &e) == 0.5f);
}

// ---------------------------------------------------------------------------
// The per-ray march state of the per-op kernels (march_rays_train, march_rays): reference raymarching.cu:363-404, 431-483,
// 759-813.  One lane marches one ray; with few rays alive the kernels are bound by that lane's serial chain, so the
// same two devices as in the fused kernel shorten it: the 64-bit bitfield word of the current 4^3 block stays in registers,
// and empty 4^3 / 16^3 blocks are crossed by verified skips (above).  Without a coarse map the emptiness of a 16^3 block is
// found by OR-ing its 64 words (512 contiguous bytes) once per block entered.
// ---------------------------------------------------------------------------
struct ngp_march_t {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, rbound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Cf, Hm1;
    const uint8_t* grid;
    uint32_t blocks_per_level;        // H^3 / 64 when the reference's binary32 cell index is exact and H a power of two, else 0
    uint32_t c_blk, c_lo, c_hi;       // the block whose word is cached, and the word
    float skip_M;                     // per-ray margin of the verified skip (ngp_skip_margin); inf = never skip
    uint32_t s_blk, s_empty;          // the 16^3 block last examined and whether it is empty
    const uint32_t* coarse;           // optional (LDS): one bit per 4^3 block, 1 = some cell of the block is occupied
    uint32_t coarse_words;            // 32-bit words of it per cascade level

    __device__ __forceinline__ void setup(const float* o, const float* d, float bound_, float dt_gamma_,
                                          uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid_) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        rdx = 1.0f / dx; rdy = 1.0f / dy; rdz = 1.0f / dz;
        bound = bound_; rbound = 1.0f / bound_; dt_gamma = dt_gamma_;
        Hf = (float)H; Cf = (float)C; Hm1 = (float)(H - 1);
        rH = 1.0f / Hf;
        H3 = (float)(H * H * H);
        dt_min = (2.0f * 1.7320508075688772f) / (float)max_steps;
        dt_max = ((2.0f * 1.7320508075688772f) * (float)(1 << (C - 1))) / Hf;
        grid = grid_;
        const bool exact = (H & (H - 1u)) == 0u && H >= 4u && (uint64_t)C * H * H * H <= (1ull << 24) &&
                           (reinterpret_cast<uintptr_t>(grid_) & 15u) == 0u;
        blocks_per_level = exact ? (H * H * H) >> 6 : 0u;
        c_blk = 0xffffffffu; c_lo = 0u; c_hi = 0u;
        skip_M = __builtin_inff();
        s_blk = 0xffffffffu; s_empty = 0u;
        coarse = nullptr; coarse_words = 0u;
    }

    // A coarse occupancy map (csrc/raymarching.hip: k_rm_build_coarse, staged in LDS by the kernel) answers "is this 4^3 / 16^3 block
    // empty" without touching the bitfield: the empty part of a ray costs no dependent global load at all.
    __device__ __forceinline__ void use_coarse(const uint32_t* lds_coarse, uint32_t words_per_level) {
        if (blocks_per_level && lds_coarse) { coarse = lds_coarse; coarse_words = words_per_level; }
    }

    // enable verified block skipping for this ray (call after setup, once `far` is known)
    __device__ __forceinline__ void allow_skip(uint32_t C, uint32_t H, float far) {
        if (blocks_per_level && ngp_skip_allowed(C, H, bound)) skip_M = ngp_skip_margin(*this, bound, far);
    }

    __device__ __forceinline__ int mip(int e) const {
        return (int)fminf(Cf - 1.0f, fmaxf(0.0f, (float)e));
    }

    __device__ __forceinline__ bool super_empty(int level, uint32_t mort) {
        const uint32_t sid = (uint32_t)level * (blocks_per_level >> 6) + (mort >> 12);
        if (sid != s_blk) {
            const uint4* w = reinterpret_cast<const uint4*>(grid) + (((size_t)level * blocks_per_level + ((size_t)(mort >> 12) << 6)) >> 1);
            uint4 acc = w[0];
            #pragma unroll
            for (int i = 1; i < 32; i++) { const uint4 v = w[i]; acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w; }
            s_blk = sid;
            s_empty = ((acc.x | acc.y | acc.z | acc.w) == 0u) ? 1u : 0u;
        }
        return s_empty != 0u;
    }

    // Probe at parameter t.  Occupied: returns true with the sample (x,y,z,dt), t untouched.
    // Empty: returns false after moving t past the cell (or, verified, through the empty block around it).
    __device__ __forceinline__ bool probe(float& t, float& x, float& y, float& z, float& dt) {
        const float tc = t;
        ngp_point<ngp_march_t> r;
        r.at(*this, tc);
        x = r.x; y = r.y; z = r.z; dt = r.dt;
        const uint32_t mort = ngp_morton3((uint32_t)r.nx, (uint32_t)r.ny, (uint32_t)r.nz);
        bool occ, block_is_empty = false;
        bool super = false;
        if (coarse) {
            const uint32_t blk = mort >> 6;
            const bool maybe = (coarse[(uint32_t)r.level * coarse_words + (blk >> 5)] >> (blk & 31u)) & 1u;
            if (maybe) {
                const uint32_t gblk = (uint32_t)r.level * blocks_per_level + blk;
                if (gblk != c_blk) {
                    const uint2 w = reinterpret_cast<const uint2*>(grid)[gblk];
                    c_blk = gblk; c_lo = w.x; c_hi = w.y;
                }
                occ = (((mort & 32u) ? c_hi : c_lo) >> (mort & 31u)) & 1u;
            } else {
                occ = false;
                block_is_empty = true;
                const uint32_t sw = (uint32_t)r.level * coarse_words + ((blk >> 6) << 1);       // the 64 coarse bits of the 16^3 block
                super = (coarse[sw] | coarse[sw + 1]) == 0u;
            }
        } else if (blocks_per_level) {
            // the reference's bit (raymarching.cu:382-383) is bit (morton & 63) of word level * H^3 / 64 + (morton >> 6)
            const uint32_t gblk = (uint32_t)r.level * blocks_per_level + (mort >> 6);
            if (gblk != c_blk) {
                const uint2 w = reinterpret_cast<const uint2*>(grid)[gblk];
                c_blk = gblk; c_lo = w.x; c_hi = w.y;
            }
            occ = (((mort & 32u) ? c_hi : c_lo) >> (mort & 31u)) & 1u;
            block_is_empty = (c_lo | c_hi) == 0u;
        } else {
            const uint32_t index = (uint32_t)((float)r.level * H3 + (float)mort);
            occ = (grid[index >> 3] >> (index & 7u)) & 1u;
        }
        if (occ) return true;
        const float tt = r.cell_exit(*this, tc);
        if (block_is_empty && skip_M < __builtin_inff()) {
            const int sh = (coarse ? super : super_empty(r.level, mort)) ? 4 : 2;
            const float ta = ngp_try_skip(*this, r, tc, tt, sh, skip_M);
            if (ta >= 0.0f) { t = ta; return false; }
        }
        float tp;
        t = ngp_advance(*this, tc, tt, tp);
        return false;
    }
};
