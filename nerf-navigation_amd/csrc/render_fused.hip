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
#include "ngp_march.h"
#include "ngp_camera.h"

#ifndef RF_MIX_BLEND
#define RF_MIX_BLEND 2                 // products of the trilinear blend, half(w * float(v)) with TWO roundings as in the reference:
                                       //   0: cvt / v_pk_mul_f32 / cvt (ngp_f2h)                                          4.02-4.04 ms
                                       //   2: v_fma_mix_f32 (binary32 product straight from the packed halves) + one cvt_pk: same bits, 3.87-3.88 ms
                                       //   1: v_fma_mixlo/hi_f16 -- timing only, NOT the reference arithmetic: it rounds the exact product
                                       //      once (1,637 of the 1.92 M values of an 800x800 image differ by up to 7.5e-5)  3.95-3.97 ms
#endif
static constexpr int RF_L = 16;       // levels (4 per lane group)
static constexpr uint32_t RF_BLOCK = 256;
// workgroups of k_field_forward_lds per CU.  Two 16-point tiles per pass need ~200 VGPRs: at 4 workgroups per CU (4 waves per SIMD,
// 128 VGPRs) the kernel spilled 81 registers to scratch (400 B per lane); at 2 it has none (tests/test_build_options.py checks).
#ifndef RF_FIELD_WG_PER_CU
#define RF_FIELD_WG_PER_CU 2
#endif
#ifndef RF_PROBES_PER_ROUND
#define RF_PROBES_PER_ROUND 512        // march iterations a round may spend so that every lane can collect its RV_S samples.
#endif                                 // A/B on MI355X, ms per frame.  Before block skipping (each probe = one cell): 4: 26.0,
                                       // 8: 19.9, 16: 12.5, 32: 12.2, 48: 12.6.  With block skipping and RV_S = 4: 6: 6.62, 10: 5.82,
                                       // 16: 5.26, 24: 4.98, 32: 4.83, 48: 4.60, 64: 4.5-4.8, 128: 4.40, 256: 4.41, 1024: 4.58.
                                       // With RV_S = 12 (512 threads): 64: 4.59, 128: 4.33, 256: 4.2, 512: 4.05

struct rf_params {
    const uint32_t* table;            // [sO] half2 rows
    const int* offsets;               // [17]
    const _Float16* w_sigma;          // 64*(32+64+16)
    const _Float16* w_color;          // 64*(32+128+16)
    float bound, density_scale;
    float inv_b2;                     // 1 / (2 bound) when that is exact (2 bound a power of two), else 0
    float scale[RF_L];                // exp2f(l*S)*H - 1 (host)
    uint32_t resolution[RF_L];        // ceil(scale)+1
    sh_norm shn;
};

// Level -> lane mapping.  Lane group g = lane >> 4 gathers, in iteration i = 0..3, level 4i + g, and holds its two
// features at slots 2i, 2i+1 of the first layer's B fragment (the first layer's A fragments are loaded in that same k
// order, rf_load_a_sigma_in).  Interleaving the levels over the lane groups makes an ITERATION nearly uniform across the
// wave: in the reference's 16-level grid, iteration 0 is levels 0..3 (all dense), iterations 2 and 3 are levels 8..15
// (all hashed) and only iteration 1 (levels 4..7) mixes both kinds, so three of four iterations run straight-line code
// for one kind of level instead of executing both sides of a per-lane branch.
struct rf_lane_levels {
    float scale[4];
    uint32_t base4[4];                         // byte offset of the level's first row in the table
    uint32_t size[4];                          // rows
    uint32_t s1b[4], s2b[4];                   // dense level: y and z strides in BYTES; s1b == 0 marks a hashed level
    uint32_t mask4[4];                         // hashed level with 2^k rows: (rows - 1) * 4, else 0
};

__device__ __forceinline__ void rf_setup_levels(const rf_params& P, int g, rf_lane_levels& lv) {
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        // select by lane group from the scalar (kernarg) arrays
        float sc = P.scale[4 * i]; uint32_t rs = P.resolution[4 * i];
        if (g == 1) { sc = P.scale[4 * i + 1]; rs = P.resolution[4 * i + 1]; }
        if (g == 2) { sc = P.scale[4 * i + 2]; rs = P.resolution[4 * i + 2]; }
        if (g == 3) { sc = P.scale[4 * i + 3]; rs = P.resolution[4 * i + 3]; }
        const int level = 4 * i + g;
        const uint32_t o0 = (uint32_t)P.offsets[level], o1 = (uint32_t)P.offsets[level + 1];
        const uint32_t size = o1 - o0;
        // reference get_grid_index (gridencoder.cu:54-72): the stride stops growing once it exceeds hashmap_size
        uint32_t stride = 1, s1 = 0, s2 = 0;
        bool dense = true;
        #pragma unroll
        for (int d = 0; d < 3; d++) {
            if (stride <= size) {
                if (d == 1) s1 = stride;
                if (d == 2) s2 = stride;
                stride *= (rs + 1);
            } else dense = false;
        }
        if (stride > size) dense = false;
        if (rs + 1 > 1024u) dense = false;     // keeps the 24-bit multiplies of the dense path exact; such a level is hashed anyway
        lv.scale[i] = sc; lv.base4[i] = o0 * 4u; lv.size[i] = size;
        lv.s1b[i] = dense ? s1 * 4u : 0u; lv.s2b[i] = dense ? s2 * 4u : 0u;
        lv.mask4[i] = (!dense && (size & (size - 1)) == 0) ? (size - 1) * 4u : 0u;
    }
}

// Which kind of code each iteration needs, decided once per kernel by the whole wave (bit i = iteration i).
struct rf_iter_class { uint32_t dense, hashed, select; };

__device__ __forceinline__ rf_iter_class rf_classify(const rf_lane_levels& lv) {   // call with all 64 lanes active
    rf_iter_class c = {0u, 0u, 0u};
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned long long bd = __ballot(lv.s1b[i] != 0u), bh = __ballot(lv.mask4[i] != 0u);
        if (bd == ~0ull) c.dense |= 1u << i;                       // every lane: dense level
        else if (bh == ~0ull) c.hashed |= 1u << i;                 // every lane: hashed level with 2^k rows
        else if ((bd | bh) == ~0ull) c.select |= 1u << i;          // a mix of those two
    }                                                              // otherwise (a hashed level whose size is not 2^k): generic
    return c;
}

__device__ __forceinline__ float rf_h(float v) { return (float)ngp_f2h(v); }   // round to half, back to float

struct rf_row2 { uint32_t lo, hi; };                                          // two consecutive rows
__device__ __forceinline__ uint32_t rf_row(const rf_params& P, uint32_t byte_off) {
    asm("" : "+v"(byte_off));      // keep the 32-bit offset a VGPR value of its own: the load is then SGPR base + VGPR offset
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(P.table) + byte_off);
}
__device__ __forceinline__ rf_row2 rf_rows(const rf_params& P, uint32_t byte_off) {   // one 8-byte load, 4-byte aligned
    asm("" : "+v"(byte_off));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    typedef u2 u2_a4 __attribute__((aligned(4)));
    const u2 v = *reinterpret_cast<const u2_a4*>(reinterpret_cast<const char*>(P.table) + byte_off);
    return rf_row2{v.x, v.y};
}

// Hash-grid encoding of one sample for the 4 levels of this lane's group -> 8 half features (slot 2i + ch = level 4i + g).
// Arithmetic identical, operation for operation, to k_grid_forward<_Float16,3,2> (gridencoder.hip); only the address
// computation is arranged differently (byte offsets, strides folded into multiply-adds, the hash computed pre-shifted:
// (y * p) << 2 == y * (p << 2) mod 2^32, and the power-of-two modulo taken on the operands: (a ^ b) & m == (a & m) ^ (b & m)).
// INRANGE: the caller guarantees |w| <= bound (march samples are clamped to the box, raymarching.cu:365-367), so the
// normalised position is in [0,1] and the out-of-range handling is dead code.
// The gathers of one PAIR of iterations (h = 0: levels g and 4+g, h = 1: levels 8+g and 12+g) for a normalised position:
// cell, fractions, byte offsets, loads issued (nothing waits here).  Splitting the encoder in pairs lets the frame kernel
// issue the next tile's pair 1 (the hashed levels, the slow gathers) before the current tile's MLP (RV_PIPELINE).
struct rf_pair { uint32_t raw[2][8]; float fx[2], fy[2], fz[2]; };

__device__ __forceinline__ void rf_normalise(const rf_params& P, float wx, float wy, float wz, float& x0, float& x1, float& x2) {
    // GridEncoder.forward (grid.py:144): (x + bound) / (2 bound).  When 2*bound is a power of two (every cascade-aligned
    // bound) the quotient equals the product with the exact reciprocal, bit for bit, and skips three IEEE divisions.
    const float b2 = 2 * P.bound;
    if (P.inv_b2 != 0.0f) { x0 = (wx + P.bound) * P.inv_b2; x1 = (wy + P.bound) * P.inv_b2; x2 = (wz + P.bound) * P.inv_b2; }
    else { x0 = (wx + P.bound) / b2; x1 = (wy + P.bound) / b2; x2 = (wz + P.bound) / b2; }
}

template <int H>
__device__ __forceinline__ void rf_gather_pair(const rf_params& P, const rf_lane_levels& lv, const rf_iter_class cls,
                                               float x0, float x1, float x2, rf_pair& o) {
    constexpr uint32_t P1 = 2654435761u, P2 = 805459861u;              // fast_hash primes (gridencoder.cu:35-51)
    // positions of the two levels at once: packed binary32 multiply and add (same roundings as the scalar operations)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 sc2 = {lv.scale[2 * H], lv.scale[2 * H + 1]};
    const f2 ppx = x0 * sc2 + 0.5f, ppy = x1 * sc2 + 0.5f, ppz = x2 * sc2 + 0.5f;
    uint32_t (&raw)[2][8] = o.raw;
    float (&fx)[2] = o.fx; float (&fy)[2] = o.fy; float (&fz)[2] = o.fz;
    #pragma unroll
    for (int i = 2 * H; i < 2 * H + 2; i++) {
        const float px = ppx[i & 1], py = ppy[i & 1], pz = ppz[i & 1];
        const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
        const uint32_t gx = (uint32_t)flx, gy = (uint32_t)fly, gz = (uint32_t)flz;
        fx[i & 1] = px - flx; fy[i & 1] = py - fly; fz[i & 1] = pz - flz;        // == px - (float)gx: the floor is an exact small integer
        const uint32_t bit = 1u << i;
        if (cls.dense & bit) {
            // x + y*s1 + z*s2 (always < size); the x-neighbour is the next row: one 8-byte load per (y, z)
            const uint32_t o00 = __umul24(gz, lv.s2b[i]) + (__umul24(gy, lv.s1b[i]) + ((gx << 2) + lv.base4[i]));
            const uint32_t o01 = o00 + lv.s1b[i], o10 = o00 + lv.s2b[i], o11 = o01 + lv.s2b[i];
#ifdef RV_EXPERIMENT_FREE_LEVELS   // timing-only build (wrong image): the gathers of levels 0 .. RV_EXPERIMENT_FREE_LEVELS-1 are not issued at all --
            // an upper bound on what serving those levels from LDS could gain (an LDS read cannot be cheaper than no read)
            // (level 0 keeps its constant-one second feature, which the bench model's density logit reads: same densities, same sample count)
            const uint32_t cst = ((threadIdx.x & 63u) >> 4) == 0u ? 0x3C000000u : 0u;
            rf_row2 r0 = {cst, cst}, r1 = {cst, cst}, r2 = {cst, cst}, r3 = {cst, cst};
            if (i != 0 || (int)((threadIdx.x & 63u) >> 4) >= RV_EXPERIMENT_FREE_LEVELS) {
                r0 = rf_rows(P, o00); r1 = rf_rows(P, o01); r2 = rf_rows(P, o10); r3 = rf_rows(P, o11);
            }
#else
            const rf_row2 r0 = rf_rows(P, o00), r1 = rf_rows(P, o01), r2 = rf_rows(P, o10), r3 = rf_rows(P, o11);
#endif
            raw[i & 1][0] = r0.lo; raw[i & 1][1] = r0.hi; raw[i & 1][2] = r1.lo; raw[i & 1][3] = r1.hi;
            raw[i & 1][4] = r2.lo; raw[i & 1][5] = r2.hi; raw[i & 1][6] = r3.lo; raw[i & 1][7] = r3.hi;
        } else {
            uint32_t off[8];                                           // byte offsets of the 8 corners
            if (cls.hashed & bit) {
                const uint32_t m = lv.mask4[i], b = lv.base4[i];
                const uint32_t hy = gy * (P1 << 2), hz = gz * (P2 << 2);
                const uint32_t hy1 = hy + (P1 << 2), hz1 = hz + (P2 << 2);
                const uint32_t a0 = (gx << 2) & m, a1 = ((gx << 2) + 4u) & m;
                const uint32_t yz0 = (hy ^ hz) & m, yz1 = (hy1 ^ hz) & m, yz2 = (hy ^ hz1) & m, yz3 = (hy1 ^ hz1) & m;
                off[0] = (a0 ^ yz0) + b; off[1] = (a1 ^ yz0) + b; off[2] = (a0 ^ yz1) + b; off[3] = (a1 ^ yz1) + b;
                off[4] = (a0 ^ yz2) + b; off[5] = (a1 ^ yz2) + b; off[6] = (a0 ^ yz3) + b; off[7] = (a1 ^ yz3) + b;
#ifdef RV_EXPERIMENT_WINDOW        // timing-only build: levels 8..15 gather inside a window of this many bytes per level
                if (i >= 2) {
                    #pragma unroll
                    for (int c = 0; c < 8; c++) off[c] = b + ((off[c] - b) & (uint32_t)(RV_EXPERIMENT_WINDOW - 1));
                }
#endif
            } else if (cls.select & bit) {
                // both kinds in one wave: compute both offsets, select per lane, no branch
                const bool dense = lv.s1b[i] != 0u;
                const uint32_t m = lv.mask4[i], b = lv.base4[i];
                const uint32_t o00 = __umul24(gz, lv.s2b[i]) + (__umul24(gy, lv.s1b[i]) + ((gx << 2) + b));
                const uint32_t o01 = o00 + lv.s1b[i], o10 = o00 + lv.s2b[i], o11 = o01 + lv.s2b[i];
                const uint32_t hy = gy * (P1 << 2), hz = gz * (P2 << 2);
                const uint32_t hy1 = hy + (P1 << 2), hz1 = hz + (P2 << 2);
                const uint32_t a0 = (gx << 2) & m, a1 = ((gx << 2) + 4u) & m;
                const uint32_t yz0 = (hy ^ hz) & m, yz1 = (hy1 ^ hz) & m, yz2 = (hy ^ hz1) & m, yz3 = (hy1 ^ hz1) & m;
                off[0] = dense ? o00 : (a0 ^ yz0) + b; off[1] = dense ? o00 + 4u : (a1 ^ yz0) + b;
                off[2] = dense ? o01 : (a0 ^ yz1) + b; off[3] = dense ? o01 + 4u : (a1 ^ yz1) + b;
                off[4] = dense ? o10 : (a0 ^ yz2) + b; off[5] = dense ? o10 + 4u : (a1 ^ yz2) + b;
                off[6] = dense ? o11 : (a0 ^ yz3) + b; off[7] = dense ? o11 + 4u : (a1 ^ yz3) + b;
            } else {
                // generic: any mix, including a hashed level whose row count is not a power of two (index % size);
                // branch-free like the rest, so that the choice of class stays the only (wave-uniform) control flow
                const bool dense = lv.s1b[i] != 0u;
                const uint32_t s1 = lv.s1b[i] >> 2, s2 = lv.s2b[i] >> 2;
                const uint32_t hy = gy * P1, hz = gz * P2;
                #pragma unroll
                for (int c = 0; c < 8; c++) {
                    const uint32_t id = (gx + (c & 1)) + (gy * s1 + ((c & 2) ? s1 : 0u)) + (gz * s2 + ((c & 4) ? s2 : 0u));
                    const uint32_t ih = ((gx + (c & 1)) ^ (hy + ((c & 2) ? P1 : 0u)) ^ (hz + ((c & 4) ? P2 : 0u))) % lv.size[i];
                    off[c] = (dense ? id : ih) * 4u + lv.base4[i];
                }
            }
            {
                #pragma unroll
                for (int c = 0; c < 8; c++) raw[i & 1][c] = rf_row(P, off[c]);
            }
        }
    }
}

// Blend of one level.  Reference arithmetic per corner and feature (gridencoder.cu:147-166, scalar_t = at::Half):
//   w = (wx * wy) * wz in binary32;  results[ch] += w * grid[...]  ==  half(float(result) + float(half(w * float(v))))
// The product is rounded to binary32 (v_fma_mix_f32: fma32(w, float(v), +0) reads the half straight out of the packed row)
// and then to binary16 by the packed conversion; the +0 addend only turns a -0 product into +0, which a sum that starts at
// +0 cannot tell apart.  v_fma_mixlo/mixhi_f16 would do both steps in one instruction but round only once (RF_MIX_BLEND 1,
// timing only).  The packed-half add is the correctly rounded binary16 sum.
__device__ __forceinline__ void rf_blend_pair(const rf_pair& in, int h, ngp_h8& out) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    #pragma unroll
    for (int j = 0; j < 2; j++) {
        const f2 wx = {1 - in.fx[j], in.fx[j]};
        const float wy0 = 1 - in.fy[j], wz0 = 1 - in.fz[j];
        const f2 wxy0 = wx * wy0, wxy1 = wx * in.fy[j];
        const f2 w[4] = {wxy0 * wz0, wxy1 * wz0, wxy0 * in.fz[j], wxy1 * in.fz[j]};   // (y, z) = (0,0) (1,0) (0,1) (1,1)
        h2 acc = {(_Float16)0.0f, (_Float16)0.0f};
        #pragma unroll
        for (int c = 0; c < 8; c++) {
            const float wc = w[c >> 1][c & 1];
#if RF_MIX_BLEND == 1
            uint32_t prod;
            asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(prod) : "v"(wc), "v"(in.raw[j][c]));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(prod) : "v"(wc), "v"(in.raw[j][c]));
            acc = acc + __builtin_bit_cast(h2, prod);
#elif RF_MIX_BLEND == 2
            // binary32 products straight from the packed halves (v_fma_mix_f32 = fma32(w, float(v), +0), rounded to binary32),
            // then ONE packed conversion: the reference's two roundings in 3 instructions instead of 5
            float p0, p1;
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(p0) : "v"(wc), "v"(in.raw[j][c]));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(p1) : "v"(wc), "v"(in.raw[j][c]));
            const h2 prod = {(_Float16)p0, (_Float16)p1};
            acc = acc + prod;
#else
            const h2 v = __builtin_bit_cast(h2, in.raw[j][c]);
            const h2 prod = {ngp_f2h(wc * (float)v.x), ngp_f2h(wc * (float)v.y)};
            acc = acc + prod;
#endif
        }
        out[4 * h + 2 * j] = acc.x;
        out[4 * h + 2 * j + 1] = acc.y;
    }
}

template <bool INRANGE = false>
__device__ __forceinline__ ngp_h8 rf_encode(const rf_params& P, const rf_lane_levels& lv, const rf_iter_class cls,
                                            float wx, float wy, float wz) {
    float x0, x1, x2;
    rf_normalise(P, wx, wy, wz, x0, x1, x2);
    // a sample outside [0,1]^3 encodes to zeros (gridencoder.cu:118-131); it gathers at the origin so that no load needs a guard
    const bool oob = !INRANGE && ((x0 < 0 || x0 > 1) || (x1 < 0 || x1 > 1) || (x2 < 0 || x2 > 1));
    if (oob) { x0 = 0.0f; x1 = 0.0f; x2 = 0.0f; }
    rf_pair a, b;
    rf_gather_pair<0>(P, lv, cls, x0, x1, x2, a);
    rf_gather_pair<1>(P, lv, cls, x0, x1, x2, b);
    ngp_h8 out;
    rf_blend_pair(a, 0, out);
    rf_blend_pair(b, 1, out);
    if (oob) {
        #pragma unroll
        for (int j = 0; j < 8; j++) out[j] = (_Float16)0.0f;
    }
    return out;
}

// density-net first layer, A fragments in rf_encode's k order: slots 2i, 2i+1 of lane group g = features of level 4i + g
__device__ __forceinline__ ngp_h8 rf_load_a_sigma_in(const _Float16* __restrict__ W, int t, int lane) {
    const int row = 16 * t + (lane & 15), g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        a[2 * i] = W[row * 32 + 2 * (4 * i + g)];
        a[2 * i + 1] = W[row * 32 + 2 * (4 * i + g) + 1];
    }
    return a;
}

// colour-net first layer, A fragments in the k order {h[4g..4g+3], SH[4g..4g+3]} (see the header comment)
__device__ __forceinline__ ngp_h8 rf_load_a_color_in(const _Float16* __restrict__ W, int t, int lane) {
    const int row = 16 * t + (lane & 15), g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int j = 0; j < 4; j++) {
        const int hi = 4 * g + j;                      // index into the density net's output h
        a[j] = (hi == 0) ? (_Float16)0.0f : W[row * 32 + 15 + hi];   // geo feature hi-1 sits at input column 16 + (hi-1)
        a[4 + j] = W[row * 32 + 4 * g + j];            // SH feature 4g+j
    }
    return a;
}

__global__ void k_field_forward_lds(rf_params P, const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                                    float* __restrict__ sigmas, float* __restrict__ rgbs);

static int rf_fill_params(const char* who, const ngp_field_t* f, rf_params& P) {
    NGP_REQUIRE(f && f->embeddings && f->offsets && f->sigma_weights && f->color_weights, "%s: null field pointer", who);
    NGP_REQUIRE(f->L == RF_L, "%s: the fused path is built for 16 levels x 2 features (the reference's hashgrid)", who);
    NGP_REQUIRE(f->bound > 0, "%s: bound must be positive", who);
    P.table = (const uint32_t*)f->embeddings;
    P.offsets = f->offsets;
    P.w_sigma = (const _Float16*)f->sigma_weights;
    P.w_color = (const _Float16*)f->color_weights;
    P.bound = f->bound;
    P.density_scale = f->density_scale;
    {
        int e;
        const float b2 = 2.0f * f->bound;
        P.inv_b2 = (frexpf(b2, &e) == 0.5f) ? 1.0f / b2 : 0.0f;
    }
    for (int l = 0; l < RF_L; l++) {
        P.scale[l] = exp2f((float)l * f->S) * (float)f->H - 1.0f;
        P.resolution[l] = (uint32_t)ceilf(P.scale[l]) + 1u;
    }
    sh_fill_norm(P.shn);
    return NGP_OK;
}

extern "C" int ngp_field_forward(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                                 float* sigmas, float* rgbs, void* stream) {
    rf_params P;
    int rc = rf_fill_params("field_forward", field_host, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && dirs && sigmas && rgbs, "field_forward: null pointer");
    const uint32_t npairs = (M + 31) >> 5;
    uint32_t blocks = ngp_div_up(npairs, RF_BLOCK / 64);
    if (blocks > 256 * RF_FIELD_WG_PER_CU) blocks = 256 * RF_FIELD_WG_PER_CU;
    hipLaunchKernelGGL(k_field_forward_lds, dim3(blocks), dim3(RF_BLOCK), 36 * 1024, (hipStream_t)stream, P, xyzs, dirs, M, sigmas, rgbs);
    NGP_CHECK_LAUNCH("field_forward");
    return NGP_OK;
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
static constexpr int RV_NFRAG = 36;
static constexpr uint32_t RV_LDS_W = RV_NFRAG * 1024;                  // weight fragments
static constexpr uint32_t RV_LDS_SH = RV_WAVES * 64 * 32;              // 16 halves per lane
static constexpr uint32_t RV_LDS_LV = 4 * 96;                          // rf_lane_levels of the 4 lane groups

struct rf_frame {
    const float* rays_o; const float* rays_d; uint32_t N;   // rays_o == null: the rays are those of `cam` (pixel = ray id)
    ngp_camera cam;
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
    const uint32_t* tile_order;      // [N/64] 8x8 tiles, most expensive first (k_tile_order), or null
};

// coarse[level][m] = any fine bit set in Morton block m (64 bits = 8 bytes of the bitfield)
__global__ __launch_bounds__(256) void k_build_coarse(const uint8_t* __restrict__ bitfield, uint32_t n_blocks_total,
                                                      uint32_t* __restrict__ coarse) {
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


__device__ __forceinline__ ngp_h8 rv_frag(const ngp_h8* __restrict__ lds_w, int f, int lane) { return lds_w[f * 64 + lane]; }

__device__ __forceinline__ void rv_mlp_tile(const ngp_h8* __restrict__ lds_w, int lane, const ngp_h8 x, ngp_h4 shq,
                                            float& sigma, float& cr, float& cg, float& cb);

// One tile of the field: encoder, then both networks with the weights streamed from LDS; the SH coefficients of the column's ray come from LDS
__device__ __forceinline__ void rv_field_tile(const rf_params& P, const rf_lane_levels& lv, const rf_iter_class cls,
                                              const ngp_h8* __restrict__ lds_w, int lane,
                                              float px, float py, float pz, ngp_h4 shq,
                                              float& sigma, float& cr, float& cg, float& cb) {
    const ngp_h8 x = rf_encode<true>(P, lv, cls, px, py, pz);
    rv_mlp_tile(lds_w, lane, x, shq, sigma, cr, cg, cb);
}

// The two networks on NT 16-column tiles at once (encoded features already in B-fragment layout, weights from LDS).  NT = 2
// runs two independent MFMA chains through every layer: each weight fragment is read from LDS once for both tiles and the
// second chain fills the issue slots the first one leaves while it waits on its MFMA results (with two waves per SIMD there
// is little else to fill them).
template <int NT>
__device__ __forceinline__ void rv_mlp_tiles(const ngp_h8* __restrict__ lds_w, int lane, const ngp_h8 (&x)[NT], const ngp_h4 (&shq)[NT],
                                             float (&sigma)[NT], float (&cr)[NT], float (&cg)[NT], float (&cb)[NT]) {
    const ngp_f4 zero = {0.f, 0.f, 0.f, 0.f};
    ngp_h8 act[NT][2];
    {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w = rv_frag(lds_w, t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w, x[n], zero);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) { act[n][0] = mlp_pack_relu(d[n][0], d[n][1]); act[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
    }
    {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w0 = rv_frag(lds_w, 4 + 2 * t, lane), w1 = rv_frag(lds_w, 5 + 2 * t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w0, act[n][0], zero);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w1, act[n][1], d[n][t]);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) { act[n][0] = mlp_pack_relu(d[n][0], d[n][1]); act[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
    }
    ngp_f4 h[NT];
    {
        const ngp_h8 w0 = rv_frag(lds_w, 12, lane), w1 = rv_frag(lds_w, 13, lane);
        #pragma unroll
        for (int n = 0; n < NT; n++) h[n] = ngp_mfma(w0, act[n][0], zero);
        #pragma unroll
        for (int n = 0; n < NT; n++) h[n] = ngp_mfma(w1, act[n][1], h[n]);
    }
    ngp_h8 cin[NT];
    #pragma unroll
    for (int n = 0; n < NT; n++)
        #pragma unroll
        for (int j = 0; j < 4; j++) { cin[n][j] = (_Float16)h[n][j]; cin[n][4 + j] = shq[n][j]; }
    {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w = rv_frag(lds_w, 14 + t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w, cin[n], zero);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) { act[n][0] = mlp_pack_relu(d[n][0], d[n][1]); act[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
    }
    #pragma unroll
    for (int l = 0; l < 2; l++) {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w0 = rv_frag(lds_w, 18 + 8 * l + 2 * t, lane), w1 = rv_frag(lds_w, 19 + 8 * l + 2 * t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w0, act[n][0], zero);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w1, act[n][1], d[n][t]);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) { act[n][0] = mlp_pack_relu(d[n][0], d[n][1]); act[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
    }
    {
        const ngp_h8 w0 = rv_frag(lds_w, 34, lane), w1 = rv_frag(lds_w, 35, lane);
        ngp_f4 o[NT];
        #pragma unroll
        for (int n = 0; n < NT; n++) o[n] = ngp_mfma(w0, act[n][0], zero);
        #pragma unroll
        for (int n = 0; n < NT; n++) o[n] = ngp_mfma(w1, act[n][1], o[n]);
        // raw network outputs (lanes g == 0: density logit and the three colour logits of column s); the activations are
        // applied once per round by the lane that owns the sample (rv_activate), not once per pass by all 64 lanes
        #pragma unroll
        for (int n = 0; n < NT; n++) { sigma[n] = h[n][0]; cr[n] = o[n][0]; cg[n] = o[n][1]; cb[n] = o[n][2]; }
    }
}

__device__ __forceinline__ void rv_mlp_tile(const ngp_h8* __restrict__ lds_w, int lane, const ngp_h8 x, ngp_h4 shq,
                                            float& sigma, float& cr, float& cg, float& cb) {
    const ngp_h8 xs[1] = {x};
    const ngp_h4 ss[1] = {shq};
    float a[1], b[1], c[1], d[1];
    rv_mlp_tiles<1>(lds_w, lane, xs, ss, a, b, c, d);
    sigma = a[0]; cr = b[0]; cg = c[0]; cb = d[0];
}

// The 36 MFMA weight fragments of both networks into LDS, each in the k order its consumer expects (fragment-major:
// lane l reads 16 B at 16 l, a conflict-free ds_read_b128).
__device__ __forceinline__ void rv_stage_weights(const rf_params& P, ngp_h8* __restrict__ lds_w, int wave, int nwaves, int lane) {
    for (int f = wave; f < RV_NFRAG; f += nwaves) {
        ngp_h8 a;
        const _Float16* Wc = P.w_color;
        const _Float16* Wch = Wc + MLP_W * 32;
        if (f < 4) a = rf_load_a_sigma_in(P.w_sigma, f, lane);
        else if (f < 12) a = mlp_load_a_permuted(P.w_sigma + MLP_W * 32, MLP_W, (f - 4) >> 1, (f - 4) & 1, lane);
        else if (f < 14) a = mlp_load_a_permuted(P.w_sigma + MLP_W * 32 + MLP_W * MLP_W, MLP_W, 0, f - 12, lane);
        else if (f < 18) a = rf_load_a_color_in(Wc, f - 14, lane);
        else if (f < 34) a = mlp_load_a_permuted(Wch + ((f - 18) >> 3) * MLP_W * MLP_W, MLP_W, ((f - 18) & 7) >> 1, (f - 18) & 1, lane);
        else a = mlp_load_a_permuted(Wch + 2 * MLP_W * MLP_W, MLP_W, 0, f - 34, lane);
        lds_w[f * 64 + lane] = a;
    }
}

// trunc_exp forward (activation.py:9-10, fp32 of the half logit) times density_scale, and torch.sigmoid on the half logits
__device__ __forceinline__ void rv_activate(const rf_params& P, float& sigma, float& cr, float& cg, float& cb) {
    sigma = P.density_scale * ngp_expf(rf_h(sigma));
    cr = rf_h(1.0f / (1.0f + ngp_expf(-rf_h(cr))));
    cg = rf_h(1.0f / (1.0f + ngp_expf(-rf_h(cg))));
    cb = rf_h(1.0f / (1.0f + ngp_expf(-rf_h(cb))));
}

// ---------------------------------------------------------------------------
// field_forward for explicit points (NeRFNetwork.forward in one launch): the frame kernel's field step on its own.
// 256-thread workgroups, weights in LDS (36 KiB each, RF_FIELD_WG_PER_CU workgroups share a CU), two 16-point tiles per pass.
// ---------------------------------------------------------------------------
template <bool FIXED>
__device__ __forceinline__ void rf_points_loop(const rf_params& P, const rf_iter_class cls_rt, const rf_lane_levels& lv, const ngp_h8* __restrict__ lds_w,
                                               const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                                               float* __restrict__ sigmas, float* __restrict__ rgbs) {
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
                rgbs[3ull * m[n]] = cr[n]; rgbs[3ull * m[n] + 1] = cg[n]; rgbs[3ull * m[n] + 2] = cb[n];
            }
    }
}

__global__ __launch_bounds__(RF_BLOCK, RF_FIELD_WG_PER_CU) void k_field_forward_lds(rf_params P, const float* __restrict__ xyzs, const float* __restrict__ dirs,
                                                                    uint32_t M, float* __restrict__ sigmas, float* __restrict__ rgbs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rf_smem);
    const int lane = threadIdx.x & 63, g = lane >> 4, wave = threadIdx.x >> 6;
    rv_stage_weights(P, lds_w, wave, RF_BLOCK / 64, lane);
    rf_lane_levels lv;
    rf_setup_levels(P, g, lv);
    __syncthreads();
    const rf_iter_class cls = rf_classify(lv);
    if (cls.dense == 1u && cls.select == 2u && cls.hashed == 12u) rf_points_loop<true>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs);
    else rf_points_loop<false>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs);
}

// queue index -> ray id.  With tile_w set (rays are a row-major image whose width and height are multiples of 8)
// consecutive queue indices walk 8x8 pixel tiles, so the 64 lanes of a wave start on a compact patch of the image
// and their gathers share cache lines; otherwise the identity.
__device__ __forceinline__ uint32_t rv_ray_of(uint32_t idx, uint32_t tile_w, const uint32_t* __restrict__ tile_order) {
    if (tile_w == 0) return idx;
    uint32_t tile = idx >> 6;
    const uint32_t in = idx & 63u, tiles_x = tile_w >> 3;
    if (tile_order) tile = tile_order[tile];
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
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
                                              float4* lds_smp, const uint32_t* lds_coarse) {
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
            if (need) {
                const uint32_t cnt = (uint32_t)__popcll(need);
                uint32_t base = 0;
#if RV_XCD_QUEUES
                // Queue q holds the rays of image band q (tiles [T q / 8, T (q + 1) / 8) of 64 rays): the workgroups of one XCD
                // start on the same band, so that the table lines neighbouring rays share are fetched into ONE L2 and
                // not into eight.  A wave that finds its queue empty moves on to the next one (stealing keeps the load
                // balanced; the lanes left without a ray wait one round).
                const uint32_t n_tiles64 = (F.N + 63u) >> 6;
                const uint32_t q_lo = (uint32_t)(((unsigned long long)n_tiles64 * rv_q) >> 3) << 6;
                uint32_t q_hi = (uint32_t)(((unsigned long long)n_tiles64 * (rv_q + 1u)) >> 3) << 6;
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
                        ray = rv_ray_of(idx, F.tile_w, F.tile_order);
                        float o[3], d[3];
                        if (F.rays_o) {
                            #pragma unroll
                            for (int k = 0; k < 3; k++) { o[k] = F.rays_o[3ull * ray + k]; d[k] = F.rays_d[3ull * ray + k]; }
                        } else {                           // camera mode: get_rays fused into the refill (nerf/utils.py:98-108)
                            o[0] = F.cam.t[0]; o[1] = F.cam.t[1]; o[2] = F.cam.t[2];
                            ngp_camera_ray(F.cam, ray, d);
                        }
                        ngp_near_far_inline(o, d, F.aabb, F.min_near, near, far);
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
#if RV_XCD_QUEUES
                if (base + cnt >= q_hi) {                  // this queue has run dry: move on, until all eight have been seen
                    rv_q = (rv_q + 1u) & 7u;
                    if (++rv_q_seen == 8u) exhausted = true;
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
        if (active) {
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
            const float M = F.skip ? ngp_skip_margin(mr, K.bound, far_r) : __builtin_inff();
            // this lane's slots, recomputed from the lane id (2 VALU) rather than kept across the field evaluation in scratch
            const uint32_t slot0 = ((uint32_t)wave_s * 64u + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) * RV_S;
            float4* const smp_w = lds_smp + slot0;
            int probes = 0;
            for (;;) {
#ifdef RV_COUNTERS
                n_trips = __builtin_amdgcn_readfirstlane(n_trips) + 1;
#endif
                if (!(t < far_r && nsamp < F.max_steps)) { ended = true; break; }
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
                    if (++cnt == RV_S) break;
                }
                if (++probes >= RF_PROBES_PER_ROUND) break;
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
        const bool capped = !done && ended && nsamp >= F.max_steps && t < far;
        if (ended) done = true;
        if (done) {
            F.image[3ull * ray] = cr + (1 - ws) * F.bg[0];
            F.image[3ull * ray + 1] = cg + (1 - ws) * F.bg[1];
            F.image[3ull * ray + 2] = cb + (1 - ws) * F.bg[2];
            F.depth[ray] = fmaxf(dacc - near, 0.0f) / (far - near);
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
    float4* lds_smp = reinterpret_cast<float4*>(rv_smem + RV_LDS_W + RV_LDS_SH + RV_LDS_LV);
    uint32_t* lds_coarse = F.coarse ? reinterpret_cast<uint32_t*>(rv_smem + RV_LDS_W + RV_LDS_SH + RV_LDS_LV + RV_LDS_SMP) : nullptr;

    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15, wave = threadIdx.x >> 6;

    rv_stage_weights(P, lds_w, wave, RV_WAVES, lane);
    if (lds_coarse) {
        const uint32_t nw = F.coarse_words * F.C;
        for (uint32_t i = threadIdx.x; i < nw; i += RV_BLOCK) lds_coarse[i] = F.coarse[i];
    }
    if (wave == 0 && s == 0) {
        rf_lane_levels tmp;
        rf_setup_levels(P, g, tmp);
        lds_lv[g] = tmp;
    }
    __syncthreads();
    const rf_iter_class cls = rf_classify(lds_lv[g]);
    if (cls.dense == 1u && cls.select == 2u && cls.hashed == 12u)
        rv_frame_loop<true>(P, F, cls, lds_w, lds_sh, lds_lv, lds_smp, lds_coarse);
    else
        rv_frame_loop<false>(P, F, cls, lds_w, lds_sh, lds_lv, lds_smp, lds_coarse);
}
#endif  // RV_S > 1

static inline bool rv_pow2(uint32_t v) { return v && !(v & (v - 1)); }

// process-wide validation switch; atomic because callers may render from several host threads (one stream each)
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

static int rv_render_frame(const ngp_field_t* field_host, const float* rays_o, const float* rays_d, const ngp_camera* cam, uint32_t N,
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
    F.coarse = nullptr; F.coarse_words = 0; F.skip = 0;
    static_assert(sizeof(rf_lane_levels) * 4 == RV_LDS_LV, "LDS carve of the level table");
    size_t lds = RV_LDS_W + RV_LDS_SH + RV_LDS_LV;
#if RV_S > 1
    lds += RV_LDS_SMP;
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
        hipLaunchKernelGGL(k_build_coarse, dim3(ngp_div_up(n_blocks_total / 32, 256)), dim3(256), 0, s, bitfield, n_blocks_total, coarse);
        F.coarse = coarse;
        F.coarse_words = (uint32_t)(blocks_per_level / 32);
        lds += coarse_bytes;
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
    return rv_render_frame(field_host, rays_o, rays_d, nullptr, N, image_width, aabb_host, min_near, bitfield, C, Hgrid, dt_gamma, max_steps,
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
    return rv_render_frame(field_host, nullptr, nullptr, &cam, H * W, W, aabb_host, min_near, bitfield, C, Hgrid, dt_gamma, max_steps,
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

// ===========================================================================
// Training step of the field in two launches (NeRFNetwork.forward + its autograd backward, nerf/network_ff.py:51-77,
// ffmlp/src/ffmlp.cu:410-518,749-895, activation.py:9-21).
//
//   k_field_train_forward   = k_field_forward_lds, and additionally keeps each sample's 32 encoded features in the layout the first
//                             layer consumes them in (one ngp_h8 per lane per 16-sample tile: 64 B per sample) -- the only activation
//                             that is kept; the gather is the expensive part of the forward and is not repeated.
//   k_field_train_backward  recomputes both networks from the kept features, runs the activation gradients back through them and
//                             accumulates all seven weight gradients, writing only d(loss)/d(encoded features) (64 B per sample, the
//                             input of the table scatter).  The op-by-op path moves ~1.5 KB per sample here (forward and backward
//                             buffers of every layer, the padded output gradients, the concatenated colour input and their copies).
//
// Orientation (ngp_mlp.h): a sample stays on lane column s = lane & 15; an activation / gradient tensor X of one 16-sample tile is a
// "B fragment" per 32 features (lane group g = lane >> 4 holds 8 of them).  Gradients flow back with G'^T = W^T G^T, the D registers of
// one step being the B fragment of the next, exactly like the forward.
// Weight gradients dW[o][i] = sum_s G[s][o] A[s][i] contract over SAMPLES, so both operands are needed with the feature on the lane and
// samples in the registers.  No LDS transpose: multiplying a B fragment (as the A operand) by a 0/1 selection fragment on the matrix
// core returns the same values transposed -- D[sample 4g+r][feature lane&15] -- exactly (one product per sum, f32).  Two 16-sample tiles
// give the 8 values per lane of a K = 32 operand; the sample order inside K is the same for both operands, so it does not matter.
// Every wave accumulates all 72 16x16 tiles of the seven weight gradients in registers (f32) over all its samples; at the end the four
// waves of a workgroup are summed through LDS and added to a global f32 workspace, which k_field_train_wgrad_finish rounds to the
// reference's half precision and clears.
// ===========================================================================
static constexpr int FT_NBWD = 38;                     // W^T fragments of the seven backward steps
static constexpr int FT_NSEL = 7;                      // selection fragments
static constexpr uint32_t FT_LDS = (36 + FT_NBWD + FT_NSEL) * 1024;
static constexpr uint32_t FT_WS_FLOATS = 7168 + 11264; // f32 workspace: the two weight vectors in FFMLP's own layout
// offsets inside FFMLP's weight vectors ([out][in] row-major per layer: in | hidden ... | out)
static constexpr int FT_S_IN = 0, FT_S_HID = 64 * 32, FT_S_OUT = 64 * 32 + 64 * 64;
static constexpr int FT_C_IN = 0, FT_C_HID1 = 64 * 32, FT_C_HID2 = 64 * 32 + 64 * 64, FT_C_OUT = 64 * 32 + 2 * 64 * 64;

// feature held at element j of lane group g in k-step c of a chained B fragment (ngp_mlp.h)
__device__ __forceinline__ int ft_phi(int c, int g, int j) { return 32 * c + 16 * (j >> 2) + 4 * g + (j & 3); }

// W^T fragments.  Rows (lane & 15) = input feature 16 t + m of the layer, k = the layer's output features in the order of the incoming
// gradient fragment.
__device__ __forceinline__ ngp_h8 ft_wt_hidden(const _Float16* __restrict__ W, int ld, int t, int c, int lane) {     // k order phi
    const int i = 16 * t + (lane & 15), g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int j = 0; j < 8; j++) a[j] = W[ft_phi(c, g, j) * ld + i];
    return a;
}
__device__ __forceinline__ ngp_h8 ft_wt_out(const _Float16* __restrict__ W, int ld, int t, int lane) {       // gradient fragment {rows 4g..4g+3, 0, 0, 0, 0}
    const int i = 16 * t + (lane & 15), g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int j = 0; j < 8; j++) a[j] = j < 4 ? W[(4 * g + j) * ld + i] : (_Float16)0.0f;
    return a;
}
// colour net's first layer: output row m <-> density-net output m (m = 0: the density logit, not a colour input; m >= 1: geo feature m - 1 =
// input column 15 + m)
__device__ __forceinline__ ngp_h8 ft_wt_color_in(const _Float16* __restrict__ W, int c, int lane) {
    const int m = lane & 15, g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int j = 0; j < 8; j++) a[j] = m == 0 ? (_Float16)0.0f : W[ft_phi(c, g, j) * 32 + 15 + m];
    return a;
}

// selection fragments: element j of lane (n = lane & 15, g) is 1 when the feature at (g, j) of the source fragment is feature n of the wanted tile
//   0, 1 : chained fragment (order phi), tile parity h = 0, 1            2 : output-style fragment {rows 4g..4g+3, 0...}
//   3    : colour input, SH half (elements 4..7 = SH 4g..4g+3)           4 : colour input, geo half: column 16 + n <-> density output n + 1
//   5, 6 : encoded features (elements 2i, 2i+1 = level 4i + g), tile 0, 1
__device__ __forceinline__ ngp_h8 ft_sel(int which, int lane) {
    const int n = lane & 15, g = lane >> 4;
    ngp_h8 a;
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        bool one = false;
        if (which <= 1) one = (j >> 2) == which && n == 4 * g + (j & 3);
        else if (which == 2) one = j < 4 && n == 4 * g + j;
        else if (which == 3) one = j >= 4 && n == 4 * g + (j - 4);
        else if (which == 4) one = j < 4 && n + 1 == 4 * g + j;
        else {
            const int f = 8 * (j >> 1) + 2 * g + (j & 1);              // feature 2 * level + e, level = 4 (j >> 1) + g
            one = (f >> 4) == which - 5 && n == (f & 15);
        }
        a[j] = one ? (_Float16)1.0f : (_Float16)0.0f;
    }
    return a;
}

__device__ __forceinline__ void ft_stage_backward(const rf_params& P, ngp_h8* __restrict__ lds_b, int wave, int nwaves, int lane) {
    const _Float16* Ws = P.w_sigma;
    const _Float16* Wc = P.w_color;
    for (int f = wave; f < FT_NBWD + FT_NSEL; f += nwaves) {
        ngp_h8 a;
        if (f < 4) a = ft_wt_out(Wc + FT_C_OUT, 64, f, lane);                                   // step 1: c3 <- colour logits
        else if (f < 12) a = ft_wt_hidden(Wc + FT_C_HID2, 64, (f - 4) >> 1, (f - 4) & 1, lane);   // step 2: c2 <- c3
        else if (f < 20) a = ft_wt_hidden(Wc + FT_C_HID1, 64, (f - 12) >> 1, (f - 12) & 1, lane); // step 3: c1 <- c2
        else if (f < 22) a = ft_wt_color_in(Wc + FT_C_IN, f - 20, lane);                        // step 4: density outputs <- c1
        else if (f < 26) a = ft_wt_out(Ws + FT_S_OUT, 64, f - 22, lane);                        // step 5: h2 <- density outputs
        else if (f < 34) a = ft_wt_hidden(Ws + FT_S_HID, 64, (f - 26) >> 1, (f - 26) & 1, lane);  // step 6: h1 <- h2
        else if (f < 38) a = ft_wt_hidden(Ws + FT_S_IN, 32, (f - 34) >> 1, (f - 34) & 1, lane);   // step 7: encoded features <- h1
        else a = ft_sel(f - FT_NBWD, lane);
        lds_b[f * 64 + lane] = a;
    }
}

// gradient of two D tiles through the ReLU of the saved activation fragment (pass where forward > 0), rounded to half like the reference's
// backward buffer: the next step's B fragment
__device__ __forceinline__ ngp_h8 ft_mask_pack(ngp_f4 d0, ngp_f4 d1, ngp_h8 act) {
    ngp_h8 b;
    #pragma unroll
    for (int r = 0; r < 4; r++) {
        b[r] = (float)act[r] > 0.0f ? (_Float16)d0[r] : (_Float16)0.0f;
        b[4 + r] = (float)act[4 + r] > 0.0f ? (_Float16)d1[r] : (_Float16)0.0f;
    }
    return b;
}

// the K = 32 weight-gradient operand of one 16-feature tile: X^T of two sample tiles, 4 samples of each per lane
__device__ __forceinline__ ngp_h8 ft_transposed(ngp_h8 x0, ngp_h8 x1, ngp_h8 sel) {
    const ngp_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const ngp_f4 a = ngp_mfma(x0, sel, zero), b = ngp_mfma(x1, sel, zero);
    ngp_h8 o;
    #pragma unroll
    for (int r = 0; r < 4; r++) { o[r] = (_Float16)a[r]; o[4 + r] = (_Float16)b[r]; }
    return o;
}

template <bool FIXED>
__device__ __forceinline__ void ft_train_forward_loop(const rf_params& P, const rf_iter_class cls_rt, const rf_lane_levels& lv, const ngp_h8* __restrict__ lds_w,
                                                      const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                                                      float* __restrict__ sigmas, float* __restrict__ rgbs, ngp_h8* __restrict__ enc) {
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
                shq[n][j] = ngp_f2h(v);
            }
            x[n] = rf_encode<false>(P, lv, cls, px, py, pz);
            enc[(size_t)(pair * 2 + n) * 64 + lane] = x[n];        // tiles are padded to pairs: the buffer holds 2 * npairs tiles
        }
        float sg[2], cr[2], cg[2], cb[2];
        rv_mlp_tiles<2>(lds_w, lane, x, shq, sg, cr, cg, cb);
        #pragma unroll
        for (int n = 0; n < 2; n++)
            if (g == 0 && m[n] < M) {
                rv_activate(P, sg[n], cr[n], cg[n], cb[n]);
                sigmas[m[n]] = sg[n];
                rgbs[3ull * m[n]] = cr[n]; rgbs[3ull * m[n] + 1] = cg[n]; rgbs[3ull * m[n] + 2] = cb[n];
            }
    }
}

#ifndef FT_FWD_WG_PER_CU
#define FT_FWD_WG_PER_CU 3                // 144 VGPRs: three workgroups (12 waves) per CU hide more of the gather latency than two
#endif
__global__ __launch_bounds__(RF_BLOCK, FT_FWD_WG_PER_CU) void k_field_train_forward(rf_params P, const float* __restrict__ xyzs, const float* __restrict__ dirs,
                                                                      uint32_t M, float* __restrict__ sigmas, float* __restrict__ rgbs,
                                                                      ngp_h8* __restrict__ enc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rf_smem);
    const int lane = threadIdx.x & 63, g = lane >> 4, wave = threadIdx.x >> 6;
    rv_stage_weights(P, lds_w, wave, RF_BLOCK / 64, lane);
    rf_lane_levels lv;
    rf_setup_levels(P, g, lv);
    __syncthreads();
    const rf_iter_class cls = rf_classify(lv);
    if (cls.dense == 1u && cls.select == 2u && cls.hashed == 12u) ft_train_forward_loop<true>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs, enc);
    else ft_train_forward_loop<false>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs, enc);
}

// NT = 2 tiles through the networks, every hidden activation kept (B fragments), plus the raw outputs.  COLOR = false stops after the
// density net (the density part of the backward needs nothing else).
struct ft_acts {
    ngp_h8 h1[2][2], h2[2][2], cin[2], c1[2][2], c2[2][2], c3[2][2];
    ngp_f4 hs[2], ho[2];                                // density-net outputs (logit, geo), colour logits
};

template <bool COLOR>
__device__ __forceinline__ void ft_recompute(const ngp_h8* __restrict__ lds_w, int lane, const ngp_h8 (&x)[2], const ngp_h4 (&shq)[2], ft_acts& A) {
    const ngp_f4 zero = {0.f, 0.f, 0.f, 0.f};
    constexpr int NT = 2;
    {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w = rv_frag(lds_w, t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w, x[n], zero);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) { A.h1[n][0] = mlp_pack_relu(d[n][0], d[n][1]); A.h1[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
    }
    {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w0 = rv_frag(lds_w, 4 + 2 * t, lane), w1 = rv_frag(lds_w, 5 + 2 * t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w0, A.h1[n][0], zero);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w1, A.h1[n][1], d[n][t]);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) { A.h2[n][0] = mlp_pack_relu(d[n][0], d[n][1]); A.h2[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
    }
    if constexpr (!COLOR) return;
    {
        const ngp_h8 w0 = rv_frag(lds_w, 12, lane), w1 = rv_frag(lds_w, 13, lane);
        #pragma unroll
        for (int n = 0; n < NT; n++) A.hs[n] = ngp_mfma(w0, A.h2[n][0], zero);
        #pragma unroll
        for (int n = 0; n < NT; n++) A.hs[n] = ngp_mfma(w1, A.h2[n][1], A.hs[n]);
    }
    #pragma unroll
    for (int n = 0; n < NT; n++)
        #pragma unroll
        for (int j = 0; j < 4; j++) { A.cin[n][j] = (_Float16)A.hs[n][j]; A.cin[n][4 + j] = shq[n][j]; }
    {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w = rv_frag(lds_w, 14 + t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w, A.cin[n], zero);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) { A.c1[n][0] = mlp_pack_relu(d[n][0], d[n][1]); A.c1[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
    }
    #pragma unroll
    for (int l = 0; l < 2; l++) {
        ngp_f4 d[NT][MLP_MT];
        #pragma unroll
        for (int t = 0; t < MLP_MT; t++) {
            const ngp_h8 w0 = rv_frag(lds_w, 18 + 8 * l + 2 * t, lane), w1 = rv_frag(lds_w, 19 + 8 * l + 2 * t, lane);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w0, l == 0 ? A.c1[n][0] : A.c2[n][0], zero);
            #pragma unroll
            for (int n = 0; n < NT; n++) d[n][t] = ngp_mfma(w1, l == 0 ? A.c1[n][1] : A.c2[n][1], d[n][t]);
        }
        #pragma unroll
        for (int n = 0; n < NT; n++) {
            const ngp_h8 p0 = mlp_pack_relu(d[n][0], d[n][1]), p1 = mlp_pack_relu(d[n][2], d[n][3]);
            if (l == 0) { A.c2[n][0] = p0; A.c2[n][1] = p1; } else { A.c3[n][0] = p0; A.c3[n][1] = p1; }
        }
    }
    {
        const ngp_h8 w0 = rv_frag(lds_w, 34, lane), w1 = rv_frag(lds_w, 35, lane);
        #pragma unroll
        for (int n = 0; n < NT; n++) A.ho[n] = ngp_mfma(w0, A.c3[n][0], zero);
        #pragma unroll
        for (int n = 0; n < NT; n++) A.ho[n] = ngp_mfma(w1, A.c3[n][1], A.ho[n]);
    }
}

// acc[to * NI + ti] += G^T A over the 32 samples of the pair
template <int NO, int NI>
__device__ __forceinline__ void ft_wgrad(ngp_f4* __restrict__ acc, const ngp_h8 (&G)[NO], const ngp_h8 (&A)[NI]) {
    #pragma unroll
    for (int to = 0; to < NO; to++)
        #pragma unroll
        for (int ti = 0; ti < NI; ti++) acc[to * NI + ti] = ngp_mfma(G[to], A[ti], acc[to * NI + ti]);
}

// Accumulator tiles in the order of the backward steps.  Holding all 72 beside the activations of two tiles does not fit a wave's 512
// registers (it spilled 110), so the backward is two launches: PART 0 = the colour net (44 tiles; hands the gradient of the density net's
// 16 outputs on through HBM, 32 B per sample), PART 1 = the density net (28 tiles; recomputes its two hidden layers only).
//   colour out [1][4] | colour hidden 2 [4][4] | colour hidden 1 [4][4] | colour in [4][2]      density out [1][4] | density hidden [4][4] | density in [4][2]
static constexpr int FT_A_COUT = 0, FT_A_CH2 = 4, FT_A_CH1 = 20, FT_A_CIN = 36, FT_NACC_COLOR = 44;
static constexpr int FT_A_SOUT = 0, FT_A_SHID = 4, FT_A_SIN = 20, FT_NACC_SIGMA = 28;

#define FT_T4(X, c0, c1) {ft_transposed(X[0][c0], X[1][c0], sel_p0), ft_transposed(X[0][c0], X[1][c0], sel_p1), \
                          ft_transposed(X[0][c1], X[1][c1], sel_p0), ft_transposed(X[0][c1], X[1][c1], sel_p1)}

template <int PART>
__global__ __launch_bounds__(RF_BLOCK, 1) void k_field_train_backward(rf_params P, const ngp_h8* __restrict__ enc, const float* __restrict__ dirs, uint32_t M,
                                                                       const float* __restrict__ grad_sigmas, const float* __restrict__ grad_rgbs,
                                                                       ngp_h4* __restrict__ grad_outs, _Float16* __restrict__ grad_enc,
                                                                       float* __restrict__ wgrad_ws) {
    constexpr int NACC = PART == 0 ? FT_NACC_COLOR : FT_NACC_SIGMA;
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rf_smem);
    ngp_h8* lds_b = lds_w + 36 * 64;
    const ngp_h8* lds_sel = lds_b + FT_NBWD * 64;
    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15, wave_in_wg = threadIdx.x >> 6;
    rv_stage_weights(P, lds_w, wave_in_wg, RF_BLOCK / 64, lane);
    ft_stage_backward(P, lds_b, wave_in_wg, RF_BLOCK / 64, lane);
    __syncthreads();
    const ngp_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const ngp_h8 hzero = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    ngp_f4 acc[NACC];
    #pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = zero;

    const uint32_t wave = (blockIdx.x * RF_BLOCK + threadIdx.x) >> 6, nwaves = gridDim.x * (RF_BLOCK / 64);
    const uint32_t npairs = (M + 31) >> 5;
    for (uint32_t pair = wave; pair < npairs; pair += nwaves) {
        ngp_h8 x[2];
        ngp_h4 shq[2];
        uint32_t m[2];
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            m[n] = pair * 32 + 16 * n + s;
            if constexpr (PART == 0) {
                const uint64_t mm = m[n] < M ? m[n] : 0;
                float sh[16];
                sh_eval<4>(dirs[3 * mm], dirs[3 * mm + 1], dirs[3 * mm + 2], P.shn, sh);
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = sh[j];
                    if (g == 1) v = sh[4 + j];
                    if (g == 2) v = sh[8 + j];
                    if (g == 3) v = sh[12 + j];
                    shq[n][j] = ngp_f2h(v);
                }
            }
            x[n] = enc[(size_t)(pair * 2 + n) * 64 + lane];
        }
        ft_acts A;
        ft_recompute<PART == 0>(lds_w, lane, x, shq, A);
        const ngp_h8 sel_p0 = lds_sel[0 * 64 + lane], sel_p1 = lds_sel[1 * 64 + lane], sel_o = lds_sel[2 * 64 + lane];
        ngp_h8 G[2][2];                                           // current gradient, chained fragments [tile][k-step]

        if constexpr (PART == 0) {
            // ---- output gradients (lanes of group 0 hold rows 0..3 of both output tiles) ----
            ngp_h8 gout[2];                                       // fragments {rows 4g..4g+3, 0, 0, 0, 0}
            float sig_grad[2];
            #pragma unroll
            for (int n = 0; n < 2; n++) {
                gout[n] = hzero;
                sig_grad[n] = 0.0f;
                if (g == 0 && m[n] < M) {
                    #pragma unroll
                    for (int k = 0; k < 3; k++) {
                        // torch.sigmoid on the half logits, its backward on halves: half(float(half(grad)) * (1 - s) * s)   (opmath float)
                        const float sv = rf_h(1.0f / (1.0f + ngp_expf(-rf_h(A.ho[n][k]))));
                        const float gh = rf_h(grad_rgbs[3ull * m[n] + k]);
                        gout[n][k] = ngp_f2h((gh * (1.0f - sv)) * sv);
                    }
                    // trunc_exp backward (activation.py:17-21): g * exp(clamp(x, max = 15)), float32, then autograd's cast to the half input
                    sig_grad[n] = rf_h(grad_sigmas[m[n]] * ngp_expf(fminf(rf_h(A.hs[n][0]), 15.0f)));
                }
            }
            {   // step 1: c3 <- colour logits; dV_out = g_out^T c3
                ngp_f4 d[2][MLP_MT];
                #pragma unroll
                for (int t = 0; t < MLP_MT; t++) {
                    const ngp_h8 w = lds_b[t * 64 + lane];
                    #pragma unroll
                    for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w, gout[n], zero);
                }
                const ngp_h8 Gt[1] = {ft_transposed(gout[0], gout[1], sel_o)};
                const ngp_h8 At[4] = FT_T4(A.c3, 0, 1);
                ft_wgrad<1, 4>(acc + FT_A_COUT, Gt, At);
                #pragma unroll
                for (int n = 0; n < 2; n++) { G[n][0] = ft_mask_pack(d[n][0], d[n][1], A.c3[n][0]); G[n][1] = ft_mask_pack(d[n][2], d[n][3], A.c3[n][1]); }
            }
            #pragma unroll
            for (int l = 0; l < 2; l++) {   // steps 2, 3: c2 <- c3 (dV_hid2 = g_c3^T c2), c1 <- c2 (dV_hid1 = g_c2^T c1)
                const ngp_h8 Gt[4] = FT_T4(G, 0, 1);
                if (l == 0) { const ngp_h8 At[4] = FT_T4(A.c2, 0, 1); ft_wgrad<4, 4>(acc + FT_A_CH2, Gt, At); }
                else        { const ngp_h8 At[4] = FT_T4(A.c1, 0, 1); ft_wgrad<4, 4>(acc + FT_A_CH1, Gt, At); }
                ngp_f4 d[2][MLP_MT];
                #pragma unroll
                for (int t = 0; t < MLP_MT; t++) {
                    const ngp_h8 w0 = lds_b[(4 + 8 * l + 2 * t) * 64 + lane], w1 = lds_b[(5 + 8 * l + 2 * t) * 64 + lane];
                    #pragma unroll
                    for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w0, G[n][0], zero);
                    #pragma unroll
                    for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w1, G[n][1], d[n][t]);
                }
                #pragma unroll
                for (int n = 0; n < 2; n++) { G[n][0] = ft_mask_pack(d[n][0], d[n][1], l == 0 ? A.c2[n][0] : A.c1[n][0]);
                                              G[n][1] = ft_mask_pack(d[n][2], d[n][3], l == 0 ? A.c2[n][1] : A.c1[n][1]); }
            }
            {   // step 4: density outputs <- c1 (the geo columns of the colour input); dV_in = g_c1^T cin
                const ngp_h8 Gt[4] = FT_T4(G, 0, 1);
                const ngp_h8 At[2] = {ft_transposed(A.cin[0], A.cin[1], lds_sel[3 * 64 + lane]), ft_transposed(A.cin[0], A.cin[1], lds_sel[4 * 64 + lane])};
                ft_wgrad<4, 2>(acc + FT_A_CIN, Gt, At);
                const ngp_h8 w0 = lds_b[20 * 64 + lane], w1 = lds_b[21 * 64 + lane];
                #pragma unroll
                for (int n = 0; n < 2; n++) {
                    ngp_f4 d = ngp_mfma(w0, G[n][0], zero);
                    d = ngp_mfma(w1, G[n][1], d);
                    ngp_h4 go;
                    #pragma unroll
                    for (int r = 0; r < 4; r++) go[r] = (_Float16)d[r];
                    if (g == 0) go[0] = ngp_f2h(sig_grad[n]);       // row 0 is the density logit: its gradient comes from trunc_exp
                    grad_outs[(size_t)(pair * 2 + n) * 64 + lane] = go;   // rows 4g..4g+3 of the 16 density-net output gradients of sample s
                }
            }
        } else {
            ngp_h8 gsig[2];
            #pragma unroll
            for (int n = 0; n < 2; n++) {
                const ngp_h4 go = grad_outs[(size_t)(pair * 2 + n) * 64 + lane];
                gsig[n] = hzero;
                #pragma unroll
                for (int r = 0; r < 4; r++) gsig[n][r] = go[r];
            }
            {   // step 5: h2 <- density outputs; dW_out = g_outs^T h2
                const ngp_h8 Gt[1] = {ft_transposed(gsig[0], gsig[1], sel_o)};
                const ngp_h8 At[4] = FT_T4(A.h2, 0, 1);
                ft_wgrad<1, 4>(acc + FT_A_SOUT, Gt, At);
                ngp_f4 d[2][MLP_MT];
                #pragma unroll
                for (int t = 0; t < MLP_MT; t++) {
                    const ngp_h8 w = lds_b[(22 + t) * 64 + lane];
                    #pragma unroll
                    for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w, gsig[n], zero);
                }
                #pragma unroll
                for (int n = 0; n < 2; n++) { G[n][0] = ft_mask_pack(d[n][0], d[n][1], A.h2[n][0]); G[n][1] = ft_mask_pack(d[n][2], d[n][3], A.h2[n][1]); }
            }
            {   // step 6: h1 <- h2; dW_hid = g_h2^T h1
                const ngp_h8 Gt[4] = FT_T4(G, 0, 1);
                const ngp_h8 At[4] = FT_T4(A.h1, 0, 1);
                ft_wgrad<4, 4>(acc + FT_A_SHID, Gt, At);
                ngp_f4 d[2][MLP_MT];
                #pragma unroll
                for (int t = 0; t < MLP_MT; t++) {
                    const ngp_h8 w0 = lds_b[(26 + 2 * t) * 64 + lane], w1 = lds_b[(27 + 2 * t) * 64 + lane];
                    #pragma unroll
                    for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w0, G[n][0], zero);
                    #pragma unroll
                    for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w1, G[n][1], d[n][t]);
                }
                #pragma unroll
                for (int n = 0; n < 2; n++) { G[n][0] = ft_mask_pack(d[n][0], d[n][1], A.h1[n][0]); G[n][1] = ft_mask_pack(d[n][2], d[n][3], A.h1[n][1]); }
            }
            {   // step 7: encoded features <- h1; dW_in = g_h1^T enc; the gradient goes to HBM level-major, [L][M][2] halves, for the table scatter
                const ngp_h8 Gt[4] = FT_T4(G, 0, 1);
                const ngp_h8 At[2] = {ft_transposed(x[0], x[1], lds_sel[5 * 64 + lane]), ft_transposed(x[0], x[1], lds_sel[6 * 64 + lane])};
                ft_wgrad<4, 2>(acc + FT_A_SIN, Gt, At);
                #pragma unroll
                for (int t = 0; t < 2; t++) {
                    const ngp_h8 w0 = lds_b[(34 + 2 * t) * 64 + lane], w1 = lds_b[(35 + 2 * t) * 64 + lane];
                    #pragma unroll
                    for (int n = 0; n < 2; n++) {
                        ngp_f4 d = ngp_mfma(w0, G[n][0], zero);
                        d = ngp_mfma(w1, G[n][1], d);
                        if (m[n] < M) {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const uint32_t level = 8 * t + 2 * g;   // rows 4g + r = features 16 t + 4 g + r = levels 8 t + 2 g, + 1
                            h2 lo, hi;
                            lo.x = (_Float16)d[0]; lo.y = (_Float16)d[1]; hi.x = (_Float16)d[2]; hi.y = (_Float16)d[3];
                            *reinterpret_cast<h2*>(grad_enc + ((size_t)level * M + m[n]) * 2) = lo;
                            *reinterpret_cast<h2*>(grad_enc + ((size_t)(level + 1) * M + m[n]) * 2) = hi;
                        }
                    }
                }
            }
        }
    }

    // ---- the workgroup's weight gradients: sum the four waves through LDS, then one f32 atomic per element into the workspace ----
    __syncthreads();                                               // every wave is done with the fragments
    float* lds_acc = reinterpret_cast<float*>(rf_smem);           // NACC x 256 f32 (44 KiB / 28 KiB)
    for (int w = 0; w < (int)(RF_BLOCK / 64); w++) {
        if (wave_in_wg == w) {
            #pragma unroll
            for (int k = 0; k < NACC; k++)
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    float* p = lds_acc + (k * 4 + r) * 64 + lane;
                    *p = w == 0 ? acc[k][r] : *p + acc[k][r];
                }
        }
        __syncthreads();
    }
    // element (k, r, lane) = dW[16 to + 4 g + r][16 ti + s] of the layer that owns tile k
    for (int e = threadIdx.x; e < NACC * 256; e += RF_BLOCK) {
        const int k = e >> 8, r = (e >> 6) & 3, l = e & 63, gg = l >> 4, ss = l & 15;
        int base, ld, to, ti;
        if constexpr (PART == 0) {
            if (k < FT_A_CH2)        { base = FT_C_OUT;  ld = 64; to = 0;                   ti = k - FT_A_COUT; }
            else if (k < FT_A_CH1)   { base = FT_C_HID2; ld = 64; to = (k - FT_A_CH2) >> 2; ti = (k - FT_A_CH2) & 3; }
            else if (k < FT_A_CIN)   { base = FT_C_HID1; ld = 64; to = (k - FT_A_CH1) >> 2; ti = (k - FT_A_CH1) & 3; }
            else                     { base = FT_C_IN;   ld = 32; to = (k - FT_A_CIN) >> 1; ti = (k - FT_A_CIN) & 1; }
            base += 7168;
        } else {
            if (k < FT_A_SHID)       { base = FT_S_OUT;  ld = 64; to = 0;                    ti = k - FT_A_SOUT; }
            else if (k < FT_A_SIN)   { base = FT_S_HID;  ld = 64; to = (k - FT_A_SHID) >> 2; ti = (k - FT_A_SHID) & 3; }
            else                     { base = FT_S_IN;   ld = 32; to = (k - FT_A_SIN) >> 1;  ti = (k - FT_A_SIN) & 1; }
        }
        const float v = lds_acc[e];
        if (v != 0.0f) unsafeAtomicAdd(wgrad_ws + base + (16 * to + 4 * gg + r) * ld + 16 * ti + ss, v);
    }
}

// workspace -> gradients of FFMLP.weights: rounded to half as the reference's grad_weights are, returned as float32; the workspace is cleared
__global__ __launch_bounds__(256) void k_field_train_wgrad_finish(float* __restrict__ ws, float* __restrict__ grad_sigma_w, float* __restrict__ grad_color_w) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= FT_WS_FLOATS) return;
    const float v = rf_h(ws[e]);
    ws[e] = 0.0f;
    if (e < 7168) grad_sigma_w[e] = v; else grad_color_w[e - 7168] = v;
}

extern "C" size_t ngp_field_train_saved_bytes(uint32_t M) { return (size_t)((M + 31) >> 5) * 2 * 64 * sizeof(ngp_h8); }
// workspace: [f32 weight-gradient accumulators, kept zero between calls | the density-net output gradients of M samples]
static size_t ft_ws_outs_offset() { return (FT_WS_FLOATS * sizeof(float) + 255) & ~(size_t)255; }
extern "C" size_t ngp_field_train_workspace(uint32_t M) { return ft_ws_outs_offset() + (size_t)((M + 31) >> 5) * 2 * 64 * sizeof(ngp_h4); }

extern "C" int ngp_field_train_forward(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                                       float* sigmas, float* rgbs, void* saved, size_t saved_bytes, void* stream) {
    rf_params P;
    int rc = rf_fill_params("field_train_forward", field_host, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && dirs && sigmas && rgbs && saved, "field_train_forward: null pointer");
    NGP_REQUIRE(saved_bytes >= ngp_field_train_saved_bytes(M), "field_train_forward: saved buffer too small (%zu < %zu bytes)", saved_bytes,
                ngp_field_train_saved_bytes(M));
    const uint32_t npairs = (M + 31) >> 5;
    uint32_t blocks = ngp_div_up(npairs, RF_BLOCK / 64);
    if (blocks > 256 * FT_FWD_WG_PER_CU) blocks = 256 * FT_FWD_WG_PER_CU;
    hipLaunchKernelGGL(k_field_train_forward, dim3(blocks), dim3(RF_BLOCK), 36 * 1024, (hipStream_t)stream, P, xyzs, dirs, M, sigmas, rgbs, (ngp_h8*)saved);
    NGP_CHECK_LAUNCH("field_train_forward");
    return NGP_OK;
}

static std::atomic<unsigned> ft_big_lds_set{0};

extern "C" int ngp_field_train_backward(const ngp_field_t* field_host, const void* saved, const float* dirs, uint32_t M,
                                        const float* grad_sigmas, const float* grad_rgbs, void* grad_enc,
                                        float* grad_sigma_weights, float* grad_color_weights, void* workspace, size_t workspace_bytes, void* stream) {
    rf_params P;
    int rc = rf_fill_params("field_train_backward", field_host, P);
    if (rc != NGP_OK) return rc;
    NGP_REQUIRE(grad_sigma_weights && grad_color_weights && workspace, "field_train_backward: null pointer");
    NGP_REQUIRE(workspace_bytes >= ngp_field_train_workspace(M), "field_train_backward: workspace too small");
    if (M > 0) {
        NGP_REQUIRE(saved && dirs && grad_sigmas && grad_rgbs && grad_enc, "field_train_backward: null pointer");
        int dev = 0;
        NGP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32, "field_train_backward: no current device");
        if (!(ft_big_lds_set.load(std::memory_order_acquire) & (1u << dev))) {
            NGP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(k_field_train_backward<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(k_field_train_backward<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS) == hipSuccess,
                        "field_train_backward: cannot reserve %u bytes of LDS", FT_LDS);
            ft_big_lds_set.fetch_or(1u << dev, std::memory_order_release);
        }
        const uint32_t npairs = (M + 31) >> 5;
        uint32_t blocks = ngp_div_up(npairs, RF_BLOCK / 64);
        if (blocks > 256) blocks = 256;
        ngp_h4* grad_outs = reinterpret_cast<ngp_h4*>(static_cast<unsigned char*>(workspace) + ft_ws_outs_offset());
        hipLaunchKernelGGL(k_field_train_backward<0>, dim3(blocks), dim3(RF_BLOCK), FT_LDS, (hipStream_t)stream, P, (const ngp_h8*)saved, dirs, M,
                           grad_sigmas, grad_rgbs, grad_outs, (_Float16*)grad_enc, (float*)workspace);
        NGP_CHECK_LAUNCH("field_train_backward (colour net)");
        hipLaunchKernelGGL(k_field_train_backward<1>, dim3(blocks), dim3(RF_BLOCK), FT_LDS, (hipStream_t)stream, P, (const ngp_h8*)saved, dirs, M,
                           grad_sigmas, grad_rgbs, grad_outs, (_Float16*)grad_enc, (float*)workspace);
        NGP_CHECK_LAUNCH("field_train_backward (density net)");
    }
    hipLaunchKernelGGL(k_field_train_wgrad_finish, dim3(ngp_div_up(FT_WS_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, (float*)workspace,
                       grad_sigma_weights, grad_color_weights);
    NGP_CHECK_LAUNCH("field_train_wgrad_finish");
    return NGP_OK;
}
