// ngp_field.h -- the field (hash grid -> density MLP -> SH -> colour MLP) as the fused kernels evaluate it: the encoder in matrix-operand
// layout, the 36 weight fragments in LDS and the two networks on 16-sample tiles.  Shared by render_fused.hip (frame and explicit-point
// inference kernels) and field_train.hip (the training step); see the header of render_fused.hip for the work mapping.
#pragma once
#include "ngp_mlp.h"
#include "ngp_sh.h"

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
    float inv_b2;                     // fl(1 / (2 bound)): the reciprocal torch multiplies by
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

#ifndef RF_PAIR_HASHED
#define RF_PAIR_HASHED 0               // 1: on hashed levels an even-x lane fetches both x-corners with ONE aligned 8-byte load; A/B: HISTORY 4
#endif
__device__ __forceinline__ rf_row2 rf_rows8(const rf_params& P, uint32_t byte_off) {      // one 8-byte load, 8-byte aligned
    asm("" : "+v"(byte_off));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const u2 v = *reinterpret_cast<const u2*>(reinterpret_cast<const char*>(P.table) + byte_off);
    return rf_row2{v.x, v.y};
}

__device__ __forceinline__ void rf_normalise(const rf_params& P, float wx, float wy, float wz, float& x0, float& x1, float& x2) {
    // GridEncoder.forward (grid.py:144): (x + bound) / (2 bound) -- as torch evaluates it on the GPU: a tensor divided by a host scalar is the
    // product with the scalar's binary32 reciprocal (ATen BinaryDivTrueKernel.cu: "compute a * reciprocal(b)").  For a power-of-two 2 * bound
    // that is the exact quotient; for any other bound it is what the reference's encoder sees, one rounding away from the quotient at times.
    x0 = (wx + P.bound) * P.inv_b2; x1 = (wy + P.bound) * P.inv_b2; x2 = (wz + P.bound) * P.inv_b2;
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
#if RF_PAIR_HASHED
                // The x-neighbour of a hashed corner: fast_hash xors x * 1 into the index (gridencoder.cu:35-51), so for an EVEN cell coordinate
                // index(x + 1, y, z) = index(x, y, z) ^ 1 (2^k rows, k >= 1: the mask keeps bit 0): both rows lie in one aligned 8-byte word and ONE
                // load serves both corners -- which half is which depends on bit 0 of the (y, z) hash.  Lanes with an odd x fetch their second
                // corner with a load of their own, the even lanes masked off: the texture path is busy per lane address (HISTORY 1), and these
                // levels go from 8 to 6 of them per sample on average.  Same rows, same bits.
                {
                    const uint32_t yz[4] = {yz0, yz1, yz2, yz3};
                    const bool odd = (gx & 1u) != 0u;
                    #pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t A = a0 ^ yz[k];                                 // byte offset of corner (x, y, z) inside the level
                        const rf_row2 r = rf_rows8(P, (A & ~4u) + b);
                        const bool hi = (A & 4u) != 0u;
                        raw[i & 1][2 * k] = hi ? r.hi : r.lo;
                        raw[i & 1][2 * k + 1] = hi ? r.lo : r.hi;                      // row A ^ 4: corner (x + 1, y, z) when x is even
                    }
                    if (odd) {
                        #pragma unroll
                        for (int k = 0; k < 4; k++) raw[i & 1][2 * k + 1] = rf_row(P, (a1 ^ yz[k]) + b);
                    }
                    continue;
                }
#endif
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


static inline int rf_fill_params(const char* who, const ngp_field_t* f, rf_params& P) {
    NGP_REQUIRE(f && f->embeddings && f->offsets && f->sigma_weights && f->color_weights, "%s: null field pointer", who);
    NGP_REQUIRE(f->L == RF_L, "%s: the fused path is built for 16 levels x 2 features (the reference's hashgrid)", who);
    NGP_REQUIRE(f->bound > 0, "%s: bound must be positive", who);
    P.table = (const uint32_t*)f->embeddings;
    P.offsets = f->offsets;
    P.w_sigma = (const _Float16*)f->sigma_weights;
    P.w_color = (const _Float16*)f->color_weights;
    P.bound = f->bound;
    P.density_scale = f->density_scale;
    P.inv_b2 = 1.0f / (2.0f * f->bound);
    for (int l = 0; l < RF_L; l++) {
        P.scale[l] = exp2f((float)l * f->S) * (float)f->H - 1.0f;
        P.resolution[l] = (uint32_t)ceilf(P.scale[l]) + 1u;
    }
    sh_fill_norm(P.shn);
    return NGP_OK;
}

static constexpr int RV_NFRAG = 36;                                    // MFMA weight fragments of both networks (14 + 22)
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

