/*
 * ngp_oracle.c -- CPU restatement of the reference's Instant-NGP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nerf-navigation_amd/ may include,
 * link or call this file; it is the checker used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.
 *
 * Each function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it follows.  The reference kernels are CUDA and cannot be
 * built or run here (no nvcc, no GPU), and the reference ships no tests or
 * golden vectors, so:
 *
 *   PARITY UNPINNED by the reference's own fixtures, except for
 *     - pcg32: the published PCG32 demo known-answer vector (seed 42, seq 54)
 *     - trunc_exp: vectors generated from the importable reference
 *       activation.py (tests/golden/trunc_exp.npz)
 *   everything else is pinned by the internal-consistency relations listed in
 *   SURVEY.md 8(c) (tests/test_oracle_*.py).
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md "Numerics"):
 *   - IEEE binary32, round-to-nearest-even, no FMA contraction
 *     (built with -ffp-contract=off; fmaf() only where written).
 *   - The reference's __expf (a CUDA fast-math intrinsic that cannot be
 *     reproduced bit-for-bit off NVIDIA silicon) is replaced on BOTH sides by
 *     the deterministic o_expf() below.
 *   - exp2f(level*S) of the grid encoder is evaluated on the host (libm) and
 *     handed to both implementations as a per-level table.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define O_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* helpers                                                            */
/* ------------------------------------------------------------------ */

static inline float clampf(float x, float lo, float hi) {
    /* raymarching/src/raymarching.cu:36-38 */
    return fminf(hi, fmaxf(lo, x));
}

static inline float sign1f(float x) {
    /* raymarching.cu:32-34 : copysignf(1, x); -0.0 maps to -1 */
    return copysignf(1.0f, x);
}

/* deterministic exp used in place of __expf (raymarching.cu:547,650,869).
 * n = rint(x*log2e); r = x - n*ln2 (two-term Cody-Waite, fmaf);
 * degree-7 Horner in fmaf; scale by 2^n through the exponent field.
 * Defined for x in [-87, 88]; below -87 the result is 0, above 88 +inf. */
static inline float o_expf(float x) {
    if (!(x >= -87.0f)) { return (x != x) ? x : 0.0f; }
    if (x > 88.0f) return INFINITY;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723e-06f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union { float f; int32_t i; } u;
    u.f = p;
    u.i += ((int32_t)n) << 23;
    return u.f;
}

/* OpenMP team size of the checker's parallel loops (the runtime is usually loaded before this library, so OMP_NUM_THREADS set later is not read) */
#ifdef _OPENMP
#include <omp.h>
O_API int o_set_threads(int n) { const int before = omp_get_max_threads(); if (n > 0) omp_set_num_threads(n); return before; }
#else
O_API int o_set_threads(int n) { (void)n; return 1; }
#endif

O_API void o_expf_array(const float* x, float* y, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) y[i] = o_expf(x[i]);
}

/* IEEE binary16 <-> binary32, software, round-to-nearest-even.
 * (c10::Half(float) in device code is __float2half = RNE.) */
static inline uint16_t f2h(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    const uint32_t sign = (v.u >> 16) & 0x8000u;
    uint32_t a = v.u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x0200u : 0u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           /* rounds to inf */
    if (a < 0x33000001u) return (uint16_t)sign;                         /* rounds to +-0 */
    if (a < 0x38800000u) {                                              /* subnormal half */
        const uint32_t e = a >> 23;
        const uint32_t m = (a & 0x7fffffu) | 0x800000u;
        const uint32_t shift = 126u - e;           /* 14..24 */
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((a - 0x38000000u) >> 13);
    const uint32_t rem = a & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

static inline float h2f(uint16_t h) {
    union { float f; uint32_t u; } v;
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) { v.u = sign; return v.f; }
        e = 113;
        while (!(m & 0x400u)) { m <<= 1; e--; }
        m &= 0x3ffu;
        v.u = sign | (e << 23) | (m << 13);
        return v.f;
    }
    if (e == 31) { v.u = sign | 0x7f800000u | (m << 13); return v.f; }
    v.u = sign | ((e + 112u) << 23) | (m << 13);
    return v.f;
}

O_API void o_f32_to_f16(const float* x, uint16_t* y, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) y[i] = f2h(x[i]);
}
O_API void o_f16_to_f32(const uint16_t* x, float* y, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) y[i] = h2f(x[i]);
}

/* ------------------------------------------------------------------ */
/* pcg32  (raymarching/src/pcg32.h:44-205)                             */
/* ------------------------------------------------------------------ */

#define O_PCG_MULT 0x5851f42d4c957f2dULL

typedef struct { uint64_t state, inc; } o_pcg32;

static inline uint32_t pcg_next(o_pcg32* g) {
    /* pcg32.h:66-72 : XSH-RR output of the old state */
    const uint64_t s = g->state;
    g->state = s * O_PCG_MULT + g->inc;
    const uint32_t xs = (uint32_t)(((s >> 18u) ^ s) >> 27u);
    const uint32_t rot = (uint32_t)(s >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
}

static inline void pcg_seed(o_pcg32* g, uint64_t initstate, uint64_t initseq) {
    /* pcg32.h:58-64 */
    g->state = 0u;
    g->inc = (initseq << 1u) | 1u;
    pcg_next(g);
    g->state += initstate;
    pcg_next(g);
}

static inline void pcg_advance(o_pcg32* g, uint64_t delta) {
    /* pcg32.h:149-170 : Brown's O(log n) jump-ahead */
    uint64_t cm = O_PCG_MULT, cp = g->inc, am = 1u, ap = 0u;
    while (delta > 0) {
        if (delta & 1u) { am *= cm; ap = ap * cm + cp; }
        cp = (cm + 1u) * cp;
        cm *= cm;
        delta >>= 1;
    }
    g->state = am * g->state + ap;
}

static inline float pcg_next_float(o_pcg32* g) {
    /* pcg32.h:107-116 : [1,2) mantissa trick minus 1 */
    union { uint32_t u; float f; } x;
    x.u = (pcg_next(g) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}

/* out[i] = pcg32{seed, seq}.advance(adv[i]).next_uint() ; floats likewise */
O_API void o_pcg32_kat(uint64_t seed, uint64_t seq, const uint64_t* adv, uint32_t n,
                       uint32_t* out_u, float* out_f) {
    for (uint32_t i = 0; i < n; i++) {
        o_pcg32 g; pcg_seed(&g, seed, seq);
        pcg_advance(&g, adv[i]);
        o_pcg32 h = g;
        out_u[i] = pcg_next(&g);
        if (out_f) out_f[i] = pcg_next_float(&h);
    }
}

/* first `n` outputs of the stream (the published PCG demo vector) */
O_API void o_pcg32_stream(uint64_t seed, uint64_t seq, uint32_t n, uint32_t* out) {
    o_pcg32 g; pcg_seed(&g, seed, seq);
    for (uint32_t i = 0; i < n; i++) out[i] = pcg_next(&g);
}

/* ------------------------------------------------------------------ */
/* Morton codes, bit packing (raymarching.cu:58-83,216-302)            */
/* ------------------------------------------------------------------ */

static inline uint32_t spread3(uint32_t v) {
    /* raymarching.cu:58-65 : 10 bits -> every third bit */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    /* raymarching.cu:67-73 */
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}

static inline uint32_t compact3(uint32_t x) {
    /* raymarching.cu:75-83 */
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

O_API void o_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    /* raymarching.cu:216-228 */
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}

O_API void o_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    /* raymarching.cu:239-256 ; `ind >> k` is an arithmetic shift of an int */
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n];
        coords[3 * n + 0] = (int32_t)compact3((uint32_t)(ind >> 0));
        coords[3 * n + 1] = (int32_t)compact3((uint32_t)(ind >> 1));
        coords[3 * n + 2] = (int32_t)compact3((uint32_t)(ind >> 2));
    }
}

O_API void o_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bitfield) {
    /* raymarching.cu:270-291 : bit i of byte n = grid[8n+i] > thresh, LSB first */
    #pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[8u * n + i] > thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* ------------------------------------------------------------------ */
/* near/far, sphere coordinates (raymarching.cu:94-211)                */
/* ------------------------------------------------------------------ */

O_API void o_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                                uint32_t N, float min_near, float* nears, float* fars) {
    /* raymarching.cu:94-147 : slab test, miss => FLT_MAX both */
    #pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float rdx = 1.0f / rays_d[3 * n], rdy = 1.0f / rays_d[3 * n + 1], rdz = 1.0f / rays_d[3 * n + 2];
        float tn = (aabb[0] - ox) * rdx, tf = (aabb[3] - ox) * rdx;
        if (tn > tf) { float s = tn; tn = tf; tf = s; }
        float yn = (aabb[1] - oy) * rdy, yf = (aabb[4] - oy) * rdy;
        if (yn > yf) { float s = yn; yn = yf; yf = s; }
        if (tn > yf || yn > tf) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (yn > tn) tn = yn;
        if (yf < tf) tf = yf;
        float zn = (aabb[2] - oz) * rdz, zf = (aabb[5] - oz) * rdz;
        if (zn > zf) { float s = zn; zn = zf; zf = s; }
        if (tn > zf || zn > tf) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (zn > tn) tn = zn;
        if (zf < tf) tf = zf;
        if (tn < min_near) tn = min_near;
        nears[n] = tn;
        fars[n] = tf;
    }
}

O_API void o_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    /* raymarching.cu:165-200 ; libm sqrtf/atan2f, so compare with tolerance */
    const float RPI = 0.3183098861837907f;
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float C = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[2 * n] = 2 * theta * RPI - 1;
        coords[2 * n + 1] = phi * RPI;
    }
}

/* ------------------------------------------------------------------ */
/* the march step shared by the three march kernels                    */
/* (raymarching.cu:363-404, 431-483, 759-813)                          */
/* ------------------------------------------------------------------ */

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Cf;
    uint32_t C, H;
    const uint8_t* grid;
} o_march;

/* Guard on the empty-space skip loop.  The reference loop (raymarching.cu:
 * 400-402) never terminates when tt is +inf (a ray with a zero direction);
 * both this oracle and the HIP kernels give up after this many substeps. */
#define O_SKIP_GUARD 65536

static inline void march_setup(o_march* m, const float* o, const float* d, float bound, float dt_gamma,
                               uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid) {
    m->ox = o[0]; m->oy = o[1]; m->oz = o[2];
    m->dx = d[0]; m->dy = d[1]; m->dz = d[2];
    m->rdx = 1.0f / m->dx; m->rdy = 1.0f / m->dy; m->rdz = 1.0f / m->dz;
    m->bound = bound; m->dt_gamma = dt_gamma;
    m->C = C; m->H = H; m->Hf = (float)H; m->Cf = (float)C;
    m->rH = 1.0f / (float)H;
    m->H3 = (float)(H * H * H);                                   /* :342 uint product -> float */
    m->dt_min = (2.0f * 1.7320508075688772f) / (float)max_steps;  /* :347 */
    m->dt_max = ((2.0f * 1.7320508075688772f) * (float)(1 << (C - 1))) / (float)H; /* :348 */
    m->grid = grid;
}

static inline int mip_from_exponent(int e, float max_cascade) {
    /* raymarching.cu:44-56 : float min/max, then truncation to int */
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}

/* Probe the occupancy grid at parameter *t.  Returns 1 with (x,y,z,dt) if the
 * cell is occupied (caller emits the sample and advances t by dt); returns 0
 * after advancing *t past the empty cell. */
static inline int march_probe(const o_march* m, float* t, float* px, float* py, float* pz, float* pdt) {
    const float tc = *t;
    const float x = clampf(m->ox + tc * m->dx, -m->bound, m->bound);
    const float y = clampf(m->oy + tc * m->dy, -m->bound, m->bound);
    const float z = clampf(m->oz + tc * m->dz, -m->bound, m->bound);
    const float dt = clampf(tc * m->dt_gamma, m->dt_min, m->dt_max);

    int e_pos, e_dt;
    (void)frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e_pos);
    (void)frexpf((float)((double)(dt * m->Hf) * 0.5), &e_dt);
    const int lp = mip_from_exponent(e_pos, m->Cf);
    const int ld = mip_from_exponent(e_dt, m->Cf);
    const int level = lp > ld ? lp : ld;

    const float mip_bound = fminf((float)(1 << level), m->bound);
    const float mip_rbound = 1.0f / mip_bound;

    /* :378-380 ; the literal 0.5 promotes the product to double, clamp() narrows it */
    const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1.0f) * (double)m->H), 0.0f, (float)(m->H - 1));
    const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1.0f) * (double)m->H), 0.0f, (float)(m->H - 1));
    const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1.0f) * (double)m->H), 0.0f, (float)(m->H - 1));

    /* :382 ; int*float + uint -> float -> uint32 */
    const uint32_t index = (uint32_t)((float)level * m->H3 + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = m->grid[index / 8] & (1 << (index % 8));

    if (occ) {
        *px = x; *py = y; *pz = z; *pdt = dt;
        return 1;
    }
    /* :394-402 ; distance to the far face of the cell along the ray, then substeps */
    const float tx = (((((float)nx + 0.5f + 0.5f * sign1f(m->dx)) * m->rH) * 2.0f - 1.0f) * mip_bound - x) * m->rdx;
    const float ty = (((((float)ny + 0.5f + 0.5f * sign1f(m->dy)) * m->rH) * 2.0f - 1.0f) * mip_bound - y) * m->rdy;
    const float tz = (((((float)nz + 0.5f + 0.5f * sign1f(m->dz)) * m->rH) * 2.0f - 1.0f) * mip_bound - z) * m->rdz;
    const float tt = tc + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    float tn = tc;
    int guard = 0;
    do {
        tn += clampf(tn * m->dt_gamma, m->dt_min, m->dt_max);
    } while (tn < tt && ++guard < O_SKIP_GUARD);
    *t = tn;
    return 0;
}

/* ------------------------------------------------------------------ */
/* training march (raymarching.cu:314-495)                             */
/* ------------------------------------------------------------------ */

/* Slot allocation: the reference reserves slots with atomicAdd in whatever
 * order threads arrive (raymarching.cu:409-410).  Any arrival order is a valid
 * outcome; oracle and HIP both realise the order "ray 0, ray 1, ...", i.e.
 * ray_index = n and point_index = counter[0] + exclusive prefix sum. */
O_API void o_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid,
                              float bound, float dt_gamma, uint32_t max_steps,
                              uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                              const float* nears, const float* fars,
                              float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                              uint32_t perturb) {
    uint32_t* nsteps = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    float* t0s = (float*)malloc(sizeof(float) * (N ? N : 1));

    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        o_march m;
        march_setup(&m, rays_o + 3 * n, rays_d + 3 * n, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t0 = nears[n];
        if (perturb) {                                            /* :352-355 ; pcg32{42}.advance(n) (:489) */
            o_pcg32 g; pcg_seed(&g, 42u, 1u);
            pcg_advance(&g, (uint64_t)n);
            t0 += m.dt_min * pcg_next_float(&g);
        }
        t0s[n] = t0;
        float t = t0, x, y, z, dt;
        uint32_t k = 0;
        while (t < far && k < max_steps) {                        /* :363 */
            if (march_probe(&m, &t, &x, &y, &z, &dt)) { k++; t += dt; }
        }
        nsteps[n] = k;
    }

    uint32_t base = (uint32_t)counter[0];
    const uint32_t ray_base = (uint32_t)counter[1];
    uint32_t* offs = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    for (uint32_t n = 0; n < N; n++) { offs[n] = base; base += nsteps[n]; }
    counter[0] = (int32_t)base;
    counter[1] = (int32_t)(ray_base + N);

    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t slot = ray_base + n;
        const uint32_t point_index = offs[n], num_steps = nsteps[n];
        rays[3 * slot] = (int32_t)n;                               /* :415-417 */
        rays[3 * slot + 1] = (int32_t)point_index;
        rays[3 * slot + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps >= M) continue;                /* :420 */
        o_march m;
        march_setup(&m, rays_o + 3 * n, rays_d + 3 * n, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t = t0s[n], last_t = t, x, y, z, dt;
        uint32_t k = 0;
        float* px = xyzs + 3 * (uint64_t)point_index;
        float* pd = dirs + 3 * (uint64_t)point_index;
        float* pl = deltas + 2 * (uint64_t)point_index;
        while (t < far && k < num_steps) {                        /* :431-483 */
            if (march_probe(&m, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                t += dt;
                pl[0] = dt; pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2; k++;
            }
        }
    }
    free(nsteps); free(t0s); free(offs);
}

/* ------------------------------------------------------------------ */
/* training composite (raymarching.cu:506-688)                         */
/* ------------------------------------------------------------------ */

O_API void o_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                          const int32_t* rays, uint32_t M, uint32_t N,
                                          float* weights_sum, float* depth, float* image) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[3 * n], offset = (uint32_t)rays[3 * n + 1], num_steps = (uint32_t)rays[3 * n + 2];
        if (num_steps == 0 || offset + num_steps >= M) {           /* :526-533 */
            weights_sum[index] = 0; depth[index] = 0;
            image[3 * index] = image[3 * index + 1] = image[3 * index + 2] = 0;
            continue;
        }
        const float* s = sigmas + offset;
        const float* c = rgbs + 3 * (uint64_t)offset;
        const float* dl = deltas + 2 * (uint64_t)offset;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t k = 0; k < num_steps; k++) {                 /* :545-572 */
            const float alpha = 1.0f - o_expf(-s[0] * dl[0]);
            const float w = alpha * T;
            r += w * c[0]; g += w * c[1]; b += w * c[2];
            t += dl[1];
            d += w * t;
            ws += w;
            T *= 1.0f - alpha;
            if (T < 1e-4f) break;
            s++; c += 3; dl += 2;
        }
        weights_sum[index] = ws; depth[index] = d;
        image[3 * index] = r; image[3 * index + 1] = g; image[3 * index + 2] = b;
    }
}

O_API void o_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                           const float* sigmas, const float* rgbs, const float* deltas,
                                           const int32_t* rays, const float* weights_sum, const float* image,
                                           uint32_t M, uint32_t N, float* grad_sigmas, float* grad_rgbs) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[3 * n], offset = (uint32_t)rays[3 * n + 1], num_steps = (uint32_t)rays[3 * n + 2];
        if (num_steps == 0 || offset + num_steps >= M) continue;   /* :627 */
        const float gws = grad_weights_sum[index];
        const float* gi = grad_image + 3 * (uint64_t)index;
        const float rf = image[3 * index], gf = image[3 * index + 1], bf = image[3 * index + 2], wsf = weights_sum[index];
        const float* s = sigmas + offset;
        const float* c = rgbs + 3 * (uint64_t)offset;
        const float* dl = deltas + 2 * (uint64_t)offset;
        float* gs = grad_sigmas + offset;
        float* gc = grad_rgbs + 3 * (uint64_t)offset;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t k = 0; k < num_steps; k++) {                 /* :648-687 */
            const float alpha = 1.0f - o_expf(-s[0] * dl[0]);
            const float w = alpha * T;
            r += w * c[0]; g += w * c[1]; b += w * c[2];
            ws += w;
            T *= 1.0f - alpha;                                     /* T already includes (1-alpha) below */
            if (T < 1e-4f) break;                                  /* samples at/after the break keep zero grad */
            gc[0] = gi[0] * w; gc[1] = gi[1] * w; gc[2] = gi[2] * w;
            gs[0] = dl[0] * (gi[0] * (T * c[0] - (rf - r)) +
                             gi[1] * (T * c[1] - (gf - g)) +
                             gi[2] * (T * c[2] - (bf - b)) +
                             gws * (1.0f - wsf));
            s++; c += 3; dl += 2; gs++; gc += 3;
        }
        (void)ws;
    }
}

/* ------------------------------------------------------------------ */
/* inference march / composite (raymarching.cu:707-922)                */
/* ------------------------------------------------------------------ */

O_API void o_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                        const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                        float* xyzs, float* dirs, float* deltas, uint32_t perturb) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        o_march m;
        march_setup(&m, rays_o + 3 * (int64_t)index, rays_d + 3 * (int64_t)index, bound, dt_gamma, max_steps, C, H, grid);
        float* px = xyzs + 3 * (uint64_t)n * n_step;
        float* pd = dirs + 3 * (uint64_t)n * n_step;
        float* pl = deltas + 2 * (uint64_t)n * n_step;
        float t = rays_t[index];
        const float far = fars[index];
        (void)nears;
        if (perturb) {                                            /* :752-755 ; pcg32{perturb}.advance(slot n) (:819) */
            o_pcg32 g; pcg_seed(&g, (uint64_t)perturb, 1u);
            pcg_advance(&g, (uint64_t)n);
            t += m.dt_min * pcg_next_float(&g);
        }
        float last_t = t, x, y, z, dt;
        uint32_t k = 0;
        while (t < far && k < n_step) {                           /* :759-813 */
            if (march_probe(&m, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                t += dt;
                pl[0] = dt; pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2; k++;
            }
        }
    }
}

O_API void o_composite_rays(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t,
                            const float* sigmas, const float* rgbs, const float* deltas,
                            float* weights_sum, float* depth, float* image) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        const float* s = sigmas + (uint64_t)n * n_step;
        const float* c = rgbs + 3 * (uint64_t)n * n_step;
        const float* dl = deltas + 2 * (uint64_t)n * n_step;
        float t = rays_t[index];
        float ws = weights_sum[index], d = depth[index];
        float r = image[3 * (int64_t)index], g = image[3 * (int64_t)index + 1], b = image[3 * (int64_t)index + 2];
        uint32_t k = 0;
        while (k < n_step) {                                      /* :865-896 */
            if (dl[0] == 0) break;                                /* zero delta = no more samples */
            const float alpha = 1.0f - o_expf(-s[0] * dl[0]);
            const float T = 1 - ws;                               /* :877 */
            const float w = alpha * T;
            ws += w;
            t += dl[1];
            d += w * t;
            r += w * c[0]; g += w * c[1]; b += w * c[2];
            if ((double)T < 1e-4) break;                          /* :890 double literal */
            s++; c += 3; dl += 2; k++;
        }
        if (k < n_step) rays_alive[n] = -1;                       /* :902-906 */
        else rays_t[index] = t;
        weights_sum[index] = ws; depth[index] = d;
        image[3 * (int64_t)index] = r; image[3 * (int64_t)index + 1] = g; image[3 * (int64_t)index + 2] = b;
    }
}

/* ------------------------------------------------------------------ */
/* grid encoder (gridencoder/src/gridencoder.cu)                       */
/* ------------------------------------------------------------------ */

#define O_MAX_D 5
#define O_MAX_C 8

static const uint32_t O_PRIMES[7] = { 1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u };

static inline uint32_t grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners,
                                  uint32_t hashmap_size, uint32_t resolution, const uint32_t* pg) {
    /* gridencoder.cu:54-72 */
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pg[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {                  /* fast_hash :35-51 */
        uint32_t h = 0;
        for (uint32_t d = 0; d < D; d++) h ^= pg[d] * O_PRIMES[d];
        index = h;
    }
    return (index % hashmap_size) * C;
}

/* per-level scale and resolution, gridencoder.cu:125-127 */
O_API void o_grid_level_table(uint32_t L, float S, uint32_t H, float* scale, uint32_t* resolution) {
    for (uint32_t l = 0; l < L; l++) {
        scale[l] = exp2f((float)l * S) * (float)H - 1.0f;
        resolution[l] = (uint32_t)ceilf(scale[l]) + 1u;
    }
}

/* table element access in either dtype; `half` emulates scalar_t = at::Half
 * (c10::Half arithmetic: every op computes in float and rounds to half). */
static inline float tab_load(const void* tab, int is_half, uint64_t i) {
    return is_half ? h2f(((const uint16_t*)tab)[i]) : ((const float*)tab)[i];
}
static inline float rnd(int is_half, float v) { return is_half ? h2f(f2h(v)) : v; }
static inline void out_store(void* out, int is_half, uint64_t i, float v) {
    if (is_half) ((uint16_t*)out)[i] = f2h(v); else ((float*)out)[i] = v;
}

/* outputs: [L, B, C] level-major (gridencoder.cu:96); dy_dx: [B, L, D, C] */
O_API void o_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                 uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                 int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int is_half) {
    float scale[64]; uint32_t reso[64];
    o_grid_level_table(L, S, H, scale, reso);
    #pragma omp parallel for collapse(2) schedule(static)
    for (uint32_t level = 0; level < L; level++) {
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (uint64_t)b * D;
            const uint64_t gbase = (uint64_t)(uint32_t)offsets[level] * C;
            const uint64_t obase = (uint64_t)level * B * C + (uint64_t)b * C;
            const uint64_t dbase = (uint64_t)b * D * L * C + (uint64_t)level * D * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;   /* :99-105 */
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) out_store(outputs, is_half, obase + ch, 0.0f);
                if (calc_grad_inputs)
                    for (uint32_t i = 0; i < D * C; i++) out_store(dy_dx, is_half, dbase + i, 0.0f);
                continue;
            }
            const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
            const float sc = scale[level];
            const uint32_t resolution = reso[level];
            float pos[O_MAX_D]; uint32_t pg[O_MAX_D];
            for (uint32_t d = 0; d < D; d++) {                     /* :133-138 */
                pos[d] = in[d] * sc + (align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            float res[O_MAX_C] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {       /* :143-170 */
                float w = 1; uint32_t pl[O_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t gi = grid_index(D, C, gridtype, align_corners, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    /* scalar_t += float : the float product narrows to scalar_t, the sum rounds to scalar_t */
                    const float prod = rnd(is_half, w * tab_load(embeddings, is_half, gbase + gi + ch));
                    res[ch] = rnd(is_half, res[ch] + prod);
                }
            }
            for (uint32_t ch = 0; ch < C; ch++) out_store(outputs, is_half, obase + ch, res[ch]);

            if (calc_grad_inputs) {                                /* :180-223 */
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg[O_MAX_C] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = sc; uint32_t pl[O_MAX_D];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                            else { w *= pos[d]; pl[d] = pg[d] + 1; }
                        }
                        pl[gd] = pg[gd];
                        const uint32_t il = grid_index(D, C, gridtype, align_corners, hashmap_size, resolution, pl);
                        pl[gd] = pg[gd] + 1;
                        const uint32_t ir = grid_index(D, C, gridtype, align_corners, hashmap_size, resolution, pl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            const float diff = rnd(is_half, tab_load(embeddings, is_half, gbase + ir + ch) -
                                                            tab_load(embeddings, is_half, gbase + il + ch));
                            const float prod = rnd(is_half, w * diff);
                            rg[ch] = rnd(is_half, rg[ch] + prod);
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) out_store(dy_dx, is_half, dbase + gd * C + ch, rg[ch]);
                }
            }
        }
    }
}

/* grad: [L, B, C]; grad_embeddings: [sO, C] pre-zeroed; grad_inputs: [B, D].
 * The scatter order of the reference's atomics is unspecified; the oracle adds
 * in (level, b, corner) order and accumulates in double, so the HIP result is
 * compared with a tolerance (fp32) or against half-rounding bounds (fp16). */
O_API void o_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                  double* grad_embeddings_f64,
                                  uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                  int calc_grad_inputs, const void* dy_dx, void* grad_inputs,
                                  uint32_t gridtype, int align_corners, int is_half) {
    (void)embeddings;
    float scale[64]; uint32_t reso[64];
    o_grid_level_table(L, S, H, scale, reso);
    #pragma omp parallel for schedule(static)
    for (uint32_t level = 0; level < L; level++) {                 /* gridencoder.cu:227-314 ; levels own disjoint rows */
        const uint64_t gbase = (uint64_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float sc = scale[level];
        const uint32_t resolution = reso[level];
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (uint64_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            float pos[O_MAX_D]; uint32_t pg[O_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = in[d] * sc + (align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[O_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t gi = grid_index(D, C, gridtype, align_corners, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float gval = tab_load(grad, is_half, (uint64_t)level * B * C + (uint64_t)b * C + ch);
                    /* half path: (__half)(w * grad) per element (:302) ; float path: w * grad (:310) */
                    grad_embeddings_f64[gbase + gi + ch] += (double)rnd(is_half, w * gval);
                }
            }
        }
    }
    if (calc_grad_inputs) {                                        /* gridencoder.cu:317-343 */
        #pragma omp parallel for schedule(static)
        for (uint32_t t = 0; t < B * D; t++) {
            const uint32_t b = t / D, d = t - b * D;
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float g = tab_load(grad, is_half, (uint64_t)l * B * C + (uint64_t)b * C + ch);
                    const float j = tab_load(dy_dx, is_half, (uint64_t)b * L * D * C + (uint64_t)l * D * C + d * C + ch);
                    result = rnd(is_half, result + rnd(is_half, g * j));
                }
            out_store(grad_inputs, is_half, t, result);
        }
    }
}

/* ------------------------------------------------------------------ */
/* fully fused MLP, semantics of ffmlp/src/ffmlp.cu:331-407            */
/* ------------------------------------------------------------------ */

/* inputs [B, in] f16 ; weights flat f16: [hidden,in] , (num_layers-1) x [hidden,hidden] , [out_pad,hidden]
 * (ffmlp.cu:631-634) ; hidden activation ReLU, output activation none (ffmlp.py:107-108).
 * forward_buffer (optional) [num_layers, B, hidden] f16 ; outputs [B, out_pad] f16.
 * Products of two halves are exact in binary32; sums are accumulated here in
 * double and rounded once to half per layer.  (The reference accumulates in
 * half inside WMMA, the HIP kernel in binary32 inside MFMA: see DESIGN.md.) */
/* hidden activations of ffmlp/src/utils.h:423-470, applied to the HALF pre-activation (the reference's WMMA accumulator is a half fragment) */
static float o_ffmlp_act(uint32_t act, float x) {
    switch (act) {
        case 0: return x > 0 ? x : 0;                                   /* ReLU */
        case 1: return expf(x);                                          /* Exponential */
        case 2: return sinf(x);                                          /* Sine */
        case 3: return 1.0f / (1.0f + expf(-x));                         /* Sigmoid (logistic) */
        case 4: { const float y = x * 10.0f; return 0.5f * (y + sqrtf(y * y + 4.0f)) / 10.0f; }   /* Squareplus, K_ACT = 10 */
        case 5: return logf(expf(x * 10.0f) + 1.0f) / 10.0f;             /* Softplus */
        default: return x;                                               /* None */
    }
}
/* utils.h:536-590: gradient times the derivative written through the forward OUTPUT y; every product is a half product there */
static float o_ffmlp_act_backward(uint32_t act, float g, float y) {
    switch (act) {
        case 0: return y > 0 ? g : 0.0f;
        case 1: return h2f(f2h(g * y));
        case 3: return h2f(f2h(g * h2f(f2h(y * h2f(f2h(1.0f - y))))));
        case 4: { const float t = y * 10.0f; return h2f(f2h(g * h2f(f2h(t * t / (t * t + 1.0f))))); }
        case 5: return h2f(f2h(g * h2f(f2h(1.0f - expf(-y * 10.0f)))));
        default: return g;                                               /* None; Sine has no backward in the reference (utils.h:552-556) */
    }
}

O_API void o_ffmlp_forward_act(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                               uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                               uint16_t* forward_buffer, uint16_t* outputs) {
    const uint32_t n_mat = num_layers + 1;
    const uint64_t nw = (uint64_t)hidden_dim * input_dim + (uint64_t)(num_layers - 1) * hidden_dim * hidden_dim + (uint64_t)output_dim * hidden_dim;
    float* wf = (float*)malloc(sizeof(float) * nw);
    for (uint64_t i = 0; i < nw; i++) wf[i] = h2f(weights[i]);
    #pragma omp parallel
    {
        float* a = (float*)malloc(sizeof(float) * (hidden_dim > input_dim ? hidden_dim : input_dim));
        float* c = (float*)malloc(sizeof(float) * (hidden_dim > output_dim ? hidden_dim : output_dim));
        #pragma omp for schedule(static)
        for (uint32_t b = 0; b < B; b++) {
            uint32_t kdim = input_dim;
            for (uint32_t k = 0; k < input_dim; k++) a[k] = h2f(inputs[(uint64_t)b * input_dim + k]);
            const float* w = wf;
            for (uint32_t m = 0; m < n_mat; m++) {
                const uint32_t odim = (m == n_mat - 1) ? output_dim : hidden_dim;
                for (uint32_t o = 0; o < odim; o++) {
                    double acc = 0;
                    for (uint32_t k = 0; k < kdim; k++) acc += (double)(a[k] * w[(uint64_t)o * kdim + k]);
                    float v = h2f(f2h((float)acc));                /* the half pre-activation */
                    if (m != n_mat - 1) v = o_ffmlp_act(activation, v);   /* hidden layers; the output layer has no activation (ffmlp.py:108) */
                    c[o] = h2f(f2h(v));
                }
                if (m != n_mat - 1) {
                    if (forward_buffer)
                        for (uint32_t o = 0; o < odim; o++)
                            forward_buffer[(uint64_t)m * B * hidden_dim + (uint64_t)b * hidden_dim + o] = f2h(c[o]);
                    for (uint32_t o = 0; o < odim; o++) a[o] = c[o];
                } else {
                    for (uint32_t o = 0; o < odim; o++) outputs[(uint64_t)b * output_dim + o] = f2h(c[o]);
                }
                w += (uint64_t)odim * kdim;
                kdim = odim;
            }
        }
        free(a); free(c);
    }
    free(wf);
}
O_API void o_ffmlp_forward(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                           uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                           uint16_t* forward_buffer, uint16_t* outputs) {
    o_ffmlp_forward_act(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, 0u, forward_buffer, outputs);
}

/* Backward, semantics of ffmlp.cu:410-518 (activation gradients) and :749-895
 * (weight-gradient GEMMs).  Outputs in binary32 (the reference rounds them to
 * half; tests apply that rounding or a tolerance).
 *   backward_buffer[0]   = (grad . W_last)        * relu'(fwd[num_layers-1])
 *   backward_buffer[k+1] = (bwd[k] . W_hid[n-1-k]) * relu'(fwd[num_layers-2-k])   (each rounded to half)
 *   dW_last = grad^T fwd[last] ; dW_hid[i] = bwd^T fwd ; dW_in = bwd[last]^T inputs
 *   grad_inputs = bwd[last] . W_in */
O_API void o_ffmlp_backward_act(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights,
                                const uint16_t* forward_buffer, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, int calc_grad_inputs,
                                uint16_t* backward_buffer, float* grad_inputs, float* grad_weights) {
    const uint64_t W = hidden_dim;
    const uint64_t off_hidden = W * input_dim;
    const uint64_t off_last = off_hidden + (uint64_t)(num_layers - 1) * W * W;
    const uint64_t nw = off_last + (uint64_t)output_dim * W;
    float* wf = (float*)malloc(sizeof(float) * nw);
    for (uint64_t i = 0; i < nw; i++) wf[i] = h2f(weights[i]);
    double* gw = (double*)calloc(nw, sizeof(double));

    /* activation gradients, row by row */
    #pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++) {
        float cur[256], nxt[256];
        /* through the last layer */
        for (uint32_t j = 0; j < hidden_dim; j++) {
            double acc = 0;
            for (uint32_t o = 0; o < output_dim; o++)
                acc += (double)(h2f(grad[(uint64_t)b * output_dim + o]) * wf[off_last + (uint64_t)o * W + j]);
            const float f = h2f(forward_buffer[(uint64_t)(num_layers - 1) * B * W + (uint64_t)b * W + j]);
            cur[j] = h2f(f2h(o_ffmlp_act_backward(activation, h2f(f2h((float)acc)), f)));
            backward_buffer[(uint64_t)b * W + j] = f2h(cur[j]);
        }
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            const float* wm = wf + off_hidden + (uint64_t)(num_layers - 2 - k) * W * W;   /* [out,in] */
            for (uint32_t j = 0; j < hidden_dim; j++) {
                double acc = 0;
                for (uint32_t o = 0; o < hidden_dim; o++) acc += (double)(cur[o] * wm[(uint64_t)o * W + j]);
                const float f = h2f(forward_buffer[(uint64_t)(num_layers - 2 - k) * B * W + (uint64_t)b * W + j]);
                nxt[j] = h2f(f2h(o_ffmlp_act_backward(activation, h2f(f2h((float)acc)), f)));
                backward_buffer[(uint64_t)(k + 1) * B * W + (uint64_t)b * W + j] = f2h(nxt[j]);
            }
            memcpy(cur, nxt, sizeof(float) * hidden_dim);
        }
        if (calc_grad_inputs) {
            for (uint32_t i = 0; i < input_dim; i++) {
                double acc = 0;
                for (uint32_t o = 0; o < hidden_dim; o++) acc += (double)(cur[o] * wf[(uint64_t)o * input_dim + i]);
                grad_inputs[(uint64_t)b * input_dim + i] = (float)acc;
            }
        }
    }
    /* weight gradients */
    for (uint32_t b = 0; b < B; b++) {
        for (uint32_t o = 0; o < output_dim; o++) {
            const float g = h2f(grad[(uint64_t)b * output_dim + o]);
            if (g == 0) continue;
            for (uint32_t j = 0; j < hidden_dim; j++)
                gw[off_last + (uint64_t)o * W + j] += (double)g * (double)h2f(forward_buffer[(uint64_t)(num_layers - 1) * B * W + (uint64_t)b * W + j]);
        }
        for (uint32_t k = 0; k + 1 < num_layers; k++) {
            const uint32_t mi = num_layers - 2 - k;                 /* hidden matrix index */
            for (uint32_t o = 0; o < hidden_dim; o++) {
                const float g = h2f(backward_buffer[(uint64_t)k * B * W + (uint64_t)b * W + o]);
                if (g == 0) continue;
                for (uint32_t j = 0; j < hidden_dim; j++)
                    gw[off_hidden + (uint64_t)mi * W * W + (uint64_t)o * W + j] += (double)g * (double)h2f(forward_buffer[(uint64_t)mi * B * W + (uint64_t)b * W + j]);
            }
        }
        for (uint32_t o = 0; o < hidden_dim; o++) {
            const float g = h2f(backward_buffer[(uint64_t)(num_layers - 1) * B * W + (uint64_t)b * W + o]);
            if (g == 0) continue;
            for (uint32_t i = 0; i < input_dim; i++)
                gw[(uint64_t)o * input_dim + i] += (double)g * (double)h2f(inputs[(uint64_t)b * input_dim + i]);
        }
    }
    for (uint64_t i = 0; i < nw; i++) grad_weights[i] = (float)gw[i];
    free(gw); free(wf);
}
O_API void o_ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights,
                            const uint16_t* forward_buffer, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                            uint32_t hidden_dim, uint32_t num_layers, int calc_grad_inputs,
                            uint16_t* backward_buffer, float* grad_inputs, float* grad_weights) {
    o_ffmlp_backward_act(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, 0u, calc_grad_inputs,
                         backward_buffer, grad_inputs, grad_weights);
}

/* ------------------------------------------------------------------ */
/* freqencoder (freqencoder/src/freqencoder.cu:27-92)                  */
/* ------------------------------------------------------------------ */

/* Deterministic stand-in for __sinf (freqencoder.cu:57), the same operations as ngp_sinf() of the HIP library: octant
 * reduction with a three-part pi/4, degree-7 sine / degree-8 cosine polynomial on [-pi/4, pi/4]. */
static inline float o_sinf(float x) {
    float ax = fabsf(x);
    const int neg = x < 0.0f;
    float y = floorf(ax * 1.27323954473516f);
    int j = (int)y;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    int flip = neg;
    if (j > 3) { flip = !flip; j -= 4; }
    const float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    float v;
    if (j == 1 || j == 2)
        v = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    else
        v = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    return flip ? -v : v;
}

O_API void o_sinf_array(const float* x, float* y, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) y[i] = o_sinf(x[i]);
}

/* kernel_freq (freqencoder.cu:27-60): outputs [B,C], C = D + 2 D deg */
O_API void o_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    (void)deg;
    for (uint64_t t = 0; t < (uint64_t)B * C; t++) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
        const float* in = inputs + (uint64_t)b * D;
        if (c < D) { outputs[t] = in[c]; continue; }
        const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
        const float phase_shift = (float)(col % 2) * (3.141592653589793f / 2);
        outputs[t] = o_sinf(scalbnf(in[d], (int)freq) + phase_shift);
    }
}

/* kernel_freq_backward (freqencoder.cu:65-92) */
O_API void o_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                  float* grad_inputs) {
    for (uint64_t t = 0; t < (uint64_t)B * D; t++) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
        const float* g = grad + (uint64_t)b * C;
        const float* o = outputs + (uint64_t)b * C;
        float result = g[d];
        g += D; o += D;
        for (uint32_t f = 0; f < deg; f++) {
            result += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
            g += 2 * D; o += 2 * D;
        }
        grad_inputs[t] = result;
    }
}

O_API int o_abi_version(void) { return 1; }
