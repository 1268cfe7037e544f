// field_train.hip -- the training step of the field in native launches (forward keeps 64 B per sample; backward recomputes both networks).
#include <atomic>
#include "ngp_field.h"

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
// waves of a workgroup are summed through LDS and stored as that workgroup's row of a partial-sum buffer; k_field_train_wgrad_finish adds
// the rows in a fixed order and rounds to the reference's half precision: no atomics, the same bits on every run.
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

// ---------------------------------------------------------------------------------------------------------------------------------------
// The kept features: LEVEL-MAJOR, enc[level][Mp] half2 (Mp = M rounded up to 32), 64 B per sample.  Lane (g, s) of a 16-sample tile owns levels
// 4i + g (slots 2i, 2i + 1 of its B fragment): four 4-byte accesses, each a 64-byte segment per lane group -- fully used lines either way round.
// ---------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ ngp_h8 ft_enc_load(const uint32_t* __restrict__ enc, uint32_t Mp, uint32_t m, int g) {
    uint32_t v[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) v[i] = enc[(size_t)(4 * i + g) * Mp + m];
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 q = {v[0], v[1], v[2], v[3]};
    return __builtin_bit_cast(ngp_h8, q);
}
__device__ __forceinline__ void ft_enc_store(uint32_t* __restrict__ enc, uint32_t Mp, uint32_t m, int g, const ngp_h8 x) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 q = __builtin_bit_cast(u4, x);
    #pragma unroll
    for (int i = 0; i < 4; i++) enc[(size_t)(4 * i + g) * Mp + m] = q[i];
}

// Pass 1 of the two-pass forward: the hash-grid encoding LEVEL BY LEVEL (blockIdx.y = level, dispatched level-major: the whole chip works on one
// level's table -- 2 MB for a hashed level -- at a time, which an XCD's 4 MB L2 holds; the one-pass kernel has all 16 levels' 25 MB live at once and
// ran at an L2 hit rate of 0.46 on ray-ordered points, 950 B per point past L2: profiles/r15_gather_rate.md).  One lane per sample.
// Arithmetic per (sample, level): rf_encode's = k_grid_forward<half,3,2>'s (gridencoder.hip), operation for operation -- the same bits.
#ifndef FT_ENC_SPT
#define FT_ENC_SPT 2                       // samples per lane of k_ft_encode_levels (their 2 x 8 gathers are in flight together); A/B: profiles/HISTORY.md 4.3
#endif
__global__ __launch_bounds__(256) void k_ft_encode_levels(rf_params P, const float* __restrict__ xyzs, uint32_t M, uint32_t Mp, uint32_t* __restrict__ enc) {
    constexpr int SPT = FT_ENC_SPT;
    const uint32_t level = blockIdx.y;
    // the level's constants (wave-uniform): get_grid_index (gridencoder.cu:54-72) -- strides grow while they fit; a level whose last stride does not fit is hashed
    const uint32_t o0 = (uint32_t)P.offsets[level], size = (uint32_t)P.offsets[level + 1] - o0;
    const float scale = P.scale[level];
    const uint32_t side = P.resolution[level] + 1u;
    uint32_t stride = 1, s1 = 0, s2 = 0;
    #pragma unroll
    for (int d = 0; d < 3; d++)
        if (stride <= size) { if (d == 1) s1 = stride; if (d == 2) s2 = stride; stride *= side; }
    const bool dense = stride <= size;
    const bool pow2 = (size & (size - 1u)) == 0u;
    const uint32_t* tab = P.table + o0;

    uint32_t m[SPT], raw[SPT][8];
    float f[SPT][3];
    bool live[SPT];                                                        // a real sample inside the box (anything else encodes to zeros and gathers at the origin)
    #pragma unroll
    for (int q = 0; q < SPT; q++) {
        m[q] = (blockIdx.x * SPT + q) * 256 + threadIdx.x;
        const uint64_t mm = m[q] < M ? m[q] : 0;
        float x[3];
        rf_normalise(P, xyzs[3 * mm], xyzs[3 * mm + 1], xyzs[3 * mm + 2], x[0], x[1], x[2]);
        const bool oob = (x[0] < 0 || x[0] > 1) || (x[1] < 0 || x[1] > 1) || (x[2] < 0 || x[2] > 1);
        live[q] = m[q] < M && !oob;
        uint32_t pg[3];
        #pragma unroll
        for (int d = 0; d < 3; d++) {
            const float p = (live[q] ? x[d] : 0.0f) * scale + 0.5f;
            const float fl = floorf(p);
            pg[d] = (uint32_t)fl;
            f[q][d] = p - fl;
        }
        if (dense) {                                                       // x + y s1 + z s2 < size by construction
            const uint32_t i0 = pg[0] + pg[1] * s1 + pg[2] * s2;
            #pragma unroll
            for (int c = 0; c < 8; c++) raw[q][c] = tab[i0 + (c & 1) + ((c & 2) ? s1 : 0u) + ((c & 4) ? s2 : 0u)];
        } else {
            constexpr uint32_t P1 = 2654435761u, P2 = 805459861u;          // fast_hash (gridencoder.cu:35-51)
            const uint32_t hy[2] = {pg[1] * P1, (pg[1] + 1u) * P1}, hz[2] = {pg[2] * P2, (pg[2] + 1u) * P2};
            #pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint32_t h = (pg[0] + (c & 1)) ^ hy[(c >> 1) & 1] ^ hz[c >> 2];
                raw[q][c] = tab[pow2 ? (h & (size - 1u)) : (h % size)];
            }
        }
    }
    // w = (wx * wy) * wz; acc = half(float(acc) + float(half(w * float(v)))) in the reference's corner order (gridencoder.cu:147-166)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    #pragma unroll
    for (int q = 0; q < SPT; q++) {
        h2 acc = {(_Float16)0.0f, (_Float16)0.0f};
        #pragma unroll
        for (int c = 0; c < 8; c++) {
            const float w = (((c & 1) ? f[q][0] : 1 - f[q][0]) * ((c & 2) ? f[q][1] : 1 - f[q][1])) * ((c & 4) ? f[q][2] : 1 - f[q][2]);
            const h2 v = __builtin_bit_cast(h2, raw[q][c]);
            const h2 prod = {ngp_f2h(w * (float)v.x), ngp_f2h(w * (float)v.y)};
            acc = acc + prod;
        }
        if (m[q] < Mp) enc[(size_t)level * Mp + m[q]] = live[q] ? __builtin_bit_cast(uint32_t, acc) : 0u;
    }
}

template <bool FIXED>
__device__ __forceinline__ void ft_train_forward_loop(const rf_params& P, const rf_iter_class cls_rt, const rf_lane_levels& lv, const ngp_h8* __restrict__ lds_w,
                                                      const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                                                      float* __restrict__ sigmas, float* __restrict__ rgbs, uint32_t* __restrict__ enc, bool encoded) {
    const rf_iter_class cls = FIXED ? rf_iter_class{1u, 12u, 2u} : cls_rt;
    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
    const uint32_t wave = (blockIdx.x * RF_BLOCK + threadIdx.x) >> 6, nwaves = gridDim.x * (RF_BLOCK / 64);
    const uint32_t npairs = (M + 31) >> 5, Mp = npairs << 5;
    // a pair's memory operands, fetched one pair ahead (pass 2 of the two-pass forward is two MFMA chains per pair behind three loads: without the
    // prefetch every pair starts with an exposed ~2 us round trip)
    struct fw_raw { float p[2][3], d[2][3]; ngp_h8 x[2]; };
    auto fetch = [&](uint32_t pr) {
        fw_raw r;
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            const uint32_t mn = pr * 32 + 16 * n + s;
            const uint64_t mm = mn < M ? mn : 0;
            #pragma unroll
            for (int k = 0; k < 3; k++) r.d[n][k] = dirs[3 * mm + k];
            if (encoded) r.x[n] = ft_enc_load(enc, Mp, mn, g);     // pass 2: k_ft_encode_levels has filled the buffer (wave-uniform choice)
            else {
                #pragma unroll
                for (int k = 0; k < 3; k++) r.p[n][k] = xyzs[3 * mm + k];
            }
        }
        return r;
    };
    fw_raw cur;
    if (wave < npairs) cur = fetch(wave);
    for (uint32_t pair = wave; pair < npairs; pair += nwaves) {
        fw_raw nxt = cur;
        if (pair + nwaves < npairs) nxt = fetch(pair + nwaves);
        ngp_h8 x[2];
        ngp_h4 shq[2];
        uint32_t m[2];
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            m[n] = pair * 32 + 16 * n + s;
            float sh[16];
            sh_eval<4>(cur.d[n][0], cur.d[n][1], cur.d[n][2], P.shn, sh);
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = sh[j];
                if (g == 1) v = sh[4 + j];
                if (g == 2) v = sh[8 + j];
                if (g == 3) v = sh[12 + j];
                shq[n][j] = ngp_f2h(v);
            }
            if (encoded) x[n] = cur.x[n];
            else {
                x[n] = rf_encode<false>(P, lv, cls, cur.p[n][0], cur.p[n][1], cur.p[n][2]);
                ft_enc_store(enc, Mp, m[n], g, x[n]);              // the buffer is padded to whole pairs of tiles
            }
        }
        cur = nxt;
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
                                                                      uint32_t* __restrict__ enc, bool encoded) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rf_smem);
    const int lane = threadIdx.x & 63, g = lane >> 4, wave = threadIdx.x >> 6;
    rv_stage_weights(P, lds_w, wave, RF_BLOCK / 64, lane);
    rf_lane_levels lv;
    rf_setup_levels(P, g, lv);
    __syncthreads();
    const rf_iter_class cls = rf_classify(lv);
    if (cls.dense == 1u && cls.select == 2u && cls.hashed == 12u) ft_train_forward_loop<true>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs, enc, encoded);
    else ft_train_forward_loop<false>(P, cls, lv, lds_w, xyzs, dirs, M, sigmas, rgbs, enc, encoded);
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
__global__ __launch_bounds__(RF_BLOCK, 1) void k_field_train_backward(rf_params P, const uint32_t* __restrict__ enc, const float* __restrict__ dirs, uint32_t M,
                                                                       const float* __restrict__ grad_sigmas, const float* __restrict__ grad_rgbs,
                                                                       ngp_h4* __restrict__ grad_outs, _Float16* __restrict__ grad_enc,
                                                                       float* __restrict__ wgrad_ws, const uint32_t* __restrict__ live_count,
                                                                       const uint32_t* __restrict__ live_list) {
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
    // Only the LIVE samples are worked on: those with a non-zero incoming gradient, compacted in order by k_ft_live_count / k_ft_live_write (below).
    // On a converged scene half of a batch's samples lie behind the compositor's early exit (tools/zero_grad_fraction.py: 0.46-0.59 of them).  A pair
    // is 32 consecutive entries of the list; grad_outs is in list order, everything else is addressed through the list.
    const uint32_t nlive = live_count[0];
    const uint32_t npairs = (nlive + 31) >> 5, Mp = ((M + 31) >> 5) << 5;
    constexpr uint32_t NONE = 0xFFFFFFFFu;                            // a pair's slot past the end of the list (>= M: nothing loaded, nothing stored)
    auto samples_of = [&](uint32_t pr, uint32_t* out) {
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            const uint32_t cp = pr * 32 + 16 * n + s;
            out[n] = (pr < npairs && cp < nlive) ? live_list[cp] : NONE;
        }
    };
    // What a pair reads from memory, fetched ONE PAIR AHEAD: the kernel runs one wave per SIMD (400+ registers), so nothing else hides the ~2 us between a
    // pair's first load and its first use -- 55 pairs per wave on an early 1.8 M-point batch.  The next pair's loads are issued before this pair's two
    // MFMA chains and have landed when the loop comes round (22 / 12 registers); its list entries were loaded one iteration earlier still.
    struct ft_raw { ngp_h8 x[2]; float d[2][3], gs[2], gc[2][3]; ngp_h4 go[2]; };
    auto fetch = [&](uint32_t pr, const uint32_t* idx) {
        ft_raw r;
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            const uint32_t mn = idx[n];
            const uint64_t mm = mn < M ? mn : 0;
            r.x[n] = ft_enc_load(enc, Mp, (uint32_t)mm, g);
            if constexpr (PART == 0) {
                #pragma unroll
                for (int k = 0; k < 3; k++) r.d[n][k] = dirs[3 * mm + k];
                const bool own = g == 0 && mn < M;             // lanes of group 0 hold rows 0..3 of the output tiles
                r.gs[n] = own ? grad_sigmas[mm] : 0.0f;
                #pragma unroll
                for (int k = 0; k < 3; k++) r.gc[n][k] = own ? grad_rgbs[3 * mm + k] : 0.0f;
            } else {
                r.go[n] = grad_outs[(size_t)(pr * 2 + n) * 64 + lane];
            }
        }
        return r;
    };
    ft_raw cur;
    uint32_t icur[2], inxt[2], inn[2];
    samples_of(wave, icur);
    samples_of(wave + nwaves, inxt);
    if (wave < npairs) cur = fetch(wave, icur);
    for (uint32_t pair = wave; pair < npairs; pair += nwaves) {
        ft_raw nxt = cur;
        if (pair + nwaves < npairs) nxt = fetch(pair + nwaves, inxt);   // (wave-uniform)
        samples_of(pair + 2 * nwaves, inn);
        ngp_h8 x[2];
        ngp_h4 shq[2];
        uint32_t m[2];
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            m[n] = icur[n];
            if constexpr (PART == 0) {
                float sh[16];
                sh_eval<4>(cur.d[n][0], cur.d[n][1], cur.d[n][2], P.shn, sh);
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = sh[j];
                    if (g == 1) v = sh[4 + j];
                    if (g == 2) v = sh[8 + j];
                    if (g == 3) v = sh[12 + j];
                    shq[n][j] = ngp_f2h(v);
                }
            }
            x[n] = cur.x[n];
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
                        const float gh = rf_h(cur.gc[n][k]);
                        gout[n][k] = ngp_f2h((gh * (1.0f - sv)) * sv);
                    }
                    // trunc_exp backward (activation.py:17-21): g * exp(clamp(x, max = 15)), float32, then autograd's cast to the half input
                    sig_grad[n] = rf_h(cur.gs[n] * ngp_expf(fminf(rf_h(A.hs[n][0]), 15.0f)));
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
                const ngp_h4 go = cur.go[n];
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
                            const uint32_t cp = pair * 32 + 16 * n + s;      // list order: the scatter reads the list too
                            h2 lo, hi;
                            lo.x = (_Float16)d[0]; lo.y = (_Float16)d[1]; hi.x = (_Float16)d[2]; hi.y = (_Float16)d[3];
                            *reinterpret_cast<h2*>(grad_enc + ((size_t)level * M + cp) * 2) = lo;
                            *reinterpret_cast<h2*>(grad_enc + ((size_t)(level + 1) * M + cp) * 2) = hi;
                        }
                    }
                }
            }
        }
        cur = nxt;
        icur[0] = inxt[0]; icur[1] = inxt[1];
        inxt[0] = inn[0]; inxt[1] = inn[1];
    }

    // ---- the workgroup's weight gradients: sum the four waves through LDS (fixed order), then this workgroup's row of the partial-sum buffer: plain
    // stores, every element.  k_field_train_wgrad_finish adds the rows in a fixed order, so the gradients are bitwise reproducible (float atomics into
    // one shared accumulator summed in arrival order: the fitted model's samples per ray moved by +-5 % from run to run) ----
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
        wgrad_ws[(size_t)blockIdx.x * FT_WS_FLOATS + base + (16 * to + 4 * gg + r) * ld + 16 * ti + ss] = lds_acc[e];
    }
}

// partial sums [nblocks][FT_WS_FLOATS] -> gradients of FFMLP.weights: the rows added in a FIXED order (16 chunks of consecutive rows, each summed front to
// back by one thread, then the 16 chunk sums front to back), rounded to half as the reference's grad_weights are, returned as float32.
static constexpr uint32_t FT_FIN_COLS = 64, FT_FIN_CHUNKS = 16;
__global__ __launch_bounds__(FT_FIN_COLS * FT_FIN_CHUNKS) void k_field_train_wgrad_finish(const float* __restrict__ ws, uint32_t nblocks, float* __restrict__ grad_sigma_w,
                                                                                           float* __restrict__ grad_color_w) {
    __shared__ float part[FT_FIN_CHUNKS][FT_FIN_COLS];
    const uint32_t col = threadIdx.x % FT_FIN_COLS, chunk = threadIdx.x / FT_FIN_COLS;
    const uint32_t e = blockIdx.x * FT_FIN_COLS + col;                     // FT_WS_FLOATS is a multiple of 64
    const uint32_t per = (nblocks + FT_FIN_CHUNKS - 1) / FT_FIN_CHUNKS;
    const uint32_t b0 = chunk * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    float sum = 0.0f;
    for (uint32_t b = b0; b < b1; b++) sum += ws[(size_t)b * FT_WS_FLOATS + e];
    part[chunk][col] = sum;
    __syncthreads();
    if (chunk == 0) {
        float v = part[0][col];
        #pragma unroll
        for (uint32_t c = 1; c < FT_FIN_CHUNKS; c++) v += part[c][col];
        v = rf_h(v);
        if (e < 7168) grad_sigma_w[e] = v; else grad_color_w[e - 7168] = v;
    }
}

extern "C" size_t ngp_field_train_saved_bytes(uint32_t M) { return (size_t)((M + 31) >> 5) * 2 * 64 * sizeof(ngp_h8); }
// workgroups of the backward launches for M samples (persistent: at most one per CU), = rows of the partial-sum buffer
static uint32_t ft_bwd_blocks(uint32_t M) {
    const uint32_t blocks = ngp_div_up((M + 31) >> 5, RF_BLOCK / 64);
    return blocks > 256 ? 256 : blocks;
}
// ---------------------------------------------------------------------------------------------------------------------------------------
// The live samples of a backward: those whose incoming gradients (d loss / d sigma, d loss / d rgb) are not all +0, listed in order.  Everything a
// dead sample would contribute is +0 bit for bit (products with +0, matrix-core sums of them on +0 accumulators, ReLU masks, half roundings; a float
// accumulator is never -0, so x + (+-0) = x), so the backward kernels work on the list alone, and d loss / d(encoded features) is written in LIST
// order: row i of every level belongs to sample list[i]; the table scatter takes the list along (ngp_grid_scatter_binned_listed) and never sees a
// dead sample either.  (Only an inf / NaN activation of a dead sample would have poisoned the weight gradients through 0 * inf.)
//   k_ft_live_count  one thread per sample: the wave's ballot goes to a bitmap, the workgroup's count to counts[chunk]
//   k_ft_live_write  workgroup c adds up counts[0 .. c) itself (at most 4,096 values) and writes its samples' indices behind them: an ordered
//                    compaction in two launches with nothing to clear; the last workgroup leaves the total in count[0]
// ---------------------------------------------------------------------------------------------------------------------------------------
static constexpr uint32_t FT_LIVE_CHUNK = 1024;
__global__ __launch_bounds__(FT_LIVE_CHUNK) void k_ft_live_count(const float* __restrict__ grad_sigmas, const float* __restrict__ grad_rgbs, uint32_t M,
                                                               uint32_t* __restrict__ counts, unsigned long long* __restrict__ bitmap, uint32_t all_live) {
    __shared__ uint32_t s_cnt[FT_LIVE_CHUNK / 64];
    const uint32_t tid = threadIdx.x, m = blockIdx.x * FT_LIVE_CHUNK + tid, wave = tid >> 6;
    uint32_t bits = 0u;
    if (m < M)
        bits = all_live | __builtin_bit_cast(uint32_t, grad_sigmas[m]) | __builtin_bit_cast(uint32_t, grad_rgbs[3ull * m]) |
               __builtin_bit_cast(uint32_t, grad_rgbs[3ull * m + 1]) | __builtin_bit_cast(uint32_t, grad_rgbs[3ull * m + 2]);
    const unsigned long long word = __ballot(bits != 0u);
    if ((tid & 63u) == 0u) { bitmap[(size_t)blockIdx.x * (FT_LIVE_CHUNK / 64) + wave] = word; s_cnt[wave] = (uint32_t)__popcll(word); }
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0;
        #pragma unroll
        for (uint32_t w = 0; w < FT_LIVE_CHUNK / 64; w++) c += s_cnt[w];
        counts[blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(FT_LIVE_CHUNK) void k_ft_live_write(const uint32_t* __restrict__ counts, const unsigned long long* __restrict__ bitmap, uint32_t nchunks,
                                                               uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
    __shared__ uint32_t s_part[FT_LIVE_CHUNK / 64], s_cnt[FT_LIVE_CHUNK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, chunk = blockIdx.x;
    uint32_t before = 0;
    for (uint32_t c = tid; c < chunk; c += FT_LIVE_CHUNK) before += counts[c];
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
    const unsigned long long word = bitmap[(size_t)chunk * (FT_LIVE_CHUNK / 64) + wave];
    if (lane == 0) { s_part[wave] = before; s_cnt[wave] = (uint32_t)__popcll(word); }
    __syncthreads();
    uint32_t base = 0, own = 0;
    #pragma unroll
    for (uint32_t w = 0; w < FT_LIVE_CHUNK / 64; w++) { base += s_part[w]; if (w < wave) base += s_cnt[w]; own += s_cnt[w]; }
    if ((word >> lane) & 1ull) list[base + (uint32_t)__popcll(word & ((1ull << lane) - 1ull))] = chunk * FT_LIVE_CHUNK + tid;
    if (chunk == nchunks - 1 && tid == 0) {
        uint32_t total = own;
        #pragma unroll
        for (uint32_t w = 0; w < FT_LIVE_CHUNK / 64; w++) total += s_part[w];
        count[0] = total;
    }
}

// workspace: [per-workgroup f32 weight-gradient partial sums, ft_bwd_blocks(M) x 18,432 | the density-net output gradients of the live samples |
//             live count (256 B) | per-chunk counts | bitmap | list of live samples]; nothing to clear
static size_t ft_ws_outs_offset(uint32_t M) { return ((size_t)ft_bwd_blocks(M) * FT_WS_FLOATS * sizeof(float) + 255) & ~(size_t)255; }
static size_t ft_ws_live_offset(uint32_t M) { return ft_ws_outs_offset(M) + (size_t)((M + 31) >> 5) * 2 * 64 * sizeof(ngp_h4); }
static size_t ft_live_counts_bytes(size_t nch) { return (nch * 4 + 255) & ~(size_t)255; }
static size_t ft_ws_live_bytes(uint32_t M) {
    const size_t nch = ngp_div_up(M ? M : 1u, FT_LIVE_CHUNK);
    return 256 + ft_live_counts_bytes(nch) + nch * (FT_LIVE_CHUNK / 64) * 8 + (((size_t)M + 31) & ~(size_t)31) * 4;
}
extern "C" size_t ngp_field_train_workspace(uint32_t M) { return M ? ft_ws_live_offset(M) + ft_ws_live_bytes(M) : 0; }
// where the last ngp_field_train_backward(.., workspace, .., M) left its list of live samples and their number (device pointers into that workspace)
extern "C" int ngp_field_train_live_list(void* workspace, uint32_t M, const uint32_t** list, const uint32_t** count) {
    NGP_REQUIRE(workspace && list && count && M > 0, "field_train_live_list: null pointer or empty batch");
    unsigned char* base = static_cast<unsigned char*>(workspace) + ft_ws_live_offset(M);
    const size_t nch = ngp_div_up(M, FT_LIVE_CHUNK);
    *count = reinterpret_cast<const uint32_t*>(base);
    *list = reinterpret_cast<const uint32_t*>(base + 256 + ft_live_counts_bytes(nch) + nch * (FT_LIVE_CHUNK / 64) * 8);
    return NGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Density only, for a few million INCOHERENT points: the occupancy-grid refresh (nerf/renderer.py:446-531 queries `density(xyzs)['sigma']` on
// H^3 / 2 ... H^3 random cell positions per cascade every 16 training steps).  The same two passes as the training forward: k_ft_encode_levels
// (level by level: random points are the case it was built for), then the density net alone on 16-sample tiles -- 14 MFMAs per tile instead of the
// op chain's encoder + permute copy + FFMLP launch + exp (1.1 ms per 2.1 M points).  The logits are the one-launch field's (k_field_forward_lds) bit
// for bit; sigma = exp(h0) through ngp_expf like there.
// ---------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RF_BLOCK, 4) void k_field_density_mlp(rf_params P, const uint32_t* __restrict__ enc, uint32_t M, float* __restrict__ sigmas) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    ngp_h8* lds_w = reinterpret_cast<ngp_h8*>(rf_smem);                   // the density net's 14 fragments (the first 14 of rv_stage_weights' order)
    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15, wave_in_wg = threadIdx.x >> 6;
    for (int f = wave_in_wg; f < 14; f += (int)(RF_BLOCK / 64)) {
        ngp_h8 a;
        if (f < 4) a = rf_load_a_sigma_in(P.w_sigma, f, lane);
        else if (f < 12) a = mlp_load_a_permuted(P.w_sigma + MLP_W * 32, MLP_W, (f - 4) >> 1, (f - 4) & 1, lane);
        else a = mlp_load_a_permuted(P.w_sigma + MLP_W * 32 + MLP_W * MLP_W, MLP_W, 0, f - 12, lane);
        lds_w[f * 64 + lane] = a;
    }
    __syncthreads();
    const ngp_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const uint32_t wave = (blockIdx.x * RF_BLOCK + threadIdx.x) >> 6, nwaves = gridDim.x * (RF_BLOCK / 64);
    const uint32_t npairs = (M + 31) >> 5, Mp = npairs << 5;
    ngp_h8 cur[2], nxt[2];
    if (wave < npairs) { cur[0] = ft_enc_load(enc, Mp, wave * 32 + s, g); cur[1] = ft_enc_load(enc, Mp, wave * 32 + 16 + s, g); }
    for (uint32_t pair = wave; pair < npairs; pair += nwaves) {
        nxt[0] = cur[0]; nxt[1] = cur[1];
        if (pair + nwaves < npairs) {                                     // the next pair's features, one pair ahead
            nxt[0] = ft_enc_load(enc, Mp, (pair + nwaves) * 32 + s, g);
            nxt[1] = ft_enc_load(enc, Mp, (pair + nwaves) * 32 + 16 + s, g);
        }
        ngp_h8 act[2][2];
        {
            ngp_f4 d[2][MLP_MT];
            #pragma unroll
            for (int t = 0; t < MLP_MT; t++) {
                const ngp_h8 w = rv_frag(lds_w, t, lane);
                #pragma unroll
                for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w, cur[n], zero);
            }
            #pragma unroll
            for (int n = 0; n < 2; n++) { act[n][0] = mlp_pack_relu(d[n][0], d[n][1]); act[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
        }
        {
            ngp_f4 d[2][MLP_MT];
            #pragma unroll
            for (int t = 0; t < MLP_MT; t++) {
                const ngp_h8 w0 = rv_frag(lds_w, 4 + 2 * t, lane), w1 = rv_frag(lds_w, 5 + 2 * t, lane);
                #pragma unroll
                for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w0, act[n][0], zero);
                #pragma unroll
                for (int n = 0; n < 2; n++) d[n][t] = ngp_mfma(w1, act[n][1], d[n][t]);
            }
            #pragma unroll
            for (int n = 0; n < 2; n++) { act[n][0] = mlp_pack_relu(d[n][0], d[n][1]); act[n][1] = mlp_pack_relu(d[n][2], d[n][3]); }
        }
        const ngp_h8 w0 = rv_frag(lds_w, 12, lane), w1 = rv_frag(lds_w, 13, lane);
        #pragma unroll
        for (int n = 0; n < 2; n++) {
            ngp_f4 h = ngp_mfma(w0, act[n][0], zero);
            h = ngp_mfma(w1, act[n][1], h);
            const uint32_t m = pair * 32 + 16 * n + s;
            if (g == 0 && m < M) sigmas[m] = P.density_scale * ngp_expf(rf_h(h[0]));     // rv_activate's density line
        }
        cur[0] = nxt[0]; cur[1] = nxt[1];
    }
}

extern "C" size_t ngp_field_density_workspace(uint32_t M) { return (size_t)RF_L * (((size_t)(M + 31) >> 5) << 5) * sizeof(uint32_t); }

extern "C" int ngp_field_density(const ngp_field_t* field_host, const float* xyzs, uint32_t M, float* sigmas, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    rf_params P;
    int rc = rf_fill_params("field_density", field_host, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && sigmas && workspace, "field_density: null pointer");
    NGP_REQUIRE(workspace_bytes >= ngp_field_density_workspace(M), "field_density: workspace too small (ngp_field_density_workspace)");
    const uint32_t npairs = (M + 31) >> 5, Mp = npairs << 5;
    hipLaunchKernelGGL(k_ft_encode_levels, dim3(ngp_div_up(Mp, 256u * FT_ENC_SPT), RF_L), dim3(256), 0, (hipStream_t)stream, P, xyzs, M, Mp, (uint32_t*)workspace);
    uint32_t blocks = ngp_div_up(npairs, RF_BLOCK / 64);
    if (blocks > 256 * 4) blocks = 256 * 4;
    hipLaunchKernelGGL(k_field_density_mlp, dim3(blocks), dim3(RF_BLOCK), 14 * 1024, (hipStream_t)stream, P, (const uint32_t*)workspace, M, sigmas);
    NGP_CHECK_LAUNCH("field_density");
    return NGP_OK;
}

// The forward in two passes (k_ft_encode_levels, then the networks) or in one (the gather fused with the networks); same values, same kept buffer.
// Process-wide switch for A/B timing and tests; batches below FT_TWO_PASS_MIN points are latency-bound either way and take the single launch.
static std::atomic<int> ft_two_pass{1};
static constexpr uint32_t FT_TWO_PASS_MIN = 65536;
extern "C" int ngp_field_train_set_two_pass(int enabled) { return ft_two_pass.exchange(enabled ? 1 : 0, std::memory_order_relaxed); }

extern "C" int ngp_field_train_forward(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                                       float* sigmas, float* rgbs, void* saved, size_t saved_bytes, void* stream) {
    rf_params P;
    int rc = rf_fill_params("field_train_forward", field_host, P);
    if (rc != NGP_OK) return rc;
    if (M == 0) return NGP_OK;
    NGP_REQUIRE(xyzs && dirs && sigmas && rgbs && saved, "field_train_forward: null pointer");
    NGP_REQUIRE(saved_bytes >= ngp_field_train_saved_bytes(M), "field_train_forward: saved buffer too small (%zu < %zu bytes)", saved_bytes,
                ngp_field_train_saved_bytes(M));
    const uint32_t npairs = (M + 31) >> 5, Mp = npairs << 5;
    uint32_t blocks = ngp_div_up(npairs, RF_BLOCK / 64);
    if (blocks > 256 * FT_FWD_WG_PER_CU) blocks = 256 * FT_FWD_WG_PER_CU;
    const bool two_pass = ft_two_pass.load(std::memory_order_relaxed) != 0 && M >= FT_TWO_PASS_MIN;
    if (two_pass) {
        hipLaunchKernelGGL(k_ft_encode_levels, dim3(ngp_div_up(Mp, 256u * FT_ENC_SPT), RF_L), dim3(256), 0, (hipStream_t)stream, P, xyzs, M, Mp, (uint32_t*)saved);
        NGP_CHECK_LAUNCH("field_train_forward (encode)");
    }
    hipLaunchKernelGGL(k_field_train_forward, dim3(blocks), dim3(RF_BLOCK), 36 * 1024, (hipStream_t)stream, P, xyzs, dirs, M, sigmas, rgbs, (uint32_t*)saved, two_pass);
    NGP_CHECK_LAUNCH("field_train_forward");
    return NGP_OK;
}

static std::atomic<unsigned> ft_big_lds_set{0};
// process-wide switch for A/B timing and tests: 0 = the backward treats every sample as live (the list is the identity)
static std::atomic<int> ft_live_only{1};
extern "C" int ngp_field_train_set_live_only(int enabled) { return ft_live_only.exchange(enabled ? 1 : 0, std::memory_order_relaxed); }

extern "C" int ngp_field_train_backward(const ngp_field_t* field_host, const void* saved, const float* dirs, uint32_t M,
                                        const float* grad_sigmas, const float* grad_rgbs, void* grad_enc,
                                        float* grad_sigma_weights, float* grad_color_weights, void* workspace, size_t workspace_bytes, int live_only,
                                        void* stream) {
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
        const uint32_t blocks = ft_bwd_blocks(M);
        unsigned char* ws = static_cast<unsigned char*>(workspace);
        ngp_h4* grad_outs = reinterpret_cast<ngp_h4*>(ws + ft_ws_outs_offset(M));
        const uint32_t nch = ngp_div_up(M, FT_LIVE_CHUNK);
        uint32_t* live_count = reinterpret_cast<uint32_t*>(ws + ft_ws_live_offset(M));
        uint32_t* live_counts = live_count + 64;
        unsigned long long* live_bitmap = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(live_counts) + ft_live_counts_bytes(nch));
        uint32_t* live_list = reinterpret_cast<uint32_t*>(live_bitmap + (size_t)nch * (FT_LIVE_CHUNK / 64));
        hipLaunchKernelGGL(k_ft_live_count, dim3(nch), dim3(FT_LIVE_CHUNK), 0, (hipStream_t)stream, grad_sigmas, grad_rgbs, M, live_counts, live_bitmap,
                           (live_only && ft_live_only.load(std::memory_order_relaxed)) ? 0u : 1u);
        hipLaunchKernelGGL(k_ft_live_write, dim3(nch), dim3(FT_LIVE_CHUNK), 0, (hipStream_t)stream, live_counts, live_bitmap, nch, live_list, live_count);
        NGP_CHECK_LAUNCH("field_train_backward (live samples)");
        hipLaunchKernelGGL(k_field_train_backward<0>, dim3(blocks), dim3(RF_BLOCK), FT_LDS, (hipStream_t)stream, P, (const uint32_t*)saved, dirs, M,
                           grad_sigmas, grad_rgbs, grad_outs, (_Float16*)grad_enc, (float*)workspace, live_count, live_list);
        NGP_CHECK_LAUNCH("field_train_backward (colour net)");
        hipLaunchKernelGGL(k_field_train_backward<1>, dim3(blocks), dim3(RF_BLOCK), FT_LDS, (hipStream_t)stream, P, (const uint32_t*)saved, dirs, M,
                           grad_sigmas, grad_rgbs, grad_outs, (_Float16*)grad_enc, (float*)workspace, live_count, live_list);
        NGP_CHECK_LAUNCH("field_train_backward (density net)");
    }
    static_assert(FT_WS_FLOATS % FT_FIN_COLS == 0, "finish: whole column groups");
    hipLaunchKernelGGL(k_field_train_wgrad_finish, dim3(FT_WS_FLOATS / FT_FIN_COLS), dim3(FT_FIN_COLS * FT_FIN_CHUNKS), 0, (hipStream_t)stream, (const float*)workspace,
                       M > 0 ? ft_bwd_blocks(M) : 0u, grad_sigma_weights, grad_color_weights);
    NGP_CHECK_LAUNCH("field_train_wgrad_finish");
    return NGP_OK;
}
