/*
 * ngp_hip.h -- C ABI of libngp_hip.so, the MI355X (gfx950) Instant-NGP rendering core.
 *
 * This is the drop-in boundary for the reference's four native extension
 * modules (_raymarching, _gridencoder, _shencoder, _ffmlp).  The reference
 * binds them with pybind11 over at::Tensor; here every entry point takes plain
 * device pointers, sizes and a HIP stream, so the same library serves the
 * Python packages under nerf-navigation_amd/ (ctypes), a C++ caller, or a
 * pybind11 shim a reference maintainer might add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t (0 = the null stream);
 *   - every function returns 0 on success, a negative NGP_E* code on bad
 *     arguments / launch failure; ngp_last_error() returns the message of the
 *     calling thread's last failure (the reference validates nothing in
 *     raymarching and never checks a launch: raymarching.cu:15-18,152);
 *   - outputs are caller-allocated, exactly as in the reference wrappers;
 *     buffers the reference requires pre-zeroed are listed per function;
 *   - no entry point allocates, frees or synchronises: all are capturable in
 *     a hipGraph.
 *   - dtype codes: NGP_F32 = 0, NGP_F16 = 1.
 *
 * Arithmetic contract (DESIGN.md "Numerics"): IEEE binary32, RNE, no FMA
 * contraction in any kernel that feeds an integer decision; the reference's
 * __expf is replaced by the deterministic ngp_expf of csrc/ngp_device.h.
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_OK 0
#define NGP_EINVAL (-1)   /* bad argument (null pointer, unsupported D/C/width, size overflow) */
#define NGP_ELAUNCH (-2)  /* hipGetLastError() reported a failure after the launch */
#define NGP_EWORKSPACE (-3) /* workspace too small */

#define NGP_F32 0
#define NGP_F16 1

int ngp_abi_version(void);
const char* ngp_last_error(void);

/* ------------------------------------------------------------------------ */
/* _raymarching  (reference: raymarching/src/raymarching.h:7-18)             */
/* ------------------------------------------------------------------------ */

/* raymarching.h:7  near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
 * rays_o, rays_d [N,3] f32; aabb [6] f32; nears, fars [N] f32. */
int ngp_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                           float min_near, float* nears, float* fars, void* stream);

/* raymarching.h:8  sph_from_ray(rays_o, rays_d, radius, N, coords) ; coords [N,2] f32 */
int ngp_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream);

/* raymarching.h:9  morton3D(coords, N, indices) ; coords [N,3] i32 -> indices [N] i32 */
int ngp_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream);

/* raymarching.h:10 morton3D_invert(indices, N, coords) */
int ngp_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream);

/* raymarching.h:11 packbits(grid, N, density_thresh, bitfield) ; grid [8N] f32 -> bitfield [N] u8 */
int ngp_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream);

/* raymarching.h:13 march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M,
 *                                   nears, fars, xyzs, dirs, deltas, rays, counter, perturb)
 * xyzs, dirs [M,3], deltas [M,2] f32 PRE-ZEROED; rays [N,3] i32; counter [2] i32 (read-modify-write).
 * Slot order is the deterministic one "ray 0, ray 1, ..." (a valid outcome of the
 * reference's atomics, raymarching.cu:409-410).
 * workspace: ngp_march_rays_train_workspace(N) bytes of scratch. */
size_t ngp_march_rays_train_workspace(uint32_t N);
/* Optional larger workspace (the above + N * max_steps floats): the count pass then records every sample's ray parameter and
 * the second pass writes the samples from them, one lane per sample, instead of marching every ray a second time. */
size_t ngp_march_rays_train_workspace_full(uint32_t N, uint32_t max_steps);
int ngp_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                         int32_t* rays, int32_t* counter, uint32_t perturb,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The same for xyzs / dirs / deltas that were NOT pre-zeroed (torch.empty): the call zeroes the slots no ray fills itself (they form one tail). */
int ngp_march_rays_train_filled(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                int32_t* rays, int32_t* counter, uint32_t perturb,
                                void* workspace, size_t workspace_bytes, void* stream);

/* Validation switch, process-wide, default 1: with dt_gamma == 0 the count pass of ngp_march_rays_train marches one WAVE per ray (64 lattice
 * points per step, csrc/raymarching.hip: k_march_train_count_wave) instead of one lane per ray (0).  1: a window's control flow is accepted by one
 * ballot when the cheap guess holds; 2: every window goes through the run-by-run replay that 1 falls back to.  Same samples, counts and order in
 * every mode; returns the previous setting. */
int ngp_march_set_wave_per_ray(int enabled);   /* also selects the wave-per-ray kernels of ngp_composite_rays_train_* */
int ngp_composite_set_scan(int enabled);       /* wave-per-ray compositors: chains as lane scans (1, default) or every lane running the recurrence (0); returns the previous setting */
/* Validation switch, process-wide, default 1: the inference march (ngp_march_rays, ngp_march_rays_fill with a workspace) stops a ray where it leaves the box of everything occupied
 * in the grid -- formed per call from the coarse map, when the cascades nest in powers of two -- instead of walking on to its far through cells that are all empty.
 * Same samples either way; returns the previous setting. */
int ngp_march_set_occupied_box(int enabled);

/* raymarching.h:14 composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image) */
int ngp_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                     uint32_t M, uint32_t N, float* weights_sum, float* depth, float* image,
                                     void* stream);

/* raymarching.h:15 composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays,
 *                                                weights_sum, image, M, N, grad_sigmas, grad_rgbs)
 * grad_sigmas [M], grad_rgbs [M,3] PRE-ZEROED. */
int ngp_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                      const float* rgbs, const float* deltas, const int32_t* rays,
                                      const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                      float* grad_sigmas, float* grad_rgbs, void* stream);

/* raymarching.h:17 march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps,
 *                             C, H, grid, nears, fars, xyzs, dirs, deltas, perturb)
 * xyzs, dirs [>= n_alive*n_step, 3], deltas [.,2] PRE-ZEROED (zero delta = terminated, raymarching.cu:867). */
int ngp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                   float* xyzs, float* dirs, float* deltas, uint32_t perturb, void* stream);

/* The same with the zero-initialisation of the wrapper (raymarching/raymarching.py:327-329) done by the kernel: xyzs, dirs [M,3],
 * deltas [M,2] need NOT be pre-zeroed; every row is written (M >= n_alive * n_step: the wrapper's padded row count).
 * workspace (optional, ngp_march_rays_workspace(C, H) bytes): the call builds a coarse occupancy map (one bit per 4^3 block) in it and the
 * march answers "block empty" from LDS instead of from the bitfield; same samples, bit for bit. */
size_t ngp_march_rays_workspace(uint32_t C, uint32_t H);
int ngp_march_rays_fill(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                        const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                        float* xyzs, float* dirs, float* deltas, uint32_t M, uint32_t perturb,
                        void* workspace, size_t workspace_bytes, void* stream);

/* raymarching.h:18 composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
 * mutates rays_alive (-1 = dead), rays_t, weights_sum, depth, image in place. */
int ngp_composite_rays(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t, const float* sigmas,
                       const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image,
                       void* stream);
/* the same with rgbs as halves (a field under autocast returns halves; the reference's wrapper widens them first, raymarching.py:343): same values */
int ngp_composite_rays_half(uint32_t n_alive, uint32_t n_step, int32_t* rays_alive, float* rays_t, const float* sigmas,
                            const void* rgbs_half, const float* deltas, float* weights_sum, float* depth, float* image, void* stream);

/* Not in the reference's native surface: stable compaction of rays_alive (the reference does it with a torch
 * boolean mask, nerf/renderer.py:365).  out [n_alive] i32, n_out [1] i32 device counter (overwritten).
 * workspace: ngp_compact_alive_workspace(n_alive) bytes. */
size_t ngp_compact_alive_workspace(uint32_t n_alive);
int ngp_compact_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same, and the count also reaches the HOST without a stream synchronisation: host_pair = 2 int32 of pinned coherent memory from ngp_host_words_alloc;
 * the kernel stores the count in host_pair[0], then `seq` in host_pair[1] (system scope, release); the caller polls host_pair[1] == seq. */
int ngp_compact_alive_publish(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out, int32_t* host_pair, int32_t seq,
                              void* workspace, size_t workspace_bytes, void* stream);
int ngp_host_words_alloc(uint32_t n_words, void** host_ptr);   /* pinned, coherent, device-visible, zeroed; 1 .. 4096 words */
int ngp_host_words_free(void* host_ptr);

/* ------------------------------------------------------------------------ */
/* density-grid maintenance.  Reference: nerf/renderer.py:381-537 -- Python loops over torch ops      */
/* (meshgrid, morton3D, rand_like, density query, indexed scatter, masked EMA, mean, packbits); no native surface.   */
/* ------------------------------------------------------------------------ */

/* Sample points of one sweep: full (iter_density < 16, :455-483) = cascade * H^3 points, point e IS cell e of the
 * [cascade][H^3] Morton-ordered grid; partial (:486-511) = per cascade H^3/4 uniformly random cells followed by H^3/4 picks
 * among the cells with density_grid > 0 (ascending list, as torch.nonzero), 2 * (H^3/4) points per cascade. */
uint32_t ngp_density_grid_points(uint32_t cascade, uint32_t H, int partial);
size_t ngp_density_grid_workspace(uint32_t cascade, uint32_t H);

/* nerf/renderer.py:473-483 / :488-507.  xyzs [n,3] f32 out (n = ngp_density_grid_points); cells [n] i32 out (partial sweep only;
 * cascade * H^3 + Morton index, -1 = no sample because the cascade has no occupied cell); density_grid [cascade*H^3] f32 is read
 * by the partial sweep only.  Random numbers: pcg32(seed, seq = iteration) (raymarching/src/pcg32.h), sample e owns draws
 * [16 e, 16 e + 16): full sweep 0..2 jitter; partial sweep (e = cas * H^3/4 + i) 0..2 random cell (next_uint * H >> 32),
 * 3 pick (next_uint * n_occ >> 32), 4..6 jitter of the random cell, 7..9 jitter of the picked cell. */
int ngp_density_grid_sample(const float* density_grid, uint32_t cascade, uint32_t H, float bound, int partial, uint64_t seed,
                            uint64_t iteration, float* xyzs, int32_t* cells, void* workspace, size_t workspace_bytes, void* stream);

/* nerf/renderer.py:511-531: tmp_grid[cells] = sigmas * density_scale (several samples of one cell: the largest);
 * where density_grid >= 0 and tmp_grid >= 0: density_grid = max(density_grid * decay, tmp_grid); mean_density[0] =
 * mean(clamp(density_grid, 0)) (summed in double in a fixed order); bitfield = packbits(density_grid, min(mean_density,
 * density_thresh)).  cells == NULL: full sweep, sigmas [cascade*H^3] already in grid order.  mean_density: [1] f32 device. */
int ngp_density_grid_update(const float* sigmas, const int32_t* cells, uint32_t n_points, float density_scale, float decay,
                            float density_thresh, uint32_t cascade, uint32_t H, float* density_grid, uint8_t* bitfield,
                            float* mean_density, void* workspace, size_t workspace_bytes, void* stream);

/* nerf/renderer.py:381-442: density_grid = -1 in every cell whose centre no camera sees (z > 0, |x| < cx/fx z + 2 half cells,
 * same for y).  poses [B,4,4] f32 row-major camera-to-world on the DEVICE. */
int ngp_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t cascade, uint32_t H,
                            float bound, float* density_grid, void* stream);

/* ------------------------------------------------------------------------ */
/* _gridencoder  (reference: gridencoder/src/gridencoder.h:12-13)            */
/* ------------------------------------------------------------------------ */

/* gridencoder.h:12 grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H,
 *                                      calc_grad_inputs, dy_dx, gridtype, align_corners)
 * inputs [B,D] f32 in [0,1]; embeddings [sO,C] `dtype`; offsets [L+1] i32 (device);
 * outputs [L,B,C] `dtype` (level-major, as the reference); dy_dx [B, L*D*C] `dtype` (may be null if !calc).
 * D in {2,3,4,5}, C in {1,2,4,8}, L <= 32; gridtype 0 = hash, 1 = tiled. */
int ngp_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners,
                            int dtype, void* stream);

/* The same forward (calc_grad_inputs = 0) writing outputs [B, L*C] directly -- what GridEncoder.forward returns after the reference's
 * permute + reshape copy (gridencoder/grid.py:42,52).  Same values, bit for bit; D = 3 and C = 2 only. */
int ngp_grid_encode_forward_rows(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                 uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                 uint32_t gridtype, int align_corners, int dtype, void* stream);

/* gridencoder.h:13 grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
 *                                       calc_grad_inputs, dy_dx, grad_inputs, gridtype, align_corners)
 * grad [L,B,C]; grad_embeddings [sO,C] PRE-ZEROED, or NULL when only grad_inputs is wanted (frozen model: skips the
 * scatter); grad_inputs [B,D] (all `dtype`). */
int ngp_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                             void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                             int align_corners, int dtype, void* stream);
/* The input gradient of grid_encode_backward WITHOUT the dy_dx tensor: recomputes the Jacobian from the table
 * (gridencoder.cu:180-223 arithmetic) and contracts it with grad in kernel_input_backward's order (gridencoder.cu:317-343):
 * bit-identical to forward(calc_grad_inputs=1) + backward(calc_grad_inputs=1), without writing and re-reading
 * B*L*D*C values.  grad [L,B,C] and grad_inputs [B,D] have the table's dtype; D in {2,3}, C in {1,2,4,8}. */
int ngp_grid_encode_backward_inputs(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                    void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, void* stream);

/* The table gradient of grid_encode_backward for the reference's hash grid under autocast (D = 3, C = 2, half `grad` [L,B,2]) WITHOUT
 * global atomics (gridencoder.cu:227-314 scatters with atomicAdd(__half2)): contributions are binned by table slice and summed in LDS, exactly, as
 * 64-bit fixed point (csrc/gridencoder.hip "Binned scatter").  grad_embeddings [sO,2] is WRITTEN WHOLE (no pre-zeroing), as float32 or half
 * (`out_dtype`), multiplied by out_scale.  max_level_rows: an upper bound of the rows of any level (2^log2_hashmap_size), at most 2^19.
 * workspace: ngp_grid_scatter_binned_workspace(B, L) bytes (768 B per sample, at most 3.2 GB: from 2^22 samples on the call runs in passes). */
size_t ngp_grid_scatter_binned_workspace(uint32_t B, uint32_t L);
int ngp_grid_scatter_binned(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                            uint32_t B, uint32_t L, float S, uint32_t H, uint32_t max_level_rows, uint32_t gridtype, int align_corners,
                            int out_dtype, float out_scale, void* workspace, size_t workspace_bytes, void* stream);
/* The same for a LISTED batch: gradient row i (of every level) belongs to the sample at inputs[list[i]], i < *list_count <= B; list and list_count are
 * device memory read by the kernels (no synchronisation).  B <= 2^22. */
int ngp_grid_scatter_binned_listed(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                   uint32_t B, uint32_t L, float S, uint32_t H, uint32_t max_level_rows, uint32_t gridtype, int align_corners,
                                   int out_dtype, float out_scale, const uint32_t* list, const uint32_t* list_count,
                                   void* workspace, size_t workspace_bytes, void* stream);
/* The same in two steps: phase 1 bins all levels (grad_embeddings unused), phase 2 sums levels [level_lo, level_hi) into their rows of grad_embeddings;
 * any number of phase-2 calls after one phase-1 call (same workspace, same stream).  Lets the data-parallel gradient exchange all-reduce one group of
 * levels while the next is summed.  B <= 2^22. */
int ngp_grid_scatter_binned_phase(int phase, const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                  uint32_t B, uint32_t L, uint32_t level_lo, uint32_t level_hi, float S, uint32_t H, uint32_t max_level_rows,
                                  uint32_t gridtype, int align_corners, int out_dtype, float out_scale, void* workspace, size_t workspace_bytes,
                                  void* stream);
int ngp_grid_scatter_binned_phase_listed(int phase, const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                         uint32_t B, uint32_t L, uint32_t level_lo, uint32_t level_hi, float S, uint32_t H, uint32_t max_level_rows,
                                         uint32_t gridtype, int align_corners, int out_dtype, float out_scale, const uint32_t* list,
                                         const uint32_t* list_count, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------ */
/* optimiser step of the training loop                                       */
/* (reference: main_nerf.py:126 torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15); nerf/utils.py:329 GradScaler,       */
/*  :789-791 scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update() -- torch library code there, three native launches here) */
/* ------------------------------------------------------------------------ */

#define NGP_ADAM_MAX_TENSORS 16
/* words of the 32-word device state buffer a caller may read or initialise (float32 unless noted) */
#define NGP_ADAM_STATE_WORDS 32
#define NGP_ADAM_STATE_SCALE 0           /* GradScaler's scale; initialise (torch: 65536) */
#define NGP_ADAM_STATE_GROWTH_TRACKER 1  /* int32: unskipped steps since the scale last changed */
#define NGP_ADAM_STATE_STEP 2            /* int32: Adam's step count t (skipped steps do not count) */
#define NGP_ADAM_STATE_FOUND_INF 4       /* 1.0 if the last call found a non-finite gradient and skipped the update, else 0.0 */

typedef struct {
    void* param;          /* float32 [n], updated in place */
    const void* grad;     /* float32 [n], still multiplied by the loss scale; read only */
    void* exp_avg;        /* float32 [n] */
    void* exp_avg_sq;     /* float32 [n] */
    void* half_copy;      /* float16 [n] or NULL: receives the updated parameters rounded to half (what an autocast forward reads) */
    uint64_t n;
    double lr;            /* this tensor's learning rate for this step (param_group['lr'] after the scheduler) */
} ngp_adam_tensor_t;

typedef struct {
    double beta1, beta2, eps;
    double growth_factor, backoff_factor;   /* GradScaler: 2.0, 0.5 */
    int32_t growth_interval;                /* GradScaler: 2000 */
    int32_t scaler_enabled;                 /* 0: no loss scaling -- gradients are used as they are, never checked, the scale words are not touched */
} ngp_adam_hyper_t;

/* One optimiser step for `count` <= NGP_ADAM_MAX_TENSORS tensors: non-finite check of all gradients, GradScaler's skip decision and scale update,
 * Adam (weight_decay 0, amsgrad off) with the arithmetic of torch.optim.Adam (csrc/adam.hip spells it out).  `tensors` and `hyper` are host memory, read
 * before the call returns; `state` is device memory (NGP_ADAM_STATE_WORDS words, zero-filled + the scale by the caller before the first step), and nothing
 * is read back: the host never waits. */
int ngp_adam_step(const ngp_adam_tensor_t* tensors, uint32_t count, const ngp_adam_hyper_t* hyper, float* state, void* stream);

/* Between the compositor and the loss of a training step (csrc/train_head.hip; torch elementwise ops in the reference, ~20 launches under autograd).
 * mix: nerf/renderer.py:318-319  out_image = image + (1 - weights_sum)[:, None] * bg_color;  out_depth = clamp(depth - nears, min=0) / (fars - nears)
 * (out_depth may be NULL).  bg_rows: 0 = bg_value for every channel, 1 = bg[3], N = bg[N,3].  backward: grad_weights_sum = -(grad_out_image . bg); the
 * gradient of `image` is grad_out_image itself; the compositor ignores the gradient of depth (raymarching.py:270), so none is formed. */
int ngp_train_mix_forward(const float* weights_sum, const float* depth, const float* image, const float* nears, const float* fars,
                          const float* bg, uint32_t bg_rows, float bg_value, uint32_t N, float* out_image, float* out_depth, void* stream);
int ngp_train_mix_backward(const float* grad_out_image, const float* bg, uint32_t bg_rows, float bg_value, uint32_t N, float* grad_weights_sum, void* stream);
/* mse head: nerf/utils.py:450,480 loss = mean((pred - target)^2), :789 scaler.scale(loss).  loss[0] = the mean, loss[1] = the mean * scale[0] (scale may be
 * NULL: loss[1] = loss[0]); grad_unit [numel] = 2 / numel * (pred - target), the gradient for a unit incoming one.  workspace: ngp_mse_head_workspace() bytes,
 * zero-filled once by the caller, one per stream.  backward: grad_pred = grad_unit * (grad_loss[0] + grad_scaled[0] * scale[0]); either may be NULL. */
/* mix forward + mse head forward + mse head backward (incoming gradient 1 on the scaled loss) + mix backward in ONE launch, for a caller that runs the backward
 * right behind the forward: the same values, bit for bit, as the four calls.  It also clears up to three caller buffers (16-byte aligned, multiples of 16 bytes)
 * that the backward launches after it expect zeroed.  workspace: the mse head's. */
int ngp_train_head_direct(const float* weights_sum, const float* image, const float* bg, uint32_t bg_rows, float bg_value, const float* target,
                          const float* scale, uint32_t N, float* out_image, float* loss, float* grad_image, float* grad_weights_sum,
                          void* const* zero_ptrs, const uint64_t* zero_bytes, uint32_t zero_count, void* workspace, size_t workspace_bytes, void* stream);
size_t ngp_mse_head_workspace(void);
int ngp_mse_head_forward(const float* pred, const float* target, uint32_t numel, const float* scale, float* loss, float* grad_unit,
                         void* workspace, size_t workspace_bytes, void* stream);
int ngp_mse_head_backward(const float* grad_unit, const float* grad_loss, const float* grad_scaled, const float* scale, uint32_t numel,
                          float* grad_pred, void* stream);

/* ------------------------------------------------------------------------ */
/* _shencoder  (reference: shencoder/src/shencoder.h:10,13)                  */
/* ------------------------------------------------------------------------ */

/* shencoder.h:10 sh_encode_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx)
 * inputs [B,3] f32; outputs [B,C*C] f32; dy_dx [B,3*C*C] f32 (may be null if !calc); C = degree in 1..8. */
int ngp_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                          int calc_grad_inputs, float* dy_dx, void* stream);

/* shencoder.h:13 sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs)
 * grad_inputs [B,3] is accumulated into (+=), so PRE-ZEROED by the caller (shencoder.cu:379). */
int ngp_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                           const float* dy_dx, float* grad_inputs, void* stream);

/* ---- freqencoder (optional module: encoding.get_encoder('frequency'), encoding.py:56-58) ----------------------------- */
/* freqencoder.h:7 freq_encode_forward(inputs, B, D, deg, C, outputs): inputs [B,D] f32 -> outputs [B,C] f32,
 * C = D + 2*D*deg: [x | sin(2^f x), sin(2^f x + pi/2) for f < deg]. */
int ngp_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs, void* stream);
/* freqencoder.h:10 freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs): grad, outputs [B,C] -> grad_inputs [B,D]
 * (fully written). */
int ngp_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                             float* grad_inputs, void* stream);

/* ------------------------------------------------------------------------ */
/* _ffmlp  (reference: ffmlp/src/ffmlp.h:8-14)                                */
/* ------------------------------------------------------------------------ */

/* ffmlp.h:8 ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
 *                         output_activation, forward_buffer, outputs)
 * all tensors f16. inputs [B,input_dim]; weights flat [hidden,in] + (num_layers-1)*[hidden,hidden] + [output_dim,hidden];
 * forward_buffer [num_layers,B,hidden]; outputs [B,output_dim]; output_dim == 16 (padded), hidden_dim == 64,
 * input_dim in {16,32,48,64}; B % 16 == 0 (the Python wrapper pads to 128 like the reference).
 * activation: 0 = ReLU (the only hidden activation reachable from the reference, ffmlp/ffmlp.py:107);
 * output_activation: 6 = none. */
int ngp_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                      uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                      void* forward_buffer, void* outputs, void* stream);

/* ffmlp.h:9 ffmlp_inference(..., inference_buffer, outputs) ; inference_buffer is unused scratch (may be null) */
int ngp_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                        uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                        void* inference_buffer, void* outputs, void* stream);

/* ffmlp.h:11 ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers,
 *                           activation, output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights)
 * grad [B,output_dim] f16; backward_buffer [num_layers,B,hidden] f16; grad_inputs [B,input_dim] f16;
 * grad_weights flat f16 (same layout as weights).
 * workspace: ngp_ffmlp_backward_workspace(...) bytes (f32 weight-gradient accumulators; zeroed by the call). */
size_t ngp_ffmlp_backward_workspace(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers);
int ngp_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer,
                       uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                       uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,
                       void* grad_inputs, void* grad_weights, void* workspace, size_t workspace_bytes, void* stream);

/* ffmlp.h:13-14 allocate_splitk(size) / free_splitk(): the reference creates side streams for CUTLASS split-K
 * GEMMs.  This implementation reduces weight gradients inside one kernel, so both are accepted no-ops. */
int ngp_allocate_splitk(size_t size);
int ngp_free_splitk(void);

/* ------------------------------------------------------------------------ */
/* Fused inference path (extra, opt-in; used by the package's own renderer)   */
/* ------------------------------------------------------------------------ */

/* Description of the field evaluated by the fused kernels: the network_ff model of the reference
 * (nerf/network_ff.py:11-148): hash grid (D=3, C=2, L levels, f16 table) -> FFMLP(32->64->64->16) ->
 * sigma = exp(h0), geo = h[1:16]; SH degree 4 (16) ++ geo (15) ++ 0 -> FFMLP(32->64->64->64->16) -> sigmoid. */
typedef struct {
    const void* embeddings;      /* [sO,2] f16 */
    const int32_t* offsets;      /* [L+1] i32, device */
    const void* sigma_weights;   /* flat f16, 64*(32+64+16)   = 7168  */
    const void* color_weights;   /* flat f16, 64*(32+128+16)  = 11264 */
    uint32_t L;                  /* 16 */
    uint32_t H;                  /* base resolution (16) */
    float S;                     /* log2(per_level_scale) */
    float bound;                 /* world -> [0,1]: (x + bound) / (2 bound) */
    float density_scale;         /* sigma multiplier (nerf/renderer.py:361) */
} ngp_field_t;

/* sigma/rgb for explicit points: the fused equivalent of NeRFNetwork.forward (nerf/network_ff.py:51-77) under
 * autocast.  xyzs, dirs [M,3] f32; sigmas [M] f32 (already times density_scale); rgbs [M,3] f32. */
int ngp_field_forward(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                      float* sigmas, float* rgbs, void* stream);
/* the same, rgbs [M,3] written as f16: the dtype the reference's network_ff returns under autocast (the values are halves either way) */
int ngp_field_forward_half(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                           float* sigmas, void* rgbs_half, void* stream);

/* The field's training step in two calls: NeRFNetwork.forward (nerf/network_ff.py:51-77) under autocast and the backward autograd
 * runs through it (ffmlp/src/ffmlp.cu:410-518,749-895; activation.py:17-21; torch.sigmoid), for the default field shapes.
 * forward: as ngp_field_forward, and keeps the encoded features of every sample (64 B each, level-major [16][M rounded up to 32] half2) in
 *   `saved` (ngp_field_train_saved_bytes(M) bytes) -- the only activation kept; both networks are recomputed in the backward.
 * backward: grad_sigmas [M], grad_rgbs [M,3] f32 (gradients of the forward's outputs) ->
 *   grad_enc [16][M][2] f16, level-major: d(loss)/d(encoded features), the `grad` argument of ngp_grid_encode_backward / ngp_grid_scatter_binned;
 *   grad_sigma_weights [7168], grad_color_weights [11264] f32 in FFMLP's weight layout, rounded to half like the reference's grad_weights.
 *   live_only != 0: the kernels work on the samples whose incoming gradients are not all +0 (bit pattern) and nothing else -- behind the compositor's
 *   early exit half of a converged batch gets none, and everything such a sample would contribute is +0 bit for bit.  grad_enc is then written in
 *   LIST order: row i of every level belongs to sample list[i], i < *count (rows from *count on are not written); ngp_field_train_live_list returns
 *   the two device pointers (into `workspace`), ngp_grid_scatter_binned_listed takes them.  live_only == 0: every sample, rows in sample order.
 *   field_host must describe the same half copies the forward used.  workspace: ngp_field_train_workspace(M) bytes, contents arbitrary
 *   (per-workgroup partial sums of the weight gradients, added in a fixed order: the gradients are bitwise reproducible; the live list; nothing to
 *   clear, ngp_field_train_workspace(0) == 0). */
/* Density only (sigma = exp(h0) * field.density_scale) for M points in two launches: the level-by-level encoder of the training forward, then the
 * density net on the matrix cores -- what the occupancy-grid refresh asks of the field (nerf/renderer.py:478-486,511-517 `self.density(xyzs)['sigma']`
 * on millions of random cell positions).  Same logits as ngp_field_forward.  workspace: ngp_field_density_workspace(M) bytes. */
size_t ngp_field_density_workspace(uint32_t M);
int ngp_field_density(const ngp_field_t* field_host, const float* xyzs, uint32_t M, float* sigmas, void* workspace, size_t workspace_bytes, void* stream);
size_t ngp_field_train_saved_bytes(uint32_t M);
/* forward in two passes (default: the encoder level by level, one level's table live in L2 at a time, then the networks) or in one launch;
 * same values and the same `saved` layout either way; returns the previous setting (process-wide: A/B timing, tests). */
int ngp_field_train_set_two_pass(int enabled);
int ngp_field_train_set_live_only(int enabled);   /* 0: ngp_field_train_backward(live_only = 1) lists every sample (A/B timing, tests); returns the previous setting */
size_t ngp_field_train_workspace(uint32_t M);
int ngp_field_train_forward(const ngp_field_t* field_host, const float* xyzs, const float* dirs, uint32_t M,
                            float* sigmas, float* rgbs, void* saved, size_t saved_bytes, void* stream);
int ngp_field_train_backward(const ngp_field_t* field_host, const void* saved, const float* dirs, uint32_t M,
                             const float* grad_sigmas, const float* grad_rgbs, void* grad_enc,
                             float* grad_sigma_weights, float* grad_color_weights, void* workspace, size_t workspace_bytes, int live_only,
                             void* stream);
int ngp_field_train_live_list(void* workspace, uint32_t M, const uint32_t** list, const uint32_t** count);

/* One whole frame of NeRFRenderer.run_cuda's inference branch (nerf/renderer.py:325-374) in one launch:
 * near/far, occupancy march, field evaluation and compositing per ray, with no intermediate tensors.
 * image [N,3], depth [N], weights_sum [N] f32 are fully written (no pre-zeroing needed);
 * image already includes the background mix image + (1-ws)*bg_color and depth the (depth-near)/(far-near) map.
 * stats [4] u32 device (zeroed by the call): [0] ray-samples evaluated, [1] rays that consumed > max_steps samples
 * (schedule-dependent in the reference), [2] rays with at least one sample, [3] 16-column matrix-core tiles evaluated (ray-samples / (16 * tiles) is the
 * packing efficiency; unlike [0..2] it depends on the traversal order).
 * image_width: when rays_o/rays_d are a row-major image (width and height multiples of 8) pass its width and the
 * kernel walks the rays in 8x8 pixel tiles for cache locality; 0 = rays in no particular order.  Results do not
 * depend on it.
 * workspace: ngp_render_frame_workspace(N) bytes. */
size_t ngp_render_frame_workspace(uint32_t N);
/* Validation switch, process-wide, default 1: ngp_render_frame jumps through empty 4^3 / 16^3 blocks of the occupancy
 * grid when that provably visits the reference's samples (render_fused.hip, rv_probe).  0 = march cell by cell like
 * kernel_march_rays (raymarching.cu:748-801).  Results are identical either way; returns the previous setting. */
int ngp_render_set_block_skip(int enabled);
/* Validation switch, process-wide, default 1: a ray of ngp_render_frame marches no further than where it leaves the box of everything occupied in the
 * grid (beyond it every cell is empty: the reference tests those cells and finds nothing).  0 = to its own far.  Results are identical; returns the previous setting. */
int ngp_render_set_occupied_box(int enabled);
int ngp_render_frame(const ngp_field_t* field_host, const float* rays_o, const float* rays_d, uint32_t N,
                     uint32_t image_width, const float* aabb, float min_near, const uint8_t* bitfield, uint32_t C, uint32_t Hgrid,
                     float dt_gamma, uint32_t max_steps, const float* bg_color3_host,
                     float* image, float* depth, float* weights_sum, uint32_t* stats,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------ */
/* Navigation-loop queries (BASELINE config 4), fused float32: the DEFAULT field of nerf/network.py                   */
/* (hash grid 16 x 2 f32 -> Linear(32,64) -> Linear(64,16) ; SH16 ++ geo15 -> Linear(31,64) -> Linear(64,64) ->      */
/* Linear(64,3), bias-free) as the planner and the pose filter call it: simulate.py:340-347.                          */
/* ------------------------------------------------------------------------ */
typedef struct {
    const float* embeddings;     /* [sO,2] f32 (GridEncoder.embeddings) */
    const int32_t* offsets_host; /* [L+1] i32 on the HOST */
    const float* sigma_w0;       /* sigma_net.0.weight [64,32] */
    const float* sigma_w1;       /* sigma_net.1.weight [16,64] */
    const float* color_w0;       /* color_net.0.weight [64,31] */
    const float* color_w1;       /* color_net.1.weight [64,64] */
    const float* color_w2;       /* color_net.2.weight [3,64] */
    uint32_t L;                  /* 16 */
    uint32_t H;                  /* base resolution */
    float S;                     /* log2(per_level_scale) */
    float bound;
    float density_scale;         /* NeRFRenderer.density_scale (nerf/renderer.py:208) */
} ngp_nav_field_t;

/* Transposed copies of sigma_w0 and color_w1 the kernels read (ngp_nav_field_workspace() bytes, caller-owned, device); call again
 * whenever those weights change.  Every entry point below takes the same `prepared` pointer. */
size_t ngp_nav_field_workspace(void);
int ngp_nav_field_prepare(const ngp_nav_field_t* field_host, void* workspace, size_t workspace_bytes, void* stream);

/* NeRFNetwork.density (nerf/network.py:125-143) on explicit points: sigma [M] = exp(h0) (trunc_exp), geo [M,15] = h[1:16] (may be NULL). */
int ngp_nav_density_forward(const ngp_nav_field_t* field_host, const void* prepared, const float* xyz, uint32_t M, float* sigma, float* geo,
                            void* stream);
/* its backward to the points: grad_sigma [M], grad_geo [M,15] or NULL -> grad_xyz [M,3] (fully written; trunc_exp's clamped backward,
 * activation.py:16-18; zero outside [-bound, bound]^3 like the encoder's dy_dx). */
int ngp_nav_density_backward(const ngp_nav_field_t* field_host, const void* prepared, const float* xyz, uint32_t M, const float* grad_sigma,
                             const float* grad_geo, float* grad_xyz, void* stream);
/* The planner's query in ONE launch (nav/quad_plot.py:224-250 over simulate.py:340-343): sigma [M] and jac [M,3] = d sigma / d xyz (trunc_exp's backward
 * factor included: grad_xyz = grad_sigma * jac), a point's 16 levels split over the four waves of a workgroup -- built for a few thousand points, where
 * the one-lane-per-point kernels above are latency-bound.  rot9_host: NULL, or a row-major 3x3 matrix applied as xyz @ rot before the field
 * (simulate.py:340's axis change), its transpose applied to the Jacobian. */
int ngp_nav_density_value_jac(const ngp_nav_field_t* field_host, const void* prepared, const float* xyz, uint32_t M, const float* rot9_host,
                              float* sigma, float* jac, void* stream);

/* NeRFRenderer.run (nerf/renderer.py:125-254) with upsample_steps = 0 and perturb = False, one workgroup per ray: nears, fars [N] from
 * ngp_near_far_from_aabb; aabb [6] and bg_color [3] on the host; image [N,3] (background mixed in), depth [N], weights_sum [N].
 * saved: NULL, or ngp_nav_run_saved_bytes(N, num_steps) bytes (108 per sample) that the forward fills for ngp_nav_run_backward. */
size_t ngp_nav_run_saved_bytes(uint32_t N, uint32_t num_steps);
int ngp_nav_run_forward(const ngp_nav_field_t* field_host, const void* prepared, const float* rays_o, const float* rays_d, const float* nears,
                        const float* fars, uint32_t N, uint32_t num_steps, const float* aabb_host, const float* bg_color3_host,
                        float* image, float* depth, float* weights_sum, void* saved, size_t saved_bytes, void* stream);
/* its backward to the rays (what the pose filter differentiates, nav/estimator_helpers.py:316): grad_image [N,3], grad_depth [N] or NULL,
 * grad_weights_sum [N] or NULL, the forward's `saved` buffer -> grad_rays_o, grad_rays_d [N,3] (fully written). */
int ngp_nav_run_backward(const ngp_nav_field_t* field_host, const void* prepared, const float* rays_o, const float* rays_d, const float* nears,
                         const float* fars, uint32_t N, uint32_t num_steps, const float* aabb_host, const float* bg_color3_host,
                         const float* grad_image, const float* grad_depth, const float* grad_weights_sum, const void* saved, size_t saved_bytes,
                         float* grad_rays_o, float* grad_rays_d, void* stream);

/* ---- camera rays (reference: get_rays, nerf/utils.py:53-116) ----
 * pose: host, row-major 4x4 (or the first 3 rows of it) camera-to-world; intrinsics: host [4] = fx, fy, cx, cy.
 * Ray k is that of pixel inds[k] (device int64, row-major pixel index, the `inds` of the reference's random branches) or
 * of pixel k when inds is null (N == H*W, the full-image branch).  rays_o, rays_d [N,3] f32 device.
 * Arithmetic and its order: csrc/ngp_camera.h (equal to the torch formula to ~1 ulp; bit-exact against the oracle). */
int ngp_get_rays(const float* pose_host, const float* intrinsics_host, uint32_t H, uint32_t W,
                 const int64_t* inds, uint32_t N, float* rays_o, float* rays_d, void* stream);
/* ngp_render_frame with get_rays fused into the kernel: the H*W rays of the camera, never materialised.
 * Bit-identical to ngp_get_rays(..., NULL, H*W, ...) followed by ngp_render_frame(..., N = H*W, image_width = W, ...).
 * workspace: ngp_render_frame_workspace(H*W) bytes. */
int ngp_render_frame_camera(const ngp_field_t* field_host, const float* pose_host, const float* intrinsics_host, uint32_t H, uint32_t W,
                            const float* aabb, float min_near, const uint8_t* bitfield, uint32_t C, uint32_t Hgrid,
                            float dt_gamma, uint32_t max_steps, const float* bg_color3_host,
                            float* image, float* depth, float* weights_sum, uint32_t* stats,
                            void* workspace, size_t workspace_bytes, void* stream);

/* P frames of one camera model in ONE launch (poses_host [P][16] row-major 4x4 cam2world on the host; outputs [P][H*W] contiguous): the
 * frame kernel's ramp and drain are paid once per launch instead of once per frame.  The same pixels, bit for bit, as P calls of
 * ngp_render_frame_camera.  H, W multiples of 8; 1 <= P <= 64; workspace: ngp_render_frames_workspace(P, H * W) bytes.
 * (No reference counterpart: nerf/utils.py:588-638 renders a test set one frame per call.) */
size_t ngp_render_frames_workspace(uint32_t P, uint32_t rays_per_frame);
int ngp_render_frames_camera(const ngp_field_t* field, const float* poses_host, uint32_t P, const float* intrinsics_host, uint32_t H, uint32_t W,
                             const float* aabb_host, float min_near, const uint8_t* bitfield, uint32_t C, uint32_t Hgrid,
                             float dt_gamma, uint32_t max_steps, const float* bg_color3_host,
                             float* image, float* depth, float* weights_sum, uint32_t* stats,
                             void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
