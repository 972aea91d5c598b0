/*
 * cfnerf.h - C ABI of the MI355X-native CF-NeRF ray-batch hot path (libcfnerf_hip.so).
 *
 * The reference (poetrywanderer/CF-NeRF) is pure Python/PyTorch and has NO FFI layer: its seam is
 * the Python call chain  render() -> render_rays() -> network_query_fn() -> NeRF_Flows.forward()
 * -> raw2outputs()  (run_nerf_uncertainty_NF.py:103-170, 457-553, 67-85, 411-454;
 * model/models.py:188-291).  Each entry point below names the reference function it replaces.
 * The Python host mirror (cf-nerf_amd/) binds these symbols with ctypes and re-exposes the
 * reference's signatures; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer: outputs, parameters, gradients, Adam moments and - through
 *     cfnerf_workspace_bytes / cfnerf_model_set_workspace - the train-step workspace (activation stash,
 *     pre-activation gradients, split-K partials).  A cfnerf_model itself owns only the packed weight copies and a
 *     few KB of scratch, allocated in cfnerf_model_create.  A caller that hands in NO workspace gets a model-owned
 *     one that grows on demand (the only allocation outside create; it synchronises the device when it grows);
 *   - every launch is asynchronous on the hipStream_t passed in (pass torch's current stream); on the steady
 *     path (same N, S, K and workspace as the previous step) no entry point allocates, frees or synchronises.
 *     NOT steady: a caller that alternates two shapes on one model - the coarse + fine sampling EXTENSION of the host
 *     mirror (train.Trainer.step_hierarchical: (N, 64, K) then (N, 192, K) every step) re-binds the one workspace twice
 *     per step, so both of its cfnerf_render_bwd calls take the plan-rebuild path (weight-gradient descriptors rebuilt
 *     and uploaded from double-buffered host copies, <= 300 MB of partials cleared: ~50 us each, asynchronous, but a host
 *     sync when the model-owned workspace has to grow).  One plan slot per model is the contract; a caller that needs two
 *     shapes at full speed uses two cfnerf_model handles, each re-packed from the one parameter buffer (cfnerf_model_set_params);
 *   - return value: 0 = OK, negative = cfnerf_status; cfnerf_last_error() gives a thread-local
 *     message.  No C++ exception crosses the ABI;
 *   - a handle lives on the device that was current at cfnerf_model_create (one handle per device, one
 *     process per GPU); calls made with another device current are rejected.  Calls that do not take a
 *     handle are stateless and re-entrant.  Calls on ONE handle share its workspaces (entropy partials,
 *     stash, split-K partials): issue them from one host thread at a time and on one stream at a time
 *     (or order the streams with events); different handles are independent;
 *   - all arithmetic is fp32 (exact-fp32 MFMA v_mfma_f32_32x32x2_f32 for the dense layers).
 */
#ifndef CFNERF_H
#define CFNERF_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: its dynamic symbol table holds the entry points declared in THIS header and
 * nothing else - no C++ symbol, no helper, no test hook (those live in a separate test library, tests/cfnerf_debug.h).
 * tests/test_abi_cpu.py compares `nm -D --defined-only` of the library with this header. */
#if defined(__GNUC__) || defined(__clang__)
#define CFNERF_API __attribute__((visibility("default")))
#else
#define CFNERF_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cfnerf_stream;            /* hipStream_t */

typedef enum cfnerf_status {
    CFNERF_OK = 0,
    CFNERF_E_INVALID = -1,              /* bad argument / shape */
    CFNERF_E_UNSUPPORTED = -2,          /* configuration the reference accepts silently but this path rejects */
    CFNERF_E_HIP = -3,                  /* HIP runtime error (message holds hipGetErrorString) */
    CFNERF_E_NOMEM = -4
} cfnerf_status;

/* The flags of config_parser() that reach the hot path (run_nerf_uncertainty_NF.py:556-719). */
typedef struct cfnerf_cfg {
    int32_t netdepth;                   /* --netdepth   (8)   skip connection after layer netdepth/2 (RUN:327) */
    int32_t netwidth;                   /* --netwidth   (256) a multiple of 64 in [64, 512] */
    int32_t multires;                   /* --multires   (10)  -> 63 input channels (run_nerf_helpers.py:54-69) */
    int32_t multires_views;             /* --multires_views (4) -> 27 channels */
    int32_t h_alpha_size;               /* --h_alpha_size (32) 32, 64, 96 or 128 */
    int32_t h_rgb_size;                 /* --h_rgb_size (64)  32, 64, 96 or 128, with netwidth / 2 + h_rgb_size <= max(netwidth, 128) */
    int32_t n_flows;                    /* --n_flows    (4)   1 .. 4 (fewer than 4 steps run as 4 with the missing steps' parameters zero: identity steps) */
} cfnerf_cfg;

typedef struct cfnerf_model cfnerf_model;   /* opaque: packed weights + workspaces for ONE device */

/* flags for the render / network entry points */
enum {
    CFNERF_F_TRAIN      = 1 << 0,       /* train branch of NeRF_Flows.forward (MOD:225-291): log-dets + entropy */
    CFNERF_F_LINDISP    = 1 << 1,       /* render_rays(lindisp=True)  RUN:513-514 */
    CFNERF_F_WHITE_BKGD = 1 << 2,       /* raw2outputs(white_bkgd=True) RUN:451-452 */
    CFNERF_F_STASH      = 1 << 3        /* keep activations for cfnerf_render_bwd (implies TRAIN); ONE stash per model: a later
                                           STASH forward replaces it and bumps cfnerf_model_stash_generation */
};

CFNERF_API int         cfnerf_version(void);
CFNERF_API const char* cfnerf_last_error(void);

/* ---- parameters ------------------------------------------------------------------------------
 * Parameters live in ONE flat fp32 device buffer owned by the caller, laid out in the order of
 * NeRF_Flows.state_dict() (model/models.py:38-67, 339-350), each tensor
 * row-major as nn.Linear stores it ([out, in]).  cfnerf_param_count / cfnerf_param_offset describe
 * that layout so the host side can map state_dict keys to slices of the buffer.  The same layout
 * is used for gradients and Adam moments, so the multi-GPU exchange is a single all-reduce.      */
CFNERF_API int64_t cfnerf_param_count(const cfnerf_cfg* cfg);
/* offset (in floats) and element count of a state_dict key such as "pts_linears.5.weight"; -1 if unknown */
CFNERF_API int64_t cfnerf_param_offset(const cfnerf_cfg* cfg, const char* key, int64_t* numel);
/* i-th key of the layout (0 <= i < number of tensors), NULL past the end */
CFNERF_API const char* cfnerf_param_key(const cfnerf_cfg* cfg, int index);

/* replaces: create_nerf()'s NeRF_Flows(args) construction, RUN:317-331 (device side only) */
CFNERF_API int cfnerf_model_create(const cfnerf_cfg* cfg, cfnerf_model** out);
CFNERF_API int cfnerf_model_destroy(cfnerf_model* m);
/* (Re)pack the flat parameter buffer into the MFMA-fragment-ordered copies the kernels stream.
 * Call after loading a checkpoint and after every optimiser step.  replaces: the implicit
 * parameter broadcast of nn.DataParallel, RUN:330 */
CFNERF_API int cfnerf_model_set_params(cfnerf_model* m, const float* flat_params, cfnerf_stream s);

/* ---- ray set-up ------------------------------------------------------------------------------
 * replaces: the ray preparation inside render(), RUN:129-158, with get_rays (HLP:288-297) and
 * ndc_rays (HLP:360-377).  Either `rays_o`/`rays_d` ([N,3] each) are given, or (c2w_host != NULL)
 * rays are generated for the N pixels pixel0 .. pixel0+N-1 (row-major) of the H x W image, so ranks can
 * tile an image by rows.  Output `rays` is the [N,11] pack o3,d3,near,far,viewdir3 of RUN:152-158. */
CFNERF_API int cfnerf_rays_setup(int H, int W, float focal, const float* c2w_host /*[3,4] row-major or NULL*/,
                      const float* rays_o, const float* rays_d, int64_t N, int64_t pixel0,
                      int ndc, float near_, float far_, float* rays /*[N,11]*/, cfnerf_stream s);

/* replaces: ndc_rays(H, W, focal, near, rays_o, rays_d) as a standalone call, HLP:360-377 (render() itself goes through
 * cfnerf_rays_setup, which applies it with near = 1 like RUN:149): rays_o, rays_d [N,3] -> out_o, out_d [N,3] in NDC.
 * (get_rays, HLP:288-297, as a standalone call is cfnerf_rays_setup with c2w_host and ndc = 0: columns 0..5 of its output.)   */
CFNERF_API int cfnerf_ndc_rays(int H, int W, float focal, float near_, const float* rays_o, const float* rays_d, int64_t N,
                    float* out_o, float* out_d, cfnerf_stream s);

/* replaces: Embedder.embed / get_embedder(multires) as a standalone call, HLP:21-69:
 * x [P,3] -> out [P, 3 + 6*multires] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]        */
CFNERF_API int cfnerf_embed(const float* x, int64_t P, int multires, float* out, cfnerf_stream s);

/* replaces: the sampling lines of render_rays as a standalone call, RUN:510-534 (used by the unfused query path):
 * rays [N,11], t_vals [S], t_rand [N,S] or NULL, LINDISP flag -> z_vals [N,S], pts [N,S,3]                      */
CFNERF_API int cfnerf_sample_points(const float* rays, const float* t_vals, const float* t_rand, int flags, int64_t N, int S,
                         float* z_vals, float* pts, cfnerf_stream s);

/* ---- fused forward ---------------------------------------------------------------------------
 * replaces: render_rays() RUN:457-553 = sampling RUN:510-534, run_network RUN:67-85 with the
 * positional encoding HLP:21-69, NeRF_Flows.forward MOD:188-291 (TriangularSylvesterNeRF
 * MOD:358-416, FLW:189-268) and raw2outputs RUN:411-454, in one launch.
 *   rays    [N,11]          t_vals [S]            t_rand [N,S] or NULL (perturb == 0)
 *   z_vals_opt [N,S] or NULL: explicit sample depths per ray (then t_vals / t_rand are not read) - used by the
 *                     hierarchical-sampling EXTENSION below, which the reference does not have
 *   eps     [K,4] = (eps_rgb0, eps_rgb1, eps_rgb2, eps_alpha) per latent sample (MOD:234,246 / 198,204)
 *   rgb_map [N,3,K]  disp_map [N,K]  depth_map [N,K]           (written unless all three are NULL and kstats is given)
 *   raw_opt [N,S,K,4], weights_opt [N,S,K], pts_opt [N,S,3]    (NULL = not wanted)
 *   kstats_opt [N,8]  fused reductions over the K latent samples, what the evaluation loop derives from the
 *                     per-K maps at RUN:1122-1131: mean_K rgb (3) | np.std_K(rgb) * n/(n-1) (3) | mean_K disp | mean_K depth
 *   entropy_out [1]   loss_entropy of MOD:286 (TRAIN only, may be NULL otherwise)                */
CFNERF_API int cfnerf_render_fwd(cfnerf_model* m, const float* rays, const float* t_vals, const float* t_rand,
                      const float* z_vals_opt, const float* eps, int64_t N, int S, int K, int flags,
                      float* rgb_map, float* disp_map, float* depth_map,
                      float* raw_opt, float* weights_opt, float* pts_opt, float* kstats_opt, float* entropy_out,
                      cfnerf_stream s);

/* replaces: what the training loop's evaluation block derives from a full-image render (render_path_train RUN:247-314 with
 * RUN:1117-1131: K-mean prediction, np.std * n/(n-1) uncertainty, mean disparity / depth) and the per-pixel integrand of
 * img2mse(rgb_mean, target) (RUN:1028, HLP:15), all reduced INSIDE the fused forward: only 32 (+12) bytes per pixel leave
 * the chip instead of 20*K.  Eval branch (fixed eps, no jitter).  kstats [N,8] as in cfnerf_render_fwd; gt_opt [N,3] and
 * sqerr_opt [N,3] = (K-mean rgb - gt)^2 go together or are both NULL.                                                  */
CFNERF_API int cfnerf_render_eval(cfnerf_model* m, const float* rays, const float* t_vals, const float* eps, int64_t N, int S, int K,
                       int flags, const float* gt_opt, float* kstats, float* sqerr_opt, cfnerf_stream s);

/* EXTENSION (not in the reference, whose N_importance / network_fine are dead parameters, RUN:467-468; the
 * semantics restated are those of the reference's upstream, yenchenlin/nerf-pytorch sample_pdf): inverse-CDF
 * resampling of N_importance depths per ray from the K-mean of the coarse weights, merged and sorted with the
 * coarse depths.  The coarse depths are recomputed from (rays, t_vals, t_rand, LINDISP flag) exactly as
 * cfnerf_render_fwd samples them.  weights [N,S,K], u [N,N_importance] in [0,1] -> z_out [N,S+N_importance].    */
CFNERF_API int cfnerf_sample_pdf(const float* rays, const float* t_vals, const float* t_rand, int flags, const float* weights,
                      const float* u, int64_t N, int S, int K, int N_importance, float* z_out, cfnerf_stream s);

/* replaces: NeRF_Flows.forward(x, is_val, is_test) MOD:188-291 on pre-embedded inputs x [P,90]
 * (what batchify()/run_network hand to the model, RUN:47-64,82).  raw [P,K,4].  With CFNERF_F_STASH (implies TRAIN) the
 * activations are kept for cfnerf_network_bwd (the model's ONE stash, bound as one "ray" of P samples: size the workspace
 * with cfnerf_workspace_bytes(cfg, 1, P, K)).                                                     */
CFNERF_API int cfnerf_network_fwd(cfnerf_model* m, const float* x, const float* eps, int64_t P, int K, int flags,
                       float* raw, float* entropy_out, cfnerf_stream s);

/* replaces: raw2outputs() RUN:411-454 as a standalone call.  raw [N,S,K,4] (16-byte aligned, S * K * 16 bytes per ray < 2 GiB: it is
 * fetched through one buffer descriptor per ray), z_vals [N,S], rays_d [N,3]; any K >= 1, any S >= 1 */
CFNERF_API int cfnerf_composite_fwd(const float* raw, const float* z_vals, const float* rays_d,
                         int64_t N, int S, int K, int white_bkgd,
                         float* rgb_map, float* disp_map, float* depth_map, float* weights_opt,
                         cfnerf_stream s);

/* ---- train step ------------------------------------------------------------------------------
 * replaces: the loss lines RUN:1026-1050 (K-mean MSE/PSNR, KDE negative log-likelihood with the
 * detached n/(n-1)-scaled bandwidth, + beta1 * entropy).  Writes d(loss)/d(rgb_map) [N,3,K] and
 * scalars_out[4] = {loss, loss_nll, mse, psnr}.  `n_total` is the GLOBAL ray count the means are
 * taken over (N for one GPU, N * world_size when rays are sharded).  scalars_out must be 8-byte aligned (its
 * 16 bytes double as the two 64-bit fixed-point accumulators of the multi-workgroup reduction).     */
CFNERF_API int cfnerf_loss_fwd_bwd(const float* rgb_map, const float* target, const float* entropy, int64_t N, int K,
                        float beta1, int64_t n_total, float* d_rgb_map, float* scalars_out, cfnerf_stream s);

/* ---- train-step workspace (ownership contract of SURVEY 8b) -------------------------------------
 * Bytes of device memory a CFNERF_F_STASH forward + cfnerf_render_bwd of an (N rays, S samples, K latents) batch
 * need: the activations autograd would have kept for loss.backward() (RUN:1066), the pre-activation gradients
 * and the split-K weight-gradient partials.  -1 on a bad argument.                                               */
CFNERF_API int64_t cfnerf_workspace_bytes(const cfnerf_cfg* cfg, int64_t N, int S, int K);
/* Hand the model a caller-owned block (256-byte aligned, e.g. a torch uint8 tensor) to use as that workspace.  The
 * library then never allocates on the train path: a batch that does not fit is refused with CFNERF_E_NOMEM.  The
 * block must stay alive until the next cfnerf_model_set_workspace / cfnerf_model_destroy.  (NULL, 0) returns to
 * the default, a model-owned block grown on demand.  Any stashed forward is dropped.                            */
CFNERF_API int cfnerf_model_set_workspace(cfnerf_model* m, void* workspace, size_t bytes);
/* Identity of the forward the model's ONE stash currently holds: every CFNERF_F_STASH forward increments it.
 * 0 = no stashed forward.  Read it right after the forward and pass it to cfnerf_render_bwd.                    */
CFNERF_API uint64_t cfnerf_model_stash_generation(const cfnerf_model* m);

/* replaces: loss.backward() (RUN:1066) through raw2outputs, the flows and the MLP for the batch
 * of the cfnerf_render_fwd(... CFNERF_F_STASH ...) whose generation is `stash_generation`; if a later STASH forward
 * has replaced that stash the call fails (CFNERF_E_INVALID) instead of differentiating the wrong batch.
 * d_depth_map may be NULL.  d_entropy points to ONE device float, d(loss)/d(loss_entropy) (e.g. beta1); NULL
 * means 0.  grad_flat [param_count] is OVERWRITTEN with the gradient in the flat parameter layout.             */
CFNERF_API int cfnerf_render_bwd(cfnerf_model* m, uint64_t stash_generation, const float* d_rgb_map, const float* d_depth_map,
                      const float* d_entropy, float* grad_flat, cfnerf_stream s);
/* replaces: the same loss.backward() when ONE optimiser step's batch is walked in slices (the reference renders any N_rand and any
 * `chunk`, RUN:88-100,602: its autograd graph grows with the batch; here the train-step workspace does - 3 MiB per ray at W = 256 -
 * so a batch larger than the workspace the caller wants to lend is cut into equal slices, each rendered with CFNERF_F_STASH and
 * differentiated before the next one replaces the stash).  Exactly cfnerf_render_bwd, except that the slice's gradient is ADDED to
 * grad_flat: call cfnerf_render_bwd for the first slice (it overwrites) and this for the others, with the loss of every slice taken
 * with n_total = the FULL batch (cfnerf_loss_fwd_bwd) and d_entropy = beta1 / n_slices.  The sum over slices is the full batch's
 * gradient up to the summation order (tests/test_hip_train.py: 8 x 1024 rays against one 8192-ray launch).  The early ranges
 * (cfnerf_grad_early_ranges) are final only after the LAST slice's call.                                                           */
CFNERF_API int cfnerf_render_bwd_accumulate(cfnerf_model* m, uint64_t stash_generation, const float* d_rgb_map, const float* d_depth_map,
                                 const float* d_entropy, float* grad_flat, cfnerf_stream s);

/* ---- the UNFUSED seam, differentiable like the reference's ---------------------------------------------------------
 * In the reference NeRF_Flows.forward (MOD:188-291) and raw2outputs (RUN:411-454) are ordinary autograd graphs, so a caller
 * that injects its own network_query_fn (RUN:382-394, called at RUN:538) still trains.  These two entry points are the
 * tail of cfnerf_render_bwd split at `raw`, with the same arithmetic.
 *
 * replaces: loss.backward() through NeRF_Flows.forward for the cfnerf_network_fwd(... CFNERF_F_STASH ...) whose generation
 * is `stash_generation`.  d_raw [P,K,4] = d loss / d raw (NULL = zeros), d_entropy = ONE device float, d loss /
 * d loss_entropy (NULL = 0).  grad_flat [param_count] is OVERWRITTEN.  Gradients with respect to the inputs x are not
 * produced (the reference's sample points are not parameters).                                                         */
CFNERF_API int cfnerf_network_bwd(cfnerf_model* m, uint64_t stash_generation, const float* d_raw, const float* d_entropy,
                       float* grad_flat, cfnerf_stream s);
/* replaces: loss.backward() through raw2outputs(raw, z_vals, rays_d) RUN:411-454, stateless: the forward is recomputed from
 * raw [N,S,K,4], z_vals [N,S], rays_d [N,3].  d_rgb_map [N,3,K]; d_disp_map [N,K], d_depth_map [N,K], d_weights [N,S,K] may
 * be NULL (= zeros).  Writes d_raw [N,S,K,4] = d loss / d raw.  S <= 4096.                                               */
CFNERF_API int cfnerf_composite_bwd(const float* raw, const float* z_vals, const float* rays_d, int64_t N, int S, int K, int white_bkgd,
                         const float* d_rgb_map, const float* d_disp_map, const float* d_depth_map, const float* d_weights,
                         float* d_raw, cfnerf_stream s);

/* ---- overlap of the multi-GPU gradient exchange with the end of the backward --------------------------------
 * replaces: nothing in the reference (its nn.DataParallel, RUN:330, gathers gradients inside autograd).  Most of
 * grad_flat - every bias, the base Gaussians, and each weight whose gradient comes from the big weight-gradient
 * launches alone - is final BEFORE the small-job launch that ends cfnerf_render_bwd.  cfnerf_grad_early_ranges
 * reports those flat ranges (they depend on the configuration only; available after the first cfnerf_render_bwd);
 * cfnerf_stream_wait_grad_early makes `waiter` (e.g. the communication stream) wait for the event recorded at
 * that point of the LAST cfnerf_render_bwd, so the all-reduce of those ranges runs while the rest still computes.
 * The early point exists from the first cfnerf_render_bwd AFTER a call of cfnerf_grad_early_ranges on (a caller that never asks
 * gets one reduction launch at the end instead of two); before that the event fires when the whole gradient is final - a
 * waiter is always correct, only not early.                                                                                 */
CFNERF_API int cfnerf_grad_early_ranges(cfnerf_model* m, int64_t* offsets, int64_t* counts, int max_ranges);
CFNERF_API int cfnerf_stream_wait_grad_early(cfnerf_model* m, cfnerf_stream waiter);

/* replaces: torch.optim.Adam.step() RUN:339,1067 on the flat buffers (betas .9/.999, eps 1e-8),
 * followed by the re-pack of cfnerf_model_set_params.  step is 1-based.  grad_scale multiplies
 * the gradient first (1/world_size after a sum all-reduce).                                      */
CFNERF_API int cfnerf_adam_step(cfnerf_model* m, float* flat_params, const float* grad_flat, float* exp_avg,
                     float* exp_avg_sq, int64_t step, float lr, float grad_scale, cfnerf_stream s);

/* OPT-IN arithmetic mode of the fused forward's dense layers.  0 (default): exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32).  1: "bf16x3" - every fp32 operand is carried as hi + lo bf16 and a product is evaluated
 * as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped lo*lo term is ~2^-18
 * relative); it is held to the SAME parity tolerances by tests/test_hip_bf16x3.py.  Applies to the fused forward,
 * the backward-data kernel and the large weight-gradient GEMMs (the narrow ones stay exact fp32).  A stashed forward belongs to
 * the mode it ran in: CHANGING the mode drops it (its generation id is retired, like after cfnerf_model_set_workspace).     */
CFNERF_API int cfnerf_model_set_precision(cfnerf_model* m, int mode);

/* Arithmetic of the flow phase of the fused kernels (the K conditional Sylvester flows, MOD:401-413 / FLW:225-268, the activations
 * and the composite, RUN:424-449, of every (point, latent sample)).  1: libm throughout (correctly rounded tanhf / logf / expf /
 * log1pf, IEEE division) - ~1300 vector instructions per (point, latent).  2: the same functions on the hardware transcendentals
 * (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each; tanh = 1 - 2 / (1 + e^2x)) - ~250 instructions; held to the same parity bounds
 * by the tests.  0 (default): 1 below 16 latent samples (the reference's plumbing and headline configurations stay on libm bit
 * for bit), 2 from 16 on, where the flow phase grows from 5 % (K = 16) to 20 % (K = 64, the reference's default) of the launch.   */
CFNERF_API int cfnerf_model_set_flow_math(cfnerf_model* m, int mode);

/* bytes currently held by the model: packed weights + the bound workspace, whoever owns it (diagnostics) */
CFNERF_API int64_t cfnerf_model_workspace_bytes(const cfnerf_model* m);

/* Measurement helpers for bench.py: kernel durations from HIP events recorded on the launch stream.
 *   mode 0: off (default).  mode 1: every stage of a step (ten events per train step: costs ~1 % of it).
 *   mode 2: the fused forward launch only (two events per step) - what the timed region of bench.py runs with.
 * cfnerf_timing_fwd_mean_ms: mean duration (ms) of the last min(n, 64) timed fused-forward launches since the mode was set.
 * cfnerf_timing_last_ms: last launch of a stage (stages 1..4 need mode 1).  Both return < 0 when nothing was timed.        */
CFNERF_API int   cfnerf_timing_enable(cfnerf_model* m, int mode);
CFNERF_API float cfnerf_timing_fwd_mean_ms(cfnerf_model* m, int n);
CFNERF_API float cfnerf_timing_last_ms(cfnerf_model* m, int which /*0=fwd 1=bwd_tail 2=bwd_data 3=bwd_dw 4=adam*/);

#ifdef __cplusplus
}
#endif
#endif /* CFNERF_H */
