// cfnerf_kernels.h - kernel argument blocks and host launchers (internal to libcfnerf_hip.so)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "cfnerf_layout.h"

// every netwidth the fused kernels are instantiated for (validate_cfg: multiples of 64 up to 512)
#ifndef CFN_FOR_EACH_WIDTH      // (development builds narrow the list: -D'CFN_FOR_EACH_WIDTH(X)=X(256)')
#define CFN_FOR_EACH_WIDTH(X) X(64) X(128) X(192) X(256) X(320) X(384) X(448) X(512)
#endif

namespace cfnerf {

// internal bits of FwdArgs::flags (above the public CFNERF_F_*): the arithmetic flavour of the flow phase forced by
// cfnerf_model_set_flow_math (neither set: libm below kFastFlowsK latents, hardware transcendentals from there on)
constexpr int CFNERF_F_FLOW_MATH_SET = 1 << 16, CFNERF_F_FLOW_MATH_FAST = 1 << 17;

struct RaysC2W { float m[12]; };

struct FwdArgs {
    const float* wp;          // packed operands
    const void* wp16;         // split-bf16 copies of the operands (bf16x3 mode)
    const float* flat;        // flat parameters (base-Gaussian mean/std live at [0,8))
    const float* rays;        // [N,11]         (ray mode)
    const float* t_vals;      // [S]
    const float* t_rand;      // [N,S] or null
    const float* z_in;        // [N,S] explicit depths or null
    const float* eps;         // [K,4]
    const float* x;           // [P,90]         (points mode)
    int64_t N, P;             // rays, points (P = N*S in ray mode)
    int32_t S, K, flags;
    float *rgb_map, *disp, *depth;       // [N,3,K] [N,K] [N,K]
    float *raw, *weights, *pts;          // optional [P,K,4] [P,K] [P,3]
    float *kstats;                       // optional [N,8] fused K-statistics
    const float* gt;                     // optional [N,3] ground-truth colours (eval): with kstats, sqerr is written
    float* sqerr;                        // optional [N,3] (K-mean rgb - gt)^2 per pixel and channel
    float* ent_partials;                 // [grid,2]  (TRAIN)
    float* enc_scratch;                  // [grid, 64*64] per-workgroup parking slot of the encoded tile (used when there is no stash)
    // activation stash for the backward pass (all optional, row-major per point)
    float *st_enc, *st_gd, *st_h, *st_feat, *st_v, *st_ha, *st_hr, *st_theta, *st_z;
    // the K-proportional pieces of the stash are TILE-TRANSPOSED, [tile][k][64 rows][.]: lane = row in the kernels that write and read
    // them, so one latent of one tile is ONE contiguous piece per wave instruction (1 KB of raw, 512 B of (e, T)); the row-major
    // [P,K,.] form of rounds 1-4 put a lane's pieces 16 K bytes apart, and at K = 64 the tail kernel's 200 MB of strided reads
    // cost backward-data ~47 us of cache state (round 5, profiles/EXPERIMENTS.md)
    float *st_raw;                       // [tiles,K,64,4] flow outputs of the stashed forward (A.raw: the CALLER's row-major [P,K,4], optional)
    float *st_at;                        // [tiles,K,64,2] e = exp(-sigma dist) = 1 - alpha, transmittance T of the composite
    uint32_t* st_mbits;                  // [D+1][tiles][W/32][64] ReLU masks as fragment-ordered bit words
    int64_t n_tiles;                     // tiles of the launch (rays * chunks per ray)
    int32_t q4;                          // 1: the trunk activations h take the Q4 layout (cfnerf_device.h: whole tiles, fp32 mode; picks the kernel variant)
};

hipError_t launch_fused_fwd(const FwdArgs& a, const NetTab& host_tab, int mode, bool train, int prec, int n_cu, int per_cu, hipStream_t st, int* grid_out);
hipError_t fused_fwd_set_attributes(int W, int ha, int* per_cu_out);      // per device, at model creation
int fused_fwd_max_grid(int W, int ha, int n_cu);
hipError_t launch_entropy_finalize(const float* partials, int n_part, const float* flat, const float* eps, int K,
                                   double count, float* out, float* eps_keep, const float* rays, float* rays_keep, int64_t n_rays_floats,
                                   hipStream_t st);
hipError_t launch_composite(const float* raw, const float* z, const float* d, int64_t N, int S, int K, int wb,
                            float* rgb, float* disp, float* depth, float* weights, hipStream_t st);
hipError_t launch_rays_setup(int H, int Wd, float focal, const RaysC2W& c2w, int use_c2w, const float* ro, const float* rd,
                             int64_t N, int64_t pixel0, int ndc, float nearv, float farv, float* out, hipStream_t st);
hipError_t launch_ndc_rays(int H, int Wd, float focal, float nearv, const float* ro, const float* rd, int64_t N, float* out_o, float* out_d,
                           hipStream_t st);
hipError_t launch_sample_pdf(const float* rays, const float* t_vals, const float* t_rand, int flags, const float* w, const float* u,
                             int64_t N, int S, int K, int Ni, float* z_out, hipStream_t st);
hipError_t launch_embed(const float* x, int64_t P, int multires, float* out, hipStream_t st);
hipError_t launch_sample_points(const float* rays, const float* t_vals, const float* t_rand, int flags, int64_t N, int S, float* z, float* pts,
                                hipStream_t st);
hipError_t launch_pack_index(const PackDesc* descs, int ndesc, uint32_t total, uint32_t* table /*[total][3]*/, hipStream_t st);      // once per model
hipError_t launch_pack(const float* flat, float* packed, void* packed16, const uint32_t* table, uint32_t total, hipStream_t st);

}  // namespace cfnerf
