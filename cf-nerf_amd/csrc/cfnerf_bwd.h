// cfnerf_bwd.h - argument blocks and per-model plan of the backward pass (internal)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "cfnerf_layout.h"

namespace cfnerf {

struct TailArgs {
    const float *raw, *theta, *at, *z, *rays, *eps, *flat;
    const float *d_rgb, *d_depth, *d_ent;
    int64_t N, P;
    int32_t S, K, flags;
    int32_t ksplit;                                       // waves per ray (1, 2 or 4 = tail_parts): wave (ray, part) handles latents [part*Kp, (part+1)*Kp);
                                                          // the parts of a ray are waves of ONE workgroup and add their g_theta rows in LDS
    float *g_theta, *gms_partials;                        // [P,128] theta gradients (one row per point), [N*ksplit, 8]
};

struct BwdArgs {
    const float* wp;
    const void* wp16;                                     // split-bf16 operand copies (bf16x3 mode)
    int64_t P;
    int32_t n_wg, nb;
    const float* g_theta;                                 // [P,128]   input: d loss / d theta from the tail kernel (or flows_bwd)
    float *g_hr, *g_ha, *g_v, *g_feat, *g_h;              // outputs: pre-activation gradients, row-major per point
    const uint32_t* mbits;                                // [D+1][tiles][W/32][64] ReLU masks (fragment-ordered bit words)
    int64_t n_tiles; int32_t S;                           // same tiling as the forward: tile = ray * chunks_per_ray + chunk
    float* dbp;                                           // [n_wg, nb] bias-gradient partials
    int32_t q4;                                           // 1: g_h / g_feat leave in the Q4 layout (cfnerf_device.h), as the forward wrote h
    int32_t db_h, db_feat, db_v, db_ha, db_hr, db_theta;  // column offsets inside a dbp row
};

// the flow-adjoint kernels live in cfnerf_tail.hip (a translation unit compiled without the SLP vectoriser)
hipError_t launch_tail_bwd(const TailArgs& ta, int64_t n_rays, int ksplit, hipStream_t st);
hipError_t launch_flows_bwd(const float* raw, const float* theta, const float* eps, const float* flat, const float* d_raw, const float* d_ent,
                            int64_t P, int K, float* g_theta, float* gms_partials, unsigned* grid_out, hipStream_t st);

// one 128 x 256 output tile of a weight-gradient job  dW[n][k] = sum_p dY[p][n] * X[p][k]
struct DwTile {
    const float* dY; int32_t ldY, N, Npad;                // Npad: readable width of a dY row from the slice start
    const float* X;  int32_t ldX, K, Kpad;                // K: columns to store, Kpad: readable width of an X row
    int32_t n0, k0;
    int32_t nseg, seg_row[4];                             // destination row segments (concatenated flow heads)
    uint32_t seg_dst[4];
    int32_t dst_ld, dst_col;
    int32_t gk, wk;                                       // small kernel: waves along k, k-tiles per wave (GN = 8 / gk);
                                                          // big kernel: gk = wave arrangement (0: 2 x 4, 1: 1 x 8 for N <= 128)
    int32_t nsplit;                                       // point splits of this tile (blocks per tile)
    int32_t lay;                                          // operand layouts: bit 0 = dY, bit 1 = X is a Q4 stream (cfnerf_device.h), else row-major
    int32_t row_f;                                        // 0, or (theta-head tiles) the model's n_flows: dY column 4 b + f is destination row b F + f of
                                                          // the concatenated heads, columns with f >= n_flows are dropped (cfnerf_layout.h)
    int32_t late;                                         // 1: which launch this job runs in depends on the stash LAYOUT (ONE Q4-capable operand: big row-major, small
                                                          // with Q4) - its tensor is never reported as early, so the early ranges depend on the configuration only
};

struct BiasMap { int32_t col0, count; uint32_t dst; };

// a parameter tensor's start offset and the number of split slots its gradient partials occupy
// nsplit: slots to sum; -1 = not written by the weight reduction (biases: reduce_bias, base Gaussians: reduce_gms).
// early: 1 = every tile of the tensor belongs to the big launches, so its sum is final BEFORE the small-job launch.
struct RedSeg { uint32_t begin; int32_t nsplit; int32_t early; int32_t pad_; };

// one workgroup of a weight-gradient launch: tile index, split slot and its point range
struct DwBlock { int32_t tile, split, kslice, pad_; int64_t pb, pe; };

// host copies of the weight-gradient descriptors of one plan; they must stay alive while their upload may be in flight
struct DwHost {
    std::vector<DwTile> tiles, tiles_small;
    std::vector<DwBlock> blocks, blocks_small;
    std::vector<RedSeg> segs;
    hipEvent_t uploaded = nullptr;                        // recorded after this set's uploads
};

struct BwdPlan {
    bool built = false;
    int nb = 0, db_h = 0, db_feat = 0, db_v = 0, db_ha = 0, db_hr = 0, db_theta = 0;
    std::vector<BiasMap> bias_maps;
    DwHost host[2];                                       // double-buffered: a rebuild never waits for the previous upload
    int cur = 0;                                          // the set the device copies were made from
    uint64_t bind_serial = ~0ull;                         // Stash::bind_serial the descriptors were built for
    bool q4 = false;                                      // ... and the layout of the wide streams (Stash::q4) they describe
    int n_blocks_wide = 0;                                // blocks[0, n_blocks_wide): 2 x 4 tiles; the rest: 1 x 8 tiles
    hipEvent_t ev_early = nullptr;                        // recorded once the "early" tensors' gradients are final
    bool early_wanted = false;                            // the caller has asked for the early ranges (cfnerf_grad_early_ranges): reduce them before the small jobs
    std::vector<int64_t> early_off, early_cnt;            // flat ranges of grad_flat that are final at ev_early (merged, sorted)
    void release() {
        for (DwHost& h : host) { if (h.uploaded) (void)hipEventDestroy(h.uploaded); h.uploaded = nullptr; }
        if (ev_early) (void)hipEventDestroy(ev_early);
        ev_early = nullptr;
        bind_serial = ~0ull;
    }
};

}  // namespace cfnerf
