// cfnerf_model.h - the opaque cfnerf_model handle: packed weights, operand table, train-step workspace.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "cfnerf_layout.h"
#include "cfnerf_bwd.h"

namespace cfnerf {

constexpr int kNumTimers = 5;   // 0 fwd, 1 bwd_tail, 2 bwd_data, 3 bwd_dw, 4 adam
constexpr int kFwdRing = 64;    // event pairs kept for the forward launch (mean over the last n launches)
constexpr int kDwSlots = 128;   // split-K slots of the weight-gradient partials (upper bound of any split count)
constexpr int kMaxDwTiles = 256, kMaxDwBlocks = 16384;     // descriptor capacities carved out of the workspace
constexpr int kTailParts = 4;   // the tail kernel splits a ray's K latents over up to this many waves of one workgroup (they meet in LDS)
constexpr int kTailMinPerPart = 2;   // a second wave per SIMD pays down to 2 latents per part (32 before the parts of a ray met in LDS)
// k-parts of the tail kernel for a batch: it runs ONE wave per SIMD, so a ray's latents are split over more waves only while all
// waves still fit in one round (measured: a second round costs more than the shorter k-loops save); at most kTailParts, never more
// than K / 2; always a power of two (1, 2 or 4), so a ray's parts are waves of ONE 4-wave workgroup.
// Round 4: the kernel fits 256 registers since it is built without the SLP vectoriser (cfnerf_tail.hip), so a SECOND wave per SIMD is
// possible.  While the parts met in memory (a partial g_theta per part, summed by backward-data) it paid only with >= 32 latents per part;
// since they meet in LDS (tail_bwd_kernel) backward-data does not see them - ONE g_theta row per point - and the second wave pays down to 2 latents per
// part: same-box A/B tail 57.8 -> 55.0 us (C2, K = 4), 93.4 -> 87.4 (C4, K = 16), 81.0 -> 76.5 (W512: four parts of 8), backward-data +-0.
inline int tail_parts(int64_t n_rays, int k, int n_cu) {
    int parts = 1;
    while (parts < kTailParts && parts * 2 <= k) {
        const bool one_round = n_rays * parts * 2 <= (int64_t)n_cu * 4;
        const bool second_wave = n_rays * parts * 2 <= (int64_t)n_cu * 8 && k / (parts * 2) >= kTailMinPerPart;
        if (!one_round && !second_wave) break;
        parts *= 2;
    }
    return parts;
}

// Everything a CFNERF_F_STASH forward keeps for cfnerf_render_bwd plus every buffer the backward writes, carved out of
// ONE block of device memory: either handed in by the caller (cfnerf_model_set_workspace, sized with
// cfnerf_workspace_bytes) or - when the caller gave none - owned by the model and grown on demand.
// Activations are row-major per point.  ~10.6 KB per point at W = 256 (1.4 GB for 1024 rays x 128 samples), the
// same again for the pre-activation gradients: sized for 288 GB of HBM.
struct Stash {
    // ---- written by the forward
    float *enc = nullptr;    // [P,64]   gamma(p) (padded)
    float *gd = nullptr;     // [P,32]   gamma(d) (padded)
    float *h = nullptr;      // [D,P,W]  trunk activations (post-ReLU)
    float *feat = nullptr;   // [P,W]
    float *v = nullptr;      // [P,W/2]  views layer (post-ReLU)
    float *ha = nullptr;     // [P,HA]
    float *hr = nullptr;     // [P,HR]
    float *theta = nullptr;  // [P,128]  flow parameters (diagonals tanh-ed)
    float *z = nullptr;      // [P]      z_vals
    float *raw = nullptr;    // [tiles,K,64,4]  tile-transposed (cfnerf_kernels.h: st_raw)
    float *rays = nullptr;   // [N,11]
    float *at = nullptr;     // [tiles,K,64,2]  e = exp(-sigma dist) (alpha = 1 - e), T; tile-transposed
    float *mbits = nullptr;  // [D+1][tiles][W/32][64] u32 ReLU mask words (fragment order)
    // ---- written by the backward (same row-major-per-point convention)
    float *gms = nullptr;       // [N*parts,8]    base-Gaussian gradient partials
    float *g_theta = nullptr;   // [P,128]  d loss / d theta (pre-tanh for the diagonal columns)
    float *g_hr = nullptr;      // [P,HR]
    float *g_ha = nullptr;      // [P,HA]
    float *g_v = nullptr;       // [P,W/2]  pre-activation gradient of the views layer
    float *g_feat = nullptr;    // [P,W]
    float *g_h = nullptr;       // [D,P,W]  pre-activation gradients of the trunk layers
    float *dbp = nullptr;       // [n_wg, nb] per-workgroup bias-gradient partials
    float *partials = nullptr;  // [kDwSlots, n_params] split-K weight-gradient partials
    float *zeros = nullptr;     // 256 B, unused since the weight-gradient loaders read out-of-range rows as zeros through their
                                // buffer descriptors (kept so the workspace layout / size does not change)
    BiasMap* bias_maps = nullptr;
    DwTile *tiles = nullptr, *tiles_small = nullptr;
    DwBlock *blocks = nullptr, *blocks_small = nullptr;
    RedSeg* segs = nullptr;

    int64_t n_tiles = 0;
    int64_t N = 0; int S = 0, K = 0, flags = 0;
    bool valid = false;             // a STASH forward has filled it
    bool points = false;            // ... in points mode (cfnerf_network_fwd: N = 1, S = P): differentiated by cfnerf_network_bwd
    bool q4 = false;                // ... with the trunk streams (h; then g_h, g_feat) in the Q4 layout (cfnerf_device.h): whole tiles, fp32
    uint64_t generation = 0;        // bumped by every STASH forward; cfnerf_render_bwd checks the caller's copy against it

    // binding
    char* base = nullptr; size_t cap = 0; bool owned = false;
    int64_t bound_N = -1; int bound_S = 0, bound_K = 0;
    uint64_t bind_serial = 0;       // bumped whenever the pointers above move (the weight-gradient tile list depends on them)
    size_t used = 0;

    // Lay the buffers out from `b` (nullptr: only measure).  Returns the bytes needed.  256-B aligned pieces.
    static size_t carve(Stash* q, char* b, const cfnerf_cfg& c, int64_t n, int s, int k, int64_t n_params) {
        const int W = c.netwidth, D = c.netdepth;
        const int64_t P = n * (int64_t)s;
        const int64_t tiles = n * (int64_t)((s + kTileM - 1) / kTileM);
        const int n_wg = kMaxCu * ((W <= 256) ? 2 : 1);
        const int nb = bias_partial_cols(c), n_bias_maps = bias_map_count(c);
        size_t off = 0;
        auto take = [&](auto** p, size_t count, size_t elem) {
            if (q) *p = reinterpret_cast<std::remove_reference_t<decltype(**p)>*>(b + off);
            off += (count * elem + 255) / 256 * 256;
        };
        Stash dummy;
        Stash* t = q ? q : &dummy;
        take(&t->enc, (size_t)P * 64, 4); take(&t->gd, (size_t)P * 32, 4);
        take(&t->h, (size_t)D * P * W, 4); take(&t->feat, (size_t)P * W, 4); take(&t->v, (size_t)P * (W / 2), 4);
        take(&t->ha, (size_t)P * c.h_alpha_size, 4); take(&t->hr, (size_t)P * c.h_rgb_size, 4);
        take(&t->theta, (size_t)P * kThetaAll, 4); take(&t->z, (size_t)P + 1, 4); take(&t->raw, (size_t)tiles * kTileM * k * 4, 4);
        take(&t->rays, (size_t)n * 11, 4); take(&t->at, (size_t)tiles * kTileM * k * 2, 4);
        take(&t->mbits, (size_t)(D + 1) * tiles * (W / 32) * 64, 4);
        take(&t->gms, (size_t)(std::max<int64_t>(n * kTailParts, tiles) + 8) * 8, 4);      // rays * k-parts (fused tail) or waves of points (flows_bwd)
        take(&t->g_theta, (size_t)P * kThetaAll, 4); take(&t->g_hr, (size_t)P * c.h_rgb_size, 4);      // ONE row per point whatever the k-parts
        take(&t->g_ha, (size_t)P * c.h_alpha_size, 4); take(&t->g_v, (size_t)P * (W / 2), 4);
        take(&t->g_feat, (size_t)P * W, 4); take(&t->g_h, (size_t)D * P * W, 4);
        take(&t->dbp, (size_t)n_wg * nb, 4);
        take(&t->partials, (size_t)kDwSlots * n_params, 4);
        take(&t->zeros, 64, 4);
        take(&t->bias_maps, (size_t)n_bias_maps, sizeof(BiasMap));
        take(&t->tiles, kMaxDwTiles, sizeof(DwTile)); take(&t->tiles_small, kMaxDwTiles, sizeof(DwTile));
        take(&t->blocks, kMaxDwBlocks, sizeof(DwBlock)); take(&t->blocks_small, kMaxDwBlocks, sizeof(DwBlock));
        take(&t->segs, 256, sizeof(RedSeg));
        return off;
    }

    void release() {
        if (owned && base) (void)hipFree(base);
        base = nullptr; cap = 0; owned = false; valid = false; bound_N = -1; used = 0;
        ++bind_serial;
    }
};

}  // namespace cfnerf

struct cfnerf_model {
    cfnerf_cfg cfg{};
    cfnerf::ParamLayout layout;
    cfnerf::PackPlan plan;
    int device = 0, n_cu = 256;
    float* d_packed = nullptr;
    void* d_packed16 = nullptr;           // split-bf16 operand copies (bf16x3 mode)
    int precision = 0;                    // 0 = fp32 MFMA (default), 1 = bf16x3 split MFMA in the forward
    int flow_math = 0;                    // 0 = auto (libm below 16 latents, hardware transcendentals from there), 1 = libm, 2 = hardware
    cfnerf::PackDesc* d_descs = nullptr;
    uint32_t* d_pack_table = nullptr;      // [plan.total_elems][3]: source, destination, bf16 destination of every copied element
    const float* flat = nullptr;          // caller-owned flat parameter buffer (last set_params)
    float* d_ent_partials = nullptr; int ent_cap = 0;
    float* d_enc_scratch = nullptr;       // [ent_cap, 64*64] forward scratch: one encoded tile per resident workgroup (8 MB, L2-resident)
    float* d_eps = nullptr;               // eps of the stashed forward
    cfnerf::Stash stash;
    cfnerf::BwdPlan bwd;
    int fwd_blocks_per_cu = 1;            // occupancy of the fused forward on THIS device (set at create)
    int timing = 0;                // 0 off, 1 every stage, 2 the fused forward only (two events per step instead of ten)
    hipEvent_t ev0[cfnerf::kNumTimers]{}, ev1[cfnerf::kNumTimers]{};
    hipEvent_t fr0[cfnerf::kFwdRing]{}, fr1[cfnerf::kFwdRing]{};   // ring of forward-launch event pairs
    uint64_t fwd_launches = 0;     // timed forward launches so far (ring position = count % kFwdRing)
    size_t ws_bytes = 0;
};

namespace cfnerf {
// (re)bind the stash pointers for an (N,S,K) batch; grows a model-owned block, rejects a caller block that is too small
int stash_bind(cfnerf_model* m, int64_t n, int s, int k, char* err, size_t errlen);
int ensure_bwd_plan(cfnerf_model* m);
size_t workspace_bytes_for(const cfnerf_cfg& c, int64_t n, int s, int k);
hipError_t bwd_set_attributes(int W, int ha);             // dynamic-LDS limits of the backward kernels, per device
}  // namespace cfnerf
