// cfnerf_model.h - the opaque cfnerf_model handle: packed weights, operand table, activation stash.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "cfnerf_layout.h"
#include "cfnerf_bwd.h"

namespace cfnerf {

constexpr int kNumTimers = 5;   // 0 fwd, 1 bwd_tail, 2 bwd_data, 3 bwd_dw, 4 adam

// Activations kept by a CFNERF_F_STASH forward for cfnerf_render_bwd.  Row-major per point.
// Sized for 288 GB of HBM: ~11 KB per point at W = 256 (1.4 GB for a 1024-ray x 128-sample batch).
struct Stash {
    float *enc = nullptr;    // [P,64]   gamma(p) (padded)
    float *gd = nullptr;     // [P,32]   gamma(d) (padded)
    float *h = nullptr;      // [D,P,W]  trunk activations (post-ReLU)
    float *feat = nullptr;   // [P,W]
    float *v = nullptr;      // [P,W/2]  views layer (post-ReLU)
    float *ha = nullptr;     // [P,HA]
    float *hr = nullptr;     // [P,HR]
    float *theta = nullptr;  // [P,128]  flow parameters (diagonals tanh-ed)
    float *z = nullptr;      // [P]      z_vals
    float *raw = nullptr;    // [P,K,4]  (used when the caller did not ask for raw)
    float *rays = nullptr;   // [N,11]
    float *at = nullptr;     // [P,K,2] alpha, T
    float *dbp = nullptr;    // [n_wg, NB] per-workgroup bias-gradient partials
    float *gms = nullptr;    // [n_waves, 8] base-Gaussian gradient partials
    float *mbits = nullptr;  // [D+1][tiles][W/32][64] u32 ReLU mask words (fragment order)
    int64_t n_tiles = 0;
    // backward workspaces (same row-major-per-point convention)
    float *g_theta = nullptr;   // [P,128]  d loss / d theta (pre-tanh for the diagonal columns)
    float *g_hr = nullptr;      // [P,HR]
    float *g_ha = nullptr;      // [P,HA]
    float *g_v = nullptr;       // [P,W/2]  pre-activation gradient of the views layer
    float *g_feat = nullptr;    // [P,W]
    float *g_h = nullptr;       // [D,P,W]  pre-activation gradients of the trunk layers
    const float* raw_used = nullptr;
    int64_t cap_P = 0, cap_N = 0; int cap_K = 0;
    int64_t N = 0; int S = 0, K = 0, flags = 0;
    bool valid = false;
    size_t bytes = 0;

    void release() {
        float** all[] = {&enc, &gd, &h, &feat, &v, &ha, &hr, &theta, &z, &raw, &rays, &at, &dbp, &gms, &mbits,
                         &g_theta, &g_hr, &g_ha, &g_v, &g_feat, &g_h};
        for (float** p : all) { if (*p) hipFree(*p); *p = nullptr; }
        cap_P = cap_N = 0; cap_K = 0; bytes = 0; valid = false;
    }

    int ensure(const cfnerf_cfg& c, int64_t n, int s, int k) {
        const int64_t P = n * (int64_t)s;
        if (P <= cap_P && n <= cap_N && k <= cap_K) return 0;
        hipDeviceSynchronize();
        release();
        const int W = c.netwidth, D = c.netdepth;
        size_t total = 0;
        auto al = [&](float** p, size_t nfloat) {
            if (hipMalloc(p, nfloat * sizeof(float)) != hipSuccess) return false;
            total += nfloat * sizeof(float);
            return true;
        };
        bool ok = al(&enc, (size_t)P * 64) && al(&gd, (size_t)P * 32) && al(&h, (size_t)D * P * W) && al(&feat, (size_t)P * W) &&
                  al(&v, (size_t)P * (W / 2)) && al(&ha, (size_t)P * c.h_alpha_size) && al(&hr, (size_t)P * c.h_rgb_size) &&
                  al(&theta, (size_t)P * kThetaAll) && al(&z, (size_t)P) && al(&raw, (size_t)P * k * 4) &&
                  al(&rays, (size_t)n * 11) && al(&at, (size_t)P * k * 2) && al(&gms, (size_t)(n + 8) * 8) && al(&mbits, (size_t)(D + 1) * (n * ((s + 63) / 64)) * (W / 32) * 64) && al(&g_theta, (size_t)P * kThetaAll) && al(&g_hr, (size_t)P * c.h_rgb_size) &&
                  al(&g_ha, (size_t)P * c.h_alpha_size) && al(&g_v, (size_t)P * (W / 2)) && al(&g_feat, (size_t)P * W) &&
                  al(&g_h, (size_t)D * P * W);
        if (!ok) { release(); return CFNERF_E_NOMEM; }
        cap_P = P; cap_N = n; cap_K = k; bytes = total;
        return 0;
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
    cfnerf::NetTab* d_tab = nullptr;
    cfnerf::PackDesc* d_descs = nullptr;
    const float* flat = nullptr;          // caller-owned flat parameter buffer (last set_params)
    float* d_ent_partials = nullptr; int ent_cap = 0;
    float* d_eps = nullptr;               // eps of the stashed forward
    cfnerf::Stash stash;
    cfnerf::BwdPlan bwd;
    bool timing = false;
    hipEvent_t ev0[cfnerf::kNumTimers]{}, ev1[cfnerf::kNumTimers]{};
    size_t ws_bytes = 0;
};
