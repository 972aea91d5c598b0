// cfnerf_abi.hip - the extern "C" boundary of libcfnerf_hip.so (see include/cfnerf.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "cfnerf_kernels.h"
#include "cfnerf_model.h"

using namespace cfnerf;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(CFNERF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" void cfnerf_set_error_(const char* msg) {      // used by the other translation units; hidden (not exported)
    std::snprintf(g_err, sizeof g_err, "%s", msg);
}

// device-side part of cfnerf_model_create; on failure the caller destroys the partially built handle
static int model_init_device(cfnerf_model* m) {
    const cfnerf_cfg* cfg = &m->cfg;
    HIPCHK(hipGetDevice(&m->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, m->device));
    m->n_cu = prop.multiProcessorCount;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CFNERF_E_UNSUPPORTED, "device is %s; this library is built for gfx950 (MI355X) only", prop.gcnArchName);
    const size_t pbytes = (size_t)m->plan.tab.packed_floats * sizeof(float);
    HIPCHK(hipMalloc(&m->d_packed, pbytes));
    HIPCHK(hipMemset(m->d_packed, 0, pbytes));
    HIPCHK(hipMalloc(&m->d_packed16, (size_t)m->plan.tab.packed16_elems * 2));
    HIPCHK(hipMemset(m->d_packed16, 0, (size_t)m->plan.tab.packed16_elems * 2));
    HIPCHK(hipMalloc(&m->d_descs, m->plan.descs.size() * sizeof(PackDesc)));
    HIPCHK(hipMemcpy(m->d_descs, m->plan.descs.data(), m->plan.descs.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&m->d_pack_table, (size_t)m->plan.total_elems * 3 * sizeof(uint32_t)));
    HIPCHK(launch_pack_index(m->d_descs, (int)m->plan.descs.size(), m->plan.total_elems, m->d_pack_table, nullptr));
    HIPCHK(hipDeviceSynchronize());
    m->ent_cap = fused_fwd_max_grid(cfg->netwidth, cfg->h_alpha_size, m->n_cu);
    HIPCHK(hipMalloc(&m->d_ent_partials, (size_t)m->ent_cap * 2 * sizeof(float)));
    HIPCHK(hipMalloc(&m->d_enc_scratch, (size_t)m->ent_cap * kTileM * 64 * sizeof(float)));
    HIPCHK(hipMalloc(&m->d_eps, kMaxK * 4 * sizeof(float)));
    for (int i = 0; i < kFwdRing; ++i) {
        HIPCHK(hipEventCreate(&m->fr0[i]));
        HIPCHK(hipEventCreate(&m->fr1[i]));
    }
    for (int i = 0; i < kNumTimers; ++i) {
        HIPCHK(hipEventCreate(&m->ev0[i]));
        HIPCHK(hipEventCreate(&m->ev1[i]));
    }
    m->ws_bytes = pbytes + sizeof(NetTab);
    // launch attributes and occupancy belong to (kernel, device): set here, with the model's device current
    HIPCHK(fused_fwd_set_attributes(cfg->netwidth, cfg->h_alpha_size, &m->fwd_blocks_per_cu));
    HIPCHK(bwd_set_attributes(cfg->netwidth, cfg->h_alpha_size));
    return CFNERF_OK;
}

size_t cfnerf::workspace_bytes_for(const cfnerf_cfg& c, int64_t n, int s, int k) {
    return Stash::carve(nullptr, nullptr, c, n, s, k, build_layout(c).total);
}

// Point the stash at its block for an (N,S,K) batch.  A caller-provided block (cfnerf_model_set_workspace) is never
// grown: too small is an error.  A model-owned block is (re)allocated here - the only place the train path allocates.
int cfnerf::stash_bind(cfnerf_model* m, int64_t n, int s, int k, char* err, size_t errlen) {
    Stash& q = m->stash;
    if (q.base && q.bound_N == n && q.bound_S == s && q.bound_K == k) return CFNERF_OK;
    const size_t need = Stash::carve(nullptr, nullptr, m->cfg, n, s, k, m->layout.total);
    if (need > q.cap) {
        if (q.base && !q.owned) {
            std::snprintf(err, errlen, "the workspace handed to cfnerf_model_set_workspace holds %zu bytes but N=%lld S=%d K=%d needs %zu "
                          "(size it with cfnerf_workspace_bytes)", q.cap, (long long)n, s, k, need);
            return CFNERF_E_NOMEM;
        }
        if (hipDeviceSynchronize() != hipSuccess) { std::snprintf(err, errlen, "hipDeviceSynchronize failed"); return CFNERF_E_HIP; }
        q.release();
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) {
            std::snprintf(err, errlen, "workspace allocation of %zu bytes failed (N=%lld S=%d K=%d)", need, (long long)n, s, k);
            return CFNERF_E_NOMEM;
        }
        q.base = static_cast<char*>(p); q.cap = need; q.owned = true;
    }
    q.used = Stash::carve(&q, q.base, m->cfg, n, s, k, m->layout.total);
    q.bound_N = n; q.bound_S = s; q.bound_K = k;
    q.valid = false;
    ++q.bind_serial;
    return CFNERF_OK;
}

// a handle lives on the device that was current when it was created (one handle per device, one process per GPU)
static int check_device(const cfnerf_model* m) {
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev != m->device)
        return fail(CFNERF_E_INVALID, "model lives on device %d but the current device is %d (hipSetDevice / torch.cuda.set_device first)",
                    m->device, dev);
    return CFNERF_OK;
}

extern "C" int cfnerf_model_destroy(cfnerf_model* m);

extern "C" {

int cfnerf_version(void) { return 100; }
const char* cfnerf_last_error(void) { return g_err; }

int64_t cfnerf_param_count(const cfnerf_cfg* cfg) {
    if (!cfg) { fail(CFNERF_E_INVALID, "cfg is NULL"); return -1; }
    if (const char* why = validate_cfg(*cfg)) { fail(CFNERF_E_UNSUPPORTED, "%s", why); return -1; }
    return build_layout(*cfg).total;
}

int64_t cfnerf_param_offset(const cfnerf_cfg* cfg, const char* key, int64_t* numel) {
    if (!cfg || !key) { fail(CFNERF_E_INVALID, "NULL argument"); return -1; }
    if (const char* why = validate_cfg(*cfg)) { fail(CFNERF_E_UNSUPPORTED, "%s", why); return -1; }
    ParamLayout L = build_layout(*cfg);
    const ParamEntry* e = L.find(key);
    if (!e) { fail(CFNERF_E_INVALID, "unknown parameter key '%s'", key); return -1; }
    if (numel) *numel = e->numel();
    return e->off;
}

const char* cfnerf_param_key(const cfnerf_cfg* cfg, int index) {
    static thread_local std::string key;
    if (!cfg || validate_cfg(*cfg)) return nullptr;
    ParamLayout L = build_layout(*cfg);
    if (index < 0 || index >= (int)L.e.size()) return nullptr;
    key = L.e[index].key;
    return key.c_str();
}

int cfnerf_model_create(const cfnerf_cfg* cfg, cfnerf_model** out) {
    if (!cfg || !out) return fail(CFNERF_E_INVALID, "NULL argument");
    if (const char* why = validate_cfg(*cfg)) return fail(CFNERF_E_UNSUPPORTED, "%s", why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(CFNERF_E_HIP, "no HIP device visible: the CF-NeRF hot path has no CPU fallback");
    cfnerf_model* m = new (std::nothrow) cfnerf_model();
    if (!m) return fail(CFNERF_E_NOMEM, "host allocation failed");
    m->cfg = *cfg;
    m->layout = build_layout(*cfg);
    m->plan = build_pack_plan(*cfg, m->layout);
    const int rc = model_init_device(m);
    if (rc != CFNERF_OK) {                 // g_err already says why; release whatever was allocated
        cfnerf_model_destroy(m);
        return rc;
    }
    *out = m;
    return CFNERF_OK;
}

int cfnerf_model_destroy(cfnerf_model* m) {
    if (!m) return CFNERF_OK;
    hipDeviceSynchronize();
    hipFree(m->d_packed); hipFree(m->d_packed16); hipFree(m->d_descs); hipFree(m->d_pack_table); hipFree(m->d_ent_partials);
    hipFree(m->d_enc_scratch);
    hipFree(m->d_eps);
    m->stash.release();
    m->bwd.release();
    for (int i = 0; i < kNumTimers; ++i) { if (m->ev0[i]) hipEventDestroy(m->ev0[i]); if (m->ev1[i]) hipEventDestroy(m->ev1[i]); }
    for (int i = 0; i < kFwdRing; ++i) { if (m->fr0[i]) hipEventDestroy(m->fr0[i]); if (m->fr1[i]) hipEventDestroy(m->fr1[i]); }
    delete m;
    return CFNERF_OK;
}

int cfnerf_model_set_params(cfnerf_model* m, const float* flat_params, cfnerf_stream s) {
    if (!m || !flat_params) return fail(CFNERF_E_INVALID, "NULL argument");
    if (int rc = check_device(m)) return rc;
    m->flat = flat_params;
    // the split-bf16 copy is only refreshed while the opt-in mode is selected
    HIPCHK(launch_pack(flat_params, m->d_packed, m->precision ? m->d_packed16 : nullptr, m->d_pack_table, m->plan.total_elems, (hipStream_t)s));
    return CFNERF_OK;
}

int cfnerf_rays_setup(int H, int W, float focal, const float* c2w_host, const float* rays_o, const float* rays_d,
                      int64_t N, int64_t pixel0, int ndc, float near_, float far_, float* rays, cfnerf_stream s) {
    if (!rays || N < 0) return fail(CFNERF_E_INVALID, "bad rays/N");
    if (N == 0) return CFNERF_OK;
    RaysC2W c{};
    if (c2w_host) {
        if (pixel0 < 0 || pixel0 + N > (int64_t)H * W)
            return fail(CFNERF_E_INVALID, "c2w given: pixels [%lld, %lld) exceed H*W = %lld", (long long)pixel0, (long long)(pixel0 + N), (long long)H * W);
        std::memcpy(c.m, c2w_host, sizeof c.m);
    } else if (!rays_o || !rays_d) {
        return fail(CFNERF_E_INVALID, "either c2w_host or rays_o/rays_d must be given");
    }
    HIPCHK(launch_rays_setup(H, W, focal, c, c2w_host ? 1 : 0, rays_o, rays_d, N, pixel0, ndc, near_, far_, rays, (hipStream_t)s));
    return CFNERF_OK;
}

int cfnerf_ndc_rays(int H, int W, float focal, float near_, const float* rays_o, const float* rays_d, int64_t N, float* out_o, float* out_d,
                    cfnerf_stream s) {
    if (N < 0 || H < 1 || W < 1) return fail(CFNERF_E_INVALID, "bad N/H/W");
    if (N == 0) return CFNERF_OK;
    if (!rays_o || !rays_d || !out_o || !out_d) return fail(CFNERF_E_INVALID, "NULL argument");
    HIPCHK(launch_ndc_rays(H, W, focal, near_, rays_o, rays_d, N, out_o, out_d, (hipStream_t)s));
    return CFNERF_OK;
}

int cfnerf_embed(const float* x, int64_t P, int multires, float* out, cfnerf_stream s) {
    if (P < 0 || multires < 1 || multires > 16) return fail(CFNERF_E_INVALID, "bad P / multires (1..16)");
    if (P == 0) return CFNERF_OK;
    if (!x || !out) return fail(CFNERF_E_INVALID, "NULL argument");
    HIPCHK(launch_embed(x, P, multires, out, (hipStream_t)s));
    return CFNERF_OK;
}

int cfnerf_sample_points(const float* rays, const float* t_vals, const float* t_rand, int flags, int64_t N, int S, float* z_vals,
                         float* pts, cfnerf_stream s) {
    if (N < 0 || S < 1) return fail(CFNERF_E_INVALID, "bad N/S");
    if (N == 0) return CFNERF_OK;
    if (!rays || !t_vals || !z_vals) return fail(CFNERF_E_INVALID, "NULL argument");
    HIPCHK(launch_sample_points(rays, t_vals, t_rand, flags, N, S, z_vals, pts, (hipStream_t)s));
    return CFNERF_OK;
}

static int flow_math_bits(const cfnerf_model* m) {
    return m->flow_math == 0 ? 0 : (CFNERF_F_FLOW_MATH_SET | (m->flow_math == 2 ? CFNERF_F_FLOW_MATH_FAST : 0));
}

static int check_common(cfnerf_model* m, int K) {
    if (!m) return fail(CFNERF_E_INVALID, "model is NULL");
    if (!m->flat) return fail(CFNERF_E_INVALID, "cfnerf_model_set_params has not been called");
    if (int rc = check_device(m)) return rc;
    if (K < 1 || K > kMaxK) return fail(CFNERF_E_UNSUPPORTED, "K_samples must be in [1,%d], got %d", kMaxK, K);
    return CFNERF_OK;
}

int cfnerf_render_fwd(cfnerf_model* m, const float* rays, const float* t_vals, const float* t_rand, const float* z_vals_opt,
                      const float* eps, int64_t N, int S, int K, int flags, float* rgb_map, float* disp_map, float* depth_map,
                      float* raw_opt, float* weights_opt, float* pts_opt, float* kstats_opt, float* entropy_out, cfnerf_stream s) {
    if (int rc = check_common(m, K)) return rc;
    if (N < 0 || S < 1) return fail(CFNERF_E_INVALID, "bad N/S");
    if (N == 0) return CFNERF_OK;            // empty batch: nothing to do (buffers may be NULL)
    if (!rays || !eps || (!t_vals && !z_vals_opt)) return fail(CFNERF_E_INVALID, "NULL argument");
    const bool maps = rgb_map && disp_map && depth_map;
    if (!maps && (rgb_map || disp_map || depth_map)) return fail(CFNERF_E_INVALID, "rgb_map/disp_map/depth_map must be given together");
    if (!maps && !kstats_opt) return fail(CFNERF_E_INVALID, "either the per-K maps or kstats_opt must be requested");
    if (kstats_opt && K < 2) return fail(CFNERF_E_INVALID, "kstats needs K >= 2 (std * n/(n-1))");
    hipStream_t st = (hipStream_t)s;
    if (flags & CFNERF_F_STASH) flags |= CFNERF_F_TRAIN;
    const bool train = flags & CFNERF_F_TRAIN;
    if (train && !entropy_out) return fail(CFNERF_E_INVALID, "TRAIN needs entropy_out");
    FwdArgs a{};
    a.wp = m->d_packed; a.wp16 = m->d_packed16; a.flat = m->flat;
    a.rays = rays; a.t_vals = t_vals; a.t_rand = z_vals_opt ? nullptr : t_rand; a.z_in = z_vals_opt; a.eps = eps;
    a.N = N; a.S = S; a.K = K; a.P = N * (int64_t)S; a.flags = (flags & 0xffff) | flow_math_bits(m);
    a.rgb_map = rgb_map; a.disp = disp_map; a.depth = depth_map;
    a.raw = raw_opt; a.weights = weights_opt; a.pts = pts_opt; a.kstats = kstats_opt;
    a.ent_partials = train ? m->d_ent_partials : nullptr;
    a.enc_scratch = m->d_enc_scratch;
    if ((flags & CFNERF_F_STASH) && !maps) return fail(CFNERF_E_INVALID, "STASH needs the per-K maps");
    if (flags & CFNERF_F_STASH) {
        char why[256];
        if (int rc = stash_bind(m, N, S, K, why, sizeof why)) return fail(rc, "%s", why);
        Stash& q = m->stash;
        a.st_enc = q.enc; a.st_gd = q.gd; a.st_h = q.h; a.st_feat = q.feat; a.st_v = q.v; a.st_ha = q.ha; a.st_hr = q.hr;
        a.st_theta = q.theta; a.st_z = q.z; a.st_at = q.at;
        a.st_mbits = reinterpret_cast<uint32_t*>(q.mbits);
        q.n_tiles = N * (int64_t)((S + kTileM - 1) / kTileM);
        a.n_tiles = q.n_tiles;
        a.st_raw = q.raw;                    // the backward reads the model's OWN (tile-transposed) copy: the caller may drop its tensor,
                                             // and a caller's raw_opt is written by the same launch (rounds 1-4: a device-to-device copy after it)
        // (q.rays and m->d_eps, the backward's own copies of the step's rays and latents, are written by the copy blocks of the
        //  entropy_finalize launch below - the forward itself reads the caller's)
        q.N = N; q.S = S; q.K = K; q.flags = flags; q.valid = true; q.points = false;
        q.q4 = (S % kTileM == 0) && m->precision == 0;      // whole tiles, fp32 mode: the wide streams take the Q4 layout (cfnerf_device.h)
        a.q4 = q.q4;
        ++q.generation;                      // this forward now owns the one stash: older backward passes are refused
    }
    int grid = 0;
    if (m->timing) HIPCHK(hipEventRecord(m->fr0[m->fwd_launches % kFwdRing], st));
    HIPCHK(launch_fused_fwd(a, m->plan.tab, 0, train, m->precision, m->n_cu, m->fwd_blocks_per_cu, st, &grid));
    if (m->timing) { HIPCHK(hipEventRecord(m->fr1[m->fwd_launches % kFwdRing], st)); ++m->fwd_launches; }
    if (train) {
        const bool keep = flags & CFNERF_F_STASH;
        HIPCHK(launch_entropy_finalize(m->d_ent_partials, grid, m->flat, eps, K, (double)a.P * K, entropy_out, keep ? m->d_eps : nullptr, rays,
                                       keep ? m->stash.rays : nullptr, N * 11, st));
    }
    return CFNERF_OK;
}

int cfnerf_render_eval(cfnerf_model* m, const float* rays, const float* t_vals, const float* eps, int64_t N, int S, int K, int flags,
                       const float* gt_opt, float* kstats, float* sqerr_opt, cfnerf_stream s) {
    if (int rc = check_common(m, K)) return rc;
    if (N < 0 || S < 1) return fail(CFNERF_E_INVALID, "bad N/S");
    if (N == 0) return CFNERF_OK;
    if (!rays || !eps || !t_vals || !kstats) return fail(CFNERF_E_INVALID, "NULL argument");
    if (K < 2) return fail(CFNERF_E_INVALID, "kstats needs K >= 2 (std * n/(n-1))");
    if ((gt_opt == nullptr) != (sqerr_opt == nullptr)) return fail(CFNERF_E_INVALID, "gt_opt and sqerr_opt must be given together");
    if (flags & (CFNERF_F_TRAIN | CFNERF_F_STASH)) return fail(CFNERF_E_INVALID, "cfnerf_render_eval is the eval branch only");
    FwdArgs a{};
    a.wp = m->d_packed; a.wp16 = m->d_packed16; a.flat = m->flat;
    a.rays = rays; a.t_vals = t_vals; a.eps = eps;
    a.N = N; a.S = S; a.K = K; a.P = N * (int64_t)S; a.flags = (flags & 0xffff) | flow_math_bits(m);
    a.kstats = kstats; a.gt = gt_opt; a.sqerr = sqerr_opt; a.enc_scratch = m->d_enc_scratch;
    int grid = 0;
    hipStream_t st = (hipStream_t)s;
    if (m->timing) HIPCHK(hipEventRecord(m->fr0[m->fwd_launches % kFwdRing], st));
    HIPCHK(launch_fused_fwd(a, m->plan.tab, 0, false, m->precision, m->n_cu, m->fwd_blocks_per_cu, st, &grid));
    if (m->timing) { HIPCHK(hipEventRecord(m->fr1[m->fwd_launches % kFwdRing], st)); ++m->fwd_launches; }
    return CFNERF_OK;
}

int cfnerf_sample_pdf(const float* rays, const float* t_vals, const float* t_rand, int flags, const float* weights, const float* u,
                      int64_t N, int S, int K, int N_importance, float* z_out, cfnerf_stream s) {
    if (N < 0 || S < 3 || K < 1 || N_importance < 1) return fail(CFNERF_E_INVALID, "bad N/S/K/N_importance (S >= 3)");
    if (S + N_importance > 1024) return fail(CFNERF_E_UNSUPPORTED, "S + N_importance must be <= 1024");
    if (N == 0) return CFNERF_OK;
    if (!rays || !t_vals || !weights || !u || !z_out) return fail(CFNERF_E_INVALID, "NULL argument");
    HIPCHK(launch_sample_pdf(rays, t_vals, t_rand, flags, weights, u, N, S, K, N_importance, z_out, (hipStream_t)s));
    return CFNERF_OK;
}

int cfnerf_network_fwd(cfnerf_model* m, const float* x, const float* eps, int64_t P, int K, int flags, float* raw,
                       float* entropy_out, cfnerf_stream s) {
    if (int rc = check_common(m, K)) return rc;
    if (P < 0 || P > 0x7fffffff) return fail(CFNERF_E_INVALID, "bad P");
    if (P == 0) return CFNERF_OK;
    if (!x || !eps || !raw) return fail(CFNERF_E_INVALID, "NULL argument");
    if (flags & CFNERF_F_STASH) flags |= CFNERF_F_TRAIN;
    const bool train = flags & CFNERF_F_TRAIN;
    if (train && !entropy_out) return fail(CFNERF_E_INVALID, "TRAIN needs entropy_out");
    hipStream_t st = (hipStream_t)s;
    FwdArgs a{};
    a.wp = m->d_packed; a.wp16 = m->d_packed16; a.flat = m->flat;
    a.eps = eps; a.x = x; a.P = P; a.N = 0; a.S = 1; a.K = K; a.flags = (flags & 0xffff) | flow_math_bits(m); a.raw = raw;
    a.ent_partials = train ? m->d_ent_partials : nullptr;
    if (flags & CFNERF_F_STASH) {            // points-mode stash: the workspace is bound as ONE "ray" of P samples
        char why[256];
        if (int rc = stash_bind(m, 1, (int)P, K, why, sizeof why)) return fail(rc, "%s", why);
        Stash& q = m->stash;
        a.st_enc = q.enc; a.st_gd = q.gd; a.st_h = q.h; a.st_feat = q.feat; a.st_v = q.v; a.st_ha = q.ha; a.st_hr = q.hr;
        a.st_theta = q.theta;
        a.st_mbits = reinterpret_cast<uint32_t*>(q.mbits);
        q.n_tiles = (P + kTileM - 1) / kTileM;
        a.n_tiles = q.n_tiles;
        a.st_raw = q.raw;                    // the backward reads the model's OWN (tile-transposed) copy; the caller's raw is written by the same launch
        if (!train) HIPCHK(hipMemcpyAsync(m->d_eps, eps, (size_t)K * 4 * sizeof(float), hipMemcpyDeviceToDevice, st));   // (else: entropy_finalize keeps them)
        q.N = 1; q.S = (int)P; q.K = K; q.flags = flags; q.valid = true; q.points = true;
        q.q4 = (P % kTileM == 0) && m->precision == 0;
        a.q4 = q.q4;
        ++q.generation;
    }
    int grid = 0;
    HIPCHK(launch_fused_fwd(a, m->plan.tab, 1, train, m->precision, m->n_cu, m->fwd_blocks_per_cu, st, &grid));
    if (train)
        HIPCHK(launch_entropy_finalize(m->d_ent_partials, grid, m->flat, eps, K, (double)P * K, entropy_out,
                                       (flags & CFNERF_F_STASH) ? m->d_eps : nullptr, nullptr, nullptr, 0, st));
    return CFNERF_OK;
}

int cfnerf_composite_fwd(const float* raw, const float* z_vals, const float* rays_d, int64_t N, int S, int K,
                         int white_bkgd, float* rgb_map, float* disp_map, float* depth_map, float* weights_opt,
                         cfnerf_stream s) {
    if (N < 0 || S < 1 || K < 1) return fail(CFNERF_E_INVALID, "bad N/S/K");
    if (N == 0) return CFNERF_OK;
    if (!raw || !z_vals || !rays_d || !rgb_map || !disp_map || !depth_map) return fail(CFNERF_E_INVALID, "NULL argument");
    if ((int64_t)S * K * 16 >= (1ll << 31)) return fail(CFNERF_E_UNSUPPORTED, "cfnerf_composite_fwd: S * K * 16 bytes per ray must stay below 2 GiB");
    HIPCHK(launch_composite(raw, z_vals, rays_d, N, S, K, white_bkgd, rgb_map, disp_map, depth_map, weights_opt, (hipStream_t)s));
    return CFNERF_OK;
}

int cfnerf_model_set_precision(cfnerf_model* m, int mode) {
    if (!m) return fail(CFNERF_E_INVALID, "model is NULL");
    if (mode != 0 && mode != 1) return fail(CFNERF_E_INVALID, "precision mode must be 0 (fp32 MFMA) or 1 (bf16x3 split MFMA)");
    const bool need_pack16 = mode != 0 && m->precision == 0 && m->flat != nullptr;
    if (mode != m->precision && m->stash.valid) {      // a stashed forward belongs to the mode it ran in (operand copies, the layout of its wide streams): its
        m->stash.valid = false;                         // backward in the other mode would read it wrongly - dropped like a re-bound workspace
        ++m->stash.generation;
    }
    m->precision = mode;
    if (need_pack16) {      // bring the bf16 copy up to date with the current parameters (rare call: fully synchronous)
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(launch_pack(m->flat, m->d_packed, m->d_packed16, m->d_pack_table, m->plan.total_elems, nullptr));
        HIPCHK(hipDeviceSynchronize());
    }
    return CFNERF_OK;
}

int cfnerf_model_set_flow_math(cfnerf_model* m, int mode) {
    if (!m) return fail(CFNERF_E_INVALID, "model is NULL");
    if (mode < 0 || mode > 2) return fail(CFNERF_E_INVALID, "flow math mode must be 0 (auto), 1 (libm) or 2 (hardware transcendentals)");
    m->flow_math = mode;
    return CFNERF_OK;
}

int64_t cfnerf_model_workspace_bytes(const cfnerf_model* m) { return m ? (int64_t)(m->ws_bytes + m->stash.cap) : 0; }

int64_t cfnerf_workspace_bytes(const cfnerf_cfg* cfg, int64_t N, int S, int K) {
    if (!cfg) { fail(CFNERF_E_INVALID, "cfg is NULL"); return -1; }
    if (const char* why = validate_cfg(*cfg)) { fail(CFNERF_E_UNSUPPORTED, "%s", why); return -1; }
    if (K < 1 || K > kMaxK) { fail(CFNERF_E_UNSUPPORTED, "K_samples must be in [1,%d], got %d", kMaxK, K); return -1; }
    if (N < 0 || S < 1) { fail(CFNERF_E_INVALID, "bad N/S"); return -1; }
    return (int64_t)workspace_bytes_for(*cfg, N, S, K);
}

int cfnerf_model_set_workspace(cfnerf_model* m, void* workspace, size_t bytes) {
    if (!m) return fail(CFNERF_E_INVALID, "model is NULL");
    if ((workspace == nullptr) != (bytes == 0)) return fail(CFNERF_E_INVALID, "workspace and bytes must both be given or both be 0");
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(CFNERF_E_INVALID, "workspace must be 256-byte aligned");
    if (int rc = check_device(m)) return rc;
    Stash& q = m->stash;
    if (q.owned && q.base) HIPCHK(hipDeviceSynchronize());      // kernels may still use the block about to be freed
    q.release();                                                // frees a model-owned block; forgets a caller-owned one
    if (workspace) { q.base = static_cast<char*>(workspace); q.cap = bytes; q.owned = false; }
    return CFNERF_OK;
}

uint64_t cfnerf_model_stash_generation(const cfnerf_model* m) { return (m && m->stash.valid) ? m->stash.generation : 0; }

int cfnerf_timing_enable(cfnerf_model* m, int mode) {
    if (!m) return fail(CFNERF_E_INVALID, "model is NULL");
    if (mode < 0 || mode > 2) return fail(CFNERF_E_INVALID, "timing mode must be 0 (off), 1 (every stage) or 2 (fused forward only)");
    m->timing = mode;
    m->fwd_launches = 0;
    return CFNERF_OK;
}

float cfnerf_timing_fwd_mean_ms(cfnerf_model* m, int n) {
    if (!m || !m->timing || n < 1 || m->fwd_launches == 0) return -1.f;
    const uint64_t have = m->fwd_launches < (uint64_t)kFwdRing ? m->fwd_launches : (uint64_t)kFwdRing;
    const uint64_t take = (uint64_t)n < have ? (uint64_t)n : have;
    double sum = 0;
    for (uint64_t i = 0; i < take; ++i) {
        const int slot = (int)((m->fwd_launches - 1 - i) % kFwdRing);
        float ms = 0.f;
        if (hipEventSynchronize(m->fr1[slot]) != hipSuccess) return -1.f;
        if (hipEventElapsedTime(&ms, m->fr0[slot], m->fr1[slot]) != hipSuccess) return -1.f;
        sum += ms;
    }
    return (float)(sum / (double)take);
}

float cfnerf_timing_last_ms(cfnerf_model* m, int which) {
    if (!m || which < 0 || which >= kNumTimers || !m->timing) return -1.f;
    if (which == 0) return cfnerf_timing_fwd_mean_ms(m, 1);
    if (m->timing != 1) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(m->ev1[which]) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, m->ev0[which], m->ev1[which]) != hipSuccess) return -1.f;
    return ms;
}

}  // extern "C"

// ---- train-step entry points: implemented in cfnerf_bwd.hip -------------------------------------
