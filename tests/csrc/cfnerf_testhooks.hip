// cfnerf_testhooks.hip - TEST HOOKS (tests/cfnerf_debug.h), built into cf-nerf_amd/build/libcfnerf_testhooks.so by cf-nerf_amd/build.py.
// NOT part of the product: libcfnerf_hip.so exports include/cfnerf.h and nothing else; nothing under cf-nerf_amd/ loads this library.
// It is compiled against the library's INTERNAL headers, so it sees the same host-side planners (operand packing, weight-gradient
// plan: header-only code in cfnerf_layout.h / cfnerf_dwplan.h) and the layout of the opaque cfnerf_model handle: the stash read-back
// takes a handle that the PRODUCT library created and copies one of its workspace buffers - it launches nothing and changes nothing.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "cfnerf_dwplan.h"
#include "cfnerf_kernels.h"
#include "cfnerf_model.h"
#include "../cfnerf_debug.h"      // (tests/cfnerf_debug.h)

using namespace cfnerf;

extern "C" {
// host-side packing with the same index map the device kernel uses (CPU tests of the operand layout)
CFNERF_API int64_t cfnerf_debug_packed_floats(const cfnerf_cfg* cfg) {
    if (!cfg || validate_cfg(*cfg)) return -1;
    ParamLayout L = build_layout(*cfg);
    return build_pack_plan(*cfg, L).tab.packed_floats;
}
CFNERF_API int cfnerf_debug_pack_host(const cfnerf_cfg* cfg, const float* flat_host, float* packed_host) {
    if (!cfg || validate_cfg(*cfg)) return CFNERF_E_UNSUPPORTED;
    ParamLayout L = build_layout(*cfg);
    PackPlan P = build_pack_plan(*cfg, L);
    std::memset(packed_host, 0, (size_t)P.tab.packed_floats * sizeof(float));
    for (const PackDesc& d : P.descs) {
        const uint32_t n = d.n_cols ? d.n_rows * d.n_cols : d.n_rows;
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t src, dst;
            pack_map(d, i, &src, &dst);
            packed_host[dst] = flat_host[src];
        }
    }
    return CFNERF_OK;
}
// copy a stash / backward-workspace buffer of the last STASH forward into dst (device), for tests
CFNERF_API int64_t cfnerf_debug_copy_stash(cfnerf_model* m, const char* name, int layer, float* dst, int64_t max_floats, cfnerf_stream s) {
    if (!m || !m->stash.valid) return -1;
    Stash& q = m->stash;
    const int W = m->cfg.netwidth;
    const int64_t P = q.N * (int64_t)q.S;
    std::string n = name;
    const float* src = nullptr; int64_t cnt = 0;
    if (n == "h") { src = q.h + (size_t)layer * P * W; cnt = P * W; }
    else if (n == "g_h") { src = q.g_h + (size_t)layer * P * W; cnt = P * W; }
    else if (n == "feat") { src = q.feat; cnt = P * W; }
    else if (n == "g_feat") { src = q.g_feat; cnt = P * W; }
    else if (n == "v") { src = q.v; cnt = P * (W / 2); }
    else if (n == "g_v") { src = q.g_v; cnt = P * (W / 2); }
    else if (n == "ha") { src = q.ha; cnt = P * m->cfg.h_alpha_size; }
    else if (n == "g_ha") { src = q.g_ha; cnt = P * m->cfg.h_alpha_size; }
    else if (n == "hr") { src = q.hr; cnt = P * m->cfg.h_rgb_size; }
    else if (n == "g_hr") { src = q.g_hr; cnt = P * m->cfg.h_rgb_size; }
    else if (n == "theta") { src = q.theta; cnt = P * kThetaAll; }
    else if (n == "g_theta") { src = q.g_theta; cnt = P * kThetaAll; }
    else if (n == "enc") { src = q.enc; cnt = P * 64; }
    else if (n == "at") { src = q.at; cnt = q.n_tiles * kTileM * q.K * 2; }        // tile-transposed [tiles,K,64,2]: (e, T)
    else if (n == "raw") { src = q.raw; cnt = q.n_tiles * kTileM * q.K * 4; }      // tile-transposed [tiles,K,64,4]
    else if (n == "z") { src = q.z; cnt = P; }                                     // z_vals [P]
    else if (n == "rays") { src = q.rays; cnt = q.N * 11; }
    else return -1;
    if (cnt > max_floats) return -cnt;
    if (hipMemcpyAsync(dst, src, cnt * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess) return -1;
    return cnt;
}
CFNERF_API int cfnerf_debug_stash_q4(cfnerf_model* m) { return (!m || !m->stash.valid) ? -1 : (m->stash.q4 ? 1 : 0); }
// operand table entry by name: out[4] = {w_off, b_off, kc, nt}
CFNERF_API int cfnerf_debug_operand(const cfnerf_cfg* cfg, const char* name, int index, uint32_t* out) {
    if (!cfg || validate_cfg(*cfg)) return CFNERF_E_UNSUPPORTED;
    ParamLayout L = build_layout(*cfg);
    PackPlan P = build_pack_plan(*cfg, L);
    const NetTab& T = P.tab;
    const SubL* s = nullptr;
    std::string n = name;
    if (n == "trunk") s = &T.trunk[index]; else if (n == "skipseg") s = &T.skipseg; else if (n == "ha") s = &T.ha;
    else if (n == "ft") s = &T.ft; else if (n == "vf") s = &T.vf; else if (n == "vd") s = &T.vd; else if (n == "hr") s = &T.hr;
    else if (n == "fr") s = &T.fr; else if (n == "fa") s = &T.fa; else if (n == "bt_fr") s = &T.bt_fr;
    else if (n == "bt_fa") s = &T.bt_fa; else if (n == "bt_hr") s = &T.bt_hr; else if (n == "bt_vf") s = &T.bt_vf;
    else if (n == "bt_ft") s = &T.bt_ft; else if (n == "bt_ha") s = &T.bt_ha; else if (n == "bt_trunk") s = &T.bt_trunk[index];
    if (!s) return CFNERF_E_INVALID;
    out[0] = s->w_off; out[1] = s->b_off; out[2] = s->kc; out[3] = s->nt;
    return CFNERF_OK;
}

// ---- debug / test helper (not part of include/cfnerf.h): the weight-gradient tile plan of a configuration, for the
// CPU test that every weight element is covered exactly once.  20 int32 per tile:
// {is_big, n0, k0, N, K, gk, wk, nseg, seg_row[0..3], dst_ld, dst_col, row_f, late, lay, 0, 0, 0} followed by 4 uint32 seg_dst in a second array.
CFNERF_API int cfnerf_debug_dw_plan(const cfnerf_cfg* cfg, int64_t P, int q4, int32_t* tiles_out, uint32_t* segdst_out, int max_tiles) {
    if (!cfg || validate_cfg(*cfg)) return CFNERF_E_UNSUPPORTED;
    ParamLayout L = build_layout(*cfg);
    Stash q;                                   // fake, distinct operand bases: only the geometry is reported
    float* base = reinterpret_cast<float*>(uintptr_t(1) << 40);
    const size_t step = size_t(1) << 36;
    float** ptrs[] = {&q.enc, &q.gd, &q.h, &q.feat, &q.v, &q.ha, &q.hr, &q.theta, &q.g_theta, &q.g_hr, &q.g_ha, &q.g_v, &q.g_feat, &q.g_h};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) *ptrs[i] = base + i * step;
    q.q4 = q4 < 0 ? (P % kTileM == 0) : (q4 != 0);        // the layout the plan is made for (see cfnerf_debug_dw_blocks)
    std::vector<DwTile> big, small;
    build_dw_jobs(*cfg, L, q, P, big, small);
    for (float** pp : ptrs) *pp = nullptr;
    int n = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (const DwTile& t : (pass == 0 ? big : small)) {
            if (n >= max_tiles) return -n;
            int32_t* o = tiles_out + 20 * n;
            o[0] = pass == 0; o[1] = t.n0; o[2] = t.k0; o[3] = t.N; o[4] = t.K; o[5] = t.gk; o[6] = t.wk; o[7] = t.nseg;
            for (int g = 0; g < 4; ++g) { o[8 + g] = t.seg_row[g]; segdst_out[4 * n + g] = t.seg_dst[g]; }
            o[12] = t.dst_ld; o[13] = t.dst_col; o[14] = t.row_f; o[15] = t.late;
            o[16] = t.lay; o[17] = 0; o[18] = 0; o[19] = 0;
            ++n;
        }
    return n;
}

// the blocks of that plan for a given point count and CU count: 5 int64 per block {kind (0: 2 x 4, 1: 1 x 8, 2: small job), tile, split,
// pb, pe}; tile indices refer to the order cfnerf_debug_dw_plan reports (big tiles, then small tiles); plus per tile its nsplit and, per
// parameter tensor, the slot count of the reduction.
CFNERF_API int cfnerf_debug_dw_blocks(const cfnerf_cfg* cfg, int64_t P, int n_cu, int q4, int64_t* blocks_out, int max_blocks, int32_t* tile_nsplit,
                                      int32_t* seg_nsplit, int32_t* seg_early, int max_segs) {
    if (!cfg || validate_cfg(*cfg)) return CFNERF_E_UNSUPPORTED;
    ParamLayout L = build_layout(*cfg);
    Stash q;
    float* base = reinterpret_cast<float*>(uintptr_t(1) << 40);
    const size_t step = size_t(1) << 36;
    float** ptrs[] = {&q.enc, &q.gd, &q.h, &q.feat, &q.v, &q.ha, &q.hr, &q.theta, &q.g_theta, &q.g_hr, &q.g_ha, &q.g_v, &q.g_feat, &q.g_h};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) *ptrs[i] = base + i * step;
    // q4: the layout of the wide streams the plan is made for - 1 / 0 as the product decides it (cfnerf_abi.hip: fp32 mode and whole tiles,
    // i.e. S % 64 == 0 in ray mode, P % 64 == 0 in points mode), -1 = the points-mode rule applied to P
    q.q4 = q4 < 0 ? (P % kTileM == 0) : (q4 != 0);
    DwHost H;
    int n_wide = 0, ns_max = 0;
    const char* why = build_dw_plan(*cfg, L, q, P, n_cu, H, &n_wide, &ns_max);
    for (float** pp : ptrs) *pp = nullptr;
    if (why) return CFNERF_E_UNSUPPORTED;
    const int nb = (int)(H.blocks.size() + H.blocks_small.size());
    if (nb > max_blocks || (int)L.e.size() > max_segs) return -nb;
    int n = 0;
    for (size_t i = 0; i < H.blocks.size(); ++i, ++n) {
        const DwBlock& b = H.blocks[i];
        int64_t* o = blocks_out + 5 * n;
        o[0] = H.tiles[b.tile].gk == 1 ? 1 : 0; o[1] = b.tile; o[2] = b.split; o[3] = b.pb; o[4] = b.pe;
    }
    for (const DwBlock& b : H.blocks_small) {
        int64_t* o = blocks_out + 5 * n++;
        o[0] = 2; o[1] = (int64_t)H.tiles.size() + b.tile; o[2] = b.split; o[3] = b.pb; o[4] = b.pe;
    }
    int t = 0;
    for (const DwTile& x : H.tiles) tile_nsplit[t++] = x.nsplit * (x.gk == 1 ? -1 : 1);     // sign: wave arrangement
    for (const DwTile& x : H.tiles_small) tile_nsplit[t++] = x.nsplit;
    for (size_t i = 0; i < H.segs.size(); ++i) { seg_nsplit[i] = H.segs[i].nsplit; if (seg_early) seg_early[i] = H.segs[i].early; }
    return n;
}
}  // extern "C"
