// cfnerf_layout.h - host-side description of the flat parameter layout and of the packed
// (MFMA-fragment-ordered) weight buffer.  Shared by every translation unit of libcfnerf_hip.so.
//
// Flat layout = NeRF_Flows.state_dict() order (reference model/models.py:38-67, 339-350),
// nn.Linear weights row-major [out, in].
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cfnerf.h"

namespace cfnerf {

constexpr int kMaxDepth = 16;
constexpr int kTileM = 64;            // rows (points) per workgroup tile
constexpr int kFlowsMax = 4;         // flow steps the kernels are built for (--n_flows 1 .. 4: see the theta column map below)
constexpr int kMaxK = 128;            // latent samples per point supported by the fused kernels (the reference's default K_samples is 64, RUN:631)

struct ParamEntry {
    std::string key;
    int64_t off;                      // floats
    int64_t rows, cols;               // cols == 0 -> vector of `rows`
    int64_t numel() const { return cols ? rows * cols : rows; }
};

struct ParamLayout {
    std::vector<ParamEntry> e;
    int64_t total = 0;
    const ParamEntry* find(const char* key) const {
        for (auto& x : e) if (x.key == key) return &x;
        return nullptr;
    }
    int64_t off(const char* key) const { auto* p = find(key); return p ? p->off : -1; }
};

inline int enc_ch(int multires) { return 3 + 6 * multires; }
// Index i of the reference's `i in self.skips` tests (MOD:39,171) with args.skips = [netdepth / 2] (RUN:327, a FLOAT
// division): for an odd netdepth the list holds x.5, nothing ever matches and the trunk is a plain MLP with no skip
// concat.  -2 never equals a layer index (nor index - 1).
inline int skip_layer(int netdepth) { return (netdepth % 2 == 0) ? netdepth / 2 : -2; }
inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// returns nullptr when valid, else the reason
inline const char* validate_cfg(const cfnerf_cfg& c) {
    if (c.netdepth < 3 || c.netdepth > kMaxDepth) return "netdepth must be in [3,16] (at netdepth 2 the reference's skip concat feeds feature_linear and crashes)";
    if (c.netwidth < 64 || c.netwidth > 512 || c.netwidth % 64 != 0)
        return "netwidth must be a multiple of 64 in [64, 512]";
    if (c.multires < 1 || enc_ch(c.multires) > 64) return "multires must be in [1,10]";
    if (c.multires_views < 1 || enc_ch(c.multires_views) > 32) return "multires_views must be in [1,4]";
    if (c.h_alpha_size < 32 || c.h_alpha_size > 128 || c.h_alpha_size % 32) return "h_alpha_size must be 32, 64, 96 or 128";
    if (c.h_rgb_size < 32 || c.h_rgb_size > 128 || c.h_rgb_size % 32) return "h_rgb_size must be 32, 64, 96 or 128";
    // the forward keeps h_rgb next to v in the in-place activation tile: columns [W/2, W/2 + h_rgb_size) of max(W, 128)
    if (c.netwidth / 2 + c.h_rgb_size > (c.netwidth > 128 ? c.netwidth : 128))
        return "h_rgb_size does not fit this netwidth (needs netwidth / 2 + h_rgb_size <= max(netwidth, 128))";
    {   // LDS of the fused forward: act[64][max(W,128)+4] | hs[64][h_alpha+4] | row info, view encoding, reductions | comp[kMaxK][8]
        const long lds = 4L * (64L * ((c.netwidth > 128 ? c.netwidth : 128) + 4) + 64L * (c.h_alpha_size + 4) + 68 * 4 + 32 + 16 + kMaxK * 8);
        if (lds > 160 * 1024) return "h_alpha_size does not fit this netwidth (the forward's LDS tile would exceed the CU's 160 KB)";
    }
    if (c.n_flows < 1 || c.n_flows > kFlowsMax) return "n_flows must be in [1,4]";
    return nullptr;
}

inline ParamLayout build_layout(const cfnerf_cfg& c) {
    ParamLayout L;
    const int W = c.netwidth, D = c.netdepth, ic = enc_ch(c.multires), icv = enc_ch(c.multires_views), F = c.n_flows;
    const int skip = skip_layer(D);   // RUN:327 args.skips = [netdepth / 2]
    auto add = [&](const std::string& k, int64_t r, int64_t cc) {
        ParamEntry p{k, L.total, r, cc};
        L.total += p.numel();
        L.e.push_back(p);
    };
    add("alpha_mean", 1, 0); add("alpha_std", 1, 0); add("rgb_mean", 3, 0); add("rgb_std", 3, 0);
    for (int i = 0; i < D; ++i) {
        int k = (i == 0) ? ic : ((i - 1) == skip ? W + ic : W);          // MOD:39
        add("pts_linears." + std::to_string(i) + ".weight", W, k);
        add("pts_linears." + std::to_string(i) + ".bias", W, 0);
    }
    add("views_linears.0.weight", W / 2, icv + W); add("views_linears.0.bias", W / 2, 0);
    add("feature_linear.weight", W, W);            add("feature_linear.bias", W, 0);
    add("alpha_linear.weight", 1, W);              add("alpha_linear.bias", 1, 0);          // dead (MOD:59)
    add("alpha_std_linear.weight", 1, W);          add("alpha_std_linear.bias", 1, 0);      // dead (MOD:60)
    add("h_alpha_linear.weight", c.h_alpha_size, W);   add("h_alpha_linear.bias", c.h_alpha_size, 0);
    add("h_rgb_linear.weight", c.h_rgb_size, W / 2);   add("h_rgb_linear.bias", c.h_rgb_size, 0);
    const char* names[2] = {"flows_rgb", "flows_alpha"};
    const int zs[2] = {3, 1};
    const int hs[2] = {c.h_rgb_size, c.h_alpha_size};
    for (int t = 0; t < 2; ++t) {
        std::string n = names[t];
        int z = zs[t];
        add(n + ".amor_d.weight", F * z * z, hs[t]);     add(n + ".amor_d.bias", F * z * z, 0);
        add(n + ".amor_diag1.0.weight", F * z, hs[t]);   add(n + ".amor_diag1.0.bias", F * z, 0);
        add(n + ".amor_diag2.0.weight", F * z, hs[t]);   add(n + ".amor_diag2.0.bias", F * z, 0);
        add(n + ".amor_b.weight", F * z, hs[t]);         add(n + ".amor_b.bias", F * z, 0);
    }
    return L;
}

#define CFN_HD_EARLY __host__ __device__      // (every translation unit of the library is compiled by hipcc)
// One packed GEMM operand: B fragments of an [N_out x K_red] matrix for v_mfma_f32_32x32x2_f32.
//   packed[((nt * kc_count + kc) * 64 + lane) * 4 + c] = M[nt*32 + (lane & 31)][kc*8 + 4*(lane >> 5) + c]
// (zero outside the valid range).  A float4 per lane feeds 4 MFMAs; the A operand uses the same
// k assignment, so the order of k inside a chunk is free.
struct SubL {
    uint32_t w_off;      // floats, into the packed buffer
    uint32_t b_off;      // floats, padded bias (nt*32 entries); 0xffffffff = none
    uint16_t kc;         // k chunks of 8
    uint16_t nt;         // n tiles of 32
    // split-bf16 copy of the same operand for v_mfma_f32_32x32x16_bf16 (opt-in "bf16x3" mode):
    //   packed16[w16_off + (((nt*kc16 + c)*2 + plane)*64 + lane)*8 + e] = bf16 part `plane` (0 hi, 1 lo) of
    //   M[nt*32 + (lane&31)][c*16 + 8*(lane>>5) + e]
    uint32_t w16_off;    // in bf16 elements
    // k chunks of 16 of that copy: the reduction axis is padded to 16 there, 8 here
    CFN_HD_EARLY int kc16() const { return (kc + 1) >> 1; }
};
static_assert(sizeof(SubL) == 16, "one table entry = ONE 16-byte scalar load from the kernarg segment (kload)");

// theta (flow-parameter) column map inside a tile row: rgb heads [0,96), alpha heads [96,128).  The kernels are built for
// kFlowsMax = 4 flow steps and address a parameter of step f as (block) * 4 + f:
//   rgb:   [0,36) amor_d (i*3+j)*4+f | [36,48) diag1 i*4+f | [48,60) diag2 | [60,72) b
//   alpha: 96 + [0,4) diag1 | [4,8) diag2 | [8,12) b            (amor_d of z=1 is fully masked: MOD:327,374)
// A model with n_flows = F < 4 (--n_flows, RUN:622) keeps that map: its nn.Linear rows (block) * F + f go to columns (block) * 4 + f,
// the columns of the steps f >= F stay zero in every packed operand, and a step whose parameters are all zero is the identity with
// log-det 0 (z + 0 * tanh(0): FLW:225-268) - so the same kernels run it.  Their gradient columns are computed and dropped.
constexpr int kThetaRgb = 96;
constexpr int kThetaAll = 128;

constexpr int kMaxCu = 256;           // gfx950 has 256 CUs; per-workgroup partial tables are carved for 2 * kMaxCu workgroups
// columns of the per-workgroup bias-gradient partial table, and the number of bias tensors it scatters into
inline int bias_partial_cols(const cfnerf_cfg& c) {
    return c.netdepth * c.netwidth + c.netwidth + c.netwidth / 2 + pad_to(c.h_alpha_size, 32) + pad_to(c.h_rgb_size, 32) + kThetaAll;
}
inline int bias_map_count(const cfnerf_cfg& c) { return c.netdepth + 4 + 21; }      // trunk | feature, views, h_alpha, h_rgb | 18 + 3 theta blocks

struct NetTab {
    // forward
    SubL trunk[kMaxDepth];   // trunk[0]: enc->W ; trunk[l]: h->W
    SubL skipseg;            // enc->W, accumulated into layer skip+1
    SubL ha, ft;             // h->h_alpha, h->feature
    SubL vf, vd;             // feature->W/2, gamma(dir)->W/2 (same accumulator)
    SubL hr;                 // v->h_rgb
    SubL fr, fa;             // h_rgb->theta_rgb(96), h_alpha->theta_alpha(32)
    // backward-data (transposed operands): outputs are the INPUT widths
    SubL bt_fr, bt_fa;       // dtheta_rgb(96)->dh_rgb ; dtheta_alpha(32)->dh_alpha
    SubL bt_hr;              // dh_rgb->dv
    SubL bt_vf;              // dv->dfeature
    SubL bt_ft, bt_ha;       // dfeature->dh ; dh_alpha->dh   (same accumulator)
    SubL bt_trunk[kMaxDepth];// bt_trunk[l]: dh_l -> dh_{l-1}   (l >= 1; skip layer: h segment only)
    int32_t D, W, skip, ic, icv, ha_sz, hr_sz, F;
    uint32_t packed_floats;
    uint32_t packed16_elems;
};

// source piece of a packed operand (device-visible POD)
struct PackDesc {
    uint32_t src_off;    // flat offset of element [0][col0] of the source matrix (already includes col0)
    uint32_t src_ld;     // row stride of the source matrix (0 for a bias vector)
    uint32_t n_rows;     // rows of the source piece  (nn.Linear "out")
    uint32_t n_cols;     // cols of the source piece  (nn.Linear "in"); 0 => bias vector of n_rows
    uint32_t dst_off;    // packed offset of the operand (w_off or b_off)
    uint32_t kc;         // k-chunk count of the destination operand
    uint32_t out_off;    // offset added to the OUTPUT index inside the operand (concatenated heads)
    uint32_t red_off;    // offset added to the REDUCTION index inside the operand
    uint32_t transpose;  // 0: out = row, red = col (forward) ; 1: out = col, red = row (backward-data)
    uint32_t first_elem; // prefix sum of element counts (for the flat thread -> desc search)
    uint32_t dst16_off;  // bf16-copy offset of the operand (elements)
    uint32_t kc16;       // k16-chunk count of the bf16 copy
};

#define CFN_HD __host__ __device__
// flat element `local` of piece `d`  ->  (source index in the flat buffer, destination index in the packed buffer)
CFN_HD inline void pack_map(const PackDesc& d, uint32_t local, uint32_t* src, uint32_t* dst) {
    if (d.n_cols == 0) {
        *src = d.src_off + local;
        *dst = d.dst_off + d.out_off + local;
        return;
    }
    const uint32_t row = local / d.n_cols, col = local - row * d.n_cols;
    *src = d.src_off + row * d.src_ld + col;
    const uint32_t o = (d.transpose ? col : row) + d.out_off;
    const uint32_t r = (d.transpose ? row : col) + d.red_off;
    const uint32_t nt = o >> 5, lane = (o & 31) + 32 * ((r & 7) >> 2), kc = r >> 3, c = r & 3;
    *dst = d.dst_off + ((nt * d.kc + kc) * 64 + lane) * 4 + c;
}

// bf16-copy destination of the same element (hi plane; the lo plane is 64*8 elements further)
CFN_HD inline bool pack_map16(const PackDesc& d, uint32_t local, uint32_t* dst16) {
    if (d.n_cols == 0) return false;
    const uint32_t row = local / d.n_cols, col = local - row * d.n_cols;
    const uint32_t o = (d.transpose ? col : row) + d.out_off;
    const uint32_t r = (d.transpose ? row : col) + d.red_off;
    const uint32_t nt = o >> 5, lane = (o & 31) + 32 * ((r & 15) >> 3), c = r >> 4, e = r & 7;
    *dst16 = d.dst16_off + (((nt * d.kc16 + c) * 2) * 64 + lane) * 8 + e;
    return true;
}

struct PackPlan {
    NetTab tab;
    std::vector<PackDesc> descs;
    uint32_t total_elems = 0;
};

inline PackPlan build_pack_plan(const cfnerf_cfg& c, const ParamLayout& L) {
    PackPlan P;
    NetTab& T = P.tab;
    std::memset(&T, 0, sizeof(T));
    const int W = c.netwidth, D = c.netdepth, ic = enc_ch(c.multires), icv = enc_ch(c.multires_views), F = c.n_flows;
    T.D = D; T.W = W; T.skip = skip_layer(D); T.ic = ic; T.icv = icv; T.ha_sz = c.h_alpha_size; T.hr_sz = c.h_rgb_size; T.F = F;
    uint32_t cur = 0, cur16 = 0;
    auto alloc_op = [&](int n_out, int k_red, bool bias) {
        SubL s;
        s.nt = (uint16_t)(pad_to(n_out, 32) / 32);
        s.kc = (uint16_t)(pad_to(k_red, 8) / 8);
        s.w16_off = cur16; cur16 += (uint32_t)s.nt * s.kc16() * 2 * 64 * 8;
        s.w_off = cur; cur += (uint32_t)s.nt * s.kc * 256;
        if (bias) { s.b_off = cur; cur += (uint32_t)s.nt * 32; } else s.b_off = 0xffffffffu;
        return s;
    };
    auto piece = [&](const SubL& s, const char* key, int col0, int ncols, int out_off, int red_off, bool transpose, int row0 = 0, int nrows = -1) {
        const ParamEntry* e = L.find(key);
        PackDesc d{};
        d.src_off = (uint32_t)(e->off + (int64_t)row0 * e->cols + col0); d.src_ld = (uint32_t)e->cols; d.n_rows = (uint32_t)(nrows < 0 ? e->rows : nrows);
        d.n_cols = (uint32_t)ncols; d.dst_off = s.w_off; d.kc = s.kc; d.out_off = out_off; d.red_off = red_off;
        d.dst16_off = s.w16_off; d.kc16 = (uint32_t)s.kc16();
        d.transpose = transpose ? 1 : 0; d.first_elem = P.total_elems;
        P.total_elems += d.n_rows * d.n_cols;
        P.descs.push_back(d);
    };
    auto bias_piece = [&](const SubL& s, const char* key, int out_off, int row0 = 0, int nrows = -1) {
        const ParamEntry* e = L.find(key);
        PackDesc d{};
        d.src_off = (uint32_t)(e->off + row0); d.src_ld = 0; d.n_rows = (uint32_t)(nrows < 0 ? e->rows : nrows); d.n_cols = 0;
        d.dst_off = s.b_off; d.kc = 0; d.out_off = out_off; d.red_off = 0; d.transpose = 0; d.first_elem = P.total_elems;
        P.total_elems += d.n_rows;
        P.descs.push_back(d);
    };
    char kw[64], kb[64];
    // ---- forward operands
    for (int l = 0; l < D; ++l) {
        std::snprintf(kw, sizeof kw, "pts_linears.%d.weight", l);
        std::snprintf(kb, sizeof kb, "pts_linears.%d.bias", l);
        if (l == 0) {
            T.trunk[0] = alloc_op(W, ic, true);
            piece(T.trunk[0], kw, 0, ic, 0, 0, false);
        } else if (l - 1 == T.skip) {                       // input = [gamma(p) ic | h W]  (MOD:172)
            T.trunk[l] = alloc_op(W, W, true);
            piece(T.trunk[l], kw, ic, W, 0, 0, false);
            T.skipseg = alloc_op(W, ic, false);
            piece(T.skipseg, kw, 0, ic, 0, 0, false);
        } else {
            T.trunk[l] = alloc_op(W, W, true);
            piece(T.trunk[l], kw, 0, W, 0, 0, false);
        }
        bias_piece(T.trunk[l], kb, 0);
    }
    T.ha = alloc_op(c.h_alpha_size, W, true);
    piece(T.ha, "h_alpha_linear.weight", 0, W, 0, 0, false); bias_piece(T.ha, "h_alpha_linear.bias", 0);
    T.ft = alloc_op(W, W, true);
    piece(T.ft, "feature_linear.weight", 0, W, 0, 0, false); bias_piece(T.ft, "feature_linear.bias", 0);
    T.vf = alloc_op(W / 2, W, true);                        // input = [feature W | gamma(d) icv] (MOD:177)
    piece(T.vf, "views_linears.0.weight", 0, W, 0, 0, false); bias_piece(T.vf, "views_linears.0.bias", 0);
    T.vd = alloc_op(W / 2, icv, false);
    piece(T.vd, "views_linears.0.weight", W, icv, 0, 0, false);
    T.hr = alloc_op(c.h_rgb_size, W / 2, true);
    piece(T.hr, "h_rgb_linear.weight", 0, W / 2, 0, 0, false); bias_piece(T.hr, "h_rgb_linear.bias", 0);
    // theta heads: block b of a head tensor = its rows [b F, (b + 1) F) -> columns base + 4 b + f (see the column map above)
    const char* ks_r[4] = {"flows_rgb.amor_d", "flows_rgb.amor_diag1.0", "flows_rgb.amor_diag2.0", "flows_rgb.amor_b"};
    const int base_r[4] = {0, 9 * kFlowsMax, 12 * kFlowsMax, 15 * kFlowsMax}, blocks_r[4] = {9, 3, 3, 3};
    const char* ks_a[3] = {"flows_alpha.amor_diag1.0", "flows_alpha.amor_diag2.0", "flows_alpha.amor_b"};
    T.fr = alloc_op(kThetaRgb, c.h_rgb_size, true);
    for (int i = 0; i < 4; ++i) {
        std::snprintf(kw, sizeof kw, "%s.weight", ks_r[i]); std::snprintf(kb, sizeof kb, "%s.bias", ks_r[i]);
        for (int b = 0; b < blocks_r[i]; ++b) {
            piece(T.fr, kw, 0, c.h_rgb_size, base_r[i] + kFlowsMax * b, 0, false, b * F, F);
            bias_piece(T.fr, kb, base_r[i] + kFlowsMax * b, b * F, F);
        }
    }
    T.fa = alloc_op(kThetaAll - kThetaRgb, c.h_alpha_size, true);
    for (int i = 0; i < 3; ++i) {
        std::snprintf(kw, sizeof kw, "%s.weight", ks_a[i]); std::snprintf(kb, sizeof kb, "%s.bias", ks_a[i]);
        piece(T.fa, kw, 0, c.h_alpha_size, kFlowsMax * i, 0, false, 0, F);
        bias_piece(T.fa, kb, kFlowsMax * i, 0, F);
    }
    // ---- backward-data operands: dX[., in] = sum_out dY[., out] * W[out][in]  -> out index = col, red index = row
    T.bt_fr = alloc_op(c.h_rgb_size, kThetaRgb, false);
    for (int i = 0; i < 4; ++i) {
        std::snprintf(kw, sizeof kw, "%s.weight", ks_r[i]);
        for (int b = 0; b < blocks_r[i]; ++b) piece(T.bt_fr, kw, 0, c.h_rgb_size, 0, base_r[i] + kFlowsMax * b, true, b * F, F);
    }
    T.bt_fa = alloc_op(c.h_alpha_size, kThetaAll - kThetaRgb, false);
    for (int i = 0; i < 3; ++i) {
        std::snprintf(kw, sizeof kw, "%s.weight", ks_a[i]);
        piece(T.bt_fa, kw, 0, c.h_alpha_size, 0, kFlowsMax * i, true, 0, F);
    }
    T.bt_hr = alloc_op(W / 2, c.h_rgb_size, false);
    piece(T.bt_hr, "h_rgb_linear.weight", 0, W / 2, 0, 0, true);
    T.bt_vf = alloc_op(W, W / 2, false);
    piece(T.bt_vf, "views_linears.0.weight", 0, W, 0, 0, true);
    T.bt_ft = alloc_op(W, W, false);
    piece(T.bt_ft, "feature_linear.weight", 0, W, 0, 0, true);
    T.bt_ha = alloc_op(W, c.h_alpha_size, false);
    piece(T.bt_ha, "h_alpha_linear.weight", 0, W, 0, 0, true);
    for (int l = 1; l < D; ++l) {
        std::snprintf(kw, sizeof kw, "pts_linears.%d.weight", l);
        T.bt_trunk[l] = alloc_op(W, W, false);
        piece(T.bt_trunk[l], kw, (l - 1 == T.skip) ? ic : 0, W, 0, 0, true);
    }
    T.packed_floats = cur;
    T.packed16_elems = cur16;
    return P;
}

}  // namespace cfnerf
