// cfnerf_dwplan.h - HOST-side plan of the weight-gradient launches (dW = dY^T X per layer): which jobs become 256 x 256 tiles of the big
// kernel and which go to the small-job kernel, how the points are split over workgroups, the per-tensor slot counts of the reduction.
// Pure geometry: it depends on the configuration, the point count and the CU count only (operand base pointers are carried along), so
// the CPU tests rebuild it through tests/csrc/cfnerf_testhooks.hip without a device.  Used by cfnerf_bwd.hip.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "cfnerf_bwd.h"
#include "cfnerf_layout.h"
#include "cfnerf_model.h"

namespace cfnerf {

constexpr int kDwRows = 32;                   // points per LDS stage of both weight-gradient kernels

// one weight-gradient job -> tiles: 256 x 256 ("big" kernel) when the job is at least 128 x 128; else ("small" kernel)
// tiles of (32 GN) x (32 GK WK), GN GK = 8 waves, with the wave arrangement picked from the job's shape
inline void add_job(std::vector<DwTile>& big, std::vector<DwTile>& small, const float* dY, int ldY, int Nread, int N, const float* X,
                    int ldX, int K, int Kvalid, int nseg, const int* seg_row, const uint32_t* seg_dst, int dst_ld, int dst_col, int row_f = 0,
                    int q4cap = 0, bool q4 = false) {
    // q4cap: which operands are streams that take the Q4 layout when the stash does (bit 0 = dY, bit 1 = X); lay = what they ARE in this plan.
    // (a job with ONE Q4 operand - g_ha x h at h_alpha_size 128 - goes to the small kernel, which takes either layout per operand: WHERE such
    // a job runs depends on the stash layout, so its tensor must never count as "early" - the ranges of cfnerf_grad_early_ranges are cached by
    // callers and have to depend on the configuration alone)
    const int lay = q4 ? q4cap : 0;
    const bool shape_big = N >= 128 && Kvalid >= 128;
    const bool is_big = shape_big && (lay == 0 || lay == 3);
    const bool late = shape_big && (q4cap == 1 || q4cap == 2);
    int gk = 0, wk = 0;
    if (!is_big) {          // wave arrangement GN x GK (GN GK = 8) from the job's K; a tile stages at most 128 + 64 or 64 + 128 columns
        wk = 1;
        gk = Kvalid <= 32 ? 1 : Kvalid <= 64 ? 2 : 4;
    }
    const int tn = is_big ? 256 : std::min(32 * (8 / gk), 128), tk = is_big ? 256 : 32 * gk;
    for (int n0 = 0; n0 < N; n0 += tn)
        for (int k0 = 0; k0 < Kvalid; k0 += tk) {
            DwTile t{};
            t.gk = is_big ? (N - n0 <= 128 ? 1 : 0) : gk; t.wk = wk;
            t.dY = dY; t.ldY = ldY; t.N = N; t.Npad = Nread; t.X = X; t.ldX = ldX; t.K = Kvalid; t.Kpad = K; t.n0 = n0; t.k0 = k0;
            t.nseg = nseg;
            for (int q = 0; q < 4; ++q) { t.seg_row[q] = q < nseg ? seg_row[q] : 0x7fffffff; t.seg_dst[q] = q < nseg ? seg_dst[q] : 0; }
            t.dst_ld = dst_ld; t.dst_col = dst_col; t.row_f = row_f; t.lay = lay; t.late = late ? 1 : 0;
            (is_big ? big : small).push_back(t);
        }
}

// every weight-gradient job of the network (dW[n][k] = sum_p dY[p][n] X[p][k]) as big / small tiles.  `q` supplies
// the operand base pointers (geometry only depends on the configuration: the CPU test of the plan passes fakes).
inline void build_dw_jobs(const cfnerf_cfg& c, const ParamLayout& L, const Stash& q, int64_t P, std::vector<DwTile>& big,
                          std::vector<DwTile>& small) {
    const int W = c.netwidth, D = c.netdepth, HA = c.h_alpha_size, HR = c.h_rgb_size, F = c.n_flows;
    const int ic = enc_ch(c.multires), icv = enc_ch(c.multires_views), skip = skip_layer(D);
    char key[64];
    const int one_row[1] = {0};
    // operand layouts (DwTile::lay: bit 0 = dY, bit 1 = X): with whole tiles in fp32 mode (Stash::q4) every WIDE stream - h[l], feature, v and
    // their pre-activation gradients g_h[l], g_feat, g_v - is Q4 (cfnerf_device.h; round 5: the trunk streams, round 6: feature / v / g_v);
    // the encodings, the narrow heads and theta / g_theta stay row-major
    const int qY = 1, qX = 2;
    const bool Q = q.q4;
    for (int l = 0; l < D; ++l) {
        std::snprintf(key, sizeof key, "pts_linears.%d.weight", l);
        const uint32_t dst[1] = {(uint32_t)L.off(key)};
        const float* dY = q.g_h + (size_t)l * P * W;
        if (l == 0) {
            add_job(big, small, dY, W, W, W, q.enc, 64, 64, ic, 1, one_row, dst, ic, 0, 0, qY, Q);
        } else if (l - 1 == skip) {
            add_job(big, small, dY, W, W, W, q.enc, 64, 64, ic, 1, one_row, dst, ic + W, 0, 0, qY, Q);
            add_job(big, small, dY, W, W, W, q.h + (size_t)(l - 1) * P * W, W, W, W, 1, one_row, dst, ic + W, ic, 0, qY | qX, Q);
        } else {
            add_job(big, small, dY, W, W, W, q.h + (size_t)(l - 1) * P * W, W, W, W, 1, one_row, dst, W, 0, 0, qY | qX, Q);
        }
    }
    const float* hlast = q.h + (size_t)(D - 1) * P * W;
    { const uint32_t dst[1] = {(uint32_t)L.off("h_alpha_linear.weight")}; add_job(big, small, q.g_ha, HA, HA, HA, hlast, W, W, W, 1, one_row, dst, W, 0, 0, qX, Q); }
    { const uint32_t dst[1] = {(uint32_t)L.off("feature_linear.weight")}; add_job(big, small, q.g_feat, W, W, W, hlast, W, W, W, 1, one_row, dst, W, 0, 0, qY | qX, Q); }
    {
        const uint32_t dst[1] = {(uint32_t)L.off("views_linears.0.weight")};
        add_job(big, small, q.g_v, W / 2, W / 2, W / 2, q.feat, W, W, W, 1, one_row, dst, W + icv, 0, 0, qY | qX, Q);
        add_job(big, small, q.g_v, W / 2, W / 2, W / 2, q.gd, 32, 32, icv, 1, one_row, dst, W + icv, W, 0, qY, Q);
    }
    { const uint32_t dst[1] = {(uint32_t)L.off("h_rgb_linear.weight")}; add_job(big, small, q.g_hr, HR, HR, HR, q.v, W / 2, W / 2, W / 2, 1, one_row, dst, W / 2, 0, 0, qX, Q); }
    {
        const int rows[4] = {0, 9 * F, 12 * F, 15 * F};
        const uint32_t dst[4] = {(uint32_t)L.off("flows_rgb.amor_d.weight"), (uint32_t)L.off("flows_rgb.amor_diag1.0.weight"),
                                 (uint32_t)L.off("flows_rgb.amor_diag2.0.weight"), (uint32_t)L.off("flows_rgb.amor_b.weight")};
        add_job(big, small, q.g_theta, kThetaAll, kThetaAll, 18 * kFlowsMax, q.hr, HR, HR, HR, 4, rows, dst, HR, 0, F);      // dY columns in the kernels' 4-step map
    }
    {
        const int rows[3] = {0, F, 2 * F};
        const uint32_t dst[3] = {(uint32_t)L.off("flows_alpha.amor_diag1.0.weight"), (uint32_t)L.off("flows_alpha.amor_diag2.0.weight"),
                                 (uint32_t)L.off("flows_alpha.amor_b.weight")};
        add_job(big, small, q.g_theta + kThetaRgb, kThetaAll, kThetaAll - kThetaRgb, 3 * kFlowsMax, q.ha, HA, HA, HA, 3, rows, dst, HA, 0, F);
    }
}

// blocks of a launch: tile t contributes t.nsplit blocks, each with an equal share of the points (rounded to whole LDS
// stages).  Slots [0, used splits of its tile) of a tensor's partials are exactly the ones a launch writes.
inline void make_blocks(std::vector<DwBlock>& blocks, std::vector<DwTile>& tiles, int64_t P, int round_to, int only_arr = -1) {
    const size_t first = blocks.size();
    for (int t = 0; t < (int)tiles.size(); ++t) {
        if (only_arr >= 0 && tiles[t].gk != only_arr) continue;
        const int nsplit = std::max(1, tiles[t].nsplit);
        int64_t chunk = (P + nsplit - 1) / nsplit;
        chunk = (chunk + round_to - 1) / round_to * round_to;
        int used = 0;
        for (int s = 0; s < nsplit; ++s) {
            const int64_t pb = (int64_t)s * chunk, pe = std::min<int64_t>(P, pb + chunk);
            if (pb >= pe) continue;
            DwBlock b; b.tile = t; b.split = used++; b.kslice = 0; b.pad_ = 0; b.pb = pb; b.pe = pe;
            blocks.push_back(b);
        }
        tiles[t].nsplit = used;
    }
    // longest blocks first: the hardware dispatches in index order
    std::stable_sort(blocks.begin() + first, blocks.end(), [](const DwBlock& a, const DwBlock& b) { return (a.pe - a.pb) > (b.pe - b.pb); });
}

// Split counts of the big tiles: one block per CU in total (every block of the launch has the same footprint, so the
// hardware places exactly one per CU whatever the mix), points shared out so that every block takes about the same time:
// a 1 x 8 block (N <= 128: the views layer) issues half the MFMAs per stage of a 2 x 4 block but pays the same fixed cost per stage, so it
// gets proportionally more points: 0.59 of a 2 x 4 block's time per point when both kinds stage row-major operands (measured in round 3;
// still the bf16x3 mode and ragged batches), 0.53 against the Q4 bodies of the 2 x 4 tiles (round 5, same-box A/B over 0.50 / 0.54 / 0.59 /
// 0.66: the default network's nine tiles then split 8 x 30 + 16 instead of 6 x 30 + 2 x 29 + 18 - with 0.59 the blocks of the two 29-split
// tiles ran 3.4 % longer than everything else: weight-gradient stage 1.224 -> 1.193 ms at C2).
//
// ROUNDS (round 5).  With one block per CU the longest block is the launch: 34 equal tiles on 256 CUs (W = 512) split 18 x 8 + 16 x 7 and the
// 7-split tiles' blocks run 14 % longer than the other half of the chip - the launch sat 7.6 % above its even share.  With two blocks per CU
// (2 x 15 + 32 x 16 ... every CU takes a long and then a shorter block, the hardware hands the next block of the list to the CU that frees
// up first) the shares even out, at the price of one more block prologue + 256-KB partial store per CU, ~23 us = ~100 points' worth
// (measured: W = 512 weight-gradient stage 2.308 -> 2.194 ms with two rounds; C2, whose nine tiles already split evenly, 1.195 -> 1.218).
// The plan is made for 1, 2 and 3 rounds and the one with the shortest modelled makespan kept.
constexpr double kDwBlockOverheadPoints = 100.0;
inline double big_split_makespan(const std::vector<DwTile>& tiles, const std::vector<int>& nsplit, int n_cu, int64_t P, double arr1) {
    std::vector<double> len;                                 // block lengths in 2 x 4-tile points, longest first = the dispatch order
    for (size_t i = 0; i < tiles.size(); ++i)
        for (int s = 0; s < nsplit[i]; ++s)
            len.push_back((tiles[i].gk == 1 ? arr1 : 1.0) * (double)((P + nsplit[i] - 1) / nsplit[i]) + kDwBlockOverheadPoints);
    std::sort(len.begin(), len.end(), [](double a, double b) { return a > b; });
    std::vector<double> cu((size_t)n_cu, 0.0);               // a free CU takes the next block of the list
    for (double l : len) *std::min_element(cu.begin(), cu.end()) += l;
    return *std::max_element(cu.begin(), cu.end());
}
inline void balance_big_splits(std::vector<DwTile>& tiles, int n_cu, int64_t P, int max_split) {
    if (tiles.empty()) return;
    bool q4 = false;
    for (const DwTile& t : tiles) q4 = q4 || t.lay == 3;
    const double arr1 = q4 ? 0.53 : 0.59;
    auto cost = [arr1](const DwTile& t) { return t.gk == 1 ? arr1 : 1.0; };
    double total = 0;
    for (const DwTile& t : tiles) total += cost(t);
    int cap = max_split;
    while (cap > 1 && P / cap < 512) --cap;                  // at least 512 points per block
    std::vector<int> best_split;
    double best_span = 0;
    for (int rounds = 1; rounds <= 3; ++rounds) {
        const int slots = n_cu * rounds;
        std::vector<int> ns(tiles.size());
        int used = 0;
        for (size_t i = 0; i < tiles.size(); ++i) { ns[i] = std::max(1, std::min(cap, (int)(slots * cost(tiles[i]) / total))); used += ns[i]; }
        while (used < slots) {                               // hand the remaining slots to the tiles whose blocks are longest
            int best = -1;
            for (size_t i = 0; i < tiles.size(); ++i)
                if (ns[i] < cap && (best < 0 || cost(tiles[i]) / ns[i] > cost(tiles[(size_t)best]) / ns[(size_t)best])) best = (int)i;
            if (best < 0) break;
            ++ns[(size_t)best]; ++used;
        }
        const double span = big_split_makespan(tiles, ns, n_cu, P, arr1);
        if (best_split.empty() || span < best_span * 0.995) { best_split = ns; best_span = span; }      // (more rounds only for a clear gain)
    }
    for (size_t i = 0; i < tiles.size(); ++i) tiles[i].nsplit = best_split[i];
}

// Split counts of the small jobs.  The launch is HBM-bound (a block streams 32 x (a_ld + b_ld) floats per stage), two
// workgroups fit a CU, and with a fixed 128 splits per tile the 10 tiles of the default network made 1280 blocks = 2.5
// rounds of 512 slots: the last half round ran on a half-empty chip.  Instead the launch is ONE round - 2 n_cu blocks that
// all start together - and a tile's share of them is proportional to the bytes it streams per point (+ a fixed per-stage
// cost), so they also end together; fewer splits are fewer partial slots for the reduction to read, too.
inline int small_stage_cols(const DwTile& t) {
    const int gk = t.gk, gn = 8 / gk, tn = std::min(32 * gn, 128), tk = 32 * gk;
    return pad_to(std::min(tn, t.N - t.n0), 32) + pad_to(std::min(tk, t.K - t.k0), 32);
}
inline void balance_small_splits(std::vector<DwTile>& tiles, int n_cu, int64_t P) {
    if (tiles.empty()) return;
    int cap = kDwSlots;
    while (cap > 1 && P / cap < 512) cap >>= 1;              // at least 512 points per block
    // (+ 64 columns' worth of fixed cost per stage: same-box sweep over 32 / 64 / 100 / 150 / 220 in round 6 - weight-gradient stage 1.1913 /
    //  1.1834 / 1.1879 / 1.1917 / 1.1926 ms at C2; rounds 3 - 5 used 32.  A narrow tile's 8-KB stage takes 0.94 us where a 24-KB one takes 1.64)
    auto cost = [](const DwTile& t) { return (double)(small_stage_cols(t) + 64); };
    double total = 0;
    for (const DwTile& t : tiles) total += cost(t);
    const int slots = 2 * n_cu;
    int used = 0;
    for (DwTile& t : tiles) { t.nsplit = std::max(1, std::min(cap, (int)(slots * cost(t) / total))); used += t.nsplit; }
    while (used < slots) {                                   // the remaining slots go to the tiles whose blocks are longest
        DwTile* best = nullptr;
        for (DwTile& t : tiles)
            if (t.nsplit < cap && (!best || cost(t) / t.nsplit > cost(*best) / best->nsplit)) best = &t;
        if (!best) break;
        ++best->nsplit; ++used;
    }
}

// The whole host-side plan of the weight-gradient launches for one workspace binding: tiles, per-tile splits, blocks
// (2 x 4 tiles first, then the 1 x 8 tiles), small-job blocks and the per-tensor slot counts of the reduction.
// Returns nullptr or the reason it cannot be built.
inline const char* build_dw_plan(const cfnerf_cfg& c, const ParamLayout& L, const Stash& q, int64_t P, int n_cu, DwHost& Hs,
                                 int* n_blocks_wide, int* ns_max_out) {
    Hs.tiles.clear(); Hs.tiles_small.clear(); Hs.blocks.clear(); Hs.blocks_small.clear(); Hs.segs.clear();
    build_dw_jobs(c, L, q, P, Hs.tiles, Hs.tiles_small);
    // split counts: big tiles ~1 block per CU in total, balanced by per-tile cost; small jobs a finer split (their
    // blocks are short and run several per CU)
    const int kMaxSplit = 64;
    balance_big_splits(Hs.tiles, n_cu, P, kMaxSplit);
    balance_small_splits(Hs.tiles_small, n_cu, P);
    make_blocks(Hs.blocks, Hs.tiles, P, kDwRows);              // one launch: longest blocks first, whatever their arrangement
    *n_blocks_wide = 0;
    for (const DwBlock& b : Hs.blocks) *n_blocks_wide += Hs.tiles[b.tile].gk == 0;
    make_blocks(Hs.blocks_small, Hs.tiles_small, P, kDwRows);
    if ((int)Hs.tiles.size() > kMaxDwTiles || (int)Hs.tiles_small.size() > kMaxDwTiles || (int)Hs.blocks.size() > kMaxDwBlocks ||
        (int)Hs.blocks_small.size() > kMaxDwBlocks)
        return "weight-gradient plan exceeds the descriptor capacity";
    // per-tensor split counts for the reduction (biases / dead tensors: 0 slots).  The tiles of one tensor may use
    // different counts (a big and a small tile of the skip / views layer; tiles that got a spare CU): the tensor is
    // reduced over the largest, so the slots some tile never writes must read as zero - the caller clears them when
    // the plan is (re)built, never on the steady path, where every launch rewrites exactly the slots it wrote before.
    int ns_max = 1;
    for (const ParamEntry& e : L.e) { RedSeg r{}; r.begin = (uint32_t)e.off; r.nsplit = 0; r.early = 0; Hs.segs.push_back(r); }
    auto mark = [&](const std::vector<DwTile>& tv, bool big) {
        for (const DwTile& t : tv)
            for (int g = 0; g < t.nseg; ++g)
                for (RedSeg& r : Hs.segs)
                    if (r.begin == t.seg_dst[g]) {
                        const bool e = big && !t.late;     // (a layout-dependent job is never early: DwTile::late)
                        if (r.nsplit == 0) r.early = e ? 1 : 0; else if (!e) r.early = 0;
                        r.nsplit = std::max(r.nsplit, t.nsplit); ns_max = std::max(ns_max, t.nsplit);
                    }
    };
    mark(Hs.tiles, true); mark(Hs.tiles_small, false);
    *ns_max_out = ns_max;
    return nullptr;
}

}  // namespace cfnerf
