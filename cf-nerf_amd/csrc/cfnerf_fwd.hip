// cfnerf_fwd.hip - fused forward of the CF-NeRF ray-batch path for gfx950 (MI355X).
//
// One workgroup (4 waves) owns a tile of 64 points: positional encoding -> 8-layer trunk ->
// heads -> flow-parameter heads all run out of ONE LDS-resident activation tile (in place), the
// dense layers on v_mfma_f32_32x32x2_f32 (exact fp32) with the weights streamed from L2 in
// fragment order; then the K conditional flows and the alpha-composite run on the VALU with
// wavefront scans.  In ray mode a workgroup walks whole rays (tiles of one ray are consecutive),
// so `raw` never has to leave the chip.
//
// Reference functions replaced: render_rays RUN:457-553, run_network RUN:67-85, Embedder HLP:21-69,
// NeRF_Flows.encode/forward MOD:165-291, TriangularSylvesterNeRF MOD:358-416, TriangularSylvester
// FLW:189-268, raw2outputs RUN:411-454.
#include <type_traits>

#include "cfnerf_device.h"
#include "cfnerf_kernels.h"

namespace cfnerf {

// Waves per workgroup of the fused forward.  W <= 256: 4 waves, two workgroups per CU (one's epilogues hide under the
// other's MFMAs).  W = 512: the in-place LDS tile (132 KB) allows one workgroup per CU, and 64 x 512 outputs held in
// registers until every wave has read the layer's input are 128 accumulator registers per lane with 4 waves - with the
// weight ping-pong that overflows the 256 architectural VGPRs (the 4-wave build spilt ~700 registers to scratch and
// shuffled thousands of values through AccVGPRs).  8 waves halve it: 2 n-tiles per wave like the W = 256 kernel, two
// waves per SIMD, 256 registers each, no scratch.
__host__ __device__ constexpr int fwd_waves(int W) { return W > 256 ? 8 : 4; }

template <int W>
struct FwdCfg {
    static constexpr int NWV = fwd_waves(W);
    static constexpr int NTHR = NWV * 64;
    static constexpr int NT = W / 32;                         // n tiles of a W-wide layer
    static constexpr int NTW = (NT + NWV - 1) / NWV;          // per wave
    static constexpr int NTV = (W / 64 + NWV - 1) / NWV;      // views layer (W/2 wide)
    static constexpr int LD = act_ld(W);
};

// comp[] holds one 8-float accumulator row per latent sample: sized by the launch's K (rounded up to 16), not by kMaxK, so that
// two workgroups of the W = 256 kernel keep fitting a CU's 160 KB whatever the limit is
__host__ __device__ inline int comp_rows(int K) { return (K + 15) / 16 * 16; }
__host__ __device__ inline size_t fwd_lds_bytes(int W, int ha, int K) {
    // act[64][LD] | hs[64][ha+4] | rowinfo[65][4] (+pad) | gdir[32] | red[16] | comp[comp_rows(K)][8]
    return sizeof(float) * ((size_t)kTileM * act_ld(W) + (size_t)kTileM * (ha + 4) + 68 * 4 + 32 + 16 + (size_t)comp_rows(K) * 8);
}

__device__ __forceinline__ float zlin_f(float t, float nearv, float farv, bool lindisp) {
    if (lindisp) return 1.f / ((1.f / nearv) * (1.f - t) + (1.f / farv) * t);       // RUN:514
    return nearv * (1.f - t) + farv * t;                                            // RUN:512
}

// positional encoding of the tile into act[:, 0:64)   (HLP:42-51, RUN:70-71)
template <int MODE, int LD, int PREC, int NTHR>
__device__ __forceinline__ void encode_tile(float* act, const float* rowinfo, const float* __restrict__ x, int64_t p0,
                                            int rows_valid, int ic, int icv) {
    const int tid = threadIdx.x;
    if (MODE == 0) {
        // one work item = (row, pair): pair p < 30 is (frequency f = p / 3, coordinate d = p % 3) and yields the sin AND the
        // cos channel (3 + 6f + d, 3 + 6f + 3 + d) from one argument reduction; pair 30 carries the identity channels and
        // the padding.  A wave holds one pair for 64 rows: no divergence.  8 items per thread instead of 16 sin-or-cos.
        const int nfreq = (ic - 3) / 6;
        for (int idx = tid; idx < kTileM * 32; idx += NTHR) {
            const int row = idx & 63, p = idx >> 6;
            const float* v = rowinfo + row * 4;
            float* dst = act + row * LD;
            if (p < 30) {
                const int f = p / 3, d = p - 3 * f;
                float sv = 0.f, cv = 0.f;
                if (f < nfreq) sincosf(v[d] * (float)(1 << f), &sv, &cv);
                act_store<PREC>(dst, LD, 3 + 6 * f + d, sv);
                act_store<PREC>(dst, LD, 3 + 6 * f + 3 + d, cv);
            } else if (p == 30) {
                act_store<PREC>(dst, LD, 0, v[0]); act_store<PREC>(dst, LD, 1, v[1]); act_store<PREC>(dst, LD, 2, v[2]);
                act_store<PREC>(dst, LD, 63, 0.f);
            }
        }
    } else {
        for (int idx = tid; idx < kTileM * 64; idx += NTHR) {
            const int row = idx >> 6, c = idx & 63;
            float v = 0.f;
            if (c < ic && row < rows_valid) v = x[(p0 + row) * (int64_t)(ic + icv) + c];
            act_store<PREC>(act + row * LD, LD, c, v);
        }
    }
}


// the kernarg segment of fused_fwd_kernel as one struct (second argument at the first argument's size rounded up to its own alignment)
struct FwdKargs { FwdArgs A; NetTab T; };
static_assert(offsetof(FwdKargs, T) == (sizeof(FwdArgs) + alignof(NetTab) - 1) / alignof(NetTab) * alignof(NetTab), "kernarg layout");

// Q4 (train, fp32, whole tiles only - launch_fused_fwd picks the variant from FwdArgs::q4): the trunk activations h[0 .. D-1], 8 of the 9.5
// W-wide units the forward stashes per point, leave as Q4 pieces straight from the accumulator registers (cfnerf_device.h) instead of by
// rows out of LDS.
template <int W, int MODE /*0 rays, 1 points*/, bool TRAIN, int PREC, bool Q4 = false>
__global__ __launch_bounds__(FwdCfg<W>::NTHR, 2)
void fused_fwd_kernel(const FwdArgs A_, const NetTab T_) {
    static_assert(!Q4 || (TRAIN && PREC == PREC_F32), "the Q4 stash layout exists for the fp32 train variants only");
    // The arguments live in the kernarg segment, so every per-layer descriptor read is a scalar load from constant
    // memory.  (Through a global pointer the compiler must assume the kernel's own stores may alias the table and issues
    // VECTOR loads with a full wait in front of each layer's first operand fetch: two or three dependent L2 round trips.)
    // They are read PER PHASE through CFN_PHASE_ARGS (kernarg_fresh, cfnerf_device.h), not held in registers across the tile loop.
    // The few scalars every phase needs (sizes, the wave index, the operand pointers) stay in SGPRs, but a phase takes them through
    // sgpr_fresh: what the optimiser DERIVES from them (row offsets rr * HR * 4 of a slab store, wave * 32, LDS sub-pointers ...) is then
    // recomputed on the scalar unit inside the phase instead of being hoisted in front of the tile loop and spilt as well.
#define CFN_PHASE_LOCALS(F)                                                                                                              \
    [[maybe_unused]] const int HA = F(HA0), HR = F(HR0), HLD = HA + 4, S = F(S0), K = F(K0), ic = F(ic0), icv = F(icv0), wave = F(wave0);  \
    [[maybe_unused]] const int chunks_per_ray = (S + kTileM - 1) / kTileM, Dn = F(Dn0), skip_l = F(skip0);                                \
    [[maybe_unused]] float* const rowinfo = hs + kTileM * HLD; /* [65][4]: x y z zval */                                                  \
    [[maybe_unused]] float* const gdir = rowinfo + 68 * 4;     /* [32] */                                                                 \
    [[maybe_unused]] float* const red = gdir + 32;             /* [16] */                                                                 \
    [[maybe_unused]] float* const comp = red + 16;             /* [comp_rows(K)][8]: r g b depth acc T disp _ */                          \
    [[maybe_unused]] const float* __restrict__ const wp = F(wp0);                                                                          \
    [[maybe_unused]] const __bf16* __restrict__ const wp16 = F(wp160)
    (void)A_; (void)T_;
#define CFN_KARGS const CFN_KCONST FwdKargs* kq_ = kernarg_fresh<FwdKargs>(); const CFN_KCONST FwdArgs& A = kq_->A; const CFN_KCONST NetTab& T = kq_->T
#define CFN_PHASE_ARGS CFN_KARGS; CFN_PHASE_LOCALS(sgpr_fresh)
    CFN_KARGS;
    using C = FwdCfg<W>;
    constexpr int LD = C::LD;
    constexpr int kWv = C::NWV, kThr = C::NTHR;     // waves / threads of this width's workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HA0 = T.ha_sz, HR0 = T.hr_sz;
    float* const act = smem;
    float* const hs = act + kTileM * LD;      // then rowinfo | gdir | red | comp: CFN_PHASE_LOCALS

    const int tid = threadIdx.x, lane = lane_id(), wave0 = wave_id();
    const float* const wp0 = A.wp;
    const __bf16* const wp160 = reinterpret_cast<const __bf16*>(A.wp16);
    const int S0 = A.S, K0 = A.K;
    const int ic0 = T.ic, icv0 = T.icv;
    const int Dn0 = T.D, skip0 = T.skip;
    const int64_t n_units = (MODE == 0) ? A.N : (A.P + kTileM - 1) / kTileM;

    float ent_r_sum = 0.f, ent_a_sum = 0.f;   // per-lane partial sums of the log-det terms (TRAIN)

    const float a_mean = A.flat[0], a_std = A.flat[1];
    const float r_mean[3] = {A.flat[2], A.flat[3], A.flat[4]};
    const float r_std[3] = {A.flat[5], A.flat[6], A.flat[7]};

    for (int64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        float ro[3], rd[3], nearv = 0.f, farv = 1.f, dnorm = 0.f;
        CFN_PHASE_ARGS;
        if (MODE == 0) {
            const float* r = A.rays + unit * 11;
#pragma unroll
            for (int d = 0; d < 3; ++d) { ro[d] = r[d]; rd[d] = r[3 + d]; }
            nearv = r[6]; farv = r[7];
            dnorm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);   // torch.norm(rays_d) RUN:429
            if (tid < 32) {
                gdir[tid] = (tid < icv) ? enc_channel(r + 8, tid) : 0.f;
            }
            if (tid < K) {
                float* c = comp + tid * 8;
                c[0] = c[1] = c[2] = c[3] = c[4] = 0.f; c[5] = 1.f;
            }
        }
        const int n_chunks = (MODE == 0) ? chunks_per_ray : 1;
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            CFN_PHASE_ARGS;
            // global point index of row 0 and number of valid rows of this tile
            const int64_t p0 = (MODE == 0) ? unit * (int64_t)S + (int64_t)chunk * kTileM : unit * (int64_t)kTileM;
            const int rows_valid = (MODE == 0) ? min(kTileM, S - chunk * kTileM) : (int)min((int64_t)kTileM, A.P - p0);
            const int64_t tile_idx = (MODE == 0) ? unit * chunks_per_ray + chunk : unit;
            constexpr int kMbStride = (W / 32) * 64;        // mask words per (layer, tile)

            // ---- 1. sampling along the ray (RUN:510-534): z for rows 0..64, pts for rows 0..63
            if (MODE == 0) {
                const float* const a_zin = A.z_in;
                const float* const a_tv = A.t_vals;
                const float* const a_tr = A.t_rand;
                float* const a_pts = A.pts;
                float* const a_stz = A.st_z;
                const int a_flags = A.flags;
                fetched_together(a_zin, a_tv, a_tr, a_pts, a_stz, a_flags);
                if (tid <= kTileM) {
                    const int s = chunk * kTileM + tid;
                    float zv = 0.f, px = 0.f, py = 0.f, pz = 0.f;
                    if (s < S) {
                        // every input of this sample is requested before the first is used (round 4): written as nested conditionals the
                        // loads sat behind one another, each with its own full wait - four serialised memory round trips at every tile
                        // start, with 191 of the workgroup's 256 threads waiting at the barrier below.  Same arithmetic.
                        const int64_t si = unit * (int64_t)S + s;
                        const float tv0 = a_tv[s], tvp = a_tv[min(s + 1, S - 1)], tvm = a_tv[max(s - 1, 0)];
                        float trv = 0.f, zin = 0.f;
                        if (a_tr != nullptr) trv = a_tr[si];
                        if (a_zin != nullptr) zin = a_zin[si];
                        if (a_zin != nullptr) {
                            zv = zin;                                                              // explicit depths (extension)
                        } else {
                            const bool lind = (a_flags & CFNERF_F_LINDISP) != 0;
                            const float zc = zlin_f(tv0, nearv, farv, lind);
                            zv = zc;
                            if (a_tr != nullptr) {                                                 // RUN:518-532
                                const float upper = (s == S - 1) ? zc : 0.5f * (zlin_f(tvp, nearv, farv, lind) + zc);
                                const float lower = (s == 0) ? zc : 0.5f * (zc + zlin_f(tvm, nearv, farv, lind));
                                zv = lower + (upper - lower) * trv;
                            }
                        }
                        px = ro[0] + rd[0] * zv; py = ro[1] + rd[1] * zv; pz = ro[2] + rd[2] * zv;  // RUN:534
                        if (tid < kTileM) {
                            if (a_pts != nullptr) {
                                float* o = a_pts + (unit * (int64_t)S + s) * 3;
                                o[0] = px; o[1] = py; o[2] = pz;
                            }
                            if (a_stz != nullptr) a_stz[unit * (int64_t)S + s] = zv;
                        }
                    }
                    float* ri = rowinfo + tid * 4;
                    ri[0] = px; ri[1] = py; ri[2] = pz; ri[3] = zv;
                }
                __syncthreads();
            }
            // Biases are fetched ONE PHASE AHEAD of the accumulator initialisation that needs them (the accumulators must
            // hold them before a layer's first MFMA, so a fetch at the top of the layer is an exposed L2 round trip per layer):
            // layer 0's rides under the encoding, layer l+1's under layer l's MFMAs, and so on down the heads.
            float bias_n[C::NTW];
            // Operand-table entries travel ONE PHASE AHEAD as well (16 bytes = one s_load_dwordx4 each, cfnerf_device.h: kload): a trunk layer
            // runs on the entry fetched two layers earlier, prefetches the next layer's bias with the entry fetched one layer earlier, and
            // fetches the one after - so no layer starts by waiting for the constant cache in front of its first weight loads.
            SubL tl_cur = kload(T.trunk[0]);
            SubL tl_nxt = kload((1 < Dn) ? T.trunk[1] : T.ft);
            const SubL tl_skip = kload(T.skipseg);
            // ... and so do the stash destinations of the layer epilogues: this tile's rows of layer 0 + a per-layer stride, fetched here
            // (four constant-cache round trips per layer when each field is loaded at its first use behind the epilogue's barrier)
            float* const te_st0 = (A.st_h != nullptr) ? A.st_h + (size_t)p0 * W : nullptr;
            uint32_t* const te_mb0 = (A.st_mbits != nullptr) ? A.st_mbits + (size_t)tile_idx * kMbStride : nullptr;
            const size_t te_st_step = (size_t)A.P * W, te_mb_step = (size_t)A.n_tiles * kMbStride;
            load_bias<C::NTW>(tl_cur, wave, kWv, wp, bias_n);
            // ---- 2. positional encoding of the tile into act[:, 0:64)   (HLP:42-51, RUN:70-71)
            encode_tile<MODE, LD, PREC, kThr>(act, rowinfo, A.x, p0, rows_valid, ic, icv);
            // gamma(p) is needed again at the skip layer, five layers later, when the in-place tile has long been overwritten.
            // Re-evaluating it there cost ~800 vector instructions per wave and tile (30 correctly rounded sincosf per point);
            // instead the tile is parked as 16 KB of row-major fp32 - in the activation stash when there is one (the backward
            // wants it there anyway), else in this workgroup's slot of a small L2-resident scratch - and fetched back with
            // four 16-byte loads per thread.
            // The EVAL variants of the exact-fp32 path keep it in REGISTERS instead (16 per thread with 4 waves, 8 with 8): the scratch
            // slot was rewritten every tile and every one of those writes went through L2 to memory - 33 MB of WRITE_SIZE per C2
            // launch against 82 KB of outputs (profiles/r02_traffic.json).  (The split-bf16 eval kernel has no registers to spare and the
            // train variants want the tile in the stash anyway: they park it in memory as before.)
            constexpr bool kParkRegs = MODE == 0 && !TRAIN && PREC == PREC_F32;
            constexpr int kParkN = kTileM * 16 / kThr;
            f32x4 park[kParkRegs ? kParkN : 1];
            float* enc_park = (MODE == 0 && !kParkRegs) ? (A.st_enc != nullptr ? A.st_enc + p0 * 64 : A.enc_scratch + (size_t)blockIdx.x * (kTileM * 64)) : nullptr;
            if (kParkRegs) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < kParkN; ++i) {
                    const int idx = tid + i * kThr, row = idx >> 4, q = idx & 15;
                    park[i] = *reinterpret_cast<const f32x4*>(act + row * LD + 4 * q);
                }
            } else if (MODE == 0 || A.st_enc != nullptr) {
                __syncthreads();
                float* dst = (MODE == 0) ? enc_park : A.st_enc + p0 * 64;
                for (int idx = tid; idx < kTileM * 16; idx += kThr) {
                    const int row = idx >> 4, q = idx & 15;
                    if (row < rows_valid) {
                        f32x4 v;
                        const float* src = act + row * LD;
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = act_load<PREC>(src, LD, 4 * q + c);
                        if (A.st_enc != nullptr) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + row * 64 + 4 * q));
                        else *reinterpret_cast<f32x4*>(dst + row * 64 + 4 * q) = v;
                    }
                }
            }
            __syncthreads();
            // ---- 3. trunk: D x (Linear + ReLU), skip concat after layer D/2   (MOD:168-172)
            for (int l = 0; l < Dn; ++l) {
                CFN_PHASE_ARGS;
                f32x16 acc[2][C::NTW];
                const SubL tl_nn = kload((l + 2 < Dn) ? T.trunk[l + 2] : T.ft);                 // (in flight under this layer's MFMAs)
                acc_init(acc, bias_n);
                // the accumulators take the bias fetched a layer ago (its wait also drains the previous layer's stash stores: one counter);
                // only THEN is the next layer's bias requested - hoisted above that wait it would expose an L2 round trip per layer
                asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[0][C::NTW - 1][0]) : "memory");
                load_bias<C::NTW>(tl_nxt, wave, kWv, wp, bias_n);                               // next layer's / the feature head's
                mma_any<C::NTW, PREC, 2>(acc, tl_cur, wave, kWv, wp, wp16, act, LD);
                // (all four dwords of the prefetched entry stay live to here: the fp32 kernels never read w16_off, and a dead destination
                // register of an s_load in flight gets reused at once - a write-after-write wait on the load that was meant to be hidden)
                asm volatile("" :: "s"(tl_nn.w16_off));
                if (l >= 1 && l - 1 == skip_l) {
                    __syncthreads();                 // every wave is done reading h_{l-1}
                    if (kParkRegs) {                 // act[:, 0:64) <- gamma(p) again, from the registers it was kept in
#pragma unroll
                        for (int i = 0; i < kParkN; ++i) {
                            const int idx = tid + i * kThr, row = idx >> 4, q = idx & 15;
                            *reinterpret_cast<f32x4*>(act + row * LD + 4 * q) = park[i];
                        }
                    } else if (MODE == 0) {          // ... or from where the tile was parked
                        for (int idx = tid; idx < kTileM * 16; idx += kThr) {
                            const int row = idx >> 4, q = idx & 15;
                            f32x4 v; v[0] = v[1] = v[2] = v[3] = 0.f;           // rows past a ragged tile: finite filler
                            // nt load: served by L2, never by this CU's L1 (the scratch slot is rewritten every tile)
                            if (row < rows_valid) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(enc_park + row * 64 + 4 * q));
                            float* dstl = act + row * LD;
#pragma unroll
                            for (int c = 0; c < 4; ++c) act_store<PREC>(dstl, LD, 4 * q + c, v[c]);
                        }
                    } else {
                        encode_tile<MODE, LD, PREC, kThr>(act, rowinfo, A.x, p0, rows_valid, ic, icv);
                    }
                    __syncthreads();
                    mma_any<C::NTW, PREC, 2>(acc, tl_skip, wave, kWv, wp, wp16, act, LD);
                }
                __syncthreads();
                float* st = (te_st0 != nullptr) ? te_st0 + (size_t)l * te_st_step : nullptr;
                uint32_t* mb = (te_mb0 != nullptr) ? te_mb0 + (size_t)l * te_mb_step : nullptr;
                constexpr bool kRows = TRAIN && PREC == PREC_F32 && kStashFromLds && !Q4;      // stash by rows out of LDS (see stash_rows) ...
                store_tiles<C::NTW, ACT_RELU, PREC, TRAIN, TRAIN && !kRows, true, Q4>(acc, tl_cur, wave, kWv, wp, act, LD, 0, st, W, rows_valid, mb);      // ... or (Q4) as pieces from the registers
                __syncthreads();
                if (kRows && st != nullptr) stash_rows<W, kThr>(act, LD, st, rows_valid);
                tl_cur = tl_nxt; tl_nxt = tl_nn;
            }
            const SubL tl_ft = tl_cur;               // after the last layer: the feature head's entry

            // ---- 4. heads: h_alpha = A h (MOD:175), feature = F h (MOD:176).  h_alpha is only 1-2 n-tiles wide: its K is
            //         split over the 4 waves (a quarter or half each) and the partial tiles are summed through act[] once
            //         h is dead, instead of one wave grinding through the whole K while three wait.
            float bias_v[C::NTV];                    // views-layer bias, in flight during the heads
            {
                CFN_PHASE_ARGS;
                f32x16 accF[2][C::NTW];
                f32x16 accA[2][1];
                const SubL s_vf = kload(T.vf), s_ha = kload(T.ha);
                float* const st_ha = A.st_ha;
                float* const st_feat = A.st_feat;
                fetched_together(s_vf.w_off, s_ha.w_off, st_ha, st_feat);
                acc_init(accF, bias_n);
                load_bias<C::NTV>(s_vf, wave, kWv, wp, bias_v);
                if (Q4) {
                    // the feature head first: its stash leaves as Q4 pieces straight from the accumulators (bias already in them, no activation)
                    // BEFORE the h_alpha accumulators exist - inside the combined epilogue below those stores spilt 13 - 37 VGPRs (rounds 5 - 6)
                    mma_any<C::NTW, PREC, 2>(accF, tl_ft, wave, kWv, wp, wp16, act, LD);
                    stash_tiles_q4<C::NTW>(accF, tl_ft, wave, kWv, st_feat ? st_feat + p0 * W : nullptr, W);
                    acc_zero(accA);
                    mma_ksplit<PREC, 2>(accA, s_ha, wave, kWv, wp, wp16, act, LD);
                } else {
                    acc_zero(accA);
                    mma_ksplit<PREC, 2>(accA, s_ha, wave, kWv, wp, wp16, act, LD);
                    mma_any<C::NTW, PREC, 2>(accF, tl_ft, wave, kWv, wp, wp16, act, LD);
                }
                __syncthreads();                     // every wave is done reading h
                {
                    const int lo = lane_id_opaque();
                    float* lp = act + (4 * (lo >> 5)) * LD + 32 * wave + (lo & 31);      // raw fp32 partial of wave w: act[:, 32w..32w+32)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) lp[(i * 32 + (r & 3) + 8 * (r >> 2)) * LD] = accA[i][0][r];
                }
                __syncthreads();
                {
                    const int ntc = (int)s_ha.nt, nparts = kWv / ntc;
                    if (kThr % HA == 0) {
                        // (round 4) every thread owns ONE column (the thread count is a multiple of h_alpha_size: 32, 64, 128): its bias is
                        // fetched once, in front of the loop.  Element by element - `wp[b_off + c]` inside the loop - the compiler emitted
                        // global_load, s_waitcnt vmcnt(0), add, ds_write, store per element: a serialised L2 round trip (which also drains the
                        // previous element's stash store) 8 ... 32 times per tile, plus an integer division by the run-time HA each.
                        // Same element -> thread map, same order of the partial sums.
                        const int c = tid % HA, rstep = kThr / HA;
                        const float bc = wp[s_ha.b_off + c];
                        const float* pp0 = act + 32 * (c >> 5) + (c & 31);                   // wave w = part * ntc + n-tile
                        for (int row = tid / HA; row < kTileM; row += rstep) {
                            const float* pp = pp0 + row * LD;
                            float v = pp[0];
                            for (int q = 1; q < nparts; ++q) v += pp[32 * ntc * q];
                            v += bc;
                            act_store<PREC>(hs + row * HLD, HLD, c, v);
                            if (st_ha != nullptr && row < rows_valid) st_stream(st_ha + (p0 + row) * HA + c, v);
                        }
                    } else {
                        for (int idx = tid; idx < kTileM * HA; idx += kThr) {
                            const int row = idx / HA, c = idx - row * HA;
                            const float* pp = act + row * LD + 32 * (c >> 5) + (c & 31);
                            float v = pp[0];
                            for (int q = 1; q < nparts; ++q) v += pp[32 * ntc * q];
                            v += wp[s_ha.b_off + c];
                            act_store<PREC>(hs + row * HLD, HLD, c, v);
                            if (st_ha != nullptr && row < rows_valid) st_stream(st_ha + (p0 + row) * HA + c, v);
                        }
                    }
                }
                __syncthreads();
                // (Q4 variant: the feature stash has left from the registers above; here the tile only goes to LDS)
                constexpr bool kRows = TRAIN && PREC == PREC_F32 && kStashFromLds && !Q4;
                store_tiles<C::NTW, ACT_NONE, PREC, false, TRAIN && !kRows && !Q4, true>(accF, tl_ft, wave, kWv, wp, act, LD, 0,
                                              (st_feat && !Q4) ? st_feat + p0 * W : nullptr, W, rows_valid);
                __syncthreads();
                if (kRows && st_feat != nullptr) stash_rows<W, kThr>(act, LD, st_feat + p0 * W, rows_valid);
            }
            // ---- 5. views layer: v = relu(V [feature | gamma(d)])   (MOD:177-181)
            float bias_h[1];                         // h_rgb bias, in flight during the views layer
            {
                CFN_PHASE_ARGS;
                f32x16 acc[2][C::NTV];
                const SubL s_hr = kload(T.hr), s_vf = kload(T.vf), s_vd = kload(T.vd);
                float* const st_gd = A.st_gd;
                float* const st_v = A.st_v;
                fetched_together(s_hr.w_off, s_vf.w_off, s_vd.w_off, st_gd, st_v);
                acc_init(acc, bias_v);
                load_bias<1>(s_hr, wave, kWv, wp, bias_h);
                mma_any<C::NTV, PREC, 2>(acc, s_vf, wave, kWv, wp, wp16, act, LD);
                __syncthreads();
                for (int idx = tid; idx < kTileM * 32; idx += kThr) {
                    const int row = idx >> 5, c = idx & 31;
                    float v;
                    if (MODE == 0) v = gdir[c];
                    else v = (c < icv && row < rows_valid) ? A.x[(p0 + row) * (int64_t)(ic + icv) + ic + c] : 0.f;
                    act_store<PREC>(act + row * LD, LD, c, v);
                    if (st_gd != nullptr && row < rows_valid) st_stream(st_gd + (p0 + row) * 32 + c, v);
                }
                __syncthreads();
                mma_any<C::NTV, PREC, 2>(acc, s_vd, wave, kWv, wp, wp16, act, LD);
                __syncthreads();
                uint32_t* mb = (te_mb0 != nullptr) ? te_mb0 + (size_t)Dn * te_mb_step : nullptr;
                constexpr bool kRows = TRAIN && PREC == PREC_F32 && kStashFromLds && !Q4;
                store_tiles<C::NTV, ACT_RELU, PREC, TRAIN, TRAIN && !kRows, true, Q4>(acc, s_vf, wave, kWv, wp, act, LD, 0,
                                                    st_v ? st_v + p0 * (W / 2) : nullptr, W / 2, rows_valid, mb);
                __syncthreads();
                if (kRows && st_v != nullptr) stash_rows<W / 2, kThr>(act, LD, st_v + p0 * (W / 2), rows_valid);
            }
            // ---- 6. h_rgb = R v   (MOD:182)  -> act[:, W/2 : W/2 + HR)
            {
                CFN_PHASE_ARGS;
                f32x16 acc[2][1];
                const SubL s_hr = kload(T.hr);
                float* const st_hr = A.st_hr;
                fetched_together(s_hr.w_off, st_hr);
                acc_init(acc, bias_h);
                mma_any<1, PREC, 2>(acc, s_hr, wave, kWv, wp, wp16, act, LD);
                __syncthreads();
                store_tiles<1, ACT_NONE, PREC, false, TRAIN, true>(acc, s_hr, wave, kWv, wp, act, LD, W / 2,
                                         st_hr ? st_hr + p0 * HR : nullptr, HR, rows_valid);
                __syncthreads();
            }
            // ---- 7. amortised flow parameters (MOD:366-383), once per point (the reference recomputes
            //         them K times on duplicated rows, MOD:210-217): theta -> act[:, 0:128)
            {
                CFN_PHASE_ARGS;
                f32x16 acc[2][1];
                acc_zero(acc);
                const bool is_rgb = wave < 3;               // waves 0-2: the three rgb n-tiles; wave 3: alpha; others idle
                const bool is_theta = wave < 4;
                const SubL s_fr = kload(T.fr), s_fa = kload(T.fa);
                float* const st_theta = A.st_theta;
                fetched_together(s_fr.w_off, s_fa.w_off, st_theta);
                const float bv = wp[(is_rgb ? s_fr.b_off + wave * 32 : s_fa.b_off) + (lane_id_opaque() & 31)];   // lands under the MFMAs
                if (is_rgb)        mma_any<1, PREC, 2>(acc, s_fr, wave, kWv, wp, wp16, act, LD, W / 2);
                else if (is_theta) mma_any<1, PREC, 2>(acc, s_fa, 0, kWv, wp, wp16, hs, HLD);
                __syncthreads();
                const int colb = is_rgb ? wave * 32 : kThetaRgb;
                const int lo = lane_id_opaque();
                const int cl = lo & 31, rbase = 4 * (lo >> 5);
                const int c_local = is_rgb ? colb + cl : cl;          // column inside the rgb / alpha head block
                const int F = 4;
                const bool tanh_col = is_rgb ? (c_local >= 9 * F && c_local < 15 * F) : (c_local < 2 * F);
                float* lp = act + rbase * LD + colb + cl;
                // stash through a slab descriptor over the tile's valid rows (scalar row offsets, ragged rows dropped by the bounds
                // check): one code path, no 64-bit address arithmetic per element (quarter-rate on this chip)
                const bool stash = TRAIN && st_theta != nullptr;                // wave-uniform
                const __amdgpu_buffer_rsrc_t sink = slab_rsrc(stash ? st_theta + p0 * kThetaAll : nullptr, stash ? rows_valid : 0, kThetaAll);
                const int voff = (rbase * kThetaAll + colb + cl) * 4;
                if (is_theta) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                            float v = acc[i][0][r] + bv;
                            if (tanh_col) v = tanhf(v);               // diag_activation, MOD:337-348
                            lp[rr * LD] = v;
                            if (TRAIN) slab_store(sink, voff, rr * kThetaAll * 4, v);      // (no stash: zero-size descriptor, every store dropped)
                        }
                }
                __syncthreads();
            }
            // ---- 8. flows + composite: lane = sample (row), waves stride over the K latent samples
            {
                CFN_PHASE_ARGS;
                const float* const f_eps = A.eps;
                float* const f_raw = A.raw;
                float* const f_weights = A.weights;
                // this tile's block of the transposed stash pieces: [k][64 rows][4] / [k][64 rows][2]
                float* const f_sraw = (A.st_raw != nullptr) ? A.st_raw + (size_t)tile_idx * K * (kTileM * 4) : nullptr;
                float* const f_at = (A.st_at != nullptr) ? A.st_at + (size_t)tile_idx * K * (kTileM * 2) : nullptr;
                const int f_flags = A.flags;
                fetched_together(f_eps, f_raw, f_weights, f_sraw, f_at, f_flags);
                const int row = lane_id_opaque();
                const bool valid = row < rows_valid;
                float th[84];
                auto load_th = [&]() {          // (the one-latent-at-a-time path keeps the point's flow parameters in registers over its loop)
                    const f32x4* tp = reinterpret_cast<const f32x4*>(act + row * LD);
#pragma unroll
                    for (int q = 0; q < 21; ++q) {
                        const f32x4 v = tp[q < 18 ? q : kThetaRgb / 4 + (q - 18)];
                        th[q * 4 + 0] = v[0]; th[q * 4 + 1] = v[1]; th[q * 4 + 2] = v[2]; th[q * 4 + 3] = v[3];
                    }
                };
                float zval = 0.f, dist = 0.f;
                if (MODE == 0) {
                    zval = rowinfo[row * 4 + 3];
                    const int s = chunk * kTileM + row;
                    const float dz = (s == S - 1) ? 1e1f : rowinfo[(row + 1) * 4 + 3] - zval;      // RUN:426-427
                    dist = dz * dnorm;                                                             // RUN:429
                }
                // the K flows + the composite of this tile in one of two arithmetic flavours (see Num<FAST> in cfnerf_device.h)
                auto flow_phase = [&](auto fast_tag) {
                    constexpr bool FAST = decltype(fast_tag)::value;
                    using M = Num<FAST>;
                    auto one = [&](const int k) {
                        const f32x4 e = *reinterpret_cast<const f32x4*>(f_eps + k * 4);
                        float z[3] = {e[0] * r_std[0] + r_mean[0], e[1] * r_std[1] + r_mean[1], e[2] * r_std[2] + r_mean[2]};  // MOD:206/251
                        float a = e[3] * a_std + a_mean;                                               // MOD:200/239
                        float ldr, lda;
                        flows_fwd<TRAIN, FAST>(th, z, a, ldr, lda);
                        if (f_raw != nullptr || f_sraw != nullptr) {
                            f32x4 o; o[0] = z[0]; o[1] = z[1]; o[2] = z[2]; o[3] = a;
                            if (f_raw != nullptr && valid) *reinterpret_cast<f32x4*>(f_raw + ((p0 + row) * (int64_t)K + k) * 4) = o;  // MOD:221/289
                            if (f_sraw != nullptr) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(f_sraw + (k * kTileM + row) * 4));   // (rows past a ragged tile: finite filler, never read)
                        }
                        const float sp_a = M::softplus(a);
                        if (TRAIN && valid) {
                            ent_a_sum += lda + (a - sp_a);                                                        // MOD:263
                            ent_r_sum += ldr + (((z[0] + z[1]) + z[2]) - 2.f * ((M::softplus(z[0]) + M::softplus(z[1])) + M::softplus(z[2])));  // MOD:278
                        }
                        if (MODE == 0) {
                            const float ea = valid ? M::exp(-sp_a * dist) : 1.f;
                            const float alpha = 1.f - ea;                                              // RUN:424,442
                            const float xk = (1.f - alpha) + 1e-10f;                                   // RUN:443
                            float incl, excl;
                            comp_scan_mul(xk, incl, excl);
                            float* cp = comp + k * 8;
                            const float Tcar = cp[5];
                            const float wgt = alpha * (Tcar * excl);
                            if (f_weights != nullptr && valid) f_weights[(p0 + row) * (int64_t)K + k] = wgt;
                            if (f_at != nullptr) {
                                f32x2 at; at[0] = ea; at[1] = Tcar * excl;      // (e, T): the backward takes d alpha / d sigma from e itself
                                __builtin_nontemporal_store(at, reinterpret_cast<f32x2*>(f_at + (k * kTileM + row) * 2));
                            }
                            const float s0 = comp_sum(wgt * M::sigmoid(z[0]));                         // RUN:431,444
                            const float s1 = comp_sum(wgt * M::sigmoid(z[1]));
                            const float s2 = comp_sum(wgt * M::sigmoid(z[2]));
                            const float sd = comp_sum(wgt * zval);                                     // RUN:447
                            const float sa = comp_sum(wgt);                                            // RUN:449
                            const float tot = comp_last(incl);
                            if (lane == 0) {
                                cp[0] += s0; cp[1] += s1; cp[2] += s2; cp[3] += sd; cp[4] += sa; cp[5] = Tcar * tot;
                            }
                        }
                    };
                    // two latents of the wave at a time (hardware-transcendental flavour, i.e. from 16 latents on): the plain multiplies and adds
                    // of the pair go out as packed fp32 instructions, the wave scans / sums of the two composites interleave.  Component
                    // for component the operations of one(): the same values, summed into the entropy terms in the same order.
                    auto two = [&](const int k0, const int k1) {
                        using M2 = Num2;
                        const f32x4 e0 = *reinterpret_cast<const f32x4*>(f_eps + k0 * 4), e1 = *reinterpret_cast<const f32x4*>(f_eps + k1 * 4);
                        f32x2 z[3], a;
#pragma unroll
                        for (int c = 0; c < 3; ++c) { f32x2 ec; ec[0] = e0[c]; ec[1] = e1[c]; z[c] = ec * r_std[c] + r_mean[c]; }
                        { f32x2 ec; ec[0] = e0[3]; ec[1] = e1[3]; a = ec * a_std + a_mean; }
                        // the point's 84 flow parameters as register PAIRS, read from its LDS row in every iteration: a packed instruction takes either
                        // half of a pair for both components (op_sel) only when the pair is formed in the loop body - hoisted out of the loop the
                        // compiler keeps a broadcast copy (x, x) of every parameter, 168 registers, and spills.  21 conflict-free ds_read_b128 per ~460
                        // vector instructions, on a pipe the flow phase does not otherwise use
                        f32x2 thp[42];
                        {
                            asm volatile("" ::: "memory");          // the reads below belong to THIS iteration (nothing for the compiler to hoist)
                            const f32x4* tp = reinterpret_cast<const f32x4*>(act + row * LD);
#pragma unroll
                            for (int q = 0; q < 21; ++q) {
                                const f32x4 v = tp[q < 18 ? q : kThetaRgb / 4 + (q - 18)];
                                thp[q * 2 + 0] = __builtin_shufflevector(v, v, 0, 1); thp[q * 2 + 1] = __builtin_shufflevector(v, v, 2, 3);
                            }
                        }
                        f32x2 ldr, lda;
                        flows_fwd2<TRAIN>(thp, z, a, ldr, lda);
                        if (f_raw != nullptr || f_sraw != nullptr) {
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                const int k = c ? k1 : k0;
                                f32x4 o; o[0] = z[0][c]; o[1] = z[1][c]; o[2] = z[2][c]; o[3] = a[c];
                                if (f_raw != nullptr && valid) *reinterpret_cast<f32x4*>(f_raw + ((p0 + row) * (int64_t)K + k) * 4) = o;
                                if (f_sraw != nullptr) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(f_sraw + (k * kTileM + row) * 4));
                            }
                        }
                        const f32x2 sp_a = M2::softplus(a);
                        if (TRAIN && valid) {
                            const f32x2 ta = lda + (a - sp_a);
                            const f32x2 tr = ldr + (((z[0] + z[1]) + z[2]) - 2.f * ((M2::softplus(z[0]) + M2::softplus(z[1])) + M2::softplus(z[2])));
                            ent_a_sum += ta[0]; ent_a_sum += ta[1];
                            ent_r_sum += tr[0]; ent_r_sum += tr[1];
                        }
                        if (MODE == 0) {
                            f32x2 ea = M2::exp(-sp_a * dist);
                            if (!valid) ea = 1.f;
                            const f32x2 alpha = 1.f - ea;
                            const f32x2 xk = (1.f - alpha) + 1e-10f;
                            const f32x2 g0 = M2::sigmoid(z[0]), g1 = M2::sigmoid(z[1]), g2 = M2::sigmoid(z[2]);
                            f32x2 incl, excl;
                            { float i0, x0, i1, x1; comp_scan_mul(xk[0], i0, x0); comp_scan_mul(xk[1], i1, x1); incl[0] = i0; incl[1] = i1; excl[0] = x0; excl[1] = x1; }
                            float* cp0 = comp + k0 * 8; float* cp1 = comp + k1 * 8;
                            f32x2 Tcar; Tcar[0] = cp0[5]; Tcar[1] = cp1[5];
                            const f32x2 Tex = Tcar * excl;
                            const f32x2 wgt = alpha * Tex;
                            if (f_weights != nullptr && valid) {
                                f_weights[(p0 + row) * (int64_t)K + k0] = wgt[0];
                                f_weights[(p0 + row) * (int64_t)K + k1] = wgt[1];
                            }
                            if (f_at != nullptr) {
#pragma unroll
                                for (int c = 0; c < 2; ++c) {
                                    f32x2 at; at[0] = ea[c]; at[1] = Tex[c];
                                    __builtin_nontemporal_store(at, reinterpret_cast<f32x2*>(f_at + ((c ? k1 : k0) * kTileM + row) * 2));
                                }
                            }
                            const f32x2 w0 = wgt * g0, w1 = wgt * g1, w2 = wgt * g2, wd = wgt * zval;
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                const float s0 = comp_sum(w0[c]), s1 = comp_sum(w1[c]), s2 = comp_sum(w2[c]);
                                const float sd = comp_sum(wd[c]), sa = comp_sum(wgt[c]);
                                const float tot = comp_last(incl[c]);
                                float* cp = c ? cp1 : cp0;
                                if (lane == 0) {
                                    cp[0] += s0; cp[1] += s1; cp[2] += s2; cp[3] += sd; cp[4] += sa; cp[5] = Tcar[c] * tot;
                                }
                            }
                        }
                    };
                    if constexpr (FAST) {
                        int k = wave;
                        for (; k + kWv < K; k += 2 * kWv) two(k, k + kWv);
                        if (k < K) { load_th(); one(k); }
                    } else {
                        load_th();
                        for (int k = wave; k < K; k += kWv) one(k);
                    }
                };
                const bool fast = (f_flags & CFNERF_F_FLOW_MATH_SET) ? (f_flags & CFNERF_F_FLOW_MATH_FAST) != 0 : K >= kFastFlowsK;     // wave-uniform
                if (fast) flow_phase(std::true_type{});
                else flow_phase(std::false_type{});
            }
            __syncthreads();
        }  // chunks

        {   // ---- 9. the ray's outputs
        CFN_PHASE_ARGS;
        if (MODE == 0 && tid < K) {
            float* cp = comp + tid * 8;
            const int k = tid;
            float r0 = cp[0], r1 = cp[1], r2 = cp[2];
            const float depth = cp[3], acc = cp[4];
            if (A.flags & CFNERF_F_WHITE_BKGD) { r0 = r0 + (1.f - acc); r1 = r1 + (1.f - acc); r2 = r2 + (1.f - acc); }  // RUN:451-452
            const float disp = 1.f / fmaxf(1e-10f + 1e-10f, depth / (acc + 1e-10f) + 1e-10f);                    // RUN:448
            if (A.rgb_map != nullptr) {
                float* o = A.rgb_map + unit * 3 * (int64_t)K;
                o[0 * K + k] = r0; o[1 * K + k] = r1; o[2 * K + k] = r2;                                           // [N,3,K] RUN:445
                A.disp[unit * (int64_t)K + k] = disp;
                A.depth[unit * (int64_t)K + k] = depth;
            }
            cp[0] = r0; cp[1] = r1; cp[2] = r2; cp[6] = disp;
        }
        if (MODE == 0 && A.kstats != nullptr) {
            // fused reductions over the K latent samples (RUN:1122-1131): mean, np.std * n/(n-1)
            __syncthreads();
            if (tid < 5) {
                const int c = (tid < 3) ? tid : (tid == 3 ? 6 : 3);       // r, g, b, disp, depth slots of comp[k]
                float mean = 0.f;
                for (int k = 0; k < K; ++k) mean += comp[k * 8 + c];
                mean /= (float)K;
                float* o = A.kstats + unit * 8;
                if (tid < 3) {
                    float var = 0.f;
                    for (int k = 0; k < K; ++k) { const float d = comp[k * 8 + c] - mean; var += d * d; }
                    o[tid] = mean;
                    o[3 + tid] = sqrtf(var / (float)K) * (float)K / (float)(K - 1);
                    if (A.sqerr != nullptr) {                                  // the integrand of img2mse(rgb_mean, target), RUN:1028 / HLP:15
                        const float e = mean - A.gt[unit * 3 + tid];
                        A.sqerr[unit * 3 + tid] = e * e;
                    }
                } else {
                    o[6 + (tid - 3)] = mean;
                }
            }
        }
        }
        __syncthreads();
    }  // units

    if (TRAIN) {
      CFN_PHASE_ARGS;
      if (A.ent_partials != nullptr) {
        const float sr = wave_sum(ent_r_sum), sa = wave_sum(ent_a_sum);
        if (lane == 0) { red[wave * 2] = sr; red[wave * 2 + 1] = sa; }
        __syncthreads();
        if (tid == 0) {
            float e0 = (red[0] + red[2]) + (red[4] + red[6]), e1 = (red[1] + red[3]) + (red[5] + red[7]);
            if (kWv == 8) {
                e0 += (red[8] + red[10]) + (red[12] + red[14]);
                e1 += (red[9] + red[11]) + (red[13] + red[15]);
            }
            A.ent_partials[blockIdx.x * 2 + 0] = e0;
            A.ent_partials[blockIdx.x * 2 + 1] = e1;
        }
      }
    }
#undef CFN_PHASE_ARGS
#undef CFN_PHASE_LOCALS
#undef CFN_KARGS
}

// loss_entropy = mean(base_a) - mean(ld_a) + mean(base_rgb) - mean(ld_rgb)      (MOD:268,283,286)
// Blocks past the first are a copy engine: the backward reads the model's OWN copies of the step's rays and latents (the caller may
// drop its tensors), and as blocks of this launch the two copies cost no launches of their own (they were two hipMemcpyAsync, ~4 us
// of stream time each, in front of every forward).
__global__ void entropy_finalize_kernel(const float* partials, int n_part, const float* flat, const float* eps,
                                        int K, double count /* P*K */, float* out, float* eps_keep, const float* rays, float* rays_keep,
                                        int64_t n_rays_floats) {
    if (blockIdx.x > 0) {
        const int64_t i = ((int64_t)blockIdx.x - 1) * blockDim.x + threadIdx.x;
        if (i < n_rays_floats) rays_keep[i] = rays[i];
        return;
    }
    if (eps_keep != nullptr)
        for (int i = threadIdx.x; i < K * 4; i += blockDim.x) eps_keep[i] = eps[i];
    __shared__ double sh[2][256];
    double s0 = 0, s1 = 0;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) { s0 += partials[2 * i]; s1 += partials[2 * i + 1]; }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { sh[0][threadIdx.x] += sh[0][threadIdx.x + d]; sh[1][threadIdx.x] += sh[1][threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float a_mean = flat[0], a_std = flat[1];
        float base_a = 0.f, base_r = 0.f;
        for (int k = 0; k < K; ++k) {
            const float a0 = eps[k * 4 + 3] * a_std + a_mean;
            base_a += -0.5f * (logf(a_std) * 2.f + (a0 - a_mean) * (a0 - a_mean) * (1.f / (a_std * a_std)));
            for (int c = 0; c < 3; ++c) {
                const float m = flat[2 + c], sd = flat[5 + c];
                const float r0 = eps[k * 4 + c] * sd + m;
                base_r += -0.5f * (logf(sd) * 2.f + (r0 - m) * (r0 - m) * (1.f / (sd * sd)));
            }
        }
        const double ent = (double)base_a / K - sh[1][0] / count + (double)base_r / (3.0 * K) - sh[0][0] / count;
        out[0] = (float)ent;
    }
}

// ---------------------------------------------------------------------------------------------
// standalone composite (raw2outputs RUN:411-454): one wave per ray, lane = sample.
// Reads 16*S*K + 4*S + 12 bytes per ray, writes 20*K (+4*S*K for weights).  `raw [N,S,K,4]` has the latent index inside the sample
// index, so "lane = sample, one k at a time" (rounds 1-3) read 16 bytes per lane at a stride of 16 K bytes and came back K times for
// the rest of every cache line (0.9 TB/s at K = 32).  Now a wave takes KG latents at a time: the [64 samples][KG latents] block of a
// chunk arrives by LDS-DMA as KG fully coalesced 1-KB pieces into the wave's own LDS block (CompStage, cfnerf_device.h: no staging
// registers, conflict-free transposed reads) and the KG latents are walked out of LDS with the arithmetic of rounds 1-3 (= the fused
// kernel's composite) operation for operation; a latent's carry and sums wait in LDS between the chunks (entry = lane), so the maps
// leave as one store per output.
// Transcendentals like the fused kernels: libm below kFastFlowsK latents (there the kernel is bound by libm's instruction count,
// not by memory), the hardware forms from there on (Num<FAST>).
constexpr int kCompKB = 64;       // latents of one pass over a ray's chunks (their carries and sums wait in LDS, entry = lane)
template <int KG, bool FAST>
__global__ __launch_bounds__(kThreads, FAST ? 4 : 3)      // (the LDS block leaves room for 4 workgroups per CU: keep the registers there too)
void composite_kernel(const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays_d,
                      int64_t N, int S, int K, int white_bkgd, float* rgb_map, float* disp_map, float* depth_map,
                      float* weights) {
    using M = Num<FAST>;
    using St = CompStage<KG>;
    constexpr int U = (KG == 8) ? 4 : 1;                     // groups per round of the weights turn-around (32 latents = one 128-byte line per sample)
    __shared__ __attribute__((aligned(16))) float stage_all[kWaves][St::kQuads * 4];
    __shared__ float sums_all[kWaves][6][kCompKB];           // [0] transmittance entering the next chunk, [1..5] rgb / depth / acc sums
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * kWaves + wave;
    if (ray >= N) return;                                    // (no workgroup barrier below: a wave leaves alone)
    float* stage = stage_all[wave];
    float (*sums)[kCompKB] = sums_all[wave];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)stage);
    const float* d = rays_d + ray * 3;
    const float dnorm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zr = z_vals + ray * (int64_t)S;
    const i32x4 rsrc = ds_rsrc(raw + ray * (int64_t)S * K * 4, S * K * 16);      // samples past S arrive as zeros (bounds check)
    const int nch = (S + 63) / 64;
    const bool vec_w = weights != nullptr && (K & 3) == 0;   // a lane's KG weights of a sample as 16-byte stores
    unsigned voff[KG];
#pragma unroll
    for (int j = 0; j < KG; ++j) voff[j] = St::piece_voff(lane, j, K);
    const int my_q0 = lane * KG, my_swz = St::swz(lane);
    // Chunk-major: all latents of a chunk before the next chunk, so that raw streams through once front to back and the KG-float
    // pieces a group adds to a sample's row of `weights` complete their cache lines while these are still in L2.
    for (int kb = 0; kb < K; kb += kCompKB) {
        const int kn = min(K - kb, kCompKB);
        sums[0][lane] = 1.f;
#pragma unroll
        for (int c = 1; c < 6; ++c) sums[c][lane] = 0.f;
        for (int ch = 0; ch < nch; ++ch) {
            const int s = ch * 64 + lane;
            const bool valid = s < S;
            float zv = 0.f, dist = 0.f;
            if (valid) {
                zv = zr[s];
                const float dz = (s == S - 1) ? 1e1f : zr[s + 1] - zv;
                dist = dz * dnorm;
            }
            for (int gf = 0; gf < kn; gf += KG * U) {
                float wrow_keep[KG * U];                     // (U > 1) this sample's weights of up to 32 latents, for full-line stores
#pragma unroll
                for (int i = 0; i < KG * U; ++i) wrow_keep[i] = 0.f;
#pragma unroll
                for (int gq = 0; gq < U; ++gq) {
                const int gi = gf + gq * KG;
                if (gi >= kn) continue;                      // (uniform)
                const int g0 = kb + gi;
                St::fetch(rsrc, lds0, voff, ch, g0, K);
                // lane q < KG: latent g0 + q
                const int slot = gi + (lane < KG ? lane : 0);
                float carv = sums[0][slot], acc0 = sums[1][slot], acc1 = sums[2][slot], acc2 = sums[3][slot], accd = sums[4][slot], acca = sums[5][slot];
                St::landed();
                float wv[KG];
#pragma unroll
                for (int q = 0; q < KG; ++q) {
                    wv[q] = 0.f;
                    if (gi + q < kn) {                       // (uniform)
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(stage + (my_q0 + (q ^ my_swz)) * 4);
                        const float car = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(carv), q));
                        const float alpha = valid ? 1.f - M::exp(-M::softplus(rv[3]) * dist) : 0.f;
                        const float xk = (1.f - alpha) + 1e-10f;
                        float incl, excl;
                        comp_scan_mul(xk, incl, excl);
                        const float wgt = alpha * (car * excl);
                        wv[q] = wgt;
                        const float s0 = comp_sum(wgt * M::sigmoid(rv[0])), s1 = comp_sum(wgt * M::sigmoid(rv[1])), s2 = comp_sum(wgt * M::sigmoid(rv[2]));
                        const float sd = comp_sum(wgt * zv), sa = comp_sum(wgt);
                        const float carn = car * comp_last(incl);
                        if (lane == q) { acc0 += s0; acc1 += s1; acc2 += s2; accd += sd; acca += sa; carv = carn; }   // (chunk by chunk like the fused kernel)
                    }
                }
                if (lane < KG) { sums[0][slot] = carv; sums[1][slot] = acc0; sums[2][slot] = acc1; sums[3][slot] = acc2; sums[4][slot] = accd; sums[5][slot] = acca; }
                if (U > 1 && vec_w) {
#pragma unroll
                    for (int q = 0; q < KG; ++q) wrow_keep[gq * KG + q] = wv[q];
                } else if (weights != nullptr && valid) {
                    float* wrow = weights + (ray * S + s) * (int64_t)K + g0;
                    if (vec_w) {                             // (K <= 4: a sample's row is one 16-byte store, rows adjacent)
#pragma unroll
                        for (int q = 0; q < KG; q += 4)
                            if (gi + q < kn) { f32x4 o; o[0] = wv[q]; o[1] = wv[q + 1]; o[2] = wv[q + 2]; o[3] = wv[q + 3]; *reinterpret_cast<f32x4*>(wrow + q) = o; }
                    } else {
#pragma unroll
                        for (int q = 0; q < KG; ++q)
                            if (gi + q < kn) wrow[q] = wv[q];
                    }
                }
                }
                if (U > 1 && vec_w) {
                    // Up to 32 latents = 128 bytes per sample: turned around in the (now free) LDS block so that 8 lanes write one sample's
                    // full line - a lane's own 16- or 32-byte pieces at a pitch of 4 K bytes took 414 instead of ~290 us at K = 32
                    // (a timing-only build with a contiguous layout).  Quad kq of sample sl sits at sl * 8 + (kq ^ swz(sl)).
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    wave_lds_turn();
#pragma unroll
                    for (int kq = 0; kq < 8; ++kq) {
                        f32x4 o; o[0] = wrow_keep[kq * 4]; o[1] = wrow_keep[kq * 4 + 1]; o[2] = wrow_keep[kq * 4 + 2]; o[3] = wrow_keep[kq * 4 + 3];
                        *reinterpret_cast<f32x4*>(stage + (lane * 8 + (kq ^ my_swz)) * 4) = o;
                    }
                    wave_lds_turn();
                    float* wblk = weights + (ray * S + (int64_t)ch * 64) * K + kb + gf;
                    f32x4 ov[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int sl = j * 8 + (lane >> 3), kq = lane & 7;
                        ov[j] = *reinterpret_cast<const f32x4*>(stage + (sl * 8 + (kq ^ St::swz(sl))) * 4);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int sl = j * 8 + (lane >> 3), kq = lane & 7;
                        if (ch * 64 + sl < S && gf + kq * 4 < kn) *reinterpret_cast<f32x4*>(wblk + (int64_t)sl * K + kq * 4) = ov[j];
                    }
                }
            }
        }
        wave_lds_turn();
        if (lane < kn) {                                     // lane l: latent kb + l
            const int k = kb + lane;
            float r0 = sums[1][lane], r1 = sums[2][lane], r2 = sums[3][lane];
            const float rd = sums[4][lane], ra = sums[5][lane];
            if (white_bkgd) { r0 = r0 + (1.f - ra); r1 = r1 + (1.f - ra); r2 = r2 + (1.f - ra); }
            float* o = rgb_map + ray * 3 * (int64_t)K;
            o[0 * K + k] = r0; o[1 * K + k] = r1; o[2 * K + k] = r2;
            disp_map[ray * (int64_t)K + k] = 1.f / fmaxf(1e-10f + 1e-10f, rd / (ra + 1e-10f) + 1e-10f);
            depth_map[ray * (int64_t)K + k] = rd;
        }
        wave_lds_turn();
    }
}


// ---------------------------------------------------------------------------------------------
// ray set-up (render() RUN:129-158; get_rays HLP:288-297; ndc_rays HLP:360-377)
__global__ void rays_setup_kernel(int H, int Wd, float focal, RaysC2W c2w, int use_c2w, const float* rays_o,
                                  const float* rays_d, int64_t N, int64_t pixel0, int ndc, float nearv, float farv, float* out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float o[3], d[3];
    if (use_c2w) {
        const int64_t pix = pixel0 + n;
        const int j = (int)(pix / Wd), i = (int)(pix - (int64_t)j * Wd);             // row-major pixels, i = x (HLP:289-291)
        const float dir[3] = {((float)i - (float)Wd * .5f) / focal, -((float)j - (float)H * .5f) / focal, -1.f};   // HLP:292
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            d[r] = (dir[0] * c2w.m[r * 4 + 0] + dir[1] * c2w.m[r * 4 + 1]) + dir[2] * c2w.m[r * 4 + 2];           // HLP:294
            o[r] = c2w.m[r * 4 + 3];                                                                              // HLP:296
        }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) { o[r] = rays_o[n * 3 + r]; d[r] = rays_d[n * 3 + r]; }
    }
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float vd[3] = {d[0] / nrm, d[1] / nrm, d[2] / nrm};                                                     // RUN:143
    if (ndc) {
        const float nr = 1.f;                                                                                     // RUN:149
        const float t = -(nr + o[2]) / d[2];                                                                      // HLP:362
        o[0] = o[0] + t * d[0]; o[1] = o[1] + t * d[1]; o[2] = o[2] + t * d[2];                                   // HLP:363
        const float cw = -1.f / ((float)Wd / (2.f * focal)), chh = -1.f / ((float)H / (2.f * focal));
        const float o0 = cw * o[0] / o[2], o1 = chh * o[1] / o[2], o2 = 1.f + 2.f * nr / o[2];                    // HLP:366-368
        const float d0 = cw * (d[0] / d[2] - o[0] / o[2]);                                                        // HLP:370
        const float d1 = chh * (d[1] / d[2] - o[1] / o[2]);
        const float d2 = -2.f * nr / o[2];
        o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
    }
    float* r = out + n * 11;
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = nearv; r[7] = farv; r[8] = vd[0]; r[9] = vd[1]; r[10] = vd[2];
}

// standalone ndc_rays (HLP:360-377) with the caller's near plane: o, d [N,3] -> o', d' [N,3]
__global__ void ndc_rays_kernel(int H, int Wd, float focal, float nr, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                int64_t N, float* __restrict__ out_o, float* __restrict__ out_d) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float o[3], d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[r] = rays_o[n * 3 + r]; d[r] = rays_d[n * 3 + r]; }
    const float t = -(nr + o[2]) / d[2];                                                                          // HLP:362
    o[0] = o[0] + t * d[0]; o[1] = o[1] + t * d[1]; o[2] = o[2] + t * d[2];                                       // HLP:363
    const float cw = -1.f / ((float)Wd / (2.f * focal)), chh = -1.f / ((float)H / (2.f * focal));
    out_o[n * 3 + 0] = cw * o[0] / o[2]; out_o[n * 3 + 1] = chh * o[1] / o[2]; out_o[n * 3 + 2] = 1.f + 2.f * nr / o[2];   // HLP:366-368
    out_d[n * 3 + 0] = cw * (d[0] / d[2] - o[0] / o[2]);                                                          // HLP:370-372
    out_d[n * 3 + 1] = chh * (d[1] / d[2] - o[1] / o[2]);
    out_d[n * 3 + 2] = -2.f * nr / o[2];
}

hipError_t launch_ndc_rays(int H, int Wd, float focal, float nearv, const float* ro, const float* rd, int64_t N, float* out_o, float* out_d,
                           hipStream_t st) {
    hipLaunchKernelGGL(ndc_rays_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, H, Wd, focal, nearv, ro, rd, N, out_o, out_d);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// standalone positional encoding (HLP:21-69) and ray sampling (RUN:510-534): the unfused boundary functions
// One thread per (point, frequency, coordinate): ONE sincosf (the libm routine of the fused kernel's encode_tile: fused and unfused
// encodings are bit-identical) gives the sin and the cos channel of that argument - rounds 1-3 ran one thread per output channel, i.e.
// the routine twice per argument, and the kernel is bound by its ~100 instructions, not by the 264 bytes per point it moves.
// Unit u of a point: u < 3 the identity channels, then u = 3 (f + 1) + d  ->  channels 3 + 6 f + d (sin) and 3 + 6 f + 3 + d (cos)  (HLP:42-51).
__global__ void embed_kernel(const float* __restrict__ x, int64_t P, int multires, float* __restrict__ out) {
    const int units = 3 * (multires + 1), nch = 3 + 6 * multires;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * units) return;
    const int64_t p = idx / units;
    const int u = (int)(idx - p * units);
    const int f = u / 3 - 1, d = u - 3 * (f + 1);
    const float v = x[p * 3 + d];
    float* o = out + p * nch;
    if (f < 0) { o[d] = v; return; }
    float sv, cv;
    sincosf(v * (float)(1 << f), &sv, &cv);                   // 2^f * v is exact in fp32
    o[3 + 6 * f + d] = sv;
    o[3 + 6 * f + 3 + d] = cv;
}

// One wave per ray, lane = sample, 64-sample chunks: no per-thread 64-bit division, the ray's 8 numbers are
// scalar loads, t_rand and z move as coalesced rows, and a chunk's 768 bytes of `pts` leave as three coalesced stores - element f of
// the block is coordinate f % 3 of sample f / 3, whose depth comes from that sample's lane through the wave's LDS row (same
// `o + d * z`, RUN:534).  (Rounds 1-3: one thread per sample, three 4-byte stores at a stride of 12 bytes.)
__global__ __launch_bounds__(kThreads)
void sample_points_kernel(const float* __restrict__ rays, const float* __restrict__ t_vals, const float* __restrict__ t_rand,
                          int flags, int64_t N, int S, float* __restrict__ z_out, float* __restrict__ pts) {
    __shared__ float zs_all[kWaves][64];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* zs = zs_all[wave];
    const int nch = (S + 63) / 64;
    const bool lind = (flags & CFNERF_F_LINDISP) != 0;
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };   // wave-uniform value in an SGPR
    for (int64_t n = (int64_t)blockIdx.x * kWaves + wave; n < N; n += (int64_t)gridDim.x * kWaves) {     // (no workgroup barrier below)
        const float* r = rays + n * 11;
        const float o0 = uni(r[0]), o1 = uni(r[1]), o2 = uni(r[2]), d0 = uni(r[3]), d1 = uni(r[4]), d2 = uni(r[5]);
        const float nearv = uni(r[6]), farv = uni(r[7]);
        for (int ch = 0; ch < nch; ++ch) {
            const int s = ch * 64 + lane;
            const bool valid = s < S;
            // every operand requested before the first is used (clamped neighbours: the selects below take what rounds 1-3 computed)
            const int sc = min(s, S - 1);
            const float tc = t_vals[sc], tp = t_vals[min(sc + 1, S - 1)], tm = t_vals[max(sc - 1, 0)];
            const float tr = (t_rand != nullptr) ? t_rand[n * S + sc] : 0.f;
            const float zc = zlin_f(tc, nearv, farv, lind);
            float zv = zc;
            if (t_rand != nullptr) {
                const float upper = (sc == S - 1) ? zc : 0.5f * (zlin_f(tp, nearv, farv, lind) + zc);
                const float lower = (sc == 0) ? zc : 0.5f * (zc + zlin_f(tm, nearv, farv, lind));
                zv = lower + (upper - lower) * tr;
            }
            if (valid) z_out[n * S + s] = zv;
            if (pts != nullptr) {
                wave_lds_turn();
                zs[lane] = zv;
                wave_lds_turn();
                float* prow = pts + (n * S + (int64_t)ch * 64) * 3;
                float pv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int f = c * 64 + lane, sl = f / 3, comp = f - 3 * sl;
                    const float o = comp == 0 ? o0 : (comp == 1 ? o1 : o2), dd = comp == 0 ? d0 : (comp == 1 ? d1 : d2);
                    pv[c] = o + dd * zs[sl];                                                                      // RUN:534
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)                  // (values first, stores after: a store's data register is not reused under it)
                    if (ch * 64 + (c * 64 + lane) / 3 < S) prow[c * 64 + lane] = pv[c];
            }
        }
    }
}

hipError_t launch_embed(const float* x, int64_t P, int multires, float* out, hipStream_t st) {
    const int64_t total = P * 3 * (multires + 1);
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, P, multires, out);
    return hipGetLastError();
}

hipError_t launch_sample_points(const float* rays, const float* t_vals, const float* t_rand, int flags, int64_t N, int S, float* z, float* pts,
                                hipStream_t st) {
    const int64_t groups = (N + kWaves - 1) / kWaves;
    const unsigned grid = (unsigned)std::min<int64_t>(groups, 1 << 30);           // (one ray per wave; the loop only covers N beyond that)
    hipLaunchKernelGGL(sample_points_kernel, dim3(grid), dim3(kThreads), 0, st, rays, t_vals, t_rand, flags, N, S, z, pts);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// EXTENSION: hierarchical resampling (semantics of nerf-pytorch's sample_pdf, the reference's upstream; the
// reference itself has no second pass).  One wave per ray.
constexpr int kPdfMax = 1024;            // S + N_importance
__global__ __launch_bounds__(64)
void sample_pdf_kernel(const float* __restrict__ rays, const float* __restrict__ t_vals, const float* __restrict__ t_rand, int flags,
                       const float* __restrict__ weights, const float* __restrict__ u, int64_t N, int S, int K, int Ni,
                       float* __restrict__ z_out) {
    __shared__ float cdf[kPdfMax], bins[kPdfMax], zall[kPdfMax], zr[kPdfMax];
    const int lane = lane_id();
    const int64_t n = blockIdx.x;
    if (n >= N) return;
    {   // coarse depths, the same arithmetic as the fused forward (RUN:510-532)
        const float nearv = rays[n * 11 + 6], farv = rays[n * 11 + 7];
        const bool lind = (flags & CFNERF_F_LINDISP) != 0;
        for (int s = lane; s < S; s += 64) {
            const float zc = zlin_f(t_vals[s], nearv, farv, lind);
            float zv = zc;
            if (t_rand != nullptr) {
                const float upper = (s == S - 1) ? zc : 0.5f * (zlin_f(t_vals[s + 1], nearv, farv, lind) + zc);
                const float lower = (s == 0) ? zc : 0.5f * (zc + zlin_f(t_vals[s - 1], nearv, farv, lind));
                zv = lower + (upper - lower) * t_rand[n * (int64_t)S + s];
            }
            zr[s] = zv;
        }
    }
    __syncthreads();
    const int nb = S - 1;                 // bins = mids of consecutive depths; cdf has nb entries (first = 0)
    const int nw = S - 2;                 // weights[..., 1:-1]
    for (int j = lane; j < nb; j += 64) bins[j] = 0.5f * (zr[j + 1] + zr[j]);
    // K-mean of the inner weights (+1e-5, "prevent nans"), their total
    float part = 0.f;
    for (int j = lane; j < nw; j += 64) {
        float w = 0.f;
        for (int k = 0; k < K; ++k) w += weights[(n * S + (j + 1)) * (int64_t)K + k];
        w = w / (float)K + 1e-5f;
        zall[j] = w;                      // scratch
        part += w;
    }
    const float total = wave_sum(part);
    __syncthreads();
    // cdf = cat([0, cumsum(pdf)])
    float carry = 0.f;
    if (lane == 0) cdf[0] = 0.f;
    for (int base = 0; base < nw; base += 64) {
        const int j = base + lane;
        float v = (j < nw) ? zall[j] / total : 0.f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
        if (j < nw) cdf[j + 1] = carry + v;
        carry += __shfl(v, 63, 64);
    }
    __syncthreads();
    for (int j = lane; j < S; j += 64) zall[j] = zr[j];
    for (int i = lane; i < Ni; i += 64) {
        const float ui = u[n * (int64_t)Ni + i];
        int lo = 0, hi = nb;              // searchsorted(cdf, u, right=True): first index with cdf > u
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= ui) lo = mid + 1; else hi = mid; }
        const int below = max(0, lo - 1), above = min(nb - 1, lo);
        const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        const float t = (ui - c0) / denom;
        zall[S + i] = b0 + t * (b1 - b0);
    }
    __syncthreads();
    // merge + sort (rank sort, stable): z_vals = sort(cat([z_vals, z_samples]))
    const int M = S + Ni;
    for (int a = lane; a < M; a += 64) {
        const float va = zall[a];
        int rank = 0;
        for (int b = 0; b < M; ++b) { const float vb = zall[b]; rank += (vb < va || (vb == va && b < a)) ? 1 : 0; }
        z_out[n * (int64_t)M + rank] = va;
    }
}

hipError_t launch_sample_pdf(const float* rays, const float* t_vals, const float* t_rand, int flags, const float* w, const float* u,
                             int64_t N, int S, int K, int Ni, float* z_out, hipStream_t st) {
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)N), dim3(64), 0, st, rays, t_vals, t_rand, flags, w, u, N, S, K, Ni, z_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// weight packing: flat state_dict-ordered parameters -> fragment-ordered operands (cfnerf_layout.h).  Which flat element goes where
// depends on the configuration only, so the map is computed ONCE per model (pack_index_kernel: one thread per copied element, a
// binary search over the source pieces + the index arithmetic of pack_map) into a table of {source, destination, bf16 destination}
// and the per-step kernel is a table-driven gather / scatter (round 4: 13.6 -> ~4 us per optimiser step at W = 256; the search per
// element, six dependent loads deep, was latency-bound).
constexpr uint32_t kNoDst16 = 0xffffffffu;
__global__ void pack_index_kernel(const PackDesc* __restrict__ descs, int ndesc, uint32_t total, uint32_t* __restrict__ table) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_elem <= idx) lo = mid; else hi = mid - 1;
    }
    const PackDesc d = descs[lo];
    uint32_t src, dst, dst16 = kNoDst16;
    pack_map(d, idx - d.first_elem, &src, &dst);
    if (!pack_map16(d, idx - d.first_elem, &dst16)) dst16 = kNoDst16;
    table[3 * (size_t)idx + 0] = src; table[3 * (size_t)idx + 1] = dst; table[3 * (size_t)idx + 2] = dst16;
}

__global__ void pack_kernel(const float* __restrict__ flat, float* __restrict__ packed, unsigned short* __restrict__ packed16,
                            const uint32_t* __restrict__ table, uint32_t total) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const uint32_t src = table[3 * (size_t)idx + 0], dst = table[3 * (size_t)idx + 1], dst16 = table[3 * (size_t)idx + 2];
    const float v = flat[src];
    packed[dst] = v;
    if (packed16 != nullptr && dst16 != kNoDst16) {                              // split-bf16 copy: hi plane, lo plane
        const unsigned w = pack_hl(v);
        packed16[dst16] = (unsigned short)(w & 0xffffu);
        packed16[dst16 + 64 * 8] = (unsigned short)(w >> 16);
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
static int max_blocks_per_cu(const void* fn, size_t lds, int threads) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

template <int W, int MODE, bool TRAIN, int PREC, bool Q4 = false>
static hipError_t launch_fwd_t(const FwdArgs& a, const NetTab& ht, int n_cu, int per_cu, hipStream_t st, int* grid_out) {
    auto fn = fused_fwd_kernel<W, MODE, TRAIN, PREC, Q4>;
    const size_t lds = fwd_lds_bytes(W, ht.ha_sz, a.K);
    const int64_t units = (MODE == 0) ? a.N : (a.P + kTileM - 1) / kTileM;
    int grid = (int)std::min<int64_t>(units, (int64_t)n_cu * per_cu);
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(FwdCfg<W>::NTHR), lds, st, a, ht);
    return hipGetLastError();
}

template <int W>
static hipError_t launch_fwd_w(const FwdArgs& a, const NetTab& ht, int mode, bool train, int prec, int n_cu, int per_cu, hipStream_t st, int* grid_out) {
    if (prec == PREC_BF16X3) {
        if (mode == 0) return train ? launch_fwd_t<W, 0, true, PREC_BF16X3>(a, ht, n_cu, per_cu, st, grid_out) : launch_fwd_t<W, 0, false, PREC_BF16X3>(a, ht, n_cu, per_cu, st, grid_out);
        return train ? launch_fwd_t<W, 1, true, PREC_BF16X3>(a, ht, n_cu, per_cu, st, grid_out) : launch_fwd_t<W, 1, false, PREC_BF16X3>(a, ht, n_cu, per_cu, st, grid_out);
    }
    if (train && a.q4)            // whole tiles + a stash in fp32 mode: the variant that writes the wide streams as Q4 pieces
        return mode == 0 ? launch_fwd_t<W, 0, true, PREC_F32, true>(a, ht, n_cu, per_cu, st, grid_out) : launch_fwd_t<W, 1, true, PREC_F32, true>(a, ht, n_cu, per_cu, st, grid_out);
    if (mode == 0) return train ? launch_fwd_t<W, 0, true, PREC_F32>(a, ht, n_cu, per_cu, st, grid_out) : launch_fwd_t<W, 0, false, PREC_F32>(a, ht, n_cu, per_cu, st, grid_out);
    return train ? launch_fwd_t<W, 1, true, PREC_F32>(a, ht, n_cu, per_cu, st, grid_out) : launch_fwd_t<W, 1, false, PREC_F32>(a, ht, n_cu, per_cu, st, grid_out);
}

hipError_t launch_fused_fwd(const FwdArgs& a, const NetTab& ht, int mode, bool train, int prec, int n_cu, int per_cu, hipStream_t st, int* grid_out) {
    switch (ht.W) {
#define CFN_W_CASE(w) case w: return launch_fwd_w<w>(a, ht, mode, train, prec, n_cu, per_cu, st, grid_out);
        CFN_FOR_EACH_WIDTH(CFN_W_CASE)
#undef CFN_W_CASE
    }
    return hipErrorInvalidValue;
}

// Per-DEVICE set-up of the fused forward kernels of one width (called from cfnerf_model_create with that device
// current): raise the dynamic-LDS limit of every variant and read the occupancy.  Nothing here is process-global.
template <int W>
static hipError_t fwd_attrs_w(int ha, int* per_cu_out) {
    const size_t lds = fwd_lds_bytes(W, ha, kMaxK);      // the limit; a launch asks for what its K needs
    const void* fns[10] = {
        reinterpret_cast<const void*>(fused_fwd_kernel<W, 0, true, PREC_F32, true>), reinterpret_cast<const void*>(fused_fwd_kernel<W, 1, true, PREC_F32, true>),
        reinterpret_cast<const void*>(fused_fwd_kernel<W, 0, false, PREC_F32>), reinterpret_cast<const void*>(fused_fwd_kernel<W, 0, true, PREC_F32>),
        reinterpret_cast<const void*>(fused_fwd_kernel<W, 1, false, PREC_F32>), reinterpret_cast<const void*>(fused_fwd_kernel<W, 1, true, PREC_F32>),
        reinterpret_cast<const void*>(fused_fwd_kernel<W, 0, false, PREC_BF16X3>), reinterpret_cast<const void*>(fused_fwd_kernel<W, 0, true, PREC_BF16X3>),
        reinterpret_cast<const void*>(fused_fwd_kernel<W, 1, false, PREC_BF16X3>), reinterpret_cast<const void*>(fused_fwd_kernel<W, 1, true, PREC_BF16X3>)};
    int per_cu = 2;
    for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        per_cu = std::min(per_cu, max_blocks_per_cu(fn, fwd_lds_bytes(W, ha, 16), FwdCfg<W>::NTHR));
    }
    *per_cu_out = std::max(1, per_cu);
    return hipSuccess;
}

hipError_t fused_fwd_set_attributes(int W, int ha, int* per_cu_out) {
    switch (W) {
#define CFN_W_CASE(w) case w: return fwd_attrs_w<w>(ha, per_cu_out);
        CFN_FOR_EACH_WIDTH(CFN_W_CASE)
#undef CFN_W_CASE
    }
    return hipErrorInvalidValue;
}

int fused_fwd_max_grid(int W, int ha, int n_cu) {
    // upper bound of the grid used by launch_fused_fwd (for sizing the entropy partial buffer)
    (void)W; (void)ha;
    return n_cu * 2;
}

hipError_t launch_entropy_finalize(const float* partials, int n_part, const float* flat, const float* eps, int K,
                                   double count, float* out, float* eps_keep, const float* rays, float* rays_keep, int64_t n_rays_floats,
                                   hipStream_t st) {
    const unsigned copy_blocks = (rays_keep != nullptr) ? (unsigned)((n_rays_floats + 255) / 256) : 0u;
    hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1 + copy_blocks), dim3(256), 0, st, partials, n_part, flat, eps, K, count, out, eps_keep,
                       rays, rays_keep, rays_keep != nullptr ? n_rays_floats : 0);
    return hipGetLastError();
}

hipError_t launch_composite(const float* raw, const float* z, const float* d, int64_t N, int S, int K, int wb,
                            float* rgb, float* disp, float* depth, float* weights, hipStream_t st) {
    const int grid = (int)((N + kWaves - 1) / kWaves);
#define CFN_COMP(KG, FAST) hipLaunchKernelGGL((composite_kernel<KG, FAST>), dim3(grid), dim3(kThreads), 0, st, raw, z, d, N, S, K, wb, rgb, disp, depth, weights)
    if (K <= 4) CFN_COMP(4, false);                 // (a group of 4 latents = the whole 64-byte row of a sample at the headline K = 4)
    else if (K < kFastFlowsK) CFN_COMP(8, false);
    else CFN_COMP(8, true);
#undef CFN_COMP
    return hipGetLastError();
}

hipError_t launch_rays_setup(int H, int Wd, float focal, const RaysC2W& c2w, int use_c2w, const float* ro, const float* rd,
                             int64_t N, int64_t pixel0, int ndc, float nearv, float farv, float* out, hipStream_t st) {
    const int grid = (int)((N + 255) / 256);
    hipLaunchKernelGGL(rays_setup_kernel, dim3(grid), dim3(256), 0, st, H, Wd, focal, c2w, use_c2w, ro, rd, N, pixel0, ndc, nearv, farv, out);
    return hipGetLastError();
}

hipError_t launch_pack_index(const PackDesc* descs, int ndesc, uint32_t total, uint32_t* table, hipStream_t st) {
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_index_kernel, dim3(grid), dim3(256), 0, st, descs, ndesc, total, table);
    return hipGetLastError();
}

hipError_t launch_pack(const float* flat, float* packed, void* packed16, const uint32_t* table, uint32_t total, hipStream_t st) {
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, st, flat, packed, reinterpret_cast<unsigned short*>(packed16), table, total);
    return hipGetLastError();
}

}  // namespace cfnerf
