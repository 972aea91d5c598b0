// cfnerf_bwd.hip - train-step kernels for gfx950: KDE-NLL loss, adjoint of composite + flows ("tail"),
// fused backward-data MLP kernel, split-K weight-gradient GEMM, gradient reduce, fused Adam.
//
// Reference lines replaced: loss RUN:1026-1050; loss.backward() RUN:1066 (autograd of RUN:411-454,
// MOD:188-291, FLW:225-268); Adam RUN:339,1067.
//
// Data flow of cfnerf_render_bwd (activations come from the CFNERF_F_STASH forward; everything lives in the workspace):
//   tail_bwd_kernel    d_rgb_map/d_depth/d_entropy + raw, alpha, T, theta -> g_theta [parts][P,128] (+ base-Gaussian partials)
//   reduce_gms         -> grad of alpha_mean/std, rgb_mean/std
//   bwd_data_kernel    sum of the g_theta parts -> g_hr, g_ha, g_v, g_feat, g_h[D-1..0]  (same LDS-tile / MFMA structure as forward)
//   reduce_bias        per-workgroup bias partials -> every bias gradient
//   dw_big_kernel      dW = dY^T X for the >= 128 x 128 jobs: ONE launch, 2 x 4 and 1 x 8 wave arrangements, split over P
//   reduce_weights(1)  tensors fed by the big launch alone -> grad_flat; ev_early recorded (cfnerf_stream_wait_grad_early)
//   dw_small_kernel    encodings / heads / flow heads
//   reduce_weights(0)  the remaining tensors -> grad_flat        (deterministic throughout: no float atomics anywhere)
#include "cfnerf_device.h"
#include "cfnerf_kernels.h"
#include "cfnerf_model.h"
#include "cfnerf_bwd.h"
#include "cfnerf_dwplan.h"

#include <cstdarg>
#include <cstdio>

namespace cfnerf {

// ================================================================================================
// 1. loss (RUN:1026-1050).  One thread per (ray, channel); the two sums (nll, squared error) are accumulated across
//    workgroups as 64-bit FIXED-POINT integers (integer addition is associative: the result does not depend on the order
//    the workgroups arrive in, so the step stays bit-reproducible) in the 16 bytes of `scalars`, which a one-thread
//    finalize kernel then turns into [loss, loss_nll, mse, psnr].  (A single 1024-thread workgroup took 87 us at K = 16.)
constexpr double kLossFix = 68719476736.0;      // 2^36: resolution 1.5e-11 per workgroup partial, |sum| < 3.3e7
constexpr int kLossThreads = 256;
// L lanes share one (ray, channel) row of K values (lane `sub` takes k = sub, sub + L, ...; sums meet through an L-wide xor butterfly).
// L = 1 is the kernel of rounds 1-3, operation for operation - every K below kLossWideK (the reference's plumbing and headline
// configurations) runs it.  L = 8 from kLossWideK latents on: one thread per row walks 2 K correctly rounded expf serially on 12
// workgroups (39 us at the reference's default K = 64, 22 at K = 32); eight lanes per row are 96 workgroups of K / 8-step loops.
constexpr int kLossWideK = 16;
template <int L>
__global__ __launch_bounds__(kLossThreads)
void loss_kernel(const float* __restrict__ rgb, const float* __restrict__ target, int64_t N, int K, int64_t n_total,
                 float* __restrict__ d_rgb, long long* __restrict__ acc) {
    __shared__ double sh[2][kLossThreads];
    const float invK = 1.f / (float)K;
    const float bw = powf(0.8f / (float)K, -1.f / 7.f);                 // torch.pow(0.8/n, tensor(-1/7))  RUN:1036
    const float c2pi = powf(2.f * 3.14159265358979323846f, -1.5f);      // RUN:1039
    const float gscale = 1.f / (3.f * (float)n_total);
    double nll = 0.0, mse = 0.0;
    const int64_t t_idx = (int64_t)blockIdx.x * kLossThreads + threadIdx.x;
    const int64_t i = t_idx / L;
    const int sub = (int)(t_idx - i * L);
    auto group_sum = [](float v) {
#pragma unroll
        for (int d = 1; d < L; d <<= 1) v += __shfl_xor(v, d, 64);
        return v;
    };
    const bool live = i < N * 3;                                        // (a row's L lanes sit in one wave: live is uniform over them)
    const float* x = rgb + (live ? i : 0) * K;                          // [N,3,K]: (n,c) row of K values
    const float t = live ? target[i] : 0.f;
    float mean = 0.f;
    for (int k = sub; k < K; k += L) mean += x[k];
    mean = group_sum(mean) * invK;                                      // RUN:1027
    float var = 0.f;
    for (int k = sub; k < K; k += L) { const float d = x[k] - mean; var += d * d; }
    var = group_sum(var);
    const float sd = sqrtf(var / (float)(K - 1));                       // torch.std (unbiased); K == 1 -> NaN like the reference (R4)
    const float rgb_std = sd * (float)K / (float)(K - 1);               // RUN:1034
    const float H = rgb_std * bw + 1e-05f;                              // RUN:1036 (detached)
    const float c2 = c2pi / H;                                          // RUN:1039
    const float inv2h2 = 1.f / (2.f * H * H);
    float rsum = 0.f;
    for (int k = sub; k < K; k += L) { const float d = x[k] - t; rsum += expf(-(d * d) * inv2h2) * c2; }
    rsum = group_sum(rsum);
    const float m = rsum * invK + 1e-05f;                               // RUN:1041
    if (live) {
        if (sub == 0) {
            nll = -logf(m);                                             // RUN:1042
            mse = (double)((mean - t) * (mean - t));                    // RUN:1028
        }
        const float gcoef = gscale * invK / (m * H * H);
        for (int k = sub; k < K; k += L) {
            const float d = x[k] - t;
            d_rgb[i * K + k] = gcoef * (expf(-(d * d) * inv2h2) * c2) * d;
        }
    }
    sh[0][threadIdx.x] = nll; sh[1][threadIdx.x] = mse;
    __syncthreads();
    for (int d = kLossThreads / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { sh[0][threadIdx.x] += sh[0][threadIdx.x + d]; sh[1][threadIdx.x] += sh[1][threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // a NaN / inf partial (K == 1, R4) must surface as NaN in the result, not as integer garbage: flag it in bit 62
        const bool bad0 = !(fabs(sh[0][0]) < 1e8), bad1 = !(fabs(sh[1][0]) < 1e8);
        atomicAdd(reinterpret_cast<unsigned long long*>(acc + 0), bad0 ? (1ull << 62) : (unsigned long long)llrint(sh[0][0] * kLossFix));
        atomicAdd(reinterpret_cast<unsigned long long*>(acc + 1), bad1 ? (1ull << 62) : (unsigned long long)llrint(sh[1][0] * kLossFix));
    }
}

__global__ void loss_finalize_kernel(const float* __restrict__ entropy, float beta1, int64_t n_total, float* __restrict__ scalars) {
    const long long a0 = reinterpret_cast<const long long*>(scalars)[0], a1 = reinterpret_cast<const long long*>(scalars)[1];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double s0 = (a0 >= (1ll << 61) || a0 < -(1ll << 61)) ? nan : (double)a0 / kLossFix;
    const double s1 = (a1 >= (1ll << 61) || a1 < -(1ll << 61)) ? nan : (double)a1 / kLossFix;
    const double nllm = s0 / (3.0 * (double)n_total), msem = s1 / (3.0 * (double)n_total);
    const float ent = entropy ? entropy[0] : 0.f;
    scalars[0] = (float)nllm + (beta1 != 0.f ? beta1 * ent : 0.f);      // RUN:1047-1050
    scalars[1] = (float)nllm;
    scalars[2] = (float)msem;
    scalars[3] = (float)(-10.0 * log(msem) / log(10.0));                // HLP:16
}

// ================================================================================================
// 2. tail (adjoint of raw2outputs + of the K flows) and 2b. flows_bwd_kernel: in cfnerf_tail.hip - a translation unit of its own,
//    compiled WITHOUT the SLP vectoriser (see there); launched through launch_tail_bwd / launch_flows_bwd (cfnerf_bwd.h).

// ------------------------------------------------------------------------------------------------
// 2c. The unfused seam's other half (the reference's raw2outputs is separately differentiable):
// composite_bwd_kernel: adjoint of raw2outputs (RUN:411-454) as a stateless call.  One wave per ray, lane = sample; the forward
// is recomputed with the forward kernels' own arithmetic (softplus / exp / wave product scan with a per-chunk carry), the carry
// of every 64-sample chunk is parked in LDS, then the chunks are walked back to front exactly like the fused tail kernel.
// Memory side (round 4, like composite_kernel in cfnerf_fwd.hip): KG latents at a time; a chunk's [64][KG] block of raw arrives by
// LDS-DMA in fully coalesced 1-KB pieces (CompStage, cfnerf_device.h), d_raw is written INTO the block's slots and leaves by the
// same pieces - rounds 1-3 read and wrote 16 bytes per lane at a stride of 16 K bytes, once per latent (0.7 TB/s at K = 32).
//   d loss / d w_s = sum_c G_c c_s + Gd' z_s + Ga + Gw_s     with  Gd' = Gd + Gdisp d disp / d depth,
//   Ga = Gdisp d disp / d acc - [white_bkgd] sum_c G_c       (disp = 1 / max(2e-10, depth / (acc + 1e-10) + 1e-10), RUN:448)
constexpr int kCompMaxChunks = 64;            // S <= 4096
template <int KG, bool FAST>
__global__ __launch_bounds__(kThreads)
void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays_d, int64_t N, int S,
                          int K, int white_bkgd, const float* __restrict__ d_rgb, const float* __restrict__ d_disp,
                          const float* __restrict__ d_depth, const float* __restrict__ d_weights, float* __restrict__ d_raw) {
#pragma clang fp contract(fast)      // (like the fused tail kernel: the two must agree to rounding, see tests/test_hip_unfused_seam.py)
    using M = Num<FAST>;
    using St = CompStage<KG>;
    __shared__ __attribute__((aligned(16))) float stage_all[kWaves][St::kQuads * 4];
    __shared__ float carryT[kWaves][KG][kCompMaxChunks];
    const int lane = lane_id_opaque(), wave = wave_id();
    const int64_t ray = (int64_t)blockIdx.x * kWaves + wave;
    if (ray >= N) return;
    float* stage = stage_all[wave];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)stage);
    const float* d = rays_d + ray * 3;
    const float dnorm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zr = z_vals + ray * (int64_t)S;
    const i32x4 rsrc = ds_rsrc(raw + ray * (int64_t)S * K * 4, S * K * 16);
    float* drow = d_raw + ray * (int64_t)S * K * 4;
    const int nch = (S + 63) / 64;
    const bool vec_w = d_weights != nullptr && (K & 3) == 0;
    unsigned voff[KG];
#pragma unroll
    for (int j = 0; j < KG; ++j) voff[j] = St::piece_voff(lane, j, K);
    const int my_q0 = lane * KG, my_swz = St::swz(lane);
    auto fetch = [&](int ch, int g0) { St::fetch(rsrc, lds0, voff, ch, g0, K); };
    auto landed = [&]() { St::landed(); };
    for (int g0 = 0; g0 < K; g0 += KG) {
        // ---- forward sums of the group's latents (depth, acc) and the transmittance entering every chunk
        float car[KG];
        float accd = 0.f, acca = 0.f;                        // lane q: latent g0 + q
#pragma unroll
        for (int q = 0; q < KG; ++q) car[q] = 1.f;
        for (int ch = 0; ch < nch; ++ch) {
            const int s = ch * 64 + lane;
            const bool valid = s < S;
            fetch(ch, g0);
            float zv = 0.f, dist = 0.f;
            if (valid) {
                zv = zr[s];
                const float dz = (s == S - 1) ? 1e1f : zr[s + 1] - zv;
                dist = dz * dnorm;
            }
            landed();
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                if (g0 + q < K) {
                    const float r3 = stage[(my_q0 + (q ^ my_swz)) * 4 + 3];
                    const float alpha = valid ? 1.f - M::exp(-M::softplus(r3) * dist) : 0.f;
                    float incl, excl;
                    comp_scan_mul((1.f - alpha) + 1e-10f, incl, excl);
                    if (lane == 0) carryT[wave][q][ch] = car[q];
                    const float wgt = alpha * (car[q] * excl);
                    const float sd = comp_sum(wgt * zv), sa = comp_sum(wgt);
                    if (lane == q) { accd += sd; acca += sa; }
                    car[q] *= comp_last(incl);
                }
            }
        }
        // ---- cotangents of the group's latents, lane q: latent g0 + q
        float G0v = 0.f, G1v = 0.f, G2v = 0.f, Gdv = 0.f, Gav = 0.f;
        {
            const int k = g0 + lane;
            if (lane < KG && k < K) {
                G0v = d_rgb[ray * 3 * (int64_t)K + 0 * K + k]; G1v = d_rgb[ray * 3 * (int64_t)K + 1 * K + k]; G2v = d_rgb[ray * 3 * (int64_t)K + 2 * K + k];
                Gdv = (d_depth != nullptr) ? d_depth[ray * (int64_t)K + k] : 0.f;
                Gav = white_bkgd ? -(G0v + G1v + G2v) : 0.f;
                if (d_disp != nullptr) {
                    const float qq = accd / (acca + 1e-10f) + 1e-10f;
                    if (qq > 1e-10f + 1e-10f) {                                    // torch.max routes the gradient to the larger argument
                        const float gq = -d_disp[ray * (int64_t)K + k] / (qq * qq);
                        Gdv += gq / (acca + 1e-10f);
                        Gav += gq * (-accd / ((acca + 1e-10f) * (acca + 1e-10f)));
                    }
                }
            }
        }
        // ---- back to front: the transmittance adjoint as a reverse affine wave scan + a carry (see tail_bwd_kernel, comp_adjoint_D)
        float cg[KG], cx[KG], cD[KG];                        // (g, x, D) of the first sample of the chunk behind (comp_adjoint_D, cfnerf_device.h)
#pragma unroll
        for (int q = 0; q < KG; ++q) { cg[q] = 0.f; cx[q] = 0.f; cD[q] = 0.f; }
        for (int ch = nch - 1; ch >= 0; --ch) {
            const int s = ch * 64 + lane;
            const bool valid = s < S;
            fetch(ch, g0);
            float zv = 0.f, dist = 0.f;
            float dwv[KG];
#pragma unroll
            for (int q = 0; q < KG; ++q) dwv[q] = 0.f;
            if (valid) {
                zv = zr[s];
                const float dz = (s == S - 1) ? 1e1f : zr[s + 1] - zv;
                dist = dz * dnorm;
                if (d_weights != nullptr) {
                    const float* wrow = d_weights + (ray * S + s) * (int64_t)K + g0;
                    if (vec_w) {
#pragma unroll
                        for (int q = 0; q < KG; q += 4)
                            if (g0 + q < K) { const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + q); dwv[q] = v[0]; dwv[q + 1] = v[1]; dwv[q + 2] = v[2]; dwv[q + 3] = v[3]; }
                    } else {
#pragma unroll
                        for (int q = 0; q < KG; ++q)
                            if (g0 + q < K) dwv[q] = wrow[q];
                    }
                }
            }
            landed();
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                if (g0 + q < K) {
                    float* slot = stage + (my_q0 + (q ^ my_swz)) * 4;
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(slot);
                    const float G0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(G0v), q)), G1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(G1v), q)),
                                G2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(G2v), q)), Gd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Gdv), q)),
                                Ga = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Gav), q));
                    const float ea = valid ? M::exp(-M::softplus(rv[3]) * dist) : 1.f;
                    const float alpha = 1.f - ea;
                    const float xk = (1.f - alpha) + 1e-10f;
                    float incl_m, excl_m;
                    comp_scan_mul(xk, incl_m, excl_m);
                    const float Tt = carryT[wave][q][ch] * excl_m;
                    const float c0 = t_sigmoid(rv[0]), c1 = t_sigmoid(rv[1]), c2 = t_sigmoid(rv[2]);
                    const float w = alpha * Tt;
                    float g = (G0 * c0 + G1 * c1 + G2 * c2) + Gd * zv;
                    g += Ga;
                    g += dwv[q];
                    const float dalpha = Tt * comp_adjoint_D(valid ? g : 0.f, xk, cg[q], cx[q], cD[q]);      // = g T - suffix / x, carried as the cancelled quantity
                    const float sg = t_sigmoid(rv[3]);                                 // softplus'
                    f32x4 o;
                    o[0] = G0 * w * c0 * (1.f - c0); o[1] = G1 * w * c1 * (1.f - c1); o[2] = G2 * w * c2 * (1.f - c2);
                    o[3] = dalpha * ea * dist * sg;                                    // d alpha / d softplus = dist e, from the exponential itself (tail_bwd_kernel)
                    *reinterpret_cast<f32x4*>(slot) = o;                               // in place: the block leaves the way it came
                }
            }
            wave_lds_turn();
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                const int p = j * 64 + lane, sl = p / KG, kk = (p % KG) ^ St::swz(sl);
                const int sj = ch * 64 + sl, k = g0 + kk;
                if (sj < S && k < K)
                    *reinterpret_cast<f32x4*>(drow + ((int64_t)sj * K + k) * 4) = *reinterpret_cast<const f32x4*>(stage + p * 4);
            }
        }
    }
}

// ================================================================================================
// 3. fused backward-data: per 64-point tile walk the network backwards out of the LDS tile.
//    dX = dY W  uses the transposed packed operands (bt_*); ReLU masks come from the stashed
//    activations; every pre-activation gradient is written out for the weight-gradient GEMM.
__host__ __device__ inline size_t bwd_lds_bytes(int W, int ha) {
    return sizeof(float) * ((size_t)kTileM * act_ld(W) + (size_t)kTileM * (ha + 4));
}

// Epilogue operands fetched BEFORE the MFMA block of a layer so their latency hides under it: the ReLU mask of
// this lane's output fragment (ONE bit word written by the forward in the same fragment layout) and the running
// bias partial.
template <int NTW>
struct EpiPre {
    uint32_t mb[NTW];
    float db[NTW];
};

template <int NTW>
__device__ __forceinline__ void epi_prefetch(EpiPre<NTW>& e, int nt_total, int nt0, int nts, const uint32_t* __restrict__ mbits,
                                             const float* __restrict__ dbp, bool first) {
    const int lane = lane_id_opaque();
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int nt = nt0 + j * nts;
        if (nt >= nt_total) nt = nt0 < nt_total ? nt0 : 0;      // clamped: value unused
        e.db[j] = first ? 0.f : dbp[nt * 32 + (lane & 31)];    // (uniform) a workgroup's first tile STARTS its partial rows: no memset of [n_wg, nb] per step
        e.mb[j] = (mbits != nullptr) ? mbits[nt * 64 + lane] : 0xffffffffu;      // bit 31 - e: element e of the fragment
    }
}

// acc (masked) -> LDS tile, global dY matrix, per-workgroup bias partials.  The dY slab goes out through a buffer
// descriptor over the tile's valid rows (scalar row offsets, ragged rows dropped by the bounds check: see slab_store);
// the mask is applied with one v_bfe_i32 + one v_and per element (relu_bit_apply).
// Q4 (whole tiles only, never RAGGED): the four rows of a group leave as ONE 16-byte store per lane, 1 KB contiguous per instruction,
// in the fragment-quad layout of cfnerf_device.h - 16 vector-memory instructions per n-tile instead of 64 (-3.5 % on this kernel).
template <int NTW, int PREC, bool RAGGED, bool Q4>
__device__ __forceinline__ void store_bwd_impl(const f32x16 (&acc)[2][NTW], const EpiPre<NTW>& e, int nt_total, int nt0, int nts,
                                               float* lds_dst, int ld, float* __restrict__ gdst, int gld, float* __restrict__ dbp,
                                               int rows_valid) {
    const int lane = lane_id_opaque();
    const int rbase = 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t sink = slab_rsrc(gdst, Q4 ? kTileM : rows_valid, gld);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = nt0 + j * nts;
        if (nt >= nt_total) continue;
        const int col = nt * 32 + (lane & 31);
        float* lrow = lds_dst + rbase * ld;
        const int voff = (rbase * gld + col) * 4;
        const uint32_t mb = e.mb[j];
        float csum = 0.f;
        [[maybe_unused]] u32x4 q4v;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                          // elements r, r + 1: two consecutive rows of this column
                const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                float v0 = relu_bit_apply(mb, i * 16 + r, acc[i][j][r]);
                float v1 = relu_bit_apply(mb, i * 16 + r + 1, acc[i][j][r + 1]);
                if (RAGGED && rr + rbase >= rows_valid) v0 = 0.f;      // rows past a ragged tile hold garbage activations
                if (RAGGED && rr + 1 + rbase >= rows_valid) v1 = 0.f;
                act_store2<PREC>(lrow + rr * ld, lrow + (rr + 1) * ld, ld, col, v0, v1);
                if (!Q4) {
                    slab_store(sink, voff, rr * gld * 4, v0);
                    slab_store(sink, voff, (rr + 1) * gld * 4, v1);
                } else if ((r & 3) == 0) {
                    q4v[0] = __float_as_uint(v0); q4v[1] = __float_as_uint(v1);
                } else {
                    q4v[2] = __float_as_uint(v0); q4v[3] = __float_as_uint(v1);
                    q4_store(q4v, sink, lane, q4_piece(i, nt, r >> 2, gld >> 5));
                }
                csum += v0; csum += v1;
            }
        csum += __shfl_xor(csum, 32, 64);
        if (lane < 32) dbp[col] = e.db[j] + csum;    // this (workgroup, column) is owned by exactly one lane: no atomics
    }
}
// the ragged-row test costs a compare and a select per element: full tiles (all but a ray's last, when S is not a multiple of
// 64) take the variant without it (wave-uniform branch)
template <int NTW, int PREC>
__device__ __forceinline__ void store_bwd(const f32x16 (&acc)[2][NTW], const EpiPre<NTW>& e, int nt_total, int nt0, int nts,
                                          float* lds_dst, int ld, float* __restrict__ gdst, int gld, float* __restrict__ dbp,
                                          int rows_valid, bool q4 = false) {
    if (q4 && PREC == PREC_F32) store_bwd_impl<NTW, PREC, false, PREC == PREC_F32>(acc, e, nt_total, nt0, nts, lds_dst, ld, gdst, gld, dbp, rows_valid);
    else if (rows_valid >= 64)  store_bwd_impl<NTW, PREC, false, false>(acc, e, nt_total, nt0, nts, lds_dst, ld, gdst, gld, dbp, rows_valid);
    else                        store_bwd_impl<NTW, PREC, true, false>(acc, e, nt_total, nt0, nts, lds_dst, ld, gdst, gld, dbp, rows_valid);
}

// the kernarg segment of bwd_data_kernel as one struct (see fused_fwd_kernel)
struct BwdKargs { BwdArgs A; NetTab T; };
static_assert(offsetof(BwdKargs, T) == (sizeof(BwdArgs) + alignof(NetTab) - 1) / alignof(NetTab) * alignof(NetTab), "kernarg layout");

template <int W, int PREC>
__global__ __launch_bounds__(kThreads, (W <= 256) ? 2 : 1)
void bwd_data_kernel(const BwdArgs A_, const NetTab T_) {
    // arguments: scalar loads from the kernarg segment, fetched per phase (CFN_PHASE_ARGS; see fused_fwd_kernel and kernarg_fresh)
#define CFN_PHASE_LOCALS(F)                                                                                                  \
    [[maybe_unused]] const int HA = F(HA0), HR = F(HR0), HLD = HA + 4, D = F(D0), wave = F(wave0), Sn = F(Sn0);               \
    [[maybe_unused]] const float* __restrict__ const wp = F(wp0);                                                              \
    [[maybe_unused]] const __bf16* __restrict__ const wp16 = F(wp160);                                                         \
    [[maybe_unused]] const int64_t P = F(P0), n_tiles = F(n_tiles0);                                                           \
    [[maybe_unused]] float* const dbp = F(dbp0);                 /* this workgroup's bias-gradient partials */                 \
    [[maybe_unused]] const uint32_t* const mb_all = F(mb_all0)
    // Which scheme pays depends on the occupancy (round 4, same-box A/B of 200-step medians): with two workgroups per CU (W <= 256) the
    // neighbour's MFMAs cover the constant-cache round trips and the ~650 v_readlane of the by-value scheme were the larger cost
    // (backward-data 1.166 -> 1.159 ms at W = 256); with ONE workgroup per CU (W > 256: one wave per SIMD, nothing covers a scalar
    // load) fetching per phase was 0.7 % SLOWER (2.215 -> 2.233 ms at W = 512), so the wide kernels keep the arguments by value.
    constexpr bool kKargMem = W <= 256;
#define CFN_KARGS const CFN_KCONST BwdKargs* kq_ = kernarg_fresh<BwdKargs, kKargMem>(); \
                  auto& A = karg_pick<kKargMem>::get(kq_->A, A_); auto& T = karg_pick<kKargMem>::get(kq_->T, T_)
#define CFN_FRESH(x) sgpr_fresh_if<kKargMem>(x)
#define CFN_PHASE_ARGS CFN_KARGS; CFN_PHASE_LOCALS(CFN_FRESH)
    CFN_KARGS;
    constexpr int LD = act_ld(W);
    constexpr int NT = W / 32, NTW = (NT + kWaves - 1) / kWaves, NTV = (W / 64 + kWaves - 1) / kWaves;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HA0 = T.ha_sz, HR0 = T.hr_sz, D0 = T.D;
    float* const act = smem;
    float* const hs = act + kTileM * LD;
    const int tid = threadIdx.x, wave0 = wave_id();
    const float* const wp0 = A.wp;
    const __bf16* const wp160 = reinterpret_cast<const __bf16*>(A.wp16);
    const int64_t P0 = A.P;
    float* const dbp0 = A.dbp + (size_t)blockIdx.x * A.nb;
    const int64_t n_tiles0 = A.n_tiles;
    constexpr int kMbStride = (W / 32) * 64;
    const uint32_t* const mb_all0 = A.mbits;

    const int Sn0 = A.S;
    for (int64_t tile = blockIdx.x; tile < n_tiles0; tile += gridDim.x) {
        CFN_PHASE_ARGS;
        const int cpr = (Sn + kTileM - 1) / kTileM;             // chunks per ray: the forward's tiling
        const int64_t ray = tile / cpr;
        const int chunk = (int)(tile - ray * cpr);
        const int64_t p0 = ray * (int64_t)Sn + (int64_t)chunk * kTileM;
        const int rows_valid = min(kTileM, Sn - chunk * kTileM);
        const bool first = tile == (int64_t)blockIdx.x;        // this workgroup's first tile: its bias-partial rows start here
        // ---- 0. g_theta tile -> act[:, 0:128).  A thread's eight 16-byte pieces are requested in BATCHES before the first of a batch is
        //      consumed (round 4): as one plain loop the compiler kept it rolled - load, s_waitcnt vmcnt(0), ds_write, next piece - i.e.
        //      eight serialised memory round trips at the top of every tile, sixteen with two k-parts (that, not the bytes, was the ~40 us
        //      an extra k-part cost this kernel).  W <= 256: one batch of eight (two workgroups per CU: 27 more VGPRs are free).  The wide
        //      kernels run at 372 + 128 registers and one workgroup per CU - all eight in flight cost them +1.1 % - and take batches of
        //      kGthB pieces.
        {
            constexpr int kGthN = kTileM * (kThetaAll / 4) / kThreads;
            constexpr int kGthB = (W <= 256) ? kGthN : 1;
            const float* const gth = A.g_theta;                 // ONE row per point: a ray's k-parts meet in LDS inside the tail kernel
#pragma unroll
            for (int b0 = 0; b0 < kGthN; b0 += kGthB) {
                f32x4 gv[kGthB];
#pragma unroll
                for (int i = 0; i < kGthB; ++i) {
                    const int idx = tid + (b0 + i) * kThreads, row = idx >> 5, q = idx & 31;
                    gv[i][0] = gv[i][1] = gv[i][2] = gv[i][3] = 0.f;
                    if (row < rows_valid) gv[i] = *reinterpret_cast<const f32x4*>(gth + (p0 + row) * kThetaAll + q * 4);
                }
#pragma unroll
                for (int i = 0; i < kGthB; ++i) {
                    const int idx = tid + (b0 + i) * kThreads, row = idx >> 5, q = idx & 31;
#pragma unroll
                    for (int c = 0; c < 4; ++c) act_store<PREC>(act + row * LD, LD, q * 4 + c, gv[i][c]);
                }
            }
        }
        __syncthreads();
        if (tid < kThetaAll) {                                 // bias gradients of the flow-parameter heads
            float s = 0.f;
            for (int r = 0; r < kTileM; ++r) s += act_load<PREC>(act + r * LD, LD, tid);
            dbp[A.db_theta + tid] = (first ? 0.f : dbp[A.db_theta + tid]) + s;
        }
        // ---- 1. dh_rgb = g_theta_rgb * [amor_d; diag1; diag2; b]   ;   dh_alpha likewise
        if (T.bt_fr.nt > 2 || T.bt_fa.nt > 2) {
            CFN_PHASE_ARGS;
            // h sizes of 96 / 128 (3 - 4 n-tiles per head): every wave takes n-tile `wave` of BOTH heads, one after the other
            f32x16 accR[2][1], accA[2][1];
            EpiPre<1> eR, eA;
            acc_zero(accR); acc_zero(accA);
            epi_prefetch<1>(eR, T.bt_fr.nt, wave, kWaves, nullptr, dbp + A.db_hr, first);
            epi_prefetch<1>(eA, T.bt_fa.nt, wave, kWaves, nullptr, dbp + A.db_ha, first);
            mma_any<1, PREC, (W > 256 ? 3 : 2)>(accR, kload(T.bt_fr), wave, kWaves, wp, wp16, act, LD);
            mma_any<1, PREC, (W > 256 ? 3 : 2)>(accA, kload(T.bt_fa), wave, kWaves, wp, wp16, act, LD, kThetaRgb);
            __syncthreads();
            store_bwd<1, PREC>(accR, eR, T.bt_fr.nt, wave, kWaves, act, LD, A.g_hr + p0 * HR, HR, dbp + A.db_hr, rows_valid);
            store_bwd<1, PREC>(accA, eA, T.bt_fa.nt, wave, kWaves, hs, HLD, A.g_ha + p0 * HA, HA, dbp + A.db_ha, rows_valid);
            __syncthreads();
        } else {
            CFN_PHASE_ARGS;
            f32x16 acc[2][1];
            EpiPre<1> e;
            acc_zero(acc);
            const bool is_rgb = wave < 2;
            if (is_rgb) {
                epi_prefetch<1>(e, T.bt_fr.nt, wave, kWaves, nullptr, dbp + A.db_hr, first);
                mma_any<1, PREC, (W > 256 ? 3 : 2)>(acc, kload(T.bt_fr), wave, kWaves, wp, wp16, act, LD);
            } else {
                epi_prefetch<1>(e, T.bt_fa.nt, wave - 2, kWaves, nullptr, dbp + A.db_ha, first);
                mma_any<1, PREC, (W > 256 ? 3 : 2)>(acc, kload(T.bt_fa), wave - 2, kWaves, wp, wp16, act, LD, kThetaRgb);
            }
            __syncthreads();
            if (is_rgb) store_bwd<1, PREC>(acc, e, T.bt_fr.nt, wave, kWaves, act, LD, A.g_hr + p0 * HR, HR, dbp + A.db_hr, rows_valid);
            else        store_bwd<1, PREC>(acc, e, T.bt_fa.nt, wave - 2, kWaves, hs, HLD, A.g_ha + p0 * HA, HA, dbp + A.db_ha, rows_valid);
            __syncthreads();
        }
        // ---- 2. dv = (dh_rgb * R) . relu'(v)
        {
            CFN_PHASE_ARGS;
            f32x16 acc[2][NTV];
            EpiPre<NTV> e;
            acc_zero(acc);
            epi_prefetch<NTV>(e, T.bt_hr.nt, wave, kWaves, mb_all + ((size_t)D * n_tiles + tile) * kMbStride, dbp + A.db_v, first);
            mma_any<NTV, PREC, (W > 256 ? 3 : 2)>(acc, kload(T.bt_hr), wave, kWaves, wp, wp16, act, LD);
            __syncthreads();
            store_bwd<NTV, PREC>(acc, e, T.bt_hr.nt, wave, kWaves, act, LD, A.g_v + p0 * (W / 2), W / 2, dbp + A.db_v, rows_valid, A.q4 != 0);
            __syncthreads();
        }
        // ---- 3. dfeature = dv * V[:, 0:W]        (feature_linear has no activation, MOD:176)
        {
            CFN_PHASE_ARGS;
            f32x16 acc[2][NTW];
            EpiPre<NTW> e;
            acc_zero(acc);
            epi_prefetch<NTW>(e, T.bt_vf.nt, wave, kWaves, nullptr, dbp + A.db_feat, first);
            mma_any<NTW, PREC, (W > 256 ? 3 : 2)>(acc, kload(T.bt_vf), wave, kWaves, wp, wp16, act, LD);
            __syncthreads();
            store_bwd<NTW, PREC>(acc, e, T.bt_vf.nt, wave, kWaves, act, LD, A.g_feat + p0 * W, W, dbp + A.db_feat, rows_valid, A.q4 != 0);
            __syncthreads();
        }
        // ---- 4. dh_{D-1} = (dfeature * F + dh_alpha * A) . relu'(h_{D-1})
        {
            CFN_PHASE_ARGS;
            f32x16 acc[2][NTW];
            EpiPre<NTW> e;
            acc_zero(acc);
            epi_prefetch<NTW>(e, NT, wave, kWaves, mb_all + ((size_t)(D - 1) * n_tiles + tile) * kMbStride, dbp + A.db_h + (D - 1) * W, first);
            mma_any<NTW, PREC, (W > 256 ? 3 : 2)>(acc, kload(T.bt_ft), wave, kWaves, wp, wp16, act, LD);
            mma_any<NTW, PREC, (W > 256 ? 3 : 2)>(acc, kload(T.bt_ha), wave, kWaves, wp, wp16, hs, HLD);
            __syncthreads();
            store_bwd<NTW, PREC>(acc, e, NT, wave, kWaves, act, LD, A.g_h + ((size_t)(D - 1) * P + p0) * W, W, dbp + A.db_h + (D - 1) * W, rows_valid, A.q4 != 0);
            __syncthreads();
        }
        // ---- 5. trunk: dh_{l-1} = (dh_l * W_l[:, h part]) . relu'(h_{l-1})
        // the next layer's first weight fragments cross L2 under this layer's epilogue: -7 .. -13 us per launch at W = 256 on one box, +-0 on
        // another (the co-resident workgroup already covers most of that latency); at W = 512 the 16 extra registers cost +25 us, so the wide
        // kernels fetch them at the top of the k-loop as before.  (The same in the forward's trunk: +-0 on both boxes, not kept.)
        constexpr bool kBPre = PREC == PREC_F32 && W <= 256;
        f32x4 bpre[NTW];
        if (kBPre) b_prefetch<NTW>(kload(T.bt_trunk[D - 1]), wave, kWaves, wp, bpre);
        for (int l = D - 1; l >= 1; --l) {
            CFN_PHASE_ARGS;
            f32x16 acc[2][NTW];
            EpiPre<NTW> e;
            acc_zero(acc);
            epi_prefetch<NTW>(e, NT, wave, kWaves, mb_all + ((size_t)(l - 1) * n_tiles + tile) * kMbStride, dbp + A.db_h + (l - 1) * W, first);
            if (kBPre) {
                mma_seg_pre<NTW>(acc, kload(T.bt_trunk[l]), wave, kWaves, wp, act, LD, bpre);
                if (l > 1) b_prefetch<NTW>(kload(T.bt_trunk[l - 1]), wave, kWaves, wp, bpre);
            } else {
                mma_any<NTW, PREC, (W > 256 ? 3 : 2)>(acc, kload(T.bt_trunk[l]), wave, kWaves, wp, wp16, act, LD);
            }
            __syncthreads();
            store_bwd<NTW, PREC>(acc, e, NT, wave, kWaves, act, LD, A.g_h + ((size_t)(l - 1) * P + p0) * W, W, dbp + A.db_h + (l - 1) * W, rows_valid, A.q4 != 0);
            __syncthreads();
        }
    }
#undef CFN_PHASE_ARGS
#undef CFN_PHASE_LOCALS
#undef CFN_KARGS
#undef CFN_FRESH
}

// ================================================================================================
// 4. weight gradients: dW[n][k] = sum_p dY[p][n] X[p][k], fp32 MFMA, P split across workgroups.
//    MFMA tile t of a wave uses vector component t of its operand vectors, i.e. the rows (cols) of a
//    tile are interleaved with stride 4 (2): lane (i, kk) holds dY[p + kk][n + 4i .. 4i+3] and
//    X[p + kk][k + 2i .. 2i+1] - a free relabelling that makes every operand fetch a full 16-B / 8-B vector.
typedef const float __attribute__((address_space(1)))* gcf_ptr;      // explicit global address space: the pointers
typedef float __attribute__((address_space(1)))* gf_ptr;             // come out of a struct in memory (else flat_load)

// Output of a big tile (always ONE destination tensor: the concatenated flow heads are small jobs).
// TK: MFMA tiles of a wave along k (2: the 2 x 4 wave arrangement, columns k + 2i + tk; 1: the 1 x 8 arrangement, column k + i)
template <int TK>
__device__ __forceinline__ void dw_store_tile(const DwTile& t, gf_ptr out, const f32x16 (&acc)[4][2], int n_base, int k_base, int lane) {
    const int kcol = k_base + TK * (lane & 31);
    gf_ptr o = out + ((size_t)t.seg_dst[0] + t.dst_col + kcol);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n_base + 4 * frag_row(r, lane) + tn;
            if (n < t.N) {
                gf_ptr row = o + (size_t)n * t.dst_ld;
                if (kcol < t.K) row[0] = acc[tn][0][r];
                if (TK == 2 && kcol + 1 < t.K) row[1] = acc[tn][1][r];
            }
        }
}

// ---- 4a. big tiles (256 x 256 per workgroup, 8 waves): operands are staged through double-buffered LDS (32 points
//      per stage) with register prefetch, so every dY / X element is read from HBM exactly once and the loads of stage
//      s+1 fly under the MFMAs of stage s.  Wave arrangement per tile (DwTile::gk): 0 = 2 (n) x 4 (k), each wave
//      128 x 64 (128 MFMAs per stage); 1 = 1 x 8, each wave 128 x 32, for tiles with N <= 128 (the views layer), whose
//      blocks then do half the MFMAs per stage and get twice the points, instead of idling half their waves.
//      Loader: buffer loads through one descriptor per operand that covers exactly the block's point range, so rows past
//      the range and columns past the matrix come back as zeros from the hardware bounds check, the per-lane offsets are
//      set up once and a stage advances ONE scalar offset - no per-stage address arithmetic on the vector pipe, which on
//      this chip would come straight out of the MFMA issue slots.
constexpr int kDwThreads = 512;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// a function's arguments arrive in VECTOR registers: tell the compiler they are wave-uniform, or every buffer load built
// from them is wrapped in a waterfall loop (cdna_hip_programming.md T20).  (The tile bodies below were `noinline` functions in
// rounds 3-4 - 48 to 220 callee-save scratch stores and loads per call; inlined since round 5 the fp32 kernel holds 171 VGPRs and no
// scratch, and the launch is 7 us (W 256) / 12 us (W 512) shorter; the readfirstlanes are then no-ops.)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// ---- LDS-DMA helpers: ds_rsrc / ds_dma16 live in cfnerf_device.h (the standalone composite kernels use them too)
__device__ __forceinline__ void ds_wait_stage(int n_w) {   // at most n_w of this wave's pieces still in flight (wave-uniform n_w)
    if (n_w == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (n_w == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}

template <int PREC, int ARR>
__device__ __forceinline__ void dw_big_body(const DwTile* __restrict__ tiles_, const DwBlock* __restrict__ blocks_,
                                            float* __restrict__ partials_, int64_t n_params_) {
    const DwTile* __restrict__ tiles = uniform_ptr(tiles_);
    const DwBlock* __restrict__ blocks = uniform_ptr(blocks_);
    float* __restrict__ partials = uniform_ptr(partials_);
    const int64_t n_params = (int64_t)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)n_params_ >> 32)) << 32) |
                                       __builtin_amdgcn_readfirstlane((unsigned)n_params_));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                 // [2][32][256]
    float* Bs = smem + 2 * kDwRows * 256;             // [2][32][256]
    const DwBlock blk = blocks[blockIdx.x];
    const DwTile t = tiles[blk.tile];
    const int tid = threadIdx.x, lane = lane_id_opaque(), wave = wave_id();
    // fp32: operands go HBM -> LDS by LDS-DMA (one 1-KB piece = one 256-float row of a stage; wave w moves rows w, w + 8, w + 16,
    // w + 24 of both operands): no staging registers, no ds_write pass between the MFMA blocks.  The split-bf16 mode converts on
    // the way into LDS and keeps the register path.
    constexpr bool DMA = PREC == PREC_F32;
    constexpr bool arr1 = ARR == 1;                   // the wave arrangement picks one of two NON-INLINED bodies (two register
                                                      // allocations: one function holding both loop nests spilt its accumulators)
    const int wn = arr1 ? 0 : wave >> 2, wk = arr1 ? wave : wave & 3;
    const int rows = (int)(blk.pe - blk.pb);          // a split's point range is far below 2^31 bytes / row
    // loader geometry: thread -> (row = tid / 64 + 8 * q, 16-B column c = tid % 64)
    const int lrow = tid >> 6, lc = tid & 63;
    const bool a_col_ok = t.n0 + 4 * lc + 4 <= t.Npad, b_col_ok = t.k0 + 4 * lc + 4 <= t.Kpad;
    // descriptor inputs through readfirstlane: they ARE wave-uniform, this makes it provable (no waterfall loops, T20)
    const int ldY = __builtin_amdgcn_readfirstlane(t.ldY), ldX = __builtin_amdgcn_readfirstlane(t.ldX);
    const int rows_u = __builtin_amdgcn_readfirstlane(rows);
    const __amdgpu_buffer_rsrc_t ra_desc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float*>(t.dY + blk.pb * t.ldY)), 0, rows_u * ldY * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_desc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float*>(t.X + blk.pb * t.ldX)), 0, rows_u * ldX * 4, 0x00020000);
    int va[4], vb[4];                                 // byte offsets of this thread's four rows of a stage; out of range = zeros
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        va[q] = a_col_ok ? ((lrow + 8 * q) * t.ldY + t.n0 + 4 * lc) * 4 : 0x7ffffff0;
        vb[q] = b_col_ok ? ((lrow + 8 * q) * t.ldX + t.k0 + 4 * lc) * 4 : 0x7ffffff0;
    }
    const int sa_step = kDwRows * ldY * 4, sb_step = kDwRows * ldX * 4;
    int sa = 0, sb = 0;                               // scalar stage offsets
    const i32x4 da = ds_rsrc(t.dY + blk.pb * t.ldY, rows_u * ldY * 4), db = ds_rsrc(t.X + blk.pb * t.ldX, rows_u * ldX * 4);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    int dma_stage = 0;                                // stage the NEXT issue fetches (advanced in uniform control flow only)
    auto issue_h = [&](int buf, int q, int which) {   // ONE of the wave's 8 pieces of stage `dma_stage` -> buffer `buf` (which: 0 dY, 1 X)
        const unsigned row_off = ((unsigned)buf * kDwRows + wave_u + 8 * q) * 256 * 4;
        if (which == 0) ds_dma16(da, lds0 + row_off, (unsigned)va[q], (unsigned)__builtin_amdgcn_readfirstlane(dma_stage * sa_step));
        else ds_dma16(db, lds0 + 2 * kDwRows * 256 * 4 + row_off, (unsigned)vb[q], (unsigned)__builtin_amdgcn_readfirstlane(dma_stage * sb_step));
    };
    auto issue_q = [&](int buf, int q) { issue_h(buf, q, 0); issue_h(buf, q, 1); };
    auto issue = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_q(buf, q);
    };
    f32x4 ra[4], rb[4];
    auto gload = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_desc, va[q], sa, 0));
            rb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_desc, vb[q], sb, 0));
        }
        sa += sa_step; sb += sb_step;
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* pa = As + ((buf * kDwRows + lrow + 8 * q) * 256 + 4 * lc);
            float* pb2 = Bs + ((buf * kDwRows + lrow + 8 * q) * 256 + 4 * lc);
            if (PREC == PREC_BF16X3) {          // operands enter LDS as {hi,lo} bf16 words
                u32x4 wa, wb;
#pragma unroll
                for (int c = 0; c < 4; ++c) { wa[c] = pack_hl(ra[q][c]); wb[c] = pack_hl(rb[q][c]); }
                *reinterpret_cast<u32x4*>(pa) = wa;
                *reinterpret_cast<u32x4*>(pb2) = wb;
            } else {
                *reinterpret_cast<f32x4*>(pa) = ra[q];
                *reinterpret_cast<f32x4*>(pb2) = rb[q];
            }
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int k_wave = arr1 ? 32 * wk : 64 * wk;
    const bool active = (t.n0 + 128 * wn < t.N) && (t.k0 + k_wave < t.K);     // wave-uniform
    const int i = lane & 31, kk = lane >> 5;
    const float* a_rd = As + kk * 256 + 128 * wn + 4 * i;
    const float* b_rd = Bs + kk * 256 + (arr1 ? 32 * wk + i : 64 * wk + 2 * i);

    if (DMA) { issue(0); dma_stage = 1; }
    else { gload(); sstore(0); __syncthreads(); }
    int buf = 0;
    for (int p = 0; p < rows_u; p += kDwRows) {      // uniform trip count: keeps the stage offsets in scalar registers
        const bool more = p + kDwRows < rows_u;
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the current stage have landed
            __syncthreads();                                   // ... everyone's have, and everyone is done reading the other buffer
            if (more && !active) issue(buf ^ 1);               // (active waves spread their pieces over the MFMA block below)
        } else {
            if (more) gload();
            __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ABOVE the MFMA block
        }
        if (active) {
            if (PREC == PREC_BF16X3) {
                // two 16-point chunks per stage; lane (i, kk) gathers points 8*kk .. 8*kk+7 of its columns
#pragma unroll
                for (int c16 = 0; c16 < kDwRows / 16; ++c16) {
                    const float* ar = As + (buf * kDwRows + c16 * 16 + 8 * kk) * 256 + 128 * wn + 4 * i;
                    const float* br = Bs + (buf * kDwRows + c16 * 16 + 8 * kk) * 256 + (arr1 ? 32 * wk + i : 64 * wk + 2 * i);
                    u32x4 wa[8];
                    unsigned wb0[8], wb1[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        wa[e] = *reinterpret_cast<const u32x4*>(ar + e * 256);
                        if (arr1) {
                            wb0[e] = __float_as_uint(br[e * 256]); wb1[e] = 0u;
                        } else {
                            const f32x2 t2 = *reinterpret_cast<const f32x2*>(br + e * 256);
                            wb0[e] = __float_as_uint(t2[0]); wb1[e] = __float_as_uint(t2[1]);
                        }
                    }
                    bf16x8 bh[2], bl[2];
                    {
                        u32x4 h, l;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { h[q] = __builtin_amdgcn_perm(wb0[2 * q + 1], wb0[2 * q], 0x05040100u); l[q] = __builtin_amdgcn_perm(wb0[2 * q + 1], wb0[2 * q], 0x07060302u); }
                        bh[0] = __builtin_bit_cast(bf16x8, h); bl[0] = __builtin_bit_cast(bf16x8, l);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { h[q] = __builtin_amdgcn_perm(wb1[2 * q + 1], wb1[2 * q], 0x05040100u); l[q] = __builtin_amdgcn_perm(wb1[2 * q + 1], wb1[2 * q], 0x07060302u); }
                        bh[1] = __builtin_bit_cast(bf16x8, h); bl[1] = __builtin_bit_cast(bf16x8, l);
                    }
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        u32x4 h, l;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            h[q] = __builtin_amdgcn_perm(wa[2 * q + 1][tn], wa[2 * q][tn], 0x05040100u);
                            l[q] = __builtin_amdgcn_perm(wa[2 * q + 1][tn], wa[2 * q][tn], 0x07060302u);
                        }
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, h), al = __builtin_bit_cast(bf16x8, l);
                        acc[tn][0] = CFN_MFMA16(al, bh[0], acc[tn][0]);
                        acc[tn][0] = CFN_MFMA16(ah, bl[0], acc[tn][0]);
                        acc[tn][0] = CFN_MFMA16(ah, bh[0], acc[tn][0]);
                        if (!arr1) {
                            acc[tn][1] = CFN_MFMA16(al, bh[1], acc[tn][1]);
                            acc[tn][1] = CFN_MFMA16(ah, bl[1], acc[tn][1]);
                            acc[tn][1] = CFN_MFMA16(ah, bh[1], acc[tn][1]);
                        }
                    }
                }
            } else {
                const float* ar = a_rd + buf * kDwRows * 256;
                const float* br = b_rd + buf * kDwRows * 256;
                // the next stage's pieces go out two at a time between the first MFMA groups instead of as a burst in front of
                // them (a burst keeps both waves of a SIMD off the matrix pipe at the same moment)
                if (!arr1) {
#pragma unroll
                    for (int pp = 0; pp < kDwRows / 2; ++pp) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(ar + pp * 512);
                        const f32x2 bv = *reinterpret_cast<const f32x2*>(br + pp * 512);
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn) {
                            acc[tn][0] = CFN_MFMA(av[tn], bv[0], acc[tn][0]);
                            acc[tn][1] = CFN_MFMA(av[tn], bv[1], acc[tn][1]);
                        }
                        if (DMA && more && pp < 8) issue_h(buf ^ 1, pp >> 1, pp & 1);            // one piece after each of the first 8 MFMA groups
                    }
                } else {
#pragma unroll
                    for (int pp = 0; pp < kDwRows / 2; ++pp) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(ar + pp * 512);
                        const float bv = br[pp * 512];
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn) acc[tn][0] = CFN_MFMA(av[tn], bv, acc[tn][0]);
                        if (DMA && more && pp < 8) issue_h(buf ^ 1, pp >> 1, pp & 1);            // one piece after each of the first 8 MFMA groups
                    }
                }

            }
        }
        if (!DMA) {
            __builtin_amdgcn_sched_barrier(0);       // ... and its consumer below it
            if (more) sstore(buf ^ 1);
            __syncthreads();
        }
        ++dma_stage;
        buf ^= 1;
    }
    if (active) {
        gf_ptr out = (gf_ptr)(partials + (size_t)blk.split * n_params);
        if (arr1) dw_store_tile<1>(t, out, acc, t.n0, t.k0 + 32 * wk, lane);
        else      dw_store_tile<2>(t, out, acc, t.n0 + 128 * wn, t.k0 + 64 * wk, lane);
    }
}

// The same tile with BOTH operands in the Q4 layout (cfnerf_device.h; fp32 mode, whole tiles).  A 32-point stage of an operand is
// ONE contiguous run of (n-tiles x 4) 1-KB pieces in memory - the LDS-DMA copies it verbatim, no column slicing - and the MFMA loop
// takes a piece as it is: per group g (8 points) one ds_read_b128 per 32-column tile and lane, four k-steps each (lane half h: points
// 8 g + 4 h + e for A and B alike).  Wave arrangement as above: 2 x 4 (wave = n-tiles 4 wn .. 4 wn + 3 x k-tiles 2 wk, 2 wk + 1) or, for
// N <= 128, 1 x 8 (n-tiles 0 .. 3 x k-tile wk).  The output tile is natural (no interleaving): dW[n0 + 32 tn' + frag_row][k0 + 32 tk' + lane & 31].
template <int ARR>
__device__ __forceinline__ void dw_big_body_q4(const DwTile* __restrict__ tiles_, const DwBlock* __restrict__ blocks_,
                                                         float* __restrict__ partials_, int64_t n_params_) {
    const DwTile* __restrict__ tiles = uniform_ptr(tiles_);
    const DwBlock* __restrict__ blocks = uniform_ptr(blocks_);
    float* __restrict__ partials = uniform_ptr(partials_);
    const int64_t n_params = (int64_t)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)n_params_ >> 32)) << 32) |
                                       __builtin_amdgcn_readfirstlane((unsigned)n_params_));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool arr1 = ARR == 1;
    const DwBlock blk = blocks[blockIdx.x];
    const DwTile t = tiles[blk.tile];
    const int lane = lane_id_opaque(), wave = wave_id();
    const int wn = arr1 ? 0 : wave >> 2, wk = arr1 ? wave : wave & 3;
    const int rows_u = __builtin_amdgcn_readfirstlane((int)(blk.pe - blk.pb));       // a multiple of 32 (whole half-tiles)
    const int ldY = __builtin_amdgcn_readfirstlane(t.ldY), ldX = __builtin_amdgcn_readfirstlane(t.ldX);
    // staged n-tiles of the two operands: [n0, n0 + 256) of dY's N columns, [k0, k0 + 256) of X's K columns
    const int ca = __builtin_amdgcn_readfirstlane(min(8, (t.N - t.n0 + 31) >> 5)), cb = __builtin_amdgcn_readfirstlane(min(8, (t.K - t.k0 + 31) >> 5));
    // a half-tile (32 points) of a C-column Q4 stream is C * 32 floats: the block's first stage starts where its row-major rows would
    const float* baseY = t.dY + blk.pb * t.ldY + t.n0 * 32;
    const float* baseX = t.X + blk.pb * t.ldX + t.k0 * 32;
    const i32x4 da = ds_rsrc(baseY, rows_u * ldY * 4), db = ds_rsrc(baseX, rows_u * ldX * 4);
    const unsigned sa_step = (unsigned)ldY * 128u, sb_step = (unsigned)ldX * 128u;       // bytes per stage: 32 points x C columns
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned kOpBytes = 2u * kDwRows * 256u * 4u;                              // both buffers of one operand
    int dma_stage = 0;
    // piece m = wave + 8 q (q = 0 .. 3) of an operand's stage: 1 KB at m * 1024 in memory and in LDS alike
    auto issue_h = [&](int buf, int q, int which) {
        const unsigned m = wave_u + 8u * (unsigned)q;
        if (which == 0) { if ((int)m < 4 * ca) ds_dma16(da, lds0 + (unsigned)buf * (kDwRows * 1024u) + m * 1024u, (unsigned)lane * 16u + m * 1024u, (unsigned)__builtin_amdgcn_readfirstlane(dma_stage) * sa_step); }
        else            { if ((int)m < 4 * cb) ds_dma16(db, lds0 + kOpBytes + (unsigned)buf * (kDwRows * 1024u) + m * 1024u, (unsigned)lane * 16u + m * 1024u, (unsigned)__builtin_amdgcn_readfirstlane(dma_stage) * sb_step); }
    };
    auto issue = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { issue_h(buf, q, 0); issue_h(buf, q, 1); }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int jn = 4 * wn, jk = arr1 ? wk : 2 * wk;              // first n-tile / k-tile of this wave inside the staged tiles
    const bool active = jn < ca && jk < cb;                      // wave-uniform
    const float* a_rd = smem + (jn * 4) * 256 + lane * 4;                                // piece (tile, g) = (tile * 4 + g) * 256 floats
    const float* b_rd = smem + (kOpBytes / 4) + (jk * 4) * 256 + lane * 4;
    issue(0); dma_stage = 1;
    int buf = 0;
    if (!active) {               // a wave without an output tile (partial 256 x 256 tiles) only moves its share of the pieces and keeps step with the barriers
        for (int p = 0; p < rows_u; p += kDwRows) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (p + kDwRows < rows_u) issue(buf ^ 1);
            ++dma_stage;
            buf ^= 1;
        }
        return;
    }
    // (the MFMA loop is NOT wrapped in `if (active)`: with the accumulators updated on one side of a branch only, the compiler copied all of
    //  them at every stage - 32 to 64 v_mov_b64 per iteration - and, at 2 x 4, spilt 48 of them inside the loop)
    for (int p = 0; p < rows_u; p += kDwRows) {
        const bool more = p + kDwRows < rows_u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of the current stage have landed
        __syncthreads();                                         // ... everyone's have, and everyone is done reading the other buffer
        const float* ar = a_rd + buf * (kDwRows * 256);
        const float* br = b_rd + buf * (kDwRows * 256);
        // per group g (8 points): one conflict-free ds_read_b128 per 32-column tile and lane - four k-steps of that tile - then 32 MFMAs
        // (8-byte reads would keep fewer operand registers live but run at half the LDS rate: lanes l and l + 16 share banks)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 av[4], bv[2];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) av[tn] = *reinterpret_cast<const f32x4*>(ar + (tn * 4 + g) * 256);
            bv[0] = *reinterpret_cast<const f32x4*>(br + g * 256);
            if (!arr1) bv[1] = *reinterpret_cast<const f32x4*>(br + (4 + g) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    acc[tn][0] = CFN_MFMA(av[tn][e], bv[0][e], acc[tn][0]);
                    if (!arr1) acc[tn][1] = CFN_MFMA(av[tn][e], bv[1][e], acc[tn][1]);
                }
                // the next stage's pieces go out one at a time between the first MFMA groups (a burst in front of them keeps both waves of a SIMD off the matrix pipe)
                if (more && g < 2) issue_h(buf ^ 1, (g * 4 + e) >> 1, (g * 4 + e) & 1);
            }
        }
        ++dma_stage;
        buf ^= 1;
    }
    {
        gf_ptr out = (gf_ptr)(partials + (size_t)blk.split * n_params);
        const int kcol = t.k0 + 32 * jk + (lane & 31);
        gf_ptr o = out + ((size_t)t.seg_dst[0] + t.dst_col + kcol);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = t.n0 + 32 * (jn + tn) + frag_row(r, lane);
                if (n < t.N) {
                    gf_ptr row = o + (size_t)n * t.dst_ld;
                    if (kcol < t.K) row[0] = acc[tn][0][r];
                    if (!arr1 && kcol + 32 < t.K) row[32] = acc[tn][1][r];
                }
            }
    }
}

// ONE launch for all arrangements: every workgroup has the same resource footprint (one per CU), so any mix of 2 x 4
// and 1 x 8 blocks fills the chip one block per CU with no second stream and no placement assumptions.
template <int PREC>
__global__ __launch_bounds__(kDwThreads, 2)
void dw_big_kernel(const DwTile* __restrict__ tiles, const DwBlock* __restrict__ blocks, float* __restrict__ partials, int64_t n_params) {
    const DwTile& t = tiles[blocks[blockIdx.x].tile];
    if (PREC == PREC_F32 && t.lay == 3) {
        if (t.gk == 1) dw_big_body_q4<1>(tiles, blocks, partials, n_params);
        else           dw_big_body_q4<0>(tiles, blocks, partials, n_params);
    } else if (t.gk == 1) dw_big_body<PREC, 1>(tiles, blocks, partials, n_params);
    else                  dw_big_body<PREC, 0>(tiles, blocks, partials, n_params);
}

// ---- 4b. small jobs (K or N well below 256: encodings, heads, flow heads).  The 8 waves of a workgroup each own ONE
//      32 x 32 output tile and are arranged GN x GK over the job (4 x 2 for a 128 x 63 encoding block, 2 x 4 for a
//      32 x 128 head slice ...), so only tiles that hold real rows / columns run MFMAs.  Operands go HBM -> LDS by
//      LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass, no vector address arithmetic in
//      the loop) into a THREE-stage ring of 32-point stages, two stages in flight: with one stage in flight the
//      kernel sat at a 3.2 us stage period against 1.9 us of MFMA work - every stage paid an HBM round trip.
//      hipcc does not count asm memory operations, so the waits are explicit (s_waitcnt vmcnt(n) with n = the wave's
//      loads of ONE stage) and the ring is drained before the workgroup ends.
//      LDS image of a stage: [32 rows][a_ld] dY | [32 rows][b_ld] X, both unpadded and row-major, which is exactly the
//      lane-linear order LDS-DMA writes (64 lanes x 16 B = 1 KB per instruction); out-of-range rows (ragged last stage,
//      stages past the block) and columns past the readable row fall outside the descriptor and arrive as zeros.
constexpr int kDsThreads = 512;
constexpr int kDsMaxCols = 192;               // staged columns per row: a_ld (dY) + b_ld (X)
constexpr int kDsStages = 3;
constexpr int kDsSlots = 3;                   // LDS-DMA instructions per wave and stage: 8 * 192 chunks / 64 lanes / 8 waves
__global__ __launch_bounds__(kDsThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void dw_small_kernel(const DwTile* __restrict__ tiles, const DwBlock* __restrict__ blocks, float* __restrict__ partials, int64_t n_params) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DwBlock blk = blocks[blockIdx.x];
    const DwTile t = tiles[blk.tile];
    const int lane = lane_id_opaque(), wave = __builtin_amdgcn_readfirstlane(wave_id());
    const int GK = __builtin_amdgcn_readfirstlane(t.gk), GN = 8 / GK;
    const int tn = min(32 * GN, 128), tk = 32 * GK;
    // staged widths: whole 32-column groups that hold real rows / columns of this tile
    const int a_ld = __builtin_amdgcn_readfirstlane(32 * ((min(tn, t.N - t.n0) + 31) / 32));
    const int b_ld = __builtin_amdgcn_readfirstlane(32 * ((min(tk, t.K - t.k0) + 31) / 32));
    const int stage_floats = kDwRows * (a_ld + b_ld);
    const int rows_u = __builtin_amdgcn_readfirstlane((int)(blk.pe - blk.pb));     // a split's point range is far below 2^31 bytes / row
    const int ldY = __builtin_amdgcn_readfirstlane(t.ldY), ldX = __builtin_amdgcn_readfirstlane(t.ldX);
    const i32x4 ra = ds_rsrc(t.dY + blk.pb * t.ldY, rows_u * ldY * 4);
    const i32x4 rb = ds_rsrc(t.X + blk.pb * t.ldX, rows_u * ldX * 4);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
    // loader: piece j = wave, wave + 8, wave + 16 of the stage's nI = (a_ld + b_ld) / 8 pieces; pieces [0, nA) carry dY
    const int nA = a_ld >> 3, nI = (a_ld + b_ld) >> 3;
    const int n_w = (nI - wave + 7) >> 3;                     // pieces of this wave per stage: 1..3 (nI >= 8)
    // An operand in the Q4 layout (DwTile::lay; cfnerf_device.h) needs no slicing: the 32-point stage of its staged n-tiles is ONE
    // contiguous run of 1-KB pieces, (n-tile, group g) = piece tile * 4 + g, which starts n0 * 32 floats into the half-tile - and the
    // half-tile where the block's row-major rows would start (32 points x C columns either way), so base, size and stage step are shared.
    const bool qa = (t.lay & 1) != 0, qb = (t.lay & 2) != 0;  // wave-uniform
    unsigned voff[kDsSlots], loff[kDsSlots];
    bool is_a[kDsSlots];
#pragma unroll
    for (int q = 0; q < kDsSlots; ++q) {
        const int j = wave + 8 * q;
        is_a[q] = j < nA;
        const int c = 64 * (is_a[q] ? j : j - nA) + lane;     // 16-B chunk inside the operand's stage image
        const int cpr = (is_a[q] ? a_ld : b_ld) >> 2;         // chunks per row: 8, 16, 24 or 32
        const int row = c / cpr, col = (is_a[q] ? t.n0 : t.k0) + 4 * (c - row * cpr);
        const bool ok = col + 4 <= (is_a[q] ? t.Npad : t.Kpad);          // the vector stays inside the readable row
        voff[q] = ok ? (unsigned)((row * (is_a[q] ? t.ldY : t.ldX) + col) * 4) : 0x7ffffff0u;
        if (is_a[q] ? qa : qb) voff[q] = (unsigned)(((is_a[q] ? t.n0 : t.k0) * 32 + c * 4) * 4);
        loff[q] = (unsigned)(((is_a[q] ? 0 : kDwRows * a_ld) + 256 * (is_a[q] ? j : j - nA)) * 4);
    }
    const unsigned sa_step = kDwRows * ldY * 4, sb_step = kDwRows * ldX * 4;
    auto issue = [&](int stage) {                             // stage index: buffer = stage % 3, rows [32 stage, 32 stage + 32)
        const unsigned base = lds0 + (unsigned)((stage % kDsStages) * stage_floats * 4);
#pragma unroll
        for (int q = 0; q < kDsSlots; ++q)
            if (q < n_w) ds_dma16(is_a[q] ? ra : rb, base + loff[q], voff[q], (unsigned)stage * (is_a[q] ? sa_step : sb_step));
    };
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const int gn = wave / GK, gk = wave - gn * GK;
    const int n_base = t.n0 + 32 * gn, k_base = t.k0 + 32 * gk;
    const bool active = gn * 32 < a_ld && n_base < t.N && k_base < t.K;      // wave-uniform
    const int i = lane & 31, kk = lane >> 5;
    // k-step (g, e) of a stage contracts points 8 g + 4 kk + e (lane half kk) - the assignment a Q4 piece has built in: a Q4 operand is
    // ONE ds_read_b128 per group (its tile's piece (tile * 4 + g), this lane's 16 bytes), a row-major one four scalar reads of rows
    // 8 g + 4 kk + e (rounds 1-4 walked rows 2 pp + kk: any order is right as long as both operands use the same)
    const float* a_rd = qa ? smem + (gn * 4) * 256 + lane * 4 : smem + 4 * kk * a_ld + 32 * gn + i;
    const float* b_rd = smem + kDwRows * a_ld + (qb ? (gk * 4) * 256 + lane * 4 : 4 * kk * b_ld + 32 * gk + i);
    // one stage = 16 k-steps: the operands are fetched from LDS into registers as a batch, so the ds_reads pipeline
    // instead of each MFMA waiting on its own read
    auto compute = [&](int stage) {
        if (!active) return;
        const float* ar = a_rd + (stage % kDsStages) * stage_floats;
        const float* br = b_rd + (stage % kDsStages) * stage_floats;
        float av[16], bv[16];
        if (qa) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { const f32x4 v = *reinterpret_cast<const f32x4*>(ar + g * 256); av[4 * g] = v[0]; av[4 * g + 1] = v[1]; av[4 * g + 2] = v[2]; av[4 * g + 3] = v[3]; }
        } else {
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) av[pp] = ar[(8 * (pp >> 2) + (pp & 3)) * a_ld];
        }
        if (qb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { const f32x4 v = *reinterpret_cast<const f32x4*>(br + g * 256); bv[4 * g] = v[0]; bv[4 * g + 1] = v[1]; bv[4 * g + 2] = v[2]; bv[4 * g + 3] = v[3]; }
        } else {
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) bv[pp] = br[(8 * (pp >> 2) + (pp & 3)) * b_ld];
        }
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) acc[pp & 1] = CFN_MFMA(av[pp], bv[pp], acc[pp & 1]);   // two chains, summed at the end
    };

    const int n_st = (rows_u + kDwRows - 1) / kDwRows;
    issue(0);
    issue(1);                                                 // (past the block's end: every lane out of range, zeros, no traffic)
    for (int s = 0; s < n_st; ++s) {
        ds_wait_stage(n_w);                                   // this wave's pieces of stage s have landed (stage s + 1 may be in flight)
        __syncthreads();                                      // ... everyone's have, and everyone is done reading stage s - 1
        issue(s + 2);                                         // into the buffer stage s - 1 occupied
        compute(s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the ring: LDS-DMA must not land after the workgroup has ended
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
    gf_ptr out = (gf_ptr)(partials + (size_t)blk.split * n_params);
    const int k = k_base + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int n = n_base + frag_row(r, lane);
        bool live = n < t.N && k < t.K;
        if (t.row_f) {                                          // theta-head tile: kernel column 4 b + f -> row b F + f, f >= F dropped
            const int f = n & (kFlowsMax - 1);
            live = live && f < t.row_f;
            n = (n >> 2) * t.row_f + f;
        }
        if (live) {
            int sg = 0;
            if (t.nseg > 1 && n >= t.seg_row[1]) sg = 1;
            if (t.nseg > 2 && n >= t.seg_row[2]) sg = 2;
            if (t.nseg > 3 && n >= t.seg_row[3]) sg = 3;
            const uint32_t dst = sg == 0 ? t.seg_dst[0] : sg == 1 ? t.seg_dst[1] : sg == 2 ? t.seg_dst[2] : t.seg_dst[3];
            const int row0 = sg == 0 ? t.seg_row[0] : sg == 1 ? t.seg_row[1] : sg == 2 ? t.seg_row[2] : t.seg_row[3];
            out[(size_t)dst + (size_t)(n - row0) * t.dst_ld + t.dst_col + k] = acc[0][r];
        }
    }
}

// ================================================================================================
// 5. reductions into grad_flat
// grad[i] = sum over the split slots that the tensor containing i actually uses (segments sorted by offset)
// phase 1: only the tensors whose tiles all belong to the big launches (final before the small-job launch); phase 0: the rest;
// phase 2: every tensor in one launch (the caller never asked for the early ranges: nothing waits between the two)
// accumulate (cfnerf_render_bwd_accumulate): the sums are ADDED to grad instead of stored (a batch walked in slices)
__global__ void reduce_weights_kernel(const float* __restrict__ partials, const RedSeg* __restrict__ segs, int n_segs, int64_t n_params,
                                      float* __restrict__ grad, int phase, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_params) return;
    int lo = 0, hi = n_segs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int64_t)segs[mid].begin <= i) lo = mid; else hi = mid - 1; }
    const int ns = segs[lo].nsplit;
    if (ns < 0 || (phase != 2 && segs[lo].early != phase)) return;
    // 8 independent loads per trip (the slots of one element are n_params apart: latency-bound otherwise); fixed
    // summation order, so the result is deterministic
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* q = partials + i;
    int k = 0;
    for (; k + 7 < ns; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = q[(size_t)(k + u) * n_params];
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
    }
    for (; k < ns; ++k) s0 += q[(size_t)k * n_params];
    const float r = (s0 + s1) + (s2 + s3);
    grad[i] = accumulate ? grad[i] + r : r;
}

__global__ __launch_bounds__(1024)
void reduce_bias_kernel(const float* __restrict__ dbp, int n_wg, int nb, const BiasMap* __restrict__ maps, int n_maps,
                        float* __restrict__ grad, int accumulate) {
    __shared__ float sh[16][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;       // 16 row groups x 64 columns
    const int j = blockIdx.x * 64 + lane;                             // column of the bias partial table
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < nb) {
        int w = part;
        for (; w + 48 < n_wg; w += 64) {                              // 4 independent loads in flight per lane
            s0 += dbp[(size_t)w * nb + j]; s1 += dbp[(size_t)(w + 16) * nb + j];
            s2 += dbp[(size_t)(w + 32) * nb + j]; s3 += dbp[(size_t)(w + 48) * nb + j];
        }
        for (; w < n_wg; w += 16) s0 += dbp[(size_t)w * nb + j];
    }
    sh[part][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && j < nb) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += sh[q][lane];
        for (int q = 0; q < n_maps; ++q)
            if (j >= maps[q].col0 && j < maps[q].col0 + maps[q].count) {
                float* g = grad + maps[q].dst + (j - maps[q].col0);
                *g = accumulate ? *g + tot : tot;
            }
    }
}

// base-Gaussian parameters: chain through z0 = eps*std + mean (tail partials) + d mean(base log-normal)/d std = -1/std
__global__ void reduce_gms_kernel(const float* __restrict__ gms, int64_t n_rows, const float* __restrict__ flat,
                                  const float* __restrict__ d_ent, float* __restrict__ grad, int accumulate) {
    __shared__ double sh[8][256];
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r = threadIdx.x; r < n_rows; r += blockDim.x)
        for (int i = 0; i < 8; ++i) s[i] += gms[r * 8 + i];
    for (int i = 0; i < 8; ++i) sh[i][threadIdx.x] = s[i];
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) for (int i = 0; i < 8; ++i) sh[i][threadIdx.x] += sh[i][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ge = d_ent ? d_ent[0] : 0.f;
        float r[8];
        r[0] = (float)sh[0][0];                                          // alpha_mean
        r[1] = (float)sh[1][0] + ge * (-1.f / flat[1]);                  // alpha_std: + d mean(base_a)/d std
        for (int c = 0; c < 3; ++c) {
            r[2 + c] = (float)sh[2 + c][0];                              // rgb_mean
            r[5 + c] = (float)sh[5 + c][0] + ge * (-1.f / (3.f * flat[5 + c]));      // rgb_std (mean over 3 channels, MOD:283,286)
        }
        for (int i = 0; i < 8; ++i) grad[i] = accumulate ? grad[i] + r[i] : r[i];
    }
}

// ================================================================================================
// 6. Adam (torch.optim.Adam defaults, RUN:339)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr_over_bc1, float inv_sqrt_bc2, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = 0.9f * m[i] + (1.f - 0.9f) * gi;
    const float vi = 0.999f * v[i] + (1.f - 0.999f) * (gi * gi);
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + 1e-08f;
    p[i] = p[i] - lr_over_bc1 * (mi / denom);
}

// ================================================================================================
// host side
static const void* bwd_data_fn(int W, bool b16) {
    switch (W) {
#define CFN_W_CASE(w) case w: return b16 ? reinterpret_cast<const void*>(bwd_data_kernel<w, PREC_BF16X3>) : reinterpret_cast<const void*>(bwd_data_kernel<w, PREC_F32>);
        CFN_FOR_EACH_WIDTH(CFN_W_CASE)
#undef CFN_W_CASE
    }
    return nullptr;
}

static hipError_t launch_bwd_data(const BwdArgs& a, const NetTab& ht, int prec, hipStream_t st, int* grid_out) {
    const size_t lds = bwd_lds_bytes(ht.W, ht.ha_sz);
    const void* fn = bwd_data_fn(ht.W, prec == PREC_BF16X3);
    if (!fn) return hipErrorInvalidValue;
    int grid = (int)std::min<int64_t>(a.n_tiles, (int64_t)a.n_wg);
    if (grid_out) *grid_out = grid;
    void* args[] = {const_cast<BwdArgs*>(&a), const_cast<NetTab*>(&ht)};
    return hipLaunchKernel(fn, dim3(grid), dim3(kThreads), args, lds, st);
}

constexpr size_t kDwBigLds = 4 * kDwRows * 256 * sizeof(float);
// (Asking for MORE dynamic LDS than the kernel uses - 96 KB, to keep two 1 x 8 workgroups off one CU - made that launch
// produce wrong sums on this stack; the XCD-granular budgets of balance_big_splits make it unnecessary anyway.)
constexpr size_t kDwSmallLds = (size_t)kDsStages * kDwRows * kDsMaxCols * sizeof(float);

// Per-DEVICE set-up of the backward kernels of one width (called from cfnerf_model_create with that device current)
hipError_t bwd_set_attributes(int W, int ha) {
    const size_t lds = bwd_lds_bytes(W, ha);
    for (int b16 = 0; b16 < 2; ++b16) {
        hipError_t e = hipFuncSetAttribute(bwd_data_fn(W, b16 != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const void* bigs[2] = {reinterpret_cast<const void*>(dw_big_kernel<PREC_F32>), reinterpret_cast<const void*>(dw_big_kernel<PREC_BF16X3>)};
    hipError_t e = hipSuccess;
    for (const void* fn : bigs) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDwBigLds);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dw_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDwSmallLds);
}

}  // namespace cfnerf

using namespace cfnerf;

static thread_local char g_berr[512] = "";
extern "C" const char* cfnerf_last_error(void);
// error text is shared with the ABI file through this hook
extern "C" void cfnerf_set_error_(const char* msg);
static int bfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_berr, sizeof g_berr, fmt, ap);
    va_end(ap);
    cfnerf_set_error_(g_berr);
    return code;
}
#define BHIP(expr)                                                                               \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return bfail(CFNERF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// build (once per model) the weight-gradient tile list, the bias map and their device copies
int cfnerf::ensure_bwd_plan(cfnerf_model* m) {
    BwdPlan& B = m->bwd;
    if (B.built) return 0;
    const cfnerf_cfg& c = m->cfg;
    const ParamLayout& L = m->layout;
    const int W = c.netwidth, D = c.netdepth, HA = c.h_alpha_size, HR = c.h_rgb_size, F = c.n_flows;
    const int ic = enc_ch(c.multires), icv = enc_ch(c.multires_views), skip = skip_layer(D);
    // bias partial table columns
    int nb = 0;
    B.db_h = nb; nb += D * W;
    B.db_feat = nb; nb += W;
    B.db_v = nb; nb += W / 2;
    B.db_ha = nb; nb += pad_to(HA, 32);
    B.db_hr = nb; nb += pad_to(HR, 32);
    B.db_theta = nb; nb += kThetaAll;
    B.nb = nb;
    char key[64];
    auto bmap = [&](int col0, int count, const char* k) {
        BiasMap bm; bm.col0 = col0; bm.count = count; bm.dst = (uint32_t)L.off(k);
        B.bias_maps.push_back(bm);
    };
    for (int l = 0; l < D; ++l) { std::snprintf(key, sizeof key, "pts_linears.%d.bias", l); bmap(B.db_h + l * W, W, key); }
    bmap(B.db_feat, W, "feature_linear.bias");
    bmap(B.db_v, W / 2, "views_linears.0.bias");
    bmap(B.db_ha, HA, "h_alpha_linear.bias");
    bmap(B.db_hr, HR, "h_rgb_linear.bias");
    {   // theta heads: one map per block of F rows (kernel columns base + 4 b + f)
        const char* kr[4] = {"flows_rgb.amor_d.bias", "flows_rgb.amor_diag1.0.bias", "flows_rgb.amor_diag2.0.bias", "flows_rgb.amor_b.bias"};
        const int base_r[4] = {0, 9 * kFlowsMax, 12 * kFlowsMax, 15 * kFlowsMax}, blocks_r[4] = {9, 3, 3, 3};
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < blocks_r[i]; ++b) {
                BiasMap bm; bm.col0 = B.db_theta + base_r[i] + kFlowsMax * b; bm.count = F; bm.dst = (uint32_t)(L.off(kr[i]) + b * F);
                B.bias_maps.push_back(bm);
            }
        const char* ka[3] = {"flows_alpha.amor_diag1.0.bias", "flows_alpha.amor_diag2.0.bias", "flows_alpha.amor_b.bias"};
        for (int i = 0; i < 3; ++i) bmap(B.db_theta + kThetaRgb + kFlowsMax * i, F, ka[i]);
    }
    B.built = true;
    return 0;
}

// Which kernel writes a tensor's gradient, and when it is final:
//   biases (reduce_bias, right after bwd_data) and the base Gaussians (reduce_gms, right after the tail): nsplit = -1, the
//   weight reduction skips them; dead tensors: zeros, written with the early phase; a weight fed only by big tiles: early.
// B.early_off / early_cnt: the merged flat ranges of grad_flat that are final when ev_early fires (before dw_small runs),
// so that a multi-GPU caller can start exchanging them while the small jobs still compute.
static void finish_segs(const ParamLayout& L, BwdPlan& B, DwHost& Hs) {
    for (size_t i = 0; i < Hs.segs.size(); ++i) {
        RedSeg& r = Hs.segs[i];
        bool by_other = i < 4;                                   // alpha_mean, alpha_std, rgb_mean, rgb_std: reduce_gms
        for (const BiasMap& bm : B.bias_maps) by_other = by_other || bm.dst == r.begin;
        if (by_other) { r.nsplit = -1; r.early = 1; }
        else if (r.nsplit == 0) r.early = 1;
    }
    B.early_off.clear(); B.early_cnt.clear();
    for (size_t i = 0; i < Hs.segs.size(); ++i) {
        if (!Hs.segs[i].early) continue;
        const int64_t b = Hs.segs[i].begin, e = (i + 1 < Hs.segs.size()) ? (int64_t)Hs.segs[i + 1].begin : L.total;
        if (!B.early_off.empty() && B.early_off.back() + B.early_cnt.back() == b) B.early_cnt.back() += e - b;
        else { B.early_off.push_back(b); B.early_cnt.push_back(e - b); }
    }
}

extern "C" {

int cfnerf_loss_fwd_bwd(const float* rgb_map, const float* target, const float* entropy, int64_t N, int K, float beta1,
                        int64_t n_total, float* d_rgb_map, float* scalars_out, cfnerf_stream s) {
    if (N < 0 || K < 1 || n_total < N) return bfail(CFNERF_E_INVALID, "bad N/K/n_total");
    if (N == 0) return CFNERF_OK;
    if (!rgb_map || !target || !d_rgb_map || !scalars_out) return bfail(CFNERF_E_INVALID, "NULL argument");
    if (reinterpret_cast<uintptr_t>(scalars_out) % 8) return bfail(CFNERF_E_INVALID, "scalars_out must be 8-byte aligned");
    hipStream_t st = (hipStream_t)s;
    BHIP(hipMemsetAsync(scalars_out, 0, 4 * sizeof(float), st));        // the two 64-bit fixed-point accumulators live in these 16 bytes
    if (K >= kLossWideK)
        hipLaunchKernelGGL(loss_kernel<8>, dim3((unsigned)((N * 3 * 8 + kLossThreads - 1) / kLossThreads)), dim3(kLossThreads), 0, st, rgb_map, target,
                           N, K, n_total, d_rgb_map, reinterpret_cast<long long*>(scalars_out));
    else
        hipLaunchKernelGGL(loss_kernel<1>, dim3((unsigned)((N * 3 + kLossThreads - 1) / kLossThreads)), dim3(kLossThreads), 0, st, rgb_map, target,
                           N, K, n_total, d_rgb_map, reinterpret_cast<long long*>(scalars_out));
    BHIP(hipGetLastError());
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, st, entropy, beta1, n_total, scalars_out);
    BHIP(hipGetLastError());
    return CFNERF_OK;
}

// Backward of the stashed forward, shared by the fused (cfnerf_render_bwd: rays, `d_out` = d_rgb_map) and the unfused
// (cfnerf_network_bwd: points, `d_out` = d_raw) entry points: first stage -> g_theta, then backward-data and the weight gradients.
static int backward_stashed(cfnerf_model* m, bool points, uint64_t stash_generation, const float* d_out, const float* d_depth_map,
                            const float* d_entropy, float* grad_flat, int accumulate, cfnerf_stream s) {
    Stash& q = m->stash;
    if (!q.valid) return bfail(CFNERF_E_INVALID, "no stashed forward: call cfnerf_render_fwd / cfnerf_network_fwd with CFNERF_F_STASH first");
    if (stash_generation != q.generation)
        return bfail(CFNERF_E_INVALID, "stale stash: this backward belongs to STASH forward #%llu but the model's one stash now holds "
                     "forward #%llu (a later grad-enabled forward overwrote it; run each backward before the next STASH forward)",
                     (unsigned long long)stash_generation, (unsigned long long)q.generation);
    if (q.points != points)
        return bfail(CFNERF_E_INVALID, points ? "the stashed forward is a cfnerf_render_fwd (ray) launch: differentiate it with cfnerf_render_bwd"
                                              : "the stashed forward is a cfnerf_network_fwd (points) launch: differentiate it with cfnerf_network_bwd");
    if (!points && q.S > 4096) return bfail(CFNERF_E_UNSUPPORTED, "backward supports S <= 4096");
    hipStream_t st = (hipStream_t)s;
    if (int rc = ensure_bwd_plan(m)) return rc;
    BwdPlan& B = m->bwd;
    const cfnerf_cfg& c = m->cfg;
    const int W = c.netwidth;
    const int64_t N = q.N, P = q.N * (int64_t)q.S, n_params = m->layout.total;
    const ParamLayout& L = m->layout;
    const int n_wg = std::min(m->n_cu, kMaxCu) * ((W <= 256) ? 2 : 1);

    // ---- weight-gradient descriptors: they hold workspace pointers, so they are rebuilt (and uploaded, asynchronously,
    //      from host vectors that outlive the copy) only when the workspace binding moved - never on the steady path
    if (B.bind_serial != q.bind_serial || B.q4 != q.q4) {      // (the tile descriptors carry operand pointers AND operand layouts)
        B.cur ^= 1;
        DwHost& Hs = B.host[B.cur];
        if (Hs.uploaded) BHIP(hipEventSynchronize(Hs.uploaded));     // the upload made from THIS set two rebuilds ago is long done
        else BHIP(hipEventCreateWithFlags(&Hs.uploaded, hipEventDisableTiming));
        int ns_max = 1;
        if (const char* why = build_dw_plan(c, L, q, P, m->n_cu, Hs, &B.n_blocks_wide, &ns_max)) return bfail(CFNERF_E_UNSUPPORTED, "%s", why);
        finish_segs(L, B, Hs);
        if (Hs.segs.size() > 256) return bfail(CFNERF_E_UNSUPPORTED, "too many parameter tensors");
        BHIP(hipMemsetAsync(q.partials, 0, (size_t)ns_max * n_params * sizeof(float), st));
        auto up = [&](void* dst, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess; };
        BHIP(up(q.tiles, Hs.tiles.data(), Hs.tiles.size() * sizeof(DwTile)));
        BHIP(up(q.tiles_small, Hs.tiles_small.data(), Hs.tiles_small.size() * sizeof(DwTile)));
        BHIP(up(q.blocks, Hs.blocks.data(), Hs.blocks.size() * sizeof(DwBlock)));
        BHIP(up(q.blocks_small, Hs.blocks_small.data(), Hs.blocks_small.size() * sizeof(DwBlock)));
        BHIP(up(q.segs, Hs.segs.data(), Hs.segs.size() * sizeof(RedSeg)));
        BHIP(up(q.bias_maps, B.bias_maps.data(), B.bias_maps.size() * sizeof(BiasMap)));
        BHIP(hipEventRecord(Hs.uploaded, st));
        B.bind_serial = q.bind_serial;
        B.q4 = q.q4;
    }
    const DwHost& Hc = B.host[B.cur];

    // ---- 1. first stage -> g_theta (+ base-Gaussian partials)
    int ksplit = 1;
    int64_t gms_rows = 0;
    if (m->timing == 1) BHIP(hipEventRecord(m->ev0[1], st));
    if (!points) {
        TailArgs ta{};
        ta.raw = q.raw; ta.theta = q.theta; ta.at = q.at; ta.z = q.z; ta.rays = q.rays; ta.eps = m->d_eps; ta.flat = m->flat;
        ta.d_rgb = d_out; ta.d_depth = d_depth_map; ta.d_ent = d_entropy; ta.N = N; ta.P = P; ta.S = q.S; ta.K = q.K; ta.flags = q.flags;
        ta.g_theta = q.g_theta; ta.gms_partials = q.gms;
        ksplit = tail_parts(N, q.K, std::min(m->n_cu, kMaxCu));      // 1, 2 or 4: the parts of a ray are waves of ONE 4-wave workgroup and
        ta.ksplit = ksplit;                                          // meet in LDS, so ONE g_theta row per point leaves the kernel
        gms_rows = N * ksplit;
        BHIP(launch_tail_bwd(ta, N, ksplit, st));
    } else {
        unsigned grid = 0;
        BHIP(launch_flows_bwd(q.raw, q.theta, m->d_eps, m->flat, d_out, d_entropy, P, q.K, q.g_theta, q.gms, &grid, st));
        gms_rows = (int64_t)grid * kWaves;                       // one row per wave (waves past P contribute zeros)
    }
    hipLaunchKernelGGL(reduce_gms_kernel, dim3(1), dim3(256), 0, st, q.gms, gms_rows, m->flat, d_entropy, grad_flat, accumulate);
    BHIP(hipGetLastError());
    if (m->timing == 1) BHIP(hipEventRecord(m->ev1[1], st));

    // ---- 2. fused backward-data (+ bias partials and their reduction: every bias gradient is final here)
    BwdArgs ba{};                                              // (q.dbp needs no memset: every launched workgroup starts its row on its first tile)
    ba.wp = m->d_packed; ba.wp16 = m->d_packed16; ba.P = P; ba.n_wg = n_wg; ba.nb = B.nb;
    ba.q4 = q.q4 ? 1 : 0;
    ba.g_theta = q.g_theta; ba.g_hr = q.g_hr; ba.g_ha = q.g_ha; ba.g_v = q.g_v; ba.g_feat = q.g_feat; ba.g_h = q.g_h;
    ba.mbits = reinterpret_cast<const uint32_t*>(q.mbits); ba.n_tiles = q.n_tiles; ba.S = q.S; ba.dbp = q.dbp;      // (points: ONE "ray" of S = P samples)
    ba.db_h = B.db_h; ba.db_feat = B.db_feat; ba.db_v = B.db_v; ba.db_ha = B.db_ha; ba.db_hr = B.db_hr; ba.db_theta = B.db_theta;
    int grid_bd = 0;
    if (m->timing == 1) BHIP(hipEventRecord(m->ev0[2], st));
    BHIP(launch_bwd_data(ba, m->plan.tab, m->precision, st, &grid_bd));
    if (m->timing == 1) BHIP(hipEventRecord(m->ev1[2], st));
    hipLaunchKernelGGL(reduce_bias_kernel, dim3((unsigned)((B.nb + 63) / 64)), dim3(1024), 0, st, q.dbp, grid_bd, B.nb,
                       q.bias_maps, (int)B.bias_maps.size(), grad_flat, accumulate);
    BHIP(hipGetLastError());

    // ---- 3. weight gradients + reductions
    if (m->timing == 1) BHIP(hipEventRecord(m->ev0[3], st));
    if (!Hc.blocks.empty()) {
        if (m->precision == PREC_BF16X3)
            hipLaunchKernelGGL(dw_big_kernel<PREC_BF16X3>, dim3((unsigned)Hc.blocks.size()), dim3(kDwThreads), kDwBigLds, st,
                               q.tiles, q.blocks, q.partials, n_params);
        else
            hipLaunchKernelGGL(dw_big_kernel<PREC_F32>, dim3((unsigned)Hc.blocks.size()), dim3(kDwThreads), kDwBigLds, st,
                               q.tiles, q.blocks, q.partials, n_params);
        BHIP(hipGetLastError());
    }
    // tensors fed by big tiles only (+ the zeros of dead tensors): final now.  ev_early lets a multi-GPU caller start
    // exchanging B.early_off / early_cnt while the small jobs below still compute - IF it ever asked for those ranges
    // (cfnerf_grad_early_ranges: the two-bucket exchange).  Otherwise nothing can be waiting between the two reductions and they
    // are ONE launch after the small jobs (the same sums in the same order; one launch of ~14 us less per step).
    const unsigned red_grid = (unsigned)((n_params + 255) / 256);
    if (!B.ev_early) BHIP(hipEventCreateWithFlags(&B.ev_early, hipEventDisableTiming));
    if (B.early_wanted) {
        hipLaunchKernelGGL(reduce_weights_kernel, dim3(red_grid), dim3(256), 0, st, q.partials, q.segs, (int)Hc.segs.size(), n_params, grad_flat, 1, accumulate);
        BHIP(hipGetLastError());
        BHIP(hipEventRecord(B.ev_early, st));
    }
    if (!Hc.blocks_small.empty()) {
        hipLaunchKernelGGL(dw_small_kernel, dim3((unsigned)Hc.blocks_small.size()), dim3(kDsThreads), kDwSmallLds, st,
                           q.tiles_small, q.blocks_small, q.partials, n_params);
        BHIP(hipGetLastError());
    }
    hipLaunchKernelGGL(reduce_weights_kernel, dim3(red_grid), dim3(256), 0, st, q.partials, q.segs, (int)Hc.segs.size(), n_params, grad_flat,
                       B.early_wanted ? 0 : 2, accumulate);
    BHIP(hipGetLastError());
    if (!B.early_wanted) BHIP(hipEventRecord(B.ev_early, st));      // (a waiter on the event still sees final gradients: everything is final here)
    if (m->timing == 1) BHIP(hipEventRecord(m->ev1[3], st));
    return CFNERF_OK;
}

int cfnerf_render_bwd(cfnerf_model* m, uint64_t stash_generation, const float* d_rgb_map, const float* d_depth_map,
                      const float* d_entropy, float* grad_flat, cfnerf_stream s) {
    if (!m || !d_rgb_map || !grad_flat) return bfail(CFNERF_E_INVALID, "NULL argument");
    return backward_stashed(m, false, stash_generation, d_rgb_map, d_depth_map, d_entropy, grad_flat, 0, s);
}

int cfnerf_render_bwd_accumulate(cfnerf_model* m, uint64_t stash_generation, const float* d_rgb_map, const float* d_depth_map,
                                 const float* d_entropy, float* grad_flat, cfnerf_stream s) {
    if (!m || !d_rgb_map || !grad_flat) return bfail(CFNERF_E_INVALID, "NULL argument");
    return backward_stashed(m, false, stash_generation, d_rgb_map, d_depth_map, d_entropy, grad_flat, 1, s);
}

int cfnerf_network_bwd(cfnerf_model* m, uint64_t stash_generation, const float* d_raw, const float* d_entropy, float* grad_flat,
                       cfnerf_stream s) {
    if (!m || !grad_flat) return bfail(CFNERF_E_INVALID, "NULL argument");
    if (!d_raw && !d_entropy) return bfail(CFNERF_E_INVALID, "d_raw and d_entropy are both NULL: nothing to differentiate");
    return backward_stashed(m, true, stash_generation, d_raw, nullptr, d_entropy, grad_flat, 0, s);
}

int cfnerf_composite_bwd(const float* raw, const float* z_vals, const float* rays_d, int64_t N, int S, int K, int white_bkgd,
                         const float* d_rgb_map, const float* d_disp_map, const float* d_depth_map, const float* d_weights, float* d_raw,
                         cfnerf_stream s) {
    if (N < 0 || S < 1 || K < 1) return bfail(CFNERF_E_INVALID, "bad N/S/K");
    if (S > 64 * kCompMaxChunks) return bfail(CFNERF_E_UNSUPPORTED, "cfnerf_composite_bwd supports S <= %d", 64 * kCompMaxChunks);
    if (N == 0) return CFNERF_OK;
    if (!raw || !z_vals || !rays_d || !d_rgb_map || !d_raw) return bfail(CFNERF_E_INVALID, "NULL argument");
    if ((int64_t)S * K * 16 >= (1ll << 31)) return bfail(CFNERF_E_UNSUPPORTED, "cfnerf_composite_bwd: S * K * 16 bytes per ray must stay below 2 GiB");
#define CFN_COMPB(KG, FAST) hipLaunchKernelGGL((composite_bwd_kernel<KG, FAST>), dim3((unsigned)((N + kWaves - 1) / kWaves)), dim3(kThreads), 0, \
                       (hipStream_t)s, raw, z_vals, rays_d, N, S, K, white_bkgd, d_rgb_map, d_disp_map, d_depth_map, d_weights, d_raw)
    if (K <= 4) CFN_COMPB(4, false);                // (the same split as launch_composite: forward and adjoint share their arithmetic)
    else if (K < kFastFlowsK) CFN_COMPB(8, false);
    else CFN_COMPB(8, true);
#undef CFN_COMPB
    BHIP(hipGetLastError());
    return CFNERF_OK;
}

// The flat ranges of grad_flat that are final when the early event of the LAST cfnerf_render_bwd fires (they do not
// depend on the batch: only on which tensors are fed by big tiles).  Returns the number of ranges (<= max), < 0 on error.
int cfnerf_grad_early_ranges(cfnerf_model* m, int64_t* offsets, int64_t* counts, int max_ranges) {
    if (!m || !offsets || !counts) return bfail(CFNERF_E_INVALID, "NULL argument");
    if (m->bwd.early_off.empty()) return bfail(CFNERF_E_INVALID, "no backward has run on this model yet");
    const int n = (int)m->bwd.early_off.size();
    if (n > max_ranges) return bfail(CFNERF_E_INVALID, "%d ranges, room for %d", n, max_ranges);
    for (int i = 0; i < n; ++i) { offsets[i] = m->bwd.early_off[i]; counts[i] = m->bwd.early_cnt[i]; }
    m->bwd.early_wanted = true;            // from the next backward on the early tensors are reduced (and ev_early recorded) BEFORE the small-job launch
    return n;
}

int cfnerf_stream_wait_grad_early(cfnerf_model* m, cfnerf_stream waiter) {
    if (!m) return bfail(CFNERF_E_INVALID, "model is NULL");
    if (!m->bwd.ev_early) return bfail(CFNERF_E_INVALID, "no backward has run on this model yet");
    BHIP(hipStreamWaitEvent((hipStream_t)waiter, m->bwd.ev_early, 0));
    return CFNERF_OK;
}

int cfnerf_adam_step(cfnerf_model* m, float* flat_params, const float* grad_flat, float* exp_avg, float* exp_avg_sq,
                     int64_t step, float lr, float grad_scale, cfnerf_stream s) {
    if (!m || !flat_params || !grad_flat || !exp_avg || !exp_avg_sq) return bfail(CFNERF_E_INVALID, "NULL argument");
    if (step < 1) return bfail(CFNERF_E_INVALID, "step is 1-based");
    hipStream_t st = (hipStream_t)s;
    const int64_t n = m->layout.total;
    const double bc1 = 1.0 - std::pow(0.9, (double)step), bc2 = 1.0 - std::pow(0.999, (double)step);
    if (m->timing == 1) BHIP(hipEventRecord(m->ev0[4], st));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flat_params, grad_flat, exp_avg, exp_avg_sq, n,
                       (float)(lr / bc1), (float)(1.0 / std::sqrt(bc2)), grad_scale);
    BHIP(hipGetLastError());
    int rc = cfnerf_model_set_params(m, flat_params, s);
    if (m->timing == 1) BHIP(hipEventRecord(m->ev1[4], st));
    return rc;
}

}  // extern "C"
