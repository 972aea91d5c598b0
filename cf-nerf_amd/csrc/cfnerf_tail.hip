// cfnerf_tail.hip - the flow-adjoint kernels of the train step for gfx950, a translation unit of their own:
//   tail_bwd_kernel   (fused path)   adjoint of raw2outputs (RUN:424-452) + of the K conditional flows (FLW:225-268, MOD:263-286)
//   flows_bwd_kernel  (unfused seam) adjoint of the K flows + entropy terms of NeRF_Flows.forward (MOD:225-291) given d loss / d raw
// Both are long straight-line scalar code (the unrolled adjoint of four 3 x 3 triangular flow steps per (sample, latent)) on one wave per
// SIMD.  This file is compiled with -fno-slp-vectorize (cf-nerf_amd/build.py): left on, the SLP vectoriser pairs the arithmetic into
// v_pk_mul / v_pk_add / v_pk_fma and then needs ~200 v_mov per (chunk, latent) to line the operands up in register pairs; that pushes the
// tail kernel to 367 ArchVGPRs + 111 AccVGPRs with another ~230 v_accvgpr moves between the two files (a vector instruction addresses
// only the first 256).  Without it: 255 VGPRs, no AccVGPR, 1 193 -> 1 103 vector instructions per (chunk, latent): -8.5 % at K = 64
// (0.280 -> 0.256 ms), -0.7 % at K = 4 (round 4, tools/ab_kernels.sh).  The rest of the backward keeps the vectoriser: switched off for
// the whole of cfnerf_bwd.hip the weight-gradient kernels ran 0.5 % slower.
#include "cfnerf_device.h"
#include "cfnerf_kernels.h"
#include "cfnerf_model.h"
#include "cfnerf_bwd.h"

namespace cfnerf {

// ================================================================================================
// 2. tail: adjoint of raw2outputs (RUN:424-452) and of the K flows (FLW:225-268, MOD:263-286).
//    One wave per ray, lane = sample, chunks of 64 samples walked back-to-front so the transmittance
//    adjoint (comp_adjoint_D, cfnerf_device.h) is a reverse wave scan + a per-k carry.
// Adjoint of the four conditional Sylvester flows (FLW:225-268, MOD:401-413) for ONE (point, latent sample): recomputes the
// forward keeping each step's input and tanh, then walks it backwards.  In: th = the point's flow parameters, e = the latent,
// ga / gz = d loss / d (alpha, rgb) flow outputs (activation and entropy-Jacobian terms already added), cE = the weight of the
// log-det terms (- d_entropy / (P K)).  Accumulates d loss / d theta into gth and the base-Gaussian terms into gms.  Shared by
// the fused tail kernel and the unfused flows_bwd_kernel, so both differentiate with the same arithmetic.
template <class GACC>
__device__ __forceinline__ void flows_adjoint(const float (&th)[84], GACC& gth, float (&gms)[8], const f32x4 e, const float a_mean,
                                              const float a_std, const float (&r_mean)[3], const float (&r_std)[3], float ga, float (&gz)[3],
                                              const float cE, const bool valid) {
#pragma clang fp contract(fast)      // gradient arithmetic: a*b + c may fuse (the library is built with -ffp-contract=off for the FORWARD's
                                     // parity with torch's separate ops; the backward has no such constraint and fused pairs halve its
                                     // multiply / add instruction count)
    // ---- recompute the flows, keeping each step's input and tanh
    float zin[4][3], tt[4][3], ain[4], ta[4];
    float z[3] = {e[0] * r_std[0] + r_mean[0], e[1] * r_std[1] + r_mean[1], e[2] * r_std[2] + r_mean[2]};
    float a = e[3] * a_std + a_mean;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const bool odd = f & 1;
        zin[f][0] = z[0]; zin[f][1] = z[1]; zin[f][2] = z[2]; ain[f] = a;
        const float zp0 = odd ? z[2] : z[0], zp1 = z[1], zp2 = odd ? z[0] : z[2];
        const float pre0 = ((th[48 + f] * zp0 + th[(1 * 3 + 0) * 4 + f] * zp1) + th[(2 * 3 + 0) * 4 + f] * zp2) + th[60 + f];
        const float pre1 = (th[52 + f] * zp1 + th[(2 * 3 + 1) * 4 + f] * zp2) + th[64 + f];
        const float pre2 = th[56 + f] * zp2 + th[68 + f];
        const float t0 = t_tanh(pre0), t1 = t_tanh(pre1), t2 = t_tanh(pre2);
        tt[f][0] = t0; tt[f][1] = t1; tt[f][2] = t2;
        const float u0 = (th[36 + f] * t0 + th[(0 * 3 + 1) * 4 + f] * t1) + th[(0 * 3 + 2) * 4 + f] * t2;
        const float u1 = th[40 + f] * t1 + th[(1 * 3 + 2) * 4 + f] * t2;
        const float u2 = th[44 + f] * t2;
        z[0] = (odd ? u2 : u0) + z[0]; z[1] = u1 + z[1]; z[2] = (odd ? u0 : u2) + z[2];
        ta[f] = t_tanh(th[76 + f] * a + th[80 + f]);
        a = th[72 + f] * ta[f] + a;
    }
    // ---- adjoint, last flow first
#pragma unroll
    for (int f = 3; f >= 0; --f) {
        const bool odd = f & 1;
        const float zp0 = odd ? zin[f][2] : zin[f][0], zp1 = zin[f][1], zp2 = odd ? zin[f][0] : zin[f][2];
        const float t0 = tt[f][0], t1 = tt[f][1], t2 = tt[f][2];
        const float d1_0 = th[36 + f], d1_1 = th[40 + f], d1_2 = th[44 + f];
        const float d2_0 = th[48 + f], d2_1 = th[52 + f], d2_2 = th[56 + f];
        const float gu0 = odd ? gz[2] : gz[0], gu1 = gz[1], gu2 = odd ? gz[0] : gz[2];   // u = flip(z-update)
        // u_i = sum_{j>=i} R1[i][j] t_j
        float gt0 = d1_0 * gu0;
        float gt1 = th[(0 * 3 + 1) * 4 + f] * gu0 + d1_1 * gu1;
        float gt2 = th[(0 * 3 + 2) * 4 + f] * gu0 + th[(1 * 3 + 2) * 4 + f] * gu1 + d1_2 * gu2;
        gth.add(36 + f, gu0 * t0); gth.add(40 + f, gu1 * t1); gth.add(44 + f, gu2 * t2);
        gth.add((0 * 3 + 1) * 4 + f, gu0 * t1); gth.add((0 * 3 + 2) * 4 + f, gu0 * t2); gth.add((1 * 3 + 2) * 4 + f, gu1 * t2);
        // log-det: ld_i = log(|q_i| + 1e-8), q_i = (1 - t_i^2) d1_i d2_i + 1     (FLW:251-259)
        if (cE != 0.f && valid) {
            // q = 1 + (1 - t^2) d1 d2 >= t^2 >= 0: the diagonals are tanh outputs (|d| <= 1, MOD:341-348), so |q| = q and sign(q) = +1 - the abs / copysign
            // of FLW:255 are the identity here (the forward keeps them: libm parity)
            const float q0 = (1.f - t0 * t0) * (d1_0 * d2_0) + 1.f, q1 = (1.f - t1 * t1) * (d1_1 * d2_1) + 1.f,
                        q2 = (1.f - t2 * t2) * (d1_2 * d2_2) + 1.f;
            const float gq0 = cE * t_rcp(q0 + 1e-08f), gq1 = cE * t_rcp(q1 + 1e-08f), gq2 = cE * t_rcp(q2 + 1e-08f);
            gt0 += gq0 * (-2.f * t0 * d1_0 * d2_0); gt1 += gq1 * (-2.f * t1 * d1_1 * d2_1); gt2 += gq2 * (-2.f * t2 * d1_2 * d2_2);
            gth.add(36 + f, gq0 * (1.f - t0 * t0) * d2_0); gth.add(40 + f, gq1 * (1.f - t1 * t1) * d2_1); gth.add(44 + f, gq2 * (1.f - t2 * t2) * d2_2);
            gth.add(48 + f, gq0 * (1.f - t0 * t0) * d1_0); gth.add(52 + f, gq1 * (1.f - t1 * t1) * d1_1); gth.add(56 + f, gq2 * (1.f - t2 * t2) * d1_2);
        }
        const float gp0 = gt0 * (1.f - t0 * t0), gp1 = gt1 * (1.f - t1 * t1), gp2 = gt2 * (1.f - t2 * t2);
        gth.add(60 + f, gp0); gth.add(64 + f, gp1); gth.add(68 + f, gp2);                         // b
        // pre_i = sum_{j>=i} R2[i][j] zp_j,  R2[i][i] = d2_i,  R2[i][j>i] = D[j][i]
        gth.add(48 + f, gp0 * zp0); gth.add(52 + f, gp1 * zp1); gth.add(56 + f, gp2 * zp2);
        gth.add((1 * 3 + 0) * 4 + f, gp0 * zp1); gth.add((2 * 3 + 0) * 4 + f, gp0 * zp2); gth.add((2 * 3 + 1) * 4 + f, gp1 * zp2);
        const float gzp0 = d2_0 * gp0;
        const float gzp1 = th[(1 * 3 + 0) * 4 + f] * gp0 + d2_1 * gp1;
        const float gzp2 = th[(2 * 3 + 0) * 4 + f] * gp0 + th[(2 * 3 + 1) * 4 + f] * gp1 + d2_2 * gp2;
        gz[0] += odd ? gzp2 : gzp0; gz[1] += gzp1; gz[2] += odd ? gzp0 : gzp2;
        // alpha: a' = a + d1 tanh(d2 a + b)
        {
            const float d1 = th[72 + f], d2 = th[76 + f], tav = ta[f], ai = ain[f];
            float gta = ga * d1;
            gth.add(72 + f, ga * tav);
            if (cE != 0.f && valid) {
                const float q = (1.f - tav * tav) * (d1 * d2) + 1.f;          // >= 0, see above
                const float gq = cE * t_rcp(q + 1e-08f);
                gta += gq * (-2.f * tav * d1 * d2);
                gth.add(72 + f, gq * (1.f - tav * tav) * d2);
                gth.add(76 + f, gq * (1.f - tav * tav) * d1);
            }
            const float gpa = gta * (1.f - tav * tav);
            gth.add(80 + f, gpa);
            gth.add(76 + f, gpa * ai);
            ga += gpa * d2;
        }
    }
    // base sample z0 = eps * std + mean  (MOD:239,251)
    gms[0] += ga; gms[1] += ga * e[3];
    gms[2] += gz[0]; gms[3] += gz[1]; gms[4] += gz[2];
    gms[5] += gz[0] * e[0]; gms[6] += gz[1] * e[1]; gms[7] += gz[2] * e[2];
}

// Where the adjoint accumulates d loss / d theta of the current point over the latent samples: 84 registers per lane.
// (Round 3 tried LDS instead - one float per (entry, wave, lane), accumulated with ds_add_f32, to take 84 registers and the ~500
// v_mov / v_accvgpr moves they cause per (chunk, latent) out of the fused tail kernel: the vector-instruction count fell from 1 329 to
// 1 011, and the kernel took 333 us instead of 64 - LDS atomics run at a fraction of the plain LDS rate.  Not kept.)
struct GReg {
    float g[84];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 84; ++i) g[i] = 0.f;
    }
    __device__ __forceinline__ void add(int i, float v) {
#pragma clang fp contract(fast)      // the add must carry the flag too, or `g[i] += a * b` of flows_adjoint stays a v_mul + v_add pair (round 4: it did)
        g[i] += v;
    }
    __device__ __forceinline__ float get(int i) const { return g[i]; }
};
// One row of d loss / d theta (the pre-activation outputs of the flow-parameter heads), in the [P,128] layout of theta:
// the diagonals were tanh-ed (MOD:341-348), the padding columns are written as zeros (the weight-gradient GEMM reads them).
template <class GACC>
__device__ __forceinline__ void store_gtheta_row(float* __restrict__ row, const float (&th)[84], const GACC& gth) {
    // element i of the row: the diagonals (36..59, 72..79) chain through their tanh
    auto val = [&](int i) { const float g = gth.get(i); return ((i >= 36 && i < 60) || (i >= 72 && i < 80)) ? g * (1.f - th[i] * th[i]) : g; };
    f32x4* gp = reinterpret_cast<f32x4*>(row);
#pragma unroll
    for (int q = 0; q < 18; ++q) { f32x4 v; v[0] = val(q * 4); v[1] = val(q * 4 + 1); v[2] = val(q * 4 + 2); v[3] = val(q * 4 + 3); gp[q] = v; }
    f32x4 zero; zero[0] = zero[1] = zero[2] = zero[3] = 0.f;
#pragma unroll
    for (int q = 18; q < 24; ++q) gp[q] = zero;
#pragma unroll
    for (int q = 0; q < 3; ++q) { f32x4 v; v[0] = val(72 + q * 4); v[1] = val(73 + q * 4); v[2] = val(74 + q * 4); v[3] = val(75 + q * 4); gp[24 + q] = v; }
#pragma unroll
    for (int q = 27; q < 32; ++q) gp[q] = zero;
}

// (two workgroups per CU = two waves per SIMD: tail_parts (cfnerf_model.h) counts on it for K >= 64; the kernel needs 255 registers
// without the SLP vectoriser - tests/test_abi_cpu.py checks the built code object for spills)
__global__ __launch_bounds__(kThreads, 2)
void tail_bwd_kernel(const TailArgs A) {
#pragma clang fp contract(fast)
    __shared__ float carry[kWaves][kMaxK][3];              // per latent: (g, x, D) of the first sample of the chunk behind (comp_adjoint_D)
    // merge (ksplit 2 or 4, so a ray's parts are waves of ONE workgroup): the parts past the first publish their 84 partial sums per
    // sample here and the first adds them before it writes the row - one g_theta part leaves the kernel, and backward-data neither reads
    // a second [P,128] array nor writes the sum back (K = 64: +60 MB read, +65 MB written, +4.9 % of that kernel by counters).
    __shared__ float gsh[kWaves - 1][84][64];
    const int lane = lane_id_opaque(), wave = wave_id();
    GReg gth;
    // one wave per (ray, k-part): a ray's K latent samples are independent up to the sums over k, which are left to the
    // consumers (bwd_data adds the partial g_theta's while loading them, reduce_gms adds the rows), so small batches and
    // large K still fill the chip (this kernel runs one wave per SIMD: ~360 registers)
    const int64_t unit = (int64_t)blockIdx.x * kWaves + wave;
    const int64_t ray_u = unit / A.ksplit;
    const int part = (int)(unit - ray_u * A.ksplit);
    static_assert(kWaves % kTailParts == 0 && (kTailParts & (kTailParts - 1)) == 0, "a ray's k-parts must be waves of one workgroup");
    const bool merge = A.ksplit > 1;                       // (tail_parts: 1, 2 or 4 - the divisors of the workgroup's 4 waves)
    const bool live = ray_u < A.N;
    if (!live && !merge) return;                           // (merge: every wave of the workgroup keeps step with the barriers below)
    const int64_t ray = live ? ray_u : 0;                  // (a wave past the last ray addresses ray 0 and touches nothing)
    const int S = A.S, K = A.K;
    const int Kp = (K + A.ksplit - 1) / A.ksplit, k_lo = part * Kp, k_hi = min(K, k_lo + Kp);
    const float* rr = A.rays + ray * 11;
    const float dnorm = sqrtf((rr[3] * rr[3] + rr[4] * rr[4]) + rr[5] * rr[5]);
    const float cE = -((A.d_ent != nullptr) ? A.d_ent[0] : 0.f) / (float)((double)A.P * (double)K);
    const bool wb = (A.flags & CFNERF_F_WHITE_BKGD) != 0;
    const float a_mean = A.flat[0], a_std = A.flat[1];
    const float r_mean[3] = {A.flat[2], A.flat[3], A.flat[4]};
    const float r_std[3] = {A.flat[5], A.flat[6], A.flat[7]};
    for (int k = lane; k < K; k += 64) { carry[wave][k][0] = 0.f; carry[wave][k][1] = 0.f; carry[wave][k][2] = 0.f; }
    float gms[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gms[i] = 0.f;

    const int nch = (S + 63) / 64;
    for (int ch = nch - 1; ch >= 0; --ch) {
        if (!live) { __syncthreads(); __syncthreads(); continue; }      // (merge only: a wave past the last ray)
        const int s = ch * 64 + lane;
        const bool valid = s < S;
        const int64_t p = ray * (int64_t)S + (valid ? s : 0);
        // this chunk's tile of the transposed stash pieces (cfnerf_kernels.h): [k][64 rows][4] / [k][64 rows][2], lane = row
        const int tile_s = __builtin_amdgcn_readfirstlane((int)(ray * nch + ch));       // (scalar: the tile base stays in SGPRs, a lane keeps one 32-bit offset)
        const __amdgpu_buffer_rsrc_t raw_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.raw + (size_t)tile_s * K * 256), 0, K * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t at_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.at + (size_t)tile_s * K * 128), 0, K * 512, 0x00020000);
        float th[84];
        {
            const f32x4* tp = reinterpret_cast<const f32x4*>(A.theta + p * kThetaAll);
#pragma unroll
            for (int q = 0; q < 18; ++q) { const f32x4 v = tp[q]; th[q * 4] = v[0]; th[q * 4 + 1] = v[1]; th[q * 4 + 2] = v[2]; th[q * 4 + 3] = v[3]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) { const f32x4 v = tp[kThetaRgb / 4 + q]; th[72 + q * 4] = v[0]; th[73 + q * 4] = v[1]; th[74 + q * 4] = v[2]; th[75 + q * 4] = v[3]; }
        }
        gth.clear();
        const float zv = A.z[p];
        const float dz = (s >= S - 1) ? 1e1f : A.z[p + 1] - zv;
        const float dist = dz * dnorm;

        // this kernel runs ONE wave per SIMD (nothing else covers a load's latency): the inputs of latent k + 1 are fetched
        // while latent k is processed
        struct KIn { f32x4 rv; f32x2 at; f32x4 e; float G0, G1, G2, Gd; };
        auto fetch = [&](int k) {
            KIn q;
            // touch-once and coalesced (1 KB / 512 B per wave instruction), through ONE buffer descriptor per array: the tile base and the
            // latent are scalar offsets, a lane keeps lane * 16 / lane * 8 (no 64-bit vector addresses in a kernel that lives at 256 registers)
            q.rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raw_rs, lane * 16, k * 1024, /*nt*/ 2));
            q.at = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(at_rs, lane * 8, k * 512, /*nt*/ 2));
            q.e = *reinterpret_cast<const f32x4*>(A.eps + k * 4);
            q.G0 = A.d_rgb[ray * 3 * (int64_t)K + 0 * K + k]; q.G1 = A.d_rgb[ray * 3 * (int64_t)K + 1 * K + k];
            q.G2 = A.d_rgb[ray * 3 * (int64_t)K + 2 * K + k];
            q.Gd = (A.d_depth != nullptr) ? A.d_depth[ray * (int64_t)K + k] : 0.f;
            return q;
        };
        KIn nx = fetch(min(k_lo, K - 1));                  // (K not a multiple of the parts: a last part may be empty - its prefetch stays in range)
        for (int k = k_lo; k < k_hi; ++k) {
            const KIn cur = nx;
            if (k + 1 < k_hi) nx = fetch(k + 1);
            const f32x4 rv = cur.rv;
            // the forward stashed e = exp(-softplus(a) dist) and T: alpha = 1 - e is the forward's own number again (one exact subtraction),
            // and d alpha / d softplus = dist * e comes from the exponential ITSELF like torch's exp backward.  Rounds 1-4 stashed alpha and
            // multiplied by (1 - alpha): for an opaque sample - above all a ray's LAST one, whose 1e1 interval (RUN:427) makes it the largest
            // entry of the density gradient - that subtraction keeps e only to eps / e (alpha = 0.99925: 8e-5), and a single ray's
            // density-path gradient sat 10-30 x further from fp64 than torch's own fp32 autograd (tests/tools/k2_grad_diag.py, round 5)
            const float ea = cur.at[0], Tt = cur.at[1];
            const float alpha = 1.f - ea;
            const float G0 = cur.G0, G1 = cur.G1, G2 = cur.G2, Gd = cur.Gd;
            const float c0 = t_sigmoid(rv[0]), c1 = t_sigmoid(rv[1]), c2 = t_sigmoid(rv[2]);
            const float w = alpha * Tt;
            float g = (G0 * c0 + G1 * c1 + G2 * c2) + Gd * zv;                 // d loss / d w_s
            if (wb) g -= (G0 + G1 + G2);                                       // rgb_map += 1 - acc  (RUN:452)
            // d loss / d alpha = T D with D = g_s - (what the samples behind s render for g), carried by its own recurrence (comp_adjoint_D,
            // cfnerf_device.h: rounds 1-5 formed g T - suffix / x from the forward's T and lost 1-2 digits per ray to the product scan's noise)
            const float xk = (1.f - alpha) + 1e-10f;                           // cumprod factor of RUN:443
            float cg = carry[wave][k][0], cx = carry[wave][k][1], cD = carry[wave][k][2];
            const float Dv = comp_adjoint_D(valid ? g : 0.f, xk, cg, cx, cD);
            if (lane == 0) { carry[wave][k][0] = cg; carry[wave][k][1] = cx; carry[wave][k][2] = cD; }
            const float dalpha = Tt * Dv;
            const float sg = t_sigmoid(rv[3]);                                 // softplus'
            float ga = dalpha * ea * dist * sg + cE * (1.f - sg);              // + d(-mean(a - softplus a))  MOD:263
            float gz[3] = {G0 * w * c0 * (1.f - c0) + cE * (1.f - 2.f * c0),   // + d(-mean(c - 2 softplus c)) MOD:278
                           G1 * w * c1 * (1.f - c1) + cE * (1.f - 2.f * c1),
                           G2 * w * c2 * (1.f - c2) + cE * (1.f - 2.f * c2)};
            if (!valid) { ga = 0.f; gz[0] = gz[1] = gz[2] = 0.f; }

            flows_adjoint(th, gth, gms, cur.e, a_mean, a_std, r_mean, r_std, ga, gz, cE, valid);
        }
        if (merge) {
            if (part != 0) {
#pragma unroll
                for (int i = 0; i < 84; ++i) gsh[wave - 1][i][lane] = gth.get(i);
            }
            __syncthreads();
            if (part == 0) {
                for (int pp = 1; pp < A.ksplit; ++pp) {    // fixed order: part 0 + part 1 (+ part 2 + part 3)
#pragma unroll
                    for (int i = 0; i < 84; ++i) gth.add(i, gsh[wave + pp - 1][i][lane]);
                }
            }
            __syncthreads();                               // (the rows are free again)
            if (part == 0 && valid) store_gtheta_row(A.g_theta + p * kThetaAll, th, gth);
        } else if (valid) {
            store_gtheta_row(A.g_theta + p * kThetaAll, th, gth);
        }
    }
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) gms[i] = wave_sum(gms[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) A.gms_partials[unit * 8 + i] = gms[i];
    }
}

// ------------------------------------------------------------------------------------------------
// 2b. The UNFUSED seam (the reference's NeRF_Flows.forward and raw2outputs are separately differentiable, so a caller-supplied
//     network_query_fn trains): the tail kernel's two halves as standalone kernels with the same arithmetic.
//
// flows_bwd_kernel: adjoint of the K flows + the entropy terms of NeRF_Flows.forward (MOD:225-291) given d loss / d raw [P,K,4].
// lane = point; every latent of a point in one lane, so g_theta needs no partial sums.
__global__ __launch_bounds__(kThreads)
void flows_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ theta, const float* __restrict__ eps, const float* __restrict__ flat,
                      const float* __restrict__ d_raw, const float* __restrict__ d_ent, int64_t P, int K, float* __restrict__ g_theta,
                      float* __restrict__ gms_partials) {
#pragma clang fp contract(fast)
    const int lane = lane_id_opaque(), wave = wave_id();
    const int64_t pi = (int64_t)blockIdx.x * kThreads + wave * 64 + lane;
    const bool valid = pi < P;
    const int64_t p = valid ? pi : 0;
    const float cE = -((d_ent != nullptr) ? d_ent[0] : 0.f) / (float)((double)P * (double)K);
    const float a_mean = flat[0], a_std = flat[1];
    const float r_mean[3] = {flat[2], flat[3], flat[4]};
    const float r_std[3] = {flat[5], flat[6], flat[7]};
    float th[84], gms[8];
    GReg gth;
    {
        const f32x4* tp = reinterpret_cast<const f32x4*>(theta + p * kThetaAll);
#pragma unroll
        for (int q = 0; q < 18; ++q) { const f32x4 v = tp[q]; th[q * 4] = v[0]; th[q * 4 + 1] = v[1]; th[q * 4 + 2] = v[2]; th[q * 4 + 3] = v[3]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) { const f32x4 v = tp[kThetaRgb / 4 + q]; th[72 + q * 4] = v[0]; th[73 + q * 4] = v[1]; th[74 + q * 4] = v[2]; th[75 + q * 4] = v[3]; }
    }
    gth.clear();
#pragma unroll
    for (int i = 0; i < 8; ++i) gms[i] = 0.f;
    for (int k = 0; k < K; ++k) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(raw + (((p >> 6) * K + k) * 64 + (p & 63)) * 4);    // tile-transposed stash: [tile p / 64][k][row p % 64][4]
        f32x4 g; g[0] = g[1] = g[2] = g[3] = 0.f;
        if (d_raw != nullptr) g = *reinterpret_cast<const f32x4*>(d_raw + (p * K + k) * 4);
        const f32x4 e = *reinterpret_cast<const f32x4*>(eps + k * 4);
        const float c0 = t_sigmoid(rv[0]), c1 = t_sigmoid(rv[1]), c2 = t_sigmoid(rv[2]), sg = t_sigmoid(rv[3]);
        float ga = g[3] + cE * (1.f - sg);                                     // + d(-mean(a - softplus a))  MOD:263
        float gz[3] = {g[0] + cE * (1.f - 2.f * c0), g[1] + cE * (1.f - 2.f * c1), g[2] + cE * (1.f - 2.f * c2)};   // MOD:278
        if (!valid) { ga = 0.f; gz[0] = gz[1] = gz[2] = 0.f; }
        flows_adjoint(th, gth, gms, e, a_mean, a_std, r_mean, r_std, ga, gz, cE, valid);
    }
    if (valid) store_gtheta_row(g_theta + p * kThetaAll, th, gth);
#pragma unroll
    for (int i = 0; i < 8; ++i) gms[i] = wave_sum(gms[i]);
    if (lane == 0) {
        const int64_t row = (int64_t)blockIdx.x * kWaves + wave;
#pragma unroll
        for (int i = 0; i < 8; ++i) gms_partials[row * 8 + i] = gms[i];
    }
}


// ---- host launchers (cfnerf_bwd.h)
hipError_t launch_tail_bwd(const TailArgs& ta, int64_t n_rays, int ksplit, hipStream_t st) {
    hipLaunchKernelGGL(tail_bwd_kernel, dim3((unsigned)((n_rays * ksplit + kWaves - 1) / kWaves)), dim3(kThreads), 0, st, ta);
    return hipGetLastError();
}

hipError_t launch_flows_bwd(const float* raw, const float* theta, const float* eps, const float* flat, const float* d_raw, const float* d_ent,
                            int64_t P, int K, float* g_theta, float* gms_partials, unsigned* grid_out, hipStream_t st) {
    const unsigned grid = (unsigned)((P + kThreads - 1) / kThreads);
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL(flows_bwd_kernel, dim3(grid), dim3(kThreads), 0, st, raw, theta, eps, flat, d_raw, d_ent, P, K, g_theta, gms_partials);
    return hipGetLastError();
}

}  // namespace cfnerf
