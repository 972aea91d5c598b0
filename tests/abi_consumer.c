/*
 * abi_consumer.c - a plain C99 host program over include/cfnerf.h: no Python, no torch, no C++ in the process.
 *
 * It is what a maintainer's non-Python binding of the drop-in boundary amounts to: device buffers from
 * hipMalloc, every step of the reference's train iteration through the C ABI -
 *     cfnerf_rays_setup -> cfnerf_render_fwd (STASH) -> cfnerf_loss_fwd_bwd -> cfnerf_render_bwd -> cfnerf_adam_step
 * (render RUN:129-158, render_rays RUN:457-553, loss RUN:1026-1050, loss.backward() RUN:1066, optimizer.step() RUN:1067) -
 * and the results compared with the numbers the REAL reference produced for the same inputs, which the test
 * (tests/test_hip_abi_consumer.py) writes into a flat binary case file from golden fixture G5/G7.
 *
 *   abi_consumer <case file>        exit code 0 = every comparison inside its tolerance
 *
 * case file (little-endian):  "CFNB" | int32 version = 1 | cfnerf_cfg (7 x int32) | int32 N, S, K, H, W, ndc, flags |
 *   float focal, near, far, beta1, lr | int64 n_params | float arrays: flat[n_params] rays_o[N,3] rays_d[N,3] t_vals[S]
 *   t_rand[N,S] eps[K,4] target[N,3] | expected: rgb_map[N,3,K] disp[N,K] depth[N,K] scalars[4] grad[n_params] adam1[n_params]
 *   (NaN in grad / adam1 = the fixture does not hold that entry)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "cfnerf.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define CFOK(x) do { int r_ = (x); if (r_ != CFNERF_OK) { fprintf(stderr, "%s:%d: status %d: %s\n", __FILE__, __LINE__, r_, cfnerf_last_error()); return 3; } } while (0)

static float* read_floats(FILE* f, size_t n) {
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    if (!p || fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "short case file\n"); exit(4); }
    return p;
}

static float* to_device(const float* h, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, (n ? n : 1) * sizeof(float)) != hipSuccess) return NULL;
    if (h && hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

/* max over i of |a - b| / (atol + rtol |b|); entries with NaN in b are skipped; *count = compared entries */
static double worst_ratio(const float* a, const float* b, size_t n, double atol, double rtol, size_t* count) {
    double w = 0.0;
    size_t c = 0, i;
    for (i = 0; i < n; ++i) {
        double r;
        if (isnan(b[i])) continue;
        r = fabs((double)a[i] - (double)b[i]) / (atol + rtol * fabs((double)b[i]));
        if (!(r <= w)) w = r;                  /* (a NaN in a propagates) */
        ++c;
    }
    if (count) *count = c;
    return w;
}

int main(int argc, char** argv) {
    FILE* f;
    char magic[4];
    int32_t version, dims[7];
    float sc[5];
    int64_t n_params;
    cfnerf_cfg cfg;
    cfnerf_model* m = NULL;
    int fails = 0;
    if (argc != 2) { fprintf(stderr, "usage: %s <case file>\n", argv[0]); return 64; }
    f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 64; }
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "CFNB", 4) != 0 || fread(&version, 4, 1, f) != 1 || version != 1 ||
        fread(&cfg, sizeof cfg, 1, f) != 1 || fread(dims, 4, 7, f) != 7 || fread(sc, 4, 5, f) != 5 || fread(&n_params, 8, 1, f) != 1) {
        fprintf(stderr, "bad case file header\n");
        return 64;
    }
    {
        const int N = dims[0], S = dims[1], K = dims[2], H = dims[3], W = dims[4], ndc = dims[5], flags = dims[6];
        const float focal = sc[0], near_ = sc[1], far_ = sc[2], beta1 = sc[3], lr = sc[4];
        const size_t np = (size_t)n_params;
        float *h_flat = read_floats(f, np), *h_ro = read_floats(f, (size_t)N * 3), *h_rd = read_floats(f, (size_t)N * 3);
        float *h_tv = read_floats(f, (size_t)S), *h_tr = read_floats(f, (size_t)N * S), *h_eps = read_floats(f, (size_t)K * 4);
        float *h_tg = read_floats(f, (size_t)N * 3);
        float *x_rgb = read_floats(f, (size_t)N * 3 * K), *x_disp = read_floats(f, (size_t)N * K), *x_depth = read_floats(f, (size_t)N * K);
        float *x_sc = read_floats(f, 4), *x_grad = read_floats(f, np), *x_adam = read_floats(f, np);
        float *d_flat, *d_ro, *d_rd, *d_tv, *d_tr, *d_eps, *d_tg, *d_rays, *d_rgb, *d_disp, *d_depth, *d_ent, *d_drgb, *d_sc, *d_grad, *d_m, *d_v, *d_beta;
        float *o_rgb = (float*)malloc((size_t)N * 3 * K * 4), *o_disp = (float*)malloc((size_t)N * K * 4), *o_depth = (float*)malloc((size_t)N * K * 4);
        float *o_grad = (float*)malloc(np * 4), *o_flat = (float*)malloc(np * 4), o_sc[4];
        uint64_t gen;
        double r;
        size_t cnt, i;
        fclose(f);

        /* the layout queries and the argument checks work without a model and without a device */
        if (cfnerf_param_count(&cfg) != n_params) { fprintf(stderr, "param count %lld != %lld\n", (long long)cfnerf_param_count(&cfg), (long long)n_params); return 5; }
        if (cfnerf_render_fwd(NULL, NULL, NULL, NULL, NULL, NULL, 1, 1, 1, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != CFNERF_E_INVALID ||
            strlen(cfnerf_last_error()) == 0) { fprintf(stderr, "a NULL model must be refused with a message\n"); return 5; }

        HIPOK(hipSetDevice(0));
        d_flat = to_device(h_flat, np); d_ro = to_device(h_ro, (size_t)N * 3); d_rd = to_device(h_rd, (size_t)N * 3);
        d_tv = to_device(h_tv, (size_t)S); d_tr = to_device(h_tr, (size_t)N * S); d_eps = to_device(h_eps, (size_t)K * 4);
        d_tg = to_device(h_tg, (size_t)N * 3); d_rays = to_device(NULL, (size_t)N * 11);
        d_rgb = to_device(NULL, (size_t)N * 3 * K); d_disp = to_device(NULL, (size_t)N * K); d_depth = to_device(NULL, (size_t)N * K);
        d_ent = to_device(NULL, 1); d_drgb = to_device(NULL, (size_t)N * 3 * K); d_sc = to_device(NULL, 4);
        d_grad = to_device(NULL, np); d_m = to_device(NULL, np); d_v = to_device(NULL, np); d_beta = to_device(&beta1, 1);
        if (!d_flat || !d_ro || !d_rd || !d_tv || !d_tr || !d_eps || !d_tg || !d_rays || !d_rgb || !d_disp || !d_depth || !d_ent || !d_drgb ||
            !d_sc || !d_grad || !d_m || !d_v || !d_beta) { fprintf(stderr, "hipMalloc / hipMemcpy failed\n"); return 2; }
        HIPOK(hipMemset(d_m, 0, np * 4));
        HIPOK(hipMemset(d_v, 0, np * 4));

        CFOK(cfnerf_model_create(&cfg, &m));
        CFOK(cfnerf_model_set_params(m, d_flat, NULL));
        CFOK(cfnerf_rays_setup(H, W, focal, NULL, d_ro, d_rd, N, 0, ndc, near_, far_, d_rays, NULL));
        CFOK(cfnerf_render_fwd(m, d_rays, d_tv, d_tr, NULL, d_eps, N, S, K, flags | CFNERF_F_STASH, d_rgb, d_disp, d_depth, NULL, NULL, NULL,
                               NULL, d_ent, NULL));
        gen = cfnerf_model_stash_generation(m);
        CFOK(cfnerf_loss_fwd_bwd(d_rgb, d_tg, d_ent, N, K, beta1, N, d_drgb, d_sc, NULL));
        CFOK(cfnerf_render_bwd(m, gen, d_drgb, NULL, d_beta, d_grad, NULL));
        HIPOK(hipDeviceSynchronize());
        HIPOK(hipMemcpy(o_rgb, d_rgb, (size_t)N * 3 * K * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(o_disp, d_disp, (size_t)N * K * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(o_depth, d_depth, (size_t)N * K * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(o_sc, d_sc, 16, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(o_grad, d_grad, np * 4, hipMemcpyDeviceToHost));
        CFOK(cfnerf_adam_step(m, d_flat, d_grad, d_m, d_v, 1, lr, 1.0f, NULL));
        HIPOK(hipDeviceSynchronize());
        HIPOK(hipMemcpy(o_flat, d_flat, np * 4, hipMemcpyDeviceToHost));

        /* forward outputs and loss lines: the tolerances of tests/util_hip.py */
        r = worst_ratio(o_rgb, x_rgb, (size_t)N * 3 * K, 1e-5, 1e-4, &cnt);   printf("rgb_map   %zu entries, worst err / tol %.3f\n", cnt, r); fails += !(r <= 1.0);
        r = worst_ratio(o_depth, x_depth, (size_t)N * K, 1e-5, 1e-4, &cnt);   printf("depth_map %zu entries, worst err / tol %.3f\n", cnt, r); fails += !(r <= 1.0);
        r = worst_ratio(o_disp, x_disp, (size_t)N * K, 1e-4, 1e-3, &cnt);     printf("disp_map  %zu entries, worst err / tol %.3f\n", cnt, r); fails += !(r <= 1.0);
        r = worst_ratio(o_sc, x_sc, 3, 1e-5, 1e-4, &cnt);                      printf("loss, nll, mse: worst err / tol %.3f   (loss %.6f, reference %.6f)\n", r, o_sc[0], x_sc[0]); fails += !(r <= 1.0);
        r = worst_ratio(o_sc + 3, x_sc + 3, 1, 1e-4, 1e-4, &cnt);              printf("psnr: err / tol %.3f\n", r); fails += !(r <= 1.0);
        /* gradient against the reference's autograd: relative L2 error over the entries the fixture holds (a ReLU unit that
           rounds to the other side of 0 in one of the two fp32 forwards moves single entries - the -m gpu gradient tests
           correct for those masks; here the norm bounds it) */
        {
            double num = 0.0, den = 0.0;
            cnt = 0;
            for (i = 0; i < np; ++i) {
                if (isnan(x_grad[i])) continue;
                num += ((double)o_grad[i] - x_grad[i]) * ((double)o_grad[i] - x_grad[i]);
                den += (double)x_grad[i] * x_grad[i];
                ++cnt;
            }
            r = sqrt(num / (den > 0 ? den : 1));
            printf("gradient  %zu entries, relative L2 error %.3e\n", cnt, r);
            fails += !(cnt > 0 && r <= 2e-3);
        }
        /* Adam, first step: every parameter moves by at most lr (m / sqrt(v) = +-1); against the reference's updated parameters
           all but the entries whose ~0 gradient has the other sign agree to rounding */
        {
            size_t far_off = 0;
            double max_move = 0.0;
            cnt = 0;
            for (i = 0; i < np; ++i) {
                const double mv = fabs((double)o_flat[i] - h_flat[i]);
                if (!(mv <= max_move)) max_move = mv;
                if (isnan(x_adam[i])) continue;
                ++cnt;
                if (fabs((double)o_flat[i] - x_adam[i]) > 1e-6 + 1e-6 * fabs((double)x_adam[i])) ++far_off;
            }
            printf("adam step %zu entries, %zu differ from the reference's by more than rounding, largest move %.3e (lr %.1e)\n", cnt, far_off, max_move, lr);
            fails += !(cnt > 0 && far_off <= cnt / 50 && max_move <= lr * 1.0001 + 1e-7);
        }
        /* a backward against a stash that a later forward replaced is refused */
        CFOK(cfnerf_render_fwd(m, d_rays, d_tv, d_tr, NULL, d_eps, N, S, K, flags | CFNERF_F_STASH, d_rgb, d_disp, d_depth, NULL, NULL, NULL,
                               NULL, d_ent, NULL));
        if (cfnerf_render_bwd(m, gen, d_drgb, NULL, d_beta, d_grad, NULL) != CFNERF_E_INVALID) { printf("a stale stash generation was accepted\n"); ++fails; }
        HIPOK(hipDeviceSynchronize());
        CFOK(cfnerf_model_destroy(m));
        printf(fails ? "abi_consumer: %d FAILED\n" : "abi_consumer: OK\n", fails);
        return fails ? 1 : 0;
    }
}
