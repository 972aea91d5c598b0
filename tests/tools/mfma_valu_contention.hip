// Development microbenchmark: how fast does a VALU-only wave run while its SIMD neighbour issues MFMAs back to back?
// Each workgroup = 8 waves (2 per SIMD): waves 0-3 run an MFMA loop (32x32x2 f32 or 16x16x4 f32, 4 independent
// accumulators), waves 4-7 a dependent-free VALU loop.  Prints the time of the VALU waves alone, of the MFMA waves
// alone, and of both together.   hipcc --offload-arch=gfx950 -O3 mfma_valu_contention.hip -o /tmp/mvc && /tmp/mvc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>   // 0: 32x32x2 f32, 1: 16x16x4 f32, 2: 32x32x16 bf16 (the split-bf16 mode's instruction)
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int mode, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6;
    const long long t0 = wall_clock64();
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            if (SHAPE == 0) {
                f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
                const float x = threadIdx.x * 1e-3f, y = 1.0001f;
                for (int i = 0; i < n_mfma; ++i) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
                }
                r = a0[0] + a1[1] + a2[2] + a3[3];
            } else if (SHAPE == 2) {
                f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
                bf16x8 x, y;
                for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(1.0f + 0.01f * e); }
                for (int i = 0; i < n_mfma; ++i) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
                }
                r = a0[0] + a1[1] + a2[2] + a3[3];
            } else {
                f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0}, a4 = {0}, a5 = {0}, a6 = {0}, a7 = {0};
                const float x = threadIdx.x * 1e-3f, y = 1.0001f;
                for (int i = 0; i < n_mfma; ++i) {      // same FLOPs per iteration: 8 x (16x16x4) = 4 x (32x32x2)
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
                    a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a5, 0, 0, 0);
                    a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a6, 0, 0, 0);
                    a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a7, 0, 0, 0);
                }
                r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + a6[2] + a7[3];
            }
        }
    } else if (mode & 8) {
        __shared__ float lds[4 * 64 * 8];
        float* p = lds + (wave - 4) * 512 + (threadIdx.x & 63);
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_nontemporal_store((float)i, p + 64 * u);   // 8 ds_write_b32
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        r = p[0];
    } else if (mode & 16) {
        float* g = out + (size_t)(blockIdx.x * 512 + threadIdx.x);
        for (int i = 0; i < n_valu / 4; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_nontemporal_store((float)i, g + (size_t)u * 256 * 512);   // 8 global_store_dword
        }
        r = 1.f;
    } else if (mode & 2) {
        if (mode & 4) __builtin_amdgcn_s_setprio(3);      // VALU waves at raised priority
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
        for (int i = 0; i < n_valu; ++i) {             // 8 independent-ish VALU ops per iteration
            v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.25f); v2 = fmaf(v2, 1.0002f, 0.125f); v3 = fmaf(v3, 0.9998f, 1.f);
            v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.25f); v2 = fmaf(v2, 1.0002f, 0.125f); v3 = fmaf(v3, 0.9998f, 1.f);
        }
        r = v0 + v1 + v2 + v3;
    }
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int SHAPE>
static void run(const char* name, float* out, long long* clk) {
    const int grid = 256, n_mfma = 20000, n_valu = 40000;
    long long h[256 * 8];
    for (int mode : {1, 2, 3, 7, 8, 9, 16, 17}) {
        hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(512), 0, 0, out, clk, mode, n_mfma, n_valu);
        hipDeviceSynchronize();
        hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w];
        m /= grid * 4 * 100.0; v /= grid * 4 * 100.0;     // us (100 MHz clock)
        printf("%-10s mode %d (%s): MFMA waves %8.1f us   VALU waves %8.1f us\n", name, mode,
               mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : mode == 3 ? "MFMA + VALU" : mode == 7 ? "MFMA + VALU prio 3" : mode == 8 ? "LDS writes only" : mode == 9 ? "MFMA + LDS writes" : mode == 16 ? "global stores only" : "MFMA + global stores", m, v);
    }
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, (size_t)256 * 512 * 4 * 9); hipMalloc(&clk, 256 * 8 * 8);
    run<0>("32x32x2", out, clk);
    run<1>("16x16x4", out, clk);
    run<2>("32x32x16bf", out, clk);
    return 0;
}
