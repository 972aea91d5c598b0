// Development microbenchmark: VALU / LDS instructions interleaved INSIDE one wave's MFMA stream (one wave per SIMD):
// how many fit in the shadow of a v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles) before the loop slows down?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NL, int BF = 0>      // NV independent VALU fmas and NL ds_write_b32 per MFMA; BF: v_mfma_f32_32x32x16_bf16 (8 passes)
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int n) {
    __shared__ float lds[256 * 4];
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    bf16x8 xb, yb;
    for (int e = 0; e < 8; ++e) { xb[e] = (__bf16)(threadIdx.x * 1e-3f + e); yb[e] = (__bf16)(1.0f + 0.01f * e); }
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float* lp = lds + threadIdx.x;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
auto step = [&](f32x16& acc) {
            if (BF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NV; ++u) v[u & 7] = fmaf(v[u & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int u = 0; u < NL; ++u) lp[256 * (u & 3)] = v[u & 7];
        };
        step(a0); step(a1); step(a2); step(a3);
    }
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float r = a0[0] + a1[1] + a2[2] + a3[3] + lds[threadIdx.x];
    for (int u = 0; u < 8; ++u) r += v[u];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NV, int NL, int BF = 0>
static void run(float* out, long long* clk) {
    const int grid = 256, n = 20000;
    long long h[256 * 4];
    hipLaunchKernelGGL((k<NV, NL, BF>), dim3(grid), dim3(256), 0, 0, out, clk, n);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid * 4; ++i) m += h[i];
    m /= grid * 4 * 100.0;
    printf("%s per MFMA: %2d VALU + %2d ds_write : %8.1f us  (%.1f cycles per MFMA at 2.2 GHz)\n", BF ? "bf16 32x32x16" : "f32 32x32x2  ", NV, NL, m, m * 2200.0 / (4.0 * n));
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 4 * 8);
    run<0, 0>(out, clk); run<4, 0>(out, clk); run<8, 0>(out, clk); run<12, 0>(out, clk); run<16, 0>(out, clk); run<24, 0>(out, clk);
    run<0, 2>(out, clk); run<0, 4>(out, clk); run<4, 4>(out, clk); run<8, 4>(out, clk); run<8, 8>(out, clk);
    run<0, 0, 1>(out, clk); run<2, 0, 1>(out, clk); run<4, 0, 1>(out, clk); run<6, 0, 1>(out, clk); run<8, 0, 1>(out, clk); run<12, 0, 1>(out, clk);
    run<0, 2, 1>(out, clk); run<0, 4, 1>(out, clk); run<4, 2, 1>(out, clk); run<4, 4, 1>(out, clk);
    return 0;
}
