// Development microbenchmark (round 5): issue cost of the vector instructions the flow + composite phase is made of, one wave per SIMD and
// two waves per SIMD: plain fp32, packed fp32 (v_pk_mul / v_pk_add / v_pk_fma), the hardware transcendentals, DPP adds, and mixes
// (a transcendental followed by independent plain instructions: do they overlap?).  Straight-line blocks of 64 instructions on 8
// independent register chains, s_memtime around 2000 repetitions.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)

enum Kind { MUL, FMA, PK_MUL, PK_ADD, PK_FMA, PK_MUL_OPSEL, EXP, RCP, LOG, DPP_ADD, EXP_THEN_3MUL, EXP_THEN_7MUL, EXP_PAIR_THEN_PK, N_KIND };
static const char* kNames[] = {"v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32 op_sel_hi:[1,0]", "v_exp_f32",
                               "v_rcp_f32", "v_log_f32", "v_add_f32 row_shr:1 (DPP)", "1 v_exp + 3 v_mul (per 4)", "1 v_exp + 7 v_mul (per 8)",
                               "2 v_exp + 2 v_pk_mul (per 4)"};

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, long long* clk, int n) {
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f, b2 = 1.0002f, b3 = 0.9998f, b4 = 1.0003f, b5 = 0.9997f, b6 = 1.0004f, b7 = 0.9996f;
    const float c = 1.00001f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == MUL)
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (KIND == FMA)
                asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (KIND == EXP)
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (KIND == RCP)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (KIND == LOG)
                asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (KIND == DPP_ADD)
                asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (KIND == EXP_THEN_3MUL)
                asm volatile("v_exp_f32 %0, %0\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_exp_f32 %4, %4\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (KIND == EXP_THEN_7MUL)
                asm volatile("v_exp_f32 %0, %0\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
                if (KIND == PK_MUL)
                    asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
                else if (KIND == PK_ADD)
                    asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
                else if (KIND == PK_FMA)
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
                else if (KIND == PK_MUL_OPSEL)
                    asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
                                 "v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
                else   // EXP_PAIR_THEN_PK: the shape of Num2::exp - two transcendentals, then packed work on other registers
                    asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_pk_mul_f32 %4, %4, %6\n v_pk_mul_f32 %5, %5, %6\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_pk_mul_f32 %4, %4, %6\n v_pk_mul_f32 %5, %5, %6"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p2), "+v"(p3) : "v"(q));
                a0 = p0[0]; a1 = p0[1]; a2 = p1[0]; a3 = p1[1]; a4 = p2[0]; a5 = p2[1]; a6 = p3[0]; a7 = p3[1];
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int KIND>
static void run(float* out, long long* clk) {
    const int grid = 256, n = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd;
        long long h[256 * 8];
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((probe<KIND>), dim3(grid), dim3(threads), 0, 0, out, clk, 10);      // warm
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND>), dim3(grid), dim3(threads), 0, 0, out, clk, n);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        double m = 0;
        const int nw = threads / 64;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < nw; ++w) m += (double)h[b * 8 + w];
        m /= grid * nw;
        // s_memtime ticks at 100 MHz on this chip; the event time gives wall time per instruction of ONE wave
        printf("%-34s %d wave(s)/SIMD: %7.2f ns per instruction and wave (events)  = %5.2f cycles at 2.4 GHz;  counter ticks per instruction %.4f\n", kNames[KIND],
               waves_per_simd, ms * 1e6 / (64.0 * n), ms * 1e6 / (64.0 * n) * 2.4, m / (64.0 * n));
    }
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8 * 8);
    run<MUL>(out, clk); run<FMA>(out, clk); run<PK_MUL>(out, clk); run<PK_ADD>(out, clk); run<PK_FMA>(out, clk); run<PK_MUL_OPSEL>(out, clk);
    run<EXP>(out, clk); run<RCP>(out, clk); run<LOG>(out, clk); run<DPP_ADD>(out, clk); run<EXP_THEN_3MUL>(out, clk); run<EXP_THEN_7MUL>(out, clk);
    run<EXP_PAIR_THEN_PK>(out, clk);
    return 0;
}
