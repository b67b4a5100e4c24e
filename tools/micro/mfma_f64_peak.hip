// What the fp64 matrix pipe sustains with operands in registers only: every wave issues `iters` x 14 independent
// v_mfma_f64_16x16x4_f64 (14 accumulators, as the GEMM kernels hold), on 8 / 64 / 256 workgroup slots x 1, 2, 4 waves per
// SIMD.  Prints TFLOP/s, the shader clock (s_memtime against the 100 MHz s_memrealtime) and cycles per instruction and
// SIMD - the ceiling the LDS-fed loops of ssmq_gemm_mfma.hip / ssmq_bq_fused.hip are measured against.  With -DAGPR_FORM the
// launch bounds leave room for 512 registers and hipcc puts the accumulators into AGPRs: 103-140 cycles per instruction
// instead of 64 (measured on MI355X, ROCm 7.2).  tools/micro/run_mfma_micro.sh builds and runs both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#ifdef AGPR_FORM
#define MFMA_BOUNDS __launch_bounds__(256)        // room for 512 registers: hipcc keeps the accumulators in AGPRs
#else
#define MFMA_BOUNDS __launch_bounds__(256, 2)     // 256 registers: accumulators in architectural VGPRs
#endif
__global__ MFMA_BOUNDS void k(double *out, int iters, double a0, double b0, unsigned long long *clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    v4d acc[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
    double a[14], b = b0;
#pragma unroll
    for (int i = 0; i < 14; ++i) a[i] = a0 + (threadIdx.x + 64 * i) * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            asm volatile("" : "+v"(a[i]));           // a different A operand per instruction, as in the GEMM kernels
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b, acc[i], 0, 0, 0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 14; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // shader-clock cycles and 100 MHz ticks of this wave
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
int main() {
    double *d;
    unsigned long long *clk, hclk[2];
    hipMalloc(&d, 4096);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    // cus: how much of the chip is busy (a power / current limit would show as a rate that depends on it)
    for (int cus : {8, 64, 256}) {
        for (int wps = 1; wps <= 4; wps *= 2) {        // waves per SIMD: blocks of 256 threads (4 waves = one per SIMD)
            const int blocks = cus * wps;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0, 1.0, clk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0, clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 14 * 2048.0;
            hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
            const double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
            printf("%3d workgroup slots x %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s (chip nominal 78.6 at 2.4 GHz); shader clock %.2f GHz -> "
                   "%.1f cycles per MFMA and SIMD\n", cus, wps, ms, flop / (ms * 1e-3) / 1e12, ghz,
                   ms * 1e-3 * ghz * 1e9 / ((double)wps * iters * 14));
        }
    }
    return 0;
}
