// What a partially filled wave costs on gfx950: a wave issues 8 independent chains of fp64 FMAs (no memory) with the first
// `active` lanes enabled.  If the vector unit skipped 16-lane quarters whose EXEC bits are all zero, 48 / 32 / 16 active lanes
// would take 3/4, 1/2, 1/4 of the time of 64.  One wave per SIMD (256 workgroups x 256 threads; the other three waves of
// each workgroup... every wave runs the same), wall clock over HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_fma(double *out, int active, int iters) {
    const int lane = threadIdx.x & 63;
    double a0 = 1.0 + lane, a1 = 2.0, a2 = 3.0, a3 = 4.0, a4 = 5.0, a5 = 6.0, a6 = 7.0, a7 = 8.0;
    const double m = 1.0000001, c = 1e-9;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 256);
    const int iters = 200000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int active : {64, 64, 49, 48, 33, 32, 17, 16, 1}) {
        hipLaunchKernelGGL(k_fma, dim3(256), dim3(256), 0, 0, d, active, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(256), dim3(256), 0, 0, d, active, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("active lanes %2d: %.3f ms, %.2f ns per wave instruction\n", active, ms, 1e6 * ms / (8.0 * iters));
    }
    return 0;
}
