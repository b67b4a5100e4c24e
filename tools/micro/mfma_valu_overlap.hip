// Do fp64 matrix instructions and fp64 vector instructions of the two waves of a SIMD overlap?  Every wave runs `steps`
// steps of [28 independent v_mfma_f64_16x16x4_f64 | NV dependent-chain-free v_fma_f64]; mode 0: matrix block only, 1: vector
// block only, 2: both, same order in every wave, 3: both, odd waves vector block first.  A workgroup barrier per step
// (as the GEMM loops have) if `bar`.  256 workgroups x 512 threads = two waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NV = 64;      // x 8 chains = 512 fma per step
__global__ __launch_bounds__(512, 1) void k(double *out, int steps, int mode, int bar, double a0) {
    v4d acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
    double a[7], b = a0, c[8];
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i] = a0 + (threadIdx.x + 64 * i) * 1e-9;
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = a0 * i;
    const bool vfirst = mode == 3 && ((threadIdx.x >> 8) & 1);
    auto mblock = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                asm volatile("" : "+v"(a[i]));
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b, acc[i], 0, 0, 0);
            }
    };
    auto vblock = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = fma(c[i], 1.0000001, 1e-9);
    };
    for (int it = 0; it < steps; ++it) {
        if (vfirst) {
            if (mode != 0) vblock();
            if (mode != 1) mblock();
        } else {
            if (mode != 1) mblock();
            if (mode != 0) vblock();
        }
        if (bar) __syncthreads();
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i];
    if (s == 12345.678) out[threadIdx.x] = s;
}
int main() {
    double *d;
    (void)hipMalloc(&d, 8192);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int steps = 2000;
    for (int bar = 0; bar < 2; ++bar)
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 10, mode, bar, 1.0);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, steps, mode, bar, 1.0);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("barrier %d, mode %d (%s): %.3f ms = %.0f ns per step (28 MFMA%s per wave, 2 waves per SIMD)\n", bar, mode,
                   mode == 0 ? "matrix only" : mode == 1 ? "vector only" : mode == 2 ? "both, same order" : "both, odd waves vector first",
                   ms, ms * 1e6 / steps, mode == 0 ? "" : " / 512 FMA");
        }
    return 0;
}
