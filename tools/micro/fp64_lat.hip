// Issue rate vs dependent-issue latency of fp64 VALU instructions on gfx950, one wave per SIMD (what k_filter_fused
// runs as at B = 1e4: 157 waves on 1024 SIMDs).  Each test runs CHAINS independent dependency chains, interleaved,
// of LEN instructions each, in ONE wave; reports shader cycles per instruction (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
template <int CH, int OP>
__global__ void k(double *out, uint64_t *cyc, double a, double b, int iters) {
    double v[CH];
    for (int c = 0; c < CH; ++c) v[c] = 1.0 + 1e-3 * (threadIdx.x + c);
    const double sa = __builtin_bit_cast(double, __builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(long, a)) | ((long)__builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(long, a) >> 32)) << 32));
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                // plain C++: hipcc puts an s_nop after every inline-asm statement, which would be measured too
                if (OP == 0) v[c] = __builtin_fma(v[c], a, b);
                if (OP == 1) v[c] = __builtin_amdgcn_rcp(v[c]);
                if (OP == 2) v[c] = __builtin_amdgcn_rsq(v[c]);
                if (OP == 3) v[c] = v[c] * a;
                if (OP == 4) v[c] = v[c] + b;
                if (OP == 5) v[c] = __builtin_fma(v[c], sa, b);
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CH; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH, int OP>
void run(const char *name, int block) {
    double *d; uint64_t *c;
    hipMalloc(&d, sizeof(double) * 4096); hipMalloc(&c, sizeof(uint64_t) * 8);
    const int iters = 1 << 15;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH, OP><<<1, block>>>(d, c, 0.999999, 1e-7, 256);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CH, OP><<<1, block>>>(d, c, 0.999999, 1e-7, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h = 0;
    hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * CH;
    const int waves_per_simd = block <= 256 ? 1 : block / 256;
    printf("%-34s block %4d: %6.2f ns per instr per wave (%6.2f per chain step); SIMD: %5.2f ns per wave-instr; memtime %.1f ticks/us\n",
           name, block, ms * 1e6 / n, ms * 1e6 / n * CH, ms * 1e6 / n / waves_per_simd, h / (ms * 1e3));
    hipFree(d); hipFree(c);
}

int main() {
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0;
    hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("clock rate %d kHz, wall clock rate %d kHz (s_memtime ticks at the latter)\n", clk, wclk);
    run<1, 0>("fma dependent x1", 64);
    run<2, 0>("fma 2 chains", 64);
    run<3, 0>("fma 3 chains", 64);
    run<4, 0>("fma 4 chains", 64);
    run<8, 0>("fma 8 chains", 64);
    run<1, 5>("fma (sgpr operand) dependent x1", 64);
    run<4, 5>("fma (sgpr operand) 4 chains", 64);
    run<1, 3>("mul dependent x1", 64);
    run<4, 3>("mul 4 chains", 64);
    run<1, 4>("add dependent x1", 64);
    run<4, 4>("add 4 chains", 64);
    run<1, 1>("rcp dependent x1", 64);
    run<2, 1>("rcp 2 chains", 64);
    run<4, 1>("rcp 4 chains", 64);
    run<1, 2>("rsq dependent x1", 64);
    run<4, 2>("rsq 4 chains", 64);
    // two / four waves per SIMD (block 512 = 8 waves on 4 SIMDs; 1024 = 16 waves)
    run<1, 0>("fma dependent x1, 2 waves/SIMD", 512);
    run<1, 0>("fma dependent x1, 4 waves/SIMD", 1024);
    run<4, 0>("fma 4 chains, 2 waves/SIMD", 512);
    run<8, 0>("fma 8 chains, 2 waves/SIMD", 512);
    run<8, 0>("fma 8 chains, 4 waves/SIMD", 1024);
    run<1, 1>("rcp dependent, 4 waves/SIMD", 1024);
    run<1, 0>("fma dependent x1, 16 lanes", 16);
    return 0;
}
