// Memory-pattern microbenchmark for the D = E = 6 moment transform (NOT product code): same traffic as
// k_apply_small<6,6,13> - 27 input planes read, 78 output planes written, 8 bytes per lane - with no arithmetic, in
// the SoA-plane layout and in a tile-blocked layout, to see what the access pattern alone costs at B = 1e5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NIN, int NOUT, int MODE>
__global__ __launch_bounds__(64) void k_mem(const double *in, double *out, long B, long ld) {
    const unsigned b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double v[NIN];
    if (MODE == 0 || MODE == 2) {          // SoA planes
#pragma unroll
        for (int e = 0; e < NIN; ++e) v[e] = in[e * ld + b];
    } else {                  // tile-blocked: [tile][e][64]
        const double *p = in + (long)blockIdx.x * NIN * 64 + threadIdx.x;
#pragma unroll
        for (int e = 0; e < NIN; ++e) v[e] = p[e * 64];
    }
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < NIN; ++e) s += v[e];
    if (MODE == 2) {          // SoA planes, non-temporal stores
#pragma unroll
        for (int e = 0; e < NOUT; ++e) __builtin_nontemporal_store(s + e, &out[e * ld + b]);
    } else if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < NOUT; ++e) out[e * ld + b] = s + e;
    } else {
        double *q = out + (long)blockIdx.x * NOUT * 64 + threadIdx.x;
#pragma unroll
        for (int e = 0; e < NOUT; ++e) q[e * 64] = s + e;
    }
}

// paired planes: element pair (2k, 2k+1) of trajectory b at ((k * ld) + b) * 2 -> 16 bytes per lane per access
template <int NIN2, int NOUT2, int NT = 0>
__global__ __launch_bounds__(64) void k_mem16(const double2 *in, double2 *out, long B, long ld) {
    const unsigned b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double2 v[NIN2];
#pragma unroll
    for (int e = 0; e < NIN2; ++e) v[e] = in[e * ld + b];
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < NIN2; ++e) s += v[e].x + v[e].y;
#pragma unroll
    for (int e = 0; e < NOUT2; ++e) {
        if (NT) {
            __builtin_nontemporal_store(s + e, &out[e * ld + b].x);
            __builtin_nontemporal_store(s - e, &out[e * ld + b].y);
        } else {
            out[e * ld + b] = make_double2(s + e, s - e);
        }
    }
}

template <int NIN2, int NOUT2>
static int run16(const char *name, long B, int sets) {
    const long ld = (B + 63) / 64 * 64;
    std::vector<double2 *> in(sets), out(sets);
    for (int i = 0; i < sets; ++i) {
        CK(hipMalloc((void **)&in[i], sizeof(double2) * ld * NIN2));
        CK(hipMalloc((void **)&out[i], sizeof(double2) * ld * NOUT2));
        CK(hipMemset(in[i], 0, sizeof(double2) * ld * NIN2));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(ld / 64);
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_mem16<NIN2, NOUT2>), dim3(grid), dim3(64), 0, 0, in[i % sets], out[i % sets], B, ld);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 60; ++i) hipLaunchKernelGGL((k_mem16<NIN2, NOUT2>), dim3(grid), dim3(64), 0, 0, in[i % sets], out[i % sets], B, ld);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 60 * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const double bytes = 16.0 * B * (NIN2 + NOUT2);
    printf("%-34s B=%ld  median %.2f us  -> %.0f GB/s\n", name, B, t[t.size() / 2], bytes / t[t.size() / 2] / 1e3);
    for (int i = 0; i < sets; ++i) { hipFree(in[i]); hipFree(out[i]); }
    return 0;
}

template <int NIN, int NOUT, int MODE>
static int run(const char *name, long B, int sets) {
    const long ld = (B + 63) / 64 * 64;
    std::vector<double *> in(sets), out(sets);
    for (int i = 0; i < sets; ++i) {
        CK(hipMalloc((void **)&in[i], sizeof(double) * ld * (NIN ? NIN : 1)));
        CK(hipMalloc((void **)&out[i], sizeof(double) * ld * (NOUT ? NOUT : 1)));
        CK(hipMemset(in[i], 0, sizeof(double) * ld * (NIN ? NIN : 1)));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(ld / 64);
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_mem<NIN, NOUT, MODE>), dim3(grid), dim3(64), 0, 0, in[i % sets], out[i % sets], B, ld);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 60; ++i) hipLaunchKernelGGL((k_mem<NIN, NOUT, MODE>), dim3(grid), dim3(64), 0, 0, in[i % sets], out[i % sets], B, ld);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 60 * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const double bytes = 8.0 * B * (NIN + NOUT);
    printf("%-34s B=%ld  median %.2f us  -> %.0f GB/s\n", name, B, t[t.size() / 2], bytes / t[t.size() / 2] / 1e3);
    for (int i = 0; i < sets; ++i) { hipFree(in[i]); hipFree(out[i]); }
    return 0;
}

int main() {
    const long B = 100000;
    run<27, 78, 0>("SoA   read 27 + write 78", B, 4);
    run<27, 78, 2>("SoA   read 27 + write 78 NT", B, 4);
    run<1, 78, 2>("SoA   write 78 only NT", B, 4);
    run<27, 78, 1>("tiled read 27 + write 78", B, 4);
    run<27, 1, 0>("SoA   read 27 only", B, 4);
    run<27, 1, 1>("tiled read 27 only", B, 4);
    run<1, 78, 0>("SoA   write 78 only", B, 4);
    run<1, 78, 1>("tiled write 78 only", B, 4);
    run16<14, 39>("pairs read 14x16B + write 39x16B", B, 4);
    run16<14, 1>("pairs read 14x16B only", B, 4);
    run16<1, 39>("pairs write 39x16B only", B, 4);
    run16<14, 31>("pairs 14 in + 31 out (packed cov)", B, 4);
    run<27, 78, 0>("SoA   read 27 + write 78 (B=1e6)", 1000000, 2);
    run<27, 78, 1>("tiled read 27 + write 78 (B=1e6)", 1000000, 2);
    return 0;
}
