// what v_fmac_f64_dpp row_newbcast does, lane by lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out) {
    const int l = threadIdx.x;
    double a = 100.0 + l, b = 1.0 + 0.001 * l, acc = 0.5;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b));
    out[l] = acc;
    double acc2 = 0.5;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "+v"(acc2) : "v"(a), "v"(b));
    out[64 + l] = acc2;
    double t, acc3 = 0.5;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a));
    acc3 = __builtin_fma(t, b, acc3);
    out[128 + l] = acc3;
}
int main() {
    double *d, h[192];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: fmac_dpp %.6f  with nops %.6f  mov+fma %.6f   expect %.6f\n", l, h[l], h[64 + l], h[128 + l],
                                           0.5 + (100.0 + 16 * (l / 16) + 3) * (1.0 + 0.001 * l));
    return 0;
}
