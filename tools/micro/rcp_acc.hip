// accuracy of v_rcp_f64 / v_rsq_f64 on this GPU: max relative error over 2^24 random-ish operands per range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *out, double lo, double ratio, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double e_rcp = 0, e_rsq = 0, e_d1 = 0, e_s1 = 0;
    for (int j = i; j < n; j += gridDim.x * blockDim.x) {
        double x = lo * pow(ratio, (double)j / n) * (1.0 + 1e-3 * (j % 977) / 977.0);
        double r = __builtin_amdgcn_rcp(x), q = __builtin_amdgcn_rsq(x);
        e_rcp = fmax(e_rcp, fabs(fma(-x, r, 1.0)));
        e_rsq = fmax(e_rsq, fabs(fma(-x * q, q, 1.0)) * 0.5);
        // one NR + residual-corrected quotient 1.2345 / x vs IEEE
        double e = fma(-x, r, 1.0); double r1 = fma(r, e, r);
        double a = 1.2345, qq = a * r1; e = fma(-x, qq, a); qq = fma(e, r1, qq);
        e_d1 = fmax(e_d1, fabs(qq - a / x) / (a / x));
        // one Goldschmidt + one correction sqrt vs IEEE
        double g = x * q, h = 0.5 * q, rr = fma(-h, g, 0.5); g = fma(g, rr, g); h = fma(h, rr, h);
        double d = fma(-g, g, x); g = fma(d, h, g);
        e_s1 = fmax(e_s1, fabs(g - sqrt(x)) / sqrt(x));
    }
    out[4 * i] = e_rcp; out[4 * i + 1] = e_rsq; out[4 * i + 2] = e_d1; out[4 * i + 3] = e_s1;
}
int main() {
    const int T = 256 * 64;
    double *d; hipMalloc(&d, sizeof(double) * 4 * T);
    double *h = new double[4 * T];
    for (double lo : {1e-6, 1.0, 1e3}) {
        k<<<256, 64>>>(d, lo, 1e6, 1 << 24);
        hipMemcpy(h, d, sizeof(double) * 4 * T, hipMemcpyDeviceToHost);
        double m[4] = {0, 0, 0, 0};
        for (int i = 0; i < T; ++i) for (int c = 0; c < 4; ++c) m[c] = fmax(m[c], h[4 * i + c]);
        printf("range [%g, %g]: rcp rel err %.3e (2^%.1f)  rsq %.3e (2^%.1f)  div(1 NR + corr) %.3e  sqrt(1 GS + 1 corr) %.3e\n",
               lo, lo * 1e6, m[0], log2(m[0]), m[1], log2(m[1]), m[2], m[3]);
    }
    return 0;
}
