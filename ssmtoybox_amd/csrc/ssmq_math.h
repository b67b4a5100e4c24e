// Elementary functions of the integrands (ssmod.py: np.sin / np.cos / np.arctan2 / np.exp / np.sqrt on sigma points)
// as short BRANCH-FREE fp64 sequences, a third of the library routines' instruction count.  The D >= 5 fused filters
// are bound by fp64 instruction issue and these expansions were most of their instructions.  A library
// call behind a range test was measured and dropped: inlined at the 22-26 integrand sites of a filter step it made the
// reentry filters 25-100 % SLOWER (every site becomes a branch, the unrolled sigma points no longer interleave); as a
// non-inlined function the code was wrong.  So the domain is stated instead:
//
//   sincos_nr   three-term Cody-Waite reduction by pi/2 (FMA), fdlibm kernel polynomials (degree 13 / 14).  Within 2.5 ulp
//               of libm for |x| <= 1e5; absolute error below 1e-12 up to |x| = 1e9 (tests/test_math_sequences.py); angles
//               beyond 2^30 (a state no filter recovers from) give NaN - and with it a status flag at the next
//               factorisation - where libm returns a value.  NaN / inf -> NaN as libm.
//   atan2_nr    ONE division - (mn - mx) / (mn + mx) beyond tan(pi/8) - and fdlibm's 11-coefficient polynomial on
//               |t| <= 0.4375.  Within 2 ulp of libm for |x| + |y| in [1e-290, 1e290]; atan2(+-0, +-0) as libm (0 or pi);
//               NaN -> NaN; an INFINITE operand gives NaN where libm returns a multiple of pi/4.
// Compiles on the host as well (tests/test_math_sequences.py builds this header with g++ and compares with libm over
// the ranges; the reciprocal seed is emulated at the accuracy measured on gfx950, 2^-24).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SSMQ_HD __host__ __device__ __forceinline__
#else
#define SSMQ_HD static inline
#endif

namespace ssmq {

// quotient with one Newton round on the hardware reciprocal and a residual correction (ssmq_device.h: div_nr)
SSMQ_HD double div_seeded(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(b);
#else
    double r = (double)(float)(1.0 / b);       // a 2^-24 seed, as measured for v_rcp_f64
#endif
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    const double q = a * r;
    e = fma(-b, q, a);
    return fma(e, r, q);
}

SSMQ_HD void sincos_nr(double x, double *sn, double *cs) {
    // beyond 2^30 (and for NaN / inf) the result is NaN, not the garbage of an overflowed quadrant count: a diverged state then
    // shows up as a failed factorisation at the next step (status flag) instead of as a plausible-looking number
    x = fabs(x) <= 1073741824.0 ? x : NAN;
    // n = nearest integer to x 2 / pi; r = x - n pi/2 in three pieces of pi/2 (33 + 33 + 53 bits: fdlibm's pio2_1, pio2_2, pio2_2t)
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632673412561417e+00, x);
    r = fma(-n, 6.07710050630396597660e-11, r);
    r = fma(-n, 2.02226624879595063154e-21, r);
    const double z = r * r;
    // fdlibm __kernel_sin / __kernel_cos coefficients
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)(n == n ? n : 0.0);       // (the conversion of a NaN is undefined in C)
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}
SSMQ_HD double sin_nr(double x) {
    double s, c;
    sincos_nr(x, &s, &c);
    return s;
}

SSMQ_HD double atan2_nr(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    // atan(mn / mx) on [0, 1]: beyond tan(pi/8) as pi/4 + atan((mn - mx) / (mn + mx)), argument in [-0.4143, 0]
    const bool big = mn > 0.41421356237309503 * mx;
    const double num = big ? mn - mx : mn;
    double den = big ? mn + mx : mx;
    den = den == 0.0 ? 1.0 : den;                  // both operands zero: t = 0
    const double t = div_seeded(num, den);
    const double z = t * t, w = z * z;
    // fdlibm atan: aT[0..10], even and odd halves
    double s1 = 1.62858201153657823623e-02;
    s1 = fma(s1, w, 4.97687799461593236017e-02);
    s1 = fma(s1, w, 6.66107313738753120669e-02);
    s1 = fma(s1, w, 9.09088713343650656196e-02);
    s1 = fma(s1, w, 1.42857142725034663711e-01);
    s1 = fma(s1, w, 3.33333333333329318027e-01);
    s1 *= z;
    double s2 = -3.65315727442169155270e-02;
    s2 = fma(s2, w, -5.83357013379057348645e-02);
    s2 = fma(s2, w, -7.69187620504482999495e-02);
    s2 = fma(s2, w, -1.11111104054623557880e-01);
    s2 = fma(s2, w, -1.99999999998764832476e-01);
    s2 *= w;
    // atan(t) = t - t (s1 + s2); with the pi/4 offset in two pieces
    const double pt = t * (s1 + s2);
    double a = big ? 7.85398163397448278999e-01 - ((pt - 3.06161699786838301793e-17) - t) : t - pt;
    a = ay > ax ? 1.57079632679489655800e+00 - (a - 6.12323399573676603587e-17) : a;
    a = signbit(x) ? 3.14159265358979311600e+00 - (a - 1.22464679914735317720e-16) : a;
    // fmax / fmin drop a NaN operand: put it back (0 otherwise; an infinite operand becomes NaN as well)
    return copysign(a, y) + ((x - x) + (y - y));
}

}  // namespace ssmq
