// Device-side building blocks shared by every kernel of libssmq: argument blocks, the in-register Cholesky, and the
// closed-form integrands of the reference's ssmod.py as __device__ functors.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/ssmq.h"
#include "ssmq_math.h"

namespace ssmq {

// The SSMQ_* switches (A/B toggles of tools/alt_paths.sh, test hooks).  The library does not call getenv() per launch: every
// switch is looked up in a process-wide SNAPSHOT that is filled on first use and thrown away only when the environment itself
// has changed since (a signature over the `environ` pointer array - setenv / putenv replace the entry's pointer), so a test may
// still flip a switch between two calls.  nullptr = unset.  The returned string lives for the life of the process.
const char *sw(const char *name);


typedef const __attribute__((address_space(4))) double *cdouble_p;   // constant address space: scalar loads

// Integrand constants as they travel in the kernel-argument segment (wave-uniform -> SGPRs).
struct FPar {
    double p[SSMQ_MAX_FPAR];
    int32_t idx[SSMQ_MAX_FIDX];
    int32_t n_idx;
    int32_t n_par;
    // Optional table of the integrand's time-dependent constant for integer times 0..T-1 (filter loops: the time index is
    // the step counter), filled on the host by time_table(); null = evaluate the transcendental on the device.
    const double *ttab;
    // Fused time loops fetch the table entry of the NEXT step while the current one computes and hand it over here
    // (use_tval = 1): the value ttab[(int) t] without a load on the step's dependency chain.  Zero from the host.
    double tval;
    int32_t use_tval;
};

// Offsets (in doubles) into a transform's constant block in HBM; every thread reads it with wave-uniform addresses,
// so the loads are scalar (s_load) and the block stays in the scalar cache / L2.
struct ConstLayout {
    int32_t xi, wm, Wc, Wcc, emv, iK, zero, ldlU, ldlD, utc, sym, sym_rs, rec, rs, total;
};
__host__ __device__ constexpr inline ConstLayout const_layout(int D, int E, int N, int form) {
    ConstLayout c{};
    c.xi = 0;
    c.wm = c.xi + D * N;
    c.Wc = c.wm + N;
    c.Wcc = c.Wc + (form == SSMQ_FORM_SIGMA ? N : N * N);
    c.emv = c.Wcc + D * N;
    c.iK = c.emv + E * E;
    c.zero = c.iK + N * N;   // E*E zeros: the default `cov_add`, so that the kernels add it unconditionally
    // optional fast-path data (see SSMQ_OPT_* in ssmq_apply_small.h)
    c.ldlU = c.zero + E * E;   // [N][N]: column j of the unit lower factor U of Wc = U diag(d) U', contiguous
    c.ldlD = c.ldlU + N * N;   // [N]
    c.utc = c.ldlD + N;        // [2]: scale c of unscented-type points [0 | c I | -c I]
    // per-point records for the kernels that pick their sigma points at run time (ssmq_filter_wsplit.hip: wave w takes points w,
    // w + W, ...): everything point n contributes, contiguous, so that ONE base pointer per point serves every scalar load -
    //   centred form:  xi_n [D] | wm_n | wc_n            uncentred BQ form:  xi_n [D] | wm_n | Wcc[:, n] [D] | Wc[:, n] [N] | iK[:, n] [N]
    // N + 1 records, the last one all zeros: the "point" of a wave that has run out of points (weight 0, evaluated at the mean).
    // reflection-symmetric weights on unscented-type points (SSMQ_OPT_SYM in ssmq_apply_small.h), M = D + 1 step records of
    // stride sym_rs: [wm_j, d_j, gam_k, beta_k, Ut[0][j] .. Ut[j-1][j], 0 ..] (k = j - 1; step 0 = the centre point)
    c.sym = c.utc + 2;
    c.sym_rs = (D + 1) + 4;
    c.rec = c.sym + (D + 1) * c.sym_rs;
    c.rs = form == SSMQ_FORM_SIGMA ? D + 2 : 2 * D + 1 + 2 * N;
    c.total = c.rec + (N + 1) * c.rs;
    return c;
}

struct ApplyArgs {
    const double *mean;     // [D][ld]
    const double *cov;      // [D*D][ld], lower triangle read
    const double *time;     // [B] or [1]
    double *mean_f;         // [E][ld]
    double *cov_f;          // [E*E][ld]
    double *cov_fx;         // [E*D][ld]
    int32_t *status;        // [B]
    const double *consts;   // transform constant block
    const double *cov_add;  // [E*E] added to cov_f after the model variance (G Q G' / R of the filters); never null
    int64_t B, ld;
    int32_t time_stride;
    int32_t emv_mode;
    double tp_nu;
    double cov_scale, ccov_scale;   // 1.0 except inside Studentian filters (ssinf.py:672-693)
    int32_t stream_out;             // 1: the outputs are final (non-temporal stores); 0: the next kernel of a filter loop
                                    // reads them right back, keep them in L2 / Infinity Cache
    FPar fp;
};

// Transform constants are read through the constant address space: wave-uniform constant-space loads are always
// selected as scalar loads (s_load), even after the kernel has started storing its outputs.
#define SSMQ_PK(i, j) ((i) * ((i) + 1) / 2 + (j))  // packed lower-triangular index, j <= i

// Square root, reciprocal square root and quotient for operands in the normal range (covariance pivots, 1 + x^2, ...).
// hipcc expands sqrt() / operator/ on doubles into v_rsq_f64 / v_rcp_f64 + FMA refinement wrapped in range rescaling
// (v_div_scale / v_div_fmas / v_div_fixup, ldexp + class tests) with two refinement rounds: 12 instructions per
// division, 17 per square root, most of them on the dependent chain.  Measured on gfx950 (tools/micro/rcp_acc.hip):
// both seeds are accurate to 2^-24.2, so ONE Newton / Goldschmidt round (2^-48) plus the residual correction already
// lands on the correctly rounded result - bit-identical to the compiler's sequence on 5e7 operands in [1e-6, 1e9].
// The recursions of the fused filter are bound by exactly this chain, so the hot callers use these (6 and 8
// instructions).  Valid for |x| in [2^-500, 2^500]; a non-positive pivot still comes out NaN / flagged.
// SSMQ_IEEE_DIVSQRT=1 at build time restores the compiler's sequences.
#ifndef SSMQ_IEEE_DIVSQRT
__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &rs) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    s = g;
    const double r2 = fma(-h, g, 0.5);   // h -> 1 / (2 sqrt x) once more, against the final g
    h = fma(h, r2, h);
    rs = h + h;
}
__device__ __forceinline__ double div_nr(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    const double q = a * r;
    e = fma(-b, q, a);
    return fma(e, r, q);
}
#else
__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &rs) {
    s = sqrt(x);
    rs = 1.0 / s;
}
__device__ __forceinline__ double div_nr(double a, double b) { return a / b; }
#endif

// exp(x) for |x| < 700 without the special-case handling of the library routine: x = n ln 2 + r, |r| <= ln 2 / 2,
// Taylor polynomial of degree 13 in r (remainder < 4e-18), scaled by 2^n.  ~19 instructions against ~31; within 1 ulp.
__device__ __forceinline__ double exp_nr(double x) {
    const double n = rint(x * 1.4426950408889634);
    double r = fma(-n, 0.6931471805599453, x);
    r = fma(-n, 2.3190468138462996e-17, r);
    double p = 1.6059043836821613e-10;            // 1 / 13!
    p = fma(p, r, 2.08767569878681e-09);          // 1 / 12!
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// In-register lower Cholesky of a packed symmetric matrix, column by column with reciprocal scaling, the operation
// order of LAPACK dpotf2 'L' that numpy.linalg.cholesky ends in (bq/bqmtran.py:98).  Returns false at the first
// non-positive (or NaN) pivot - where the reference raises LinAlgError.
template <int D>
__device__ __forceinline__ bool chol_packed(double (&L)[D * (D + 1) / 2]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double ajj = L[SSMQ_PK(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) ajj -= L[SSMQ_PK(j, k)] * L[SSMQ_PK(j, k)];
        ok = ok && (ajj > 0.0);
        double r;
        sqrt_rsqrt(ajj, ajj, r);
        L[SSMQ_PK(j, j)] = ajj;
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double s = L[SSMQ_PK(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[SSMQ_PK(i, k)] * L[SSMQ_PK(j, k)];
            L[SSMQ_PK(i, j)] = s * r;
        }
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
// Integrands.  Each functor: DIN = leading inputs it reads, init(t, par) once per trajectory, eval<E>(x, out).
// Formulas: ssmod.py lines cited in include/ssmq.h.  Additive-noise models are evaluated with zero noise
// (ssmod.py:153-158, 993-998).
// ---------------------------------------------------------------------------------------------------------------
template <int F>
struct Fn;

template <>
struct Fn<SSMQ_F_UNGM_DYN> {
    static constexpr int DIN = 1;
    double c;
    __device__ __forceinline__ void init(double t, const FPar &p) {
        c = p.use_tval ? p.tval : (p.ttab ? ((cdouble_p)p.ttab)[(int)t] : 8.0 * cos(1.2 * t));
    }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        o[0] = 0.5 * x[0] + 25.0 * div_nr(x[0], 1.0 + x[0] * x[0]) + c;
    }
};
template <>
struct Fn<SSMQ_F_UNGM_MEAS> {
    static constexpr int DIN = 1;
    __device__ __forceinline__ void init(double, const FPar &) {}
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        o[0] = 0.05 * (x[0] * x[0]);
    }
};
template <>
struct Fn<SSMQ_F_UNGMNA_DYN> {
    static constexpr int DIN = 2;
    double c;
    __device__ __forceinline__ void init(double t, const FPar &p) {
        c = p.use_tval ? p.tval : (p.ttab ? ((cdouble_p)p.ttab)[(int)t] : cos(1.2 * t));
    }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        o[0] = 0.5 * x[0] + 25.0 * div_nr(x[0], 1.0 + x[0] * x[0]) + 8.0 * x[1] * c;
    }
};
template <>
struct Fn<SSMQ_F_UNGMNA_MEAS> {
    static constexpr int DIN = 2;
    __device__ __forceinline__ void init(double, const FPar &) {}
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        o[0] = 0.05 * x[1] * (x[0] * x[0]);
    }
};
template <>
struct Fn<SSMQ_F_PENDULUM_DYN> {
    static constexpr int DIN = 2;
    double dt;
    __device__ __forceinline__ void init(double, const FPar &p) { dt = p.p[0]; }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        o[0] = x[0] + x[1] * dt;
        o[1] = x[1] - 9.81 * dt * sin_nr(x[0]);
    }
};
template <>
struct Fn<SSMQ_F_PENDULUM_MEAS> {
    static constexpr int DIN = 1;
    __device__ __forceinline__ void init(double, const FPar &) {}
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        o[0] = sin_nr(x[0]);
    }
};
template <>
struct Fn<SSMQ_F_REENTRY1D_DYN> {
    static constexpr int DIN = 3;
    double dt;
    __device__ __forceinline__ void init(double, const FPar &p) { dt = p.p[0]; }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        const double gam = 1.0 / 6.096;
        o[0] = x[0] - dt * x[1];
        o[1] = x[1] - dt * exp_nr(-gam * x[0]) * (x[1] * x[1]) * x[2];
        o[2] = x[2];
    }
};
template <>
struct Fn<SSMQ_F_RANGE_MEAS> {
    static constexpr int DIN = 1;
    __device__ __forceinline__ void init(double, const FPar &) {}
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        const double sx = 30.0, sy = 30.0;
        double rg, irg;
        sqrt_rsqrt(sx * sx + (x[0] - sy) * (x[0] - sy), rg, irg);      // >= 900: normal range
        o[0] = rg;
    }
};
struct ReentryCore {
    double dt;
    __device__ __forceinline__ void init(double, const FPar &p) { dt = p.p[0]; }
    __device__ __forceinline__ void core(const double *x, double *o) const {
        const double r0 = 6374.0, h0 = 13.406, gm0 = 3.9860e5, b0 = -0.59783;
        // Same formula as ssmod.py:547-557, arranged for the fp64 vector ALU: the two exponentials b0 exp(x4) and
        // exp((R0 - R) / H0) are one exp of the summed argument, and R, 1 / R^3 come from one reciprocal square root
        // (differences from the reference's evaluation order are a few ulp, far inside the 1e-10 parity bar).
        const double r2 = x[0] * x[0] + x[1] * x[1];
        double rr, ir;          // the radius is never near zero on this model (Earth radius 6374): normal-range helper
        sqrt_rsqrt(r2, rr, ir);
        double vv, ivv;         // speed: clamped away from zero so that the helper's reciprocal seed stays finite
        sqrt_rsqrt(fmax(x[2] * x[2] + x[3] * x[3], 1e-290), vv, ivv);
        const double dr = b0 * exp_nr(x[4] + (r0 - rr) * (1.0 / h0)) * vv;
        const double gr = -gm0 * (ir * ir * ir);
        o[0] = x[0] + dt * x[2];
        o[1] = x[1] + dt * x[3];
        o[2] = x[2] + dt * (dr * x[2] + gr * x[0]);
        o[3] = x[3] + dt * (dr * x[3] + gr * x[1]);
        o[4] = x[4];
    }
};
template <>
struct Fn<SSMQ_F_REENTRY2D_DYN> : ReentryCore {
    static constexpr int DIN = 5;
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        core(x, o);
    }
};
template <>
struct Fn<SSMQ_F_REENTRY2D_BIAS_DYN> : ReentryCore {
    static constexpr int DIN = 6;
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        core(x, o);
        o[5] = x[5];
    }
};
template <>
struct Fn<SSMQ_F_SMOOTH10D_DYN> {
    static constexpr int DIN = 10;
    __device__ __forceinline__ void init(double, const FPar &) {}
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            double sn, cs;
            sincos_nr(x[i], &sn, &cs);
            o[i] = sn + x[5 + i] * x[5 + i];
            o[5 + i] = x[5 + i] * cs;
        }
    }
};
template <>
struct Fn<SSMQ_F_RADAR2D_MEAS> {
    static constexpr int DIN = 2;
    double lx, ly;
    __device__ __forceinline__ void init(double, const FPar &p) {
        lx = p.n_par >= 2 ? p.p[0] : 0.0;
        ly = p.n_par >= 2 ? p.p[1] : 0.0;
    }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        const double dx = x[0] - lx, dy = x[1] - ly;
        const double r2 = dx * dx + dy * dy;
        double rg, irg;
        sqrt_rsqrt(r2, rg, irg);                   // normal-range helper ...
        o[0] = r2 > 0.0 ? rg : r2;                  // ... and a target exactly at the radar
        o[1] = atan2_nr(dy, dx);
    }
};
template <>
struct Fn<SSMQ_F_CT_DYN> {
    static constexpr int DIN = 5;
    double dt;
    __device__ __forceinline__ void init(double, const FPar &p) { dt = p.p[0]; }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        const double om = x[4];
        double a, b;
        sincos_nr(om * dt, &a, &b);
        const double c = a / om, d = (1.0 - b) / om;
        o[0] = x[0] + c * x[1] - d * x[3];
        o[1] = b * x[1] - a * x[3];
        o[2] = d * x[1] + x[2] + c * x[3];
        o[3] = a * x[1] + b * x[3];
        o[4] = x[4];
    }
};
template <>
struct Fn<SSMQ_F_BEARING_MEAS> {
    static constexpr int DIN = 2;
    const FPar *fp;
    __device__ __forceinline__ void init(double, const FPar &p) { fp = &p; }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
#pragma unroll
        for (int s = 0; s < E; ++s) {
            // The register kernels (k_apply_small, k_filter_fused) instantiate with E = the sensor count: the loop bound is the
            // only bound, no run-time test inside the unrolled body (round 4 had one here: +34 % VALU, 44 scratch instructions in
            // the configs[3] filter).  The generic kernels pass E = SSMQ_MAX_FIDX and an output array of their compile-time
            // bound on D and E, below 8 for small shapes: there, one bearing per sensor (2 s < n_par) and no more.
            if constexpr (E == SSMQ_MAX_FIDX) {
                if (s < SSMQ_MAX_FPAR / 2 && 2 * s < fp->n_par) o[s] = atan2_nr(x[1] - fp->p[2 * s + 1], x[0] - fp->p[2 * s]);
            } else {
                static_assert(E <= SSMQ_MAX_FPAR / 2, "one (x, y) pair of integrand constants per sensor");
                o[s] = atan2_nr(x[1] - fp->p[2 * s + 1], x[0] - fp->p[2 * s]);
            }
        }
    }
};
template <>
struct Fn<SSMQ_F_CTRS_DYN> {
    static constexpr int DIN = 7;
    double dt;
    __device__ __forceinline__ void init(double, const FPar &p) { dt = p.p[0]; }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        const double q0 = x[5], q1 = x[6];
        double s3, c3;
        sincos_nr(x[3], &s3, &c3);
        double f0, f1;
        if (x[4] == 0.0) {
            f0 = dt * x[2] * c3;
            f1 = dt * x[2] * s3;
        } else {
            const double c = x[2] / x[4];
            double s34, c34;
            sincos_nr(x[3] + x[4] * dt, &s34, &c34);
            f0 = c * (s34 - s3) + 0.5 * dt * dt * c3 * q0;
            f1 = c * (-c34 + c3) + 0.5 * dt * dt * s3 * q0;
        }
        o[0] = x[0] + f0;
        o[1] = x[1] + f1;
        o[2] = x[2] + dt * q0;
        o[3] = x[3] + (dt * x[3] + 0.5 * dt * dt * q1);
        o[4] = x[4] + dt * q1;
    }
};
template <>
struct Fn<SSMQ_F_CV_DYN> {
    static constexpr int DIN = 4;
    double dt;
    __device__ __forceinline__ void init(double, const FPar &p) { dt = p.p[0]; }
    template <int E>
    __device__ __forceinline__ void eval(const double *x, double *o) const {
        o[0] = x[0] + dt * x[1];
        o[1] = x[1];
        o[2] = x[2] + dt * x[3];
        o[3] = x[3];
    }
};

// Integrand chosen at run time (generic kernels): xs holds the integrand's inputs (kMaxIntegrandIn of them: up to
// SSMQ_MAX_FIDX selected through a state index, or the leading ones), o up to SSMQ_MAX_DIM outputs.
constexpr int kMaxIntegrandIn = 16;
__device__ __forceinline__ void eval_integrand(int id, const double *xs, double t, const FPar &fp, double *o) {
#define SSMQ_CASE(F)                      \
    case F: {                             \
        Fn<F> fn;                         \
        fn.init(t, fp);                   \
        fn.template eval<SSMQ_MAX_FIDX>(xs, o); \
    } break;
    switch (id) {
        SSMQ_CASE(SSMQ_F_UNGM_DYN)
        SSMQ_CASE(SSMQ_F_UNGM_MEAS)
        SSMQ_CASE(SSMQ_F_UNGMNA_DYN)
        SSMQ_CASE(SSMQ_F_UNGMNA_MEAS)
        SSMQ_CASE(SSMQ_F_PENDULUM_DYN)
        SSMQ_CASE(SSMQ_F_PENDULUM_MEAS)
        SSMQ_CASE(SSMQ_F_REENTRY1D_DYN)
        SSMQ_CASE(SSMQ_F_RANGE_MEAS)
        SSMQ_CASE(SSMQ_F_REENTRY2D_DYN)
        SSMQ_CASE(SSMQ_F_RADAR2D_MEAS)
        SSMQ_CASE(SSMQ_F_CT_DYN)
        SSMQ_CASE(SSMQ_F_BEARING_MEAS)
        SSMQ_CASE(SSMQ_F_CTRS_DYN)
        SSMQ_CASE(SSMQ_F_CV_DYN)
        SSMQ_CASE(SSMQ_F_REENTRY2D_BIAS_DYN)
        SSMQ_CASE(SSMQ_F_SMOOTH10D_DYN)
        default: break;
    }
#undef SSMQ_CASE
}

// Jacobian of integrand `id` at xs with respect to its own inputs (the leading DIN of them): J[e * ldj + k] = d out_e / d xs_k.
// Only the models whose dyn_fcn_dx / meas_fcn_dx the reference implements (ssmod.py:271-272, 305-306, 363-365, 848-852,
// 1063-1064, 1088-1089, 1117-1118; every other model's returns None there and its ExtendedKalman raises).  Formulas as written
// in the reference - including ConstantVelocity's, which returns the TRANSPOSE of its transition matrix (ssmod.py:848-852).
// Returns false for an integrand without one.
__device__ __forceinline__ bool jac_integrand(int id, const double *xs, double t, const FPar &fp, double *J, int ldj) {
    switch (id) {
        case SSMQ_F_UNGM_DYN: {
            const double x2 = xs[0] * xs[0], d = 1.0 + x2;
            J[0] = 0.5 + div_nr(25.0 * (1.0 - x2), d * d);
            return true;
        }
        case SSMQ_F_UNGMNA_DYN: {
            Fn<SSMQ_F_UNGMNA_DYN> fn;
            fn.init(t, fp);                                  // c = cos(1.2 t), from the table where there is one
            const double x2 = xs[0] * xs[0], d = 1.0 + x2;
            J[0] = 0.5 + div_nr(25.0 * (1.0 - x2), d * d);
            J[1] = 8.0 * fn.c;
            return true;
        }
        case SSMQ_F_PENDULUM_DYN: {
            const double dt = fp.p[0];
            double sn, cs;
            sincos_nr(xs[0], &sn, &cs);
            J[0] = 1.0; J[1] = dt;
            J[ldj] = -9.81 * dt * cs; J[ldj + 1] = 1.0;
            return true;
        }
        case SSMQ_F_CV_DYN: {
            const double dt = fp.p[0];
            for (int e = 0; e < 4; ++e)
                for (int k = 0; k < 4; ++k) J[e * ldj + k] = (e == k) ? 1.0 : 0.0;
            J[1 * ldj + 0] = dt;                             // (the transpose of [[1, dt, 0, 0], [0, 1, 0, 0], [0, 0, 1, dt], [0, 0, 0, 1]])
            J[3 * ldj + 2] = dt;
            return true;
        }
        case SSMQ_F_UNGM_MEAS:
            J[0] = 0.1 * xs[0];
            return true;
        case SSMQ_F_UNGMNA_MEAS:
            J[0] = 0.1 * xs[1] * xs[0];
            J[1] = 0.05 * (xs[0] * xs[0]);
            return true;
        case SSMQ_F_PENDULUM_MEAS: {
            double sn, cs;
            sincos_nr(xs[0], &sn, &cs);
            J[0] = cs;
            return true;
        }
        default: return false;
    }
}
__host__ __device__ inline bool integrand_has_jacobian(int id) {
    return id == SSMQ_F_UNGM_DYN || id == SSMQ_F_UNGMNA_DYN || id == SSMQ_F_PENDULUM_DYN || id == SSMQ_F_CV_DYN ||
           id == SSMQ_F_UNGM_MEAS || id == SSMQ_F_UNGMNA_MEAS || id == SSMQ_F_PENDULUM_MEAS;
}

// Host: the time-dependent constant of integrand `id` for times 0..T-1 (what Fn<id>::init would compute), or false if the
// integrand has none.  Evaluated with the host libm in fp64 - the reference evaluates np.cos on the host as well.
__host__ inline bool time_table(int id, int T, double *out) {
    if (id == SSMQ_F_UNGM_DYN) {
        for (int k = 0; k < T; ++k) out[k] = 8.0 * cos(1.2 * (double)k);
        return true;
    }
    if (id == SSMQ_F_UNGMNA_DYN) {
        for (int k = 0; k < T; ++k) out[k] = cos(1.2 * (double)k);
        return true;
    }
    return false;
}

// Host-visible table: inputs read / outputs produced by each integrand (0 = "set by the transform's E").
struct FInfo {
    int din, dout;
    bool uses_time;
};
__host__ inline bool integrand_info(int id, FInfo *o) {
    switch (id) {
        case SSMQ_F_UNGM_DYN: *o = {1, 1, true}; return true;
        case SSMQ_F_UNGM_MEAS: *o = {1, 1, false}; return true;
        case SSMQ_F_UNGMNA_DYN: *o = {2, 1, true}; return true;
        case SSMQ_F_UNGMNA_MEAS: *o = {2, 1, false}; return true;
        case SSMQ_F_PENDULUM_DYN: *o = {2, 2, false}; return true;
        case SSMQ_F_PENDULUM_MEAS: *o = {1, 1, false}; return true;
        case SSMQ_F_REENTRY1D_DYN: *o = {3, 3, false}; return true;
        case SSMQ_F_RANGE_MEAS: *o = {1, 1, false}; return true;
        case SSMQ_F_REENTRY2D_DYN: *o = {5, 5, false}; return true;
        case SSMQ_F_RADAR2D_MEAS: *o = {2, 2, false}; return true;
        case SSMQ_F_CT_DYN: *o = {5, 5, false}; return true;
        case SSMQ_F_BEARING_MEAS: *o = {2, 0, false}; return true;
        case SSMQ_F_CTRS_DYN: *o = {7, 5, false}; return true;
        case SSMQ_F_CV_DYN: *o = {4, 4, false}; return true;
        case SSMQ_F_REENTRY2D_BIAS_DYN: *o = {6, 6, false}; return true;
        case SSMQ_F_SMOOTH10D_DYN: *o = {10, 10, false}; return true;
        default: return false;
    }
}

}  // namespace ssmq
