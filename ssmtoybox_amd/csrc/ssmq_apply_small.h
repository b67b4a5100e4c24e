// Batched moment transform, one trajectory per lane, everything in registers ("small" shapes: E*N up to ~100).
//
// HBM layout: SoA planes [element][ld]; lane b of a wave reads element e at ptr[e*ld + b], so every global access of a
// wave is one contiguous 512-byte segment.  Per trajectory the kernel reads D + D(D+1)/2 doubles and writes
// E + E*E + E*D doubles; the transform constants (xi, wm, Wc, Wcc, emv, iK) are read with wave-uniform addresses
// (scalar loads, scalar cache / L2 resident) and never count towards per-trajectory traffic.
//
// Algorithm per trajectory (bq/bqmtran.py:60-109, 158-223; mtran.py:105-149):
//   L = chol(cov); x_n = mean + L xi_n; fx_n = f(x_n);
//   BQ form:    mean_f = fx wm; cov_f = (fx Wc) fx' - mean_f mean_f' + emv (+ cov_add); cov_fx = (fx Wcc') L'
//   SIGMA form: mean_f = fx wm; dfx = fx - mean_f; cov_f = dfx diag(wc) dfx'; cov_fx = dfx diag(wc) (x - mean)'
//   TP:         emv_e = (nu - 2 + fx_e iK fx_e') / (nu - 2 + N) * emv_e   (bq/bqmod.py:1132-1160)
#pragma once
#include "ssmq_device.h"

namespace ssmq {

constexpr int kSmallBlock = 64;

// Keeps hipcc's scheduler from interleaving the fully unrolled per-sigma-point / per-column bodies: without it the
// live ranges of all N bodies overlap and the D = 6 kernel needs > 512 registers (170 spills).
#define SSMQ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Sub-state selection (MeasurementModel.state_index, ssmod.py:990-991) is a compile-time pattern here: a run-time
// index would need 2*D*DIN scalar condition masks live across the whole kernel.  SEL 0: the leading entries (the
// default, state_index=None); SEL 1: entries (0, 2) (position components of [x, vx, y, vy, ...] states, the only other
// pattern the reference's tests and research scripts use).  Anything else runs on the generic kernel.
template <int D, int DIN, int SEL>
__device__ __forceinline__ void select_inputs(const double (&x)[D], double (&xs)[DIN]) {
#pragma unroll
    for (int k = 0; k < DIN; ++k) {
        const int src = (SEL == 1) ? 2 * k : k;
        xs[k] = x[src < D ? src : 0];
    }
}

template <int D, int E, int N, int F, int FORM, int TP, int SEL>
__global__ __launch_bounds__(kSmallBlock) void k_apply_small(const ApplyArgs a) {
    const uint32_t b = blockIdx.x * kSmallBlock + threadIdx.x;  // 32-bit lane offset: plane base stays scalar
    if ((int64_t)b >= a.B) return;
    const int64_t ld = a.ld;
    const cdouble_p c = (cdouble_p)a.consts;
    const cdouble_p cadd = (cdouble_p)a.cov_add;
    constexpr ConstLayout cl = const_layout(D, E, N, FORM);
    using Fun = Fn<F>;
    constexpr int DIN = Fun::DIN;

    double m[D], L[D * (D + 1) / 2];
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = a.mean[d * ld + b];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[SSMQ_PK(i, j)] = a.cov[(i * D + j) * ld + b];
    const double t = a.time[a.time_stride ? b : 0];

    const bool ok = chol_packed<D>(L);

    Fun fn;
    fn.init(t, a.fp);

    double fx[E][N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double s = m[d];
#pragma unroll
            for (int k = 0; k <= d; ++k) s += L[SSMQ_PK(d, k)] * c[cl.xi + n * D + k];
            x[d] = s;
        }
        double xs[DIN], o[E];
        select_inputs<D, DIN, SEL>(x, xs);
        fn.template eval<E>(xs, o);
#pragma unroll
        for (int e = 0; e < E; ++e) fx[e][n] = o[e];
        SSMQ_SCHED_FENCE();
    }

    double mf[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double s = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n) s += fx[e][n] * c[cl.wm + n];
        mf[e] = s;
    }

    const double nan = __builtin_nan("");
    if (!ok) a.status[b] = 1;
    else a.status[b] = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) a.mean_f[e * ld + b] = ok ? mf[e] : nan;

    if (FORM == SSMQ_FORM_BQ) {
        // ---- covariance: (fx Wc) fx' - mean mean' + emv --------------------------------------------------------
        double cv[E * (E + 1) / 2];
#pragma unroll
        for (int i = 0; i < E * (E + 1) / 2; ++i) cv[i] = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double tj[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) s += fx[e][i] * c[cl.Wc + j * N + i];
                tj[e] = s;
            }
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int e2 = 0; e2 <= e; ++e2) cv[SSMQ_PK(e, e2)] += tj[e] * fx[e2][j];
            SSMQ_SCHED_FENCE();
        }
        // expected model variance: constant, or scaled by the data for a Student-t process model
        double em[E * (E + 1) / 2];
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int e2 = 0; e2 <= e; ++e2) {
                const bool use = (e == e2) || (a.emv_mode == SSMQ_EMV_BROADCAST);
                em[SSMQ_PK(e, e2)] = use ? c[cl.emv + e * E + e2] : 0.0;
            }
        if (TP) {
            double sv[E * (E + 1) / 2];
#pragma unroll
            for (int i = 0; i < E * (E + 1) / 2; ++i) sv[i] = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double tj[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) s += fx[e][i] * c[cl.iK + j * N + i];
                    tj[e] = s;
                }
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int e2 = 0; e2 <= e; ++e2) sv[SSMQ_PK(e, e2)] += tj[e] * fx[e2][j];
                SSMQ_SCHED_FENCE();
            }
            const double den = 1.0 / (a.tp_nu - 2.0 + (double)N);
#pragma unroll
            for (int i = 0; i < E * (E + 1) / 2; ++i) em[i] = (a.tp_nu - 2.0 + sv[i]) * den * em[i];
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int e2 = 0; e2 <= e; ++e2) {
                double v = cv[SSMQ_PK(e, e2)] - mf[e] * mf[e2] + em[SSMQ_PK(e, e2)];
                if (a.cov_add) v += cadd[e * E + e2];
                v = ok ? v : nan;
                a.cov_f[(e * E + e2) * ld + b] = v;
                if (e2 != e) a.cov_f[(e2 * E + e) * ld + b] = v;
            }
        // ---- cross-covariance: (fx Wcc') L' -------------------------------------------------------------------
        double g[E][D];
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int d = 0; d < D; ++d) g[e][d] = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const double w = c[cl.Wcc + n * D + d];
#pragma unroll
                for (int e = 0; e < E; ++e) g[e][d] += fx[e][n] * w;
            }
            SSMQ_SCHED_FENCE();
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
#pragma unroll
                for (int d = 0; d <= j; ++d) s += g[e][d] * L[SSMQ_PK(j, d)];
                a.cov_fx[(e * D + j) * ld + b] = ok ? s : nan;
            }
        }
    } else {
        // ---- classical centred form, diagonal covariance weights -----------------------------------------------
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int n = 0; n < N; ++n) fx[e][n] -= mf[e];
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int e2 = 0; e2 <= e; ++e2) {
                double s = 0.0;
#pragma unroll
                for (int n = 0; n < N; ++n) s += (fx[e][n] * c[cl.Wc + n]) * fx[e2][n];
                if (a.cov_add) s += cadd[e * E + e2];
                s = ok ? s : nan;
                a.cov_f[(e * E + e2) * ld + b] = s;
                if (e2 != e) a.cov_f[(e2 * E + e) * ld + b] = s;
            }
        double cx[E][D];
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int d = 0; d < D; ++d) cx[e][d] = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                // x_n - mean exactly as the reference forms it: (mean + L xi_n) - mean   (mtran.py:139,148)
                double s = m[d];
#pragma unroll
                for (int k = 0; k <= d; ++k) s += L[SSMQ_PK(d, k)] * c[cl.xi + n * D + k];
                const double dx = s - m[d];
#pragma unroll
                for (int e = 0; e < E; ++e) cx[e][d] += (fx[e][n] * c[cl.Wc + n]) * dx;
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int d = 0; d < D; ++d) a.cov_fx[(e * D + d) * ld + b] = ok ? cx[e][d] : nan;
    }
}

template <int D, int E, int N, int F, int FORM, int TP, int SEL>
inline hipError_t launch_apply_small(const ApplyArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + kSmallBlock - 1) / kSmallBlock);
    hipLaunchKernelGGL((k_apply_small<D, E, N, F, FORM, TP, SEL>), dim3(grid), dim3(kSmallBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace ssmq
