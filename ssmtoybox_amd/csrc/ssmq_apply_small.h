// Batched moment transform, one trajectory per lane, everything in registers ("small" shapes: E*N up to ~100).
//
// HBM layout: SoA planes [element][ld]; lane b of a wave reads element e at ptr[e*ld + b], so every global access of a
// wave is one contiguous 512-byte segment.  Per trajectory the kernel reads D + D(D+1)/2 doubles and writes
// E + E*E + E*D doubles; the transform constants (xi', wm, Wc', Wcc', emv, iK' - stored TRANSPOSED so that what one
// unrolled body consumes is contiguous: point n's D coordinates, column j of Wc) are read with wave-uniform addresses
// (scalar loads, scalar cache / L2 resident) and never count towards per-trajectory traffic.
//
// Algorithm per trajectory (bq/bqmtran.py:60-109, 158-223; mtran.py:105-149):
//   L = chol(cov); x_n = mean + L xi_n; fx_n = f(x_n);
//   BQ form:    mean_f = fx wm; cov_f = (fx Wc) fx' - mean_f mean_f' + emv (+ cov_add); cov_fx = (fx Wcc') L'
//   SIGMA form: mean_f = fx wm; dfx = fx - mean_f; cov_f = dfx diag(wc) dfx'; cov_fx = dfx diag(wc) (x - mean)'
//   TP:         emv_e = (nu - 2 + fx_e iK fx_e') / (nu - 2 + N) * emv_e   (bq/bqmod.py:1132-1160)
//
// The arithmetic lives in moment_transform_core<>, parameterised by a "sink" that receives every result element in
// registers: the stand-alone kernel's sink stores straight to the SoA planes, the fused filter loop
// (ssmq_filter_fused.h) keeps the results in registers for the next stage.
#pragma once
#include "ssmq_device.h"

namespace ssmq {

#ifndef SSMQ_SMALL_BLOCK
#define SSMQ_SMALL_BLOCK 64
#endif
constexpr int kSmallBlock = SSMQ_SMALL_BLOCK;

// Keeps hipcc's scheduler from interleaving the fully unrolled per-sigma-point / per-column bodies: without it the
// live ranges of all N bodies overlap and the D = 6 kernel needs > 512 registers (170 spills).
#define SSMQ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Hides the constant-block base pointer from value numbering so that loads repeated in successive unrolled bodies are
// re-issued (cheap scalar-cache hits) instead of being merged into one long-lived SGPR set that has to be spilled.
__device__ __forceinline__ cdouble_p launder(cdouble_p p) {
    asm volatile("" : "+s"(p));
    return p;
}

// Sub-state selection (MeasurementModel.state_index, ssmod.py:990-991) is a compile-time pattern here: a run-time
// index would need 2*D*DIN scalar condition masks live across the whole kernel.  SEL 0: the leading entries (the
// default, state_index=None); SEL 1: entries (0, 2, ...) (position components of [x, vx, y, vy, ...] states, the only
// other pattern the reference's tests and research scripts use).  Anything else runs on the generic kernel.
template <int D, int DIN, int SEL>
__device__ __forceinline__ void select_inputs(const double (&x)[D], double (&xs)[DIN]) {
#pragma unroll
    for (int k = 0; k < DIN; ++k) {
        const int src = (SEL == 1) ? 2 * k : k;
        xs[k] = x[src < D ? src : 0];
    }
}

// Software prefetch of wave-uniform constants.  The unrolled bodies are fenced from each other (SSMQ_SCHED_FENCE), so
// without help every body would start by waiting for its own scalar loads.  Instead body r issues the loads of body
// r + 1 at its top and pins them at its bottom (an empty asm that needs the values in SGPRs forces the s_waitcnt there,
// after the arithmetic of body r has covered the scalar-cache latency).
template <int CNT>
struct SBuf {
    double v[CNT];
};
template <int CNT>
__device__ __forceinline__ void sload(SBuf<CNT> &b, cdouble_p p) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) b.v[i] = p[i];
}
template <int CNT>
__device__ __forceinline__ void spin(const SBuf<CNT> &b) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) asm volatile("" ::"s"(b.v[i]));
}

// Pins a value in a VGPR at this point of the program: IR-level sinking otherwise moves whole dot products (e.g. the
// mean) down to their last use and re-creates the constant loads they need there.
__device__ __forceinline__ void pin_v(double &x) { asm volatile("" : "+v"(x)); }

struct CoreParams {
    cdouble_p c;      // transform constant block (const_layout)
    cdouble_p cadd;   // [E*E] added to the covariance (a block of zeros when the caller has nothing to add)
    int32_t emv_mode;
    double tp_nu;
    // Studentian filters turn the transformed covariances into scale matrices before the noise term is added
    // (ssinf.py:672-693): cov = cov_scale * cov + cadd, ccov = ccov_scale * ccov.  1.0 for everything else.
    double cov_scale, ccov_scale;
};

// Optional fast paths, selected by the host per transform handle (both verified there before use):
//   SSMQ_OPT_LDL  the covariance quadratic form fx Wc fx' is evaluated through Wc = U diag(d) U' (unit lower U, factored
//                 on the host in fp64 without pivoting and accepted only if U diag(d) U' reproduces Wc to 1e-14):
//                 g_j = fx u_j costs N - 1 - j FMAs per row instead of N, i.e. E N (N - 1) / 2 instead of E N^2.
//   SSMQ_OPT_UT   the unit points are [0 | c I | -c I] (unscented / fully-symmetric degree 3 / spherical-radial with a
//                 centre point): x_n = m +- c L[:, k] needs one FMA per coordinate instead of a row of L times xi_n.
// Either changes only the order of floating-point operations (differences of a few ulp of the intermediate sums).
//   SSMQ_OPT_SYM  (round 6; with SSMQ_OPT_UT, BQ form) the weights are invariant under each reflection x_k -> -x_k of the unscented
//                 point set (swap of points 1 + k and 1 + D + k): what a kernel with a diagonal length-scale matrix gives on
//                 these points (RBF: K, q, R, Q all commute with the reflections), up to the round-off of the weight
//                 computation - the host accepts it only if symmetrising changes no weight by more than 2e-13 of the largest.
//                 Then with S_k = f_{1+k} + f_{1+D+k}, A_k = f_{1+k} - f_{1+D+k}, G = [f_0 | S_1 .. S_D]:
//                   mean = G wm_s;  fx Wcc' = [gam_d A_d]  (row d of Wcc is gam_d (e_{1+d} - e_{1+D+d})');
//                   fx Wc fx' = G Mt G' + sum_k beta_k A_k A_k'  (Wc block-diagonalises: a dense (D+1) x (D+1) block on the
//                   symmetric combinations, a DIAGONAL one on the antisymmetric), Mt = U diag(d) U' as SSMQ_OPT_LDL does for Wc.
//                 D = E = 6, N = 13: 822 multiply-adds for the three reductions instead of 1 590 (the kernel: 2 559 -> ~1 800
//                 vector instructions per wave, and it is bound by exactly those at two waves per SIMD: DESIGN.md 3.1).
#define SSMQ_OPT_LDL 1
#define SSMQ_OPT_UT 2
#define SSMQ_OPT_SYM 4

// m: mean; L: in = packed lower triangle of cov, out = its Cholesky factor.  Returns false if cov is not PD (results
// are then garbage; the caller writes NaN).  Sink interface: mean(e, v), cov(e, e2, v) for e2 <= e, ccov(e, d, v).
template <int D, int E, int N, int F, int FORM, int TP, int SEL, bool NEED_CCOV, int OPT, class Sink>
__device__ __forceinline__ bool moment_transform_core(const double (&m)[D], double (&L)[D * (D + 1) / 2], double t,
                                                      const FPar &fp, const CoreParams &cp, Sink &out) {
    constexpr ConstLayout cl = const_layout(D, E, N, FORM);
    using Fun = Fn<F>;
    constexpr int DIN = Fun::DIN;
    const cdouble_p c = cp.c;
    // Tiny shapes (UNGM: D = E = 1, N = 3) need none of the register-pressure / prefetch machinery below, and inside the
    // fused time loop it would only stop the compiler from hoisting the ~40 constant loads out of the loop.
    constexpr bool kTiny = (N * N + 2 * D * N + E * E) <= 48;
#define SSMQ_FENCE_T() do { if (!kTiny) SSMQ_SCHED_FENCE(); } while (0)
#define SSMQ_LAUNDER_T(p) (kTiny ? (p) : launder(p))
#define SSMQ_SPIN_T(b) do { if (!kTiny) spin(b); } while (0)
#define SSMQ_PIN_T(x) do { if (!kTiny) pin_v(x); } while (0)

    const bool ok = chol_packed<D>(L);

    Fun fn;
    fn.init(t, fp);

    constexpr bool kUT = (OPT & SSMQ_OPT_UT) != 0 && N == 2 * D + 1;
    const double utc = kUT ? c[cl.utc] : 0.0;
    constexpr bool kSYM = (OPT & SSMQ_OPT_SYM) != 0 && kUT && FORM == SSMQ_FORM_BQ && !TP;
    double mf[E];
    if constexpr (kSYM) {
        // ---- reflection-symmetric weights (SSMQ_OPT_SYM): ONE pass over the centre point and the D point pairs -----------------
        // Step 0 is the centre point, step j = 1 + k the pair (1 + k, 1 + D + k): G_j = f+ + f-, A_k = f+ - f-.  Everything a step
        // can finish is finished in that step: the mean and covariance sums advance (the symmetric block as Mt = Ut diag(d) Ut'
        // with Ut unit UPPER triangular, so that g_j = G_j + sum_{i<j} Ut[i][j] G_i needs only the steps done so far), and column
        // k of the cross-covariance - which only involves the pairs 0 .. k (L is lower triangular) - is complete and LEAVES after
        // pair k.  The launch is bound by its 63 MB of stores (5.8 TB/s = 10.9 us) counted from the moment the first store can
        // issue (tools/mt6_timeline.py): with the cross-covariance planes - 46 % of the output - leaving during the integrand
        // evaluations instead of after them, the write pipe fills ~2 us earlier.  Constants per step: one record
        // [wm_j, d_j, gam_k, beta_k, Ut[0..j-1][j]] (const_layout: sym, stride sym_rs).
        constexpr int M = D + 1;
        constexpr int RS = cl.sym_rs;
        double G[E][M], cv[E * (E + 1) / 2], acc[E][D];
#pragma unroll
        for (int i = 0; i < E * (E + 1) / 2; ++i) cv[i] = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mf[e] = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) acc[e][j] = 0.0;
        }
        SBuf<RS> rc_;
        sload(rc_, SSMQ_LAUNDER_T(c) + cl.sym);
        SSMQ_SPIN_T(rc_);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            SBuf<RS> rn_;
            if (j + 1 < M) sload(rn_, SSMQ_LAUNDER_T(c) + cl.sym + (j + 1) * RS);
            const int k = j == 0 ? 0 : j - 1;
            double A[E];
            {
                double x[D], xs[DIN], o[E];
#pragma unroll
                for (int d = 0; d < D; ++d) x[d] = (j != 0 && d >= k) ? m[d] + L[SSMQ_PK(d >= k ? d : k, k)] * utc : m[d];
                select_inputs<D, DIN, SEL>(x, xs);
                fn.template eval<E>(xs, o);
                if (j == 0) {
#pragma unroll
                    for (int e = 0; e < E; ++e) G[e][0] = o[e];
                } else {
                    double o2[E];
#pragma unroll
                    for (int d = 0; d < D; ++d) x[d] = (d >= k) ? m[d] - L[SSMQ_PK(d >= k ? d : k, k)] * utc : m[d];
                    select_inputs<D, DIN, SEL>(x, xs);
                    fn.template eval<E>(xs, o2);
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        G[e][j] = o[e] + o2[e];
                        A[e] = o[e] - o2[e];
                    }
                }
            }
            // mean
#pragma unroll
            for (int e = 0; e < E; ++e) mf[e] += G[e][j] * rc_.v[0];
            // cross-covariance: column k is complete after pair k
            if (NEED_CCOV && j != 0) {
                const double gs = rc_.v[2] * cp.ccov_scale;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double g = A[e] * gs;
#pragma unroll
                    for (int jj = 0; jj < D; ++jj)
                        if (jj >= k) acc[e][jj] += g * L[SSMQ_PK(jj >= k ? jj : k, k)];
                    out.ccov(e, k, acc[e][k]);
                }
            }
            // covariance: symmetric block (column j of Ut), antisymmetric block (beta_k A_k A_k')
            {
                double g[E], dg[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double sacc = G[e][j];
#pragma unroll
                    for (int i = 0; i < M; ++i)
                        if (i < j) sacc += G[e][i] * rc_.v[4 + i];
                    g[e] = sacc;
                    dg[e] = sacc * rc_.v[1];
                }
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int e2 = 0; e2 <= e; ++e2) cv[SSMQ_PK(e, e2)] += dg[e] * g[e2];
                if (j != 0) {
                    double ba[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) ba[e] = A[e] * rc_.v[3];
#pragma unroll
                    for (int e = 0; e < E; ++e)
#pragma unroll
                        for (int e2 = 0; e2 <= e; ++e2) cv[SSMQ_PK(e, e2)] += ba[e] * A[e2];
                }
            }
#pragma unroll
            for (int i = 0; i < E * (E + 1) / 2; ++i) SSMQ_PIN_T(cv[i]);
#pragma unroll
            for (int e = 0; e < E; ++e) SSMQ_PIN_T(mf[e]);
            if (j + 1 < M) {
                SSMQ_SPIN_T(rn_);
                rc_ = rn_;
            }
            SSMQ_FENCE_T();
        }
#pragma unroll
        for (int e = 0; e < E; ++e) out.mean(e, mf[e]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
#pragma unroll
            for (int e2 = 0; e2 <= e; ++e2) {
                const bool use = (e == e2) || (cp.emv_mode == SSMQ_EMV_BROADCAST);
                const double em = use ? c[cl.emv + e * E + e2] : 0.0;
                double v = cv[SSMQ_PK(e, e2)] - mf[e] * mf[e2] + em;
                v = v * cp.cov_scale + cp.cadd[e * E + e2];
                out.cov(e, e2, v);
            }
            SSMQ_FENCE_T();
        }
        return ok;
    }
    double fx[E][N];
    SBuf<D> xic;
    if (!kUT) {
        sload(xic, c + cl.xi);
        SSMQ_SPIN_T(xic);
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        SBuf<D> xin;
        if (!kUT && n + 1 < N) sload(xin, c + cl.xi + (n + 1) * D);
        double x[D];
        if (kUT) {
            // point 0: the mean; point 1 + k: m + c L[:, k]; point 1 + D + k: m - c L[:, k]   (L lower triangular)
            const int k = (n == 0) ? 0 : (n - 1) % D;
            const double sc = (n == 0) ? 0.0 : ((n - 1) < D ? utc : -utc);
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] = (n != 0 && d >= k) ? m[d] + L[SSMQ_PK(d >= k ? d : k, k)] * sc : m[d];
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double s = m[d];
#pragma unroll
                for (int k = 0; k <= d; ++k) s += L[SSMQ_PK(d, k)] * xic.v[k];
                x[d] = s;
            }
        }
        double xs[DIN], o[E];
        select_inputs<D, DIN, SEL>(x, xs);
        fn.template eval<E>(xs, o);
#pragma unroll
        for (int e = 0; e < E; ++e) fx[e][n] = o[e];
        if (!kUT && n + 1 < N) {
            SSMQ_SPIN_T(xin);
            xic = xin;
        }
        SSMQ_FENCE_T();
    }

#pragma unroll
    for (int e = 0; e < E; ++e) {
        double s = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n) s += fx[e][n] * c[cl.wm + n];
        SSMQ_PIN_T(s);
        mf[e] = s;
        out.mean(e, s);
    }
    SSMQ_FENCE_T();

    if (FORM == SSMQ_FORM_BQ) {
        // ---- cross-covariance first, one output row at a time: (fx_e Wcc') L'.  After it L is dead, which keeps the
        //      covariance stage (fx + accumulators) inside 256 registers ---------------------------------------------
        if (NEED_CCOV) {
            // Two passes over halves of the output rows e, each streaming Wcc once, one input dimension d per body:
            //   g_e = fx_e . Wcc[d, :]  (N FMAs per row), then ccov[e][j] += g_e L[j][d] for j >= d; column d of the
            // result is complete after body d and leaves at once.  Per body: N scalar constants for ~N EH FMAs, long
            // enough to cover the scalar-cache latency of the next body's constants; live accumulators EH x D.
            constexpr int EH = (E + 1) / 2;
            constexpr int NPASS = (E + EH - 1) / EH;
            SBuf<N> wcur;
            sload(wcur, SSMQ_LAUNDER_T(c) + cl.Wcc);   // Wcc block is [D][N] here (row d contiguous)
            SSMQ_SPIN_T(wcur);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                double acc[EH][D];
#pragma unroll
                for (int q = 0; q < EH; ++q)
#pragma unroll
                    for (int j = 0; j < D; ++j) acc[q][j] = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const bool more = !(ps == NPASS - 1 && d == D - 1);
                    SBuf<N> wnext;
                    if (more) sload(wnext, SSMQ_LAUNDER_T(c) + cl.Wcc + ((d + 1) % D) * N);
#pragma unroll
                    for (int q = 0; q < EH; ++q) {
                        const int e = ps * EH + q;
                        if (e < E) {
                            double g = 0.0;
#pragma unroll
                            for (int n = 0; n < N; ++n) g += fx[e][n] * wcur.v[n];
#pragma unroll
                            for (int j = d; j < D; ++j) acc[q][j] += g * L[SSMQ_PK(j, d)];
                            out.ccov(e, d, acc[q][d] * cp.ccov_scale);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < EH; ++q)
#pragma unroll
                        for (int j = 0; j < D; ++j) SSMQ_PIN_T(acc[q][j]);
                    if (more) {
                        SSMQ_SPIN_T(wnext);
                        wcur = wnext;
                    }
                    SSMQ_FENCE_T();
                }
            }
        }
        constexpr bool kLDL = (OPT & SSMQ_OPT_LDL) != 0 && !TP;
        if (kLDL) {
            // ---- covariance through Wc = U diag(d) U':  fx Wc fx' = sum_j d_j g_j g_j',  g_j = fx u_j -------------------
            double cv[E * (E + 1) / 2];
#pragma unroll
            for (int i = 0; i < E * (E + 1) / 2; ++i) cv[i] = 0.0;
            SBuf<N> ucur;
            sload(ucur, SSMQ_LAUNDER_T(c) + cl.ldlU);
            SSMQ_SPIN_T(ucur);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                SBuf<N> unext;
                if (j + 1 < N) sload(unext, SSMQ_LAUNDER_T(c) + cl.ldlU + (j + 1) * N);
                const double dj = c[cl.ldlD + j];
                double g[E], dg[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double sacc = fx[e][j];
#pragma unroll
                    for (int i = 0; i < N; ++i)   // constant bounds + predicate: a (j + 1 .. N) loop is not unrolled
                        if (i > j) sacc += fx[e][i] * ucur.v[i];
                    g[e] = sacc;
                    dg[e] = sacc * dj;
                }
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int e2 = 0; e2 <= e; ++e2) cv[SSMQ_PK(e, e2)] += dg[e] * g[e2];
#pragma unroll
                for (int i = 0; i < E * (E + 1) / 2; ++i) SSMQ_PIN_T(cv[i]);
                if (j + 1 < N) {
                    SSMQ_SPIN_T(unext);
                    ucur = unext;
                }
                SSMQ_FENCE_T();
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
#pragma unroll
                for (int e2 = 0; e2 <= e; ++e2) {
                    const bool use = (e == e2) || (cp.emv_mode == SSMQ_EMV_BROADCAST);
                    const double em = use ? c[cl.emv + e * E + e2] : 0.0;
                    double v = cv[SSMQ_PK(e, e2)] - mf[e] * mf[e2] + em;
                    v = v * cp.cov_scale + cp.cadd[e * E + e2];
                    out.cov(e, e2, v);
                }
                SSMQ_FENCE_T();
            }
        } else {
        // ---- covariance: (fx Wc) fx' - mean mean' + emv, TWO OUTPUT ROWS AT A TIME -------------------------------
        // Rows (e0, e0 + 1) are accumulated over all columns j of Wc and stored as soon as they are complete, so the
        // E*E covariance stores of a wave are spread over the whole stage instead of forming a burst at its end (with
        // every wave of the launch in the same phase that burst ran at the HBM write limit while the ALUs idled).
        // Wc is streamed E/2 times from the scalar cache; only 2E - 1 accumulators are live instead of E(E+1)/2.
        constexpr int RB = 2;
        const double den = TP ? 1.0 / (cp.tp_nu - 2.0 + (double)N) : 0.0;
        SBuf<N> colc;
        sload(colc, SSMQ_LAUNDER_T(c) + cl.Wc);
        SSMQ_SPIN_T(colc);
        SBuf<N> kolc;   // same for iK when the model variance is the Student-t process one
        if (TP) {
            sload(kolc, SSMQ_LAUNDER_T(c) + cl.iK);
            SSMQ_SPIN_T(kolc);
        }
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += RB) {
            double cv[RB][E], sv[RB][E];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int e2 = 0; e2 < E; ++e2) cv[r][e2] = sv[r][e2] = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const bool more = !(e0 + RB >= E && j == N - 1);
                SBuf<N> coln, koln;
                if (more) sload(coln, SSMQ_LAUNDER_T(c) + cl.Wc + ((j + 1) % N) * N);
                if (TP && more) sload(koln, SSMQ_LAUNDER_T(c) + cl.iK + ((j + 1) % N) * N);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int e = e0 + r;
                    if (e < E) {
                        double tj = 0.0, uj = 0.0;
#pragma unroll
                        for (int i = 0; i < N; ++i) tj += fx[e][i] * colc.v[i];
#pragma unroll
                        for (int e2 = 0; e2 <= e; ++e2) cv[r][e2] += tj * fx[e2][j];
                        if (TP) {
#pragma unroll
                            for (int i = 0; i < N; ++i) uj += fx[e][i] * kolc.v[i];
#pragma unroll
                            for (int e2 = 0; e2 <= e; ++e2) sv[r][e2] += uj * fx[e2][j];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e2 = 0; e2 < E; ++e2) {
                        SSMQ_PIN_T(cv[r][e2]);   // keeps column j's work in body j (see pin_v)
                        if (TP) SSMQ_PIN_T(sv[r][e2]);
                    }
                if (more) {
                    SSMQ_SPIN_T(coln);
                    colc = coln;
                    if (TP) {
                        SSMQ_SPIN_T(koln);
                        kolc = koln;
                    }
                }
                SSMQ_FENCE_T();
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int e = e0 + r;
                if (e < E) {
#pragma unroll
                    for (int e2 = 0; e2 <= e; ++e2) {
                        const bool use = (e == e2) || (cp.emv_mode == SSMQ_EMV_BROADCAST);
                        double em = use ? c[cl.emv + e * E + e2] : 0.0;
                        if (TP) em = (cp.tp_nu - 2.0 + sv[r][e2]) * den * em;
                        double v = cv[r][e2] - mf[e] * mf[e2] + em;
                        v = v * cp.cov_scale + cp.cadd[e * E + e2];
                        out.cov(e, e2, v);
                    }
                }
            }
            SSMQ_FENCE_T();
        }
        }   // !kLDL
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
                s = s * cp.cov_scale + cp.cadd[e * E + e2];
                out.cov(e, e2, s);
            }
        if (NEED_CCOV) {
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
                    if (kUT) {
                        const int k = (n == 0) ? 0 : (n - 1) % D;
                        const double sc = (n == 0) ? 0.0 : ((n - 1) < D ? utc : -utc);
                        if (n != 0 && d >= k) s = m[d] + L[SSMQ_PK(d >= k ? d : k, k)] * sc;
                    } else {
#pragma unroll
                        for (int k = 0; k <= d; ++k) s += L[SSMQ_PK(d, k)] * c[cl.xi + n * D + k];
                    }
                    const double dx = s - m[d];
#pragma unroll
                    for (int e = 0; e < E; ++e) cx[e][d] += (fx[e][n] * c[cl.Wc + n]) * dx;
                }
                SSMQ_FENCE_T();
            }
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int d = 0; d < D; ++d) out.ccov(e, d, cx[e][d] * cp.ccov_scale);
        }
    }
    return ok;
#undef SSMQ_FENCE_T
#undef SSMQ_LAUNDER_T
#undef SSMQ_SPIN_T
#undef SSMQ_PIN_T
}

// Sink of the stand-alone kernel: every element goes straight to its SoA plane.
// Stand-alone transforms write their results once and nobody re-reads them soon: non-temporal stores keep them from
// displacing the inputs in L2 / Infinity Cache (22.9 -> 19.4 us on the D = E = 6 launch; non-temporal LOADS of the
// inputs made no difference).  Inside a filter's launch loop the next kernel reads the outputs right back, and there
// plain stores are 8 % faster (reentry UKF, B = 1e5) - hence two instantiations of the fast-path kernels, chosen at launch
// (a run-time branch around every store cost the whole gain: it splits the unrolled body into ~80 basic blocks).
// The fused filter kernels always stream (SSMQ_STORE).
#ifndef SSMQ_TEMPORAL_STORE
#define SSMQ_STORE(dst, v) __builtin_nontemporal_store((v), &(dst))
#else
#define SSMQ_STORE(dst, v) (dst) = (v)
#endif
template <int D, int E, bool NTS>
struct GlobalSink {
    double *mean_f, *cov_f, *cov_fx;
    int64_t ld;
    uint32_t b;
    template <typename T>
    __device__ __forceinline__ static void put(T &dst, T v) {
        if constexpr (NTS) SSMQ_STORE(dst, v);
        else dst = v;
    }
    // SSMQ_DIAG_* : timing-only diagnostic builds (tools/ab.sh); never defined in the product build.
    __device__ __forceinline__ void keep(double v) { asm volatile("" ::"v"(v)); }
    __device__ __forceinline__ void mean(int e, double v) {
#ifdef SSMQ_DIAG_NOSTORE_ALL
        keep(v);
#else
        put(mean_f[e * ld + b], v);
#endif
    }
    __device__ __forceinline__ void cov(int e, int e2, double v) {
#if defined(SSMQ_DIAG_NOSTORE_ALL) || defined(SSMQ_DIAG_NOSTORE_COV)
        keep(v);
#else
        put(cov_f[(e * E + e2) * ld + b], v);
        if (e2 != e) put(cov_f[(e2 * E + e) * ld + b], v);
#endif
    }
    __device__ __forceinline__ void ccov(int e, int d, double v) {
#ifdef SSMQ_DIAG_NOSTORE_ALL
        keep(v);
#else
        put(cov_fx[(e * D + d) * ld + b], v);
#endif
    }
};

// __launch_bounds__(64, 2): at least two waves per SIMD, i.e. at most 256 registers per lane.  The D = E = 6, N = 13
// kernel needs ~270 without the bound (one wave per SIMD, no latency hiding at all); with it hipcc spills 8 registers
// and B = 1e5 trajectories (1563 waves) are all resident at once.
// Timing-only diagnostic builds of this header (tools/build_file_variant.sh; never defined in the product build):
//   SSMQ_SMALL_LPW=n   n active lanes per wave (ceil(B / n) waves): the partial-wave launch the round-5 review asked to see measured
//   SSMQ_DIAG_STAMP    every wave leaves [start, inputs arrived, last store issued, stores acknowledged] (100 MHz counter) and its
//                      hardware id in g_small_stamps; read back with ssmq_diag_stamps() (tools/mt6_timeline.py)
#ifdef SSMQ_DIAG_STAMP
__device__ uint64_t g_small_stamps[8 * 32768];
extern "C" int ssmq_diag_stamps(uint64_t *dst, int n_waves) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_small_stamps), sizeof(uint64_t) * 8 * (size_t)n_waves);
}
#endif
#ifdef SSMQ_SMALL_LPW
constexpr int kSmallLpw = SSMQ_SMALL_LPW;
#else
constexpr int kSmallLpw = kSmallBlock;
#endif

template <int D, int E, int N, int F, int FORM, int TP, int SEL, int OPT, bool NTS = false>
__global__ __launch_bounds__(kSmallBlock, 2) void k_apply_small(const ApplyArgs a) {
#ifdef SSMQ_DIAG_STAMP
    const uint64_t t_start = wall_clock64();
#endif
    const uint32_t b = blockIdx.x * kSmallLpw + threadIdx.x;  // 32-bit lane offset: plane base stays scalar
    if (kSmallLpw != kSmallBlock && (int)threadIdx.x >= kSmallLpw) return;
    if ((int64_t)b >= a.B) return;
    const int64_t ld = a.ld;
    double m[D], L[D * (D + 1) / 2];
#ifdef SSMQ_DIAG_NOLOAD
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = 6500.0 + 1e-7 * (double)b + d;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[SSMQ_PK(i, j)] = (i == j) ? 1e-4 : 1e-9 * (double)(b & 255);
#else
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = a.mean[d * ld + b];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[SSMQ_PK(i, j)] = a.cov[(i * D + j) * ld + b];
#endif
    const double t = a.time[a.time_stride ? b : 0];
#ifdef SSMQ_DIAG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t_loaded = wall_clock64();
#endif

    CoreParams cp{(cdouble_p)a.consts, (cdouble_p)a.cov_add, a.emv_mode, a.tp_nu, a.cov_scale, a.ccov_scale};
    GlobalSink<D, E, NTS> sink{a.mean_f, a.cov_f, a.cov_fx, ld, b};
    const bool ok = moment_transform_core<D, E, N, F, FORM, TP, SEL, true, OPT>(m, L, t, a.fp, cp, sink);
    a.status[b] = ok ? 0 : 1;
#ifdef SSMQ_DIAG_STAMP
    {
        asm volatile("" ::: "memory");
        const uint64_t t_issued = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t_done = wall_clock64();
        if (threadIdx.x == 0 && blockIdx.x < 32768) {
            uint64_t *o = g_small_stamps + 8 * (size_t)blockIdx.x;
            o[0] = t_start; o[1] = t_loaded; o[2] = t_issued; o[3] = t_done;
            o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID: wave / SIMD / CU / SH / SE
            o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        }
    }
#endif
    if (!ok) {
        // the reference raises LinAlgError here (bq/bqmtran.py:98); a batch marks the item and poisons its outputs
        const double nan = __builtin_nan("");
#pragma unroll
        for (int e = 0; e < E; ++e) {
            sink.mean(e, nan);
#pragma unroll
            for (int e2 = 0; e2 <= e; ++e2) sink.cov(e, e2, nan);
#pragma unroll
            for (int d = 0; d < D; ++d) sink.ccov(e, d, nan);
        }
    }
}

template <int D, int E, int N, int F, int FORM, int TP, int SEL, int OPT>
inline hipError_t launch_apply_small(const ApplyArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + kSmallLpw - 1) / kSmallLpw);
    if constexpr (OPT != 0) {   // the bandwidth-bound shapes: streaming stores for stand-alone calls (a.stream_out)
        if (a.stream_out) {
            hipLaunchKernelGGL((k_apply_small<D, E, N, F, FORM, TP, SEL, OPT, true>), dim3(grid), dim3(kSmallBlock), 0, s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((k_apply_small<D, E, N, F, FORM, TP, SEL, OPT, false>), dim3(grid), dim3(kSmallBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace ssmq
