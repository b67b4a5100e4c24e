/*
 * CPU oracle in plain C for the per-step hot loop: one moment transform (BQ and classical form), the Gaussian
 * measurement update and the filter recursion around them.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is loaded only by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py ("port": the reference's algorithm restated in C and timed on the GPU box's host
 * cores).  The product (ssmtoybox_amd/, libssmq.so) never links or calls it.
 *
 * Restates, from the reference's published algorithm (paths under /root/reference/ssmtoybox):
 *   BQTransform.apply + _mean/_covariance/_cross_covariance   bq/bqmtran.py:60-109, 158-223 (TP: 394-415,
 *                                                             bq/bqmod.py:1132-1160)
 *   SigmaPointTransform.apply                                 mtran.py:105-149
 *   integrands                                                ssmod.py (lines cited at each case)
 *   GaussianInference._time_update / _measurement_update      ssinf.py:254-323, forward_pass ssinf.py:66-118
 * Pinned in tests/test_oracle_c.py against the golden vectors generated from the reference (the .npz fixtures under tests/golden) and
 * against the NumPy oracle.  All arithmetic is IEEE fp64; compile WITHOUT -ffast-math.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 16
#define MAXN 1024

enum {
    F_UNGM_DYN = 1, F_UNGM_MEAS, F_UNGMNA_DYN, F_UNGMNA_MEAS, F_PENDULUM_DYN, F_PENDULUM_MEAS, F_REENTRY1D_DYN,
    F_RANGE_MEAS, F_REENTRY2D_DYN, F_RADAR2D_MEAS, F_CT_DYN, F_BEARING_MEAS, F_CTRS_DYN, F_CV_DYN,
    F_REENTRY2D_BIAS_DYN,
    F_SMOOTH10D_DYN
};

typedef struct {
    int fid, npar, nidx;
    double par[16];
    int idx[16];
} orc_integrand;

/* lower Cholesky, LAPACK dpotf2 order (numpy.linalg.cholesky); returns 0 ok, 1 not positive definite */
static int chol_lower(int n, const double *a, double *l) {
    memset(l, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        double ajj = a[j * n + j];
        for (int k = 0; k < j; ++k) ajj -= l[j * n + k] * l[j * n + k];
        if (!(ajj > 0.0)) return 1;
        ajj = sqrt(ajj);
        l[j * n + j] = ajj;
        for (int i = j + 1; i < n; ++i) {
            double s = a[i * n + j];
            for (int k = 0; k < j; ++k) s -= l[i * n + k] * l[j * n + k];
            l[i * n + j] = s / ajj;
        }
    }
    return 0;
}

/* closed-form integrands, one input column -> one output column; returns the output dimension */
static int eval_integrand(const orc_integrand *f, const double *xin, double t, double *o) {
    double x[MAXD];
    const double *p = f->par;
    if (f->nidx > 0) { /* MeasurementModel.state_index, ssmod.py:990-991 */
        for (int k = 0; k < f->nidx; ++k) x[k] = xin[f->idx[k]];
    } else {
        memcpy(x, xin, sizeof(double) * MAXD);
    }
    switch (f->fid) {
        case F_UNGM_DYN: /* ssmod.py:268-269 */
            o[0] = 0.5 * x[0] + 25 * (x[0] / (1 + x[0] * x[0])) + 8 * cos(1.2 * t);
            return 1;
        case F_UNGM_MEAS: /* ssmod.py:1060-1061 */
            o[0] = 0.05 * (x[0] * x[0]);
            return 1;
        case F_UNGMNA_DYN: /* ssmod.py:299-300 */
            o[0] = 0.5 * x[0] + 25 * (x[0] / (1 + x[0] * x[0])) + 8 * x[1] * cos(1.2 * t);
            return 1;
        case F_UNGMNA_MEAS: /* ssmod.py:1085-1086 */
            o[0] = 0.05 * x[1] * (x[0] * x[0]);
            return 1;
        case F_PENDULUM_DYN: /* ssmod.py:357-358 */
            o[0] = x[0] + x[1] * p[0];
            o[1] = x[1] - 9.81 * p[0] * sin(x[0]);
            return 2;
        case F_PENDULUM_MEAS: /* ssmod.py:1114-1115 */
            o[0] = sin(x[0]);
            return 1;
        case F_REENTRY1D_DYN: /* ssmod.py:424-427 */
            o[0] = x[0] - p[0] * x[1];
            o[1] = x[1] - p[0] * exp(-(1 / 6.096) * x[0]) * (x[1] * x[1]) * x[2];
            o[2] = x[2];
            return 3;
        case F_RANGE_MEAS: /* ssmod.py:1147-1149 */
            o[0] = sqrt(30.0 * 30.0 + (x[0] - 30.0) * (x[0] - 30.0));
            return 1;
        case F_REENTRY2D_DYN:
        case F_REENTRY2D_BIAS_DYN: { /* ssmod.py:530-564 */
            const double dt = p[0], r0 = 6374.0, h0 = 13.406, gm0 = 3.9860e5, b0 = -0.59783;
            const double b = b0 * exp(x[4]);
            const double rr = sqrt(x[0] * x[0] + x[1] * x[1]);
            const double vv = sqrt(x[2] * x[2] + x[3] * x[3]);
            const double dr = b * exp((r0 - rr) / h0) * vv;
            const double gr = -gm0 / (rr * rr * rr);
            o[0] = x[0] + dt * x[2];
            o[1] = x[1] + dt * x[3];
            o[2] = x[2] + dt * (dr * x[2] + gr * x[0]);
            o[3] = x[3] + dt * (dr * x[3] + gr * x[1]);
            o[4] = x[4];
            if (f->fid == F_REENTRY2D_BIAS_DYN) {
                o[5] = x[5];
                return 6;
            }
            return 5;
        }
        case F_RADAR2D_MEAS: { /* ssmod.py:1227-1252 */
            const double lx = f->npar >= 2 ? p[0] : 0.0, ly = f->npar >= 2 ? p[1] : 0.0;
            o[0] = sqrt((x[0] - lx) * (x[0] - lx) + (x[1] - ly) * (x[1] - ly));
            o[1] = atan2(x[1] - ly, x[0] - lx);
            return 2;
        }
        case F_CT_DYN: { /* ssmod.py:675-690 */
            const double dt = p[0], om = x[4];
            const double a = sin(om * dt), b = cos(om * dt), c = sin(om * dt) / om, d = (1 - cos(om * dt)) / om;
            o[0] = x[0] + c * x[1] - d * x[3];
            o[1] = b * x[1] - a * x[3];
            o[2] = d * x[1] + x[2] + c * x[3];
            o[3] = a * x[1] + b * x[3];
            o[4] = x[4];
            return 5;
        }
        case F_BEARING_MEAS: { /* ssmod.py:1189-1195 */
            const int ns = f->npar / 2;
            for (int s = 0; s < ns; ++s) o[s] = atan2(x[1] - p[2 * s + 1], x[0] - p[2 * s]);
            return ns;
        }
        case F_CTRS_DYN: { /* ssmod.py:755-774, input [x(5), q(2)] */
            const double dt = p[0], q0 = x[5], q1 = x[6];
            double f0, f1;
            if (x[4] == 0.0) {
                f0 = dt * x[2] * cos(x[3]);
                f1 = dt * x[2] * sin(x[3]);
            } else {
                const double c = x[2] / x[4];
                f0 = c * (sin(x[3] + x[4] * dt) - sin(x[3])) + 0.5 * dt * dt * cos(x[3]) * q0;
                f1 = c * (-cos(x[3] + x[4] * dt) + cos(x[3])) + 0.5 * dt * dt * sin(x[3]) * q0;
            }
            o[0] = x[0] + f0;
            o[1] = x[1] + f1;
            o[2] = x[2] + dt * q0;
            o[3] = x[3] + (dt * x[3] + 0.5 * dt * dt * q1);
            o[4] = x[4] + dt * q1;
            return 5;
        }
        case F_CV_DYN: /* ssmod.py:839-846 */
            o[0] = x[0] + p[0] * x[1];
            o[1] = x[1];
            o[2] = x[2] + p[0] * x[3];
            o[3] = x[3];
            return 4;
        case F_SMOOTH10D_DYN: /* this build's synthetic 10-D integrand; the reference has none */
            for (int i = 0; i < 5; ++i) {
                o[i] = sin(x[i]) + x[5 + i] * x[5 + i];
                o[5 + i] = x[5 + i] * cos(x[i]);
            }
            return 10;
        default: return 0;
    }
}

/*
 * One moment transform.  form 0: BQ (uncentred covariance + model variance), form 1: classical centred form with
 * diagonal covariance weights wc = Wc[0..N).  emv: (E, E) matrix; emv_broadcast 0 keeps its diagonal only
 * (I_out = eye(E)), 1 adds all of it.  tp_nu > 0: Student-t process scaling with iK.  Returns 1 if cov is not PD.
 */
int orc_apply(int form, int D, int E, int N, const orc_integrand *f, const double *mean, const double *cov, double t,
              const double *pts, const double *wm, const double *Wc, const double *Wcc, const double *emv,
              int emv_broadcast, double tp_nu, const double *iK, const double *cov_add, double *mf, double *cf,
              double *cfx) {
    double L[MAXD * MAXD];
    if (chol_lower(D, cov, L)) return 1;
    double *fx = (double *)malloc(sizeof(double) * (size_t)(E + D) * N);
    double *xs = fx + (size_t)E * N;
    for (int n = 0; n < N; ++n) {
        double x[MAXD], o[MAXD];
        memset(x, 0, sizeof(x));
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
            for (int k = 0; k <= d; ++k) s += L[d * D + k] * pts[k * N + n];
            x[d] = mean[d] + s;
            xs[d * N + n] = x[d];
        }
        eval_integrand(f, x, t, o);
        for (int e = 0; e < E; ++e) fx[e * N + n] = o[e];
    }
    for (int e = 0; e < E; ++e) {
        double s = 0.0;
        for (int n = 0; n < N; ++n) s += fx[e * N + n] * wm[n];
        mf[e] = s;
    }
    if (form == 0) {
        double *T = (double *)malloc(sizeof(double) * (size_t)E * N);
        double S[MAXD * MAXD];
        for (int pass = 0; pass < (tp_nu > 0.0 ? 2 : 1); ++pass) {
            const double *W = pass == 0 ? Wc : iK;
            for (int e = 0; e < E; ++e)
                for (int j = 0; j < N; ++j) {
                    double s = 0.0;
                    for (int i = 0; i < N; ++i) s += fx[e * N + i] * W[i * N + j];
                    T[e * N + j] = s;
                }
            for (int e = 0; e < E; ++e)
                for (int e2 = 0; e2 < E; ++e2) {
                    double s = 0.0;
                    for (int j = 0; j < N; ++j) s += T[e * N + j] * fx[e2 * N + j];
                    if (pass == 0) cf[e * E + e2] = s;
                    else S[e * E + e2] = s;
                }
        }
        for (int e = 0; e < E; ++e)
            for (int e2 = 0; e2 < E; ++e2) {
                double em = (e == e2 || emv_broadcast) ? (emv ? emv[e * E + e2] : 0.0) : 0.0;
                if (tp_nu > 0.0) em = (tp_nu - 2 + S[e * E + e2]) / (tp_nu - 2 + N) * em;
                cf[e * E + e2] = cf[e * E + e2] - mf[e] * mf[e2] + em;
                if (cov_add) cf[e * E + e2] += cov_add[e * E + e2];
            }
        for (int e = 0; e < E; ++e) {
            double g[MAXD];
            for (int d = 0; d < D; ++d) {
                double s = 0.0;
                for (int n = 0; n < N; ++n) s += fx[e * N + n] * Wcc[d * N + n];
                g[d] = s;
            }
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
                for (int d = 0; d <= j; ++d) s += g[d] * L[j * D + d];
                cfx[e * D + j] = s;
            }
        }
        free(T);
    } else {
        for (int e = 0; e < E; ++e)
            for (int n = 0; n < N; ++n) fx[e * N + n] -= mf[e];
        for (int e = 0; e < E; ++e)
            for (int e2 = 0; e2 < E; ++e2) {
                double s = 0.0;
                for (int n = 0; n < N; ++n) s += (fx[e * N + n] * Wc[n]) * fx[e2 * N + n];
                if (cov_add) s += cov_add[e * E + e2];
                cf[e * E + e2] = s;
            }
        for (int e = 0; e < E; ++e)
            for (int d = 0; d < D; ++d) {
                double s = 0.0;
                for (int n = 0; n < N; ++n) s += (fx[e * N + n] * Wc[n]) * (xs[d * N + n] - mean[d]);
                cfx[e * D + d] = s;
            }
    }
    free(fx);
    return 0;
}

/* B independent transforms, reference layout (trajectory-major); time per trajectory (stride 1) or shared (0).
 * Returns the number of non-PD items; status[b] set. */
int orc_apply_batch(int form, int D, int E, int N, const orc_integrand *f, int64_t B, const double *mean,
                    const double *cov, const double *time, int time_stride, const double *pts, const double *wm,
                    const double *Wc, const double *Wcc, const double *emv, int emv_broadcast, double tp_nu,
                    const double *iK, double *mf, double *cf, double *cfx, int32_t *status, int threads) {
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) reduction(+ : bad)
#endif
    for (int64_t b = 0; b < B; ++b) {
        const int st = orc_apply(form, D, E, N, f, mean + b * D, cov + b * D * D, time[time_stride ? b : 0], pts, wm,
                                 Wc, Wcc, emv, emv_broadcast, tp_nu, iK, NULL, mf + b * E, cf + b * E * E,
                                 cfx + b * E * D);
        if (status) status[b] = st;
        bad += st;
    }
    return bad;
}

/* Gaussian measurement update, ssinf.py:297-323.  Returns 1 if P_y is not PD. */
int orc_kalman_update(int D, int Y, const double *m_pr, const double *P_pr, const double *y_mean, const double *P_y,
                      const double *P_yx, const double *y, double *m_fi, double *P_fi) {
    double S[MAXD * MAXD], G[MAXD * MAXD], v[MAXD], w[MAXD];
    if (chol_lower(Y, P_y, S)) return 1;
    for (int d = 0; d < D; ++d) {
        for (int i = 0; i < Y; ++i) {
            double s = P_yx[i * D + d];
            for (int k = 0; k < i; ++k) s -= S[i * Y + k] * v[k];
            v[i] = s / S[i * Y + i];
        }
        for (int i = Y - 1; i >= 0; --i) {
            double s = v[i];
            for (int k = i + 1; k < Y; ++k) s -= S[k * Y + i] * v[k];
            v[i] = s / S[i * Y + i];
        }
        for (int i = 0; i < Y; ++i) G[d * Y + i] = v[i];
    }
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (int i = 0; i < Y; ++i) s += G[d * Y + i] * (y[i] - y_mean[i]);
        m_fi[d] = m_pr[d] + s;
    }
    for (int d = 0; d < D; ++d) {
        for (int j = 0; j < Y; ++j) {
            double s = 0.0;
            for (int i = 0; i < Y; ++i) s += G[d * Y + i] * P_y[i * Y + j];
            w[j] = s;
        }
        for (int d2 = 0; d2 < D; ++d2) {
            double s = 0.0;
            for (int j = 0; j < Y; ++j) s += w[j] * G[d2 * Y + j];
            P_fi[d * D + d2] = P_pr[d * D + d2] - s;
        }
    }
    return 0;
}

typedef struct {
    int form, D, E, N, emv_broadcast;
    double tp_nu;
    const double *pts, *wm, *Wc, *Wcc, *emv, *iK;
    orc_integrand f;
} orc_transform;

/*
 * Forward pass of an additive-noise Gaussian filter for B trajectories (ssinf.py:66-118, 254-323).
 * y [B][T][Y]; m0 [D], P0 [D*D] shared; outputs fm [B][T][D], fP [B][T][D*D]; status[b] = 0 or 1 + failing step.
 */
int orc_filter_forward(const orc_transform *dyn, const orc_transform *obs, int64_t B, int T, const double *y,
                       const double *m0, const double *P0, const double *GQG, const double *R, double *fm,
                       double *fP, int32_t *status, int threads) {
    const int D = dyn->D, Y = obs->E;
    int bad = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) reduction(+ : bad)
#endif
    for (int64_t b = 0; b < B; ++b) {
        double m[MAXD], P[MAXD * MAXD], m_pr[MAXD], P_pr[MAXD * MAXD], C[MAXD * MAXD], ym[MAXD], Py[MAXD * MAXD],
            Pyx[MAXD * MAXD];
        memcpy(m, m0, sizeof(double) * D);
        memcpy(P, P0, sizeof(double) * D * D);
        int st = 0;
        for (int k = 0; k < T && !st; ++k) {
            /* both transforms of step k+1 use time index k (ssinf.py:104, 276-288) */
            if (orc_apply(dyn->form, D, D, dyn->N, &dyn->f, m, P, (double)k, dyn->pts, dyn->wm, dyn->Wc, dyn->Wcc,
                          dyn->emv, dyn->emv_broadcast, dyn->tp_nu, dyn->iK, GQG, m_pr, P_pr, C) ||
                orc_apply(obs->form, D, Y, obs->N, &obs->f, m_pr, P_pr, (double)k, obs->pts, obs->wm, obs->Wc,
                          obs->Wcc, obs->emv, obs->emv_broadcast, obs->tp_nu, obs->iK, R, ym, Py, Pyx) ||
                orc_kalman_update(D, Y, m_pr, P_pr, ym, Py, Pyx, y + (b * T + k) * Y, m, P)) {
                st = k + 1;
                break;
            }
            memcpy(fm + (b * T + k) * D, m, sizeof(double) * D);
            memcpy(fP + (b * T + k) * D * D, P, sizeof(double) * D * D);
        }
        if (status) status[b] = st;
        bad += st != 0;
    }
    return bad;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
