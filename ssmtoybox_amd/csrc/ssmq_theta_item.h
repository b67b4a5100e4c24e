// Device functions of the per-item theta step (one parameter item per lane, everything in registers): shared by k_theta_item
// (ssmq_theta_item.hip) and the one-launch marginalised filter (ssmq_marginal.hip: k_mg_persistent).  See ssmq_theta_item.hip.
#pragma once
#include "ssmq_device.h"

namespace ssmq {
namespace theta_item {

// sum of v[0 .. N-1] in the order block_sum's butterfly adds the lanes 0 .. N-1 of a wave (the rest zero), N <= 8
template <int N>
__device__ __forceinline__ double butterfly_sum(const double (&v)[N]) {
    double l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) l[i] = i < N ? v[i < N ? i : 0] : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) l[i] += l[i + 4];
#pragma unroll
    for (int i = 0; i < 2; ++i) l[i] += l[i + 2];
    return l[0] + l[1];
}

template <int DI, int N>
struct GpWeights {
    double wm[N], Wc[N][N], Wcc[DI][N], mv;
    bool pd;
};

// GP quadrature weights of one parameter row (weights_body with NB = 0, use_lds = 1: same operations, same order)
template <int DI, int N>
__device__ __forceinline__ void gp_weights_item(const double *xi, const double *par, double jitter, GpWeights<DI, N> &w) {
    const double alpha = par[0];
    double sil[DI], zs[DI][N], nrm[N], x[DI][N];
#pragma unroll
    for (int d = 0; d < DI; ++d) {
        sil[d] = 1.0 / par[1 + d];                       // par[1:] ** -1   (bq/bqkern.py:454)
#pragma unroll
        for (int n = 0; n < N; ++n) {
            x[d][n] = xi[d * N + n];
            zs[d][n] = sil[d] * x[d][n];
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DI; ++d) s += zs[d][n] * zs[d][n];
        nrm[n] = s;
    }
    // kernel matrix, scaling=False (alpha = 1): exp(2 log(1) - maha / 2), + jitter I
    double A[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double dot = 0.0;
#pragma unroll
            for (int d = 0; d < DI; ++d) dot += zs[d][i] * zs[d][j];
            const double mh = (nrm[i] + nrm[j]) - 2.0 * dot;
            A[i][j] = exp(0.0 - 0.5 * mh) + (i == j ? jitter : 0.0);
        }
    // chol_block: right-looking, lower triangle
    bool pd = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double p = A[k][k];
        if (!(p > 0.0)) pd = false;
        A[k][k] = sqrt(p);
        const double r = 1.0 / A[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) A[i][k] *= r;
#pragma unroll
        for (int i = k + 1; i < N; ++i)
#pragma unroll
            for (int j = k + 1; j <= i; ++j) A[i][j] -= A[i][k] * A[j][k];
    }
    w.pd = pd;
    // chol_inverse: (L L')^-1 column by column, forward then backward substitution
    double X[N][N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = c; k < i; ++k) s -= A[i][k] * X[k][c];
            X[i][c] = (i < c) ? 0.0 : s / A[i][i];
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            double s = X[i][c];
#pragma unroll
            for (int k = i + 1; k < N; ++k) s -= A[k][i] * X[k][c];
            X[i][c] = s / A[i][i];
        }
    }
    double iK[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) iK[i][j] = 0.5 * (X[i][j] + X[j][i]);
    // Gaussian expectations of the kernel
    double cq = 1.0, cQ = 1.0;
#pragma unroll
    for (int d = 0; d < DI; ++d) {
        const double il = sil[d] * sil[d];
        cq *= il + 1.0;
        cQ *= il + il + 1.0;
    }
    cq = 1.0 / sqrt(cq);   // det(Lam^-1 + I) ** -0.5
    cQ = 1.0 / sqrt(cQ);   // det(2 Lam^-1 + I) ** -0.5
    double q[N], R[DI][N], Q[N][N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DI; ++d) {
            const double il = sil[d] * sil[d];
            const double lam = 1.0 / il;
            s += x[d][n] * ((1.0 / (lam + 1.0)) * x[d][n]);
        }
        q[n] = cq * exp(-0.5 * s);
    }
#pragma unroll
    for (int d = 0; d < DI; ++d) {
        const double lam = 1.0 / (sil[d] * sil[d]);
#pragma unroll
        for (int n = 0; n < N; ++n) R[d][n] = q[n] * ((1.0 / (lam + 1.0)) * x[d][n]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            // xi_i + xi_j + maha(Lam^-1 x_i, -Lam^-1 x_j; (2 Lam^-1 + I)^-1) / 2
            double m2i = 0.0, m2j = 0.0, mij = 0.0;
#pragma unroll
            for (int d = 0; d < DI; ++d) {
                const double il = sil[d] * sil[d];
                const double v = 1.0 / (il + il + 1.0);
                const double yi = il * x[d][i], yj = -(il * x[d][j]);
                m2i += (yi * v) * yi;
                m2j += (yj * v) * yj;
                mij += (yi * v) * yj;
            }
            const double mh = (m2i + m2j) - 2.0 * mij;
            const double e = ((0.0 - 0.5 * nrm[i]) + (0.0 - 0.5 * nrm[j])) + 0.5 * mh;
            Q[i][j] = cQ * exp(e);
        }
    // GP weights (bq/bqmod.py:495-523): wm = q iK, Wcc = R iK, Wc = sym(iK Q iK), model_var = alpha^2 (1 - tr(Q iK))
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) s += q[k] * iK[k][j];
        w.wm[j] = s;
    }
#pragma unroll
    for (int d = 0; d < DI; ++d)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s += R[d][k] * iK[k][j];
            w.Wcc[d][j] = s;
        }
    double M1[N][N], M2[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s += Q[i][k] * iK[k][j];
            M1[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) s += iK[i][k] * M1[k][j];
            M2[i][j] = s;
        }
    double dg[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        dg[i] = M1[i][i];
#pragma unroll
        for (int j = 0; j < N; ++j) w.Wc[i][j] = 0.5 * (M2[i][j] + M2[j][i]);
    }
    w.mv = (alpha * alpha) * (1.0 - butterfly_sum<N>(dg));
    if (!pd) {        // weights_body poisons every output of a kernel matrix that is not positive definite
        const double nan = __builtin_nan("");
        w.mv = nan;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            w.wm[i] = nan;
#pragma unroll
            for (int j = 0; j < N; ++j) w.Wc[i][j] = nan;
#pragma unroll
            for (int d = 0; d < DI; ++d) w.Wcc[d][i] = nan;
        }
    }
}

// BQ moment transform of one item with its own weights (apply_wave_body, BQ form, tp_nu = 0, cov_scale = ccov_scale = 1).
// covf: the input covariance, full row-major [DI][DI] (lower triangle read).  Returns the Cholesky flag; outputs are NaN where
// that kernel writes NaN.
template <int DI, int E, int N, bool CCOV>
__device__ __forceinline__ bool bq_transform_item(int fid, const FPar &fp, double t, const double (&m)[DI], const double (&covf)[DI][DI],
                                                  const double *xi, const GpWeights<DI, N> &w, int emv_mode, const double *cov_add,
                                                  double (&mf)[E], double (&cf)[E][E], double (&cfx)[E][DI]) {
    double sL[DI][DI];
#pragma unroll
    for (int r = 0; r < DI; ++r)
#pragma unroll
        for (int c = 0; c < DI; ++c) sL[r][c] = (c <= r) ? covf[r][c] : 0.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < DI; ++j) {
        const double ajj = sL[j][j];
        ok = ok && (ajj > 0.0);
        const double ljj = sqrt(ajj), r = 1.0 / ljj;
        sL[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < DI; ++i) sL[i][j] *= r;
#pragma unroll
        for (int i = j + 1; i < DI; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) sL[i][k] -= sL[i][j] * sL[k][j];
    }
    double sfx[E][N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double x[DI], o[SSMQ_MAX_DIM], xs[kMaxIntegrandIn];
#pragma unroll
        for (int d = 0; d < DI; ++d) {
            double sacc = m[d];
#pragma unroll
            for (int k = 0; k <= d; ++k) sacc += sL[d][k] * xi[k * N + n];
            x[d] = sacc;
        }
#pragma unroll
        for (int k = 0; k < SSMQ_MAX_DIM; ++k) o[k] = 0.0;
#pragma unroll
        for (int k = 0; k < kMaxIntegrandIn; ++k) {
            double v = k < DI ? x[k < DI ? k : 0] : 0.0;
            if (fp.n_idx > 0) {                    // state-index selection (MeasurementModel.state_index)
                const int src = k < fp.n_idx ? fp.idx[k] : 0;
                v = x[0];
#pragma unroll
                for (int qq = 1; qq < DI; ++qq) v = (src == qq) ? x[qq] : v;
            }
            xs[k] = v;
        }
        eval_integrand(fid, xs, t, fp, o);
#pragma unroll
        for (int e = 0; e < E; ++e) sfx[e][n] = o[e];
    }
    const double nan = __builtin_nan("");
    double smf[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double sacc = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n) sacc += sfx[e][n] * w.wm[n];
        smf[e] = sacc;
        mf[e] = ok ? sacc : nan;
    }
    double sA[E][N];       // fx Wc
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += sfx[e][i] * w.Wc[i][j];
#pragma unroll
        for (int e = 0; e < E; ++e) sA[e][j] = acc[e];
    }
    double sC[E][E];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int e2 = 0; e2 <= e; ++e2) {
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) sacc += sA[e][j] * sfx[e2][j];
            sC[e][e2] = sacc;
            sC[e2][e] = sacc;
        }
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int e2 = 0; e2 < E; ++e2) {
            const bool use = (e == e2) || (emv_mode == SSMQ_EMV_BROADCAST);
            const double em = use ? w.mv : 0.0;
            double v = (sC[e][e2] - smf[e] * smf[e2] + em) * 1.0;
            if (cov_add) v += cov_add[e * E + e2];
            cf[e][e2] = ok ? v : nan;
        }
    if (CCOV) {
        double sg[E][DI];
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int d = 0; d < DI; ++d) {
                double sacc = 0.0;
#pragma unroll
                for (int n = 0; n < N; ++n) sacc += sfx[e][n] * w.Wcc[d][n];
                sg[e][d] = sacc;
            }
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int j = 0; j < DI; ++j) {
                double sacc = 0.0;
#pragma unroll
                for (int d = 0; d <= j; ++d) sacc += sg[e][d] * sL[j][d];
                cfx[e][j] = ok ? sacc * 1.0 : nan;
            }
    }
    return ok;
}

// The whole step of ONE item: weights of the dynamics transform, the transform, + G Q G', the same for the measurement model, + R,
// the Kalman update and the log-likelihood.  par_dyn [1 + DIN], par_obs [1 + D]: kernel parameters; m / cv: the (augmented) state
// moments; yv: the measurement.  Returns the merged flags: 1 weights(dyn) | 2 weights(obs) | 4 transform(dyn) | 8 transform(obs) |
// 16 update.
template <int DIN, int D, int Y, int ND, int NO>
__device__ __forceinline__ int32_t theta_item_core(int fid_dyn, int fid_obs, const FPar &fpd, const FPar &fpo, int emv_dyn, int emv_obs,
                                                   const double *xid, const double *xio, const double *par_dyn, const double *par_obs,
                                                   const double (&m)[DIN], const double (&cv)[DIN][DIN], const double (&yv)[Y], double t,
                                                   const double *gq, const double *rr, double jitter, double (&m_fi)[D],
                                                   double (&P_fi)[D][D], double &ll) {
    const double nan = __builtin_nan("");
    // ---- time update: weights of the dynamics transform, the transform, + G Q G' ----------------------------------------------
    GpWeights<DIN, ND> wd;
    gp_weights_item<DIN, ND>(xid, par_dyn, jitter, wd);
    double m_pr[D], P_pr[D][D], c_unused[D][DIN];
    const bool ok_td = bq_transform_item<DIN, D, ND, false>(fid_dyn, fpd, t, m, cv, xid, wd, emv_dyn, gq, m_pr, P_pr, c_unused);
    // ---- predictive measurement moments, + R --------------------------------------------------------------------------------
    GpWeights<D, NO> wo;
    gp_weights_item<D, NO>(xio, par_obs, jitter, wo);
    double y_mean[Y], P_y[Y][Y], P_yx[Y][D];
    const bool ok_to = bq_transform_item<D, Y, NO, true>(fid_obs, fpo, t, m_pr, P_pr, xio, wo, emv_obs, rr, y_mean, P_y, P_yx);
    // ---- measurement update (kalman_update_item<D, Y>) --------------------------------------------------------------------------
    double S[Y * (Y + 1) / 2];
#pragma unroll
    for (int i = 0; i < Y; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) S[SSMQ_PK(i, j)] = P_y[i][j];
    bool ok_up;
    double G[D][Y];
    if (Y == 1) {
        ok_up = S[0] > 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) G[d][0] = div_nr(P_yx[0][d], S[0]);
    } else {
        ok_up = chol_packed<Y>(S);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double v[Y];
#pragma unroll
            for (int i = 0; i < Y; ++i) {
                double s = P_yx[i][d];
#pragma unroll
                for (int k = 0; k < i; ++k) s -= S[SSMQ_PK(i, k)] * v[k];
                v[i] = div_nr(s, S[SSMQ_PK(i, i)]);
            }
#pragma unroll
            for (int i = Y - 1; i >= 0; --i) {
                double s = v[i];
#pragma unroll
                for (int k = i + 1; k < Y; ++k) s -= S[SSMQ_PK(k, i)] * v[k];
                v[i] = div_nr(s, S[SSMQ_PK(i, i)]);
            }
#pragma unroll
            for (int i = 0; i < Y; ++i) G[d][i] = v[i];
        }
    }
    double dy[Y];
#pragma unroll
    for (int i = 0; i < Y; ++i) dy[i] = yv[i] - y_mean[i];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < Y; ++i) s += G[d][i] * dy[i];
        m_fi[d] = ok_up ? m_pr[d] + s : nan;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        double wv[Y];
#pragma unroll
        for (int j = 0; j < Y; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < Y; ++i) s += G[d][i] * P_y[i][j];
            wv[j] = s;
        }
#pragma unroll
        for (int d2 = 0; d2 < D; ++d2) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < Y; ++j) s += wv[j] * G[d2][j];
            P_fi[d][d2] = ok_up ? P_pr[d][d2] - s : nan;
        }
    }
    // ---- log N(y | y_mean, P_y) (gauss_logpdf_item) ---------------------------------------------------------------------------------
    {
        double L[Y][Y], v[Y];
#pragma unroll
        for (int i = 0; i < Y; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) L[i][j] = P_y[i][j];
        bool okl = true;
        double logdet = 0.0, q = 0.0;
#pragma unroll
        for (int j = 0; j < Y; ++j) {
            double ajj = L[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) ajj -= L[j][k] * L[j][k];
            okl = okl && (ajj > 0.0);
            ajj = sqrt(ajj);
            L[j][j] = ajj;
            logdet += log(ajj);
            const double r = 1.0 / ajj;
#pragma unroll
            for (int i = j + 1; i < Y; ++i) {
                double s = L[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
                L[i][j] = s * r;
            }
        }
#pragma unroll
        for (int i = 0; i < Y; ++i) {
            double s = yv[i] - y_mean[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s -= L[i][k] * v[k];
            v[i] = s / L[i][i];
            q += v[i] * v[i];
        }
        ll = okl ? -0.5 * (q + 2.0 * logdet + Y * 1.8378770664093453) : nan;
    }
    return (wd.pd ? 0 : 1) | (wo.pd ? 0 : 2) | (ok_td ? 0 : 4) | (ok_to ? 0 : 8) | (ok_up ? 0 : 16);
}

}  // namespace theta_item
}  // namespace ssmq
