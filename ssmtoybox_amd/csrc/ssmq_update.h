// Per-trajectory bodies of the measurement update and of the Gaussian log-density, shared by their batched kernels
// (ssmq_filter.hip: one trajectory per lane) and by the theta-batched step's chained kernel (ssmq_apply_wide.hip:
// k_theta_chain, where the lane that finished a trajectory's two transforms goes on with its update).  One definition,
// so both routes round identically.
#pragma once
#include "ssmq_host.h"

namespace ssmq {

struct UpdArgs {
    const double *m_pr, *P_pr, *y_mean, *P_y, *P_yx, *y;
    double *m_fi, *P_fi;
    int32_t *status;            // aggregated: 0 ok, else 1 + first failing step
    const int32_t *st_a, *st_b; // per-step status of the two transforms (may be null)
    int64_t B, ld;
    int32_t step, D, Y;
    // Studentian update (ssinf.py:700-736): student_dof > 0 -> the inputs are scale matrices, and besides the filtered
    // "covariance" P_fi the rescaled scale matrix (dof + delta'delta) / (dof + Y) * P_fi is written to smat_out
    double student_dof;
    double *smat_out;
    int32_t Dx;   // columns of P_yx (> D when the measurement transform ran on a noise-augmented state: the first D are used)
};

template <int D, int Y>
__device__ __forceinline__ void kalman_update_item(const UpdArgs &a, const uint32_t b) {
    const int64_t ld = a.ld;
    double S[Y * (Y + 1) / 2];
#pragma unroll
    for (int i = 0; i < Y; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) S[SSMQ_PK(i, j)] = a.P_y[(i * Y + j) * ld + b];
    double Py[Y][Y];
#pragma unroll
    for (int i = 0; i < Y; ++i)
#pragma unroll
        for (int j = 0; j < Y; ++j) Py[i][j] = a.P_y[(i * Y + j) * ld + b];
    bool ok;
    // X = P_y^-1 P_yx, column by column (forward then backward substitution); gain[d][i] = X[i][d]
    double G[D][Y];
    if (Y == 1) {
        // scalar measurement: one division instead of factor + two substitutions (same shortcut as k_filter_fused)
        ok = S[0] > 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) G[d][0] = div_nr(a.P_yx[d * ld + b], S[0]);   // (Y == 1: row 0 only)
    } else {
        ok = chol_packed<Y>(S);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double v[Y];
#pragma unroll
            for (int i = 0; i < Y; ++i) {
                double s = a.P_yx[(i * a.Dx + d) * ld + b];
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
    for (int i = 0; i < Y; ++i) dy[i] = a.y[i * ld + b] - a.y_mean[i * ld + b];
    int32_t agg = a.status[b];
    int32_t bad = ok ? 0 : 1;
    if (a.st_a) bad |= a.st_a[b];
    if (a.st_b) bad |= a.st_b[b];
    if (agg == 0 && bad) agg = a.step + 1;
    a.status[b] = agg;
    const double nan = __builtin_nan("");
    const bool good = (agg == 0);
    double sc2 = 1.0;
    if (a.student_dof > 0.0) {
        // delta = chol(P_y)^-1 (y - y_mean)   (ssinf.py:729-731)
        double dl[Y], dd = 0.0;
        if (Y == 1) {
            dd = div_nr(dy[0] * dy[0], S[0]);
        } else {
#pragma unroll
            for (int i = 0; i < Y; ++i) {
                double s = dy[i];
#pragma unroll
                for (int k = 0; k < i; ++k) s -= S[SSMQ_PK(i, k)] * dl[k];
                dl[i] = div_nr(s, S[SSMQ_PK(i, i)]);
                dd += dl[i] * dl[i];
            }
        }
        sc2 = (a.student_dof + dd) / (a.student_dof + (double)Y);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < Y; ++i) s += G[d][i] * dy[i];
        const double mp = a.m_pr[d * ld + b];
        a.m_fi[d * ld + b] = good ? mp + s : nan;
    }
    // W = gain P_y (D x Y);  P = P_pr - W gain'
#pragma unroll
    for (int d = 0; d < D; ++d) {
        double w[Y];
#pragma unroll
        for (int j = 0; j < Y; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < Y; ++i) s += G[d][i] * Py[i][j];
            w[j] = s;
        }
#pragma unroll
        for (int d2 = 0; d2 < D; ++d2) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < Y; ++j) s += w[j] * G[d2][j];
            const double pp = a.P_pr[(d * D + d2) * ld + b];
            const double pf = good ? pp - s : nan;
            a.P_fi[(d * D + d2) * ld + b] = pf;
            if (a.student_dof > 0.0) a.smat_out[(d * D + d2) * ld + b] = sc2 * pf;
        }
    }
}

// Run-time-shape fallback (D, Y <= SSMQ_MAX_DIM); private arrays live in scratch.
__device__ inline void kalman_update_item_generic(const UpdArgs &a, const uint32_t b) {
    const int64_t ld = a.ld;
    const int D = a.D, Y = a.Y;
    double S[SSMQ_MAX_DIM * SSMQ_MAX_DIM], G[SSMQ_MAX_DIM * SSMQ_MAX_DIM], v[SSMQ_MAX_DIM], w[SSMQ_MAX_DIM];
    for (int i = 0; i < Y; ++i)
        for (int j = 0; j < Y; ++j) S[i * Y + j] = a.P_y[(i * Y + j) * ld + b];
    bool ok = true;
    for (int j = 0; j < Y; ++j) {
        double ajj = S[j * Y + j];
        for (int k = 0; k < j; ++k) ajj -= S[j * Y + k] * S[j * Y + k];
        ok = ok && (ajj > 0.0);
        ajj = sqrt(ajj);
        S[j * Y + j] = ajj;
        const double r = 1.0 / ajj;
        for (int i = j + 1; i < Y; ++i) {
            double s = S[i * Y + j];
            for (int k = 0; k < j; ++k) s -= S[i * Y + k] * S[j * Y + k];
            S[i * Y + j] = s * r;
        }
    }
    for (int d = 0; d < D; ++d) {
        for (int i = 0; i < Y; ++i) {
            double s = a.P_yx[(i * a.Dx + d) * ld + b];
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
    int32_t agg = a.status[b];
    int32_t bad = ok ? 0 : 1;
    if (a.st_a) bad |= a.st_a[b];
    if (a.st_b) bad |= a.st_b[b];
    if (agg == 0 && bad) agg = a.step + 1;
    a.status[b] = agg;
    const double nan = __builtin_nan("");
    const bool good = (agg == 0);
    double sc2 = 1.0;
    if (a.student_dof > 0.0) {
        double dd = 0.0;
        for (int i = 0; i < Y; ++i) {
            double s = a.y[i * ld + b] - a.y_mean[i * ld + b];
            for (int k = 0; k < i; ++k) s -= S[i * Y + k] * v[k];
            v[i] = s / S[i * Y + i];
            dd += v[i] * v[i];
        }
        sc2 = (a.student_dof + dd) / (a.student_dof + (double)Y);
    }
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (int i = 0; i < Y; ++i) s += G[d * Y + i] * (a.y[i * ld + b] - a.y_mean[i * ld + b]);
        a.m_fi[d * ld + b] = good ? a.m_pr[d * ld + b] + s : nan;
    }
    for (int d = 0; d < D; ++d) {
        for (int j = 0; j < Y; ++j) {
            double s = 0.0;
            for (int i = 0; i < Y; ++i) s += G[d * Y + i] * a.P_y[(i * Y + j) * ld + b];
            w[j] = s;
        }
        for (int d2 = 0; d2 < D; ++d2) {
            double s = 0.0;
            for (int j = 0; j < Y; ++j) s += w[j] * G[d2 * Y + j];
            const double pf = good ? a.P_pr[(d * D + d2) * ld + b] - s : nan;
            a.P_fi[(d * D + d2) * ld + b] = pf;
            if (a.student_dof > 0.0) a.smat_out[(d * D + d2) * ld + b] = sc2 * pf;
        }
    }
}

__device__ inline void gauss_logpdf_item(const double *y, const double *y_mean, const double *P_y, double *out, int Y,
                                         int64_t ld, const int32_t *merge, int32_t *merge_out, const uint32_t b) {
    if (merge)
        merge_out[b] = (merge[b] ? 1 : 0) | (merge[ld + b] ? 2 : 0) | (merge[2 * ld + b] ? 4 : 0) |
                       (merge[3 * ld + b] ? 8 : 0) | (merge[4 * ld + b] ? 16 : 0);
    double S[SSMQ_MAX_DIM * SSMQ_MAX_DIM], v[SSMQ_MAX_DIM];
    for (int i = 0; i < Y; ++i)
        for (int j = 0; j <= i; ++j) S[i * Y + j] = P_y[((int64_t)i * Y + j) * ld + b];
    bool ok = true;
    double logdet = 0.0, q = 0.0;
    for (int j = 0; j < Y; ++j) {
        double ajj = S[j * Y + j];
        for (int k = 0; k < j; ++k) ajj -= S[j * Y + k] * S[j * Y + k];
        ok = ok && (ajj > 0.0);
        ajj = sqrt(ajj);
        S[j * Y + j] = ajj;
        logdet += log(ajj);
        const double r = 1.0 / ajj;
        for (int i = j + 1; i < Y; ++i) {
            double s = S[i * Y + j];
            for (int k = 0; k < j; ++k) s -= S[i * Y + k] * S[j * Y + k];
            S[i * Y + j] = s * r;
        }
    }
    for (int i = 0; i < Y; ++i) {
        double s = y[(int64_t)i * ld + b] - y_mean[(int64_t)i * ld + b];
        for (int k = 0; k < i; ++k) s -= S[i * Y + k] * v[k];
        v[i] = s / S[i * Y + i];
        q += v[i] * v[i];
    }
    out[b] = ok ? -0.5 * (q + 2.0 * logdet + Y * 1.8378770664093453) : __builtin_nan("");
}

}  // namespace ssmq
