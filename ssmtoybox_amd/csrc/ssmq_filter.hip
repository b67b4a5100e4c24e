// Gaussian measurement update, batched, one trajectory per lane (ssinf.py:297-323):
//   gain = (P_y^-1 P_yx)'  via Cholesky of P_y;  m = m_pr + gain (y - y_mean);  P = P_pr - (gain P_y) gain'
// The covariance is left unsymmetrised exactly as the reference leaves it (ssinf.py:323); the next time update reads
// only its lower triangle (LAPACK 'L').  SoA planes, pitch ld.
#include "ssmq_host.h"
#include "ssmq_update.h"

namespace ssmq {

constexpr int kUpdBlock = 64;

template <int D, int Y>
__global__ __launch_bounds__(kUpdBlock) void k_kalman_update(const UpdArgs a) {
    const uint32_t b = blockIdx.x * kUpdBlock + threadIdx.x;
    if ((int64_t)b >= a.B) return;
    kalman_update_item<D, Y>(a, b);
}

// Run-time-shape fallback (D, Y <= SSMQ_MAX_DIM); private arrays live in scratch.
__global__ __launch_bounds__(kUpdBlock) void k_kalman_update_generic(const UpdArgs a) {
    const uint32_t b = blockIdx.x * kUpdBlock + threadIdx.x;
    if ((int64_t)b >= a.B) return;
    kalman_update_item_generic(a, b);
}

template <int D, int Y>
static void launch_upd(const UpdArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + kUpdBlock - 1) / kUpdBlock);
    hipLaunchKernelGGL((k_kalman_update<D, Y>), dim3(grid), dim3(kUpdBlock), 0, s, a);
}

int launch_kalman_update_ex(int D, int Y, int64_t B, int64_t ld, const double *m_pr, const double *P_pr,
                            const double *y_mean, const double *P_y, const double *P_yx, const double *y,
                            double *m_fi, double *P_fi, int32_t *status, const int32_t *st_a, const int32_t *st_b,
                            int step, hipStream_t s, double student_dof, double *smat_out, int Dx) {
    UpdArgs a{m_pr, P_pr, y_mean, P_y, P_yx, y, m_fi, P_fi, status, st_a, st_b, B, ld, step, D, Y, student_dof, smat_out,
              Dx > 0 ? Dx : D};
#define SSMQ_UPD(d, y_)                  \
    if (D == d && Y == y_) {             \
        launch_upd<d, y_>(a, s);         \
        return hip_fail(hipGetLastError(), "k_kalman_update"); \
    }
    SSMQ_UPD(1, 1)
    SSMQ_UPD(2, 1)
    SSMQ_UPD(2, 2)
    SSMQ_UPD(3, 1)
    SSMQ_UPD(4, 2)
    SSMQ_UPD(5, 2)
    SSMQ_UPD(5, 4)
    SSMQ_UPD(6, 2)
#undef SSMQ_UPD
    if (D > SSMQ_MAX_DIM || Y > SSMQ_MAX_DIM) {
        set_error("kalman update: D or Y above SSMQ_MAX_DIM");
        return SSMQ_E_UNSUPPORTED;
    }
    const unsigned grid = (unsigned)((B + kUpdBlock - 1) / kUpdBlock);
    hipLaunchKernelGGL(k_kalman_update_generic, dim3(grid), dim3(kUpdBlock), 0, s, a);
    return hip_fail(hipGetLastError(), "k_kalman_update_generic");
}

int launch_kalman_update(int D, int Y, int64_t B, int64_t ld, const double *m_pr, const double *P_pr,
                         const double *y_mean, const double *P_y, const double *P_yx, const double *y, double *m_fi,
                         double *P_fi, int32_t *status, hipStream_t s) {
    return launch_kalman_update_ex(D, Y, B, ld, m_pr, P_pr, y_mean, P_y, P_yx, y, m_fi, P_fi, status, nullptr, nullptr,
                                   0, s, 0.0, nullptr, 0);
}


// ---- log N(y | y_mean, P_y) per trajectory (scipy multivariate_normal.logpdf as used at ssinf.py:1198) ----------------
// Cholesky route: -(delta' P_y^-1 delta + log det P_y + Y log 2 pi) / 2; NaN where P_y is not positive definite.
// merge: five status vectors of the theta-batched step [w_dyn | w_obs | t_dyn | t_obs | upd], pitch ld, folded into
// bit flags at merge_out (ssmq_gp_theta_step) in the same launch - or null
__global__ __launch_bounds__(kUpdBlock) void k_gauss_logpdf(const double *y, const double *y_mean, const double *P_y,
                                                            double *out, int Y, int64_t B, int64_t ld,
                                                            const int32_t *merge, int32_t *merge_out) {
    const uint32_t b = blockIdx.x * kUpdBlock + threadIdx.x;
    if ((int64_t)b >= B) return;
    gauss_logpdf_item(y, y_mean, P_y, out, Y, ld, merge, merge_out, b);
}

int launch_gauss_logpdf(int Y, int64_t B, int64_t ld, const double *y, const double *y_mean, const double *P_y,
                        double *out, hipStream_t s, const int32_t *merge, int32_t *merge_out) {
    if (Y < 1 || Y > SSMQ_MAX_DIM) {
        set_error("logpdf: Y out of range");
        return SSMQ_E_ARG;
    }
    const unsigned grid = (unsigned)((B + kUpdBlock - 1) / kUpdBlock);
    hipLaunchKernelGGL(k_gauss_logpdf, dim3(grid), dim3(kUpdBlock), 0, s, y, y_mean, P_y, out, Y, B, ld, merge, merge_out);
    return hip_fail(hipGetLastError(), "k_gauss_logpdf");
}

// ---- noise augmentation for non-additive models (ssinf.py:271-272, 282-283) -----------------------------------------
// [m; noise_mean], blockdiag(P, noise_cov): plain copy kernel, one trajectory per lane, planes written whole.
__global__ __launch_bounds__(kUpdBlock) void k_augment(const double *m, const double *P, const double *nmean,
                                                       const double *ncov, double *ma, double *Pa, int D, int Dn,
                                                       int64_t B, int64_t ld) {
    const uint32_t b = blockIdx.x * kUpdBlock + threadIdx.x;
    if ((int64_t)b >= B) return;
    const int Da = D + Dn;
    for (int i = 0; i < Da; ++i) ma[(int64_t)i * ld + b] = i < D ? m[(int64_t)i * ld + b] : nmean[i - D];
    for (int i = 0; i < Da; ++i)
        for (int j = 0; j < Da; ++j) {
            double v = 0.0;
            if (i < D && j < D) v = P[((int64_t)i * D + j) * ld + b];
            else if (i >= D && j >= D) v = ncov[(i - D) * Dn + (j - D)];
            Pa[((int64_t)i * Da + j) * ld + b] = v;
        }
}

int launch_augment(const double *m, const double *P, const double *nmean, const double *ncov, double *ma, double *Pa,
                   int D, int Dn, int64_t B, int64_t ld, hipStream_t s) {
    const unsigned grid = (unsigned)((B + kUpdBlock - 1) / kUpdBlock);
    hipLaunchKernelGGL(k_augment, dim3(grid), dim3(kUpdBlock), 0, s, m, P, nmean, ncov, ma, Pa, D, Dn, B, ld);
    return hip_fail(hipGetLastError(), "k_augment");
}

// ---- Rauch-Tung-Striebel backward pass (ssinf.py:120-147, 325-344) ---------------------------------------------------
// One trajectory per lane, run-time loop over time.  Index convention: arrays hold steps 1..T of the reference's
// 0..T arrays (element t here = index t + 1 there).  The reference's loop `for k in range(N - 2, 0, -1)` starts from the
// filtered estimate at index N and pairs it with the predictive moments of index N - 1; the
// smoothed arrays therefore keep the filtered values at the last two steps.  Reproduced as is.
struct RtsArgs {
    const double *fm, *fP;       // filtered   [T][D][ld], [T][D*D][ld]
    const double *pm, *pP, *pC;  // predictive mean, covariance, cross-covariance of the dynamics transform
    double *sm, *sP;             // smoothed
    int32_t *status;             // |= (1 << 30) if a predictive covariance is not PD
    int64_t B, ld;
    int32_t T;
    int32_t c_cols;              // columns stored per row of pC (D; D + dim_noise for a model that takes its noise as an
                                 // argument - the smoother uses the state columns only, ssinf.py:294-295)
};

// every element of the five input sequences is read exactly once and the outputs are written once: streaming accesses
#define RTS_LD(src) __builtin_nontemporal_load(&(src))
template <int D>
__global__ __launch_bounds__(kUpdBlock) void k_rts_backward(const RtsArgs a) {
    const uint32_t b = blockIdx.x * kUpdBlock + threadIdx.x;
    if ((int64_t)b >= a.B) return;
    const int64_t ld = a.ld;
    const int T = a.T;
    double ms[D], Ps[D][D];
    // smoothed = filtered at the last two steps; the recursion starts from the last filtered estimate
    for (int t = T - 1; t >= 0 && t >= T - 2; --t) {
#pragma unroll
        for (int d = 0; d < D; ++d) a.sm[((int64_t)t * D + d) * ld + b] = a.fm[((int64_t)t * D + d) * ld + b];
#pragma unroll
        for (int i = 0; i < D * D; ++i) a.sP[((int64_t)t * D * D + i) * ld + b] = a.fP[((int64_t)t * D * D + i) * ld + b];
    }
    if (T < 1) return;
#pragma unroll
    for (int d = 0; d < D; ++d) ms[d] = a.fm[((int64_t)(T - 1) * D + d) * ld + b];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Ps[i][j] = a.fP[((int64_t)(T - 1) * D * D + i * D + j) * ld + b];
    bool allok = true;
#pragma unroll 1
    for (int k = T - 2; k >= 1; --k) {
        // predictive moments of reference index k + 1 = element k here; filtered moments of index k = element k - 1
        double S[D * (D + 1) / 2], mp[D], Pp[D][D];
#pragma unroll
        for (int d = 0; d < D; ++d) mp[d] = RTS_LD(a.pm[((int64_t)k * D + d) * ld + b]);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {     // the forward pass wrote a symmetric matrix: lower triangle, mirrored
                Pp[i][j] = RTS_LD(a.pP[((int64_t)k * D * D + i * D + j) * ld + b]);
                Pp[j][i] = Pp[i][j];
                S[SSMQ_PK(i, j)] = Pp[i][j];
            }
        allok = chol_packed<D>(S) && allok;
        // gain = (P_pr^-1 C)'  -> G[d][i] = X[i][d], X = P_pr^-1 C, C = cross-covariance (D_out x D_in)
        double G[D][D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double v[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double s = RTS_LD(a.pC[((int64_t)k * D * a.c_cols + i * a.c_cols + d) * ld + b]);
#pragma unroll
                for (int q = 0; q < i; ++q) s -= S[SSMQ_PK(i, q)] * v[q];
                v[i] = div_nr(s, S[SSMQ_PK(i, i)]);
            }
#pragma unroll
            for (int i = D - 1; i >= 0; --i) {
                double s = v[i];
#pragma unroll
                for (int q = i + 1; q < D; ++q) s -= S[SSMQ_PK(q, i)] * v[q];
                v[i] = div_nr(s, S[SSMQ_PK(i, i)]);
            }
#pragma unroll
            for (int i = 0; i < D; ++i) G[d][i] = v[i];
        }
        double mn[D], Pn[D][D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) s += G[d][i] * (ms[i] - mp[i]);
            mn[d] = RTS_LD(a.fm[((int64_t)(k - 1) * D + d) * ld + b]) + s;
        }
        // P_s = P_f + G (P_s_next - P_pr) G'
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double w[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) s += G[d][i] * (Ps[i][j] - Pp[i][j]);
                w[j] = s;
            }
#pragma unroll
            for (int d2 = 0; d2 < D; ++d2) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) s += w[j] * G[d2][j];
                Pn[d][d2] = RTS_LD(a.fP[((int64_t)(k - 1) * D * D + d * D + d2) * ld + b]) + s;
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            ms[d] = mn[d];
            __builtin_nontemporal_store(mn[d], &a.sm[((int64_t)(k - 1) * D + d) * ld + b]);
#pragma unroll
            for (int d2 = 0; d2 < D; ++d2) {
                Ps[d][d2] = Pn[d][d2];
                __builtin_nontemporal_store(Pn[d][d2], &a.sP[((int64_t)(k - 1) * D * D + d * D + d2) * ld + b]);
            }
        }
    }
    if (!allok) a.status[b] |= (1 << 30);
}

#undef RTS_LD
template <int D>
static void launch_rts(const RtsArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + kUpdBlock - 1) / kUpdBlock);
    hipLaunchKernelGGL((k_rts_backward<D>), dim3(grid), dim3(kUpdBlock), 0, s, a);
}

int launch_rts_backward(int D, int64_t B, int64_t ld, int T, const double *fm, const double *fP, const double *pm,
                        const double *pP, const double *pC, double *sm, double *sP, int32_t *status, hipStream_t s,
                        int c_cols) {
    RtsArgs a{fm, fP, pm, pP, pC, sm, sP, status, B, ld, T, c_cols > 0 ? c_cols : D};
    switch (D) {
        case 1: launch_rts<1>(a, s); break;
        case 2: launch_rts<2>(a, s); break;
        case 3: launch_rts<3>(a, s); break;
        case 4: launch_rts<4>(a, s); break;
        case 5: launch_rts<5>(a, s); break;
        case 6: launch_rts<6>(a, s); break;
        case 7: launch_rts<7>(a, s); break;
        default:
            set_error("rts_backward: state dimension above 7 is not instantiated");
            return SSMQ_E_UNSUPPORTED;
    }
    return hip_fail(hipGetLastError(), "k_rts_backward");
}

}  // namespace ssmq
