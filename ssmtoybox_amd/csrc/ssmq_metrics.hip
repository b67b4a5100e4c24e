// Error statistics of a batch of filtered trajectories, reduced over the Monte-Carlo axis on the device.
//
// Reference: ssmtoybox/utils.py:18-38 (squared_error), :41-64 (mse_matrix), :66-120 (log_cred_ratio),
// :123-148 (neg_log_likelihood) as the studies aggregate them (research/tpq/tpq_base.py:154-172: RMSE averaged over
// simulations, MSE matrix per time step, log credibility ratio against the per-step MSE matrix).
//
// Layout: the filter outputs stay where ssmq_filter_forward_dev left them - planes [T][D][ld] / [T][D*D][ld], one
// trajectory per lane - so each thread streams its trajectory's D + D + D*D doubles of a time step with 512-B wave
// accesses and nothing is copied to the host except [T][NV] sums.  Reduction order is fixed (thread-serial over a
// strided slice, wave shuffle tree, LDS across waves, then a second kernel over the partials in index order): results
// are deterministic and independent of launch timing.  HBM-bound: algorithmic 8 (2 D + D^2) bytes per trajectory and
// step; the kernels move 8 (2 D + D (D + 1) / 2) - only the lower triangle of P is read (the Cholesky factor needs no
// more, as LAPACK 'L' under the filters) and only the lower triangle of the outer products is accumulated.
#include "ssmq_host.h"

namespace ssmq {
namespace {

// every input is read exactly once: non-temporal loads keep the stream out of the way of L2 (238 -> 210 us at D = 6)
#define SSMQ_MLOAD(src) __builtin_nontemporal_load(&(src))

constexpr int kMetBlock = 256;
constexpr int kMetPerThread = 8;       // trajectories each thread accumulates before the block reduction
constexpr int kMetMaxD = SSMQ_MAX_DIM;

__host__ __device__ constexpr int met_nv(int D) { return D + 2 + D * D + 2; }

// in-place lower Cholesky of a dense D x D matrix held in registers / scratch; false if not positive definite
template <int DT>
__device__ __forceinline__ bool chol_dense(double *A, int D) {
    const int n = DT > 0 ? DT : D;
    bool ok = true;
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0.0)) ok = false;
        const double l = sqrt(s), r = 1.0 / l;
        A[j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double v = A[i * n + j];
            for (int k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = v * r;
        }
    }
    return ok;
}

// |L^-1 dx|^2 = dx' (L L')^-1 dx
template <int DT>
__device__ __forceinline__ double whitened_norm2(const double *L, const double *dx, int D) {
    const int n = DT > 0 ? DT : D;
    double v[DT > 0 ? DT : kMetMaxD];
    double q = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = dx[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * v[k];
        v[i] = s / L[i * n + i];
        q += v[i] * v[i];
    }
    return q;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sums of NV per-thread values -> out[NV] (written by thread 0..NV-1); sh: [kMetBlock / 64][NV]
__device__ __forceinline__ void block_sums(const double *acc, int NV, double *sh, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v = 0; v < NV; ++v) {
        const double s = wave_sum(acc[v]);
        if (lane == 0) sh[wave * NV + v] = s;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < NV; v += kMetBlock) {
        double s = 0.0;
        for (int w = 0; w < kMetBlock / 64; ++w) s += sh[w * NV + v];
        out[v] = s;
    }
}

struct MetArgs {
    const double *x, *fm, *fP;     // [T][D][ld], [T][D][ld], [T][D*D][ld]
    const int32_t *status;         // [ld] or null; nonzero = trajectory excluded (its filter failed)
    const double *mse;             // phase 2: [T][D*D] global MSE matrices (already regularised), device
    double *partial;               // [T][chunks][NV]
    int64_t B, ld;
    int32_t D, T, chunks;
};

// block-wide sums with a compile-time count (accumulators stay in registers)
template <int NV>
__device__ __forceinline__ void block_sums_fixed(const double (&acc)[NV], double *sh, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const double s = wave_sum(acc[v]);
        if (lane == 0) sh[wave * NV + v] = s;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < NV; v += kMetBlock) {
        double s = 0.0;
        for (int w = 0; w < kMetBlock / 64; ++w) s += sh[w * NV + v];
        out[v] = s;
    }
}

// |L^-1 dx|^2 with L packed lower
template <int D>
__device__ __forceinline__ double whitened_norm2_packed(const double (&L)[D * (D + 1) / 2], const double (&dx)[D]) {
    double v[D];
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double s = dx[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[SSMQ_PK(i, k)] * v[k];
        v[i] = s / L[SSMQ_PK(i, i)];
        q += v[i] * v[i];
    }
    return q;
}

// phase 1 values per time step (output): se[D] | rmse | nll | mse[D*D] | n_ok | n_pd.
// Compile-time D: accumulators se[D], rmse, nll, lower triangle of the outer products, two counts - all in registers;
// the partial row is written in the packed order and expanded by k_reduce_partials.
template <int D>
__global__ __launch_bounds__(kMetBlock) void k_error_sums(MetArgs a) {
    constexpr int TRI = D * (D + 1) / 2, NA = D + 2 + TRI + 2;
    const int t = blockIdx.y;
    extern __shared__ double sh[];
    double acc[NA];
#pragma unroll
    for (int v = 0; v < NA; ++v) acc[v] = 0.0;
    const double *x = a.x + (int64_t)t * D * a.ld, *fm = a.fm + (int64_t)t * D * a.ld;
    const double *fP = a.fP + (int64_t)t * D * D * a.ld;
    // 32-bit lane index against wave-uniform plane pointers: loads take the (SGPR base + VGPR offset) form and no
    // 64-bit address is kept per plane
    const uint32_t base = blockIdx.x * (uint32_t)(kMetBlock * kMetPerThread) + threadIdx.x;
    for (int r = 0; r < kMetPerThread; ++r) {
        const uint32_t b = base + (uint32_t)r * kMetBlock;
        if ((int64_t)b >= a.B) break;
        if (a.status && a.status[b] != 0) continue;
        double dx[D], L[TRI];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) L[SSMQ_PK(i, j)] = SSMQ_MLOAD(fP[((int64_t)i * D + j) * a.ld + b]);
        double n2 = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            dx[d] = SSMQ_MLOAD(x[(int64_t)d * a.ld + b]) - SSMQ_MLOAD(fm[(int64_t)d * a.ld + b]);
            acc[d] += dx[d] * dx[d];
            n2 += dx[d] * dx[d];
        }
        acc[D] += sqrt(n2);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) acc[D + 2 + SSMQ_PK(i, j)] += dx[i] * dx[j];
        acc[D + 2 + TRI] += 1.0;
        // negative log-likelihood (utils.py:143-148) for positive-definite P: log det = 2 sum log L_ii
        if (chol_packed<D>(L)) {
            double logdet = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) logdet += log(L[SSMQ_PK(i, i)]);
            const double q = whitened_norm2_packed<D>(L, dx);
            acc[D + 1] += 0.5 * (2.0 * logdet + q + D * 1.8378770664093453 /* log(2 pi) */);
            acc[D + 3 + TRI] += 1.0;
        }
    }
    block_sums_fixed<NA>(acc, sh, a.partial + ((int64_t)t * a.chunks + blockIdx.x) * NA);
}

// run-time D (> 6): dense arrays in scratch, same packed partial row
__global__ __launch_bounds__(kMetBlock) void k_error_sums_generic(MetArgs a) {
    const int D = a.D, TRI = D * (D + 1) / 2, NA = D + 2 + TRI + 2;
    const int t = blockIdx.y;
    extern __shared__ double sh[];
    double acc[kMetMaxD + 2 + kMetMaxD * (kMetMaxD + 1) / 2 + 2];
    for (int v = 0; v < NA; ++v) acc[v] = 0.0;
    const double *x = a.x + (int64_t)t * D * a.ld, *fm = a.fm + (int64_t)t * D * a.ld;
    const double *fP = a.fP + (int64_t)t * D * D * a.ld;
    const int64_t base = (int64_t)blockIdx.x * kMetBlock * kMetPerThread;
    for (int r = 0; r < kMetPerThread; ++r) {
        const int64_t b = base + (int64_t)r * kMetBlock + threadIdx.x;
        if (b >= a.B) break;
        if (a.status && a.status[b] != 0) continue;
        double dx[kMetMaxD], P[kMetMaxD * kMetMaxD];
        double n2 = 0.0;
        for (int d = 0; d < D; ++d) {
            dx[d] = SSMQ_MLOAD(x[(int64_t)d * a.ld + b]) - SSMQ_MLOAD(fm[(int64_t)d * a.ld + b]);
            acc[d] += dx[d] * dx[d];
            n2 += dx[d] * dx[d];
        }
        acc[D] += sqrt(n2);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j <= i; ++j) {
                acc[D + 2 + SSMQ_PK(i, j)] += dx[i] * dx[j];
                P[i * D + j] = fP[((int64_t)i * D + j) * a.ld + b];
            }
        acc[D + 2 + TRI] += 1.0;
        if (chol_dense<0>(P, D)) {
            double logdet = 0.0;
            for (int i = 0; i < D; ++i) logdet += log(P[i * D + i]);
            const double q = whitened_norm2<0>(P, dx, D);
            acc[D + 1] += 0.5 * (2.0 * logdet + q + D * 1.8378770664093453);
            acc[D + 3 + TRI] += 1.0;
        }
    }
    block_sums(acc, NA, sh, a.partial + ((int64_t)t * a.chunks + blockIdx.x) * NA);
}

// phase 2 values per time step: lcr | n counted | n left to the second pass (P not positive definite)
template <int D>
__global__ __launch_bounds__(kMetBlock) void k_lcr_sums(MetArgs a) {
    constexpr int TRI = D * (D + 1) / 2;
    const int t = blockIdx.y;
    extern __shared__ double sh[];
    double acc[3] = {0.0, 0.0, 0.0};
    const double *x = a.x + (int64_t)t * D * a.ld, *fm = a.fm + (int64_t)t * D * a.ld;
    const double *fP = a.fP + (int64_t)t * D * D * a.ld;
    // the step's MSE matrix is the same for every trajectory: wave-uniform loads, factored once per thread
    double M[TRI];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) M[SSMQ_PK(i, j)] = a.mse[(int64_t)t * D * D + i * D + j];
    const bool m_ok = chol_packed<D>(M);
    const uint32_t base = blockIdx.x * (uint32_t)(kMetBlock * kMetPerThread) + threadIdx.x;
    for (int r = 0; r < kMetPerThread && m_ok; ++r) {
        const uint32_t b = base + (uint32_t)r * kMetBlock;
        if ((int64_t)b >= a.B) break;
        if (a.status && a.status[b] != 0) continue;
        double dx[D], L[TRI];
#pragma unroll
        for (int d = 0; d < D; ++d) dx[d] = x[(int64_t)d * a.ld + b] - fm[(int64_t)d * a.ld + b];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) L[SSMQ_PK(i, j)] = SSMQ_MLOAD(fP[((int64_t)i * D + j) * a.ld + b]);
        if (!chol_packed<D>(L)) {              // the reference's SVD square root (utils.py:426-432): k_indef_sums
            acc[2] += 1.0;
            continue;
        }
        const double qa = whitened_norm2_packed<D>(L, dx), qb = whitened_norm2_packed<D>(M, dx);
        acc[0] += 10.0 * (log10(qa) - log10(qb));
        acc[1] += 1.0;
    }
    block_sums_fixed<3>(acc, sh, a.partial + ((int64_t)t * a.chunks + blockIdx.x) * 3);
}

__global__ __launch_bounds__(kMetBlock) void k_lcr_sums_generic(MetArgs a) {
    const int D = a.D;
    const int t = blockIdx.y;
    extern __shared__ double sh[];
    double acc[3] = {0.0, 0.0, 0.0};
    const double *x = a.x + (int64_t)t * D * a.ld, *fm = a.fm + (int64_t)t * D * a.ld;
    const double *fP = a.fP + (int64_t)t * D * D * a.ld;
    double M[kMetMaxD * kMetMaxD];
    for (int i = 0; i < D * D; ++i) M[i] = a.mse[(int64_t)t * D * D + i];
    const bool m_ok = chol_dense<0>(M, D);
    const int64_t base = (int64_t)blockIdx.x * kMetBlock * kMetPerThread;
    for (int r = 0; r < kMetPerThread && m_ok; ++r) {
        const int64_t b = base + (int64_t)r * kMetBlock + threadIdx.x;
        if (b >= a.B) break;
        if (a.status && a.status[b] != 0) continue;
        double dx[kMetMaxD], P[kMetMaxD * kMetMaxD];
        for (int d = 0; d < D; ++d) dx[d] = x[(int64_t)d * a.ld + b] - fm[(int64_t)d * a.ld + b];
        for (int i = 0; i < D; ++i)
            for (int j = 0; j <= i; ++j) P[i * D + j] = fP[((int64_t)i * D + j) * a.ld + b];
        if (!chol_dense<0>(P, D)) {
            acc[2] += 1.0;
            continue;
        }
        const double qa = whitened_norm2<0>(P, dx, D), qb = whitened_norm2<0>(M, dx, D);
        acc[0] += 10.0 * (log10(qa) - log10(qb));
        acc[1] += 1.0;
    }
    block_sums(acc, 3, sh, a.partial + ((int64_t)t * a.chunks + blockIdx.x) * 3);
}

// ---- covariances that are not positive definite ----------------------------------------------------------------------
// The reference's metrics do not need a positive-definite P: neg_log_likelihood uses inv(P) and sign * logdet of
// slogdet (utils.py:143-148), log_cred_ratio's mat_sqrt falls back to u sqrt(s) of an SVD (utils.py:426-432), i.e. the
// quadratic form with |P| = U S U'.  The streaming kernels above go through the Cholesky factor and leave such entries
// out; when the counts say that some were left out, this kernel makes a second pass that adds exactly those terms:
//   phase 1: 0.5 (sign log|det P| + dx' P^-1 dx + D log 2 pi) from an LU factorisation with partial pivoting of the
//            full matrix;  phase 2: 10 (log10 dx'|P|^-1 dx - log10 dx' M^-1 dx), |P| through a Jacobi
//            eigen-decomposition of the symmetric part (eigenvalues in absolute value).
// Rare path, run-time D, dense arrays in scratch; same deterministic two-level reduction.  partial row: term sum | count.
__device__ bool lu_solve_logdet(double *A, const double *rhs, double *z, int n, double *sign, double *logabs) {
    int perm_sign = 1;
    double la = 0.0;
    for (int i = 0; i < n; ++i) z[i] = rhs[i];
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (!(best > 0.0)) return false;           // singular (or NaN): numpy.linalg.inv raises
        if (p != k) {
            for (int j = 0; j < n; ++j) { const double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            const double t = z[k]; z[k] = z[p]; z[p] = t;
            perm_sign = -perm_sign;
        }
        const double piv = A[k * n + k];
        if (piv < 0.0) perm_sign = -perm_sign;
        la += log(fabs(piv));
        for (int i = k + 1; i < n; ++i) {
            const double l = A[i * n + k] / piv;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
            z[i] -= l * z[k];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s2 = z[i];
        for (int j = i + 1; j < n; ++j) s2 -= A[i * n + j] * z[j];
        z[i] = s2 / A[i * n + i];
    }
    *sign = (double)perm_sign;
    *logabs = la;
    return true;
}

// dx' |S|^-1 dx for the symmetric matrix S (destroyed): cyclic Jacobi rotations, V accumulates the eigenvectors
__device__ double abs_quadratic_form(double *S, double *V, const double *dx, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += S[i * n + i] * S[i * n + i];
            for (int j = 0; j < i; ++j) off += S[i * n + j] * S[i * n + j];
        }
        if (!(off > 1e-32 * diag)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = S[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (S[q * n + q] - S[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {
                    const double skp = S[k * n + p], skq = S[k * n + q];
                    S[k * n + p] = c * skp - sn * skq;
                    S[k * n + q] = sn * skp + c * skq;
                }
                for (int k = 0; k < n; ++k) {
                    const double spk = S[p * n + k], sqk = S[q * n + k];
                    S[p * n + k] = c * spk - sn * sqk;
                    S[q * n + k] = sn * spk + c * sqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - sn * vkq;
                    V[k * n + q] = sn * vkp + c * vkq;
                }
            }
    }
    double q = 0.0;
    for (int i = 0; i < n; ++i) {
        double u = 0.0;
        for (int k = 0; k < n; ++k) u += V[k * n + i] * dx[k];
        q += u * u / fabs(S[i * n + i]);
    }
    return q;
}

__global__ __launch_bounds__(kMetBlock) void k_indef_sums(MetArgs a, int phase) {
    const int D = a.D;
    const int t = blockIdx.y;
    extern __shared__ double sh[];
    double acc[2] = {0.0, 0.0};
    const double *x = a.x + (int64_t)t * D * a.ld, *fm = a.fm + (int64_t)t * D * a.ld;
    const double *fP = a.fP + (int64_t)t * D * D * a.ld;
    double M[kMetMaxD * kMetMaxD];
    bool m_ok = true;
    if (phase == 2) {
        for (int i = 0; i < D * D; ++i) M[i] = a.mse[(int64_t)t * D * D + i];
        m_ok = chol_dense<0>(M, D);
    }
    const int64_t base = (int64_t)blockIdx.x * kMetBlock * kMetPerThread;
    for (int r = 0; r < kMetPerThread && m_ok; ++r) {
        const int64_t b = base + (int64_t)r * kMetBlock + threadIdx.x;
        if (b >= a.B) break;
        if (a.status && a.status[b] != 0) continue;
        double dx[kMetMaxD], P[kMetMaxD * kMetMaxD], W[kMetMaxD * kMetMaxD], z[kMetMaxD];
        for (int i = 0; i < D; ++i)
            for (int j = 0; j <= i; ++j) W[i * D + j] = fP[((int64_t)i * D + j) * a.ld + b];
        if (chol_dense<0>(W, D)) continue;          // positive definite: the streaming kernel has counted it
        for (int d = 0; d < D; ++d) dx[d] = x[(int64_t)d * a.ld + b] - fm[(int64_t)d * a.ld + b];
        for (int i = 0; i < D * D; ++i) P[i] = fP[(int64_t)i * a.ld + b];
        if (phase == 1) {
            double sign, logabs;
            if (!lu_solve_logdet(P, dx, z, D, &sign, &logabs)) continue;
            double q = 0.0;
            for (int d = 0; d < D; ++d) q += dx[d] * z[d];
            acc[0] += 0.5 * (sign * logabs + q + D * 1.8378770664093453);
            acc[1] += 1.0;
        } else {
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < D; ++j) W[i * D + j] = 0.5 * (P[i * D + j] + P[j * D + i]);
            const double qa = abs_quadratic_form(W, P, dx, D), qb = whitened_norm2<0>(M, dx, D);
            acc[0] += 10.0 * (log10(qa) - log10(qb));
            acc[1] += 1.0;
        }
    }
    block_sums(acc, 2, sh, a.partial + ((int64_t)t * a.chunks + blockIdx.x) * 2);
}

// out[t][v] = sum over chunks, in chunk order.  D > 0: phase-1 rows, partials hold the outer products as a packed
// lower triangle (NA values per row) and the output row is the full layout of met_nv(D) values.
__global__ void k_reduce_partials(const double *partial, double *out, int chunks, int NV, int total, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = i / NV, v = i % NV;
    int NA = NV, src = v;
    if (D > 0) {
        const int TRI = D * (D + 1) / 2;
        NA = D + 2 + TRI + 2;
        if (v >= D + 2 && v < D + 2 + D * D) {
            const int r = (v - D - 2) / D, c = (v - D - 2) % D;
            src = D + 2 + (r >= c ? SSMQ_PK(r, c) : SSMQ_PK(c, r));
        } else if (v >= D + 2 + D * D) {
            src = v - D * D + TRI;
        }
    }
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += partial[((int64_t)t * chunks + c) * NA + src];
    out[i] = s;
}

template <int DT>
hipError_t launch_phase(const MetArgs &a, int phase, hipStream_t s) {
    const int D = DT > 0 ? DT : a.D;
    const int NV = phase == 1 ? met_nv(D) : 3;       // upper bound of the packed row
    const size_t lds = sizeof(double) * (kMetBlock / 64) * NV;
    dim3 grid((unsigned)a.chunks, (unsigned)a.T);
    if constexpr (DT > 0) {
        if (phase == 1)
            hipLaunchKernelGGL(k_error_sums<DT>, grid, dim3(kMetBlock), lds, s, a);
        else
            hipLaunchKernelGGL(k_lcr_sums<DT>, grid, dim3(kMetBlock), lds, s, a);
    } else {
        if (phase == 1)
            hipLaunchKernelGGL(k_error_sums_generic, grid, dim3(kMetBlock), lds, s, a);
        else
            hipLaunchKernelGGL(k_lcr_sums_generic, grid, dim3(kMetBlock), lds, s, a);
    }
    return hipGetLastError();
}

}  // namespace

int metrics_chunks(int64_t B);

// second pass over the entries whose covariance is not positive definite: d_out [T][2] = term sum | count
int launch_metrics_indef(int phase, int D, int64_t B, int64_t ld, int T, const double *x, const double *fm, const double *fP,
                         const int32_t *status, const double *mse, double *partial, double *out, hipStream_t s) {
    MetArgs a{x, fm, fP, status, mse, partial, B, ld, D, T, metrics_chunks(B)};
    hipLaunchKernelGGL(k_indef_sums, dim3((unsigned)a.chunks, (unsigned)T), dim3(kMetBlock), sizeof(double) * (kMetBlock / 64) * 2,
                       s, a, phase);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_indef_sums");
    hipLaunchKernelGGL(k_reduce_partials, dim3((T * 2 + 255) / 256), dim3(256), 0, s, partial, out, a.chunks, 2, T * 2, 0);
    return hip_fail(hipGetLastError(), "k_reduce_partials");
}

int metrics_values_per_step(int D) { return met_nv(D); }

// d_out [T][NV] device; d_partial scratch of T * chunks * NV doubles (chunks from metrics_chunks)
int metrics_chunks(int64_t B) { return (int)((B + (int64_t)kMetBlock * kMetPerThread - 1) / ((int64_t)kMetBlock * kMetPerThread)); }

int launch_metrics(int phase, int D, int64_t B, int64_t ld, int T, const double *x, const double *fm, const double *fP,
                   const int32_t *status, const double *mse, double *partial, double *out, hipStream_t s) {
    MetArgs a{x, fm, fP, status, mse, partial, B, ld, D, T, metrics_chunks(B)};
    hipError_t e;
    switch (D) {
        case 1: e = launch_phase<1>(a, phase, s); break;
        case 2: e = launch_phase<2>(a, phase, s); break;
        case 3: e = launch_phase<3>(a, phase, s); break;
        case 4: e = launch_phase<4>(a, phase, s); break;
        case 5: e = launch_phase<5>(a, phase, s); break;
        case 6: e = launch_phase<6>(a, phase, s); break;
        default: e = launch_phase<0>(a, phase, s); break;
    }
    if (e != hipSuccess) return hip_fail(e, phase == 1 ? "k_error_sums" : "k_lcr_sums");
    const int NV = phase == 1 ? met_nv(D) : 3, total = T * NV;
    hipLaunchKernelGGL(k_reduce_partials, dim3((total + 255) / 256), dim3(256), 0, s, partial, out, a.chunks, NV, total,
                       phase == 1 ? D : 0);
    return hip_fail(hipGetLastError(), "k_reduce_partials");
}

}  // namespace ssmq
