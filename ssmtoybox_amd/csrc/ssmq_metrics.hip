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
// are deterministic and independent of launch timing.  HBM-bound: 8 (2 D + D^2) bytes per trajectory and step.
#include "ssmq_host.h"

namespace ssmq {
namespace {

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

// phase 1 values per time step: se[D] | rmse | nll | mse[D*D] | n_ok | n_pd
template <int DT>
__global__ __launch_bounds__(kMetBlock) void k_error_sums(MetArgs a) {
    const int D = DT > 0 ? DT : a.D;
    const int NV = met_nv(D);
    const int t = blockIdx.y;
    extern __shared__ double sh[];
    double acc[DT > 0 ? met_nv(DT) : met_nv(kMetMaxD)];
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    const double *x = a.x + (int64_t)t * D * a.ld, *fm = a.fm + (int64_t)t * D * a.ld;
    const double *fP = a.fP + (int64_t)t * D * D * a.ld;
    const int64_t base = (int64_t)blockIdx.x * kMetBlock * kMetPerThread;
    for (int r = 0; r < kMetPerThread; ++r) {
        const int64_t b = base + (int64_t)r * kMetBlock + threadIdx.x;
        if (b >= a.B) break;
        if (a.status && a.status[b] != 0) continue;
        double dx[DT > 0 ? DT : kMetMaxD], P[DT > 0 ? DT * DT : kMetMaxD * kMetMaxD];
        double n2 = 0.0;
        for (int d = 0; d < D; ++d) {
            dx[d] = x[(int64_t)d * a.ld + b] - fm[(int64_t)d * a.ld + b];
            acc[d] += dx[d] * dx[d];
            n2 += dx[d] * dx[d];
        }
        acc[D] += sqrt(n2);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                acc[D + 2 + i * D + j] += dx[i] * dx[j];
                P[i * D + j] = fP[((int64_t)i * D + j) * a.ld + b];
            }
        acc[D + 2 + D * D] += 1.0;
        // negative log-likelihood (utils.py:143-148) for positive-definite P: log det = 2 sum log L_ii
        if (chol_dense<DT>(P, D)) {
            double logdet = 0.0;
            for (int i = 0; i < D; ++i) logdet += log(P[i * D + i]);
            const double q = whitened_norm2<DT>(P, dx, D);
            acc[D + 1] += 0.5 * (2.0 * logdet + q + D * 1.8378770664093453 /* log(2 pi) */);
            acc[D + 3 + D * D] += 1.0;
        }
    }
    block_sums(acc, NV, sh, a.partial + ((int64_t)t * a.chunks + blockIdx.x) * NV);
}

// phase 2 values per time step: lcr | n counted
template <int DT>
__global__ __launch_bounds__(kMetBlock) void k_lcr_sums(MetArgs a) {
    const int D = DT > 0 ? DT : a.D;
    const int t = blockIdx.y;
    extern __shared__ double sh[];
    double acc[2] = {0.0, 0.0};
    const double *x = a.x + (int64_t)t * D * a.ld, *fm = a.fm + (int64_t)t * D * a.ld;
    const double *fP = a.fP + (int64_t)t * D * D * a.ld;
    // the step's MSE matrix is the same for every trajectory: factor it once per thread from L2 / scalar cache
    double M[DT > 0 ? DT * DT : kMetMaxD * kMetMaxD];
    for (int i = 0; i < D * D; ++i) M[i] = a.mse[(int64_t)t * D * D + i];
    const bool m_ok = chol_dense<DT>(M, D);
    const int64_t base = (int64_t)blockIdx.x * kMetBlock * kMetPerThread;
    for (int r = 0; r < kMetPerThread && m_ok; ++r) {
        const int64_t b = base + (int64_t)r * kMetBlock + threadIdx.x;
        if (b >= a.B) break;
        if (a.status && a.status[b] != 0) continue;
        double dx[DT > 0 ? DT : kMetMaxD], P[DT > 0 ? DT * DT : kMetMaxD * kMetMaxD];
        for (int d = 0; d < D; ++d) dx[d] = x[(int64_t)d * a.ld + b] - fm[(int64_t)d * a.ld + b];
        for (int i = 0; i < D * D; ++i) P[i] = fP[(int64_t)i * a.ld + b];
        if (!chol_dense<DT>(P, D)) continue;      // the reference falls back to an SVD square root here (utils.py:426-432)
        const double qa = whitened_norm2<DT>(P, dx, D), qb = whitened_norm2<DT>(M, dx, D);
        acc[0] += 10.0 * (log10(qa) - log10(qb));
        acc[1] += 1.0;
    }
    block_sums(acc, 2, sh, a.partial + ((int64_t)t * a.chunks + blockIdx.x) * 2);
}

// out[t][v] = sum over chunks, in chunk order
__global__ void k_reduce_partials(const double *partial, double *out, int chunks, int NV, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = i / NV, v = i % NV;
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += partial[((int64_t)t * chunks + c) * NV + v];
    out[i] = s;
}

template <int DT>
hipError_t launch_phase(const MetArgs &a, int phase, hipStream_t s) {
    const int D = DT > 0 ? DT : a.D;
    const int NV = phase == 1 ? met_nv(D) : 2;
    const size_t lds = sizeof(double) * (kMetBlock / 64) * NV;
    dim3 grid((unsigned)a.chunks, (unsigned)a.T);
    if (phase == 1)
        hipLaunchKernelGGL(k_error_sums<DT>, grid, dim3(kMetBlock), lds, s, a);
    else
        hipLaunchKernelGGL(k_lcr_sums<DT>, grid, dim3(kMetBlock), lds, s, a);
    return hipGetLastError();
}

}  // namespace

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
    const int NV = phase == 1 ? met_nv(D) : 2, total = T * NV;
    hipLaunchKernelGGL(k_reduce_partials, dim3((total + 255) / 256), dim3(256), 0, s, partial, out, a.chunks, NV, total);
    return hip_fail(hipGetLastError(), "k_reduce_partials");
}

}  // namespace ssmq
