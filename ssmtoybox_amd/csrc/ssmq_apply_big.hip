// The per-trajectory rest of a moment transform whose point set is too large for the LDS-resident kernels and has no
// instantiation of the fused matrix-core route: everything that is not the batch GEMM T = FX Wc (ssmq_gemm_mfma.hip,
// launch_fxwc_blocks).  Point counts up to SSMQ_MAX_PTS (4096) - Gauss-Hermite grids in 6-7 dimensions, fully-symmetric
// rules of degree 7 (1181 points at D = 10): bq/bqmtran.py:178-223, mtran.py:141-149 for any N.
//
// One wave per trajectory; the sums over the points run on the matrix cores (v_mfma_f64_16x16x4_f64) with the rows of
// the trajectory as both operands - lane (row = l & 15, k group = l >> 4) loads 4 consecutive doubles of "its" row per block
// of 16 points (the k order inside a block permuted consistently for both operands, as ssmq_gemm_mfma.hip does):
//   BQ form       cov[e][e2] = sum_n T[e][n] fx[e2][n] - m_e m_e2 + emv     (T = fx Wc from the blocked GEMM; formed for
//                 e2 <= e and mirrored);  t-process: S = (fx iK) fx' likewise from a second GEMM
//                 P'[d][e] = sum_n Wcc[d][n] fx[e][n] comes out of the GEMM as D extra columns of T
//   centred form  the same with T[e][n] := wc_n (fx[e][n] - m_e), fx := fx - m, and P' formed here with
//                 Wcc[d][n] := xi[d][n] wc_n (x_n - m = L xi_n; the constants block carries it)
//   ccov' = L P'  on the matrix cores as well (A = fragments of the factor), as k_apply_tile does
// The first version walked the points with lanes over n and 48 accumulators per output row: 256 VGPRs, one wave per SIMD,
// 8.8 ms at B = 1e4 / N = 1181 - more than the GEMM (5.6 ms).  This one: ~300 matrix instructions and 190 KB of row
// reads per trajectory.
#include "ssmq_host.h"

namespace ssmq {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kRestWaves = 4;

__device__ __forceinline__ void load4(const double *p, bool on, double (&v)[4]) {
    if (on) {
        const double2 a = ((const double2 *)p)[0], b = ((const double2 *)p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
        v[0] = v[1] = v[2] = v[3] = 0.0;
    }
}

// (launch bounds: two workgroups per CU.  Allowed 512 registers the compiler keeps the MFMA accumulators in AGPRs, and on
// gfx950 a v_mfma_f64_16x16x4_f64 with AGPR accumulators issues every 103-140 cycles instead of every 64:
// tools/micro/mfma_f64_peak.hip - 48 against 77 TFLOP/s chip-wide.  Every other matrix-core kernel of the library already
// had its accumulators in architectural VGPRs.)
__global__ __launch_bounds__(64 * kRestWaves, 2) void k_big_rest(const BigRest r, int64_t B) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;
    const int64_t b = (int64_t)blockIdx.x * kRestWaves + (threadIdx.x >> 6);
    if (b >= B) return;                                   // whole waves leave; no workgroup barrier below
    const int D = r.D, E = r.E, N = r.N;
    const bool sigma = r.form == SSMQ_FORM_SIGMA, tp = r.tp_nu > 0.0 && r.t2 != nullptr;
    const WideLayout cl = wide_layout(D, E, N, r.form);
    const double *cs = r.consts;
    const double *L = r.chol + b * D * D;
    const double nan = __builtin_nan("");
    const bool ok = !r.status || r.status[b] == 0;
    const bool row_on = c < E;
    const double mc = row_on ? r.mean_rows[b * E + c] : 0.0;                 // mean of row / column c
    const double *fx = r.fx + (b * E + (row_on ? c : 0)) * r.lda + 4 * q;
    const double *tt = r.t ? r.t + (b * E + (row_on ? c : 0)) * r.ldt + 4 * q : nullptr;
    const double *t2 = tp ? r.t2 + (b * E + (row_on ? c : 0)) * r.ldt + 4 * q : nullptr;
    const double *wcc = cs + cl.Wcc + (int64_t)(c < D ? c : 0) * N + 4 * q;   // centred form: xi[d][n] wc_n, natural (D, N)
    const int KB = (N + 15) / 16;
    v4d cov = {0.0, 0.0, 0.0, 0.0}, sq = {0.0, 0.0, 0.0, 0.0}, pacc = {0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < KB; ++kb) {
        const int n0 = 16 * kb + 4 * q;
        double f4[4], a4[4], s4[4], w4[4];
        load4(fx + 16 * kb, row_on, f4);                  // rows are zero-padded to a multiple of 16 points
        if (sigma) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool in = n0 + s < N;
                f4[s] = (row_on && in) ? f4[s] - mc : 0.0;
                a4[s] = in ? cs[cl.Wc + (in ? n0 + s : 0)] * f4[s] : 0.0;
                w4[s] = (c < D && in) ? wcc[16 * kb + s] : 0.0;
            }
        } else {
            load4(tt + 16 * kb, row_on, a4);
            if (tp) load4(t2 + 16 * kb, row_on, s4);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            cov = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s], f4[s], cov, 0, 0, 0);
            if (tp) sq = __builtin_amdgcn_mfma_f64_16x16x4f64(s4[s], f4[s], sq, 0, 0, 0);
            if (sigma) pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(w4[s], f4[s], pacc, 0, 0, 0);
        }
    }
    // cross-covariance: ccov' = L P'.  BQ: P'[d][e] = T[e][p_col + d] (the GEMM's extra columns), read straight into the
    // B-operand layout of step t (lane (e, q): d = 4 t + q); centred form: the accumulator registers are that layout
    v4d cc = {0.0, 0.0, 0.0, 0.0};
    for (int t4 = 0; 4 * t4 < D; ++t4) {
        const int d = 4 * t4 + q;
        const double lf = (c < D && d < D) ? L[c * D + d] : 0.0;
        double pb;
        if (sigma) {
            pb = t4 == 0 ? pacc[0] : t4 == 1 ? pacc[1] : t4 == 2 ? pacc[2] : pacc[3];
        } else {
            pb = (row_on && d < D) ? r.t[(b * E + c) * r.ldt + r.p_col + d] : 0.0;
        }
        cc = __builtin_amdgcn_mfma_f64_16x16x4f64(lf, pb, cc, 0, 0, 0);
    }
    const double tp_den = tp ? 1.0 / (r.tp_nu - 2.0 + (double)N) : 0.0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int e1 = q + 4 * rr, e2 = c;
        const double m1 = __shfl(mc, e1 & 15, 64);           // lane e1 (q = 0) holds the mean of row e1
        if (e1 < E && e2 <= e1) {
            const int idx = e1 * E + e2;
            double v;
            if (sigma) {
                v = cov[rr] * r.cov_scale;
            } else {
                const bool use = (e1 == e2) || r.emv_mode == SSMQ_EMV_BROADCAST;
                double em = use ? cs[cl.emv + idx] : 0.0;
                if (tp) em = (r.tp_nu - 2.0 + sq[rr]) * tp_den * em;
                v = (cov[rr] - m1 * mc + em) * r.cov_scale;
            }
            if (r.cov_add) v += r.cov_add[idx];
            v = ok ? v : nan;
            r.cov_f[(int64_t)idx * r.es + b * r.bs_cf] = v;
            if (e2 != e1) r.cov_f[(int64_t)(e2 * E + e1) * r.es + b * r.bs_cf] = v;
        }
        const int dd = q + 4 * rr;                            // ccov[e = c][j = dd]
        if (c < E && dd < D) r.cov_fx[(int64_t)(c * D + dd) * r.es + b * r.bs_cfx] = ok ? cc[rr] * r.ccov_scale : nan;
    }
}

}  // namespace

int launch_big_rest(const BigRest &r, int64_t B, hipStream_t s) {
    if (B <= 0) return SSMQ_OK;
    hipLaunchKernelGGL(k_big_rest, dim3((unsigned)((B + kRestWaves - 1) / kRestWaves)), dim3(64 * kRestWaves), 0, s, r, B);
    return hip_fail(hipGetLastError(), "k_big_rest");
}

}  // namespace ssmq
