// The fused filter time loop with ONE TRAJECTORY ON FOUR LANES (round 6), for batches whose whole-pass waves would each sit
// alone on a SIMD: BASELINE configs[2] on eight GPUs is 12 500 trajectories per GPU = 196 waves of k_filter_fused on 1 024 SIMDs,
// and a lone wave is bound by the ~2 300 instructions it issues per step, one every four cycles, whatever their kind.  Here a
// wave carries 16 trajectories instead of 64 (782 waves for 12 500: still one per SIMD) and issues about half as many
// instructions per step:
//   * the sigma points of a transform (bq/bqmtran.py:132-156 evaluates them one after the other) are dealt to the four lanes of a
//     trajectory's quad, point n to lane n mod 4: three (D = 5, 11 points) or four (D = 6, 13 points) evaluations of the integrand
//     per lane instead of 11 / 13 - the integrands are half of the register kernel's instructions;
//   * every lane adds up the weighted sums (mtran.py:141-148: mean, centred covariance, cross-covariance) over ITS points and the
//     partial sums are all-reduced over the quad - two exchanges per value, each two `v_mov_b32_dpp quad_perm` and an add (the
//     only DPP form that moves fp64 data between the lanes of a quad).  The reduction tree is symmetric, so all four lanes end
//     up with the same bits and carry on with an identical, replicated state;
//   * the factorisations (chol of the state covariance before either transform, ssinf.py:276-288 / mtran.py:139) and the
//     measurement update (ssinf.py:297-323) are serial chains: every lane of the quad runs them (replication costs a lone wave
//     nothing that splitting them would not cost in exchanges);
//   * the one lane-dependent access - lane q needs column (n - 1) mod D of the Cholesky factor for its point n - goes through
//     LDS: the quad's first lane writes the factor, dense and column-major, and each lane reads its columns back
//     (a select chain over registers would cost 6 instructions per coordinate).
// What splitting the OUTPUT entries over the lanes (as the round-5 review sketched) would need on top - each lane forming
// different entries from different operands - is lane-dependent REGISTER indexing, which a SIMD lane does not have: it turns into
// the same LDS round trips or select chains, for every operand.  Hence partial sums + all-reduce.
// Centred sigma-point form on unscented points (the filters that are stable on the reentry model: configs[2]), Gaussian recursion.
// Summation order differs from k_filter_fused (per-lane partial sums), so the results agree to rounding, not bit for bit:
// tests/test_gpu_parity.py::test_quad_filters_match_oracle_and_register_kernel.
#include <cstring>
#include "ssmq_filter_fused_kernel.h"

namespace ssmq {
namespace {

template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// sum over the four lanes of a quad, the same bits in all four: (v0 + v1) + (v2 + v3) in every lane (addition commutes)
__device__ __forceinline__ double quad_sum(double v) {
    v += quad_perm<0xB1>(v);      // quad_perm:[1,0,3,2]
    v += quad_perm<0x4E>(v);      // quad_perm:[2,3,0,1]
    return v;
}

// ... for NV values at once: all exchanges of a level first, then its additions - the DPP moves read registers written several
// instructions earlier (no hazard wait states) and the additions of a level are independent of each other (a lone wave issues a
// DEPENDENT fp64 operation every ~7 cycles, an independent one every ~5: profiles/r05_lane_prims.txt)
template <int NV>
__device__ __forceinline__ void quad_sum_all(double (&v)[NV]) {
    double t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] = quad_perm<0xB1>(v[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += t[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] = quad_perm<0x4E>(v[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += t[i];
}

constexpr int kQuadTraj = 16;     // trajectories per wave

// One moment transform in the centred form on unscented points, the N = 2 DI + 1 points dealt to the quad's lanes.
//   m, L: mean and Cholesky factor (packed lower) of the input, replicated; lds: this trajectory's record (DI x CS doubles);
//   col_off[s] / sgn[s] / wm[s] / wc[s]: per lane, the column, +-c (0 for the centre point and for empty slots) and the weights of
//   the lane's s-th point.  Results (replicated): mf, cv (packed lower, + cadd), and with CROSS cx[e][d].
template <int DI, int E, int F, int SEL, bool CROSS, int S, int CS>
__device__ __forceinline__ void quad_transform(const double (&m)[DI], const double (&L)[DI * (DI + 1) / 2], double t, const FPar &fp, double *lds,
                                               bool writer, const int (&col_off)[S], const double (&sgn)[S], const double (&wm)[S],
                                               const double (&wc)[S], cdouble_p cadd, double (&mf)[E], double (&cv)[E * (E + 1) / 2],
                                               double (&cx)[E][DI]) {
    using Fun = Fn<F>;
    constexpr int DIN = Fun::DIN;
    // the factor, dense and column-major with pitch CS (the zeros above the diagonal were written once, before the time loop)
    if (writer) {
#pragma unroll
        for (int k = 0; k < DI; ++k)
#pragma unroll
            for (int d = k; d < DI; ++d) lds[k * CS + d] = L[SSMQ_PK(d, k)];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    Fun fn;
    fn.init(t, fp);
    double fx[S][E], dx[S][DI];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double x[DI], xs[DIN];
#pragma unroll
        for (int d = 0; d < DI; ++d) {
            const double c = lds[col_off[s] + d];
            x[d] = fma(c, sgn[s], m[d]);
            if (CROSS) dx[s][d] = x[d] - m[d];       // x_n - mean as the reference forms it: (mean + L xi_n) - mean  (mtran.py:139,148)
        }
        select_inputs<DI, DIN, SEL>(x, xs);
        fn.template eval<E>(xs, fx[s]);
    }
    __builtin_amdgcn_wave_barrier();                  // (the record is rewritten by the next transform: every read above is done)
    // mean
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double p = fx[0][e] * wm[0];
#pragma unroll
        for (int s = 1; s < S; ++s) p = fma(fx[s][e], wm[s], p);
        mf[e] = p;
    }
    quad_sum_all<E>(mf);
    // centred, weighted
    double fw[S][E];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            fx[s][e] -= mf[e];
            fw[s][e] = fx[s][e] * wc[s];
        }
    // covariance and cross-covariance: the partial sums of every entry, one all-reduce for all of them
    constexpr int NC = E * (E + 1) / 2, NX = CROSS ? E * DI : 0;
    double red[NC + NX];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int e2 = 0; e2 <= e; ++e2) {
            double p = fw[0][e] * fx[0][e2];
#pragma unroll
            for (int s = 1; s < S; ++s) p = fma(fw[s][e], fx[s][e2], p);
            red[SSMQ_PK(e, e2)] = p;
        }
    if (CROSS) {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int d = 0; d < DI; ++d) {
                double p = fw[0][e] * dx[0][d];
#pragma unroll
                for (int s = 1; s < S; ++s) p = fma(fw[s][e], dx[s][d], p);
                red[NC + e * DI + d] = p;
            }
    }
    quad_sum_all<NC + NX>(red);
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int e2 = 0; e2 <= e; ++e2) cv[SSMQ_PK(e, e2)] = red[SSMQ_PK(e, e2)] + cadd[e * E + e2];
    if (CROSS) {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int d = 0; d < DI; ++d) cx[e][d] = red[NC + e * DI + d];
    }
}

template <int D, int Y, int FD, int FO, int SELO>
__global__ __launch_bounds__(kSmallBlock, 1) void k_filter_quad(const FusedArgs a) {      // (one wave per SIMD is the regime: all 512 registers)
    constexpr int N = 2 * D + 1, S = (N + 3) / 4, NP = D * (D + 1) / 2;
    constexpr int CS = D + (D & 1);                    // column pitch (even: 16-byte aligned columns)
    constexpr int ROW = (D * CS) | 1;                  // doubles per trajectory record: odd, so that the 16 records start on different banks
    __shared__ double s_rec[kQuadTraj][ROW];
    const int lane = threadIdx.x, q = lane & 3, tr = lane >> 2;
    const int64_t b = (int64_t)blockIdx.x * kQuadTraj + tr;
    const bool valid = b < a.B;
    const int64_t bb = valid ? b : a.B - 1;            // lanes beyond the batch shadow its last trajectory and store nothing
    const int64_t ld = a.ld;
    const bool writer = q == 0;
    double *lds = &s_rec[tr][0];
    if (writer) {
#pragma unroll
        for (int i = 0; i < ROW; ++i) lds[i] = 0.0;
    }
    // the lane's points: n = 4 s + q; n = 0 the centre, 1 .. D: + c L[:, n - 1], D + 1 .. 2 D: - c L[:, n - D - 1]; n >= N: nothing
    constexpr ConstLayout cld = const_layout(D, D, N, SSMQ_FORM_SIGMA), clo = const_layout(D, Y, N, SSMQ_FORM_SIGMA);
    const double *cdv = a.c_dyn, *cov_ = a.c_obs;
    int col_off[S];
    double sgn_d[S], sgn_o[S], wm_d[S], wc_d[S], wm_o[S], wc_o[S];
    const double utc_d = cdv[cld.utc], utc_o = cov_[clo.utc];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = 4 * s + q;
        const bool real = n < N, centre = n == 0;
        const int k = (!real || centre) ? 0 : (n - 1) % D;
        col_off[s] = k * CS;
        const double sg = (!real || centre) ? 0.0 : (n <= D ? 1.0 : -1.0);
        sgn_d[s] = sg * utc_d;
        sgn_o[s] = sg * utc_o;
        const int nn = real ? n : 0;
        wm_d[s] = real ? cdv[cld.wm + nn] : 0.0;
        wc_d[s] = real ? cdv[cld.Wc + nn] : 0.0;
        wm_o[s] = real ? cov_[clo.wm + nn] : 0.0;
        wc_o[s] = real ? cov_[clo.Wc + nn] : 0.0;
    }
    double m[D], Pl[NP];
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = a.m0[d * ld + bb];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Pl[SSMQ_PK(i, j)] = a.P0[(i * D + j) * ld + bb];
    // (none of the instantiated models has a per-step time table: their integrand constants are read from the kernel arguments
    // where they are used - copies held across the loop cost 60 spilled scalar registers)
    static_assert(!HasTimeTable<FD>::value && !HasTimeTable<FO>::value, "k_filter_quad: time-table integrands are not wired");
    const FPar &fd = a.fd, &fo = a.fo;
    const cdouble_p gqg = (cdouble_p)a.gqg, rr = (cdouble_p)a.rr;
    uint32_t off_m[D], off_P[D * D];              // byte offsets of this lane's entries inside one step's planes (< 4 GB: D^2 ld doubles)
#pragma unroll
    for (int d = 0; d < D; ++d) off_m[d] = (uint32_t)(((int64_t)d * ld + bb) * 8);
#pragma unroll
    for (int i = 0; i < D * D; ++i) off_P[i] = (uint32_t)(((int64_t)i * ld + bb) * 8);
    const double nan = __builtin_nan("");
    int32_t agg = 0;
    double ynext[Y];
#pragma unroll
    for (int i = 0; i < Y; ++i) ynext[i] = a.y[((int64_t)0 * Y + i) * ld + bb];
#pragma unroll 1
    for (int k = 0; k < a.T; ++k) {
        const double t = (double)k;          // both transforms of step k + 1 use time index k (ssinf.py:104, 276-288)
        double ycur[Y];
#pragma unroll
        for (int i = 0; i < Y; ++i) ycur[i] = ynext[i];
        {
            const int kn = (k + 1 < a.T) ? k + 1 : k;
#pragma unroll
            for (int i = 0; i < Y; ++i) ynext[i] = a.y[((int64_t)kn * Y + i) * ld + bb];
        }
        // G Q G' / R are re-read from the scalar cache in the step that adds them: hoisted out of the loop their D^2 + Y^2 values
        // overflow the scalar registers and come back as ~100 v_readlane per step
        const cdouble_p gqg_k = launder(gqg), rr_k = launder(rr);
        // ---- time update: predictive state moments, + G Q G' (ssinf.py:276-279) ------------------------------------------
        bool ok = chol_packed<D>(Pl);
        double pm[D], pP[NP], dummy[D][D];
        quad_transform<D, D, FD, 0, false, S, CS>(m, Pl, t, fd, lds, writer, col_off, sgn_d, wm_d, wc_d, gqg_k, pm, pP, dummy);
        // ---- predictive measurement moments, + R (ssinf.py:287-291) --------------------------------------------------------
        double L2[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) L2[i] = pP[i];
        ok = chol_packed<D>(L2) && ok;
        double ym[Y], Sy[Y * (Y + 1) / 2], cx[Y][D];
        quad_transform<D, Y, FO, SELO, true, S, CS>(pm, L2, t, fo, lds, writer, col_off, sgn_o, wm_o, wc_o, rr_k, ym, Sy, cx);
        // ---- measurement update (ssinf.py:321-323), as k_filter_fused forms it --------------------------------------------
        double Sf[Y * (Y + 1) / 2];
#pragma unroll
        for (int i = 0; i < Y * (Y + 1) / 2; ++i) Sf[i] = Sy[i];
        double G[D][Y];
        if (Y == 1) ok = (Sf[0] > 0.0) && ok;
        else ok = chol_packed<Y>(Sf) && ok;
        if (agg == 0 && !ok) agg = k + 1;
        // From the first failing step on every result is NaN (the reference raises there).  ONE poisoned row of the cross-covariance
        // does it: the gain, and with it the mean and every covariance entry of this and all later steps, inherit the NaN - D
        // additions instead of two selects on each of the D + D^2 results.
        {
            const double pois = agg == 0 ? 0.0 : nan;
#pragma unroll
            for (int d = 0; d < D; ++d) cx[0][d] += pois;
        }
        if (Y == 1) {
#pragma unroll
            for (int d = 0; d < D; ++d) G[d][0] = div_nr(cx[0][d], Sf[0]);
        } else {
            // the two substitutions of cho_solve with the pivots' reciprocals formed once (Y divisions instead of 2 Y per state
            // coordinate; a product with the correctly rounded reciprocal is within an ulp of the quotient k_filter_fused forms)
            double ri[Y];
#pragma unroll
            for (int i = 0; i < Y; ++i) ri[i] = div_nr(1.0, Sf[SSMQ_PK(i, i)]);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double v[Y];
#pragma unroll
                for (int i = 0; i < Y; ++i) {
                    double s = cx[i][d];
#pragma unroll
                    for (int r = 0; r < i; ++r) s -= Sf[SSMQ_PK(i, r)] * v[r];
                    v[i] = s * ri[i];
                }
#pragma unroll
                for (int i = Y - 1; i >= 0; --i) {
                    double s = v[i];
#pragma unroll
                    for (int r = i + 1; r < Y; ++r) s -= Sf[SSMQ_PK(r, i)] * v[r];
                    v[i] = s * ri[i];
                }
#pragma unroll
                for (int i = 0; i < Y; ++i) G[d][i] = v[i];
            }
        }
        const bool st = writer && valid;
        double pout[D][D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < Y; ++i) s += G[d][i] * (ycur[i] - ym[i]);
            m[d] = pm[d] + s;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double w[Y];
#pragma unroll
            for (int j = 0; j < Y; ++j) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < Y; ++i) s += G[d][i] * Sy[i >= j ? SSMQ_PK(i, j) : SSMQ_PK(j, i)];
                w[j] = s;
            }
            // the lower triangle, mirrored: K P_y K' is symmetric; the reference's product leaves the two triangles an ulp apart
            // (ssinf.py:323), which only a bit-for-bit kernel has to reproduce
#pragma unroll
            for (int d2 = 0; d2 <= d; ++d2) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < Y; ++j) s += w[j] * G[d2][j];
                const double p = pP[SSMQ_PK(d, d2)] - s;
                pout[d][d2] = p;
                pout[d2][d] = p;
                Pl[SSMQ_PK(d, d2)] = p;
            }
        }
        if (st) {          // the quad's first lane stores the step's results: ONE masked region for the D + D^2 stores; addresses =
                           // the step's plane base (scalar) + a per-lane offset formed once before the loop
            char *bm = (char *)a.fm + (int64_t)k * D * ld * 8, *bP = (char *)a.fP + (int64_t)k * D * D * ld * 8;
#pragma unroll
            for (int d = 0; d < D; ++d) SSMQ_STORE(*(double *)(bm + off_m[d]), m[d]);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int d2 = 0; d2 < D; ++d2) SSMQ_STORE(*(double *)(bP + off_P[d * D + d2]), pout[d][d2]);
        }
    }
    if (writer && valid) a.status[b] = agg;
}

typedef void (*quad_kernel)(const FusedArgs);
struct QuadEntry {
    int fd, fo, D, Y, selo;
    quad_kernel k;
    const char *name;
};
#define SSMQ_QD(FD, FO, D, Y, SELO) \
    {FD, FO, D, Y, SELO, &k_filter_quad<D, Y, FD, FO, SELO>, "k_filter_quad<D=" #D ",Y=" #Y "," #FD "," #FO ",SSMQ_FORM_SIGMA,SELO=" #SELO ">"}
const QuadEntry kQuad[] = {
    SSMQ_QD(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 0),
    SSMQ_QD(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 0),
    SSMQ_QD(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 1),
};

}  // namespace

// 1: launched (dry_run: would be); 0: this filter / batch keeps its other routes; < 0: error.
// Taken when the batch's whole-pass waves would each have a SIMD to themselves and the quad waves still do: ceil(B / 16) <= SIMDs
// (B <= 16 384 on 256 compute units).  SSMQ_FUSED_QUAD=0 never, =1 for any batch (tests).
int try_launch_quad(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho, const ssmq_integrand *fo, int sel_obs,
                    int64_t B, int64_t ld, int T, const double *d_y, const double *d_m0, const double *d_P0, const double *d_gqg,
                    const double *d_rr, double *d_fm, double *d_fP, int32_t *d_status, hipStream_t s, const char **name, bool dry_run,
                    const double *d_sscale, double student_dof, int cus) {
    const char *ev = ssmq::sw("SSMQ_FUSED_QUAD");
    const int force = ev ? atoi(ev) : -1;
    if (force == 0 || d_sscale || student_dof > 0.0 || B <= 0) return 0;
    if (!dry_run && ctx().no_strips) return 0;       // a job of a multi-filter launch shares the chip: whole-pass kernels only
    if (force != 1 && ssmq::sw("SSMQ_FUSED_WSPLIT")) return 0;          // a forced wave-split mode (0 = the register kernel) is what runs
    if (hd->form != SSMQ_FORM_SIGMA || ho->form != SSMQ_FORM_SIGMA || hd->tp_nu > 0.0 || ho->tp_nu > 0.0 || sel_obs < 0 || fd->n_idx > 0) return 0;
    if (!(hd->opt_mask & ho->opt_mask & SSMQ_OPT_UT)) return 0;          // unscented-type points [0 | c I | -c I], verified on the host
    if (hd->N != 2 * hd->D + 1 || ho->N != hd->N || ho->D != hd->D || hd->E != hd->D) return 0;
    if ((int64_t)hd->D * hd->D * ld * 8 >= ((int64_t)1 << 32)) return 0;        // (the kernel's 32-bit store offsets inside one step's planes)
    const int64_t waves = (B + kQuadTraj - 1) / kQuadTraj;
    if (force != 1 && waves > 4 * (int64_t)cus) return 0;
    for (const QuadEntry &e : kQuad) {
        if (!(e.fd == fd->id && e.fo == fo->id && e.D == hd->D && e.Y == ho->E && e.selo == sel_obs)) continue;
        if (name) *name = e.name;
        if (dry_run) return 1;
        FusedArgs a;
        memset(&a, 0, sizeof(a));
        a.y = d_y; a.m0 = d_m0; a.P0 = d_P0; a.fm = d_fm; a.fP = d_fP; a.status = d_status;
        a.c_dyn = hd->d_small; a.c_obs = ho->d_small; a.gqg = d_gqg; a.rr = d_rr; a.B = B; a.ld = ld; a.T = T;
        a.emv_dyn = hd->emv_mode; a.emv_obs = ho->emv_mode; a.lpw = kQuadTraj;
        fill_fpar(fd, &a.fd);
        fill_fpar(fo, &a.fo);
        hipLaunchKernelGGL(e.k, dim3((unsigned)waves), dim3(kSmallBlock), 0, s, a);
        const int rc = hip_fail(hipGetLastError(), e.name);
        return rc ? rc : 1;
    }
    return 0;
}

}  // namespace ssmq
