// Fused filter time loop: ONE kernel runs all T steps of an additive-noise Gaussian sigma-point / BQ filter for a batch of
// independent trajectories, one trajectory per lane, the filter state (mean, covariance) held in registers from step to
// step (reference recursion: ssinf.py:66-118, 254-323).  Per step and trajectory it reads the measurement (Y doubles)
// and writes the filtered mean and the full, unsymmetrised covariance (D + D*D doubles) - the per-step HBM traffic the
// reference's forward_pass implies (bytes_step = 8 (dim_y + D + D^2)); the predictive moments never
// leave the register file and the 3 T kernel launches of the unfused loop collapse into one.
//
// The input-state cross-covariance of the dynamics transform is only consumed by the smoother (ssinf.py:105-107,
// 325-344) and is not formed in the forward pass.
#include <cstring>
#include <vector>
#include "ssmq_filter_fused_kernel.h"

namespace ssmq {

template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int OPT, int STU>
static hipError_t launch_fused(const FusedArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + a.lpw - 1) / a.lpw);
    hipLaunchKernelGGL((k_filter_fused<D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, STU>), dim3(grid), dim3(kSmallBlock), 0, s, a);
    return hipGetLastError();
}

typedef hipError_t (*fused_fn)(const FusedArgs &, hipStream_t);
struct FusedEntry {
    int fd, fo, D, Y, ND, NO, form, tp, selo, opt;
    fused_fn fn[2];   // [Studentian]; shapes with the recursion type decided at run time fill fn[0] only
    const char *name;
};

#define SSMQ_FUSED_NAME(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT)                                              \
    "k_filter_fused<D=" #D ",Y=" #Y ",ND=" #ND ",NO=" #NO "," #FD "," #FO "," #FORM ",TP=" #TP ",SELO=" #SELO \
    ",OPT=" #OPT ">"
#define SSMQ_FUSED_FN(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT, STU) \
    &launch_fused<D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, STU>
#define SSMQ_FUSED_ONE(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT)                                        \
    {FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT,                                                         \
     {SSMQ_FUSED_FN(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT, -1), nullptr},                           \
     SSMQ_FUSED_NAME(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT)}
// scalar state: recursion type fixed at compile time
#define SSMQ_FUSED_ONE_S(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT)                                      \
    {FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT,                                                         \
     {SSMQ_FUSED_FN(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT, 0),                                      \
      SSMQ_FUSED_FN(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT, 1)},                                     \
     SSMQ_FUSED_NAME(FD, FO, D, Y, ND, NO, FORM, TP, SELO, OPT)}
#define SSMQ_FUSED(FD, FO, D, Y, N, SELO)                                      \
    SSMQ_FUSED_ONE(FD, FO, D, Y, N, N, SSMQ_FORM_BQ, 0, SELO, 0),              \
    SSMQ_FUSED_ONE(FD, FO, D, Y, N, N, SSMQ_FORM_BQ, 1, SELO, 0),              \
    SSMQ_FUSED_ONE(FD, FO, D, Y, N, N, SSMQ_FORM_SIGMA, 0, SELO, 0)
// scalar-state models (D = Y = 1)
#define SSMQ_FUSED_S(FD, FO, N)                                                \
    SSMQ_FUSED_ONE_S(FD, FO, 1, 1, N, N, SSMQ_FORM_BQ, 0, 0, 0),               \
    SSMQ_FUSED_ONE_S(FD, FO, 1, 1, N, N, SSMQ_FORM_BQ, 1, 0, 0),               \
    SSMQ_FUSED_ONE_S(FD, FO, 1, 1, N, N, SSMQ_FORM_SIGMA, 0, 0, 0)
// larger shapes: also with the LDL' / unscented-point fast paths of ssmq_apply_small.h
#define SSMQ_FUSED_FAST(FD, FO, D, Y, N, SELO)                                 \
    SSMQ_FUSED(FD, FO, D, Y, N, SELO),                                         \
    SSMQ_FUSED_ONE(FD, FO, D, Y, N, N, SSMQ_FORM_BQ, 0, SELO, 7),              \
    SSMQ_FUSED_ONE(FD, FO, D, Y, N, N, SSMQ_FORM_BQ, 0, SELO, 3),              \
    SSMQ_FUSED_ONE(FD, FO, D, Y, N, N, SSMQ_FORM_BQ, 1, SELO, 2),              \
    SSMQ_FUSED_ONE(FD, FO, D, Y, N, N, SSMQ_FORM_SIGMA, 0, SELO, 2)

static bool HasTimeTableRT(int fid) { return fid == SSMQ_F_UNGM_DYN || fid == SSMQ_F_UNGMNA_DYN; }

static const FusedEntry kFused[] = {
    SSMQ_FUSED_S(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 2),
    SSMQ_FUSED_S(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 3),
    SSMQ_FUSED_S(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 5),
#ifndef SSMQ_FUSED_UNGM_ONLY   // tools/build_variant.sh: quick builds for A/B timing of the UNGM kernels
    SSMQ_FUSED(SSMQ_F_PENDULUM_DYN, SSMQ_F_PENDULUM_MEAS, 2, 1, 5, 0),
    SSMQ_FUSED(SSMQ_F_REENTRY1D_DYN, SSMQ_F_RANGE_MEAS, 3, 1, 7, 0),          // tests/test_ssinf.py:40-50 of the reference
    SSMQ_FUSED(SSMQ_F_CV_DYN, SSMQ_F_RADAR2D_MEAS, 4, 2, 9, 0),               // constant velocity + radar (Student filters)
    SSMQ_FUSED_FAST(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0),
    SSMQ_FUSED_FAST(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0),
    SSMQ_FUSED_FAST(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 11, 1),
    // spherical-radial point sets (2 D points: the cubature Kalman filter and every BQ transform built with 'sr')
    SSMQ_FUSED(SSMQ_F_PENDULUM_DYN, SSMQ_F_PENDULUM_MEAS, 2, 1, 4, 0),
    SSMQ_FUSED(SSMQ_F_REENTRY1D_DYN, SSMQ_F_RANGE_MEAS, 3, 1, 6, 0),
    SSMQ_FUSED(SSMQ_F_CV_DYN, SSMQ_F_RADAR2D_MEAS, 4, 2, 8, 0),
    SSMQ_FUSED(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 10, 0),
    SSMQ_FUSED(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 12, 0),
    SSMQ_FUSED(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 10, 1),
#endif
};

// ---- several filters of ONE model family in one launch (round 6; ssmq_filter_forward_multi_dev) -------------------------------
// A graph with one branch per filter still reaches the device as one dispatch after the other (measured: six configs[1]-sized UNGM
// filters 160 us as a forked graph, 187 us as six streams, 31 us each - the launch path serialises them).  The filters of the
// reference's UNGM studies (research/bsq/bsq_ungm.py:132-137, research/tpq/tpq_base.py:175-192: UKF, CKF, GHKF, GPQKF, TPQKF,
// BSQKF) differ only in point count and form, so they fit ONE kernel: the grid is the concatenation of the jobs' blocks, a block
// looks its job up in a table in memory and runs that job's time loop - the same fused_pass<> instantiation k_filter_fused runs,
// hence the same bits.  942 waves for six filters of 1e4 trajectories: every one has a SIMD to itself.
// The job table travels in the kernel-argument segment (constant memory: a block reads its job's fields with scalar loads on
// demand - a table in global memory, copied into a local FusedArgs, cost 495 spilled SGPRs): only what the UNGM time loop reads.
struct MultiJob {
    const double *y, *m0, *P0, *c_dyn, *c_obs, *gqg, *rr, *ttd, *tto;
    double *fm, *fP;
    int32_t *status;
    int64_t B, ld;
    int32_t T, emv_dyn, emv_obs, kind;
    double nu_dyn, nu_obs;
    int32_t first, pad;          // first block of the job in the grid
};
constexpr int kMultiMaxJobs = 24;
struct MultiKernArgs {
    int32_t n, blocks;
    MultiJob job[kMultiMaxJobs];
};
static_assert(sizeof(MultiKernArgs) <= 4096, "kernel-argument segment");

// One NON-INLINED function per kind: nine time loops inlined into one kernel body shared one scalar-register allocation and
// spilled 500 SGPRs (to vector-register lanes, read back inside the loops); as functions each has the allocation of its own
// whole-pass kernel.  The job record is read through the constant address space (scalar loads).
typedef const __attribute__((address_space(4))) MultiJob *multi_job_p;
template <int N, int FORM, int TP>
__device__ __attribute__((noinline)) void multi_case(multi_job_p q, uint32_t blk) {
    FusedArgs a;
    a.y = q->y; a.m0 = q->m0; a.P0 = q->P0; a.fm = q->fm; a.fP = q->fP; a.status = q->status;
    a.c_dyn = q->c_dyn; a.c_obs = q->c_obs; a.gqg = q->gqg; a.rr = q->rr; a.B = q->B; a.ld = q->ld; a.T = q->T;
    a.emv_dyn = q->emv_dyn; a.emv_obs = q->emv_obs; a.lpw = 64; a.nu_dyn = q->nu_dyn; a.nu_obs = q->nu_obs;
    a.sscale = nullptr; a.student_dof = 0.0;
    a.fd.n_par = a.fd.n_idx = a.fo.n_par = a.fo.n_idx = 0;
    a.fd.ttab = q->ttd; a.fd.tval = 0.0; a.fd.use_tval = 0;
    a.fo.ttab = q->tto; a.fo.tval = 0.0; a.fo.use_tval = 0;
    a.t_chunk = a.n_blocks = 0; a.queue = nullptr; a.hand = nullptr;
    fused_pass<1, 1, N, N, SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, FORM, TP, 0, 0, 0, false>(a, blk, 0, a.T, true, true);
}

__global__ __launch_bounds__(kSmallBlock, 2) void k_filter_multi_ungm(const MultiKernArgs) {
    const __attribute__((address_space(4))) MultiKernArgs *m =
        (const __attribute__((address_space(4))) MultiKernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    int j = 0;
    while (j + 1 < m->n && (int)blockIdx.x >= m->job[j + 1].first) ++j;
    multi_job_p q = &m->job[j];
    const uint32_t blk = blockIdx.x - (uint32_t)q->first;
    switch (q->kind) {
        case 0: multi_case<2, SSMQ_FORM_BQ, 0>(q, blk); break;
        case 1: multi_case<2, SSMQ_FORM_BQ, 1>(q, blk); break;
        case 2: multi_case<2, SSMQ_FORM_SIGMA, 0>(q, blk); break;
        case 3: multi_case<3, SSMQ_FORM_BQ, 0>(q, blk); break;
        case 4: multi_case<3, SSMQ_FORM_BQ, 1>(q, blk); break;
        case 5: multi_case<3, SSMQ_FORM_SIGMA, 0>(q, blk); break;
        case 6: multi_case<5, SSMQ_FORM_BQ, 0>(q, blk); break;
        case 7: multi_case<5, SSMQ_FORM_BQ, 1>(q, blk); break;
        case 8: multi_case<5, SSMQ_FORM_SIGMA, 0>(q, blk); break;
        default: break;
    }
}

// Host side: 1 = every job is a Gaussian-recursion UNGM filter with 2 / 3 / 5 points, at most kMultiMaxJobs of them (the kernel
// arguments are written to `table`, the grid size to *blocks); 0 = not one family (the caller forks a graph instead).
int multi_family_table(int n, const ssmq_transform *const *hd, const ssmq_integrand *const *fd, const ssmq_transform *const *ho,
                       const ssmq_integrand *const *fo, const FusedArgs *args, std::vector<char> *table, int *blocks) {
    if (n < 1 || n > kMultiMaxJobs) return 0;
    MultiKernArgs m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    int64_t first = 0;
    for (int i = 0; i < n; ++i) {
        const FusedArgs &a = args[i];
        if (fd[i]->id != SSMQ_F_UNGM_DYN || fo[i]->id != SSMQ_F_UNGM_MEAS || hd[i]->D != 1 || ho[i]->E != 1 || hd[i]->N != ho[i]->N ||
            hd[i]->form != ho[i]->form || (hd[i]->tp_nu > 0.0) != (ho[i]->tp_nu > 0.0) || a.sscale != nullptr || a.student_dof > 0.0 ||
            fd[i]->n_idx > 0 || fo[i]->n_idx > 0 || !a.fd.ttab || a.lpw != 64)
            return 0;
        const int N = hd[i]->N, in = N == 2 ? 0 : (N == 3 ? 1 : (N == 5 ? 2 : -1));
        if (in < 0) return 0;
        MultiJob &q = m.job[i];
        q.y = a.y; q.m0 = a.m0; q.P0 = a.P0; q.c_dyn = a.c_dyn; q.c_obs = a.c_obs; q.gqg = a.gqg; q.rr = a.rr; q.ttd = a.fd.ttab; q.tto = a.fo.ttab;
        q.fm = a.fm; q.fP = a.fP; q.status = a.status; q.B = a.B; q.ld = a.ld; q.T = a.T; q.emv_dyn = a.emv_dyn; q.emv_obs = a.emv_obs;
        q.kind = 3 * in + (hd[i]->form == SSMQ_FORM_SIGMA ? 2 : (hd[i]->tp_nu > 0.0 ? 1 : 0));
        q.nu_dyn = a.nu_dyn; q.nu_obs = a.nu_obs;
        q.first = (int32_t)first;
        first += (a.B + 63) / 64;
        if (first > (1 << 30)) return 0;
    }
    m.blocks = (int32_t)first;
    table->assign(sizeof(m), 0);
    memcpy(table->data(), &m, sizeof(m));
    *blocks = m.blocks;
    return 1;
}
// `table`: the HOST copy multi_family_table() produced (kernel arguments are captured at launch)
int multi_family_launch(const char *table, int blocks, hipStream_t s) {
    if (blocks <= 0) return SSMQ_OK;
    MultiKernArgs m;
    memcpy(&m, table, sizeof(m));
    hipLaunchKernelGGL(k_filter_multi_ungm, dim3((unsigned)blocks), dim3(kSmallBlock), 0, s, m);
    return hip_fail(hipGetLastError(), "k_filter_multi_ungm");
}

// ---- models that take their noise as an argument (ssinf.py:271-272, 282-283, 294-295) -------------------------------------
// Same time loop with the inputs of either transform augmented in registers: [m; noise_mean], blockdiag(P, noise_cov)
// (the Cholesky of a block-diagonal matrix is block-diagonal, so the augmented factor costs nothing extra), and the
// measurement cross-covariance cut back to the state columns.  A separate kernel so that the additive-noise kernels
// above - the measured ones - compile exactly as before.  DQ / DR = 0 means that model is additive (G Q G' / R added).
struct AugArgs {
    const double *y, *m0, *P0;
    double *fm, *fP;
    int32_t *status;
    const double *c_dyn, *c_obs;
    const double *add_dyn, *add_obs;   // [D*D] / [Y*Y]: G Q G' / R for an additive model, zeros otherwise
    const double *noise;               // q_mean[DQ] | q_cov[DQ*DQ] | r_mean[DR] | r_cov[DR*DR]
    double *pm, *pP, *pC;              // KEEP: predictive mean / covariance / dynamics cross-covariance of every step
                                       // ([T][D][ld], [T][D*D][ld] x 2) for the RTS pass (ssinf.py:105-107)
    int64_t B, ld;
    int32_t T, emv_dyn, emv_obs;
    double nu_dyn, nu_obs;
    FPar fd, fo;
};

template <int D, int DN>
__device__ __forceinline__ void augment(const double (&m)[D], const double *Pl, cdouble_p nmean, cdouble_p ncov,
                                        double (&ma)[D + DN], double (&Pa)[(D + DN) * (D + DN + 1) / 2]) {
#pragma unroll
    for (int i = 0; i < D + DN; ++i) {
        ma[i] = i < D ? m[i < D ? i : 0] : nmean[i - D];
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double v = 0.0;
            if (i < D) v = Pl[SSMQ_PK(i, j)];
            else if (j >= D) v = ncov[(i - D) * DN + (j - D)];
            Pa[SSMQ_PK(i, j)] = v;
        }
    }
}

template <int D, int Y, int DQ, int DR, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, bool KEEP = false>
__global__ __launch_bounds__(kSmallBlock, (D + DQ >= 5 ? 1 : 2)) void k_filter_fused_aug(const AugArgs a) {
    constexpr int DA = D + DQ, DO = D + DR;
    const uint32_t b = blockIdx.x * kSmallBlock + threadIdx.x;
    if ((int64_t)b >= a.B) return;
    const int64_t ld = a.ld;
    double m[D], Pl[D * (D + 1) / 2];
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = a.m0[d * ld + b];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Pl[SSMQ_PK(i, j)] = a.P0[(i * D + j) * ld + b];
    const CoreParams cpd{(cdouble_p)a.c_dyn, (cdouble_p)a.add_dyn, a.emv_dyn, a.nu_dyn, 1.0, 1.0};
    const CoreParams cpo{(cdouble_p)a.c_obs, (cdouble_p)a.add_obs, a.emv_obs, a.nu_obs, 1.0, 1.0};
    const cdouble_p qm = (cdouble_p)a.noise, qc = qm + DQ, rm = qc + DQ * DQ, rc = rm + DR;
    const double nan = __builtin_nan("");
    int32_t agg = 0;
    // the initial moments have to have arrived before the loop is entered (see k_filter_fused)
#pragma unroll
    for (int d = 0; d < D; ++d) pin_v(m[d]);
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) pin_v(Pl[i]);
#pragma unroll 1
    for (int k = 0; k < a.T; ++k) {
        const double t = (double)k;   // both transforms of step k + 1 use time index k (ssinf.py:104, 276-288)
        double yk[Y];
#pragma unroll
        for (int i = 0; i < Y; ++i) yk[i] = a.y[((int64_t)k * Y + i) * ld + b];
        double ma[DA], Pa[DA * (DA + 1) / 2];
        augment<D, DQ>(m, Pl, qm, qc, ma, Pa);
        typename std::conditional<KEEP, RegSink<DA, D>, RegSinkNoCross<DA, D>>::type pr;
        bool ok = moment_transform_core<DA, D, ND, FD, FORM, TP, 0, KEEP, 0>(ma, Pa, t, a.fd, cpd, pr);
        if constexpr (KEEP) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                SSMQ_STORE(a.pm[((int64_t)k * D + d) * ld + b], pr.mf[d]);
#pragma unroll
                for (int d2 = 0; d2 < D; ++d2) {
                    SSMQ_STORE(a.pP[((int64_t)k * D * D + d * D + d2) * ld + b], pr.cv[d >= d2 ? SSMQ_PK(d, d2) : SSMQ_PK(d2, d)]);
                    SSMQ_STORE(a.pC[((int64_t)k * D * D + d * D + d2) * ld + b], pr.cx[d][d2]);
                }
            }
        }
        double mo[DO], Po[DO * (DO + 1) / 2];
        augment<D, DR>(pr.mf, pr.cv, rm, rc, mo, Po);
        RegSink<DO, Y> ob;
        ok = moment_transform_core<DO, Y, NO, FO, FORM, TP, SELO, true, 0>(mo, Po, t, a.fo, cpo, ob) && ok;
        double S[Y * (Y + 1) / 2];
#pragma unroll
        for (int i = 0; i < Y * (Y + 1) / 2; ++i) S[i] = ob.cv[i];
        double G[D][Y];
        if (Y == 1) {
            ok = (S[0] > 0.0) && ok;
#pragma unroll
            for (int d = 0; d < D; ++d) G[d][0] = div_nr(ob.cx[0][d], S[0]);   // state columns only (ssinf.py:294)
        } else {
            ok = chol_packed<Y>(S) && ok;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double v[Y];
#pragma unroll
                for (int i = 0; i < Y; ++i) {
                    double s = ob.cx[i][d];
#pragma unroll
                    for (int q = 0; q < i; ++q) s -= S[SSMQ_PK(i, q)] * v[q];
                    v[i] = div_nr(s, S[SSMQ_PK(i, i)]);
                }
#pragma unroll
                for (int i = Y - 1; i >= 0; --i) {
                    double s = v[i];
#pragma unroll
                    for (int q = i + 1; q < Y; ++q) s -= S[SSMQ_PK(q, i)] * v[q];
                    v[i] = div_nr(s, S[SSMQ_PK(i, i)]);
                }
#pragma unroll
                for (int i = 0; i < Y; ++i) G[d][i] = v[i];
            }
        }
        if (agg == 0 && !ok) agg = k + 1;
        const bool good = (agg == 0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < Y; ++i) s += G[d][i] * (yk[i] - ob.mf[i]);
            m[d] = good ? pr.mf[d] + s : nan;
            SSMQ_STORE(a.fm[((int64_t)k * D + d) * ld + b], m[d]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double w[Y];
#pragma unroll
            for (int j = 0; j < Y; ++j) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < Y; ++i) s += G[d][i] * ob.cv[i >= j ? SSMQ_PK(i, j) : SSMQ_PK(j, i)];
                w[j] = s;
            }
#pragma unroll
            for (int d2 = 0; d2 < D; ++d2) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < Y; ++j) s += w[j] * G[d2][j];
                double p = pr.cv[d >= d2 ? SSMQ_PK(d, d2) : SSMQ_PK(d2, d)] - s;
                p = good ? p : nan;
                SSMQ_STORE(a.fP[((int64_t)k * D * D + d * D + d2) * ld + b], p);
                if (d2 <= d) Pl[SSMQ_PK(d, d2)] = p;
            }
        }
    }
    a.status[b] = agg;
}

template <int D, int Y, int DQ, int DR, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, bool KEEP>
static hipError_t launch_fused_aug(const AugArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + kSmallBlock - 1) / kSmallBlock);
    hipLaunchKernelGGL((k_filter_fused_aug<D, Y, DQ, DR, ND, NO, FD, FO, FORM, TP, SELO, KEEP>), dim3(grid), dim3(kSmallBlock), 0,
                       s, a);
    return hipGetLastError();
}

typedef hipError_t (*aug_fn)(const AugArgs &, hipStream_t);
struct AugEntry {
    int fd, fo, D, Y, DQ, DR, ND, NO, form, tp, selo, keep;
    aug_fn fn;
    const char *name;
};
#define SSMQ_AUG_ONE(FD, FO, D, Y, DQ, DR, ND, NO, FORM, TP, SELO)                                                        \
    {FD, FO, D, Y, DQ, DR, ND, NO, FORM, TP, SELO, 0, &launch_fused_aug<D, Y, DQ, DR, ND, NO, FD, FO, FORM, TP, SELO, false>, \
     "k_filter_fused_aug<D=" #D ",Y=" #Y ",DQ=" #DQ ",DR=" #DR ",ND=" #ND ",NO=" #NO "," #FD "," #FO "," #FORM ",TP=" #TP ">"}
// models that take their noise as an argument, predictive moments kept for the smoother (the cross-covariance is stored
// with its state columns only)
#define SSMQ_AUG_KEEP_ONE(FD, FO, D, Y, DQ, DR, ND, NO, FORM, TP, SELO)                                                      \
    {FD, FO, D, Y, DQ, DR, ND, NO, FORM, TP, SELO, 1, &launch_fused_aug<D, Y, DQ, DR, ND, NO, FD, FO, FORM, TP, SELO, true>, \
     "k_filter_fused_aug_keep<D=" #D ",Y=" #Y ",DQ=" #DQ ",DR=" #DR ",ND=" #ND ",NO=" #NO "," #FD "," #FO "," #FORM ",TP=" #TP ">"}
#define SSMQ_AUG_KEEP(FD, FO, D, Y, DQ, DR, ND, NO, SELO)                          \
    SSMQ_AUG_KEEP_ONE(FD, FO, D, Y, DQ, DR, ND, NO, SSMQ_FORM_BQ, 0, SELO),        \
    SSMQ_AUG_KEEP_ONE(FD, FO, D, Y, DQ, DR, ND, NO, SSMQ_FORM_BQ, 1, SELO),        \
    SSMQ_AUG_KEEP_ONE(FD, FO, D, Y, DQ, DR, ND, NO, SSMQ_FORM_SIGMA, 0, SELO)
// additive models, predictive moments kept for the smoother
#define SSMQ_KEEP_ONE(FD, FO, D, Y, N, FORM, TP, SELO)                                                            \
    {FD, FO, D, Y, 0, 0, N, N, FORM, TP, SELO, 1, &launch_fused_aug<D, Y, 0, 0, N, N, FD, FO, FORM, TP, SELO, true>, \
     "k_filter_fused_keep<D=" #D ",Y=" #Y ",N=" #N "," #FD "," #FO "," #FORM ",TP=" #TP ">"}
#define SSMQ_KEEP(FD, FO, D, Y, N, SELO)                     \
    SSMQ_KEEP_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO),   \
    SSMQ_KEEP_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO),   \
    SSMQ_KEEP_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO)
#define SSMQ_AUG(FD, FO, D, Y, DQ, DR, ND, NO, SELO)                          \
    SSMQ_AUG_ONE(FD, FO, D, Y, DQ, DR, ND, NO, SSMQ_FORM_BQ, 0, SELO),        \
    SSMQ_AUG_ONE(FD, FO, D, Y, DQ, DR, ND, NO, SSMQ_FORM_BQ, 1, SELO),        \
    SSMQ_AUG_ONE(FD, FO, D, Y, DQ, DR, ND, NO, SSMQ_FORM_SIGMA, 0, SELO)

static const AugEntry kAug[] = {
    SSMQ_AUG(SSMQ_F_UNGMNA_DYN, SSMQ_F_UNGMNA_MEAS, 1, 1, 1, 1, 4, 4, 0),      // spherical-radial points in 2-D
#ifndef SSMQ_FUSED_UNGM_ONLY
    SSMQ_AUG(SSMQ_F_UNGMNA_DYN, SSMQ_F_UNGMNA_MEAS, 1, 1, 1, 1, 5, 5, 0),      // unscented points in 2-D
    SSMQ_AUG_KEEP(SSMQ_F_UNGMNA_DYN, SSMQ_F_UNGMNA_MEAS, 1, 1, 1, 1, 4, 4, 0),
    SSMQ_AUG_KEEP(SSMQ_F_UNGMNA_DYN, SSMQ_F_UNGMNA_MEAS, 1, 1, 1, 1, 5, 5, 0),
    SSMQ_AUG_ONE(SSMQ_F_CTRS_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 2, 0, 15, 11, SSMQ_FORM_SIGMA, 0, 0),
    SSMQ_KEEP(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 1, 1, 2, 0),
    SSMQ_KEEP(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 1, 1, 3, 0),
    SSMQ_KEEP(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 1, 1, 5, 0),
    SSMQ_KEEP(SSMQ_F_PENDULUM_DYN, SSMQ_F_PENDULUM_MEAS, 2, 1, 5, 0),
    SSMQ_KEEP(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0),
    SSMQ_KEEP(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0),
    // spherical-radial point sets (cubature smoother)
    SSMQ_KEEP(SSMQ_F_PENDULUM_DYN, SSMQ_F_PENDULUM_MEAS, 2, 1, 4, 0),
    SSMQ_KEEP(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 10, 0),
#endif
};

// as try_launch_fused, for filters whose models take the noise as an argument; d_noise: q_mean | q_cov | r_mean | r_cov
int try_launch_fused_aug(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho,
                         const ssmq_integrand *fo, int sel_obs, int D, int dq, int dr, int64_t B, int64_t ld, int T,
                         const double *d_y, const double *d_m0, const double *d_P0, const double *d_add_dyn,
                         const double *d_add_obs, const double *d_noise, double *d_fm, double *d_fP, int32_t *d_status,
                         hipStream_t s, const char **name, bool dry_run, const double *d_ttab_dyn,
                         const double *d_ttab_obs, double *d_pm, double *d_pP, double *d_pC) {
    const int keep = (d_pm && d_pP && d_pC) ? 1 : 0;
    if (hd->form != ho->form || (hd->tp_nu > 0.0) != (ho->tp_nu > 0.0) || sel_obs < 0 || fd->n_idx > 0) return 0;
    const int tp = hd->tp_nu > 0.0 ? 1 : 0;
    for (const AugEntry &e : kAug) {
        if (e.fd == fd->id && e.fo == fo->id && e.D == D && e.Y == ho->E && e.DQ == dq && e.DR == dr && e.ND == hd->N &&
            e.NO == ho->N && e.form == hd->form && e.tp == tp && e.selo == sel_obs && e.keep == keep) {
            if (name) *name = e.name;
            if (dry_run) return 1;
            AugArgs a;
            a.pm = d_pm; a.pP = d_pP; a.pC = d_pC;
            a.y = d_y; a.m0 = d_m0; a.P0 = d_P0; a.fm = d_fm; a.fP = d_fP; a.status = d_status;
            a.c_dyn = hd->d_small; a.c_obs = ho->d_small; a.add_dyn = d_add_dyn; a.add_obs = d_add_obs;
            a.noise = d_noise; a.B = B; a.ld = ld; a.T = T; a.emv_dyn = hd->emv_mode; a.emv_obs = ho->emv_mode;
            a.nu_dyn = hd->tp_nu; a.nu_obs = ho->tp_nu;
            fill_fpar(fd, &a.fd);
            fill_fpar(fo, &a.fo);
            a.fd.ttab = d_ttab_dyn;
            a.fo.ttab = d_ttab_obs;
            int rc = hip_fail(e.fn(a, s), e.name);
            return rc ? rc : 1;
        }
    }
    return 0;
}

int try_launch_wsplit(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho, const ssmq_integrand *fo,
                      int sel_obs, int64_t B, int64_t ld, int T, const double *d_y, const double *d_m0, const double *d_P0,
                      const double *d_gqg, const double *d_rr, double *d_fm, double *d_fP, int32_t *d_status, hipStream_t s,
                      const char **name, bool dry_run, const double *d_sscale, double student_dof, int cus);

int try_launch_quad(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho, const ssmq_integrand *fo, int sel_obs,
                    int64_t B, int64_t ld, int T, const double *d_y, const double *d_m0, const double *d_P0, const double *d_gqg,
                    const double *d_rr, double *d_fm, double *d_fP, int32_t *d_status, hipStream_t s, const char **name, bool dry_run,
                    const double *d_sscale, double student_dof, int cus);

// ... and for batches of the heavy shapes that do not fill the chip evenly, the time loop in chunks dealt from a queue
// (ssmq_filter_chunked.hip)
int try_launch_chunked(const FusedArgs &a0, int fd, int fo, int D, int Y, int ND, int NO, int form, int tp, int selo, int opt, int cus,
                       hipStream_t s, bool dry_run, const char **name);

// compute units of the library's device (the wave-split kernel is chosen by how many workgroups the device can spread out)
static int device_cus() {
    static thread_local int cus = 0;
    static thread_local unsigned epoch = ~0u;
    if (epoch != device_epoch() || !cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        epoch = device_epoch();
    }
    return cus > 0 ? cus : 256;
}

// Returns 1 if a fused kernel was launched, 0 if none exists for this combination, < 0 on error.
int try_launch_fused(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho,
                     const ssmq_integrand *fo, int sel_obs, int64_t B, int64_t ld, int T, const double *d_y,
                     const double *d_m0, const double *d_P0, const double *d_gqg, const double *d_rr, double *d_fm,
                     double *d_fP, int32_t *d_status, hipStream_t s, const char **name, bool dry_run,
                     const double *d_sscale, double student_dof, const double *d_ttab_dyn, const double *d_ttab_obs) {
    if (hd->form != ho->form || (hd->tp_nu > 0.0) != (ho->tp_nu > 0.0) || sel_obs < 0 || fd->n_idx > 0) return 0;
    if (!dry_run || B > 0) {
        // batches whose waves would each sit alone on a SIMD: one trajectory on four lanes (ssmq_filter_quad.hip) ...
        const int rq = try_launch_quad(hd, fd, ho, fo, sel_obs, B, ld, T, d_y, d_m0, d_P0, d_gqg, d_rr, d_fm, d_fP, d_status, s, name, dry_run,
                                       d_sscale, student_dof, device_cus());
        if (rq != 0) return rq;
        // ... or the sigma points shared out over the waves of a workgroup (ssmq_filter_wsplit.hip)
        const int rw = try_launch_wsplit(hd, fd, ho, fo, sel_obs, B, ld, T, d_y, d_m0, d_P0, d_gqg, d_rr, d_fm, d_fP, d_status, s, name,
                                         dry_run, d_sscale, student_dof, device_cus());    // (the dry run asks the same device: 256 only without one)
        if (rw != 0) return rw;
    }
    const int tp = hd->tp_nu > 0.0 ? 1 : 0;
    const int both = hd->opt_mask & ho->opt_mask;
    // best fast path BOTH handles qualify for: reflection-symmetric weights (7), LDL' + unscented points (3), the dense kernel
    const int plain = !(tp || hd->form == SSMQ_FORM_SIGMA);
    const int want[3] = {(plain && (both & 7) == 7) ? 7 : -1, both & (plain ? 3 : SSMQ_OPT_UT), 0};
    for (int w = 0; w < 3; ++w)
    for (const FusedEntry &e : kFused) {
        if (want[w] < 0) break;      // (no such variant for these handles: next w)
        if (e.fd == fd->id && e.fo == fo->id && e.D == hd->D && e.Y == ho->E && e.ND == hd->N && e.NO == ho->N &&
            e.form == hd->form && e.tp == tp && e.selo == sel_obs && e.opt == want[w]) {
            const int stu = (d_sscale != nullptr && student_dof > 0.0) ? 1 : 0;
            if ((d_sscale != nullptr) != (student_dof > 0.0)) return 0;   // never produced by the entry points
            fused_fn fn = e.fn[stu] ? e.fn[stu] : e.fn[0];
            const char *kname = e.name;
            if (HasTimeTableRT(fd->id) && !d_ttab_dyn && !dry_run) return 0;   // the kernels read the table
            if (name) *name = kname;
            FusedArgs a;
            memset(&a, 0, sizeof(a));
            a.B = B; a.T = T; a.lpw = 64; a.sscale = d_sscale; a.student_dof = student_dof;
            if (dry_run) {
                if (B > 0) (void)try_launch_chunked(a, e.fd, e.fo, e.D, e.Y, e.ND, e.NO, e.form, e.tp, e.selo, e.opt, device_cus(), s, true, name);
                return 1;
            }
            a.y = d_y; a.m0 = d_m0; a.P0 = d_P0; a.fm = d_fm; a.fP = d_fP; a.status = d_status;
            a.c_dyn = hd->d_small; a.c_obs = ho->d_small; a.gqg = d_gqg; a.rr = d_rr; a.B = B; a.ld = ld; a.T = T;
            a.emv_dyn = hd->emv_mode; a.emv_obs = ho->emv_mode; a.nu_dyn = hd->tp_nu; a.nu_obs = ho->tp_nu;
            a.sscale = d_sscale; a.student_dof = student_dof;
            a.lpw = 64;
            if (const char *ev = ssmq::sw("SSMQ_FUSED_LPW")) {
                const int v = atoi(ev);
                if (v == 16 || v == 32 || v == 64) a.lpw = v;
            }
            fill_fpar(fd, &a.fd);
            fill_fpar(fo, &a.fo);
            a.fd.ttab = d_ttab_dyn;
            a.fo.ttab = d_ttab_obs;
            if (a.lpw == 64) {
                const int rcq = try_launch_chunked(a, e.fd, e.fo, e.D, e.Y, e.ND, e.NO, e.form, e.tp, e.selo, e.opt, device_cus(), s, false, name);
                if (rcq != 0) return rcq;
            }
            int rc = hip_fail(fn(a, s), kname);
            return rc ? rc : 1;
        }
    }
    return 0;
}

}  // namespace ssmq
