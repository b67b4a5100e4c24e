// The theta-batched filter step for SMALL systems, one parameter item per LANE, everything in registers (round 5).
//
// ssmq_gp_theta_step's two-launch route (k_theta_weights: a workgroup - a wave for up to 8 points - per item and transform;
// k_theta_chain: a wave per item through the generic LDS kernel) takes 13 + 17 us however few items there are: dozens of
// barrier-separated phases and LDS / global round trips for what is, on the reference's demo systems of the marginalised filter
// (UNGM: 1 state, 2-3 points; pendulum: 2 states, 4-5 points), a few hundred to two thousand flops per item.  Its callers are
// latency-bound on exactly that: MarginalInference.theta_step (param_dim + 1 items per BFGS evaluation) and the rounds of the
// batched marginalised filter (csrc/ssmq_marginal.hip: most rounds serve a handful of straggler trajectories).
//
// Here lane = item: RBF kernel matrix on the unit points, its Cholesky factor and inverse, the Gaussian expectations q, R, Q, the
// GP quadrature weights wm / Wc / Wcc and the model variance (bq/bqkern.py:38-64,96-120,329-424, bq/bqmod.py:495-523), the BQ
// moment transform of the dynamics (bq/bqmtran.py:60-109,158-223), + G Q G', the same for the measurement model, + R, the Kalman
// update (ssinf.py:321-323) and log N(y | y_mean, P_y) (ssinf.py:1153-1198) - with the shapes as template constants, so that
// every loop unrolls and every array is registers.  The operations and their ORDER are those of the kernels this replaces
// (weights_body, chol_block, chol_inverse, gemm in ssmq_weights.hip; apply_wave_body in ssmq_apply_wide.hip;
// kalman_update_item, gauss_logpdf_item in ssmq_update.h), so the two routes agree to rounding (tests compare them and both
// with the reference's vectors); SSMQ_NO_THETA_ITEM=1 keeps the two-launch route.
#include <cstdlib>
#include <cstring>
#include "ssmq_host.h"
#include "ssmq_update.h"
#include "ssmq_theta_item.h"

namespace ssmq {
namespace {

struct ThetaItemArgs {
    int32_t fid_dyn, fid_obs, emv_dyn, emv_obs;
    FPar fpd, fpo;
    const double *xid, *xio;            // unit points, natural layout [Din][Nd], [D][No]
    const double *pard, *paro;          // kernel parameters [P][1 + Din], [P][1 + D]
    const double *mean, *cov;           // state moments [.][Din], [.][Din Din]; item stride bs_mean / bs_cov (0: shared)
    int64_t bs_mean, bs_cov;
    const double *y;                    // measurement planes [Y][ld]
    const double *time;                 // [P] or [1]
    int32_t time_stride;
    const double *gq, *rr;              // G Q G' [D D] (zeros for dynamics that take their noise as an argument), R [Y Y]
    double jitter;
    double *m_fi, *P_fi, *ll;           // planes [D][ld], [D D][ld], [ld]
    int32_t *st_all;                    // merged flags [ld]: 1 weights(dyn) | 2 weights(obs) | 4 transform(dyn) | 8 transform(obs) | 16 update
    int64_t ld, P;
    const int32_t *count;               // null, or the number of items on the device
};

template <int DIN, int D, int Y, int ND, int NO>
__global__ __launch_bounds__(64) void k_theta_item(const ThetaItemArgs a) {
    const int64_t P = a.count ? (int64_t)*a.count : a.P;
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= P) return;
    double m[DIN], cv[DIN][DIN], yv[Y], m_fi[D], P_fi[D][D], ll;
#pragma unroll
    for (int d = 0; d < DIN; ++d) m[d] = a.mean[b * a.bs_mean + d];
#pragma unroll
    for (int i = 0; i < DIN; ++i)
#pragma unroll
        for (int j = 0; j < DIN; ++j) cv[i][j] = a.cov[b * a.bs_cov + i * DIN + j];
#pragma unroll
    for (int i = 0; i < Y; ++i) yv[i] = a.y[(int64_t)i * a.ld + b];
    const double t = a.time ? a.time[a.time_stride ? b : 0] : 0.0;
    const int32_t st = theta_item::theta_item_core<DIN, D, Y, ND, NO>(a.fid_dyn, a.fid_obs, a.fpd, a.fpo, a.emv_dyn, a.emv_obs, a.xid, a.xio,
                                                                    a.pard + b * (1 + DIN), a.paro + b * (1 + D), m, cv, yv, t, a.gq,
                                                                    a.rr, a.jitter, m_fi, P_fi, ll);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        a.m_fi[(int64_t)d * a.ld + b] = m_fi[d];
#pragma unroll
        for (int d2 = 0; d2 < D; ++d2) a.P_fi[(int64_t)(d * D + d2) * a.ld + b] = P_fi[d][d2];
    }
    a.ll[b] = ll;
    a.st_all[b] = st;
}

typedef void (*theta_item_kernel)(const ThetaItemArgs);
struct ThetaItemEntry {
    int Din, D, Y, Nd, No;
    theta_item_kernel k;
};
#define SSMQ_TI(DIN, D, Y, ND, NO) {DIN, D, Y, ND, NO, &k_theta_item<DIN, D, Y, ND, NO>}
// the reference's demo systems of the marginalised filter (tests/test_ssinf.py:262-300, research/gpq): UNGM with spherical-radial
// / unscented points, UNGM with its noise as an argument, the pendulum
const ThetaItemEntry kThetaItem[] = {
    SSMQ_TI(1, 1, 1, 2, 2), SSMQ_TI(1, 1, 1, 3, 3), SSMQ_TI(2, 1, 1, 4, 2), SSMQ_TI(2, 1, 1, 5, 3), SSMQ_TI(2, 2, 1, 4, 4), SSMQ_TI(2, 2, 1, 5, 5),
};

const ThetaItemEntry *find_theta_item(int Din, int D, int Y, int Nd, int No) {
    if (ssmq::sw("SSMQ_NO_THETA_ITEM")) return nullptr;
    for (const ThetaItemEntry &e : kThetaItem)
        if (e.Din == Din && e.D == D && e.Y == Y && e.Nd == Nd && e.No == No) return &e;
    return nullptr;
}

}  // namespace

bool theta_item_supported(int Din, int D, int Y, int Nd, int No) { return find_theta_item(Din, D, Y, Nd, No) != nullptr; }

// One launch for `bound` items (the kernel reads the true count from *d_count when that is not null).  Pointers as in
// gp_theta_step_impl's device arena; mean / cov item strides 0 = shared by all items.
int launch_theta_item(int Din, int D, int Y, int Nd, int No, const ssmq_integrand *f_dyn, const ssmq_integrand *f_obs, int emv_dyn,
                      int emv_obs, const double *xid, const double *xio, const double *pard, const double *paro, const double *mean,
                      const double *cov, int64_t bs_mean, int64_t bs_cov, const double *ysoa, const double *time, int time_stride,
                      const double *gq, const double *rr, double jitter, double *m_fi, double *P_fi, double *ll, int32_t *st_all,
                      int64_t ld, int64_t bound, const int32_t *d_count, hipStream_t s) {
    const ThetaItemEntry *e = find_theta_item(Din, D, Y, Nd, No);
    if (!e) {
        set_error("theta_item: no instantiation for this shape");
        return SSMQ_E_UNSUPPORTED;
    }
    ThetaItemArgs a;
    std::memset(&a, 0, sizeof(a));
    a.fid_dyn = f_dyn->id; a.fid_obs = f_obs->id; a.emv_dyn = emv_dyn; a.emv_obs = emv_obs;
    fill_fpar(f_dyn, &a.fpd);
    fill_fpar(f_obs, &a.fpo);
    a.xid = xid; a.xio = xio; a.pard = pard; a.paro = paro; a.mean = mean; a.cov = cov; a.bs_mean = bs_mean; a.bs_cov = bs_cov;
    a.y = ysoa; a.time = time; a.time_stride = time_stride; a.gq = gq; a.rr = rr; a.jitter = jitter;
    a.m_fi = m_fi; a.P_fi = P_fi; a.ll = ll; a.st_all = st_all; a.ld = ld; a.P = bound; a.count = d_count;
    hipLaunchKernelGGL(e->k, dim3((unsigned)((bound + 63) / 64)), dim3(64), 0, s, a);
    return hip_fail(hipGetLastError(), "k_theta_item");
}

}  // namespace ssmq
