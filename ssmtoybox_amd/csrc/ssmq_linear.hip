// Linearisation moment transform (mtran.py:49-59: LinearizationTransform - the transform of ExtendedKalman, ssinf.py:347-357):
//   mean_f = f(mean),  J = f(mean, dx=True),  cov_fx = J cov,  cov_f = cov_fx J'.
// One trajectory per lane, element planes in and out as for every other transform (element e of trajectory b at ptr[e ld + b]):
// 8 (D + D^2) bytes read and 8 (E + E^2 + E D) written per trajectory, a few dozen operations - an HBM-bound map.
// The Jacobian is the model's own (ssmq_device.h: jac_integrand - the seven models whose dyn_fcn_dx / meas_fcn_dx the reference
// implements), placed into the columns of the full state as MeasurementModel.meas_eval does (ssmod.py:985-1009): through the
// state index where there is one; without one the reference assigns `out[:, None] = jac`, which for a one-column Jacobian and a
// wider state BROADCASTS it into every column (Pendulum2DMeasurement on the 2-D state: both columns cos(x0)) - kept.
#include "ssmq_device.h"
#include "ssmq_host.h"
#include "ssmq_math.h"

namespace ssmq {

struct LinArgs {
    int32_t D, E, din, fid, time_stride, bcast;      // bcast: no state index and din == 1 < D
    const double *mean, *cov, *time, *cov_add;       // planes [D][ld], [D*D][ld]; time [B] or [1]; cov_add [E*E] or null
    double *mean_f, *cov_f, *cov_fx;                 // planes [E][ld], [E*E][ld], [E*D][ld]
    int32_t *status;
    int64_t B, ld;
    double cov_scale, ccov_scale;
    FPar fp;
};

// DT, ET > 0: the transform's dimensions at compile time (everything in registers: the shapes of the seven models that have a
// Jacobian); 0: run-time sizes, private arrays of the maximal size (scratch memory - a fallback, 0.13-0.21 of HBM where the
// specialised bodies reach 0.6-0.7)
template <int DT, int ET>
__global__ __launch_bounds__(256) void k_linearize(const LinArgs a) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= a.B) return;
    constexpr int DM = DT > 0 ? DT : SSMQ_MAX_DIM, EM = ET > 0 ? ET : SSMQ_MAX_DIM;
    const int D = DT > 0 ? DT : a.D, E = ET > 0 ? ET : a.E, din = a.din;
    const int64_t ld = a.ld;
    double x[DM], xs[kMaxIntegrandIn], o[SSMQ_MAX_DIM];
    double Js[EM * DM], J[EM * DM], C[EM * DM];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = a.mean[d * ld + b];
#pragma unroll
    for (int k = 0; k < kMaxIntegrandIn; ++k) {
        if (DT > 0) {                    // static register indices: a select chain over the DT candidates
            const int src = a.fp.n_idx > 0 ? (k < a.fp.n_idx ? a.fp.idx[k] : 0) : (k < D ? k : 0);
            double v = x[0];
#pragma unroll
            for (int q = 1; q < DM; ++q) v = (src == q) ? x[q] : v;
            xs[k] = k < DM ? v : 0.0;
        } else {
            const int src = a.fp.n_idx > 0 ? (k < a.fp.n_idx ? a.fp.idx[k] : 0) : (k < D ? k : 0);
            xs[k] = x[src];
        }
    }
    const double t = a.time ? a.time[a.time_stride ? b : 0] : 0.0;
    for (int e = 0; e < SSMQ_MAX_DIM; ++e) o[e] = 0.0;
    eval_integrand(a.fid, xs, t, a.fp, o);
    for (int i = 0; i < E * din; ++i) Js[i] = 0.0;
    jac_integrand(a.fid, xs, t, a.fp, Js, din);
    for (int i = 0; i < E * D; ++i) J[i] = 0.0;
    for (int e = 0; e < E; ++e) {
        if (a.fp.n_idx > 0) {
            for (int k = 0; k < din; ++k) {
                if (DT > 0) {
#pragma unroll
                    for (int d = 0; d < DM; ++d)
                        if (a.fp.idx[k] == d) J[e * D + d] = Js[e * din + k];
                } else {
                    J[e * D + a.fp.idx[k]] = Js[e * din + k];
                }
            }
        } else if (a.bcast) {
            for (int d = 0; d < D; ++d) J[e * D + d] = Js[e * din];
        } else {
            for (int k = 0; k < din; ++k) J[e * D + k] = Js[e * din + k];
        }
    }
    // cov_fx = J cov (E x D), cov_f = cov_fx J' (E x E)
    for (int e = 0; e < E; ++e)
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
            for (int k = 0; k < D; ++k) s += J[e * D + k] * a.cov[(int64_t)(k * D + d) * ld + b];
            C[e * D + d] = s;
        }
    for (int e = 0; e < E; ++e) a.mean_f[e * ld + b] = o[e];
    for (int e = 0; e < E; ++e)
        for (int e2 = 0; e2 < E; ++e2) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) s += C[e * D + d] * J[e2 * D + d];
            s *= a.cov_scale;
            if (a.cov_add) s += a.cov_add[e * E + e2];
            a.cov_f[(int64_t)(e * E + e2) * ld + b] = s;
        }
    for (int e = 0; e < E; ++e)
        for (int d = 0; d < D; ++d) a.cov_fx[(int64_t)(e * D + d) * ld + b] = C[e * D + d] * a.ccov_scale;
    a.status[b] = 0;
}

// D, E: the transform's; din: the integrand's own input count (ssmq_api.hip: FInfo)
int launch_linearize(int D, int E, int din, const ssmq_integrand *f, const FPar &fp, int64_t B, int64_t ld, const double *d_mean,
                     const double *d_cov, const double *d_time, int time_stride, double *d_mean_f, double *d_cov_f, double *d_cov_fx,
                     int32_t *d_status, const double *d_cov_add, double cov_scale, double ccov_scale, hipStream_t s) {
    if (!integrand_has_jacobian(f->id)) {
        set_error("linearisation: this model has no Jacobian (its dyn_fcn_dx / meas_fcn_dx returns None in the reference too)");
        return SSMQ_E_UNSUPPORTED;
    }
    if (f->n_idx == 0 && din != D && din != 1) {
        set_error("linearisation: a Jacobian of 1 < din < D columns without a state index has no placement (numpy raises there)");
        return SSMQ_E_UNSUPPORTED;
    }
    LinArgs a;
    a.D = D; a.E = E; a.din = din; a.fid = f->id; a.time_stride = time_stride; a.bcast = (f->n_idx == 0 && din == 1 && D > 1) ? 1 : 0;
    a.mean = d_mean; a.cov = d_cov; a.time = d_time; a.cov_add = d_cov_add;
    a.mean_f = d_mean_f; a.cov_f = d_cov_f; a.cov_fx = d_cov_fx; a.status = d_status; a.B = B; a.ld = ld;
    a.cov_scale = cov_scale; a.ccov_scale = ccov_scale; a.fp = fp;
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    const bool generic = ssmq::sw("SSMQ_LINEAR_GENERIC") != nullptr;      // tools/alt_paths.sh: the run-time-size body for every shape
    if (generic) hipLaunchKernelGGL((k_linearize<0, 0>), grid, block, 0, s, a);
    else if (D == 1 && E == 1) hipLaunchKernelGGL((k_linearize<1, 1>), grid, block, 0, s, a);
    else if (D == 2 && E == 1) hipLaunchKernelGGL((k_linearize<2, 1>), grid, block, 0, s, a);
    else if (D == 2 && E == 2) hipLaunchKernelGGL((k_linearize<2, 2>), grid, block, 0, s, a);
    else if (D == 4 && E == 4) hipLaunchKernelGGL((k_linearize<4, 4>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_linearize<0, 0>), grid, block, 0, s, a);
    return hip_fail(hipGetLastError(), "k_linearize");
}

}  // namespace ssmq
