// What the fused time-loop kernels share: the argument block of the additive-noise loop, the register sinks that receive a
// transform's results, and the per-step time tables (ssmq_filter_fused.hip: one trajectory per lane, all points in one wave;
// ssmq_filter_wsplit.hip: the same trajectory on the same lane of W waves, each evaluating every W-th sigma point).
#pragma once
#include "ssmq_apply_small.h"

namespace ssmq {

struct FusedArgs {
    const double *y;        // [T][Y][ld]
    const double *m0, *P0;  // [D][ld], [D*D][ld]
    double *fm, *fP;        // [T][D][ld], [T][D*D][ld]
    int32_t *status;        // [B]: 0 or 1 + first failing step
    const double *c_dyn, *c_obs, *gqg, *rr;
    int64_t B, ld;
    int32_t T, emv_dyn, emv_obs, lpw;   // lpw: active lanes (trajectories) per wave
    double nu_dyn, nu_obs;
    // Studentian recursion (ssinf.py:634-736): per-step scale (dof_pr - 2) / dof_pr [T] (null = Gaussian filter), the
    // filter's dof for the measurement-update rescaling; gqg / rr then hold G q_smat G' and r_smat
    const double *sscale;
    double student_dof;
    FPar fd, fo;
    // the strip schedule (ssmq_filter_chunked.hip): blocks of lpw trajectories; queue = hand-over flags [n_blocks] + the strip
    // counter; hand = the state a piece leaves for the piece that continues its block, [n_blocks][D + D (D + 1) / 2 + 1][64]
    int32_t t_chunk, n_blocks;     // (t_chunk: unused since the strips)
    int32_t *queue;
    double *hand;
};

template <int D, int E>
struct RegSink {
    double mf[E];
    double cv[E * (E + 1) / 2];
    double cx[E][D];
    __device__ __forceinline__ void mean(int e, double v) { mf[e] = v; }
    __device__ __forceinline__ void cov(int e, int e2, double v) { cv[SSMQ_PK(e, e2)] = v; }
    __device__ __forceinline__ void ccov(int e, int d, double v) { cx[e][d] = v; }
};
template <int D, int E>
struct RegSinkNoCross {
    double mf[E];
    double cv[E * (E + 1) / 2];
    __device__ __forceinline__ void mean(int e, double v) { mf[e] = v; }
    __device__ __forceinline__ void cov(int e, int e2, double v) { cv[SSMQ_PK(e, e2)] = v; }
    __device__ __forceinline__ void ccov(int, int, double) {}
};

// Integrands whose time dependence is a per-step constant tabulated on the host (time_table() in ssmq_device.h): the
// fused loops fetch the entry of the next step one step ahead instead of loading (or re-evaluating) it on the chain.
template <int F> struct HasTimeTable { static constexpr bool value = false; };
template <> struct HasTimeTable<SSMQ_F_UNGM_DYN> { static constexpr bool value = true; };
template <> struct HasTimeTable<SSMQ_F_UNGMNA_DYN> { static constexpr bool value = true; };

}  // namespace ssmq
