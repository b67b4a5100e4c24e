// The fused filter time loop as a device function + its whole-pass kernel (ssmq_filter_fused.hip: dispatch table; the chunked,
// self-scheduling variant of the heavy shapes: ssmq_filter_chunked.hip).
#pragma once
#include <cstdlib>
#include "ssmq_fused.h"
#include <type_traits>
#include "ssmq_host.h"

namespace ssmq {

// STU: -1 Gaussian or Studentian recursion decided at run time (a.sscale / a.student_dof), 0 / 1 fixed at compile time
// (scalar-state kernels: on a 105-instruction step the run-time form costs two branches, three multiplications by a
// scale of one and their operand moves - 5 us of a 43 us pass).
#ifdef SSMQ_FUSED_NO_STORE          // A/B builds: the time loop without its stores / with ordinary (temporal) stores
#undef SSMQ_STORE
#define SSMQ_STORE(dst, v) asm volatile("" ::"v"(v))
#endif
#ifdef SSMQ_FUSED_PLAIN_STORE
#undef SSMQ_STORE
#define SSMQ_STORE(dst, v) (dst) = (v)
#endif
#ifndef SSMQ_FUSED_FORCE_OCC
#define SSMQ_FUSED_FORCE_OCC 0   // A/B builds (tools/build_variant.sh): waves per SIMD requested for every instantiation
#endif
#ifndef SSMQ_FUSED_OCC_D5_SIGMA
#define SSMQ_FUSED_OCC_D5_SIGMA 1   // waves per SIMD for the centred D = 5 kernels: 276 registers unconstrained; held to 256
                                    // (2 waves) 20 of them spill and the UKF pass is 2-3 % slower (round 3, tools/build_variant.sh)
#endif
// One pass over the steps k_begin ... k_end - 1 of the wave's trajectories (block `blk` of a.lpw trajectories).  The whole filter is
// the pass (0, T, first, last); CHUNKED passes (ssmq_filter_chunked.hip) start from the state - mean, lower triangle of the
// covariance, status word: what the registers held, bit for bit - that the previous pass of the block left in a.hand.
template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int OPT, int STU, bool CHUNKED>
__device__ __forceinline__ void fused_pass(const FusedArgs &a, uint32_t blk, int k_begin, int k_end, bool first, bool last) {
    const uint32_t b = blk * a.lpw + threadIdx.x;
    if ((int64_t)b >= a.B) return;
    const int64_t ld = a.ld;
    double m[D], Pl[D * (D + 1) / 2];
    if (!CHUNKED || first) {
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = a.m0[d * ld + b];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Pl[SSMQ_PK(i, j)] = a.P0[(i * D + j) * ld + b];
    } else {
        // the state the previous chunk of this block left (ssmq_filter_chunked.hip: system-scope accesses, coherent without cache
        // maintenance): [D + D (D + 1) / 2 + 1][64] doubles per block
        const double *h = a.hand + (size_t)blk * (D + D * (D + 1) / 2 + 1) * 64 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = __hip_atomic_load(h + d * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) Pl[i] = __hip_atomic_load(h + (D + i) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    CoreParams cpd{(cdouble_p)a.c_dyn, (cdouble_p)a.gqg, a.emv_dyn, a.nu_dyn, 1.0, 1.0};
    CoreParams cpo{(cdouble_p)a.c_obs, (cdouble_p)a.rr, a.emv_obs, a.nu_obs, 1.0, 1.0};
    const cdouble_p ssc = (cdouble_p)a.sscale;
    const bool stu_scale = STU < 0 ? ssc != nullptr : STU == 1;          // ssinf.py:672-693
    const bool stu_update = STU < 0 ? a.student_dof > 0.0 : STU == 1;    // ssinf.py:729-733
    // scalar state and measurement: failures are carried by NaN instead of per-step selects (see the update below)
    constexpr bool kScalar = (D == 1 && Y == 1);
    const double nan = __builtin_nan("");
    int32_t agg = 0;       // !kScalar: 1 + first failing step;  kScalar: number of steps completed without a NaN
    if (CHUNKED && !first)
        agg = (int32_t)__hip_atomic_load(a.hand + (size_t)blk * (D + D * (D + 1) / 2 + 1) * 64 + (D + D * (D + 1) / 2) * 64 + threadIdx.x, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_SYSTEM);
#ifndef SSMQ_FUSED_YAHEAD
#define SSMQ_FUSED_YAHEAD 1          // A/B builds: how many steps ahead the measurements are requested
#endif
    constexpr int YA = SSMQ_FUSED_YAHEAD;
    double ynext[YA][Y];   // the measurement of step k + 1 is requested one step ahead: its HBM latency hides behind step k
#pragma unroll
    for (int q = 0; q < YA; ++q)
#pragma unroll
        for (int i = 0; i < Y; ++i) ynext[q][i] = a.y[((int64_t)(k_begin + q < a.T ? k_begin + q : a.T - 1) * Y + i) * ld + b];
    // Per-step scalars (time-table entries of the integrands, the Studentian scale) are requested one step ahead as
    // well: consumed in the step that loads them they put a scalar-cache round trip on the chain of every step.
    FPar fd = a.fd, fo = a.fo;
    constexpr bool kTTd = HasTimeTable<FD>::value, kTTo = HasTimeTable<FO>::value;
    const cdouble_p ttd = (cdouble_p)a.fd.ttab, tto = (cdouble_p)a.fo.ttab;   // host: non-null when kTTd / kTTo
    double tdn = 0.0, ton = 0.0, scn = 1.0;
    if constexpr (kTTd) { tdn = ttd[k_begin]; fd.use_tval = 1; }
    if constexpr (kTTo) { ton = tto[k_begin]; fo.use_tval = 1; }
    if (stu_scale) scn = ssc[k_begin];
    // Everything requested so far has to have ARRIVED before the loop is entered: hipcc's wait-count insertion merges the
    // loop-entry state into the loop header, and a vector load still pending there (m0, P0: used by the first step
    // only) leaves an s_waitcnt vmcnt(n) in the body that - memory operations retire in order - makes EVERY iteration
    // wait for the acknowledgement of the stores its predecessor has just issued.
#pragma unroll
    for (int d = 0; d < D; ++d) pin_v(m[d]);
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) pin_v(Pl[i]);
#pragma unroll
    for (int q = 0; q < YA; ++q)
#pragma unroll
        for (int i = 0; i < Y; ++i) pin_v(ynext[q][i]);
#pragma unroll 1
    for (int k = k_begin; k < k_end; ++k) {
        const double t = (double)k;  // both transforms of step k + 1 use time index k (ssinf.py:104, 276-288)
        double ycur[Y];
#pragma unroll
        for (int i = 0; i < Y; ++i) ycur[i] = ynext[0][i];
#pragma unroll
        for (int q = 0; q + 1 < YA; ++q)
#pragma unroll
            for (int i = 0; i < Y; ++i) ynext[q][i] = ynext[q + 1][i];
        if constexpr (kTTd) fd.tval = tdn;
        if constexpr (kTTo) fo.tval = ton;
        const double sc = scn;
        {   // next step's inputs; the last step re-requests its own (no branch in the loop body)
            const int kn = (k + 1 < a.T) ? k + 1 : k;
            // (YA == 1 keeps the ONE index kn for the plane and the time tables: a second one, equal in value, cost the loop its
            // strength-reduced addresses and the headline pass 2 us - 32.3 -> 34.3 - before it was noticed in the bench line)
            const int ky = YA == 1 ? kn : ((k + YA < a.T) ? k + YA : a.T - 1);
#pragma unroll
            for (int i = 0; i < Y; ++i) ynext[YA - 1][i] = a.y[((int64_t)ky * Y + i) * ld + b];
            if constexpr (kTTd) tdn = ttd[kn];
            if constexpr (kTTo) ton = tto[kn];
            if (stu_scale) scn = ssc[kn];
        }
        if (stu_scale) {   // Studentian: transformed covariances become scale matrices before the noise term (ssinf.py:672-693)
            cpd.cov_scale = sc;
            cpo.cov_scale = sc;
            cpo.ccov_scale = sc;
        }
        // ---- time update: predictive state moments, + G Q G' (ssinf.py:276-279) ----------------------------------
        RegSinkNoCross<D, D> pr;
        bool ok = moment_transform_core<D, D, ND, FD, FORM, TP, 0, false, OPT, RegSinkNoCross<D, D>>(m, Pl, t, fd, cpd, pr);
        // ---- predictive measurement moments, + R (ssinf.py:287-291) ------------------------------------------------
        double L2[D * (D + 1) / 2];
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) L2[i] = pr.cv[i];
        RegSink<D, Y> ob;
        ok = moment_transform_core<D, Y, NO, FO, FORM, TP, SELO, true, OPT, RegSink<D, Y>>(pr.mf, L2, t, fo, cpo, ob) && ok;
        // ---- measurement update (ssinf.py:321-323) ---------------------------------------------------------------------
        double S[Y * (Y + 1) / 2];
#pragma unroll
        for (int i = 0; i < Y * (Y + 1) / 2; ++i) S[i] = ob.cv[i];
        double G[D][Y];
        // (Round 6) From the first failing step on every result of a trajectory is NaN (the reference raises there).  Where the state
        // is not scalar, ONE poisoned row of the cross-covariance does it - the gain, and through it the mean and every covariance
        // entry of this and all later steps, inherit the NaN: 2 D selects instead of 2 (D + D^2) on the results.  Bit-neutral for
        // the trajectories that do not fail (a select, not an addition: not even the sign of a zero changes).
        constexpr bool kPoisonRow = !kScalar;
        if (kPoisonRow && Y > 1) {
            ok = chol_packed<Y>(S) && ok;
        } else if (kPoisonRow) {
            ok = (S[0] > 0.0) && ok;
        }
        if (kPoisonRow) {
            if (agg == 0 && !ok) agg = k + 1;
#pragma unroll
            for (int d = 0; d < D; ++d) ob.cx[0][d] = (agg == 0) ? ob.cx[0][d] : nan;
        }
        if (Y == 1) {
            // scalar measurement: P_y^-1 P_yx is one division; the factor-and-two-substitutions route of cho_solve
            // (sqrt + two divisions by it) would only lengthen the serial dependency chain of the time loop
            if constexpr (kScalar) {
                // A non-positive (or NaN) covariance turns into NaN by itself in the square root of the next transform
                // (sqrt_rsqrt: rsq of a negative number, 0 * inf) and from there into every later result of the
                // trajectory; the one failure that does not is P_y <= 0, so that one is poisoned here - ONE select on
                // the high word instead of four on the results, and no flag logic on the step's chain.
                int hi = __double2hiint(S[0]);
                hi = (S[0] > 0.0) ? hi : 0x7ff80000;
                S[0] = __hiloint2double(hi, __double2loint(S[0]));
            }
#pragma unroll
            for (int d = 0; d < D; ++d) G[d][0] = div_nr(ob.cx[0][d], S[0]);
        } else {
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
        const bool good = true;       // (failures travel as NaN: the scalar kernels through the innovation variance, the others through the poisoned row)
        double sc2 = 1.0;
        if (stu_update) {   // (dof + delta'delta) / (dof + Y), delta = chol(S)^-1 (y - y_mean)  (ssinf.py:729-733)
            double dl[Y], dd = 0.0;
            if (Y == 1) {
                const double dy0 = ycur[0] - ob.mf[0];
                dd = div_nr(dy0 * dy0, S[0]);
            } else {
#pragma unroll
                for (int i = 0; i < Y; ++i) {
                    double s = ycur[i] - ob.mf[i];
#pragma unroll
                    for (int q = 0; q < i; ++q) s -= S[SSMQ_PK(i, q)] * dl[q];
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
            for (int i = 0; i < Y; ++i) s += G[d][i] * (ycur[i] - ob.mf[i]);
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
                if (kScalar) agg += (p == p) ? 1 : 0;   // a NaN never goes away again: counts the steps completed
                if (d2 <= d) Pl[SSMQ_PK(d, d2)] = sc2 * p;   // next Cholesky reads the lower triangle only (LAPACK 'L')
            }
        }
    }
    if (CHUNKED && !last) {
        double *h = a.hand + (size_t)blk * (D + D * (D + 1) / 2 + 1) * 64 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < D; ++d) __hip_atomic_store(h + d * 64, m[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) __hip_atomic_store(h + (D + i) * 64, Pl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(h + (D + D * (D + 1) / 2) * 64, (double)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (kScalar) agg = (agg == a.T) ? 0 : agg + 1;
    a.status[b] = agg;
}

template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int OPT, int STU = -1>
__global__ __launch_bounds__(kSmallBlock, (SSMQ_FUSED_FORCE_OCC ? SSMQ_FUSED_FORCE_OCC
                                           : (D >= 6 ? 1 : ((D >= 5 && FORM == SSMQ_FORM_SIGMA) ? SSMQ_FUSED_OCC_D5_SIGMA : 2)))) void k_filter_fused(const FusedArgs a) {
    if ((int)threadIdx.x >= a.lpw) return;
    fused_pass<D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, STU, false>(a, blockIdx.x, 0, a.T, true, true);
}


}  // namespace ssmq
