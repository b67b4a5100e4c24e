// Fused filter time loop for batches that do NOT fill the chip: the sigma points of every transform are split over the W
// waves of a workgroup.
//
// k_filter_fused (one trajectory per lane, every sigma point evaluated by that lane) needs ceil(B / 64) waves.  At the batch
// sizes BASELINE names - 1e4 trajectories on one GPU, 1e5 over eight = 12 500 per GPU - that is 157 / 196 waves for 1 024
// SIMDs, and a wave alone on its SIMD is bound by the NUMBER of instructions of a step (one per ~2.4 ns, whatever the mix:
// DESIGN.md 3.4), 70-80 % of them the N evaluations of the integrand and the N-term sums (bq/bqmtran.py:132-223).
//
// Here a workgroup of W waves owns 64 trajectories: trajectory b sits on the SAME lane of every wave, wave w evaluates the
// sigma points n = w, w + W, w + 2W, ... and the partial weighted sums over them; the partial sums (and, for the uncentred BQ
// form, the integrand values themselves) cross waves through LDS planes [slot][lane] - lane-contiguous, conflict-free - with two
// workgroup barriers per transform.  Every wave then adds the W partials in the same order (bitwise the same result in every
// wave) and carries the whole filter state redundantly; the Cholesky factorisations and the Kalman update are replicated
// (they are serial chains: replication costs no time where every wave has a SIMD to itself), the stores are shared out by row.
// HBM access stays what it was: lane = trajectory, every global access of a wave one contiguous 512-byte segment.
//
//   per wave and step, reentry 5-D + radar, unscented filter (N = 11, W = 4):  ~1 300 instructions against ~2 200
//   total issue work: ~2.4 x that of k_filter_fused - which is why the host picks this kernel only when ceil(B / 64) W waves
//   still find a SIMD each (try_launch_wsplit) and k_filter_fused for saturated batches.
//
// Arithmetic: the reference's formulas as in moment_transform_core (ssmq_apply_small.h); the sums over sigma points are formed
// as W partial sums added in wave order, so results differ from k_filter_fused in the last bits (tests bound both against the
// oracle, tests/test_gpu_parity.py::test_wsplit_*).
#include <cstddef>
#include <cstdlib>
#include "ssmq_fused.h"
#include "ssmq_host.h"

namespace ssmq {

// number of LDS slots (planes of 64 doubles) one transform needs in each of the two exchange areas
__host__ __device__ constexpr int ws_slots_a(int E, int N, int form, int W) { return form == SSMQ_FORM_SIGMA ? W * E : E * (N + 1); }
__host__ __device__ constexpr int ws_slots_b(int D, int E, int form, int tp, bool ccov, int W) {
    return W * (E * (E + 1) / 2 * (tp ? 2 : 1) + (ccov ? E * D : 0));
}
__host__ __device__ constexpr int ws_max(int a, int b) { return a > b ? a : b; }

// One moment transform, points split over W waves.  m: mean; L: in = packed lower triangle of the covariance, out = its
// Cholesky factor (every wave factors its own copy).  xa / xb: the two LDS exchange areas, this lane's column (plane stride 64).
// rec0: this wave's first point record (const_layout: rec + w rs); its q-th point is W records further on, and a wave that has
// run out of points reads the all-zero record N.  Every wave returns the complete moments in `out`.
template <int D, int E, int N, int F, int FORM, int TP, int SEL, bool NEED_CCOV, int W, class Sink>
__device__ __forceinline__ bool transform_split(const double (&m)[D], double (&L)[D * (D + 1) / 2], double t, const FPar &fp,
                                                const CoreParams &cp, Sink &out, double *xa, double *xb, int w) {
    constexpr ConstLayout cl = const_layout(D, E, N, FORM);
    using Fun = Fn<F>;
    constexpr int DIN = Fun::DIN;
    constexpr int NL = (N + W - 1) / W;   // points of the busiest wave
    constexpr int NP = E * (E + 1) / 2;
    constexpr int RS = cl.rs;
    // (laundered: the constants this wave reads depend on w only, and hipcc would otherwise hoist every one of those scalar
    // loads out of the time loop and keep them in SGPRs spilled to VGPR lanes - 520 v_readlane / v_writelane in the 5-D loop)
    const cdouble_p c = launder(cp.c);
    const cdouble_p cadd = launder(cp.cadd);

    const bool ok = chol_packed<D>(L);
    Fun fn;
    fn.init(t, fp);

    double fx[E][NL], dx[NEED_CCOV && FORM == SSMQ_FORM_SIGMA ? D : 1][NL];
    cdouble_p rec[NL];
    int nq[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) {
        const int n = w + q * W;            // wave-uniform
        nq[q] = n < N ? n : N;              // N: the zero record
        rec[q] = c + cl.rec + nq[q] * RS;
        double x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double s = m[d];
#pragma unroll
            for (int k = 0; k <= d; ++k) s += L[SSMQ_PK(d, k)] * rec[q][k];
            x[d] = s;
        }
        double xs[DIN], o[E];
        select_inputs<D, DIN, SEL>(x, xs);
        fn.template eval<E>(xs, o);
#pragma unroll
        for (int e = 0; e < E; ++e) fx[e][q] = o[e];
        if (NEED_CCOV && FORM == SSMQ_FORM_SIGMA) {
#pragma unroll
            for (int d = 0; d < D; ++d) dx[d][q] = x[d] - m[d];   // (mean + L xi_n) - mean as the reference forms it (mtran.py:139,148)
        }
    }

    double mf[E];
    if (FORM == SSMQ_FORM_SIGMA) {
        // ---- mean: partial sums over this wave's points, W partials added in wave order -----------------------------------
#pragma unroll
        for (int e = 0; e < E; ++e) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < NL; ++q) s += fx[e][q] * rec[q][D];
            xa[(w * E + e) * 64] = s;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            double s = xa[e * 64];
#pragma unroll
            for (int v = 1; v < W; ++v) s += xa[(v * E + e) * 64];
            mf[e] = s;
            out.mean(e, s);
        }
        // ---- centred covariance and cross-covariance, diagonal weights (mtran.py:105-149) ---------------------------------
        constexpr int V = NP + (NEED_CCOV ? E * D : 0);
        double part[V];
#pragma unroll
        for (int i = 0; i < V; ++i) part[i] = 0.0;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            double df[E], dw[E];
            const double wc = rec[q][D + 1];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                df[e] = fx[e][q] - mf[e];
                dw[e] = df[e] * wc;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
#pragma unroll
                for (int e2 = 0; e2 <= e; ++e2) part[SSMQ_PK(e, e2)] += dw[e] * df[e2];
                if (NEED_CCOV) {
#pragma unroll
                    for (int d = 0; d < D; ++d) part[NP + e * D + d] += dw[e] * dx[d][q];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) xb[(w * V + i) * 64] = part[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < V; ++i) {
            double s = xb[i * 64];
#pragma unroll
            for (int v = 1; v < W; ++v) s += xb[(v * V + i) * 64];
            part[i] = s;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
#pragma unroll
            for (int e2 = 0; e2 <= e; ++e2) out.cov(e, e2, part[SSMQ_PK(e, e2)] * cp.cov_scale + cadd[e * E + e2]);
            if (NEED_CCOV) {
#pragma unroll
                for (int d = 0; d < D; ++d) out.ccov(e, d, part[NP + e * D + d] * cp.ccov_scale);
            }
        }
    } else {
        // ---- uncentred BQ form: the integrand values of all N points go round (every column of Wc meets every value); a wave
        //      without a q-th point parks its (unused) values in plane N ----------------------------------------------------------
#pragma unroll
        for (int q = 0; q < NL; ++q)
#pragma unroll
            for (int e = 0; e < E; ++e) xa[(e * (N + 1) + nq[q]) * 64] = fx[e][q];
        __syncthreads();
        // One output row e at a time: its N values come back from LDS (22 registers, not E N of them), give the mean and, against
        // this wave's COLUMNS j of Wc (and of iK for the t-process model variance), the partial quadratic forms
        //   cv += (fx_e Wc[:, j]) fx[:, j]',  sv likewise with iK;  partial g = fx[:, j] Wcc[:, j]' for the cross-covariance
        // (the zero record makes every term of a point that does not exist vanish)
        constexpr int V = NP * (TP ? 2 : 1) + (NEED_CCOV ? E * D : 0);
        double part[V];
#pragma unroll
        for (int i = 0; i < V; ++i) part[i] = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            double row[N];
#pragma unroll
            for (int i = 0; i < N; ++i) row[i] = xa[(e * (N + 1) + i) * 64];
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) s += row[i] * c[cl.wm + i];
            mf[e] = s;
            out.mean(e, s);
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const cdouble_p wcol = rec[q] + 2 * D + 1, kcol = wcol + N, ccol = rec[q] + D + 1;
                double tj = 0.0, uj = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) tj += row[i] * wcol[i];
                if (TP) {
#pragma unroll
                    for (int i = 0; i < N; ++i) uj += row[i] * kcol[i];
                }
#pragma unroll
                for (int e2 = 0; e2 <= e; ++e2) {
                    part[SSMQ_PK(e, e2)] += tj * fx[e2][q];
                    if (TP) part[NP + SSMQ_PK(e, e2)] += uj * fx[e2][q];
                }
                if (NEED_CCOV) {
#pragma unroll
                    for (int d = 0; d < D; ++d) part[NP * (TP ? 2 : 1) + e * D + d] += fx[e][q] * ccol[d];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) xb[(w * V + i) * 64] = part[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < V; ++i) {
            double s = xb[i * 64];
#pragma unroll
            for (int v = 1; v < W; ++v) s += xb[(v * V + i) * 64];
            part[i] = s;
        }
        const double den = TP ? 1.0 / (cp.tp_nu - 2.0 + (double)N) : 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
#pragma unroll
            for (int e2 = 0; e2 <= e; ++e2) {
                const bool use = (e == e2) || (cp.emv_mode == SSMQ_EMV_BROADCAST);
                double em = use ? c[cl.emv + e * E + e2] : 0.0;
                if (TP) em = (cp.tp_nu - 2.0 + part[NP + SSMQ_PK(e, e2)]) * den * em;     // bq/bqmod.py:1132-1160
                double v = part[SSMQ_PK(e, e2)] - mf[e] * mf[e2] + em;                    // bq/bqmtran.py:199
                v = v * cp.cov_scale + cadd[e * E + e2];
                out.cov(e, e2, v);
            }
            if (NEED_CCOV) {   // cov_fx = (fx Wcc') L'   (bq/bqmtran.py:203-223)
#pragma unroll
                for (int jd = 0; jd < D; ++jd) {
                    double s = 0.0;
#pragma unroll
                    for (int d = 0; d <= jd; ++d) s += part[NP * (TP ? 2 : 1) + e * D + d] * L[SSMQ_PK(jd, d)];
                    out.ccov(e, jd, s * cp.ccov_scale);
                }
            }
        }
    }
    return ok;
}

// An integrand's constants from the kernel-argument segment, field by field through the constant address space (scalar loads;
// what the integrand does not use is never loaded).  Time tables are not used by these kernels.
typedef const __attribute__((address_space(4))) char *ckarg_p;
__device__ __forceinline__ FPar load_fpar(ckarg_p base) {
    FPar f;
    const cdouble_p pp = (cdouble_p)(base + offsetof(FPar, p));
    const __attribute__((address_space(4))) int32_t *ip = (const __attribute__((address_space(4))) int32_t *)(base + offsetof(FPar, idx));
#pragma unroll
    for (int i = 0; i < SSMQ_MAX_FPAR; ++i) f.p[i] = pp[i];
#pragma unroll
    for (int i = 0; i < SSMQ_MAX_FIDX; ++i) f.idx[i] = ip[i];
    f.n_idx = ip[SSMQ_MAX_FIDX];
    f.n_par = ip[SSMQ_MAX_FIDX + 1];
    f.ttab = nullptr;
    f.tval = 0.0;
    f.use_tval = 0;
    return f;
}

template <int D, int Y, int ND, int NO, int FORM, int TP, int W>
__host__ __device__ constexpr int ws_lds_doubles() {
    const int a = ws_max(ws_slots_a(D, ND, FORM, W), ws_slots_a(Y, NO, FORM, W));
    const int b = ws_max(ws_slots_b(D, D, FORM, TP, false, W), ws_slots_b(D, Y, FORM, TP, true, W));
    return (a + b) * 64;
}

template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int W>
__global__ __launch_bounds__(64 * W, 1) void k_filter_wsplit(const FusedArgs a) {
    extern __shared__ double ws_lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int SA = ws_max(ws_slots_a(D, ND, FORM, W), ws_slots_a(Y, NO, FORM, W));
    double *xa = ws_lds + lane, *xb = ws_lds + SA * 64 + lane;
    // lanes past the batch run on trajectory B - 1 (every wave of the workgroup has to reach every barrier) and store nothing
    const uint32_t b0 = blockIdx.x * 64 + lane;
    const bool live = (int64_t)b0 < a.B;
    const uint32_t b = live ? b0 : (uint32_t)(a.B - 1);
    const int64_t ld = a.ld;
    double m[D], Pl[D * (D + 1) / 2];
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = a.m0[d * ld + b];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Pl[SSMQ_PK(i, j)] = a.P0[(i * D + j) * ld + b];
    CoreParams cpd{(cdouble_p)a.c_dyn, (cdouble_p)a.gqg, a.emv_dyn, a.nu_dyn, 1.0, 1.0};
    CoreParams cpo{(cdouble_p)a.c_obs, (cdouble_p)a.rr, a.emv_obs, a.nu_obs, 1.0, 1.0};
    const cdouble_p ssc = (cdouble_p)a.sscale;
    const bool stu_scale = ssc != nullptr;          // ssinf.py:672-693
    const bool stu_update = a.student_dof > 0.0;    // ssinf.py:729-733
    const double nan = __builtin_nan("");
    int32_t agg = 0;       // 1 + first failing step
    double ynext[Y];       // the measurement of step k + 1 is requested one step ahead
#pragma unroll
    for (int i = 0; i < Y; ++i) ynext[i] = a.y[(int64_t)i * ld + b];
    double scn = stu_scale ? ssc[0] : 1.0;
#pragma unroll
    for (int d = 0; d < D; ++d) pin_v(m[d]);          // everything requested so far has arrived before the loop (see k_filter_fused)
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) pin_v(Pl[i]);
#pragma unroll
    for (int i = 0; i < Y; ++i) pin_v(ynext[i]);
#pragma unroll 1
    for (int k = 0; k < a.T; ++k) {
        const double t = (double)k;  // both transforms of step k + 1 use time index k (ssinf.py:104, 276-288)
        double ycur[Y];
#pragma unroll
        for (int i = 0; i < Y; ++i) ycur[i] = ynext[i];
        const double sc = scn;
        {
            const int kn = (k + 1 < a.T) ? k + 1 : k;
#pragma unroll
            for (int i = 0; i < Y; ++i) ynext[i] = a.y[((int64_t)kn * Y + i) * ld + b];
            if (stu_scale) scn = ssc[kn];
        }
        if (stu_scale) {
            cpd.cov_scale = sc;
            cpo.cov_scale = sc;
            cpo.ccov_scale = sc;
        }
        // the integrands' constants are read from the kernel-argument segment where they are used: kept in SGPRs across the loop
        // (two FPar blocks, ~100 registers) they were spilled to VGPR lanes - 210 v_readlane / v_writelane per step
        ckarg_p ka = (ckarg_p)__builtin_amdgcn_kernarg_segment_ptr();      // `a` is the kernel's only argument: offset 0
        asm volatile("" : "+s"(ka));
        const FPar fd = load_fpar(ka + offsetof(FusedArgs, fd)), fo = load_fpar(ka + offsetof(FusedArgs, fo));
        // ---- time update (ssinf.py:276-279) ---------------------------------------------------------------------------
        RegSinkNoCross<D, D> pr;
        bool ok = transform_split<D, D, ND, FD, FORM, TP, 0, false, W>(m, Pl, t, fd, cpd, pr, xa, xb, w);
        // ---- predictive measurement moments (ssinf.py:287-291) ---------------------------------------------------------
        double L2[D * (D + 1) / 2];
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) L2[i] = pr.cv[i];
        RegSink<D, Y> ob;
        ok = transform_split<D, Y, NO, FO, FORM, TP, SELO, true, W>(pr.mf, L2, t, fo, cpo, ob, xa, xb, w) && ok;
        // ---- measurement update (ssinf.py:321-323), replicated in every wave ------------------------------------------------
        double S[Y * (Y + 1) / 2];
#pragma unroll
        for (int i = 0; i < Y * (Y + 1) / 2; ++i) S[i] = ob.cv[i];
        double G[D][Y];
        if (Y == 1) {
            ok = (S[0] > 0.0) && ok;
#pragma unroll
            for (int d = 0; d < D; ++d) G[d][0] = div_nr(ob.cx[0][d], S[0]);
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
        const double gs = agg == 0 ? 1.0 : nan;      // a failed trajectory's moments are NaN from the failing step on: x * 1.0 == x
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
            m[d] = (pr.mf[d] + s) * gs;
        }
        if (w == 0 && live) {
#pragma unroll
            for (int d = 0; d < D; ++d) SSMQ_STORE(a.fm[((int64_t)k * D + d) * ld + b], m[d]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double wv[Y];
#pragma unroll
            for (int j = 0; j < Y; ++j) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < Y; ++i) s += G[d][i] * ob.cv[i >= j ? SSMQ_PK(i, j) : SSMQ_PK(j, i)];
                wv[j] = s;
            }
            double prow[D];
#pragma unroll
            for (int d2 = 0; d2 < D; ++d2) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < Y; ++j) s += wv[j] * G[d2][j];
                const double p = (pr.cv[d >= d2 ? SSMQ_PK(d, d2) : SSMQ_PK(d2, d)] - s) * gs;
                prow[d2] = p;
                if (d2 <= d) Pl[SSMQ_PK(d, d2)] = sc2 * p;   // next Cholesky reads the lower triangle only (LAPACK 'L')
            }
            if (w == (d + 1) % W && live) {      // row d of the covariance leaves from wave (d + 1) mod W (wave 0 has the mean)
#pragma unroll
                for (int d2 = 0; d2 < D; ++d2) SSMQ_STORE(a.fP[((int64_t)k * D * D + d * D + d2) * ld + b], prow[d2]);
            }
        }
    }
    if (w == 0 && live) a.status[b] = agg;
}

template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int W>
static hipError_t launch_wsplit(const FusedArgs &a, hipStream_t s) {
    // W = 4 fills the four SIMDs of a compute unit with ONE workgroup: claim more than half the LDS so that no second one joins it
    constexpr size_t need = sizeof(double) * ws_lds_doubles<D, Y, ND, NO, FORM, TP, W>();
    constexpr size_t lds = (W == 4 && need < 81 * 1024) ? 81 * 1024 : need;
    static_assert(lds <= 160 * 1024, "exchange areas exceed the LDS of a CU");
    auto kern = k_filter_wsplit<D, Y, ND, NO, FD, FO, FORM, TP, SELO, W>;
    static thread_local unsigned attr_epoch = ~0u;
    if (attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_epoch = device_epoch();
    }
    const unsigned grid = (unsigned)((a.B + 63) / 64);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}

typedef hipError_t (*wsplit_fn)(const FusedArgs &, hipStream_t);
struct WsplitEntry {
    int fd, fo, D, Y, ND, NO, form, tp, selo, W;
    wsplit_fn fn;
    const char *name;
    size_t lds_need;      // bytes of LDS the exchange areas take
};
#define SSMQ_WS_ONE(FD, FO, D, Y, N, FORM, TP, SELO, W)                                                          \
    {FD, FO, D, Y, N, N, FORM, TP, SELO, W, &launch_wsplit<D, Y, N, N, FD, FO, FORM, TP, SELO, W>,               \
     "k_filter_wsplit<D=" #D ",Y=" #Y ",N=" #N "," #FD "," #FO "," #FORM ",TP=" #TP ",SELO=" #SELO ",W=" #W ">",                 \
     sizeof(double) * ws_lds_doubles<D, Y, N, N, FORM, TP, W>()}
#define SSMQ_WS(FD, FO, D, Y, N, SELO, W)                    \
    SSMQ_WS_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, W),  \
    SSMQ_WS_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO, W),  \
    SSMQ_WS_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO, W)

// the multi-dimensional systems of BASELINE configs[2] / [3] (ssmq_filter_fused.hip has the full list of shapes; scalar
// models have too little work per step to share out)
static const WsplitEntry kWsplit[] = {
    // W <= 4: a workgroup lives on ONE compute unit, which has four SIMDs - a fifth wave would share a SIMD with a sibling
    SSMQ_WS(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0, 4),
    SSMQ_WS(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0, 2),
    SSMQ_WS(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0, 4),
    SSMQ_WS(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0, 2),
    SSMQ_WS(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 11, 1, 4),
    SSMQ_WS(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 11, 1, 2),
};

// Workgroups of this entry that share a compute unit with every wave on a SIMD of its own: one of four waves, or two of two
// waves where their exchange areas fit the LDS side by side.
static int ws_groups_per_cu(const WsplitEntry &e) { return e.W == 4 ? 1 : (e.W == 2 && e.lds_need <= 80 * 1024 ? 2 : 1); }

// Whether entry `e` is the kernel for a batch of B trajectories on a device of `cus` compute units: only while each of its waves
// still finds a SIMD to itself - the regime in which sharing out the points shortens the step; beyond it k_filter_fused does the
// same work with 2-3 x fewer instructions in total.  SSMQ_FUSED_WSPLIT=0 switches the kernel off, =W forces that W (A/B timing,
// tests).
static bool ws_wanted(const WsplitEntry &e, int64_t B, int cus) {
    int forced = -1;
    if (const char *ev = ssmq::sw("SSMQ_FUSED_WSPLIT")) forced = atoi(ev);
    if (forced == 0) return false;
    if (forced > 0) return e.W == forced;
    // Measured (tools/wsplit_time.py, profiles/r05_wsplit.txt): the split pays where a point costs hundreds of instructions - the
    // t-process form, two N x N quadratic forms per output row: configs[3] 0.318 -> 0.272 ms with W = 2 - and loses 10-25 % on the
    // unscented and Bayes-Sard filters of the reentry model (~60 instructions per point against four barriers per step), so only
    // the former is picked by default; W = 4 of the t-process form needs more than 512 registers (spills) and is slower than W = 2.
    if (!(e.tp && e.W == 2)) return false;
    return (B + 63) / 64 <= (int64_t)cus * ws_groups_per_cu(e);
}

// Returns 1 if the wave-split kernel was launched (or, dry_run, would be), 0 if not applicable, < 0 on error.
int try_launch_wsplit(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho, const ssmq_integrand *fo,
                      int sel_obs, int64_t B, int64_t ld, int T, const double *d_y, const double *d_m0, const double *d_P0,
                      const double *d_gqg, const double *d_rr, double *d_fm, double *d_fP, int32_t *d_status, hipStream_t s,
                      const char **name, bool dry_run, const double *d_sscale, double student_dof, int cus) {
    if (hd->form != ho->form || (hd->tp_nu > 0.0) != (ho->tp_nu > 0.0) || sel_obs < 0 || fd->n_idx > 0 || B < 1) return 0;
    if ((d_sscale != nullptr) != (student_dof > 0.0)) return 0;
    const int tp = hd->tp_nu > 0.0 ? 1 : 0;
    const WsplitEntry *pick = nullptr;
    for (const WsplitEntry &e : kWsplit)
        if (e.fd == fd->id && e.fo == fo->id && e.D == hd->D && e.Y == ho->E && e.ND == hd->N && e.NO == ho->N && e.form == hd->form &&
            e.tp == tp && e.selo == sel_obs && ws_wanted(e, B, cus) && (!pick || e.W > pick->W))
            pick = &e;
    if (!pick) return 0;
    {
        {
            const WsplitEntry &e = *pick;
            if (name) *name = e.name;
            if (dry_run) return 1;
            FusedArgs a;
            a.y = d_y; a.m0 = d_m0; a.P0 = d_P0; a.fm = d_fm; a.fP = d_fP; a.status = d_status;
            a.c_dyn = hd->d_small; a.c_obs = ho->d_small; a.gqg = d_gqg; a.rr = d_rr; a.B = B; a.ld = ld; a.T = T;
            a.emv_dyn = hd->emv_mode; a.emv_obs = ho->emv_mode; a.nu_dyn = hd->tp_nu; a.nu_obs = ho->tp_nu;
            a.sscale = d_sscale; a.student_dof = student_dof;
            a.lpw = 64;
            fill_fpar(fd, &a.fd);
            fill_fpar(fo, &a.fo);
            a.fd.ttab = nullptr;
            a.fo.ttab = nullptr;
            int rc = hip_fail(e.fn(a, s), e.name);
            return rc ? rc : 1;
        }
    }
    return 0;
}

}  // namespace ssmq
