// Generic batched moment transform: run-time D, E, N and integrand id, one 64-lane workgroup per trajectory, the
// sigma points / integrand values / intermediate products resident in LDS.  This is the path for shapes that have no
// register-resident specialisation (ssmq_apply_small.h) - e.g. Bayes-Sard quadrature at D = 10 with N = 21 or 201 -
// and for the split entry points that serve an arbitrary Python integrand (sigma points out, reductions in).
//
// Work split inside a workgroup: lanes over sigma points n for x_n = mean + L xi_n and f(x_n); lanes over output
// entries (e, j) for T = fx Wc (Wc rows streamed from L2, coalesced over j) and for every (E x E) / (E x D) result.
#include "ssmq_device.h"
#include "ssmq_wide.h"
#include "ssmq_update.h"

namespace ssmq {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// E rows of N values each, row pitch fl in global memory (one contiguous E * fl block), to a dense [E][N] LDS image.
// Eight independent loads are in flight per lane before the first LDS store: one load at a time costs a full memory
// round trip per 64 values (the FX pass of the N = 201 case spent most of its time right here).
template <int BLOCK>
__device__ __forceinline__ void rows_to_lds(const double *__restrict__ src, int64_t fl, int E, int N, double *dst, int lane) {
    const int64_t total = (int64_t)E * fl;
    if ((fl & 1) == 0 && (((uintptr_t)src) & 15) == 0 && total <= 16 * 2 * BLOCK) {
        // the whole block as 16-byte loads, ALL of them in flight before the first LDS store: one memory round trip
        // for up to 32 * BLOCK doubles (the GEMM route's 10 x 208 rows: 9 loads per lane)
        const double2 *s2 = (const double2 *)src;
        const int64_t pairs = total / 2;
        double2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t i = (int64_t)q * BLOCK + lane;
            v[q] = i < pairs ? s2[i] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t i = (int64_t)q * BLOCK + lane;
            if (i < pairs) {
                const int e = (int)((2 * i) / fl), n = (int)(2 * i - (int64_t)e * fl);     // fl even: a pair never straddles rows
                if (n < N) dst[e * N + n] = v[q].x;
                if (n + 1 < N) dst[e * N + n + 1] = v[q].y;
            }
        }
        return;
    }
    for (int64_t base = 0; base < total; base += 8 * BLOCK) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t i = base + q * BLOCK + lane;
            v[q] = i < total ? src[i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t i = base + q * BLOCK + lane;
            if (i < total) {
                const int e = (int)(i / fl), n = (int)(i - (int64_t)e * fl);
                if (n < N) dst[e * N + n] = v[q];
            }
        }
    }
}

// BLOCK = 64 (one wave: small shapes, many trajectories per CU) or 256 (large point sets: the LDS footprint allows only
// ~3 trajectories per CU, so each one must bring its own four waves or the CU idles behind LDS / memory latency)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_apply_wide(const WideArgs a) {
    extern __shared__ __align__(16) double lds[];
    const int D = a.D, E = a.E, N = a.N;
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    // LDS image; the pass-specific parts are left out where a pass does not touch them (wide_lds_bytes_for): fewer
    // bytes = more trajectories resident per CU
    const bool need_x = a.mode != SSMQ_WIDE_FX || a.form == SSMQ_FORM_SIGMA;
    double *sL = lds;               // D*D   lower factor, row-major, zeros above the diagonal
    double *sm = sL + D * D;        // D
    double *sx = sm + D;            // D*N   sigma points (not in the FX pass of the BQ form)
    double *sfx = sx + (need_x ? D * N : 0);   // E*N   integrand values (centred in place for the SIGMA form)
    const bool need_T = a.mode == SSMQ_WIDE_FULL || a.mode == SSMQ_WIDE_FX;
    double *sT = sfx + E * N;       // E*N   fx Wc  (then fx iK for the TP model variance); FULL / FX passes only
    double *smf = sT + (need_T ? E * N : 0);   // E; from here on: not in EVAL / POINTS
    double *sS = smf + E;           // E*E   TP quadratic form
    double *sC = sS + E * E;        // E*E   fx Wc fx'
    double *sg = sC + E * E;        // E*D
    __shared__ int s_ok;
    const double *c = a.consts + b * a.consts_stride;   // per-trajectory weights: theta-batched callers
    const WideLayout cl = wide_layout(D, E, N, a.form);
    const double nan = __builtin_nan("");

    // ---- 1. inputs -> LDS, Cholesky --------------------------------------------------------------------------
    if (a.mode != SSMQ_WIDE_FX) {
        for (int d = lane; d < D; d += BLOCK) sm[d] = a.mean[d * a.es_in + b * a.bs_mean];
        for (int i = lane; i < D * D; i += BLOCK) {
            const int r = i / D, cc = i % D;
            sL[i] = (cc <= r) ? a.cov[(int64_t)i * a.es_in + b * a.bs_cov] : 0.0;
        }
        __syncthreads();
        // right-looking Cholesky across the wave: column j is scaled by all lanes, the trailing block updated by all
        // lanes (the subtractions reach every element in the order k = 0, 1, ... of the left-looking dot products, so
        // the factor is the same to the bit); a serial lane-0 loop costs ~D^3 / 3 dependent LDS round trips
        {
            bool ok = true;
            for (int j = 0; j < D; ++j) {
                const double ajj = sL[j * D + j];
                ok = ok && (ajj > 0.0);
                const double ljj = sqrt(ajj), r = 1.0 / ljj;
                __syncthreads();
                if (lane == 0) sL[j * D + j] = ljj;
                for (int i = j + 1 + lane; i < D; i += BLOCK) sL[i * D + j] *= r;
                __syncthreads();
                const int m = D - j - 1;
                for (int idx = lane; idx < m * m; idx += BLOCK) {
                    const int i = j + 1 + idx / m, k = j + 1 + idx % m;
                    if (k <= i) sL[i * D + k] -= sL[i * D + j] * sL[k * D + j];
                }
                __syncthreads();
            }
            if (lane == 0) {
                s_ok = ok ? 1 : 0;
                if (a.status) a.status[b] = ok ? 0 : 1;
            }
        }
        __syncthreads();
        // ---- 2. sigma points and integrand -----------------------------------------------------------------------
        const double t = a.time ? a.time[a.time_stride ? b : 0] : 0.0;
        for (int n = lane; n < N; n += BLOCK) {
            double xin[SSMQ_MAX_DIM];              // point n's coordinates: D independent loads, one round trip (PMC,
                                                   // tools/pmc_wide.sh: the pass waited on 55 dependent loads per point)
#pragma unroll
            for (int k = 0; k < SSMQ_MAX_DIM; ++k) xin[k] = k < D ? c[cl.xiT + n * D + k] : 0.0;
            for (int d = 0; d < D; ++d) {
                double s = sm[d];
#pragma unroll
                for (int k = 0; k < SSMQ_MAX_DIM; ++k)
                    if (k <= d) s += sL[d * D + k] * xin[k];
                sx[d * N + n] = s;
            }
            if (a.mode == SSMQ_WIDE_FULL || a.mode == SSMQ_WIDE_EVAL) {
                double xs[kMaxIntegrandIn], o[SSMQ_MAX_DIM];
#pragma unroll
                for (int k = 0; k < kMaxIntegrandIn; ++k) {
                    const int src = a.fp.n_idx > 0 ? (k < a.fp.n_idx ? a.fp.idx[k] : 0) : (k < D ? k : 0);
                    xs[k] = sx[src * N + n];
                }
#pragma unroll
                for (int e = 0; e < SSMQ_MAX_DIM; ++e) o[e] = 0.0;
                eval_integrand(a.fid, xs, t, a.fp, o);
#pragma unroll
                for (int e = 0; e < SSMQ_MAX_DIM; ++e)
                    if (e < E) sfx[e * N + n] = o[e];
            }
        }
        __syncthreads();
        if (a.mode == SSMQ_WIDE_EVAL) {
            // integrand values as (b E + e)-th row of the batch matrix, zero-padded to the GEMM's column count
            const int64_t fl = a.fx_ld ? a.fx_ld : N;
            for (int e = 0; e < E; ++e)
                for (int n = lane; n < fl; n += BLOCK)
                    a.fx_out[((int64_t)b * E + e) * fl + n] = n < N ? (s_ok ? sfx[e * N + n] : nan) : 0.0;
            for (int i = lane; i < D * D; i += BLOCK) a.chol_out[b * D * D + i] = s_ok ? sL[i] : nan;
            return;
        }
        if (a.mode == SSMQ_WIDE_POINTS) {
            // outputs in the reference layout: x [b][D][N], chol [b][D][D]
            for (int i = lane; i < D * N; i += BLOCK) a.x_out[b * D * N + i] = s_ok ? sx[i] : nan;
            for (int i = lane; i < D * D; i += BLOCK) a.chol_out[b * D * D + i] = s_ok ? sL[i] : nan;
            return;
        }
    } else {
        // reductions only: L, fx (and x, mean for the centred form) come from the caller, reference layout
        for (int i = lane; i < D * D; i += BLOCK) sL[i] = a.chol_in[b * D * D + i];
        {
            const int64_t fl = a.fx_ld ? a.fx_ld : N;
            rows_to_lds<BLOCK>(a.fx_in + (int64_t)b * E * fl, fl, E, N, sfx, lane);
        }
        if (a.form == SSMQ_FORM_SIGMA) {
            for (int i = lane; i < D * N; i += BLOCK) sx[i] = a.x_in[b * D * N + i];
            for (int d = lane; d < D; d += BLOCK) sm[d] = a.mean[b * D + d];
        }
        if (lane == 0) s_ok = 1;
        __syncthreads();
    }
    const bool ok = s_ok != 0;
#define OUT_ADDR(ptr, e, bs) ptr[(int64_t)(e) * a.es_out + b * (bs)]

    // ---- 3. mean ---------------------------------------------------------------------------------------------
    for (int e = lane >> 6; e < E; e += BLOCK / 64) {      // one wave per output row
        double s = 0.0;
        for (int n = lane & 63; n < N; n += 64) s += sfx[e * N + n] * c[cl.wm + n];
        s = wave_sum(s);
        if ((lane & 63) == 0) smf[e] = s;
    }
    __syncthreads();
    for (int e = lane; e < E; e += BLOCK) OUT_ADDR(a.mean_f, e, a.bs_mf) = ok ? smf[e] : nan;

    if (a.form == SSMQ_FORM_BQ) {
        // ---- 4. T = fx Wc; cov = T fx' - mean mean' + emv -------------------------------------------------------
        if (a.t_in) {      // fx Wc came from the matrix-core GEMM over the whole batch
            const int64_t fl = a.fx_ld ? a.fx_ld : N;
            rows_to_lds<BLOCK>(a.t_in + (int64_t)b * E * fl, fl, E, N, sT, lane);
        } else {
            for (int idx = lane; idx < E * N; idx += BLOCK) {
                const int e = idx / N, j = idx % N;
                double s = 0.0;
                for (int i = 0; i < N; ++i) s += sfx[e * N + i] * c[cl.Wc + (int64_t)i * N + j];
                sT[idx] = s;
            }
        }
        __syncthreads();
        // (fx Wc) fx' is symmetric (Wc is): lower triangle only, mirrored - as the register kernels do
        for (int idx = lane; idx < E * (E + 1) / 2; idx += BLOCK) {
            int e = 0;
            while ((e + 1) * (e + 2) / 2 <= idx) ++e;
            const int e2 = idx - e * (e + 1) / 2;
            double s = 0.0;
            for (int j = 0; j < N; ++j) s += sT[e * N + j] * sfx[e2 * N + j];
            sC[e * E + e2] = s;
            sC[e2 * E + e] = s;
        }
        __syncthreads();
        if (a.tp_nu > 0.0) {
            for (int idx = lane; idx < E * N; idx += BLOCK) {
                const int e = idx / N, j = idx % N;
                double s = 0.0;
                for (int i = 0; i < N; ++i) s += sfx[e * N + i] * c[cl.iK + (int64_t)i * N + j];
                sT[idx] = s;
            }
            __syncthreads();
            for (int idx = lane; idx < E * E; idx += BLOCK) {
                const int e = idx / E, e2 = idx % E;
                double s = 0.0;
                for (int j = 0; j < N; ++j) s += sT[e * N + j] * sfx[e2 * N + j];
                sS[idx] = s;
            }
            __syncthreads();
        }
        for (int idx = lane; idx < E * E; idx += BLOCK) {
            const int e = idx / E, e2 = idx % E;
            const bool use = (e == e2) || (a.emv_mode == SSMQ_EMV_BROADCAST);
            double em = use ? c[cl.emv + idx] : 0.0;
            if (a.tp_nu > 0.0) em = (a.tp_nu - 2.0 + sS[idx]) * (1.0 / (a.tp_nu - 2.0 + (double)N)) * em;
            double v = (sC[idx] - smf[e] * smf[e2] + em) * a.cov_scale;
            if (a.cov_add) v += a.cov_add[idx];
            OUT_ADDR(a.cov_f, idx, a.bs_cf) = ok ? v : nan;
        }
        // ---- 5. cross-covariance (fx Wcc') L' ------------------------------------------------------------------
        for (int idx = lane; idx < E * D; idx += BLOCK) {
            const int e = idx / D, d = idx % D;
            double s = 0.0;
            for (int n = 0; n < N; ++n) s += sfx[e * N + n] * c[cl.Wcc + d * N + n];
            sg[idx] = s;
        }
        __syncthreads();
        for (int idx = lane; idx < E * D; idx += BLOCK) {
            const int e = idx / D, j = idx % D;
            double s = 0.0;
            for (int d = 0; d <= j; ++d) s += sg[e * D + d] * sL[j * D + d];
            OUT_ADDR(a.cov_fx, idx, a.bs_cfx) = ok ? s * a.ccov_scale : nan;
        }
    } else {
        // ---- classical centred form (mtran.py:141-149), Wc = diag(wc) -------------------------------------------
        for (int idx = lane; idx < E * N; idx += BLOCK) sfx[idx] -= smf[idx / N];
        __syncthreads();
        for (int idx = lane; idx < E * E; idx += BLOCK) {
            const int e = idx / E, e2 = idx % E;
            double s = 0.0;
            for (int n = 0; n < N; ++n) s += (sfx[e * N + n] * c[cl.Wc + n]) * sfx[e2 * N + n];
            s *= a.cov_scale;
            if (a.cov_add) s += a.cov_add[idx];
            OUT_ADDR(a.cov_f, idx, a.bs_cf) = ok ? s : nan;
        }
        for (int idx = lane; idx < E * D; idx += BLOCK) {
            const int e = idx / D, d = idx % D;
            double s = 0.0;
            for (int n = 0; n < N; ++n) s += (sfx[e * N + n] * c[cl.Wc + n]) * (sx[d * N + n] - sm[d]);
            OUT_ADDR(a.cov_fx, idx, a.bs_cfx) = ok ? s * a.ccov_scale : nan;
        }
    }
#undef OUT_ADDR
}

// ---- evaluation pass of the two-pass matrix-core route: ONE WAVE per trajectory -------------------------------------------
// Cholesky factor, sigma points, integrand values (written as rows b E + e of the batch matrix FX, zero-padded to the
// GEMM's column count), transformed mean.  Everything the covariances need beyond that happens in the GEMM's epilogue
// (ssmq_gemm_mfma.hip).  No workgroup barrier anywhere: the waves of a block are independent trajectories, the factor
// lives in a 2 KB LDS slice of the wave, sigma points and integrand values never touch LDS (lane n owns point n), so a
// CU keeps 8+ trajectories in flight instead of the 3 that the LDS image of k_apply_wide allows - that kernel spent
// its time waiting (10 us per trajectory behind ~35 barriers and serial LDS / L2 round trips, 277 us for B = 1e4).
constexpr int kEvalWaves = 4;
// DM: compile-time bound on D and E (the unrolled per-point loops run to DM, predicated on the run-time sizes)
// FC: integrand fixed at compile time (no switch, no select chain for a state index it does not take), or -1
#ifndef SSMQ_EVAL_OCC
#define SSMQ_EVAL_OCC 0          // A/B builds: waves per SIMD requested of the compiler (0: its own choice, 4 at 110 registers;
                                // 5 / 6 / 8 spill and cost the N = 1181 route 2 / 10 / 14 %: round 4, tools/c5_deg7.py)
#endif
template <int DM, int FC = -1>
__global__ __launch_bounds__(64 * kEvalWaves, (SSMQ_EVAL_OCC ? SSMQ_EVAL_OCC : 1)) void k_eval_wave(const WideArgs a, int64_t B) {
    __shared__ double s_all[kEvalWaves][SSMQ_MAX_DIM * SSMQ_MAX_DIM + SSMQ_MAX_DIM];
    const int D = a.D, E = a.E, N = a.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * kEvalWaves + wave;
    if (b >= B) return;                         // whole waves leave
    double *sL = s_all[wave], *sm = sL + SSMQ_MAX_DIM * SSMQ_MAX_DIM;
    const double *c = a.consts + b * a.consts_stride;
    const WideLayout cl = wide_layout(D, E, N, a.form);
    const double nan = __builtin_nan("");
    // wave-scope ordering of the LDS slice: LDS instructions of one wave execute in order, the fence keeps the compiler
    // from moving accesses across it
#define SSMQ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    for (int d = lane; d < D; d += 64) sm[d] = a.mean[d * a.es_in + b * a.bs_mean];
    for (int i = lane; i < D * D; i += 64) {
        const int r = i / D, cc = i % D;
        sL[i] = (cc <= r) ? a.cov[(int64_t)i * a.es_in + b * a.bs_cov] : 0.0;
    }
    SSMQ_WAVE_SYNC();
    // right-looking Cholesky, the subtraction order of the left-looking dot products (as k_apply_wide: same factor)
    bool ok = true;
    for (int j = 0; j < D; ++j) {
        const double ajj = sL[j * D + j];
        ok = ok && (ajj > 0.0);
        const double ljj = sqrt(ajj), r = 1.0 / ljj;
        SSMQ_WAVE_SYNC();
        if (lane == 0) sL[j * D + j] = ljj;
        for (int i = j + 1 + lane; i < D; i += 64) sL[i * D + j] *= r;
        SSMQ_WAVE_SYNC();
        const int m = D - j - 1;
        for (int idx = lane; idx < m * m; idx += 64) {
            const int i = j + 1 + idx / m, k = j + 1 + idx % m;
            if (k <= i) sL[i * D + k] -= sL[i * D + j] * sL[k * D + j];
        }
        SSMQ_WAVE_SYNC();
    }
    if (lane == 0 && a.status) a.status[b] = ok ? 0 : 1;
    for (int i = lane; i < D * D; i += 64) a.chol_out[b * D * D + i] = ok ? sL[i] : nan;
    const double t = a.time ? a.time[a.time_stride ? b : 0] : 0.0;
    const int64_t fl = a.fx_ld ? a.fx_ld : N;
    double *rows = a.fx_out + b * E * fl;
    double macc[DM];
#pragma unroll
    for (int e = 0; e < DM; ++e) macc[e] = 0.0;
    for (int n0 = 0; n0 < fl; n0 += 64) {
        const int n = n0 + lane;
        double o[DM];
#pragma unroll
        for (int e = 0; e < DM; ++e) o[e] = 0.0;
        if (n < N) {
            double xin[DM], x[DM];
#pragma unroll
            for (int k = 0; k < DM; ++k) xin[k] = k < D ? c[cl.xiT + n * D + k] : 0.0;
#pragma unroll
            for (int d = 0; d < DM; ++d) {
                double s = 0.0;
                if (d < D) {
                    s = sm[d];
#pragma unroll
                    for (int k = 0; k < DM; ++k)
                        if (k <= d) s += sL[d * D + k] * xin[k];
                }
                x[d] = s;
            }
            double xs[kMaxIntegrandIn];
#pragma unroll
            for (int k = 0; k < kMaxIntegrandIn; ++k) {
                double v = k < DM ? x[k < DM ? k : 0] : 0.0;
                if (FC < 0 && a.fp.n_idx > 0) {    // state-index selection (MeasurementModel.state_index)
                    const int src = k < a.fp.n_idx ? a.fp.idx[k] : 0;
                    v = x[0];
#pragma unroll
                    for (int q = 1; q < DM; ++q) v = (src == q) ? x[q] : v;
                }
                xs[k] = v;
            }
            if constexpr (FC >= 0) {
                Fn<FC> fn;
                fn.init(t, a.fp);
                fn.template eval<SSMQ_MAX_FIDX>(xs, o);
            } else {
                eval_integrand(a.fid, xs, t, a.fp, o);
            }
            const double w = c[cl.wm + n];
#pragma unroll
            for (int e = 0; e < DM; ++e) macc[e] += o[e] * w;
        }
        if (n < fl) {
            if (a.fx_frag) {
                const int64_t blk = b / a.fx_frag;
                const int lr0 = (int)(b - blk * a.fx_frag) * E;
                double *tile0 = a.fx_out + blk * 64 * fl + (int64_t)(n >> 4) * 256 + 64 * ((n & 15) >> 2) + (n & 3);
#pragma unroll
                for (int e = 0; e < DM; ++e)
                    if (e < E) {
                        const int lr = lr0 + e;
                        tile0[(int64_t)(lr >> 4) * 16 * fl + 4 * (lr & 15)] = n < N ? (ok ? o[e] : nan) : 0.0;
                    }
            } else {
#pragma unroll
                for (int e = 0; e < DM; ++e)
                    if (e < E) rows[(int64_t)e * fl + n] = n < N ? (ok ? o[e] : nan) : 0.0;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < DM; ++e) {
        if (e < E) {
            const double s = wave_sum(macc[e]);
            if (lane == 0) {
                a.mean_f[(int64_t)e * a.es_out + b * a.bs_mf] = ok ? s : nan;
                a.mrow_out[b * E + e] = ok ? s : nan;
            }
        }
    }
#undef SSMQ_WAVE_SYNC
}

hipError_t launch_eval_wave(const WideArgs &a, int64_t B, hipStream_t s) {
    const dim3 grid((unsigned)((B + kEvalWaves - 1) / kEvalWaves)), block(64 * kEvalWaves);
    const int dm = a.D > a.E ? a.D : a.E;
    if (a.fid == SSMQ_F_SMOOTH10D_DYN && a.fp.n_idx == 0 && dm <= 10)
        hipLaunchKernelGGL((k_eval_wave<10, SSMQ_F_SMOOTH10D_DYN>), grid, block, 0, s, a, B);
    else if (dm <= 4) hipLaunchKernelGGL(k_eval_wave<4>, grid, block, 0, s, a, B);
    else if (dm <= 8) hipLaunchKernelGGL(k_eval_wave<8>, grid, block, 0, s, a, B);
    else if (dm <= 10) hipLaunchKernelGGL(k_eval_wave<10>, grid, block, 0, s, a, B);
    else if (dm <= 12) hipLaunchKernelGGL(k_eval_wave<12>, grid, block, 0, s, a, B);
    else hipLaunchKernelGGL(k_eval_wave<SSMQ_MAX_DIM>, grid, block, 0, s, a, B);
    return hipGetLastError();
}

// ---- whole transform, ONE WAVE per trajectory, for point sets of up to 64 points ---------------------------------------
// The shapes without a register-resident specialisation that are not large either: Gauss-Hermite grids, fully-symmetric
// degree-5 sets in low dimension, BQ transforms at D = 7 ... 16 with 2 D (+ 1) points (the unisolvent Bayes-Sard case of
// BASELINE configs[4]: D = 10, N = 21), the theta-batched step (per-trajectory weights).  Same arithmetic and summation
// orders as k_apply_wide, but lane n owns sigma point n through Cholesky -> point -> integrand (registers), the products
// run with lanes over output columns / entries out of a wave-private LDS slice, and nothing ever waits at a workgroup
// barrier: four independent trajectories per block, as k_eval_wave.
constexpr int kWaveWaves = 4;
// A point set of N <= 32 points leaves half of the wave idle, so a wave carries K <= 64 / N trajectories side by side:
// "group" g = lane / G, G = 64 / K lanes, works on trajectory K (4 block + wave) + g with its own LDS slice; the group's
// first N lanes own its sigma points / columns, all G lanes share its entry loops.
__host__ __device__ inline int wave_lds_doubles(int D, int E, int N, bool tp) {      // per group
    const int m = E > D ? E : D;
    return D * D + D + E + E * m + (tp ? E * E : 0) + (E + m) * N + 2;
}

// The body, for the wave with global index `wid` (trajectories wid K ... wid K + K - 1) and its LDS slice `lw` (K
// groups of wave_lds_doubles): k_apply_wave below runs it once, k_theta_chain twice in a row.  False when the wave has no
// trajectory (nothing done, nothing synchronised).
template <int DM, int FC>
__device__ __forceinline__ bool apply_wave_body(const WideArgs &a, int64_t B, int64_t wid, double *lw) {
    const int D = a.D, E = a.E, N = a.N;
    const int lane = threadIdx.x & 63;
    const int K = a.wave_k, G = 64 / K, gi = lane / G, gl = lane - gi * G;   // G >= N lanes per trajectory
    const int64_t b0 = wid * K;
    if (b0 >= B) return false;                  // whole waves leave; no workgroup barrier below
    const int64_t b = b0 + gi;
    const bool active = gi < K && b < B;        // idle lanes still take part in the wave-scope synchronisation
    const int stride = (wave_lds_doubles(D, E, N, a.tp_nu > 0.0) + 1) & ~1;
    double *sL = lw + (size_t)(gi < K ? gi : 0) * stride;   // D*D factor (pitch D)
    double *sm = sL + D * D;                    // D    input mean
    double *smf = sm + D;                       // E    transformed mean
    double *sC = smf + E;                       // E*E  fx Wc fx', then (same place) sg: E*D fx Wcc'
    double *sg = sC;
    double *sS = sC + E * (E > D ? E : D);      // E*E  t-process quadratic form (only then)
    double *sfx = sS + (a.tp_nu > 0.0 ? E * E : 0);   // E*N  integrand values (centred in place: SIGMA)
    double *sA = sfx + E * N;                   // BQ: E*N fx Wc;  SIGMA: D*N points minus mean
    const double *c = a.consts + (active ? b : b0) * a.consts_stride;
    const WideLayout cl = wide_layout(D, E, N, a.form);
    const double nan = __builtin_nan("");
#define SSMQ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define OUT_ADDR(ptr, e, bs) ptr[(int64_t)(e) * a.es_out + b * (bs)]
    // ---- 1. inputs, Cholesky (as k_eval_wave), the group's lanes over the entries ---------------------------------------------
    if (active) {
        for (int d = gl; d < D; d += G) sm[d] = a.mean[d * a.es_in + b * a.bs_mean];
        for (int i = gl; i < D * D; i += G) {
            const int r = i / D, cc = i % D;
            sL[i] = (cc <= r) ? a.cov[(int64_t)i * a.es_in + b * a.bs_cov] : 0.0;
        }
    }
    SSMQ_WAVE_SYNC();
    bool ok = true;
    for (int j = 0; j < D; ++j) {
        const double ajj = active ? sL[j * D + j] : 1.0;
        ok = ok && (ajj > 0.0);
        const double ljj = sqrt(ajj), r = 1.0 / ljj;
        SSMQ_WAVE_SYNC();
        if (active) {
            if (gl == 0) sL[j * D + j] = ljj;
            for (int i = j + 1 + gl; i < D; i += G) sL[i * D + j] *= r;
        }
        SSMQ_WAVE_SYNC();
        if (active) {
            const int m = D - j - 1;
            for (int idx = gl; idx < m * m; idx += G) {
                const int i = j + 1 + idx / m, k = j + 1 + idx % m;
                if (k <= i) sL[i * D + k] -= sL[i * D + j] * sL[k * D + j];
            }
        }
        SSMQ_WAVE_SYNC();
    }
    if (active && gl == 0 && a.status) a.status[b] = ok ? 0 : 1;
    // ---- 2. lane n of the group: sigma point and integrand -----------------------------------------------------------------
    if (active && gl < N) {
        const double t = a.time ? a.time[a.time_stride ? b : 0] : 0.0;
        const int n = gl;
        double xin[DM], x[DM], o[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) xin[k] = k < D ? c[cl.xiT + n * D + k] : 0.0;
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            double sacc = 0.0;
            if (d < D) {
                sacc = sm[d];
#pragma unroll
                for (int k = 0; k < DM; ++k)
                    if (k <= d) sacc += sL[d * D + k] * xin[k];
            }
            x[d] = sacc;
            o[d] = 0.0;
        }
        double xs[kMaxIntegrandIn];
#pragma unroll
        for (int k = 0; k < kMaxIntegrandIn; ++k) {
            double v = k < DM ? x[k < DM ? k : 0] : 0.0;
            if (FC < 0 && a.fp.n_idx > 0) {        // state-index selection (MeasurementModel.state_index)
                const int src = k < a.fp.n_idx ? a.fp.idx[k] : 0;
                v = x[0];
#pragma unroll
                for (int q = 1; q < DM; ++q) v = (src == q) ? x[q] : v;
            }
            xs[k] = v;
        }
        if constexpr (FC >= 0) {
            Fn<FC> fn;
            fn.init(t, a.fp);
            fn.template eval<SSMQ_MAX_FIDX>(xs, o);
        } else {
            eval_integrand(a.fid, xs, t, a.fp, o);
        }
#pragma unroll
        for (int e = 0; e < DM; ++e)
            if (e < E) sfx[e * N + n] = o[e];
        if (a.form == SSMQ_FORM_SIGMA) {
#pragma unroll
            for (int d = 0; d < DM; ++d)
                if (d < D) sA[d * N + n] = x[d] - sm[d];
        }
    }
    SSMQ_WAVE_SYNC();
    // ---- 3. mean: the group's lanes over the output rows -------------------------------------------------------------------
    if (active) {
        for (int e = gl; e < E; e += G) {
            double sacc = 0.0;
            for (int n = 0; n < N; ++n) sacc += sfx[e * N + n] * c[cl.wm + n];
            smf[e] = sacc;
            OUT_ADDR(a.mean_f, e, a.bs_mf) = ok ? sacc : nan;
        }
    }
    SSMQ_WAVE_SYNC();
    if (a.form == SSMQ_FORM_BQ) {
        // ---- 4. T = fx Wc: lane j owns column j (rows of Wc read coalesced), then (fx Wc) fx' over the lower triangle ---------
        auto times_matrix = [&](int off) {
            if (!active || gl >= N) return;
            double acc[DM];
#pragma unroll
            for (int e = 0; e < DM; ++e) acc[e] = 0.0;
            for (int i = 0; i < N; ++i) {
                const double w = c[off + (int64_t)i * N + gl];
#pragma unroll
                for (int e = 0; e < DM; ++e)
                    if (e < E) acc[e] += sfx[e * N + i] * w;
            }
#pragma unroll
            for (int e = 0; e < DM; ++e)
                if (e < E) sA[e * N + gl] = acc[e];
        };
        auto lower_products = [&](double *dst) {       // dst[e][e2] = sum_j sA[e][j] sfx[e2][j], e2 <= e, mirrored
            if (!active) return;
            for (int idx = gl; idx < E * (E + 1) / 2; idx += G) {
                int e = 0;
                while ((e + 1) * (e + 2) / 2 <= idx) ++e;
                const int e2 = idx - e * (e + 1) / 2;
                double sacc = 0.0;
                for (int j = 0; j < N; ++j) sacc += sA[e * N + j] * sfx[e2 * N + j];
                dst[e * E + e2] = sacc;
                dst[e2 * E + e] = sacc;
            }
        };
        times_matrix(cl.Wc);
        SSMQ_WAVE_SYNC();
        lower_products(sC);
        if (a.tp_nu > 0.0) {
            SSMQ_WAVE_SYNC();
            times_matrix(cl.iK);
            SSMQ_WAVE_SYNC();
            if (active) {
                for (int idx = gl; idx < E * E; idx += G) {
                    const int e = idx / E, e2 = idx % E;
                    double sacc = 0.0;
                    for (int j = 0; j < N; ++j) sacc += sA[e * N + j] * sfx[e2 * N + j];
                    sS[idx] = sacc;
                }
            }
        }
        SSMQ_WAVE_SYNC();
        if (active) {
            for (int idx = gl; idx < E * E; idx += G) {
                const int e = idx / E, e2 = idx % E;
                const bool use = (e == e2) || (a.emv_mode == SSMQ_EMV_BROADCAST);
                double em = use ? c[cl.emv + idx] : 0.0;
                if (a.tp_nu > 0.0) em = (a.tp_nu - 2.0 + sS[idx]) * (1.0 / (a.tp_nu - 2.0 + (double)N)) * em;
                double v = (sC[idx] - smf[e] * smf[e2] + em) * a.cov_scale;
                if (a.cov_add) v += a.cov_add[idx];
                OUT_ADDR(a.cov_f, idx, a.bs_cf) = ok ? v : nan;
            }
        }
        SSMQ_WAVE_SYNC();                               // sg takes the place of sC
        if (active) {
            // ---- 5. cross-covariance (fx Wcc') L' --------------------------------------------------------------------------
            for (int idx = gl; idx < E * D; idx += G) {
                const int e = idx / D, d = idx % D;
                double sacc = 0.0;
                for (int n = 0; n < N; ++n) sacc += sfx[e * N + n] * c[cl.Wcc + d * N + n];
                sg[idx] = sacc;
            }
        }
        SSMQ_WAVE_SYNC();
        if (active) {
            for (int idx = gl; idx < E * D; idx += G) {
                const int e = idx / D, j = idx % D;
                double sacc = 0.0;
                for (int d = 0; d <= j; ++d) sacc += sg[e * D + d] * sL[j * D + d];
                OUT_ADDR(a.cov_fx, idx, a.bs_cfx) = ok ? sacc * a.ccov_scale : nan;
            }
        }
    } else {
        // ---- classical centred form (mtran.py:141-149), Wc = diag(wc) ----------------------------------------------------------
        if (active)
            for (int idx = gl; idx < E * N; idx += G) sfx[idx] -= smf[idx / N];
        SSMQ_WAVE_SYNC();
        if (active) {
            for (int idx = gl; idx < E * E; idx += G) {
                const int e = idx / E, e2 = idx % E;
                double sacc = 0.0;
                for (int n = 0; n < N; ++n) sacc += (sfx[e * N + n] * c[cl.Wc + n]) * sfx[e2 * N + n];
                sacc *= a.cov_scale;
                if (a.cov_add) sacc += a.cov_add[idx];
                OUT_ADDR(a.cov_f, idx, a.bs_cf) = ok ? sacc : nan;
            }
            for (int idx = gl; idx < E * D; idx += G) {
                const int e = idx / D, d = idx % D;
                double sacc = 0.0;
                for (int n = 0; n < N; ++n) sacc += (sfx[e * N + n] * c[cl.Wc + n]) * sA[d * N + n];
                OUT_ADDR(a.cov_fx, idx, a.bs_cfx) = ok ? sacc * a.ccov_scale : nan;
            }
        }
    }
#undef OUT_ADDR
#undef SSMQ_WAVE_SYNC
    return true;
}

template <int DM, int FC = -1>
__global__ __launch_bounds__(64 * kWaveWaves) void k_apply_wave(const WideArgs a, int64_t B) {
    extern __shared__ __align__(16) double lds[];
    const int wave = threadIdx.x >> 6;
    const int stride = (wave_lds_doubles(a.D, a.E, a.N, a.tp_nu > 0.0) + 1) & ~1;
    apply_wave_body<DM, FC>(a, B, (int64_t)blockIdx.x * kWaveWaves + wave, lds + (size_t)wave * a.wave_k * stride);
}

// trajectories per wave: as many as there are N-lane groups, but not so many that the LDS slices leave the SIMDs with
// fewer than ~3 waves each (the kernel waits on LDS / L2 round trips; measured at D = 10, N = 21: K = 1 / 2 / 3)
static int wave_groups(int D, int E, int N, bool tp) {
    if (const char *k = ssmq::sw("SSMQ_WAVE_K")) return std::max(1, std::min(64 / N, atoi(k)));
    const size_t slice = sizeof(double) * (size_t)((wave_lds_doubles(D, E, N, tp) + 1) & ~1);
    const size_t budget = (160 * 1024) / 12;                 // 12 waves per CU
    return (int)std::max<size_t>(1, std::min<size_t>(64 / N, budget / std::max<size_t>(slice, 1)));
}
static size_t wave_lds_bytes(int D, int E, int N, bool tp, int K) {
    return sizeof(double) * kWaveWaves * (size_t)K * (size_t)((wave_lds_doubles(D, E, N, tp) + 1) & ~1);
}
template <int DM, int FC>
static hipError_t launch_wave_one(const WideArgs &a, int64_t B, hipStream_t s) {
    WideArgs aw = a;
    aw.wave_k = wave_groups(a.D, a.E, a.N, a.tp_nu > 0.0);
    const size_t lds = wave_lds_bytes(a.D, a.E, a.N, a.tp_nu > 0.0, aw.wave_k);
    const int64_t per_block = (int64_t)kWaveWaves * aw.wave_k;
    if (lds > 48 * 1024) {     // beyond the default limit: raise it (per device and instantiation; a cheap call, rare shapes)
        hipError_t e = hipFuncSetAttribute((const void *)k_apply_wave<DM, FC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 64);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_apply_wave<DM, FC>), dim3((unsigned)((B + per_block - 1) / per_block)), dim3(64 * kWaveWaves), lds,
                       s, aw, B);
    return hipGetLastError();
}
// whole transforms (built-in integrand) of this shape run one wave per trajectory (k_apply_wave) rather than one workgroup
bool wide_full_uses_wave(int D, int E, int N) {
    return N >= 1 && N <= 64 && !ssmq::sw("SSMQ_NO_WAVE") && wave_lds_bytes(D, E, N, true, 1) <= 160 * 1024 - 64;
}
static bool wave_route(const WideArgs &a) { return a.mode == SSMQ_WIDE_FULL && wide_full_uses_wave(a.D, a.E, a.N); }
static hipError_t launch_apply_wave(const WideArgs &a, int64_t B, hipStream_t s) {
    const int dm = a.D > a.E ? a.D : a.E;
    hipError_t e;
    if (a.fid == SSMQ_F_SMOOTH10D_DYN && a.fp.n_idx == 0 && dm <= 10) e = launch_wave_one<10, SSMQ_F_SMOOTH10D_DYN>(a, B, s);
    else if (dm <= 4) e = launch_wave_one<4, -1>(a, B, s);
    else if (dm <= 8) e = launch_wave_one<8, -1>(a, B, s);
    else if (dm <= 12) e = launch_wave_one<12, -1>(a, B, s);
    else e = launch_wave_one<SSMQ_MAX_DIM, -1>(a, B, s);
    return e;
}

// ---- the theta-batched step after its weights, ONE launch (ssmq_gp_theta_step) ------------------------------
// time update -> measurement transform -> Kalman update -> log-likelihood of item b were four launches of a few
// microseconds each, every one waiting for the previous; here the wave that owns the item runs the two transforms back to
// back (apply_wave_body, handing m_pr / P_pr over through their global planes: release + acquire at agent scope around a
// wave barrier, the second transform's lanes read what other lanes of the same wave wrote) and the first lane of the
// item's group goes on with the update and the log-density (ssmq_update.h: the bodies of k_kalman_update /
// k_gauss_logpdf, so the results are the same bits as the launch-per-stage route's).
struct ThetaChainArgs {
    WideArgs dyn, obs;          // same wave_k
    UpdArgs upd;                // status = the update's own flag vector (set here, not accumulated)
    const double *y;
    double *loglik;
    const int32_t *merge;       // five flag vectors, pitch upd.ld
    int32_t *merge_out;
    int32_t lds_per_wave;       // doubles
    const int32_t *count;       // null, or the number of items on the device (B of the launch is then an upper bound)
};

template <int DM, bool GEN>
__global__ __launch_bounds__(64 * kWaveWaves) void k_theta_chain(const ThetaChainArgs c, int64_t B_launch) {
    extern __shared__ __align__(16) double lds[];
    const int64_t B = c.count ? (int64_t)*c.count : B_launch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wid = (int64_t)blockIdx.x * kWaveWaves + wave;
    double *lw = lds + (size_t)wave * c.lds_per_wave;
#ifndef SSMQ_CHAIN_SCOPE
// The predictive moments travel from the first transform to the second through their global planes, written and read back by
// lanes of ONE wave: workgroup scope orders that (one CU, one L1).  Agent scope - which on this part also writes the L2 back -
// was what rounds 3-4 used: 3 us more per call (65 -> 62 us at 5 items), same results.
#define SSMQ_CHAIN_SCOPE "workgroup"
#endif
#define SSMQ_AGENT_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, SSMQ_CHAIN_SCOPE); __builtin_amdgcn_wave_barrier(); \
                               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SSMQ_CHAIN_SCOPE); } while (0)
    const int K = c.dyn.wave_k, G = 64 / K, gi = lane / G, gl = lane - gi * G;
    const int64_t b = wid * K + gi;
    if (wid * K >= B) return;
    apply_wave_body<DM, -1>(c.dyn, B, wid, lw);
    SSMQ_AGENT_SYNC();
    apply_wave_body<DM, -1>(c.obs, B, wid, lw);
    SSMQ_AGENT_SYNC();
#undef SSMQ_AGENT_SYNC
    if (gi >= K || gl != 0 || b >= B) return;
    const uint32_t ub = (uint32_t)b;
    c.upd.status[ub] = 0;
    const int D = c.upd.D, Y = c.upd.Y;
    bool done = false;
    if constexpr (!GEN) {
#define SSMQ_UPD(d, y_) if (!done && d <= DM && D == d && Y == y_) { kalman_update_item<d, y_>(c.upd, ub); done = true; }
        SSMQ_UPD(1, 1) SSMQ_UPD(2, 1) SSMQ_UPD(2, 2) SSMQ_UPD(3, 1) SSMQ_UPD(4, 2) SSMQ_UPD(5, 2) SSMQ_UPD(5, 4) SSMQ_UPD(6, 2)
#undef SSMQ_UPD
    } else {
        kalman_update_item_generic(c.upd, ub);
    }
    gauss_logpdf_item(c.y, c.upd.y_mean, c.upd.P_y, c.loglik, Y, c.upd.ld, c.merge, c.merge_out, ub);
}

// the (D, Y) pairs with a register-resident update (launch_kalman_update_ex's table)
static bool update_is_specialised(int D, int Y) {
    return (D == 1 && Y == 1) || (D == 2 && (Y == 1 || Y == 2)) || (D == 3 && Y == 1) || (D == 4 && Y == 2) ||
           (D == 5 && (Y == 2 || Y == 4)) || (D == 6 && Y == 2);
}

bool theta_chain_supported(int Din, int D, int Y, int Nd, int No) {
    return !ssmq::sw("SSMQ_NO_THETA_FUSED") && wide_full_uses_wave(Din, D, Nd) && wide_full_uses_wave(D, Y, No) && Nd <= 64 && No <= 64;
}

template <int DM, bool GEN>
static hipError_t launch_chain_one(const ThetaChainArgs &c, int64_t B, size_t lds, hipStream_t s) {
    static thread_local unsigned attr_epoch = 0;
    if (lds > 48 * 1024 && attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_theta_chain<DM, GEN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 64);
        if (e != hipSuccess) return e;
        attr_epoch = device_epoch();
    }
    const int64_t per_block = (int64_t)kWaveWaves * c.dyn.wave_k;
    hipLaunchKernelGGL((k_theta_chain<DM, GEN>), dim3((unsigned)((B + per_block - 1) / per_block)), dim3(64 * kWaveWaves), lds, s,
                       c, B);
    return hipGetLastError();
}

// dyn / obs: WideArgs of the two transforms as for launch_apply_wide (mode FULL, form BQ); upd: the update's arguments
hipError_t launch_theta_chain(const WideArgs &dyn, const WideArgs &obs, const UpdArgs &upd, const double *y, double *loglik,
                              const int32_t *merge, int32_t *merge_out, int64_t B, hipStream_t s, const int32_t *d_count) {
    ThetaChainArgs c;
    c.dyn = dyn; c.obs = obs; c.upd = upd; c.y = y; c.loglik = loglik; c.merge = merge; c.merge_out = merge_out;
    c.count = d_count;
    const int K = std::min(wave_groups(dyn.D, dyn.E, dyn.N, false), wave_groups(obs.D, obs.E, obs.N, false));
    c.dyn.wave_k = c.obs.wave_k = K;
    const int per_item = std::max((wave_lds_doubles(dyn.D, dyn.E, dyn.N, false) + 1) & ~1, (wave_lds_doubles(obs.D, obs.E, obs.N, false) + 1) & ~1);
    c.lds_per_wave = K * per_item;
    const size_t lds = sizeof(double) * kWaveWaves * (size_t)c.lds_per_wave;
    const int dm = std::max(std::max(dyn.D, dyn.E), std::max(obs.D, obs.E));
    const bool gen = !update_is_specialised(upd.D, upd.Y);
    if (dm <= 4) return gen ? launch_chain_one<4, true>(c, B, lds, s) : launch_chain_one<4, false>(c, B, lds, s);
    if (dm <= 8) return gen ? launch_chain_one<8, true>(c, B, lds, s) : launch_chain_one<8, false>(c, B, lds, s);
    if (dm <= 12) return gen ? launch_chain_one<12, true>(c, B, lds, s) : launch_chain_one<12, false>(c, B, lds, s);
    return gen ? launch_chain_one<SSMQ_MAX_DIM, true>(c, B, lds, s) : launch_chain_one<SSMQ_MAX_DIM, false>(c, B, lds, s);
}

size_t wide_lds_bytes(int D, int E, int N) {
    return sizeof(double) * (size_t)(D * D + D + D * N + 2 * E * N + E + 2 * E * E + E * D);
}

// what one pass really needs (same carve-up as the kernel)
static size_t wide_lds_bytes_for(const WideArgs &a) {
    const size_t D = a.D, E = a.E, N = a.N;
    const bool need_x = a.mode != SSMQ_WIDE_FX || a.form == SSMQ_FORM_SIGMA;
    size_t n = D * D + D + (need_x ? D * N : 0) + E * N;
    if (a.mode == SSMQ_WIDE_FULL || a.mode == SSMQ_WIDE_FX) n += E * N;
    if (a.mode != SSMQ_WIDE_EVAL && a.mode != SSMQ_WIDE_POINTS) n += E + 2 * E * E + E * D;
    return sizeof(double) * n;
}

// hipFuncSetAttribute is per device: ensure_device() clears the flag when the current device changes
static thread_local unsigned attr_set_epoch = 0;       // (per thread context: ssmq_host.h)
void reset_wide_attributes() { attr_set_epoch = 0; }

hipError_t launch_apply_wide(const WideArgs &a, int64_t B, hipStream_t s) {
    if (a.mode == SSMQ_WIDE_FULL && a.consts_stride == 0 && wide_full_uses_tile(a.D, a.E, a.N) && tile_pitch_ok(a)) return launch_apply_tile(a, B, s);
    if (wave_route(a)) return launch_apply_wave(a, B, s);
    const size_t lds = wide_lds_bytes_for(a);
    if (attr_set_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_apply_wide<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 64);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_apply_wide<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 64);
        if (e != hipSuccess) return e;
        attr_set_epoch = device_epoch();
    }
    if ((int64_t)a.E * a.N >= 1024 && !ssmq::sw("SSMQ_WIDE_ONE_WAVE"))
        hipLaunchKernelGGL(k_apply_wide<256>, dim3((unsigned)B), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(k_apply_wide<64>, dim3((unsigned)B), dim3(64), lds, s, a);
    return hipGetLastError();
}

}  // namespace ssmq
