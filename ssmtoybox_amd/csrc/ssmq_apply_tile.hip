// Whole moment transform for point sets of up to 64 points with every product on the matrix cores
// (v_mfma_f64_16x16x4_f64): the generic shapes between the register-resident kernels (ssmq_apply_small.h) and the
// batch GEMM of large point sets (ssmq_gemm_mfma.hip) - Gauss-Hermite grids, fully-symmetric degree-5 sets in low
// dimension, BQ transforms at D = 7 ... 16 with 2 D (+ 1) points (BASELINE configs[4], unisolvent half: Bayes-Sard,
// D = E = 10, N = 21).  k_apply_wave ran these with lanes over output entries out of LDS: 2 200 wave instructions per
// trajectory, 5 % of either roofline.  Here (bq/bqmtran.py:158-223, mtran.py:141-149):
//
//   a wave takes G = min(4, 64 / N) trajectories at a time.  Lanes in G groups: inputs -> LDS slice, Cholesky, then
//   lane (g, n) owns sigma point n of trajectory g: x_n = m + L xi_n, f(x_n) in registers, written to the
//   trajectory's FX tile in LDS in the order the matrix instruction wants its B operand: lane (c, q) of a 16 x 4
//   fragment reads FX[c][4 s + q], s = 0 .. KS - 1, as consecutive doubles.
//   Per trajectory, with f[s] those fragments (the SAME registers serve every product below as the B operand):
//     T' = Wc FX'        NB x KS instructions, A = fragments of Wc (LDS, shared by the workgroup).  The accumulator
//                        of row block blk, register r, lane (c, q) is T[c][16 blk + 4 r + q] - exactly the A operand
//                        (row c, k = q) of the step that sums over n = 4 (4 blk + r) .. + 3, so
//     cov = T FX'        KS instructions straight from the accumulators, no transposition, no LDS round trip
//     P'  = Wcc FX'      KS instructions;  ccov' = L P'  ceil(D / 4) instructions (A = fragments of the factor)
//     mean               KS multiply-adds per lane + two cross-lane adds
//   t-process: S = fx iK fx' the same way (NB KS + KS more).  Centred form (classical rules): Wc = diag(wc),
//   Wcc = xi diag(wc), the mean subtracted from the fragments first.
// 27 matrix instructions per trajectory at D = E = 10, N = 21 instead of ~9 000 multiply-adds spread over lanes that
// were mostly idle.  The fp64 matrix rate equals the vector rate on CDNA4: what is gained is issue slots and
// lane utilisation, not peak.  Results differ from k_apply_wave / k_apply_wide at rounding level (other summation
// order inside the products); the covariance is formed for e2 <= e1 and mirrored, as every other kernel does.
#include "ssmq_device.h"
#include "ssmq_wide.h"

namespace ssmq {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
#ifndef SSMQ_TILE_ALIAS
#define SSMQ_TILE_ALIAS 0
#endif
#ifndef SSMQ_TILE_LATE_FETCH
#define SSMQ_TILE_LATE_FETCH 1      // round 5: 0.2507 -> 0.2385 ms (D = E = 10, N = 21, B = 1e5; tools/tile_ab.sh)
#endif
#ifndef SSMQ_TILE_PAD_TRAJ
#define SSMQ_TILE_PAD_TRAJ 0        // A/B builds (tools/tile_variants.sh): extra doubles per trajectory part / per wave slice - moves the
#endif                              // LDS banks the trajectories of a workgroup start on (step 4 reads 12 of them per instruction)
#ifndef SSMQ_TILE_PAD_WAVE
#define SSMQ_TILE_PAD_WAVE 0
#endif
constexpr int kTileWaves = 4;

struct TileGeom {
    int KS, KSP, NB, G, GL;
    int frag_doubles;     // workgroup-shared operand fragments
    int per_traj;         // a trajectory's part of the wave slice: factor + mean + FX tile (in_doubles), then its outputs [mean | cov | ccov]
    int in_doubles;
    int out_off;          // where in a trajectory's part its outputs go
    int wave_doubles;     // per-wave slice
};
// k-steps a point set is padded to (the kernel is instantiated for these; padding = zero fragments, no loop guards)
__host__ __device__ constexpr inline int tile_ksm(int N) {
    return N <= 16 ? 4 : N <= 24 ? 6 : N <= 32 ? 8 : N <= 52 ? 13 : 16;
}
__host__ __device__ constexpr inline TileGeom tile_geom(int D, int E, int N, bool tp, bool mrow = false) {
    TileGeom g{};
    g.KS = tile_ksm(N);
    g.KSP = g.KS | 1;                       // odd pitch: the 64 lanes of a fragment read fall on distinct banks
    g.NB = (g.KS + 3) / 4;
    g.G = 64 / N < 4 ? 64 / N : 4;
    g.GL = 64 / g.G;
    g.frag_doubles = 64 * ((tp ? 2 : 1) * g.NB * g.KS + (mrow ? 1 : 2) * g.KS) + N * (D | 1);     // + unit points [N][D | 1]; no wm fragments where the mean rides in Wcc
    g.frag_doubles = (g.frag_doubles + 1) & ~1;
    g.in_doubles = D * D + D + E * 4 * g.KSP;
#if SSMQ_TILE_ALIAS
    // the outputs of a trajectory overwrite its own inputs (read into registers by then): 54 KB per workgroup at D = E = 10,
    // N = 21 instead of 74 - three workgroups per CU
    g.out_off = 0;
    g.per_traj = g.in_doubles > E + E * E + E * D ? g.in_doubles : E + E * E + E * D;
#else
    g.out_off = g.in_doubles;
    g.per_traj = g.in_doubles + E + E * E + E * D;
#endif
    g.per_traj += SSMQ_TILE_PAD_TRAJ;
    g.wave_doubles = g.G * g.per_traj + 2 + 64 + 16 + SSMQ_TILE_PAD_WAVE;   // per trajectory: factor + mean + FX tile; status words, column slots
    g.wave_doubles = (g.wave_doubles + 1) & ~1;
    return g;
}

#define SSMQ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// DM: compile-time bound on D and E; KS: k-steps of 4 points the point set is padded to (tile_ksm(N));
// FC: integrand fixed at compile time, or -1
#ifndef SSMQ_TILE_OCC
#define SSMQ_TILE_OCC 2      // waves per SIMD the register allocation is held to
#endif
#ifndef SSMQ_TILE_WGS_PER_CU
#define SSMQ_TILE_WGS_PER_CU 2      // the resident number at this register count: every wave walks ~16 groups
#endif
// NX > 0 (round 6): the EXACT shape D = E = DM, N = NX in the BQ form without the t-process variance, mean riding in the Wcc
// operand - everything the geometry, the index maps and the masks are made of is then a compile-time constant.  With the shape
// at run time the kernel keeps ~60 wave-uniform quantities in scalar registers, spills 167 of them and reads them back
// (v_readlane) inside its loops; BASELINE configs[4]'s unisolvent half (D = E = 10, N = 21) gets the exact variant.
template <int DM, int KS, int FC = -1, int NX = 0>
__global__ __launch_bounds__(64 * kTileWaves, SSMQ_TILE_OCC) void k_apply_tile(const WideArgs a, int64_t B) {
    extern __shared__ __align__(16) double lds[];
    constexpr int KSM = KS, NBM = (KS + 3) / 4, NB = NBM, KSP = KS | 1;
    constexpr bool kExact = NX > 0;
    const int D = kExact ? DM : a.D, E = kExact ? DM : a.E, N = kExact ? NX : a.N;
    const bool tp = kExact ? false : a.tp_nu > 0.0, sigma = kExact ? false : a.form == SSMQ_FORM_SIGMA;
    // BQ form, D <= 15: row 15 of the Wcc operand is free and carries wm, so the transformed mean comes out of the
    // cross-covariance product (accumulator register 3 of the lanes q = 3) instead of a separate sum + cross-lane adds
    const bool mrow = kExact ? true : (!sigma && D <= 15 && a.wave_k == 0);     // (wave_k = 1: SSMQ_TILE_NO_MROW, to test the D = 16 path on smaller models)
    const TileGeom tg = tile_geom(D, E, N, tp, mrow);
    const int G = tg.G, GL = tg.GL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const WideLayout cl = wide_layout(D, E, N, a.form);
    const double *cs = a.consts;
    const double nan = __builtin_nan("");
    const int kq = mrow ? 3 : 0;                      // the k sub-index whose lanes hold the mean for the m m' product

    // ---- operand fragments of the constants, once per workgroup ---------------------------------------------------------
    double *fWc = lds;                              // [NB][KS][64]  lane (c, q): Wc[16 blk + c][4 s + q]
    double *fIK = fWc + NB * KS * 64;               // the same of iK (t-process only)
    double *fWcc = fIK + (tp ? NB * KS * 64 : 0);   // [KS][64]      lane (c, q): Wcc[c][4 s + q]
    double *fWm = fWcc + KS * 64;                   // [KS][64]      lane (c, q): wm[4 s + q]  (not with mrow)
    double *sXi = fWm + (mrow ? 0 : KS * 64);       // [N][D | 1]    unit sigma points
    for (int i = threadIdx.x; i < N * D; i += 64 * kTileWaves) sXi[(i / D) * (D | 1) + i % D] = cs[cl.xiT + i];
    for (int i = threadIdx.x; i < NB * KS * 64; i += 64 * kTileWaves) {
        const int l = i & 63, s = (i >> 6) % KS, blk = (i >> 6) / KS;
        const int row = 16 * blk + (l & 15), col = 4 * s + (l >> 4);
        const bool in = row < N && col < N;
        double w = 0.0;
        if (in) w = sigma ? (row == col ? cs[cl.Wc + row] : 0.0) : cs[cl.Wc + row * N + col];
        fWc[i] = w;
        if (tp) fIK[i] = in ? cs[cl.iK + row * N + col] : 0.0;
    }
    for (int i = threadIdx.x; i < KS * 64; i += 64 * kTileWaves) {
        const int l = i & 63, s = i >> 6;
        const int d = l & 15, n = 4 * s + (l >> 4);
        double w = 0.0;
        if (d < D && n < N) w = sigma ? cs[cl.xiT + n * D + d] * cs[cl.Wc + n] : cs[cl.Wcc + d * N + n];
        if (mrow && d == 15 && n < N) w = cs[cl.wm + n];      // BQ form: the mean rides along as row 15 of P' = Wcc FX'
        fWcc[i] = w;
        if (!mrow) fWm[i] = n < N ? cs[cl.wm + n] : 0.0;
    }
    // ---- the wave's slice: factor + mean and FX tile per trajectory --------------------------------------------------------
    double *wbase = lds + tg.frag_doubles + (size_t)wave * tg.wave_doubles;
    const int per_traj = tg.per_traj;
    for (int i = lane; i < tg.wave_doubles; i += 64) wbase[i] = 0.0;     // the tile's padding (n >= N) stays zero for good
    __syncthreads();

    // entries of the model variance / additive term this lane finishes: rows e1 = q + 4 r, column e2 = c
    double emv_r[4], add_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e1 = q + 4 * r, e2 = c;
        const bool in = e1 < E && e2 < E;
        const bool use = !sigma && in && ((e1 == e2) || a.emv_mode == SSMQ_EMV_BROADCAST);
        emv_r[r] = use ? cs[cl.emv + e1 * E + e2] : 0.0;
        add_r[r] = (in && a.cov_add) ? a.cov_add[e1 * E + e2] : 0.0;
    }
    const double tp_den = tp ? 1.0 / (a.tp_nu - 2.0 + (double)N) : 0.0;
    // what this lane stores per trajectory: plane indices and a mask (bit r: covariance entry (e1, e2) with e2 <= e1,
    // bit 4 + r: its mirror image, bit 8 + r: cross-covariance entry (c, q + 4 r))
    int pl_cov[4], pl_cvt[4], pl_cc[4], smask = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e1 = q + 4 * r, e2 = c;
        pl_cov[r] = e1 * E + e2;
        pl_cvt[r] = e2 * E + e1;
        pl_cc[r] = c * D + e1;
        if (e1 < E && e2 <= e1) smask |= 1 << r;
        if (e1 < E && e2 < e1) smask |= 16 << r;
        if (c < E && e1 < D) smask |= 256 << r;
    }

    const int gi = lane / GL, gl = lane - gi * GL;
    const int64_t n_groups = (B + G - 1) / G;
    // Every wave of the workgroup runs the same number of iterations (the stores below are a workgroup affair); a wave whose
    // group lies beyond the batch walks through the barriers only.
    const int64_t grp0 = (int64_t)blockIdx.x * kTileWaves + wave, grp_step = (int64_t)gridDim.x * kTileWaves;
    const int64_t base_end = n_groups;      // (loop bound on the workgroup's FIRST group of an iteration)
    // Inputs: lane (g, i), i < D, owns row i of trajectory g's covariance (lower triangle) and entry i of its mean.  The
    // rows of the NEXT group are requested before this group's matrix work starts, so the HBM round trip is never waited for.
    const uint32_t es8 = (uint32_t)(a.es_in * 8), eo8 = (uint32_t)(a.es_out * 8);   // plane pitches in bytes (< 2^32: launcher)
    // Every lane loads from a valid address (trajectory and row clamped, entries beyond the diagonal re-read the diagonal
    // one: same cache line): no divergent branches around the loads; what a lane does not own is never used.
    double in_row[DM], in_m = 0.0;
    const int gl_row = gl < D ? gl : D - 1;
    auto fetch = [&](int64_t grp) {
        int64_t bq = grp * G + (gi < G ? gi : G - 1);
        bq = bq < B ? bq : B - 1;
        const char *pm = (const char *)(a.mean + bq) + (uint64_t)(uint32_t)gl_row * es8;
        const char *pc = (const char *)(a.cov + bq) + (uint64_t)((uint32_t)gl_row * (uint32_t)D) * es8;
        in_m = *(const double *)pm;
#pragma unroll
        for (int k = 0; k < DM; ++k) in_row[k] = *(const double *)(pc + (uint64_t)(uint32_t)(k < gl_row ? k : gl_row) * es8);
    };
    double *scol = wbase + G * per_traj + 2;        // [64] one slot per lane: the current elimination column
    fetch(grp0 < n_groups ? grp0 : n_groups - 1);
    // outputs: planes [mean (E) | cov (E E) | ccov (E D)] of the workgroup's 4 G consecutive trajectories, see step 4
    const int n_planes = E + E * E + E * D;
    const bool out16 = ((((uintptr_t)a.mean_f) | ((uintptr_t)a.cov_f) | ((uintptr_t)a.cov_fx)) & 15) == 0 && (a.es_out & 1) == 0;
    for (int64_t grp = grp0; grp - wave < base_end; grp += grp_step) {
        const int64_t b0 = grp * G;
        // ---- 1. Cholesky, one row per lane in registers; column j travels through the slice (right-looking, the subtractions
        //         in the order k = 0, 1, ... of the left-looking dot products: the factor of the other generic kernels) --------
        const int64_t b = b0 + gi;
        const bool active = gi < G && b < B;
        double *sL = wbase + (gi < G ? gi : 0) * per_traj;      // D*D factor (pitch D), zeros above the diagonal
        double *sm = sL + D * D;
        double *sfx = sm + D;                                   // [E][4][KSP]
        const double *gcol = scol + gi * GL;                    // the group's column: entry k at gcol[k]
        double rowv[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) rowv[k] = in_row[k];
        const double my_m = in_m;
#if !SSMQ_TILE_LATE_FETCH
        fetch(grp + grp_step < n_groups ? grp + grp_step : (grp < n_groups ? grp : n_groups - 1));
#endif
        bool ok = true;
#pragma unroll
        for (int j = 0; j < DM; ++j) {
            if (j < D) {                                        // wave-uniform
                scol[lane] = rowv[j];                           // column j before scaling, every lane its own slot
                SSMQ_WAVE_SYNC();
                const double ajj = active ? gcol[j] : 1.0;
                ok = ok && (ajj > 0.0);
                double ljj, rinv;
                sqrt_rsqrt(ajj, ljj, rinv);
                const double lij = (gl == j) ? ljj : rowv[j] * rinv;
                rowv[j] = lij;
#pragma unroll
                for (int k = j + 1; k < DM; ++k)                // l_kj formed here as lane k forms it for itself
                    rowv[k] = fma(-lij, gcol[k] * rinv, rowv[k]);    // entries k > gl (and k >= D) are never used
                SSMQ_WAVE_SYNC();
            }
        }
        if (active && gl < D) {
#pragma unroll
            for (int k = 0; k < DM; ++k)
                if (k < D) sL[gl * D + k] = k <= gl ? rowv[k] : 0.0;
            sm[gl] = my_m;
        }
        if (active && gl == 0 && a.status) a.status[b] = ok ? 0 : 1;
        SSMQ_WAVE_SYNC();
        // ---- 2. lane (g, n): sigma point and integrand, values into the trajectory's tile -------------------------------------
        if (active && gl < N) {
            const double t = a.time ? a.time[a.time_stride ? b : 0] : 0.0;
            const int n = gl;
            double xin[DM], x[DM], o[DM];
#pragma unroll
            for (int k = 0; k < DM; ++k) xin[k] = k < D ? sXi[n * (D | 1) + k] : 0.0;
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
                    for (int qq = 1; qq < DM; ++qq) v = (src == qq) ? x[qq] : v;
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
            const int pos = (n & 3) * KSP + (n >> 2);
#pragma unroll
            for (int e = 0; e < DM; ++e)                 // a covariance that is not positive definite poisons every output
                if (e < E) sfx[e * 4 * KSP + pos] = ok ? o[e] : nan;
#if SSMQ_TILE_ALIAS
            // the previous iteration's outputs lay over this tile: its padding (points N ... 4 KS - 1) is zeroed again
            for (int np = N + gl; np < 4 * KS; np += GL) {
                const int pp = (np & 3) * KSP + (np >> 2);
#pragma unroll
                for (int e = 0; e < DM; ++e)
                    if (e < E) sfx[e * 4 * KSP + pp] = 0.0;
            }
#endif
        }
        SSMQ_WAVE_SYNC();
        // The next group's inputs (requested at the top of this iteration) are taken into registers NOW, while nothing else is
        // outstanding: the compiler would otherwise wait for them at their first use, after this iteration's stores have been
        // issued - and the in-order memory counter then makes that wait cover every store acknowledgement as well.
#if SSMQ_TILE_LATE_FETCH
        // (requested HERE: the rows are not live across the integrand, the kernel's register peak; the matrix phase covers the trip)
        fetch(grp + grp_step < n_groups ? grp + grp_step : (grp < n_groups ? grp : n_groups - 1));
#else
#pragma unroll
        for (int k = 0; k < DM; ++k) asm volatile("" : "+v"(in_row[k]));
        asm volatile("" : "+v"(in_m));
#endif
        // ---- 3. one trajectory at a time on the matrix cores; lane (c, q) = column c, k sub-index q ------------------------------
        for (int g = 0; g < G; ++g) {
            const int64_t bb = b0 + g;
            if (bb >= B) break;                                  // wave-uniform
            const double *tL = wbase + g * per_traj, *tfx = tL + D * D + D;
            double f[KSM];
#pragma unroll
            for (int s = 0; s < KSM; ++s) f[s] = c < E ? tfx[(c * 4 + q) * KSP + s] : 0.0;
            double mc = 0.0;
            if (!mrow) {
                // mean: column c, the four k sub-indices summed across the lane groups
                double mpart = 0.0;
#pragma unroll
                for (int s = 0; s < KSM; ++s) mpart = fma(fWm[s * 64 + lane], f[s], mpart);
                mpart += __shfl_xor(mpart, 16, 64);
                mpart += __shfl_xor(mpart, 32, 64);
                mc = mpart;
                if (sigma) {
#pragma unroll
                    for (int s = 0; s < KSM; ++s)
                        if (4 * s + q < N) f[s] -= mc;
                }
            }
            v4d acc[NBM], cov = {0.0, 0.0, 0.0, 0.0}, sq = {0.0, 0.0, 0.0, 0.0};
            auto quadratic = [&](const double *fA, v4d &out) {     // out = FX A FX' (A symmetric N x N, as fragments)
#pragma unroll
                for (int blk = 0; blk < NBM; ++blk) {
                    acc[blk] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < KSM; ++s)
                        acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(fA[(blk * KS + s) * 64 + lane], f[s], acc[blk], 0, 0, 0);
                }
#pragma unroll
                for (int blk = 0; blk < NBM; ++blk)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * blk + r < KS) out = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[blk][r], f[4 * blk + r], out, 0, 0, 0);
            };
            quadratic(fWc, cov);
            if (tp) quadratic(fIK, sq);
            // cross-covariance: P' = Wcc FX' (rows d; row 15 = the mean), then ccov' = L P'
            v4d pacc = {0.0, 0.0, 0.0, 0.0}, cc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KSM; ++s) pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(fWcc[s * 64 + lane], f[s], pacc, 0, 0, 0);
            if (mrow) mc = pacc[3];                               // lanes q = 3: row 15 of P', column c
#pragma unroll
            for (int t4 = 0; t4 < (DM + 3) / 4; ++t4) {
                if (4 * t4 < D) {
                    const int d = 4 * t4 + q;
                    const double lf = (c < D && d < D) ? tL[c * D + d] : 0.0;
                    cc = __builtin_amdgcn_mfma_f64_16x16x4f64(lf, pacc[t4], cc, 0, 0, 0);
                }
            }
            // cov - m m': one more step of the same accumulation (A = -m, B = m in the lanes of one k sub-index)
            const double am = (q == kq) ? mc : 0.0;
            if (!sigma) cov = __builtin_amdgcn_mfma_f64_16x16x4f64(-am, am, cov, 0, 0, 0);
            // ---- outputs of this trajectory into its part of the slice: [mean | cov (e2 <= e1 mirrored) | ccov] ------------------------
            double *so = wbase + g * per_traj + tg.out_off;
            if (q == kq && c < E) so[c] = mc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (smask & (1 << r)) {
                    double v;
                    if (sigma) {
                        v = cov[r] * a.cov_scale + add_r[r];
                    } else {
                        double em = emv_r[r];
                        if (tp) em = (a.tp_nu - 2.0 + sq[r]) * tp_den * em;
                        v = (cov[r] + em) * a.cov_scale + add_r[r];
                    }
                    so[E + pl_cov[r]] = v;
                    if (smask & (16 << r)) so[E + pl_cvt[r]] = v;
                }
                if (smask & (256 << r)) so[E + E * E + pl_cc[r]] = cc[r] * a.ccov_scale;
            }
        }
        // ---- 4. stores, by the whole workgroup: its four waves hold the outputs of 4 G CONSECUTIVE trajectories; per plane these are
        //         4 G consecutive doubles (96 bytes at G = 3), written as 16-byte pieces by neighbouring lanes.  (Round 3 stored
        //         from the accumulators: every lane 8 bytes to a plane of its own, 24-byte fragments per plane and wave - 1.03 M
        //         store instructions and 426 MB of partial-sector writes for a 168 MB payload at D = E = 10, N = 21, B = 1e5.) -----
#if SSMQ_TILE_LATE_FETCH
#pragma unroll
        for (int k = 0; k < DM; ++k) asm volatile("" : "+v"(in_row[k]));      // (taken before the stores are issued, see above)
        asm volatile("" : "+v"(in_m));
#endif
        __syncthreads();
        {
            const int64_t bc = (grp - wave) * G;                        // first trajectory of the workgroup's chunk
            const int64_t left = B - bc;
            const int nt = (int)(left < (int64_t)(kTileWaves * G) ? left : (int64_t)(kTileWaves * G));
            const int pp = (nt + 1) >> 1;                               // pieces of two trajectories per plane
            const double *sall = lds + tg.frag_doubles + tg.out_off;
            // GC > 0: a full chunk of 4 GC trajectories with the group count at compile time - the index maps below are divisions
            // by constants (multiply-shift); at run-time divisors they were 150 of the kernel's 540 vector instructions per
            // trajectory (profiles/r04_tile_sq.txt), in a kernel whose vector and matrix instructions share one pipe
            auto store_chunk = [&](auto gc_c) {
                constexpr int GC = decltype(gc_c)::value;
                const int gg = GC > 0 ? GC : G, ppc = GC > 0 ? 2 * GC : pp;
                for (int id = threadIdx.x; id < n_planes * ppc; id += 64 * kTileWaves) {
                    const int p = id / ppc, k = id - p * ppc;
                    const int t0 = 2 * k, t1 = t0 + 1;
                    const int w0 = t0 / gg, w1 = t1 / gg;
                    const double v0 = sall[(size_t)w0 * tg.wave_doubles + (t0 - w0 * gg) * per_traj + p];
                    const double v1 = t1 < nt ? sall[(size_t)w1 * tg.wave_doubles + (t1 - w1 * gg) * per_traj + p] : 0.0;
                    double *dst;
                    if (p < E) dst = (double *)((char *)(a.mean_f + bc + t0) + (uint64_t)(uint32_t)p * eo8);
                    else if (p < E + E * E) dst = (double *)((char *)(a.cov_f + bc + t0) + (uint64_t)(uint32_t)(p - E) * eo8);
                    else dst = (double *)((char *)(a.cov_fx + bc + t0) + (uint64_t)(uint32_t)(p - E - E * E) * eo8);
                    if (t1 < nt && out16) {
                        typedef double v2d __attribute__((ext_vector_type(2)));
                        *(v2d *)dst = v2d{v0, v1};
                    } else {
                        dst[0] = v0;
                        if (t1 < nt) dst[1] = v1;
                    }
                }
            };
            const bool full = nt == kTileWaves * G;
            if (full && G == 3) store_chunk(std::integral_constant<int, 3>{});
            else if (full && G == 4) store_chunk(std::integral_constant<int, 4>{});
            else if (full && G == 2) store_chunk(std::integral_constant<int, 2>{});
            else if (full && G == 1) store_chunk(std::integral_constant<int, 1>{});
            else store_chunk(std::integral_constant<int, 0>{});
        }
        __syncthreads();         // the next group's inputs and outputs overwrite the slices
    }
}

template <int DM, int KS, int FC, int NX = 0>
hipError_t launch_tile_one(const WideArgs &a, int64_t B, hipStream_t s) {
    WideArgs aw = a;
    aw.wave_k = ssmq::sw("SSMQ_TILE_NO_MROW") ? 1 : 0;
    const bool mrow = a.form != SSMQ_FORM_SIGMA && a.D <= 15 && aw.wave_k == 0;
    const TileGeom tg = tile_geom(a.D, a.E, a.N, a.tp_nu > 0.0, mrow);
    const size_t lds = sizeof(double) * ((size_t)tg.frag_doubles + (size_t)kTileWaves * tg.wave_doubles);
    if (lds > 48 * 1024) {     // per device and instantiation; a cheap call, rare shapes
        hipError_t e = hipFuncSetAttribute((const void *)k_apply_tile<DM, KS, FC, NX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 64);
        if (e != hipSuccess) return e;
    }
    const int64_t groups = (B + tg.G - 1) / tg.G, blocks = (groups + kTileWaves - 1) / kTileWaves;
    // a few workgroups per CU, each walking its share of the batch: the constants' fragments are built once per workgroup
    const int64_t cap = 256 * SSMQ_TILE_WGS_PER_CU;
    if (ssmq::sw("SSMQ_TILE_DEBUG")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_apply_tile<DM, KS, FC, NX>, 64 * kTileWaves, lds);
        fprintf(stderr, "k_apply_tile: %zu bytes of LDS, %d workgroups per CU\n", lds, nb);
    }
    hipLaunchKernelGGL((k_apply_tile<DM, KS, FC, NX>), dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(64 * kTileWaves), lds, s,
                       aw, B);
    return hipGetLastError();
}

template <int KS>
hipError_t launch_tile_ks(const WideArgs &a, int64_t B, hipStream_t s) {
    const int dm = a.D > a.E ? a.D : a.E;
    if constexpr (KS == 6) {
        // BASELINE configs[4], unisolvent half, as an exact shape (the kernel's comment); SSMQ_TILE_NO_EXACT=1: the run-time-shape body
        if (a.fid == SSMQ_F_SMOOTH10D_DYN && a.fp.n_idx == 0 && a.D == 10 && a.E == 10 && a.N == 21 && a.form == SSMQ_FORM_BQ && !(a.tp_nu > 0.0) &&
            !ssmq::sw("SSMQ_TILE_NO_MROW") && !ssmq::sw("SSMQ_TILE_NO_EXACT"))
            return launch_tile_one<10, KS, SSMQ_F_SMOOTH10D_DYN, 21>(a, B, s);
        if (a.fid == SSMQ_F_SMOOTH10D_DYN && a.fp.n_idx == 0 && dm <= 10) return launch_tile_one<10, KS, SSMQ_F_SMOOTH10D_DYN>(a, B, s);
    }
    if (dm <= 4) return launch_tile_one<4, KS, -1>(a, B, s);
    if (dm <= 8) return launch_tile_one<8, KS, -1>(a, B, s);
    if (dm <= 12) return launch_tile_one<12, KS, -1>(a, B, s);
    return launch_tile_one<SSMQ_MAX_DIM, KS, -1>(a, B, s);
}

}  // namespace

size_t tile_lds_bytes(int D, int E, int N, bool tp) {
    const TileGeom tg = tile_geom(D, E, N, tp);
    return sizeof(double) * ((size_t)tg.frag_doubles + (size_t)kTileWaves * tg.wave_doubles);
}

// whole transforms (built-in integrand, one constant block for the batch) of this shape run on the matrix cores
bool wide_full_uses_tile(int D, int E, int N) {
    return N > 8 && N <= 64 && D <= 16 && E <= 16 && !ssmq::sw("SSMQ_NO_TILE") && !ssmq::sw("SSMQ_NO_WAVE") &&
           tile_lds_bytes(D, E, N, true) <= 160 * 1024 - 64;
}

// plane pitches must fit 32 bits in bytes (the kernel forms addresses as 32 x 32 -> 64-bit products)
// ... and consecutive trajectories must be consecutive doubles of a plane (the library's batch layout)
bool tile_pitch_ok(const WideArgs &a) {
    return tile_ld_ok(a.es_in) && tile_ld_ok(a.es_out) && a.bs_mean == 1 && a.bs_cov == 1 && a.bs_mf == 1 &&
           a.bs_cf == 1 && a.bs_cfx == 1;
}

hipError_t launch_apply_tile(const WideArgs &a, int64_t B, hipStream_t s) {
    switch (tile_ksm(a.N)) {
        case 4: return launch_tile_ks<4>(a, B, s);
        case 6: return launch_tile_ks<6>(a, B, s);
        case 8: return launch_tile_ks<8>(a, B, s);
        case 13: return launch_tile_ks<13>(a, B, s);
        default: return launch_tile_ks<16>(a, B, s);
    }
}

}  // namespace ssmq
