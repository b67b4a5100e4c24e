// Whole BQ moment transform for point sets of 65 ... 208 points in ONE launch: the integrand values never leave the CU.
//
// The two-pass route (k_eval_wave -> k_fxwc_cov_mfma, ssmq_gemm_mfma.hip) wrote the batch matrix FX ((B E) x NP doubles,
// 166 MB at D = E = 10, N = 201, B = 1e4) to HBM and read it back twice: 25 x the transform's algorithmic bytes, and the
// evaluation pass - some 20 us of arithmetic - took 76 us at 2.1 TB/s of writes.  Here the workgroup that owns a block of rows of
// the GEMM produces them itself (bq/bqmtran.py:132-156: x_n = m + L xi_n, f(x_n); :178-199, 223: mean, fx Wc fx', fx Wcc' L'):
//
//   workgroup = 8 waves, TPW whole trajectories = TPW E <= 64 rows of FX, kept in LDS ([row][NP + 2]; N = 201, E = 10: 6
//   trajectories, 60 rows - what fits 160 KB beside the slabs);
//   0. the first slab of X, this lane's first sigma point and the covariance's additive terms are requested;
//   1. one wave per trajectory: mean / covariance in; the Cholesky factorisations of all the tile's trajectories in the lanes
//      of ONE wave, in registers (D <= 10; else one wave per trajectory over LDS as k_eval_wave); factors packed in LDS for
//      steps 2 and 4;
//   2. the (trajectory, point) pairs over the workgroup's lanes (ten wave-iterations for 6 x 201 pairs): sigma point from
//      the LDS-resident factor, integrand, E values into the tile; zero padding;
//   3. the transposed product [T G]' = [S | Wcc' | wm]' FX' - S the lower triangle of Wc with half its diagonal, Wc = S + S', so
//      that fx Wc fx' = C + C' with C = (fx S) fx' and the zero k-blocks above the diagonal are neither loaded nor multiplied
//      (round 4: 104 tile steps instead of 182) - slabs of 16 rows of X double-buffered in LDS, requested two steps ahead, the
//      k-blocks walked long-short-long (12, 0, 11, 1, ...); wave w on row tile w & 3 and on the column tiles of ONE PARITY
//      (w >> 2): two waves per SIMD, 56 accumulator registers each.  The FX fragments come from the LDS tile: lane group lg
//      feeds column 16 kb + lg + 4 s in MFMA step s (stride 4, so that with the row pitch NP + 2 = 18 mod 32 the 32 lanes of
//      a ds_read_b64 half fall on 32 different bank pairs); the slab rows are stored permuted to match.  Column 15 of the
//      Wcc' tile of X is wm (D <= 15): the transformed mean is a by-product of the product;
//   4. cross-covariance from the G tile and the packed factors; C = T FX2' on the accumulators as in k_fxwc_cov_mfma, each
//      wave over its own column tiles; all parts meet in LDS, one thread per (trajectory, e >= e2) forms C + C' and stores.
// Grid: ceil(B / TPW) workgroups, one per CU: 1 667 at B = 1e4, E = 10 = 6.5 rounds of 256 (7 taken), against 3.05 rounds
// of 512 (4 taken) of the two-pass route.  Measured (tools/c5_full.py, tools/bqf_ab.sh with one step compiled out at a
// time; D = E = 10, N = 201, B = 1e4): 322 us against 337 us for the two passes; per tile ~27 us in the product (the matrix
// pipe 75 % busy there), 3 factorisation, 5 integrand, 8 epilogue (3 of them the 104 matrix instructions of S), 5 the rest;
// rocprofv3 counters: matrix pipe busy 43 % of the launch, other VALU 17 %, LDS 22 % (tools/pmc_bqf.sh).  A persistent
// variant (one workgroup per CU walking the tiles, next tile's inputs prefetched) measured 346 us (the hoisted per-thread
// addresses cost 66-117 spilled registers), dropped.
#include "ssmq_host.h"
#include "ssmq_wide.h"
#include <type_traits>

namespace ssmq {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

struct BqFusedArgs {
    WideArgs w;             // shape, integrand, constants (points, wm), inputs, outputs, scales - as for k_eval_wave
    const double *X;        // [NP][NP + 16] = [S | Wcc' | wm] zero-padded, S = tril(Wc) with half the diagonal (Wc = S + S')
    const double *emv;      // [E * E]
    int32_t emv_broadcast, tpw, fx_doubles;
    int64_t B;
};

// WAVES = 4: 256 threads, 32-row tiles, slabs of 8 rows of X - TWO workgroups per CU, so that one's factorisations,
// integrand evaluations and stores run under the other's matrix instructions; WAVES = 8: 512 threads, 64-row tiles, slabs
// of 16 rows, one workgroup per CU (half the L2 -> LDS traffic for X; for shapes whose 32-row tile does not fit 80 KB)
template <int NT, int DM, int FC, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void k_bq_fused(const BqFusedArgs g) {
    constexpr int TB = 64 * WAVES, RT = WAVES / 2, KS = 2 * WAVES, NKB = NT * 16 / KS, SPB = KS / 4;
    constexpr int NP = NT * 16, NX = NP + 16, NTX = NT + 1, LB = NX + 4, FP = NP + 2;
    // Column tiles of X = [S | Wcc' | wm] by PARITY: wave half ch owns the tiles ct = 2 t + ch (t = 0 .. C0 - 1, ct <= NT; tile NT
    // is the G tile [Wcc' | wm]).  S is the lower triangle of Wc with half its diagonal (Wc = S + S', so fx Wc fx' = C + C' with
    // C = (fx S) fx'): the k-block kb contributes to tile ct < NT only for 16 ct < KS (kb + 1) - NT (NT + 1) / 2 + NT tile steps
    // instead of NT (NT + 1), with both halves busy in every step.
    constexpr int C0 = (NTX + 1) / 2;
    constexpr int GCH = NT & 1, GT = NT >> 1;      // the G tile's half and local index
    constexpr int PK = DM * (DM + 1) / 2;          // packed lower triangle
    constexpr bool REGCHOL = DM <= 10;             // the factorisation in one lane's registers (else: the wave's lanes over LDS)
    extern __shared__ __align__(16) double lds[];
    const WideArgs &a = g.w;
    const int D = a.D, E = a.E, N = a.N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: everything per wave below branches, never masks
    const int li = lane & 15, lg = lane >> 4;
    const int lip = 4 * (li & 3) + (li >> 2);
    const int TPW = g.tpw, rows = TPW * E;
    const int64_t bw0 = (int64_t)blockIdx.x * TPW;
    const int nb = (int)((g.B - bw0) < (int64_t)TPW ? (g.B - bw0) : (int64_t)TPW);
    const int vrows = nb * E;
    double *sFX = lds;                              // [rows][FP]
    double *slab = lds + g.fx_doubles;              // [2][KS][LB]
    double *smr = slab + 2 * KS * LB;               // [16 RT] transformed means by row
    double *sLp = smr + 16 * RT;                    // [TPW][PK] factors, packed lower triangles (points AND the epilogue)
    double *sm = sLp + TPW * PK;                    // [TPW][DM] input means
    int *sok = (int *)(sm + TPW * DM);              // [TPW]
    int *srow = sok + 16;                           // [16 RT] row -> (trajectory << 8) | output index: no divisions in step 4
    double *sA = slab;                              // step 1 only: covariances in (the slab is not in use yet)
    const double nan = __builtin_nan("");
    const double *c = a.consts;
    const WideLayout cl = wide_layout(D, E, N, a.form);
    // ---- 0. what does not depend on the factors goes out first: the first slab of X, this lane's first sigma point --------
    static_assert(WAVES == 8, "slab staging and tile schedule below are written for 16-row slabs (KS = 16)");
    // Slabs of 16 rows of X are double-buffered in LDS and staged through registers: written right after the barrier that
    // released their buffer, so the LDS writes have a whole step to complete before the next barrier.
    // Thread (r, hf, c16) moves the columns 16 (2 j + hf) + c16 of slab row r, j = 0 .. C0 - 1: pairs of column tiles.  S is lower
    // triangular, so slab kb holds nothing right of column 16 (kb + 1): pair j is moved only for 2 j <= kb, and the last pair (the
    // G tile) always - conditions on the loop counter alone, i.e. scalar branches around whole instructions.
    const int sr = tid >> 5, shf = (tid >> 4) & 1, sc16 = tid & 15;
    // The k-blocks are walked in the order 12, 0, 11, 1, 10, 2, ... (kseq): a long step (many tiles of S reach it) next to a
    // short one, so that any two consecutive steps carry about the same number of matrix instructions, and the slabs are
    // requested TWO steps ahead into two register sets - a slab has a long + a short step (~1.8 us) to arrive, where one set and
    // the natural order left the loads of the short early steps (8 matrix instructions per SIMD at kb = 0) exposed.
    double breg[2][C0];
    auto phys = [](int k) { return 4 * (k & 3) + (k >> 2); };
    // row k of a slab lives at a permuted position, so that the lanes of a ds_read_b64 half (lane groups lg = 0, 1 or 2, 3,
    // reading k = lg + 4 s) are 16 bank pairs apart with the pitch LB = 4 mod 32
    auto kseq = [](int p) { return (p & 1) ? (p >> 1) : (NKB - 1 - (p >> 1)); };
    auto load_b = [&](int p) {                 // slab of position p into register set p & 1
        if (p < NKB) {
            const int kb = kseq(p);
            const double *src = g.X + ((int64_t)kb * KS + sr) * NX + 16 * shf + sc16;
#pragma unroll
            for (int j = 0; j < C0; ++j)
                if (2 * j <= kb || j == C0 - 1) breg[p & 1][j] = (2 * j + 1 < NTX || shf == 0) ? src[32 * j] : 0.0;
        }
    };
    auto park_b = [&](int p) {                 // ... and from there into LDS buffer p & 1
        const int kb = kseq(p);
        double *dst = slab + (p & 1) * KS * LB + phys(sr) * LB + 16 * shf + sc16;
#pragma unroll
        for (int j = 0; j < C0; ++j)
            if ((2 * j <= kb || j == C0 - 1) && (2 * j + 1 < NTX || shf == 0)) dst[32 * j] = breg[p & 1][j];
    };
    if (tid < 16 * RT) {
        const int gq = tid / E;
        srow[tid] = (gq << 8) | (tid - gq * E);
    }
    load_b(0);
    // the (trajectory, point) pairs of the tile in ONE index space over the workgroup's lanes, idx = tid, tid + TB, ...: ten
    // wave-iterations for 6 x 201 pairs.  (Vector instructions cost their four cycles whatever the number of active lanes,
    // and - tools/micro/mfma_valu_overlap.hip - they do not hide under matrix instructions: TB / TPW lanes per trajectory
    // with the factor in registers was 24 wave-iterations, and one factorisation per WAVE six times the instructions of
    // six factorisations in the lanes of one wave.)
    const float rN = 1.0f / (float)N;
    auto split = [&](int idx, int &gi, int &n) {           // idx -> (trajectory, point); exact for idx < 2^20
        gi = (int)(((float)idx + 0.5f) * rN);
        n = idx - gi * N;
    };
    double xin[DM];
    auto load_xi = [&](int n, double (&dst)[DM]) {
#pragma unroll
        for (int k = 0; k < DM; ++k) dst[k] = (k < D && n < N) ? c[cl.xiT + n * D + k] : 0.0;
    };
    {
        int gi0, n0;
        split(tid, gi0, n0);
        load_xi(tid < nb * N ? n0 : N, xin);
    }
    // the additive terms of the covariance (4c): in the store loop every one would be a dependent L2 round trip
    double ev = 0.0, ca = 0.0;
    if (tid < E * E) {
        ev = g.emv[tid];
        ca = a.cov_add ? a.cov_add[tid] : 0.0;
    }
    // ---- 1. factors ----------------------------------------------------------------------------------------------------------
#define SSMQ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#ifdef BQF_SKIP_CHOL
    if (tid < TPW) sok[tid] = 1;
    for (int gi = wave; gi < 0; gi += WAVES) {
#else
    for (int gi = wave; gi < nb; gi += WAVES) {
#endif
        const int64_t b = bw0 + gi;
        double *A = sA + gi * D * D, *m = sm + gi * DM;
        for (int d = lane; d < D; d += 64) m[d] = a.mean[d * a.es_in + b * a.bs_mean];
        for (int i = lane; i < D * D; i += 64) {
            const int r = i / D, cc = i - r * D;
            A[i] = (cc <= r) ? a.cov[(int64_t)i * a.es_in + b * a.bs_cov] : 0.0;
        }
        if constexpr (!REGCHOL) {
            SSMQ_WAVE_SYNC();
            bool ok = true;
            for (int j = 0; j < D; ++j) {
                const double ajj = A[j * D + j];
                ok = ok && (ajj > 0.0);
                const double ljj = sqrt(ajj), r = 1.0 / ljj;
                SSMQ_WAVE_SYNC();
                if (lane == 0) A[j * D + j] = ljj;
                for (int i = j + 1 + lane; i < D; i += 64) A[i * D + j] *= r;
                SSMQ_WAVE_SYNC();
                const int mm = D - j - 1;
                for (int idx = lane; idx < mm * mm; idx += 64) {
                    const int i = j + 1 + idx / mm, k = j + 1 + idx % mm;
                    if (k <= i) A[i * D + k] -= A[i * D + j] * A[k * D + j];
                }
                SSMQ_WAVE_SYNC();
            }
            for (int q = lane; q < PK; q += 64) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= q) ++i;
                const int j = q - i * (i + 1) / 2;
                sLp[gi * PK + q] = (i < D) ? A[i * D + j] : (i == j ? 1.0 : 0.0);
            }
            if (lane == 0) {
                sok[gi] = ok ? 1 : 0;
                if (a.status) a.status[b] = ok ? 0 : 1;
            }
        }
    }
#undef SSMQ_WAVE_SYNC
    if constexpr (REGCHOL) {
        // D <= 10: the factorisations in registers, lane j of wave 0 = trajectory j (left-looking, the subtraction order of
        // the other kernels' factorisations; identity beyond D)
        __syncthreads();
#ifndef BQF_SKIP_CHOL
        if (wave == 0) {
            const int gi = lane < nb ? lane : 0;
            const double *A = sA + gi * D * D;
            double S[PK];
#pragma unroll
            for (int i = 0; i < DM; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) S[SSMQ_PK(i, j)] = (i < D) ? A[i * D + j] : (i == j ? 1.0 : 0.0);
            const bool ok = chol_packed<DM>(S);
            if (lane < nb) {
#pragma unroll
                for (int q = 0; q < PK; ++q) sLp[gi * PK + q] = S[q];
                sok[gi] = ok ? 1 : 0;
                if (a.status) a.status[bw0 + gi] = ok ? 0 : 1;
            }
        }
#endif
    }
    __syncthreads();
    // ---- 2. integrand values ---------------------------------------------------------------------------------------------------
    {
        const double t0 = (a.time && !a.time_stride) ? a.time[0] : 0.0;
#ifdef BQF_SKIP_POINTS
        for (int idx = tid; idx < 0; idx += TB) {
#else
        for (int idx = tid; idx < nb * N; idx += TB) {
#endif
            int gi, n;
            split(idx, gi, n);
            const double t = (a.time && a.time_stride) ? a.time[bw0 + gi] : t0;
            const double *Lp = sLp + gi * PK, *mp = sm + gi * DM;
            double x[DM], o[DM];
#pragma unroll
            for (int d = 0; d < DM; ++d) {
                double s = d < D ? mp[d] : 0.0;
#pragma unroll
                for (int k = 0; k <= d; ++k) s += Lp[SSMQ_PK(d, k)] * xin[k];          // (rows beyond D: identity x 0)
                x[d] = s;
                o[d] = 0.0;
            }
            {                                      // the next pair's coordinates while this one is evaluated (same registers)
                int g2, n2;
                split(idx + TB, g2, n2);
                load_xi(idx + TB < nb * N ? n2 : N, xin);
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
            const bool ok = sok[gi] != 0;
#pragma unroll
            for (int e = 0; e < DM; ++e)
                if (e < E) sFX[(gi * E + e) * FP + n] = ok ? o[e] : nan;
        }
    }
    // the GEMM's padding columns, and whole rows of a last workgroup with fewer trajectories
    for (int idx = tid; idx < rows * (NP - N); idx += TB) {
        const int r = idx / (NP - N);
        sFX[r * FP + N + (idx - r * (NP - N))] = 0.0;
    }
    for (int idx = tid; idx < (rows - vrows) * N; idx += TB) {
        const int r = idx / N;
        sFX[(vrows + r) * FP + (idx - r * N)] = 0.0;
    }
    park_b(0);                 // (the covariances in the slab region were last read before the barrier that ended step 1)
    load_b(1);
    load_b(2);
    __syncthreads();
    // ---- 3. [T G]' = X' FX'; the G tile's last column is wm: the transformed mean comes out of the same product -----------------
    const int rt = wave % RT, ch = wave / RT;
    const int lrow = (16 * rt + li) < rows ? (16 * rt + li) : rows - 1;   // rows beyond the tile: any row, never stored
    const double *frow = sFX + lrow * FP;
    v4d acc[C0];
#pragma unroll
    for (int t = 0; t < C0; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    // this lane's corner of a slab: row 4 lg + s of the permuted order is k = lg + 4 s; tile t of this half starts 32 t columns on
    const int woff = 4 * lg * LB + 16 * ch + lip, goff = 4 * lg * LB + 16 * NT + lip;
    // NA tiles of S in one k-block: the four k sub-steps, each with its NA reads issued ahead of its NA matrix instructions
    auto mma = [&](auto na_c, const double *sb, const double (&af)[4]) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double w[NA > 0 ? NA : 1];
#pragma unroll
            for (int t = 0; t < NA; ++t) w[t] = sb[woff + s * LB + 32 * t];              // X[16 kb + lg + 4 s][16 (2 t + ch) + pi(li)]
#pragma unroll
            for (int t = 0; t < NA; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[t], af[s], acc[t], 0, 0, 0);
        }
    };
#ifdef BQF_SKIP_MAIN
    for (int p = 0; p < 0; ++p) {
#else
#pragma unroll
    for (int p = 0; p < NKB; ++p) {
#endif
        const int kb = kseq(p);
        if (p + 1 < NKB) {
            park_b(p + 1);
            load_b(p + 3);
        }
        double af[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) af[s] = frow[KS * kb + lg + 4 * s];
        const double *sb = slab + (p & 1) * KS * LB;
        // tiles of S this k-block reaches: ct = 2 t + ch <= kb (the rest of the column is zero)
        const int na = (kb - ch + 2) >> 1;
        if (ch == GCH) {                      // the G tile [Wcc' | wm]: every k-block
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[GT] = __builtin_amdgcn_mfma_f64_16x16x4f64(sb[goff + s * LB], af[s], acc[GT], 0, 0, 0);
        }
        if constexpr (C0 >= 7) { if (na == 7) mma(std::integral_constant<int, 7>{}, sb, af); }
        if constexpr (C0 >= 6) { if (na == 6) mma(std::integral_constant<int, 6>{}, sb, af); }
        if constexpr (C0 >= 5) { if (na == 5) mma(std::integral_constant<int, 5>{}, sb, af); }
        if (na == 4) mma(std::integral_constant<int, 4>{}, sb, af);
        if (na == 3) mma(std::integral_constant<int, 3>{}, sb, af);
        if (na == 2) mma(std::integral_constant<int, 2>{}, sb, af);
        if (na == 1) mma(std::integral_constant<int, 1>{}, sb, af);
        __syncthreads();
    }
#ifdef BQF_SKIP_EPI
    {
        double tsum = 0.0;
#pragma unroll
        for (int ct = 0; ct < C0; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) tsum += acc[ct][r];
        if (tsum == 12345.678) smr[lane] = tsum;
        return;
    }
#endif
    // ---- 4a. the waves that hold the G tile (global tile NT = local tile GT of half GCH): mean, cross-covariance ---------------
    double *sP = slab;                             // [2][RT][64][8]: the epilogue's exchange (the slabs are free: last barrier passed)
    if (ch == GCH) {
        const int lr = 16 * rt + li;
        const bool valid = lr < vrows;
        const int gi = srow[lr] >> 8, e = srow[lr] & 255;
        const int64_t b = bw0 + gi;
        const double *Lb = sLp + (valid ? gi : 0) * PK;
        const v4d gt = acc[GT];
        if (lg == 3) {                               // column 15 of the tile: FX wm (NaN where the factorisation failed)
            smr[lr] = gt[3];
            if (valid) a.mean_f[(int64_t)e * a.es_out + b * a.bs_mf] = gt[3];
        }
        // (fx Wcc') L' for row li: the G tile goes through LDS (the slab region is free: the loop's last barrier has been
        // passed) and lane group lg takes the columns j = lg, lg + 4, ... - the sum over d with two cross-lane exchanges per j
        // instead was 3 us per tile of dependent ds_bpermute round trips
        double *sG = sP + ((GCH * RT + rt) * 64) * 8;     // this wave's own exchange slot, before it is written below
#pragma unroll
        for (int r = 0; r < 4; ++r) sG[li * 16 + 4 * lg + r] = gt[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j = lg; j < D; j += 4) {
            double p = 0.0;
            for (int d = 0; d <= j; ++d) p += sG[li * 16 + d] * Lb[SSMQ_PK(j, d)];
            if (valid) a.cov_fx[(int64_t)(e * D + j) * a.es_out + b * a.bs_cfx] = p * a.ccov_scale;
        }
    }
    // ---- 4b. C = T FX2' with T = fx S, this wave's column tiles -------------------------------------------------------------------
    const int s0 = (16 * rt / E) * E;             // first row of the first trajectory that intersects the row tile
    v4d acc2[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
    {
        const int r0 = (s0 + li) < rows ? (s0 + li) : rows - 1, r1 = (s0 + 16 + li) < rows ? (s0 + 16 + li) : rows - 1;
        const double *f0p = sFX + r0 * FP + 4 * lg, *f1p = sFX + r1 * FP + 4 * lg;
#pragma unroll
        for (int t = 0; t < C0; ++t) {
            const int ct = 2 * t + ch;
#ifdef BQF_SKIP_4B
            if (0) {
#else
            if (ct < NT) {
#endif
                const int col = 16 * ct;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[t][r], f0p[col + r], acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[t][r], f1p[col + r], acc2[1], 0, 0, 0);
                }
            }
        }
    }
    // Every wave's part of C goes to LDS (lane (lg, li) of wave (rt, ch): C[16 rt + lg + 4 r][s0 + 16 h + li] in slot 4 h + r;
    // the slabs are free, the loop's last barrier has been passed), the additive terms and the (e, e2) table beside it
    double *sev = sP + 2 * RT * 64 * 8, *sca = sev + 256;
    int *spair = (int *)(sca + 256);
    const int npair = E * (E + 1) / 2;
    if (tid < E * E) {
        sev[tid] = ev;
        sca[tid] = ca;
    }
    if (tid < npair) {
        int e = 0;
        while ((e + 1) * (e + 2) / 2 <= tid) ++e;
        spair[tid] = (e << 4) | (tid - e * (e + 1) / 2);
    }
    {
        double *sx = sP + ((ch * RT + rt) * 64 + lane) * 8;
        __builtin_amdgcn_wave_barrier();          // (the G waves read their slot as sG above)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) sx[4 * h + r] = acc2[h][r];
    }
    __syncthreads();
#ifdef BQF_SKIP_4C
    if (acc2[0][0] != 12345.678) return;
#endif
    // ---- 4c. covariance entries: fx Wc fx' = C + C'; one (trajectory, e >= e2) pair per thread, trajectory fastest --------------
    for (int idx = tid; idx < nb * npair; idx += TB) {
        const int p = idx / nb, gi = idx - p * nb;
        const int e = spair[p] >> 4, e2 = spair[p] & 15;
        const int l1 = gi * E + e, l2 = gi * E + e2;
        auto cval = [&](int la, int lb) {          // C[la][lb]: both halves of the column range, half 0 first
            const int rta = la >> 4, i = la & 15, j = lb - (16 * rta / E) * E;
            const int at = ((rta * 64) + (j & 15) + 16 * (i & 3)) * 8 + 4 * (j >> 4) + (i >> 2);
            return sP[at] + sP[RT * 64 * 8 + at];
        };
        const int64_t b = bw0 + gi;
        const int ie = e * E + e2, it = e2 * E + e;
        const bool use = (e == e2) || g.emv_broadcast;
        const double em = use ? sev[ie] : 0.0;
        double v = (cval(l1, l2) + cval(l2, l1) - smr[l1] * smr[l2] + em) * a.cov_scale;
        if (a.cov_add) v += sca[ie];
        a.cov_f[(int64_t)ie * a.es_out + b * a.bs_cf] = v;
        if (e2 != e) a.cov_f[(int64_t)it * a.es_out + b * a.bs_cf] = v;     // mirrored entry, same value
    }
}

struct FusedGeom {
    int waves, tpw, fx_doubles;
    size_t lds;
};
FusedGeom fused_geom_for(int waves, int NT, int D, int E, int DM) {
    FusedGeom q;
    q.waves = waves;
    const int NP = NT * 16, LB = NP + 16 + 4, FP = NP + 2, RT = waves / 2, KS = 2 * waves;
    const size_t cap = waves == 4 ? 80 * 1024 : 160 * 1024;
    // as many whole trajectories as fit the tile's 16 RT rows AND the LDS share beside the two slabs
    for (q.tpw = 16 * RT / E; q.tpw >= 1; --q.tpw) {
        q.fx_doubles = q.tpw * E * FP;
        q.lds = sizeof(double) * ((size_t)q.fx_doubles + 2 * KS * LB + 16 * RT + (size_t)q.tpw * (DM * (DM + 1) / 2 + DM)) + sizeof(int) * (16 + 16 * RT);
        if (q.lds <= cap) break;
    }
    return q;
}
// at least 3/4 of the tile's rows in use; the covariances of step 1 (slab region) and the epilogue's exchange must fit the slabs
bool fused_geom_ok(const FusedGeom &q, int NT, int D, int E) {
    const int RT = q.waves / 2, KS = 2 * q.waves;
    const size_t slab = (size_t)2 * KS * (NT * 16 + 20);
    return q.tpw >= 1 && 4 * q.tpw * E >= 3 * 16 * RT && (size_t)q.tpw * D * D <= slab && (size_t)2 * RT * 64 * 8 + 512 + 32 <= slab;
}
// one 512-thread workgroup per CU.  (Two of 256 threads - 32-row tiles, 8-row slabs, 80 KB each - were measured at D = E =
// 10, N = 201, B = 1e4: 339 us against 322 us; the two workgroups of a CU run in phase, so their producer and store steps
// coincide instead of hiding under each other's matrix instructions, and X crosses L2 -> LDS twice as often.  The kernel
// keeps the wave count as a template parameter; only WAVES = 8 is instantiated.)
FusedGeom fused_geom(int NT, int D, int E, int DM) { return fused_geom_for(8, NT, D, E, DM); }
// the compile-time bound on D and E a shape runs with (launch_fused_nt's table)
int fused_dm(const WideArgs *a, int D, int E) {
    const int dm = D > E ? D : E;
    if (a && a->fid == SSMQ_F_SMOOTH10D_DYN && a->fp.n_idx == 0 && dm <= 10) return 10;
    return dm <= 8 ? 8 : SSMQ_MAX_DIM;
}

template <int NT, int DM, int FC, int WAVES>
hipError_t launch_fused_one(const BqFusedArgs &g, size_t lds, hipStream_t s) {
    static thread_local unsigned attr_epoch = 0;   // per-device attribute: set again after a device change
    if (attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_bq_fused<NT, DM, FC, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           WAVES == 4 ? 80 * 1024 : 160 * 1024);
        if (e != hipSuccess) return e;
        attr_epoch = device_epoch();
    }
    hipLaunchKernelGGL((k_bq_fused<NT, DM, FC, WAVES>), dim3((unsigned)((g.B + g.tpw - 1) / g.tpw)), dim3(64 * WAVES), lds, s, g);
    return hipGetLastError();
}

template <int NT, int WAVES>
hipError_t launch_fused_nt(const BqFusedArgs &g, size_t lds, hipStream_t s) {
    const int dm = fused_dm(&g.w, g.w.D, g.w.E);
    if (dm == 10) return launch_fused_one<NT, 10, SSMQ_F_SMOOTH10D_DYN, WAVES>(g, lds, s);
    if (dm == 8) return launch_fused_one<NT, 8, -1, WAVES>(g, lds, s);
    return launch_fused_one<NT, SSMQ_MAX_DIM, -1, WAVES>(g, lds, s);
}

}  // namespace

// BQ transform (not the t-process one) with one constant block for the batch, 64 < N <= 208 points (the 128- and
// 208-column instantiations of the matrix-core route), whole trajectories filling at least 48 of a tile's 64 rows
bool bq_fused_supported(int D, int E, int N) {
    if (ssmq::sw("SSMQ_NO_BQ_FUSED")) return false;
    const int np = gemm_mfma_padded(N);
    if (np != 128 && np != 208) return false;
    // D <= 15: column 15 of the Wcc' tile of X carries wm (ssmq_api.hip: upload of d_wcx_pad)
    // E <= 10: no built-in integrand has more outputs (10-D model: 10; bearings: SSMQ_MAX_FPAR / 2 = 8 sensors), so nothing
    // beyond that could be tested
    if (D < 1 || D > 15 || E < 6 || E > 10 || !fxwc_cov_supported(E)) return false;
    return fused_geom_ok(fused_geom(np / 16, D, E, fused_dm(nullptr, D, E)), np / 16, D, E);   // (the generic bound: the larger footprint)
}

// a: WideArgs of the whole transform as for launch_apply_wide (mode FULL, form BQ, consts_stride 0); X = [Wc | Wcc'] padded
// with wm in its last column (ssmq_transform::d_wcx_pad); emv: the E x E model-variance block
int launch_bq_fused(const WideArgs &a, const double *X, const double *emv, int emv_broadcast, int64_t B, hipStream_t s) {
    if (B <= 0) return SSMQ_OK;
    const int np = gemm_mfma_padded(a.N);
    if (!bq_fused_supported(a.D, a.E, a.N) || a.consts_stride != 0 || a.form != SSMQ_FORM_BQ || a.tp_nu > 0.0) {
        set_error("bq_fused: shape not supported");
        return SSMQ_E_UNSUPPORTED;
    }
    FusedGeom q = fused_geom(np / 16, a.D, a.E, fused_dm(&a, a.D, a.E));
    if (!fused_geom_ok(q, np / 16, a.D, a.E)) q = fused_geom(np / 16, a.D, a.E, fused_dm(nullptr, a.D, a.E));
    BqFusedArgs g;
    g.w = a; g.X = X; g.emv = emv; g.emv_broadcast = emv_broadcast; g.tpw = q.tpw; g.fx_doubles = q.fx_doubles;
    g.B = B;
    const hipError_t e = np == 208 ? launch_fused_nt<13, 8>(g, q.lds, s) : launch_fused_nt<8, 8>(g, q.lds, s);
    return hip_fail(e, "k_bq_fused");
}

}  // namespace ssmq
