// Whole BQ moment transform for LARGE point sets (209 ... 4096 points) in ONE launch: k_bq_fused's scheme with the point axis
// tiled.  (bq/bqmtran.py:132-156: x_n = m + L xi_n, f(x_n); :158-223: mean = fx wm, cov = fx Wc fx' - mean mean' + emv,
// ccov = fx Wcc' L'.)
//
// Round 3's route for these sizes was three launches - k_eval_wave (FX to HBM), k_fxwc_mfma<16,1> (T = FX [Wc | Wcc'] by column
// blocks, T to HBM), k_big_rest (per trajectory: T FX', the rest) - with 9.1 GB of HBM traffic per 1e4 transforms at D = E = 10,
// N = 1181 for 25.6 MB of algorithmic bytes, and the full product fx Wc.  Here:
//   * a persistent 512-thread workgroup per CU walks tiles of TPW whole trajectories (TPW E <= 64 rows of FX);
//   * factor + integrand values as in k_bq_fused, but in chunks of 208 points through the LDS tile, each chunk copied to the
//     workgroup's OWN scratch block in global memory in FRAGMENT ORDER ([k-block][row tile][lane][4]: the 32 bytes a lane
//     feeds to the four matrix instructions of a k-block are contiguous) - 606 KB per workgroup at N = 1181, 155 MB for the
//     256 workgroups of the device however large the batch: it lives in the Infinity Cache / L2, not in HBM;
//   * Wc = S + S' (S: lower triangle, half the diagonal), so fx Wc fx' = C + C' with C = (fx S) fx': the column tiles of S
//     are processed in PANELS of 13 (208 columns, the accumulators of k_bq_fused: 7 tiles per wave, two waves per SIMD);
//     panel p needs the k-blocks kb >= 13 p only, and of its first 13 k-blocks only the tiles on or below the diagonal.
//     T never exists: when a panel's k loop ends, its accumulators are multiplied with the panel's FX columns (C += T_p FX_p')
//     and cleared.  Panel 0 carries the G tile [Wcc' | wm]: cross-covariance and mean are by-products;
//   * X slabs (16 k-rows x 224 columns, L2-resident: the workgroups of an XCD walk the same sequence) double-buffered in LDS and
//     requested two steps ahead, FX fragments requested two steps ahead straight into registers;
//   * epilogue as k_bq_fused: all parts of C meet in LDS, one thread per (trajectory, e >= e2) forms C + C' and stores.
// Matrix work per 64-row tile at N = 1181 (74 column tiles): 74 75 / 2 + 74 = 2 849 tile steps x 4 instructions per row tile against
// 74 90 = 6 660 for the full [Wc | Wcc'] product.
#include "ssmq_host.h"
#include "ssmq_wide.h"
#include <type_traits>

namespace ssmq {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kPanT = 13;                       // column tiles of S per panel
constexpr int kPanW = 16 * kPanT;               // 208 columns = points per chunk of the evaluation
constexpr int kPanX = kPanW + 16;               // panel row in memory: 208 columns of S + the G tile (panel 0 only)

struct BqStreamArgs {
    WideArgs w;             // shape, integrand, constants (points, wm), inputs, outputs, scales - as for k_eval_wave
    const double *X;        // [npan][16 nkb][224]: panel p = columns 208 p .. of S, then (p = 0) [Wcc' | wm]; zero-padded
    const double *emv;      // [E * E]
    double *scratch;        // [gridDim.x][nkb][4][64][4]: this workgroup's FX in fragment order
    int32_t emv_broadcast, tpw, fx_doubles, nkb, npan, ntot;   // k-blocks of 16 points, panels, column tiles of S (= nkb)
    int64_t B, tiles;
};

struct StepIt {             // (panel, k-block) of one step of the flattened main loop; kb runs DOWN within a panel
    int p, kb;
};

template <int DM, int FC>
__global__ __launch_bounds__(512, 1) void k_bq_stream(const BqStreamArgs g) {
    constexpr int WAVES = 8, TB = 512, RT = 4, KS = 16;
    constexpr int NT = kPanT, NX = kPanX, LB = NX + 4, FP = kPanW + 2;
    constexpr int C0 = (NT + 2) / 2;               // 7 accumulator tiles per wave: S tiles 2 t + ch of the panel; G at GT of half GCH
    constexpr int GCH = NT & 1, GT = NT >> 1;
    constexpr int PK = DM * (DM + 1) / 2;
    constexpr bool REGCHOL = DM <= 10;
    extern __shared__ __align__(16) double lds[];
    const WideArgs &a = g.w;
    const int D = a.D, E = a.E, N = a.N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int lip = 4 * (li & 3) + (li >> 2);
    const int rt = wave % RT, ch = wave / RT;
    const int TPW = g.tpw, rows = TPW * E;
    const int nkb = g.nkb, npan = g.npan;
    double *sFX = lds;                              // [rows][FP]: one chunk of integrand values; the epilogues' exchange later
    double *slab = lds + g.fx_doubles;              // [2][KS][LB]
    double *smr = slab + 2 * KS * LB;               // [64] transformed means by row
    double *sLp = smr + 64;                         // [TPW][PK] packed factors
    double *sm = sLp + TPW * PK;                    // [TPW][DM] input means
    int *sok = (int *)(sm + TPW * DM);              // [TPW]
    int *srow = sok + 16;                           // [64] row -> (trajectory << 8) | output index
    double *sA = slab;                              // step 1 only: covariances in
    double *sG = sFX;                               // [RT][256]  panel 0's G tiles            } the tile is free during
    double *sP = sFX + RT * 256;                    // [2][RT][64][8] parts of C               } the main loop
    double *sev = sP + 2 * RT * 64 * 8, *sca = sev + 256;
    int *spair = (int *)(sca + 256);
    double *myfx = g.scratch + (size_t)blockIdx.x * nkb * (RT * 64 * 4);
    const double nan = __builtin_nan("");
    const double *c = a.consts;
    const WideLayout cl = wide_layout(D, E, N, a.form);
    auto next_it = [&](StepIt it) {
        if (it.kb > NT * it.p) return StepIt{it.p, it.kb - 1};
        return StepIt{it.p + 1, nkb - 1};
    };
    const float rW = 1.0f / (float)kPanW;
    const int npair = E * (E + 1) / 2;

    for (int64_t tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
        const int64_t bw0 = tile * TPW;
        const int nb = (int)((g.B - bw0) < (int64_t)TPW ? (g.B - bw0) : (int64_t)TPW);
        const int vrows = nb * E;
        // ---- 0. requests that do not depend on the factors ---------------------------------------------------------------------
        if (tid < 64) {
            const int gq = tid / E;
            srow[tid] = (gq << 8) | (tid - gq * E);
        }
        StepIt c0{0, nkb - 1};
        StepIt c1 = next_it(c0), c2 = next_it(c1);
        // ---- 1. factors (as k_bq_fused) ------------------------------------------------------------------------------------------
#define SSMQ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
        for (int gi = wave; gi < nb; gi += WAVES) {
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
            __syncthreads();
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
        }
        __syncthreads();
        // ---- 2. integrand values, 208 points at a time: LDS tile -> this workgroup's scratch block in fragment order ---------------
        const double t0 = (a.time && !a.time_stride) ? a.time[0] : 0.0;
        // (the thread index through an opaque statement per tile: otherwise every per-thread address of this phase is hoisted out of
        // the tile loop and stays in registers across the main loop - spills in its steps, each one a drained request queue)
        int ptid = tid;
        asm volatile("" : "+v"(ptid));
        for (int ck = 0; ck < npan; ++ck) {
            const int n0 = ck * kPanW;
            const int cn = (N - n0) < kPanW ? (N - n0) : kPanW;      // points of this chunk (> 0: npan = ceil(N / 208))
            const float rC = cn == kPanW ? rW : 1.0f / (float)cn;
            auto split = [&](int idx, int &gi, int &n) {
                gi = (int)(((float)idx + 0.5f) * rC);
                n = idx - gi * cn;
            };
            double xin[DM];
            auto load_xi = [&](int n, double (&dst)[DM]) {
#pragma unroll
                for (int k = 0; k < DM; ++k) dst[k] = (k < D && n < N) ? c[cl.xiT + n * D + k] : 0.0;
            };
            {
                int gi0, nl0;
                split(ptid, gi0, nl0);
                load_xi(ptid < nb * cn ? n0 + nl0 : N, xin);
            }
            for (int idx = ptid; idx < nb * cn; idx += TB) {
                int gi, nl;
                split(idx, gi, nl);
                const double t = (a.time && a.time_stride) ? a.time[bw0 + gi] : t0;
                const double *Lp = sLp + gi * PK, *mp = sm + gi * DM;
                double x[DM], o[DM];
#pragma unroll
                for (int d = 0; d < DM; ++d) {
                    double s = d < D ? mp[d] : 0.0;
#pragma unroll
                    for (int k = 0; k <= d; ++k) s += Lp[SSMQ_PK(d, k)] * xin[k];
                    x[d] = s;
                    o[d] = 0.0;
                }
                {
                    int g2, n2;
                    split(idx + TB, g2, n2);
                    load_xi(idx + TB < nb * cn ? n0 + n2 : N, xin);
                }
                double xs[kMaxIntegrandIn];
#pragma unroll
                for (int k = 0; k < kMaxIntegrandIn; ++k) {
                    double v = k < DM ? x[k < DM ? k : 0] : 0.0;
                    if (FC < 0 && a.fp.n_idx > 0) {
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
                    if (e < E) sFX[(gi * E + e) * FP + nl] = ok ? o[e] : nan;
            }
            // padding columns of the last chunk, and whole rows of a tile with fewer trajectories
            for (int idx = ptid; idx < rows * (kPanW - cn); idx += TB) {
                const int r = idx / (kPanW - cn);
                sFX[r * FP + cn + (idx - r * (kPanW - cn))] = 0.0;
            }
            for (int idx = ptid; idx < (rows - vrows) * cn; idx += TB) {
                const int r = idx / cn;
                sFX[(vrows + r) * FP + (idx - r * cn)] = 0.0;
            }
            __syncthreads();
            // copy out: block (k-block kbl of the chunk, row tile q) -> [kb][q][lane][4]; rows beyond the tile repeat its last row
            for (int blk = wave; blk < NT * RT; blk += WAVES) {
                const int kbl = blk >> 2, q = blk & 3, kb = NT * ck + kbl;
                if (kb < nkb) {
                    const int lrow = (16 * q + li) < rows ? (16 * q + li) : rows - 1;
                    const double *fr = sFX + lrow * FP + KS * kbl + lg;
                    v4d v;
#pragma unroll
                    for (int s = 0; s < 4; ++s) v[s] = fr[4 * s];
                    *(v4d *)(myfx + (((size_t)kb * RT + q) * 64 + lane) * 4) = v;
                }
            }
            __syncthreads();
        }
        {
        // (everything per-thread the main loop uses is derived here from an opaque copy of the thread index: computed before the tile
        // loop it would stay in registers across the evaluation phase, be spilled there and reloaded inside the loop's steps)
        int mtid = tid;
        asm volatile("" : "+v"(mtid));
        const int lane = mtid & 63, li = lane & 15, lg = lane >> 4, lip = 4 * (li & 3) + (li >> 2);
        const int sr = mtid >> 5, shf = (mtid >> 4) & 1, sc16 = mtid & 15;
        double *myfx = g.scratch + (size_t)blockIdx.x * nkb * (RT * 64 * 4);
        auto phys = [](int k) { return 4 * (k & 3) + (k >> 2); };
        // X slabs (L2-resident) are requested ONE step ahead through one register set; the FX fragments (this workgroup's scratch
        // block: Infinity Cache / HBM latency, ~2.5 us) TWO steps ahead through two.  Every request of a step is UNCONDITIONAL (an
        // invalid step repeats the last valid addresses; slab columns a diagonal step does not need are read all the same) and the
        // steps run in straight-line groups of kGroup: the compiler's s_waitcnt placement counts requests, any request under a
        // condition - or a loop header - makes it wait for ALL outstanding ones, i.e. every step would wait for the requests of the
        // step before it in full (measured that way: 2.6 us per step, 4.6 ms per 1e4 transforms at N = 1181).
        constexpr int kGroup = 4;
        const StepIt last_it{npan - 1, NT * (npan - 1)};
        auto valid_it = [&](StepIt it) { return it.p < npan ? it : last_it; };
        double breg[C0];
        v4d afs[2];
        auto load_b = [&](StepIt it0) {
#ifdef BQS_SAME_SLAB
            const StepIt it{0, 0};
#else
            const StepIt it = valid_it(it0);
#endif
            const double *src = g.X + ((size_t)it.p * nkb * KS + (size_t)it.kb * KS + sr) * NX + 16 * shf + sc16;
#pragma unroll
            for (int j = 0; j < C0; ++j) breg[j] = src[32 * j];
        };
        auto park_b = [&](int buf) {
            double *dst = slab + buf * KS * LB + phys(sr) * LB + 16 * shf + sc16;
#pragma unroll
            for (int j = 0; j < C0; ++j) dst[32 * j] = breg[j];
        };
        auto load_af = [&](auto set_c, StepIt it0) {
            constexpr int S_ = decltype(set_c)::value;
#ifdef BQS_SAME_AF
            const StepIt it{0, 0};
#else
            const StepIt it = valid_it(it0);
#endif
            afs[S_] = *(const v4d *)(myfx + (((size_t)it.kb * RT + rt) * 64 + lane) * 4);
        };
        // ---- 3. main loop over (panel, k-block): [T_p G]' = X_p' FX', C += T_p FX_p' at the end of each panel ------------------------
        load_b(c0);                // (a tile is ~350 us of matrix work: nothing is requested across the phases)
        park_b(0);
        load_af(std::integral_constant<int, 0>{}, c0);
        load_b(c1);
        load_af(std::integral_constant<int, 1>{}, c1);
        __syncthreads();
        v4d acc[C0];
#pragma unroll
        for (int t = 0; t < C0; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
        // this wave's part of C lives in its exchange slot (LDS) between the panels: 16 registers less across the main loop
        double *sx = sP + ((ch * RT + rt) * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) sx[i] = 0.0;
        const int woff = 4 * lg * LB + 16 * ch + lip, goff = 4 * lg * LB + 16 * NT + lip;
        const int s0 = (16 * rt / E) * E;             // first row of the first trajectory that intersects this wave's row tile
        auto mma = [&](auto na_c, const double *sb, const double (&af)[4]) {
            constexpr int NA = decltype(na_c)::value;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double w[NA];
#pragma unroll
                for (int t = 0; t < NA; ++t) w[t] = sb[woff + s * LB + 32 * t];
#pragma unroll
                for (int t = 0; t < NA; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[t], af[s], acc[t], 0, 0, 0);
            }
        };
        // element (row, k) of FX in the scratch block: [k >> 4][row >> 4][(row & 15) + 16 (k & 3)][(k & 15) >> 2]
        auto fx_at = [&](int row, int k) {
            return myfx[(((size_t)(k >> 4) * RT + (row >> 4)) * 64 + (row & 15) + 16 * (k & 3)) * 4 + ((k & 15) >> 2)];
        };
        auto panel_end = [&](int p) {
            const int ntp = g.ntot - NT * p;
            if (p == 0 && ch == GCH) {
                // mean and cross-covariance from the G tile (as k_bq_fused 4a)
                const int lr = 16 * rt + li;
                const bool valid = lr < vrows;
                const int gi = srow[lr] >> 8, e = srow[lr] & 255;
                const int64_t b = bw0 + gi;
                const double *Lb = sLp + (valid ? gi : 0) * PK;
                const v4d gt = acc[GT];
                if (lg == 3) {
                    smr[lr] = gt[3];
                    if (valid) a.mean_f[(int64_t)e * a.es_out + b * a.bs_mf] = gt[3];
                }
                double *sg = sG + rt * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) sg[li * 16 + 4 * lg + r] = gt[r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int j = lg; j < D; j += 4) {
                    double pq = 0.0;
                    for (int d = 0; d <= j; ++d) pq += sg[li * 16 + d] * Lb[SSMQ_PK(j, d)];
                    if (valid) a.cov_fx[(int64_t)(e * D + j) * a.es_out + b * a.bs_cfx] = pq * a.ccov_scale;
                }
            }
            // C += T_p FX_p': this wave's tiles of the panel against the rows of the trajectories that meet its row tile.  Straight-
            // line code: the eight FX fragments of tile t + 1 are requested before the eight matrix instructions of tile t are issued
            // (a request per instruction, each waited for, was 56 dependent round trips per panel: 350 us per tile).  Tiles this
            // panel does not have (the last panel; tile GT of the G half) multiply accumulators that are zero / are skipped.
            const int r0 = (s0 + li) < rows ? (s0 + li) : rows - 1, r1 = (s0 + 16 + li) < rows ? (s0 + 16 + li) : rows - 1;
            const int nmine = ((ntp < NT ? ntp : NT) - ch + 1) >> 1;         // S tiles of this wave in this panel
            v4d acc2[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
            double f[2][8];
            auto ldf = [&](int t, double (&dst)[8]) {
                const int tc = t < nmine ? t : 0;
                const int col = 16 * (NT * p + 2 * tc + ch) + 4 * lg;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dst[r] = fx_at(r0, col + r);
                    dst[4 + r] = fx_at(r1, col + r);
                }
            };
            ldf(0, f[0]);
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                ldf(t + 1, f[(t + 1) & 1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[t][r], f[t & 1][r], acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[t][r], f[t & 1][4 + r], acc2[1], 0, 0, 0);
                }
            }
            if (ch != GCH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[GT][r], f[GT & 1][r], acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[GT][r], f[GT & 1][4 + r], acc2[1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) sx[4 * h + r] += acc2[h][r];
#pragma unroll
            for (int t = 0; t < C0; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
        };
        auto step = [&](auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            park_b(PAR ^ 1);                                   // the slab of step q + 1 (requested a step ago)
            load_b(c2);
            const double af[4] = {afs[PAR][0], afs[PAR][1], afs[PAR][2], afs[PAR][3]};
            load_af(std::integral_constant<int, PAR>{}, c2);   // the fragments of step q + 2
            const double *sb = slab + PAR * KS * LB;
            const int kbl = c0.kb - NT * c0.p, ntp = g.ntot - NT * c0.p;
            int na = (kbl - ch + 2) >> 1;
            const int cap = ((ntp < NT ? ntp : NT) - ch + 1) >> 1;
            na = na < cap ? na : cap;
            if (c0.p >= npan) na = 0;                       // padding step of the last group
            if (c0.p == 0 && ch == GCH) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[GT] = __builtin_amdgcn_mfma_f64_16x16x4f64(sb[goff + s * LB], af[s], acc[GT], 0, 0, 0);
            }
            if (na == 7) mma(std::integral_constant<int, 7>{}, sb, af);
            if (na == 6) mma(std::integral_constant<int, 6>{}, sb, af);
            if (na == 5) mma(std::integral_constant<int, 5>{}, sb, af);
            if (na == 4) mma(std::integral_constant<int, 4>{}, sb, af);
            if (na == 3) mma(std::integral_constant<int, 3>{}, sb, af);
            if (na == 2) mma(std::integral_constant<int, 2>{}, sb, af);
            if (na == 1) mma(std::integral_constant<int, 1>{}, sb, af);
            __syncthreads();
            if (c0.p < npan && c0.kb == NT * c0.p) panel_end(c0.p);
            c0 = c1; c1 = c2; c2 = next_it(c2);
        };
        static_assert(kGroup % 2 == 0, "register sets and slab buffers alternate with the step");
        while (c0.p < npan) {
#pragma unroll
            for (int u = 0; u < kGroup; u += 2) {
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
            }
        }
        }
        // ---- 4. all parts of C meet in LDS; fx Wc fx' = C + C' -----------------------------------------------------------------------
        if (tid < E * E) {
            sev[tid] = g.emv[tid];
            sca[tid] = a.cov_add ? a.cov_add[tid] : 0.0;
        }
        if (tid < npair) {
            int e = 0;
            while ((e + 1) * (e + 2) / 2 <= tid) ++e;
            spair[tid] = (e << 4) | (tid - e * (e + 1) / 2);
        }
        __syncthreads();
        for (int idx = tid; idx < nb * npair; idx += TB) {
            const int p = idx / nb, gi = idx - p * nb;
            const int e = spair[p] >> 4, e2 = spair[p] & 15;
            const int l1 = gi * E + e, l2 = gi * E + e2;
            auto cval = [&](int la, int lb) {
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
            if (e2 != e) a.cov_f[(int64_t)it * a.es_out + b * a.bs_cf] = v;
        }
        __syncthreads();          // the next tile reuses every LDS region
    }
}

struct StreamGeom {
    int tpw, fx_doubles;
    size_t lds;
};
StreamGeom stream_geom(int D, int E, int DM) {
    StreamGeom q;
    const int LB = kPanX + 4, FP = kPanW + 2;
    for (q.tpw = 64 / E; q.tpw >= 1; --q.tpw) {
        q.fx_doubles = q.tpw * E * FP;
        q.lds = sizeof(double) * ((size_t)q.fx_doubles + 2 * 16 * LB + 64 + (size_t)q.tpw * (DM * (DM + 1) / 2 + DM)) + sizeof(int) * (16 + 64);
        if (q.lds <= 160 * 1024) break;
    }
    return q;
}
bool stream_geom_ok(const StreamGeom &q, int D, int E) {
    const size_t slab = (size_t)2 * 16 * (kPanX + 4);
    // 3/4 of the rows in use; the covariances of step 1 fit the slab region; the epilogue's exchange fits the tile
    return q.tpw >= 1 && 4 * q.tpw * E >= 3 * 64 && (size_t)q.tpw * D * D <= slab &&
           (size_t)4 * 256 + 2 * 4 * 64 * 8 + 512 + 32 <= (size_t)q.fx_doubles;
}
int stream_dm(const WideArgs *a, int D, int E) {
    const int dm = D > E ? D : E;
    if (a && a->fid == SSMQ_F_SMOOTH10D_DYN && a->fp.n_idx == 0 && dm <= 10) return 10;
    return dm <= 8 ? 8 : SSMQ_MAX_DIM;
}

template <int DM, int FC>
hipError_t launch_stream_one(const BqStreamArgs &g, size_t lds, int grid, hipStream_t s) {
    static unsigned attr_epoch = 0;
    if (attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_bq_stream<DM, FC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_epoch = device_epoch();
    }
    hipLaunchKernelGGL((k_bq_stream<DM, FC>), dim3((unsigned)grid), dim3(512), lds, s, g);
    return hipGetLastError();
}

}  // namespace

int bq_stream_panels(int N) { return (N + kPanW - 1) / kPanW; }
int bq_stream_kblocks(int N) { return (N + 15) / 16; }
size_t bq_stream_x_doubles(int N) { return (size_t)bq_stream_panels(N) * bq_stream_kblocks(N) * 16 * kPanX; }
size_t bq_stream_scratch_doubles(int N, int grid) { return (size_t)grid * bq_stream_kblocks(N) * 4 * 64 * 4; }

// BQ transform (not the t-process one), one constant block for the batch, 208 < N <= SSMQ_MAX_PTS, a symmetric Wc
bool bq_stream_supported(int D, int E, int N) {
    if (getenv("SSMQ_NO_BQ_STREAM") || getenv("SSMQ_NO_MFMA")) return false;
    if (N <= kPanW || N > SSMQ_MAX_PTS) return false;
    if (D < 1 || D > 15 || E < 6 || E > 10) return false;
    return stream_geom_ok(stream_geom(D, E, stream_dm(nullptr, D, E)), D, E);
}

// X in the panel layout of BqStreamArgs from the natural-layout weights (host): S = tril(Wc) with half the diagonal
void bq_stream_pack(int D, int N, const double *Wc, const double *Wcc, const double *wm, double *X) {
    const int npan = bq_stream_panels(N), nkb = bq_stream_kblocks(N);
    const size_t per = (size_t)nkb * 16 * kPanX;
    for (size_t i = 0; i < per * npan; ++i) X[i] = 0.0;
    for (int k = 0; k < N; ++k) {
        for (int j = 0; j <= k; ++j) {
            const int p = j / kPanW;
            X[(size_t)p * per + (size_t)k * kPanX + (j - p * kPanW)] = j == k ? 0.5 * Wc[(size_t)k * N + k] : Wc[(size_t)k * N + j];
        }
        for (int d = 0; d < D && d < 16; ++d) X[(size_t)k * kPanX + kPanW + d] = Wcc[(size_t)d * N + k];
        if (D <= 15) X[(size_t)k * kPanX + kPanW + 15] = wm[k];
    }
}

int launch_bq_stream(const WideArgs &a, const double *X, const double *emv, int emv_broadcast, int64_t B, double *scratch,
                     int grid, hipStream_t s) {
    if (B <= 0) return SSMQ_OK;
    if (!bq_stream_supported(a.D, a.E, a.N) || a.consts_stride != 0 || a.form != SSMQ_FORM_BQ || a.tp_nu > 0.0 || grid < 1) {
        set_error("bq_stream: shape not supported");
        return SSMQ_E_UNSUPPORTED;
    }
    int dm = stream_dm(&a, a.D, a.E);
    StreamGeom q = stream_geom(a.D, a.E, dm);
    if (!stream_geom_ok(q, a.D, a.E)) {
        dm = stream_dm(nullptr, a.D, a.E);
        q = stream_geom(a.D, a.E, dm);
    }
    BqStreamArgs g;
    g.w = a; g.X = X; g.emv = emv; g.scratch = scratch; g.emv_broadcast = emv_broadcast; g.tpw = q.tpw; g.fx_doubles = q.fx_doubles;
    g.nkb = bq_stream_kblocks(a.N); g.npan = bq_stream_panels(a.N); g.ntot = g.nkb;
    g.B = B; g.tiles = (B + q.tpw - 1) / q.tpw;
    if ((int64_t)grid > g.tiles) grid = (int)g.tiles;
    hipError_t e;
    if (dm == 10 && a.fid == SSMQ_F_SMOOTH10D_DYN && a.fp.n_idx == 0) e = launch_stream_one<10, SSMQ_F_SMOOTH10D_DYN>(g, q.lds, grid, s);
    else if (dm == 8) e = launch_stream_one<8, -1>(g, q.lds, grid, s);
    else e = launch_stream_one<SSMQ_MAX_DIM, -1>(g, q.lds, grid, s);
    return hip_fail(e, "k_bq_stream");
}

}  // namespace ssmq
