// BQ moment transform for LARGE point sets (209 ... 4096 points) in TWO launches: k_eval_wave (factor, sigma points, integrand
// values FX to memory, row-major) and this kernel - k_bq_fused's product with the point axis tiled, nothing else through
// memory.  (bq/bqmtran.py:158-223: mean = fx wm, cov = fx Wc fx' - mean mean' + emv, ccov = fx Wcc' L'.)
//
// Round 3's route for these sizes was three launches - k_eval_wave, k_fxwc_mfma<16,1> (T = FX [Wc | Wcc'] by column blocks,
// T to HBM: 0.95 GB written, read again), k_big_rest (per trajectory: T FX', the rest) - 9.1 GB of HBM traffic per 1e4
// transforms at D = E = 10, N = 1181 for 25.6 MB of algorithmic bytes, and the full product fx Wc.  Here:
//   * a 512-thread workgroup owns TPW whole trajectories = TPW E <= 64 consecutive rows of FX (4 row tiles of 16);
//   * Wc = S + S' (S: lower triangle, half the diagonal), so fx Wc fx' = C + C' with C = (fx S) fx': the column tiles of S
//     are processed in PANELS of 13 (208 columns: the accumulators of k_bq_fused, 7 tiles per wave, wave w on row tile w & 3 and
//     on the column tiles of one parity); panel p needs the k-blocks kb >= 13 p only, and of its first 13 k-blocks only the
//     tiles on or below the diagonal.  T never exists: when a panel's k loop ends, its accumulators are multiplied with the
//     panel's FX columns (C += T_p FX_p') and cleared.  Panel 0 carries the G tile [Wcc' | wm]: cross-covariance and mean are
//     by-products;
//   * X slabs (16 k-rows x 224 columns; L2-resident, every workgroup walks the same sequence) double-buffered in LDS, requested
//     one step ahead; FX fragments (16 rows x 16 k per wave and step) requested two steps ahead straight into registers;
//   * epilogue as k_bq_fused: all parts of C meet in LDS, one thread per (trajectory, e >= e2) forms C + C' and stores.
// Matrix work per 64-row tile at N = 1181 (74 column tiles): 74 75 / 2 + 74 = 2 849 tile steps x 4 instructions per row tile against
// 74 90 = 6 660 for the full [Wc | Wcc'] product.
// (A first version of this round did everything in ONE launch - persistent workgroups, factor and integrand in chunks of 208
// points through an LDS tile into a per-workgroup scratch block in fragment order: 4.19 ms per 1e4 transforms, of which 0.86 ms
// were the chunk loop - its stores drained at every barrier - against 0.27 ms for k_eval_wave, whose waves fill the chip.)
#include "ssmq_host.h"
#include "ssmq_wide.h"
#include <type_traits>

namespace ssmq {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kPanT = 13;                       // column tiles of S per panel
constexpr int kPanW = 16 * kPanT;               // 208 columns
constexpr int kPanX = kPanW + 16;               // panel row in memory: 208 columns of S + the G tile (panel 0 only)

struct BqStreamArgs {
    int32_t D, E, N, emv_broadcast, tpw, nkb, npan;   // nkb: k-blocks of 16 points = column tiles of S; npan: panels
    int64_t B, lda;                                    // FX [B E][lda], lda >= 16 nkb, zero beyond N
    const double *fx, *chol;                           // k_eval_wave's outputs: values (NaN rows where the factorisation failed), factors [B][D][D]
    const double *X;        // [npan][16 nkb][224]: panel p = columns 208 p .. of S, then (p = 0) [Wcc' | wm]; zero-padded
    const double *emv, *cov_add;                       // [E * E]; cov_add or null
    double cov_scale, ccov_scale;
    double *mean_f, *cov_f, *cov_fx;                   // element e of trajectory b at ptr[e * es + b]
    int64_t es;
};

struct StepIt {             // (panel, k-block) of one step of the flattened main loop; kb runs DOWN within a panel
    int p, kb;
};

__global__ __launch_bounds__(512, 1) void k_bq_stream(const BqStreamArgs g) {
    constexpr int TB = 512, RT = 4, KS = 16;
    constexpr int NT = kPanT, NX = kPanX, LB = NX + 4;
    constexpr int C0 = (NT + 2) / 2;               // 7 accumulator tiles per wave: S tiles 2 t + ch of the panel; G at GT of half GCH
    constexpr int GCH = NT & 1, GT = NT >> 1;
    constexpr int kGroup = 2;
    constexpr int KBS = 1;                          // k-blocks per step (2: half the barriers, but 30 spilled registers with the staggered halves)
    extern __shared__ __align__(16) double lds[];
    const int D = g.D, E = g.E;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int lip = 4 * (li & 3) + (li >> 2);
    const int rt = wave % RT, ch = wave / RT;
    const int TPW = g.tpw, rows = TPW * E;
    const int nkb = g.nkb, npan = g.npan;
    const int64_t bw0 = (int64_t)blockIdx.x * TPW;
    const int nb = (int)((g.B - bw0) < (int64_t)TPW ? (g.B - bw0) : (int64_t)TPW);
    const int vrows = nb * E;
    double *slab = lds;                             // [2 buffers][2 k-blocks][KS][LB]
    double *sP = slab + 4 * KS * LB;                // [2][RT][64][8] parts of C (a G wave's slot first serves as its G tile)
    double *sev = sP + 2 * RT * 64 * 8, *sca = sev + 256;
    double *smr = sca + 256;                        // [64] transformed means by row
    int *spair = (int *)(smr + 64);                 // [64] (e, e2) pairs
    int *srow = spair + 64;                         // [64] row -> (trajectory << 8) | output index
    if (tid < 64) {
        const int gq = tid / E;
        srow[tid] = (gq << 8) | (tid - gq * E);
    }
    const int npair = E * (E + 1) / 2;
    if (tid < npair) {
        int e = 0;
        while ((e + 1) * (e + 2) / 2 <= tid) ++e;
        spair[tid] = (e << 4) | (tid - e * (e + 1) / 2);
    }
    if (tid < E * E) {
        sev[tid] = g.emv[tid];
        sca[tid] = g.cov_add ? g.cov_add[tid] : 0.0;
    }
    const int sr = tid >> 5, shf = (tid >> 4) & 1, sc16 = tid & 15;
    auto phys = [](int k) { return 4 * (k & 3) + (k >> 2); };
    // a step = TWO k-blocks (kb, kb - 1; the second one missing at the end of a panel with an odd number of them): one workgroup
    // barrier per 32 points.  (One per 16 measured 17 % of the launch in barrier waits: 4.28 against 3.55 ms with the barriers
    // compiled out.)
    auto next_it = [&](StepIt it) {
        if (it.kb - KBS >= NT * it.p) return StepIt{it.p, it.kb - KBS};
        return StepIt{it.p + 1, nkb - 1};
    };
    // this wave's 16 rows of FX: rows beyond the tile's valid ones (a last, partial tile; the 4 padding rows of a 60-row tile) read
    // the tile's last valid row - their results are never stored
    const int64_t row0 = bw0 * E;
    const int myrow = (16 * rt + li) < vrows ? (16 * rt + li) : vrows - 1;
    const double *fxrow = g.fx + (row0 + myrow) * g.lda + lg;
    // Every request of a step is UNCONDITIONAL (an invalid step repeats the last valid addresses; slab columns a diagonal step does
    // not need are read all the same) and the steps run in straight-line groups of kGroup: the compiler's s_waitcnt placement
    // counts requests, and any request under a condition - or a loop header - makes it wait for ALL outstanding ones.
    const StepIt last_it{npan - 1, (KBS == 2 && NT * (npan - 1) + 1 < nkb) ? NT * (npan - 1) + 1 : NT * (npan - 1)};
    auto valid_it = [&](StepIt it) { return it.p < npan ? it : last_it; };
    double breg[2][C0], afs[2][4];        // (fragments: ONE step ahead is a 2.8 us lead with two k-blocks per step)
    // second k-block of a step: kb - 1, or (at the end of a panel with an odd number of k-blocks) kb once more - read, never used
    auto second = [&](StepIt it) { return it.kb - 1 >= NT * it.p ? it.kb - 1 : it.kb; };
    auto load_b = [&](StepIt it0) {
        const StepIt it = valid_it(it0);
        const double *pa = g.X + ((size_t)it.p * nkb * KS + (size_t)it.kb * KS + sr) * NX + 16 * shf + sc16;
        const double *pb = g.X + ((size_t)it.p * nkb * KS + (size_t)second(it) * KS + sr) * NX + 16 * shf + sc16;
#pragma unroll
        for (int j = 0; j < C0; ++j) {
            breg[0][j] = pa[32 * j];
            if constexpr (KBS == 2) breg[1][j] = pb[32 * j];
        }
    };
    auto park_b = [&](int buf) {
        double *dst = slab + buf * 2 * KS * LB + phys(sr) * LB + 16 * shf + sc16;
#pragma unroll
        for (int j = 0; j < C0; ++j) {
            dst[32 * j] = breg[0][j];
            if constexpr (KBS == 2) dst[KS * LB + 32 * j] = breg[1][j];
        }
    };
    auto load_af = [&](StepIt it0) {            // lane (li, lg): FX[row][16 kb + lg + 4 s], s = 0 .. 3, both k-blocks
        const StepIt it = valid_it(it0);
        const double *pa = fxrow + KS * it.kb, *pb = fxrow + KS * second(it);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            afs[0][s] = pa[4 * s];
            if constexpr (KBS == 2) afs[1][s] = pb[4 * s];
        }
    };
    StepIt c0{0, nkb - 1};
    StepIt c1 = next_it(c0), c2 = next_it(c1);
    load_b(c0);
    load_af(c0);
    park_b(0);
    load_b(c1);
    double *sx = sP + ((ch * RT + rt) * 64 + lane) * 8;     // this wave's part of C lives in LDS between the panels
#pragma unroll
    for (int i = 0; i < 8; ++i) sx[i] = 0.0;
    __syncthreads();
    v4d acc[C0];
#pragma unroll
    for (int t = 0; t < C0; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
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
    auto panel_end = [&](int p) {
        const int ntp = nkb - NT * p;
        if (p == 0 && ch == GCH) {
            // mean and cross-covariance from the G tile (as k_bq_fused 4a), the factor from k_eval_wave's output
            const int lr = 16 * rt + li;
            const bool valid = lr < vrows;
            const int gi = srow[lr] >> 8, e = srow[lr] & 255;
            const int64_t b = bw0 + (valid ? gi : 0);
            const double *Lb = g.chol + b * D * D;
            const v4d gt = acc[GT];
            if (lg == 3) {
                smr[lr] = gt[3];
                if (valid) g.mean_f[(int64_t)e * g.es + b] = gt[3];
            }
            double *sg = sP + ((GCH * RT + rt) * 64) * 8;     // this wave's own slot (512 doubles), zeroed again below
#pragma unroll
            for (int r = 0; r < 4; ++r) sg[li * 16 + 4 * lg + r] = gt[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int j = lg; j < D; j += 4) {
                double pq = 0.0;
                for (int d = 0; d <= j; ++d) pq += sg[li * 16 + d] * Lb[j * D + d];
                if (valid) g.cov_fx[(int64_t)(e * D + j) * g.es + b] = pq * g.ccov_scale;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < 8; ++i) sx[i] = 0.0;
        }
        // C += T_p FX_p': this wave's tiles of the panel against the rows of the trajectories that meet its row tile.  Straight-
        // line code: the FX fragments of tile t + 1 are requested before the eight matrix instructions of tile t are issued.  Tiles
        // this panel does not have (the last panel; tile GT of the G half) multiply accumulators that are zero / are skipped.
        const int r0 = (s0 + li) < vrows ? (s0 + li) : vrows - 1, r1 = (s0 + 16 + li) < vrows ? (s0 + 16 + li) : vrows - 1;
        const double *f0p = g.fx + (row0 + r0) * g.lda + 4 * lg, *f1p = g.fx + (row0 + r1) * g.lda + 4 * lg;
        const int nmine = ((ntp < NT ? ntp : NT) - ch + 1) >> 1;         // S tiles of this wave in this panel
        v4d acc2[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
        v4d f[2][2];
        auto ldf = [&](int t, v4d (&dst)[2]) {
            const int tc = t < nmine ? t : 0;
            const int col = 16 * (NT * p + 2 * tc + ch);
            dst[0] = *(const v4d *)(f0p + col);
            dst[1] = *(const v4d *)(f1p + col);
        };
        ldf(0, f[0]);
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            ldf(t + 1, f[(t + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[t][r], f[t & 1][0][r], acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[t][r], f[t & 1][1][r], acc2[1], 0, 0, 0);
            }
        }
        if (ch != GCH) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[GT][r], f[GT & 1][0][r], acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[GT][r], f[GT & 1][1][r], acc2[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) sx[4 * h + r] += acc2[h][r];
#pragma unroll
        for (int t = 0; t < C0; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    };
    auto half_step = [&](const double *sb, const double (&af)[4], int kb, bool on) {
        const int kbl = kb - NT * c0.p, ntp = nkb - NT * c0.p;
        int na = (kbl - ch + 2) >> 1;
        const int cap = ((ntp < NT ? ntp : NT) - ch + 1) >> 1;
        na = na < cap ? na : cap;
        if (!on) na = 0;                                // padding step of the last group / missing second k-block
        if (on && c0.p == 0 && ch == GCH) {
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
    };
    auto step = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        const double afa[4] = {afs[0][0], afs[0][1], afs[0][2], afs[0][3]};
        const double afb[4] = {afs[1][0], afs[1][1], afs[1][2], afs[1][3]};
        const double *sb = slab + PAR * 2 * KS * LB;
        const bool on = c0.p < npan;
        // The two waves of a SIMD (row tile rt, halves ch = 0 / 1) run the step's two parts in OPPOSITE order - half 0 moves data
        // first (slabs of step q + 1 into LDS, requests for q + 2), then multiplies; half 1 multiplies first - so that one wave's
        // matrix instructions cover the other's stores, requests and waits.  Both in the same order left the matrix pipe idle
        // while both moved data: the barrier keeps the eight waves in phase (3.55 ms with the barriers compiled out, 4.28 with).
        if (ch == 0) {
            park_b(PAR ^ 1);
            load_b(c2);
            load_af(c1);
            half_step(sb, afa, c0.kb, on);
            if constexpr (KBS == 2) half_step(sb + KS * LB, afb, c0.kb - 1, on && c0.kb - 1 >= NT * c0.p);
        } else {
            half_step(sb, afa, c0.kb, on);
            if constexpr (KBS == 2) half_step(sb + KS * LB, afb, c0.kb - 1, on && c0.kb - 1 >= NT * c0.p);
            park_b(PAR ^ 1);
            load_b(c2);
            load_af(c1);
        }
        // a barrier for the LDS slabs only: __syncthreads() is also a fence on global memory, i.e. s_waitcnt vmcnt(0) - every
        // request issued ahead (next slabs, fragments of the step after next) would be waited for at the end of EVERY step
#ifdef BQS_NO_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        if (on && c0.kb - KBS < NT * c0.p) panel_end(c0.p);
        c0 = c1; c1 = c2; c2 = next_it(c2);
    };
    static_assert(kGroup % 2 == 0, "register sets and slab buffers alternate with the step");
#ifdef BQS_SKIP_MAIN
    while (false) {
#else
    while (c0.p < npan) {
#endif
#pragma unroll
        for (int u = 0; u < kGroup; u += 2) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
        }
    }
    // ---- all parts of C meet in LDS; fx Wc fx' = C + C' ---------------------------------------------------------------------------
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
        double v = (cval(l1, l2) + cval(l2, l1) - smr[l1] * smr[l2] + em) * g.cov_scale;
        if (g.cov_add) v += sca[ie];
        g.cov_f[(int64_t)ie * g.es + b] = v;
        if (e2 != e) g.cov_f[(int64_t)it * g.es + b] = v;
    }
}

constexpr size_t kStreamLds = sizeof(double) * (4 * 16 * (kPanX + 4) + 2 * 4 * 64 * 8 + 256 + 256 + 64) + sizeof(int) * 128;

}  // namespace

int bq_stream_panels(int N) { return (N + kPanW - 1) / kPanW; }
int bq_stream_kblocks(int N) { return (N + 15) / 16; }
size_t bq_stream_x_doubles(int N) { return (size_t)bq_stream_panels(N) * bq_stream_kblocks(N) * 16 * kPanX; }

// BQ transform (not the t-process one), one constant block for the batch, 208 < N <= SSMQ_MAX_PTS, a symmetric Wc; whole
// trajectories fill at least 3/4 of a 64-row tile for every E <= 10
bool bq_stream_supported(int D, int E, int N) {
    if (getenv("SSMQ_NO_BQ_STREAM") || getenv("SSMQ_NO_MFMA")) return false;
    if (N <= kPanW || N > SSMQ_MAX_PTS) return false;
    return D >= 1 && D <= 15 && E >= 1 && E <= 10;      // D <= 15: column 15 of the G tile carries wm
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

// a: WideArgs of the whole transform (outputs, scales, cov_add; unit batch strides); fx [B E][lda] and chol [B][D][D] as
// k_eval_wave left them (lda >= 16 ceil(N / 16), zero beyond N)
int launch_bq_stream(const WideArgs &a, const double *X, const double *emv, int emv_broadcast, int64_t B, const double *fx,
                     const double *chol, int64_t lda, hipStream_t s) {
    if (B <= 0) return SSMQ_OK;
    if (!bq_stream_supported(a.D, a.E, a.N) || a.consts_stride != 0 || a.form != SSMQ_FORM_BQ || a.tp_nu > 0.0 ||
        lda < 16 * bq_stream_kblocks(a.N) || (lda & 3) || a.bs_mf != 1 || a.bs_cf != 1 || a.bs_cfx != 1) {
        set_error("bq_stream: shape not supported");
        return SSMQ_E_UNSUPPORTED;
    }
    static unsigned attr_epoch = 0;
    if (attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_bq_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStreamLds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_bq_stream)");
        attr_epoch = device_epoch();
    }
    BqStreamArgs g;
    g.D = a.D; g.E = a.E; g.N = a.N; g.emv_broadcast = emv_broadcast; g.tpw = 64 / a.E;
    g.nkb = bq_stream_kblocks(a.N); g.npan = bq_stream_panels(a.N);
    g.B = B; g.lda = lda; g.fx = fx; g.chol = chol; g.X = X; g.emv = emv; g.cov_add = a.cov_add;
    g.cov_scale = a.cov_scale; g.ccov_scale = a.ccov_scale; g.mean_f = a.mean_f; g.cov_f = a.cov_f; g.cov_fx = a.cov_fx; g.es = a.es_out;
    const int64_t tiles = (B + g.tpw - 1) / g.tpw;
    hipLaunchKernelGGL(k_bq_stream, dim3((unsigned)tiles), dim3(512), kStreamLds, s, g);
    return hip_fail(hipGetLastError(), "k_bq_stream");
}

}  // namespace ssmq
