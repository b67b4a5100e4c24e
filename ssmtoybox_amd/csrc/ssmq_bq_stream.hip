// BQ moment transform for LARGE point sets (209 ... 4096 points) in TWO launches: k_eval_wave (factor, sigma points, integrand
// values FX to memory in fragment order - WideArgs::fx_frag) and this kernel - k_bq_fused's product with the point axis tiled,
// nothing else through memory.  (bq/bqmtran.py:158-223: mean = fx wm, cov = fx Wc fx' - mean mean' + emv, ccov = fx Wcc' L'.)
//
// Round 3's route for these sizes was three launches - k_eval_wave, k_fxwc_mfma<16,1> (T = FX [Wc | Wcc'] by column blocks,
// T to HBM: 0.95 GB written, read again), k_big_rest (per trajectory: T FX', the rest) - 9.1 GB of HBM traffic per 1e4
// transforms at D = E = 10, N = 1181 for 25.6 MB of algorithmic bytes, and the full product fx Wc.  Here:
//   * a 512-thread workgroup owns TPW whole trajectories = TPW E <= 64 rows of FX (4 row tiles of 16);
//   * Wc = S + S' (S: lower triangle, half the diagonal), so fx Wc fx' = C + C' with C = (fx S) fx': the column tiles of S
//     are processed in PANELS of 16 (256 columns); wave w owns column tiles w and 15 - w of every panel for all four row tiles
//     (8 accumulator tiles).  Panel p needs the k-blocks kb >= 16 p only, and tile t of it only kb >= 16 p + t.  T never exists:
//     when a wave's k loop of a panel ends, its accumulators are multiplied with the panel's FX columns (C += T_p FX_p') and
//     cleared.  The G tile [Wcc' | wm] (cross-covariance, mean) is spread over the waves inside panel 0;
//   * NO LDS staging, no data-carrying barrier in the main loop: both operands straight from memory into registers in fragment order
//     (a lane's four k values of a tile = one 32-byte read, a wave's reads of a tile = 2 KB contiguous), requested one step ahead; a bare
//     rendezvous every 16 regular steps keeps the eight waves close enough for the FX fragments one has fetched to be still in L2
//     when the others ask (3.7 instead of 6-10 GB of L2-miss reads at the same 3.1 ms);
//   * epilogue as k_bq_fused: all parts of C meet in LDS, one thread per (trajectory, e >= e2) forms C + C' and stores.
// Matrix work per 64-row block at N = 1181 (74 column tiles): 74 75 / 2 + 74 = 2 849 tile steps x 4 instructions per row tile against
// 74 90 = 6 660 for the full [Wc | Wcc'] product.
// (Earlier versions of this round, DESIGN.md 3.12: ONE launch with the integrand inside - 4.19 ms per 1e4 transforms; X slabs
// double-buffered in LDS with a barrier per 16 points - 4.06 ms, the matrix pipe 59 % busy; this ownership with a row-major FX -
// 3.71 ms, the texture path 4 x longer busy per fragment read than with contiguous ones.  Now 3.05-3.17 ms.)
#include "ssmq_host.h"
#include "ssmq_wide.h"
#include <type_traits>

namespace ssmq {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kPanT = 16;                       // column tiles of S per panel: two per wave
constexpr int kPanW = 16 * kPanT;               // 256 columns
constexpr int kFrag = 256;                      // doubles of one 16 x 16 tile in fragment order
constexpr int kCholLds = 960;                   // doubles of LDS for the block's Cholesky factors

struct BqStreamArgs {
    int32_t D, E, N, emv_broadcast, tpw, nkb, npan;   // nkb: k-blocks of 16 points = column tiles of S; npan: panels
    int64_t B, lda;                                    // FX in fragment order (WideArgs::fx_frag = tpw), lda = 16 nkb, zero beyond N
    const double *fx, *chol;                           // k_eval_wave's outputs: values (NaN rows where the factorisation failed), factors [B][D][D]
    const double *X;        // [npan][nkb][16 tiles][2][64][2] S in fragment order, then [nkb][2][64][2] the G tile [Wcc' | wm]
    const double *emv, *cov_add;                       // [E * E]; cov_add or null
    double cov_scale, ccov_scale;
    double *mean_f, *cov_f, *cov_fx;                   // element e of trajectory b at ptr[e * es + b]
    int64_t es;
    // Workgroups [0, n_whole) run whole blocks; the n_tail blocks behind them - what would be a last, partly empty round of
    // workgroups on the chip - are cut by PANEL: unit n_whole + p n_tail + j is panel p of block n_whole + j (panel-major, i.e.
    // longest first) and leaves its waves' parts of C in parts[p][trajectory of the tail][pair][2][8]; k_bq_stream_finish adds them
    // up in the order a whole block does - a trajectory's result does not depend on where in a batch it sits.
    int32_t n_whole, n_tail;
    double *parts;
};

struct StepIt {             // (panel, k-block) of one step of a wave's flattened main loop; kb runs DOWN within a panel
    int p, kb;
};

__global__ __launch_bounds__(512, 1) void k_bq_stream(const BqStreamArgs g) {
    constexpr int TB = 512, RT = 4, NW = 8, NT = kPanT;
    extern __shared__ __align__(16) double lds[];
    const int D = g.D, E = g.E;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int TPW = g.tpw;
    const int nkb = g.nkb, npan = g.npan;
    int blk = blockIdx.x, p_lo = 0, p_hi = npan - 1;
    const bool partial = (int)blockIdx.x >= g.n_whole;
    if (partial) {
        const int u = (int)blockIdx.x - g.n_whole;
        p_lo = p_hi = u / g.n_tail;
        blk = g.n_whole + (u - p_lo * g.n_tail);
    }
    const int64_t bw0 = (int64_t)blk * TPW;
    const int nb = (int)((g.B - bw0) < (int64_t)TPW ? (g.B - bw0) : (int64_t)TPW);
    const int vrows = nb * E;
    double *sP = lds;                               // [NW][RT][64][8] every wave's part of C
    double *sG = sP + NW * RT * 64 * 8;             // [NW][64][4] every wave's part of the G tile
    double *sev = sG + NW * 64 * 4, *sca = sev + 256;
    double *smr = sca + 256;                        // [64] transformed means by row
    double *sL = smr + 64;                          // [TPW][D][D] Cholesky factors of this block's trajectories, if they fit kCholLds
    int *spair = (int *)(sL + kCholLds);            // [64] (e, e2) pairs
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
    // the factors the cross-covariance needs at the very end: requested now, read from LDS then (straight from memory they were
    // ~30 dependent reads per lane with every other wave waiting at the barrier)
    const bool factors_in_lds = TPW * D * D <= kCholLds;     // (many one-output trajectories per block: from memory, as before)
    if (factors_in_lds)
        for (int idx = tid; idx < nb * D * D; idx += TB) sL[idx] = g.chol[bw0 * D * D + idx];
    // This wave's column tiles of every panel: tA = wave and tB = 15 - wave.  Within a panel the k-blocks run DOWN to the panel's
    // own 16 (the diagonal region), where tile t has work for kb >= 16 p + t only: the pair (w, 15 - w) gives every wave the same
    // 17 tile steps there.  A wave's steps of panel p: kb = nkb - 1 ... 16 p + wave.
    // (Panel 0 runs down to kb = 0 for every wave: the G tile [Wcc' | wm] is spread over the waves as (row tile wave & 3) x
    // (k-blocks of parity wave >> 2), 37 tile steps each, with the FX fragments the step has loaded anyway.)
    const int tA = wave, tB = NT - 1 - wave;
#ifndef BQS_RENDEZVOUS
#define BQS_RENDEZVOUS 16       // the eight waves meet every BQS_RENDEZVOUS steps of the regular (below-diagonal) part of a panel;
                                // 0: never (A/B builds: 3.1 ms either way, 6-10 GB of L2-miss reads instead of 3.7)
#endif
    auto lo = [&](int p) { return p == 0 ? 0 : NT * p + wave; };
    int nstep = 0;
    auto next_it = [&](StepIt it) {
        if (it.kb - 1 >= lo(it.p)) return StepIt{it.p, it.kb - 1};
        const int p = it.p + 1;
        if (p <= p_hi && lo(p) <= nkb - 1) return StepIt{p, nkb - 1};
        return StepIt{npan, nkb - 1};
    };
    auto valid_it = [&](StepIt it) { return it.p < npan ? it : StepIt{0, nkb - 1}; };
    // FX in fragment order (WideArgs::fx_frag): this block's 64 rows are [4 row tiles][nkb][64 lanes][4].  Rows beyond the block's
    // valid ones (a last, partial block; the 4 padding rows of a 60-row block) hold whatever the buffer held: a row of FX only ever
    // meets its own accumulator column, and those are never stored.
    const double *fxb = g.fx + (int64_t)blk * 64 * g.lda;
    const int tile_ld = nkb * kFrag;               // doubles from one row tile of the block to the next
    const double *Xg = g.X + (size_t)npan * nkb * NT * kFrag;
    // operands of a step, requested ONE step ahead straight into registers (fragment order in memory: a lane's four k values of a
    // tile are two 16-byte pieces, each piece contiguous over the wave).  Every request is unconditional - an invalid step repeats
    // a valid address - and the steps run in straight-line pairs, so that the compiler's s_waitcnt placement can count them.
    v4d av[2][RT], gv[2];
    v2d bv[2][2][2];
    auto load_step = [&](StepIt it0, auto set_c) {
        constexpr int set = decltype(set_c)::value;
#ifdef BQS_SAME_ADDR
        const StepIt it = StepIt{0, nkb - 1};
#else
        const StepIt it = valid_it(it0);
#endif
#ifdef BQS_SAME_A
        const int kba = nkb - 1;
#else
        const int kba = it.kb;
#endif
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) av[set][rt] = *(const v4d *)(fxb + rt * tile_ld + kba * kFrag + 4 * lane);
#ifdef BQS_SAME_B
        const double *pb = g.X + ((size_t)0 * nkb + nkb - 1) * (NT * kFrag) + 2 * lane;
#else
        const double *pb = g.X + ((size_t)it.p * nkb + it.kb) * (NT * kFrag) + 2 * lane;
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bv[set][0][h] = *(const v2d *)(pb + tA * kFrag + 128 * h);
            bv[set][1][h] = *(const v2d *)(pb + tB * kFrag + 128 * h);
        }
        gv[set] = *(const v4d *)(Xg + (size_t)it.kb * kFrag + 4 * lane);
    };
    v4d accA[RT], accB[RT], gacc = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) accA[rt] = accB[rt] = v4d{0.0, 0.0, 0.0, 0.0};
    double *sx = sP + (size_t)wave * RT * 64 * 8 + lane * 8;       // + rt * 512: this wave's part of C for row tile rt
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 8; ++i) sx[rt * 512 + i] = 0.0;
    const int grt = wave & 3, gpar = wave >> 2;         // the G tile: row tile grt at the k-blocks of parity gpar
    StepIt c0 = (lo(p_lo) <= nkb - 1) ? StepIt{p_lo, nkb - 1} : StepIt{npan, nkb - 1};   // (a wave without a tile in this panel)
    StepIt c1 = next_it(c0);
    load_step(c0, std::integral_constant<int, 0>{});
    __syncthreads();                                // the tables
    // C += T_p FX_p' for this wave's tiles of panel p, then the accumulators are cleared: acc (T', S columns x rows) is the A
    // operand as it stands, the FX rows of the trajectories that meet the row tile are B
    auto panel_end = [&](int p) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int s0 = (16 * rt / E) * E;     // first row of the first trajectory that intersects row tile rt
            const int r0 = (s0 + li) < vrows ? (s0 + li) : vrows - 1, r1 = (s0 + 16 + li) < vrows ? (s0 + 16 + li) : vrows - 1;
            // row r, points 16 kb + 4 lg ...: tile (r >> 4, kb), lane (r & 15) + 16 lg
            const double *f0p = fxb + (r0 >> 4) * tile_ld + 4 * ((r0 & 15) + 16 * lg) + NT * p * kFrag;
            const double *f1p = fxb + (r1 >> 4) * tile_ld + 4 * ((r1 & 15) + 16 * lg) + NT * p * kFrag;
            const int cB = (NT * p + tB < nkb) ? tB : tA;          // a tile beyond the last one: accB is zero, any address does
            const v4d fa0 = *(const v4d *)(f0p + tA * kFrag), fa1 = *(const v4d *)(f1p + tA * kFrag);
            const v4d fb0 = *(const v4d *)(f0p + cB * kFrag), fb1 = *(const v4d *)(f1p + cB * kFrag);
            v4d c2a = v4d{0.0, 0.0, 0.0, 0.0}, c2b = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c2a = __builtin_amdgcn_mfma_f64_16x16x4f64(accA[rt][r], fa0[r], c2a, 0, 0, 0);
                c2b = __builtin_amdgcn_mfma_f64_16x16x4f64(accA[rt][r], fa1[r], c2b, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c2a = __builtin_amdgcn_mfma_f64_16x16x4f64(accB[rt][r], fb0[r], c2a, 0, 0, 0);
                c2b = __builtin_amdgcn_mfma_f64_16x16x4f64(accB[rt][r], fb1[r], c2b, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sx[rt * 512 + r] += c2a[r];
                sx[rt * 512 + 4 + r] += c2b[r];
            }
            accA[rt] = accB[rt] = v4d{0.0, 0.0, 0.0, 0.0};
        }
    };
    auto step = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        load_step(c1, std::integral_constant<int, PAR ^ 1>{});
        const bool on = c0.p < npan;
        const int kbl = c0.kb - NT * c0.p;
#ifdef BQS_NO_MMA
        if (on) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) asm volatile("" ::"v"(av[PAR][rt]));
            asm volatile("" ::"v"(bv[PAR][0][0]), "v"(bv[PAR][0][1]), "v"(bv[PAR][1][0]), "v"(bv[PAR][1][1]), "v"(gv[PAR]));
            if (c0.kb == lo(c0.p)) panel_end(c0.p);
        }
        if (false) {
#else
        if (on) {
#endif
            if (kbl >= tA) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double b = bv[PAR][0][s >> 1][s & 1];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) accA[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, av[PAR][rt][s], accA[rt], 0, 0, 0);
                }
            }
            if (kbl >= tB) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double b = bv[PAR][1][s >> 1][s & 1];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) accB[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, av[PAR][rt][s], accB[rt], 0, 0, 0);
                }
            }
            if (c0.p == 0 && (c0.kb & 1) == gpar) {
                // (four branches, not a selected operand: the compiler turns a select over av[] into an indexed read of the array in
                // scratch memory - and stores every fragment there the moment it arrives)
#define SSMQ_G_TILE(RTG) \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) \
                    gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[PAR][s], av[PAR][RTG][s], gacc, 0, 0, 0);
                if (grt == 0) { SSMQ_G_TILE(0) } else if (grt == 1) { SSMQ_G_TILE(1) } else if (grt == 2) { SSMQ_G_TILE(2) } else { SSMQ_G_TILE(3) }
#undef SSMQ_G_TILE
            }
            if (c0.kb == lo(c0.p)) panel_end(c0.p);
        }
        if constexpr (BQS_RENDEZVOUS > 0) {
            // Every wave runs every step below a panel's diagonal region (kb >= 16 (p + 1)), so counting those gives all eight the
            // same number of barriers: a bare rendezvous, no memory operation waits for it.  It keeps the waves within a few steps
            // of each other where they all do the same work, so that the FX fragments one of them has fetched are still in L2
            // when the others ask; in the diagonal region (different work per wave, equal per SIMD) they run freely.
            if (on && kbl >= NT && (++nstep % BQS_RENDEZVOUS) == 0) asm volatile("s_barrier" ::: "memory");
        }
        c0 = c1; c1 = next_it(c1);
    };
#ifdef BQS_SKIP_MAIN
    while (false) {
#else
    while (c0.p < npan) {
#endif
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sG[(wave * 64 + lane) * 4 + r] = gacc[r];
    __syncthreads();
    // ---- mean and cross-covariance from the G tile (as k_bq_fused 4a), the factor from k_eval_wave's output -------------------------
    if (wave < RT && p_lo == 0) {
        const int rt = wave, lr = 16 * rt + li;
        const bool valid = lr < vrows;
        const int gi = srow[lr] >> 8, e = srow[lr] & 255;
        const int64_t b = bw0 + (valid ? gi : 0);
        const double *Lb = factors_in_lds ? sL + (valid ? gi : 0) * D * D : g.chol + b * D * D;
        const v4d gt = *(const v4d *)(sG + (wave * 64 + lane) * 4) + *(const v4d *)(sG + ((wave + 4) * 64 + lane) * 4);
        if (lg == 3) {
            smr[lr] = gt[3];
            if (valid) g.mean_f[(int64_t)e * g.es + b] = gt[3];
        }
        double *sg = sG + wave * 256;                  // this wave's own slot: [row li][column 4 lg + r]
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
    }
    __syncthreads();
    // ---- all parts of C meet in LDS; fx Wc fx' = C + C' ---------------------------------------------------------------------------
    for (int idx = tid; idx < nb * npair; idx += TB) {
        const int p = idx / nb, gi = idx - p * nb;
        const int e = spair[p] >> 4, e2 = spair[p] & 15;
        const int l1 = gi * E + e, l2 = gi * E + e2;
        auto cat = [&](int la, int lb) {
            const int rta = la >> 4, i = la & 15, j = lb - (16 * rta / E) * E;
            return ((rta * 64) + (j & 15) + 16 * (i & 3)) * 8 + 4 * (j >> 4) + (i >> 2);
        };
        auto cval = [&](int at) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += sP[w * RT * 64 * 8 + at];
            return v;
        };
        const int64_t b = bw0 + gi;
        if (partial) {
            // every wave's part of the two entries, this panel only: k_bq_stream_finish adds the panels per wave (the order in
            // which a whole block's wave accumulates them), then the waves, then applies the formula below - the same bits
            double *dst = g.parts + (((size_t)p_lo * g.n_tail * TPW + (size_t)(blk - g.n_whole) * TPW + gi) * npair + p) * (2 * NW);
            const int a1 = cat(l1, l2), a2 = cat(l2, l1);
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                dst[w] = sP[w * RT * 64 * 8 + a1];
                dst[NW + w] = sP[w * RT * 64 * 8 + a2];
            }
            continue;
        }
        const int ie = e * E + e2, it = e2 * E + e;
        const bool use = (e == e2) || g.emv_broadcast;
        const double em = use ? sev[ie] : 0.0;
        double v = (__builtin_fma(-smr[l1], smr[l2], cval(cat(l1, l2)) + cval(cat(l2, l1))) + em) * g.cov_scale;
        if (g.cov_add) v += sca[ie];
        g.cov_f[(int64_t)ie * g.es + b] = v;
        if (e2 != e) g.cov_f[(int64_t)it * g.es + b] = v;
    }
}

// grid over (tail trajectory, pair): cov = (sum over panels of the parts, in panel order, - mean mean' + emv) scale + cov_add
__global__ __launch_bounds__(256) void k_bq_stream_finish(const BqStreamArgs g) {
    const int E = g.E, npair = E * (E + 1) / 2, TPW = g.tpw;
    const int64_t nt = (int64_t)g.n_tail * TPW;                     // trajectory slots of the tail (the last block may be partial)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nt * npair) return;
    const int64_t j = idx / npair;
    const int p = (int)(idx - j * npair);
    const int64_t b = (int64_t)g.n_whole * TPW + j;
    if (b >= g.B) return;
    int e = 0;
    while ((e + 1) * (e + 2) / 2 <= p) ++e;
    const int e2 = p - e * (e + 1) / 2;
    constexpr int NW = 8;
    double c[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            double sw = 0.0;                                          // what this wave's LDS slot holds in a whole block
            for (int q = 0; q < g.npan; ++q) sw += g.parts[(((size_t)q * nt + j) * npair + p) * (2 * NW) + h * NW + w];
            c[h] += sw;
        }
    const int ie = e * E + e2, it = e2 * E + e;
    const bool use = (e == e2) || g.emv_broadcast;
    const double em = use ? g.emv[ie] : 0.0;
    double v = (__builtin_fma(-g.mean_f[(int64_t)e * g.es + b], g.mean_f[(int64_t)e2 * g.es + b], c[0] + c[1]) + em) * g.cov_scale;
    if (g.cov_add) v += g.cov_add[ie];
    g.cov_f[(int64_t)ie * g.es + b] = v;
    if (e2 != e) g.cov_f[(int64_t)it * g.es + b] = v;
}

constexpr size_t kStreamLds = sizeof(double) * (8 * 4 * 64 * 8 + 8 * 64 * 4 + 256 + 256 + 64 + kCholLds) + sizeof(int) * 128;

}  // namespace

int bq_stream_panels(int N) { return (N + kPanW - 1) / kPanW; }
int bq_stream_kblocks(int N) { return (N + 15) / 16; }
int bq_stream_tpw(int E) { return 64 / E; }

// How many blocks of a batch run whole and how many - the last, partly empty round of workgroups on `cus` compute units - are
// cut by panel (BqStreamArgs::n_whole / n_tail).  1 667 blocks on 256 CUs are 6.51 rounds: the seventh costs a whole block's time
// with half the chip idle; cut into 5 x 131 units dealt longest first it costs about half of that.
static void bq_stream_split(int64_t nblocks, int npan, int cus, int *n_whole, int *n_tail) {
    *n_whole = (int)nblocks;
    *n_tail = 0;
    if (npan < 2 || cus < 1 || ssmq::sw("SSMQ_BQ_STREAM_NO_SPLIT")) return;
    const int64_t rem = nblocks % cus;
    if (rem == 0 || 5 * rem > 3 * cus) return;      // a last round that is more than 60 % full: leave it
    *n_tail = (int)rem;
    *n_whole = (int)(nblocks - rem);
}
size_t bq_stream_parts_doubles(int E, int N, int64_t B, int cus) {
    const int tpw = bq_stream_tpw(E);
    int nw, nt;
    bq_stream_split((B + tpw - 1) / tpw, bq_stream_panels(N), cus, &nw, &nt);
    return (size_t)bq_stream_panels(N) * nt * tpw * (E * (E + 1) / 2) * 16;      // per pair: 8 waves x the two entries (e, e2), (e2, e)
}
size_t bq_stream_x_doubles(int N) { return ((size_t)bq_stream_panels(N) * kPanT + 1) * bq_stream_kblocks(N) * kFrag; }

// BQ transform (not the t-process one), one constant block for the batch, 208 < N <= SSMQ_MAX_PTS, a symmetric Wc; whole
// trajectories fill at least 3/4 of a 64-row tile for every E <= 10
bool bq_stream_supported(int D, int E, int N) {
    if (ssmq::sw("SSMQ_NO_BQ_STREAM") || ssmq::sw("SSMQ_NO_MFMA")) return false;
    // (A/B: SSMQ_BQ_STREAM_MIN_N lowers the bound - the two-launch route at N = 201 against k_bq_fused)
    const char *mn = ssmq::sw("SSMQ_BQ_STREAM_MIN_N");
    if (N <= (mn ? atoi(mn) : 208) || N > SSMQ_MAX_PTS) return false;
    return D >= 1 && D <= 15 && E >= 1 && E <= 10;      // D <= 15: column 15 of the G tile carries wm
}

// X in the fragment layout of BqStreamArgs from the natural-layout weights (host): S = tril(Wc) with half the diagonal.
// Tile (k-block kb, column tile ct): lane (li, lg) holds rows k = 16 kb + 4 lg + s, s = 0 .. 3, of column 16 ct + lip(li),
// lip(li) = 4 (li & 3) + (li >> 2) - the accumulator rows of the f64 matrix instruction are lg + 4 r, and the second product
// reads its operand straight out of the accumulators with the points 4 lg + r in its k slots.
void bq_stream_pack(int D, int N, const double *Wc, const double *Wcc, const double *wm, double *X) {
    const int npan = bq_stream_panels(N), nkb = bq_stream_kblocks(N);
    const size_t total = bq_stream_x_doubles(N);
    for (size_t i = 0; i < total; ++i) X[i] = 0.0;
    auto at = [](int lane, int s) { return (size_t)(s >> 1) * 128 + 2 * lane + (s & 1); };
    for (int k = 0; k < N; ++k) {
        const int kb = k >> 4, lg = (k & 15) >> 2, s = k & 3;
        for (int j = 0; j <= k; ++j) {
            const int ct = j >> 4, p = ct / kPanT, t = ct - p * kPanT, c = j & 15;
            const int li = 4 * (c & 3) + (c >> 2);      // lip(li) == c  (the permutation is its own inverse)
            X[(((size_t)p * nkb + kb) * kPanT + t) * kFrag + at(li + 16 * lg, s)] =
                j == k ? 0.5 * Wc[(size_t)k * N + k] : Wc[(size_t)k * N + j];
        }
        double *G = X + (size_t)npan * nkb * kPanT * kFrag + (size_t)kb * kFrag;
        for (int c = 0; c < 16; ++c) {
            const int li = 4 * (c & 3) + (c >> 2);
            const double v = c < D ? Wcc[(size_t)c * N + k] : (c == 15 && D <= 15) ? wm[k] : 0.0;
            G[4 * (li + 16 * lg) + s] = v;
        }
    }
}

// a: WideArgs of the whole transform (outputs, scales, cov_add; unit batch strides); fx [B E][lda] and chol [B][D][D] as
// k_eval_wave left them (lda >= 16 ceil(N / 16), zero beyond N)
int launch_bq_stream(const WideArgs &a, const double *X, const double *emv, int emv_broadcast, int64_t B, const double *fx,
                     const double *chol, int64_t lda, int cus, double *parts, hipStream_t s) {
    if (B <= 0) return SSMQ_OK;
    if (!bq_stream_supported(a.D, a.E, a.N) || a.consts_stride != 0 || a.form != SSMQ_FORM_BQ || a.tp_nu > 0.0 ||
        lda != 16 * bq_stream_kblocks(a.N) || a.bs_mf != 1 || a.bs_cf != 1 || a.bs_cfx != 1) {
        set_error("bq_stream: shape not supported");
        return SSMQ_E_UNSUPPORTED;
    }
    static thread_local unsigned attr_epoch = 0;
    if (attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_bq_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStreamLds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_bq_stream)");
        attr_epoch = device_epoch();
    }
    BqStreamArgs g;
    g.D = a.D; g.E = a.E; g.N = a.N; g.emv_broadcast = emv_broadcast; g.tpw = bq_stream_tpw(a.E);
    g.nkb = bq_stream_kblocks(a.N); g.npan = bq_stream_panels(a.N);
    g.B = B; g.lda = lda; g.fx = fx; g.chol = chol; g.X = X; g.emv = emv; g.cov_add = a.cov_add;
    g.cov_scale = a.cov_scale; g.ccov_scale = a.ccov_scale; g.mean_f = a.mean_f; g.cov_f = a.cov_f; g.cov_fx = a.cov_fx; g.es = a.es_out;
    const int64_t tiles = (B + g.tpw - 1) / g.tpw;
    bq_stream_split(tiles, g.npan, parts ? cus : 0, &g.n_whole, &g.n_tail);
    g.parts = parts;
    hipLaunchKernelGGL(k_bq_stream, dim3((unsigned)(g.n_whole + (int64_t)g.n_tail * g.npan)), dim3(512), kStreamLds, s, g);
    int rc = hip_fail(hipGetLastError(), "k_bq_stream");
    if (!rc && g.n_tail > 0) {
        const int64_t items = (int64_t)g.n_tail * g.tpw * (g.E * (g.E + 1) / 2);
        hipLaunchKernelGGL(k_bq_stream_finish, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, g);
        rc = hip_fail(hipGetLastError(), "k_bq_stream_finish");
    }
    return rc;
}

}  // namespace ssmq
