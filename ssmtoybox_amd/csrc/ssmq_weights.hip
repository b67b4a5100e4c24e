// Quadrature weights on the device, batched over kernel-parameter rows (theta): one 256-thread workgroup per theta.
//
//   RBF kernel matrix over the unit sigma points        bq/bqkern.py:329-343 + utils.maha utils.py:385-409
//   (K + jitter I)^-1 by a right-looking Cholesky and two triangular solves against I, then symmetrised
//                                                        bq/bqkern.py:38-64, 96-120
//   Gaussian expectations q, R, Q, kbar                  bq/bqkern.py:345-424
//   GP weights  wm = q iK, Wc = iK Q iK, Wcc = R iK      bq/bqmod.py:495-523
//   Bayes-Sard weights (unisolvent + general branch)     bq/bqmod.py:893-992, polynomial moments :635-797
//
// For N <= 64 the kernel matrix, its factor and the inverse are LDS-resident (2 N^2 doubles <= 64 KiB) and one launch
// does everything.  Larger point sets (fully-symmetric degree 5 at D = 10: N = 201) take three launches per batch of
// parameter rows ("large point sets" below):
//   1. k_weights<1024>, stage 1: K, then its Cholesky factor in the PACKED lower triangle (N (N + 1) / 2 doubles =
//      162 408 bytes at N = 201) held in the CU's 160 KiB of LDS; the factor goes to the workspace;
//   2. k_weights_inverse: the two triangular solves against I are independent per column, so the columns are dealt out
//      to ceil(N / 16) workgroups per parameter row - each re-reads the packed factor into its LDS and keeps its 16
//      solution columns in registers (one CU alone is instruction-bound on the 2 N^3 / 3 multiply-adds);
//   3. k_weights<1024>, stage 2: expectations and the weight algebra, every N x N x N product through LDS tiles with
//      4 x 4 register blocks and operands fetched one slab ahead.
// Beyond N = 201 (the degree-7 rule at D = 10: 1 181 points) the factor no longer fits a CU: blocked Cholesky, inverse and the two
// N^3 products as many-workgroup launches (k_wb_*, further down), the rest in stages 3 / 4 of k_weights<1024>.
// The algebra follows the reference step by step (explicit inverse, symmetrisation, jitter placement) because the
// results are only reproducible to cond(K) eps, not because it is the best-conditioned route.
#include <algorithm>
#include <cstring>
#include <vector>
#include "ssmq_host.h"
#include "ssmq_wide.h"

namespace ssmq {

// workgroup size: 256 threads per parameter row, 1024 for large point sets (N > 64: one row keeps a whole CU busy with
// L2-latency-bound loops, so it takes all the waves the CU can give); device code strides by the actual block size
#define kWgtBlock ((int)blockDim.x)

struct WgtArgs {
    int32_t D, N, P, NB, bs, use_lds;
    int32_t stage;      // 0: everything in one launch; 1: K + factor only (large point sets); 2: from the inverse onwards
    int32_t tiled;      // 1: the products go through LDS tiles (gemm_tiled) although K and its inverse live in global memory
    int32_t var_mode;   // 1: BayesSardModel.exp_model_variance / integral_variance semantics (bq/bqmod.py:995-1050): always
                        // the general formulas, no jitter on V' iK V; weights are still written but are not the reference's
    double jitter;
    const double *xi;       // [D][N]
    const double *par;      // [P][1+D]
    const int32_t *mulind;  // [D][NB]
    const double *px, *xpx, *pxpx;  // [NB], [D][NB], [NB][NB]
    double *wm, *Wc, *Wcc, *iK, *q, *Q, *R, *mv, *iv;
    int32_t *status;
    double *work;           // per-theta workspace
    int64_t work_stride;    // doubles
    double *lpack;          // stages 1 / 2: packed Cholesky factors, N (N + 1) / 2 doubles per parameter row
};

__device__ __forceinline__ void bsync() { __syncthreads(); }

// offsets (doubles) of the blocks of one parameter row's workspace; the launcher of the large-N route needs M1, M2 and T4 too
struct WgtCarve {
    int64_t gA, gX, M1, M2, zs, nrm, V, Z, G, iG, kx, T1, T2, bv, Dm, T3, T4, end;
    __host__ __device__ WgtCarve(int64_t D, int64_t N, int64_t NB) {
        int64_t w = 0;
        const int64_t nn = N * N;
        gA = w; w += nn;
        gX = w; w += nn;
        M1 = w; w += nn;
        M2 = w; w += nn;
        zs = w; w += D * N;       // length-scale-normalised points
        nrm = w; w += N;
        V = w; w += N * NB;       // Vandermonde (N x NB)
        Z = w; w += NB * N;       // V' iK
        G = w; w += NB * NB;      // V' iK V + 1e-8 I  -> Cholesky factor
        iG = w; w += NB * NB;     // its inverse ("iViKV")
        kx = w; w += N * NB;      // E[k(x, x_n) p_q(x)]
        T1 = w; w += nn;
        T2 = w; w += nn;
        bv = w; w += NB;
        Dm = w; w += D * NB;      // D = R Z' - xpx       (D x NB)
        T3 = w; w += nn;
        T4 = w; w += nn;
        end = w;
    }
};

// C (M x N, ldc) = op(A) op(B); op(A) is M x K, op(B) is K x N.  Block-cooperative; ends with a barrier.
__device__ void gemm(double *C, int ldc, const double *A, int lda, bool ta, const double *B, int ldb, bool tb, int M,
                     int N, int K) {
    for (int idx = threadIdx.x; idx < M * N; idx += kWgtBlock) {
        const int i = idx / N, j = idx % N;
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double a = ta ? A[k * lda + i] : A[i * lda + k];
            const double b = tb ? B[j * ldb + k] : B[k * ldb + j];
            s += a * b;
        }
        C[i * ldc + j] = s;
    }
    bsync();
}

// In-place right-looking Cholesky (lower) of the n x n matrix A; the strict upper triangle is left untouched.
// Returns false (to every thread) at the first non-positive pivot.
__device__ bool chol_block(double *A, int n, int *flag) {
    if (threadIdx.x == 0) *flag = 1;
    bsync();
    for (int k = 0; k < n; ++k) {
        if (threadIdx.x == 0) {
            const double p = A[k * n + k];
            if (!(p > 0.0)) *flag = 0;
            A[k * n + k] = sqrt(p);
        }
        bsync();
        if (*flag == 0) return false;
        const double r = 1.0 / A[k * n + k];
        for (int i = k + 1 + threadIdx.x; i < n; i += kWgtBlock) A[i * n + k] *= r;
        bsync();
        const int m = n - k - 1;
        for (int idx = threadIdx.x; idx < m * m; idx += kWgtBlock) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) A[i * n + j] -= A[i * n + k] * A[j * n + k];
        }
        bsync();
    }
    return true;
}

// X = (L L')^-1 for the lower factor L (n x n): each thread owns columns of X; forward then backward substitution.
__device__ void chol_inverse(const double *L, double *X, int n) {
    for (int c = threadIdx.x; c < n; c += kWgtBlock) {
        for (int i = 0; i < n; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= L[i * n + k] * X[k * n + c];   // X[k][c] = 0 for k < c
            X[i * n + c] = (i < c) ? 0.0 : s / L[i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = X[i * n + c];
            for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * X[k * n + c];
            X[i * n + c] = s / L[i * n + i];
        }
    }
    bsync();
}

// X = A^-1 for a general n x n matrix by LU with partial pivoting (numpy.linalg.solve(V, I), bq/bqmod.py:954).
// A is destroyed.  Single-thread pivot search, block-parallel elimination; n is small (<= N).
__device__ bool lu_inverse(double *A, double *X, int n, int *piv, int *flag) {
    for (int idx = threadIdx.x; idx < n * n; idx += kWgtBlock) X[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
    if (threadIdx.x == 0) *flag = 1;
    bsync();
    for (int k = 0; k < n; ++k) {
        if (threadIdx.x == 0) {
            int p = k;
            double best = fabs(A[k * n + k]);
            for (int i = k + 1; i < n; ++i)
                if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
            *piv = p;
            if (best == 0.0) *flag = 0;
        }
        bsync();
        if (*flag == 0) return false;
        const int p = *piv;
        if (p != k) {
            for (int j = threadIdx.x; j < n; j += kWgtBlock) {
                double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t;
                t = X[k * n + j]; X[k * n + j] = X[p * n + j]; X[p * n + j] = t;
            }
        }
        bsync();
        const double r = 1.0 / A[k * n + k];
        for (int i = k + 1 + threadIdx.x; i < n; i += kWgtBlock) A[i * n + k] *= r;
        bsync();
        const int m = n - k - 1;
        for (int idx = threadIdx.x; idx < m * n; idx += kWgtBlock) {
            const int i = k + 1 + idx / n, j = idx % n;
            const double l = A[i * n + k];
            if (j > k) A[i * n + j] -= l * A[k * n + j];
            X[i * n + j] -= l * X[k * n + j];
        }
        bsync();
    }
    // back substitution U X = Y, thread per column
    for (int c = threadIdx.x; c < n; c += kWgtBlock) {
        for (int i = n - 1; i >= 0; --i) {
            double s = X[i * n + c];
            for (int k = i + 1; k < n; ++k) s -= A[i * n + k] * X[k * n + c];
            X[i * n + c] = s / A[i * n + i];
        }
    }
    bsync();
    return true;
}

// ---- large point sets (N > 64, 1024 threads) --------------------------------------------------------------------------
#define SSMQ_PKL(i, j) ((i) * ((i) + 1) / 2 + (j))   // packed lower triangle, j <= i

// Right-looking Cholesky of a packed lower triangle held in LDS (same subtraction order as chol_block: same factor).
// Two barriers per column: every thread forms the pivot's reciprocal root itself, and the trailing update walks rows and
// columns without an integer division per element.
__device__ bool chol_packed_lds(double *Lp, int n, int *flag) {
    if (threadIdx.x == 0) *flag = 1;
    bsync();
    for (int k = 0; k < n; ++k) {
        const double p = Lp[SSMQ_PKL(k, k)];
        if (!(p > 0.0)) return false;            // uniform: every thread reads the same pivot
        const double lkk = sqrt(p), r = 1.0 / lkk;
        bsync();                                 // everyone has read the pivot before it is overwritten
        if (threadIdx.x == 0) Lp[SSMQ_PKL(k, k)] = lkk;
        for (int i = k + 1 + threadIdx.x; i < n; i += kWgtBlock) Lp[SSMQ_PKL(i, k)] *= r;
        bsync();
        for (int i = k + 1 + (threadIdx.x >> 4); i < n; i += kWgtBlock >> 4) {
            const double lik = Lp[SSMQ_PKL(i, k)];
            for (int j = k + 1 + (threadIdx.x & 15); j <= i; j += 16) Lp[SSMQ_PKL(i, j)] -= lik * Lp[SSMQ_PKL(j, k)];
        }
        bsync();
    }
    return true;
}

// X = (L L')^-1 column by column - cho_solve(cho_factor(A), I): forward substitution L y = e_c, backward substitution
// L' x = y.  Sixteen lanes (a DPP row) own a column and keep it IN REGISTERS (lane `part` holds rows part, part + 16, ...);
// the factor is only read (packed, in LDS), so columns never synchronise with each other.  The row index of a register
// slot is static; the one dynamic write per step (row i of the solution) goes through a switch.  n <= 16 kInvSlots.
constexpr int kInvSlots = 16, kInvLanes = 16, kInvBlock = 256, kInvCols = kInvBlock / kInvLanes;
__device__ __forceinline__ void slot_write(double (&x)[kInvSlots], int slot, double v) {
    switch (slot) {
#define SSMQ_SW(q) case q: x[q] = v; break;
#define SSMQ_SW4(q) SSMQ_SW(q) SSMQ_SW(q + 1) SSMQ_SW(q + 2) SSMQ_SW(q + 3)
        SSMQ_SW4(0) SSMQ_SW4(4) SSMQ_SW4(8) SSMQ_SW4(12)
#undef SSMQ_SW4
#undef SSMQ_SW
        default: break;
    }
}
// sum over the 16 lanes of a group (a DPP row), to every lane: two quad permutes, the mirrored half, the mirrored row
// (DPP: no LDS round trip)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_f64<0xB1>(v);     // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E>(v);     // quad_perm [2, 3, 0, 1]: every lane of a quad holds the quad's sum
    v += dpp_f64<0x141>(v);    // row_half_mirror: lane i <-> 7 - i, the other quad of the half row
    v += dpp_f64<0x140>(v);    // row_mirror: lane i <-> 15 - i, the other half
    return v;
}
// grid (ceil(n / kInvCols), P): workgroup (cb, p) solves columns cb kInvCols ... of parameter row p
__global__ __launch_bounds__(kInvBlock) void k_weights_inverse(int n, const double *__restrict__ lpack, const int32_t *status,
                                                              double *__restrict__ work, int64_t work_stride) {
    extern __shared__ __align__(16) double Lp[];
    const int p = blockIdx.y;
    if (status[p] != 0) return;                  // not positive definite: stage 2 poisons the outputs
    const int np = n * (n + 1) / 2;
    const double *src = lpack + (int64_t)p * np;
    for (int idx = threadIdx.x; idx < np; idx += kInvBlock) Lp[idx] = src[idx];
    __syncthreads();
    double *X = work + (int64_t)p * work_stride + (int64_t)n * n;     // gX of k_weights' workspace carve-up
    const int part = threadIdx.x & (kInvLanes - 1);
    const int c = blockIdx.x * kInvCols + threadIdx.x / kInvLanes;
    const bool live = c < n;                     // whole lane groups are live or not
    double x[kInvSlots];
#pragma unroll
    for (int q = 0; q < kInvSlots; ++q) x[q] = 0.0;
    // forward: y_i = (e_c[i] - sum_{k < i} L[i][k] y_k) / L[i][i]; y_k = 0 for k < c
    for (int i = 0; i < n; ++i) {
        // all loads of the row first (clamped index, the factor of an out-of-range term is zero), four partial sums
        double sp[4] = {0.0, 0.0, 0.0, 0.0};
        const int rowi = SSMQ_PKL(i, 0);
#pragma unroll
        for (int q = 0; q < kInvSlots; ++q) {
            const int k = kInvLanes * q + part;
            if (kInvLanes * q < i) {             // uniform: a whole slot beyond row i contributes nothing
                const double l = Lp[rowi + (k < i ? k : i)];
                sp[q & 3] += (k < i ? l : 0.0) * x[q];
            }
        }
        const double s = row_sum((sp[0] + sp[1]) + (sp[2] + sp[3]));
        const double y = (live && i >= c) ? div_nr((i == c ? 1.0 : 0.0) - s, Lp[SSMQ_PKL(i, i)]) : 0.0;
        if ((i & (kInvLanes - 1)) == part) slot_write(x, i / kInvLanes, y);
    }
    // backward: x_i = (y_i - sum_{k > i} L[k][i] x_k) / L[i][i]
    for (int i = n - 1; i >= 0; --i) {
        double sp[4] = {0.0, 0.0, 0.0, 0.0}, yi = 0.0;
#pragma unroll
        for (int q = 0; q < kInvSlots; ++q) {
            const int k = kInvLanes * q + part;
            if (kInvLanes * q + kInvLanes - 1 >= i && kInvLanes * q < n) {   // uniform: slots entirely above row i are done
                const bool in = k > i && k < n;
                const int kc = in ? k : i;
                const double l = Lp[SSMQ_PKL(kc, i)];
                sp[q & 3] += (in ? l : 0.0) * x[q];
                yi = (k == i) ? x[q] : yi;
            }
        }
        const double s = row_sum((sp[0] + sp[1]) + (sp[2] + sp[3]));
        yi = row_sum((i & (kInvLanes - 1)) == part ? yi : 0.0);   // y_i sits in exactly one lane of the group
        const double v = live ? div_nr(yi - s, Lp[SSMQ_PKL(i, i)]) : 0.0;
        if ((i & (kInvLanes - 1)) == part) slot_write(x, i / kInvLanes, v);
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < kInvSlots; ++q) {
            const int k = kInvLanes * q + part;
            if (k < n) X[k * n + c] = x[q];
        }
    }
}

// C (M x N, ldc) = op(A) op(B) through LDS tiles: 128 x 128 outputs per pass, 4 x 4 per thread (1024 threads), K in
// slabs of 16 whose loads are issued one slab ahead (registers) so that the L2 latency hides behind the arithmetic of
// the current slab; k ascends inside every output's sum exactly as in gemm(), so the result is bit-identical to it.
// tile: 2 * 16 * 132 doubles of LDS.
constexpr int kTileMN = 128, kTileK = 16, kTilePitch = kTileMN + 4;
__device__ void gemm_tiled(double *tile, double *C, int ldc, const double *A, int lda, bool ta, const double *B, int ldb,
                           bool tb, int M, int N, int K) {
    double *sA = tile, *sB = tile + kTileK * kTilePitch;     // sA[k][i], sB[k][j]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 32 threads, each 4 x 4 outputs
    constexpr int kPer = kTileK * kTileMN / 1024;            // elements of either slab per thread (2)
    for (int i0 = 0; i0 < M; i0 += kTileMN)
        for (int j0 = 0; j0 < N; j0 += kTileMN) {
            double acc[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[r][q] = 0.0;
            double ra[kPer], rb[kPer];
            auto fetch = [&](int k0) {
#pragma unroll
                for (int u = 0; u < kPer; ++u) {
                    const int idx = threadIdx.x + u * kWgtBlock;
                    int kk, ii;                  // coalesced along the contiguous direction of each operand
                    if (ta) { kk = idx / kTileMN; ii = idx % kTileMN; } else { ii = idx / kTileK; kk = idx % kTileK; }
                    const int gi = i0 + ii, gk = k0 + kk;
                    ra[u] = (gi < M && gk < K) ? (ta ? A[gk * lda + gi] : A[gi * lda + gk]) : 0.0;
                    int kb, jj;
                    if (tb) { jj = idx / kTileK; kb = idx % kTileK; } else { kb = idx / kTileMN; jj = idx % kTileMN; }
                    const int gj = j0 + jj, gkb = k0 + kb;
                    rb[u] = (gj < N && gkb < K) ? (tb ? B[gj * ldb + gkb] : B[gkb * ldb + gj]) : 0.0;
                }
            };
            auto park = [&]() {
#pragma unroll
                for (int u = 0; u < kPer; ++u) {
                    const int idx = threadIdx.x + u * kWgtBlock;
                    int kk, ii;
                    if (ta) { kk = idx / kTileMN; ii = idx % kTileMN; } else { ii = idx / kTileK; kk = idx % kTileK; }
                    sA[kk * kTilePitch + ii] = ra[u];
                    int kb, jj;
                    if (tb) { jj = idx / kTileK; kb = idx % kTileK; } else { kb = idx / kTileMN; jj = idx % kTileMN; }
                    sB[kb * kTilePitch + jj] = rb[u];
                }
            };
            fetch(0);
            for (int k0 = 0; k0 < K; k0 += kTileK) {
                bsync();                         // the previous slab has been consumed
                park();
                bsync();
                if (k0 + kTileK < K) fetch(k0 + kTileK);
#pragma unroll
                for (int kk = 0; kk < kTileK; ++kk) {
                    double av[4], bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[r] = sA[kk * kTilePitch + ty * 4 + r];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[q] = sB[kk * kTilePitch + tx * 4 + q];
                    if (k0 + kk < K) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[r][q] += av[r] * bv[q];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gi = i0 + ty * 4 + r, gj = j0 + tx * 4 + q;
                    if (gi < M && gj < N) C[gi * ldc + gj] = acc[r][q];
                }
        }
    bsync();
}

// ---- point sets beyond the CU-resident route (N > 201; the degree-7 rule at D = 10 has 1 181 points) ---------------------
// One workgroup per parameter row spent 0.98 s on N = 1 181 (factor, inverse and products of one row on ONE CU out of 256).
// The three O(N^3) parts go to many workgroups, one launch per dependency level:
//   k_wb_kmatrix                       K + jitter I, element per thread
//   k_wb_chol_panel / k_wb_chol_update right-looking Cholesky in 64-column blocks: every workgroup of a panel launch factors the
//                                      64 x 64 diagonal block itself (LDS), workgroup 0 stores it, the others solve 64 rows of the
//                                      panel each; the update launch subtracts the panel's outer product from the trailing
//                                      lower triangle, one 64 x 64 tile per workgroup.  Every element receives its updates one k
//                                      at a time in ascending k, as chol_block applies them: the factor is the same, bit for bit.
//                                      The transposed factor is written into the (otherwise unused) upper triangle.
//   k_wb_inverse                       cho_solve against I: a WAVE owns CPW columns in registers (slot q of lane l = row 64 q + l),
//                                      rows of L for the forward pass, rows of L' (the upper triangle) for the backward pass
//   k_wb_gemm                          C = A B, one 64 x 64 tile per workgroup, the sums in gemm_tiled's order
// The remaining algebra (all of it O(N^2 NB) or less) stays in k_weights<1024>, split at the two large products (stage 3 / 4).
constexpr int kCb = 64, kCbPitch = kCb + 1;
constexpr size_t kCbLds = sizeof(double) * (2 * kCb * kCbPitch + kCb);

__global__ __launch_bounds__(256) void k_wb_kmatrix(const WgtArgs a) {
    const int p = blockIdx.y, N = a.N, D = a.D;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.status[p] = 0;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)N * N) return;
    const double *par = a.par + (int64_t)p * (1 + D);
    const int i = (int)(idx / N), j = (int)(idx % N);
    double ni = 0.0, nj = 0.0, dot = 0.0;
    for (int d = 0; d < D; ++d) {
        const double sl = 1.0 / par[1 + d];
        const double zi = sl * a.xi[d * N + i], zj = sl * a.xi[d * N + j];
        ni += zi * zi;
        nj += zj * zj;
        dot += zi * zj;
    }
    const double mh = (ni + nj) - 2.0 * dot;
    a.work[(int64_t)p * a.work_stride + idx] = exp(0.0 - 0.5 * mh) + (i == j ? a.jitter : 0.0);
}

// grid (row chunks from the diagonal block downwards, P)
__global__ __launch_bounds__(256) void k_wb_chol_panel(int N, int kb, double *__restrict__ work, int64_t work_stride,
                                                       int32_t *status) {
    extern __shared__ __align__(16) double sm[];
    double *sD = sm, *sR = sm + kCb * kCbPitch, *sInv = sR + kCb * kCbPitch;
    const int p = blockIdx.y, tid = threadIdx.x;
    if (status[p] != 0) return;
    double *A = work + (int64_t)p * work_stride;
    const int c0 = kb * kCb, bs = min(kCb, N - c0);
    for (int idx = tid; idx < bs * bs; idx += 256) {
        const int i = idx / bs, j = idx % bs;
        sD[i * kCbPitch + j] = (j <= i) ? A[(int64_t)(c0 + i) * N + c0 + j] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < bs; ++k) {
        const double pv = sD[k * kCbPitch + k];
        if (!(pv > 0.0)) {                       // uniform: every thread of every workgroup of the launch reads the same pivot
            if (blockIdx.x == 0 && tid == 0) status[p] = 1;
            return;
        }
        const double lkk = sqrt(pv), r = 1.0 / lkk;
        __syncthreads();
        if (tid == 0) { sD[k * kCbPitch + k] = lkk; sInv[k] = r; }
        for (int i = k + 1 + tid; i < bs; i += 256) sD[i * kCbPitch + k] *= r;
        __syncthreads();
        const int m = bs - k - 1;
        for (int idx = tid; idx < m * m; idx += 256) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) sD[i * kCbPitch + j] -= sD[i * kCbPitch + k] * sD[j * kCbPitch + k];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int idx = tid; idx < bs * bs; idx += 256) {
            const int i = idx / bs, j = idx % bs;
            if (j <= i) {
                const double v = sD[i * kCbPitch + j];
                A[(int64_t)(c0 + i) * N + c0 + j] = v;
                if (j < i) A[(int64_t)(c0 + j) * N + c0 + i] = v;
            }
        }
        return;
    }
    // 64 rows of the panel below the diagonal block (bs == 64 whenever there is such a row)
    const int r0 = c0 + kCb * blockIdx.x, nr = min(kCb, N - r0);
    for (int idx = tid; idx < nr * kCb; idx += 256) {
        const int t = idx / kCb, k = idx % kCb;
        sR[t * kCbPitch + k] = A[(int64_t)(r0 + t) * N + c0 + k];
    }
    __syncthreads();
    if (tid < nr) {
        double *row = sR + tid * kCbPitch;
        for (int k = 0; k < kCb; ++k) {
            const double l = row[k] * sInv[k];
            row[k] = l;
            for (int j = k + 1; j < kCb; ++j) row[j] -= l * sD[j * kCbPitch + k];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < nr * kCb; idx += 256) {
        const int t = idx / kCb, k = idx % kCb;
        A[(int64_t)(r0 + t) * N + c0 + k] = sR[t * kCbPitch + k];
    }
    for (int idx = tid; idx < nr * kCb; idx += 256) {
        const int k = idx / nr, t = idx % nr;
        A[(int64_t)(c0 + k) * N + r0 + t] = sR[t * kCbPitch + k];   // L' into the upper triangle
    }
}

// grid (tile pairs ti >= tj of the trailing triangle, P)
__global__ __launch_bounds__(256) void k_wb_chol_update(int N, int kb, double *__restrict__ work, int64_t work_stride,
                                                        const int32_t *status) {
    extern __shared__ __align__(16) double sm[];
    double *sI = sm, *sJ = sm + kCb * kCbPitch;
    const int p = blockIdx.y, tid = threadIdx.x;
    if (status[p] != 0) return;
    double *A = work + (int64_t)p * work_stride;
    const int c0 = kb * kCb, t0 = c0 + kCb;
    const int bx = blockIdx.x;
    int ti = (int)((sqrt(8.0 * bx + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > bx) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= bx) ++ti;
    const int tj = bx - ti * (ti + 1) / 2;
    const int i0 = t0 + kCb * ti, j0 = t0 + kCb * tj;
    for (int idx = tid; idx < kCb * kCb; idx += 256) {
        const int t = idx / kCb, k = idx % kCb;
        sI[t * kCbPitch + k] = (i0 + t < N) ? A[(int64_t)(i0 + t) * N + c0 + k] : 0.0;
        sJ[t * kCbPitch + k] = (j0 + t < N) ? A[(int64_t)(j0 + t) * N + c0 + k] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = i0 + 4 * ty + r, gj = j0 + 4 * tx + q;
            acc[r][q] = (gi < N && gj <= gi) ? A[(int64_t)gi * N + gj] : 0.0;
        }
    for (int k = 0; k < kCb; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = sI[(4 * ty + r) * kCbPitch + k];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = sJ[(4 * tx + q) * kCbPitch + k];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] -= av[r] * bv[q];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = i0 + 4 * ty + r, gj = j0 + 4 * tx + q;
            if (gi < N && gj <= gi) A[(int64_t)gi * N + gj] = acc[r][q];
        }
}

// sum over the 64 lanes of a wave, to every lane (as a uniform value): the DPP row sums, then one lane of each row
__device__ __forceinline__ double wave_sum(double v) {
    v = row_sum(v);
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), 16 * r);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 16 * r);
        s += __hiloint2double(hi, lo);
    }
    return s;
}
__device__ __forceinline__ double lane_value(double v, int lane) {      // lane: uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// grid (ceil(N / (4 CPW)), P), 256 threads = 4 waves, each with CPW columns of the inverse; N <= 64 S.  The matrix holds L in
// its lower triangle and L' in the upper one; the solution columns go to gX TRANSPOSED (row c of gX = column c of the
// inverse: coalesced; the caller symmetrises 0.5 (X + X'), which does not care).
template <int S, int CPW>
__global__ __launch_bounds__(256) void k_wb_inverse(int N, double *__restrict__ work, int64_t work_stride, const int32_t *status) {
    const int p = blockIdx.y;
    if (status[p] != 0) return;
    const double *A = work + (int64_t)p * work_stride;
    double *Xt = work + (int64_t)p * work_stride + (int64_t)N * N;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cb = (blockIdx.x * 4 + wave) * CPW;
    if (cb >= N) return;
    double x[CPW][S];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int q = 0; q < S; ++q) x[c][q] = 0.0;
    const int qb = cb >> 6;
    // forward: y_i = (e_c[i] - sum_{k < i} L[i][k] y_k) / L[i][i], y_k = 0 for k < c
#pragma unroll
    for (int qi = 0; qi < S; ++qi) {
        if (qi < qb || 64 * qi >= N) continue;
        for (int ii = 0; ii < 64; ++ii) {
            const int i = 64 * qi + ii;
            if (i < cb || i >= N) continue;
            const double *row = A + (int64_t)i * N;
            double sp[CPW];
#pragma unroll
            for (int c = 0; c < CPW; ++c) sp[c] = 0.0;
#pragma unroll
            for (int q = 0; q <= qi; ++q) {
                if (q < qb) continue;
                const bool in = q < qi || lane < ii;
                const double l = in ? row[64 * q + lane] : 0.0;
#pragma unroll
                for (int c = 0; c < CPW; ++c) sp[c] += l * x[c][q];
            }
            const double dii = row[i];
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const double s = wave_sum(sp[c]);
                const double y = (i >= cb + c) ? div_nr((i == cb + c ? 1.0 : 0.0) - s, dii) : 0.0;
                x[c][qi] = (lane == ii) ? y : x[c][qi];
            }
        }
    }
    // backward: x_i = (y_i - sum_{k > i} L[k][i] x_k) / L[i][i];  L[k][i] = A[i][k] for k > i
#pragma unroll
    for (int qi = S - 1; qi >= 0; --qi) {
        if (64 * qi >= N) continue;
        for (int ii = 63; ii >= 0; --ii) {
            const int i = 64 * qi + ii;
            if (i >= N) continue;
            const double *row = A + (int64_t)i * N;
            double sp[CPW];
#pragma unroll
            for (int c = 0; c < CPW; ++c) sp[c] = 0.0;
#pragma unroll
            for (int q = qi; q < S; ++q) {
                if (64 * q >= N) continue;
                const int k = 64 * q + lane;
                const bool in = (q > qi || lane > ii) && k < N;
                const double l = in ? row[k] : 0.0;
#pragma unroll
                for (int c = 0; c < CPW; ++c) sp[c] += l * x[c][q];
            }
            const double dii = row[i];
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const double s = wave_sum(sp[c]);
                const double v = div_nr(lane_value(x[c][qi], ii) - s, dii);
                x[c][qi] = (lane == ii) ? v : x[c][qi];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        if (cb + c >= N) continue;
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const int k = 64 * q + lane;
            if (k < N) Xt[(int64_t)(cb + c) * N + k] = x[c][q];
        }
    }
}

// C (M x N, ldc) = A (M x K, row-major) B (K x N, row-major): grid (ceil(N / 64), ceil(M / 64), P), 256 threads, 4 x 4 outputs per
// thread, K in slabs of 16 fetched one slab ahead; k ascends inside every output's sum exactly as in gemm_tiled.
__global__ __launch_bounds__(256) void k_wb_gemm(double *__restrict__ Cb, int64_t c_stride, int ldc, const double *__restrict__ Ab,
                                                 int64_t a_stride, int lda, const double *__restrict__ Bb, int64_t b_stride, int ldb,
                                                 int M, int N, int K, const int32_t *status) {
    constexpr int TM = 64, TK = 16, TP = TM + 4, PER = TK * TM / 256;
    __shared__ __align__(16) double sA[TK * TP], sB[TK * TP];
    const int p = blockIdx.z, tid = threadIdx.x;
    if (status[p] != 0) return;
    const double *A = Ab + p * a_stride, *B = Bb + p * b_stride;
    double *C = Cb + p * c_stride;
    const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TM;
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = 0.0;
    double ra[PER], rb[PER];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * 256;
            const int ii = idx / TK, kk = idx % TK;
            const int gi = i0 + ii, gk = k0 + kk;
            ra[u] = (gi < M && gk < K) ? A[(int64_t)gi * lda + gk] : 0.0;
            const int kb = idx / TM, jj = idx % TM;
            const int gj = j0 + jj, gkb = k0 + kb;
            rb[u] = (gj < N && gkb < K) ? B[(int64_t)gkb * ldb + gj] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += TK) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * 256;
            sA[(idx % TK) * TP + idx / TK] = ra[u];
            sB[(idx / TM) * TP + idx % TM] = rb[u];
        }
        __syncthreads();
        if (k0 + TK < K) fetch(k0 + TK);
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = sA[kk * TP + ty * 4 + r];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = sB[kk * TP + tx * 4 + q];
            if (k0 + kk < K) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[r][q] += av[r] * bv[q];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = i0 + ty * 4 + r, gj = j0 + tx * 4 + q;
            if (gi < M && gj < N) C[(int64_t)gi * ldc + gj] = acc[r][q];
        }
}

__device__ double block_sum(double v, double *red) {
    // all threads of the block -> one value (to every thread); waves added in index order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    bsync();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    bsync();
    double s = 0.0;
    for (int w = 0; w < kWgtBlock / 64; ++w) s += red[w];
    bsync();
    return s;
}

__device__ double ipow(double x, int k) {
    double r = 1.0;
    for (int i = 0; i < k; ++i) r *= x;
    return r;
}

// Entry (n, q) of the Vandermonde matrix (utils.py:478-502) and of E[k(x, x_n) p_q(x)], the closed form of
// bq/bqmod.py:733-797; its `ell` is sqrt_inv_lam ** -2 = ell^2, reproduced as written there.  sil[d] = 1 / ell_d.
__device__ void bs_basis_entry(int D, int N, int NB, int n, int qb, const double *__restrict__ xi,
                               const int32_t *__restrict__ mulind, const double *sil, double &vand, double &kxpx) {
    double v = 1.0, kprod = 1.0;
    for (int d = 0; d < D; ++d) {
        const int al = mulind[d * NB + qb];
        const double x = xi[d * N + n];
        v *= ipow(x, al);
        const double sl = sil[d];
        const double el = 1.0 / (sl * sl);
        const double e1 = 1.0 + el * el;
        const double ea = el * pow(e1, -(1.0 + al) / 2.0) * exp(-(x * x) / (2.0 * e1));
        double eb = 0.0;
        const double xs = x / sqrt(e1);
        for (int m = 0; m <= al / 2; ++m) {
            // al! / (2^m m! (al - 2m)!)
            double num = 1.0, den = 1.0;
            for (int t = 2; t <= al; ++t) num *= t;
            for (int t = 0; t < m; ++t) den *= 2.0;
            for (int t = 2; t <= m; ++t) den *= t;
            for (int t = 2; t <= al - 2 * m; ++t) den *= t;
            eb += (num / den) * (ipow(el, 2 * m) * ipow(xs, al - 2 * m));
        }
        kprod *= ea * eb;
    }
    vand = v;
    kxpx = kprod;
}

// where one parameter row's workspace and results live: the global arrays of WgtArgs at row p (k_weights), or the
// workgroup's LDS (k_theta_weights, which only keeps the packed constants of the transform kernel)
struct WgtOut {
    double *work, *q, *Q, *R, *iK, *wm, *Wc, *Wcc, *mv, *iv;
    int32_t *status;
};

__device__ void weights_body(const WgtArgs &a, const WgtOut &o, int p, double *lds) {
    __shared__ double s_sil[SSMQ_MAX_DIM], s_red[16];
    __shared__ int s_flag, s_piv;
    const int D = a.D, N = a.N, NB = a.NB, tid = threadIdx.x;
    const double *par = a.par + (int64_t)p * (1 + D);
    const double alpha = par[0];
    double *w = o.work;
    // workspace carve-up (global); A and X move to LDS when they fit
    const WgtCarve cv(D, N, NB);
    double *gA = w + cv.gA, *gX = w + cv.gX, *M1 = w + cv.M1, *M2 = w + cv.M2, *zs = w + cv.zs, *nrm = w + cv.nrm;
    const bool large = a.use_lds == 2;       // N > 64: packed factor in LDS, tiles afterwards (1024 threads)
    double *A = a.use_lds == 1 ? lds : gA;
    double *X = a.use_lds == 1 ? lds + N * N : gX;
    // products: the LDS-tiled routine for large point sets (LDS is free again once the inverse exists), else the plain one
    const bool tiled = large || a.tiled != 0;
#define GEMM(...) do { if (tiled) gemm_tiled(lds, __VA_ARGS__); else gemm(__VA_ARGS__); } while (0)
    double *oq = o.q, *oQ = o.Q, *oR = o.R, *oiK = o.iK, *owm = o.wm, *oWc = o.Wc, *oWcc = o.Wcc;

    if (tid < D) s_sil[tid] = 1.0 / par[1 + tid];   // par[1:] ** -1   (bq/bqkern.py:454)
    bsync();
    // ---- kernel matrix, scaling=False (alpha = 1): exp(2 log(1) - maha / 2) ------------------------------------
    for (int idx = tid; idx < D * N; idx += kWgtBlock) zs[idx] = s_sil[idx / N] * a.xi[idx];
    bsync();
    for (int n = tid; n < N; n += kWgtBlock) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += zs[d * N + n] * zs[d * N + n];
        nrm[n] = s;
    }
    bsync();
    bool pd = true;
    // stages 3 / 4 (point sets beyond the CU-resident route): the factor, the inverse (in gX, transposed - symmetrised below) and
    // the two N x N x N products come from the k_wb_* launches; stage 3 runs up to the products, stage 4 from there
    const bool front = a.stage != 4;
    if (a.stage == 4 && *o.status != 0) return;  // stage 3 has poisoned the outputs (or flagged the polynomial part)
    if (a.stage < 2) {
    for (int idx = tid; idx < N * N; idx += kWgtBlock) {
        const int i = idx / N, j = idx % N;
        if (large && j > i) continue;
        double dot = 0.0;
        for (int d = 0; d < D; ++d) dot += zs[d * N + i] * zs[d * N + j];
        const double mh = (nrm[i] + nrm[j]) - 2.0 * dot;
        const double v = exp(0.0 - 0.5 * mh) + (i == j ? a.jitter : 0.0);
        if (large) lds[SSMQ_PKL(i, j)] = v;
        else A[idx] = v;
    }
    bsync();
    // ---- (K + jitter I)^-1 ----------------------------------------------------------------------------------------
    pd = large ? chol_packed_lds(lds, N, &s_flag) : chol_block(A, N, &s_flag);
    if (tid == 0) *o.status = pd ? 0 : 1;
    if (a.stage == 1) {      // large point sets: the factor goes to the workspace, the inverse is a launch of its own
        if (pd) {
            double *dst = a.lpack + (int64_t)p * (N * (N + 1) / 2);
            for (int idx = tid; idx < N * (N + 1) / 2; idx += kWgtBlock) dst[idx] = lds[idx];
        }
        return;
    }
    } else {
        pd = *o.status == 0;
    }
    if (!pd) {
        const double nan = __builtin_nan("");
        for (int idx = tid; idx < N * N; idx += kWgtBlock) { oiK[idx] = nan; oWc[idx] = nan; oQ[idx] = nan; }
        for (int idx = tid; idx < D * N; idx += kWgtBlock) { oWcc[idx] = nan; oR[idx] = nan; }
        for (int n = tid; n < N; n += kWgtBlock) { owm[n] = nan; oq[n] = nan; }
        if (tid == 0) { *o.mv = nan; *o.iv = nan; }
        return;
    }
    if (a.stage == 0) chol_inverse(A, X, N);     // stages 2 / 3: k_weights_inverse / k_wb_inverse has filled X
    if (front) {
        for (int idx = tid; idx < N * N; idx += kWgtBlock) {
            const int i = idx / N, j = idx % N;
            oiK[idx] = 0.5 * (X[i * N + j] + X[j * N + i]);
        }
    }
    bsync();
    const double *iK = oiK;
    // ---- Gaussian expectations of the kernel ------------------------------------------------------------------------
    double cq = 1.0, cQ = 1.0, ck = 1.0;
    for (int d = 0; d < D; ++d) {
        const double il = s_sil[d] * s_sil[d];
        cq *= il + 1.0;
        cQ *= il + il + 1.0;
        ck *= 2.0 * il + 1.0;
    }
    cq = 1.0 / sqrt(cq);   // det(Lam^-1 + I) ** -0.5
    cQ = 1.0 / sqrt(cQ);   // det(2 Lam^-1 + I) ** -0.5
    const double kbar = alpha * alpha * (1.0 / sqrt(ck));
    if (front) {
    for (int n = tid; n < N; n += kWgtBlock) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            const double il = s_sil[d] * s_sil[d];
            const double lam = 1.0 / il;
            const double x = a.xi[d * N + n];
            s += x * ((1.0 / (lam + 1.0)) * x);
        }
        oq[n] = cq * exp(-0.5 * s);
    }
    bsync();
    for (int idx = tid; idx < D * N; idx += kWgtBlock) {
        const int d = idx / N, n = idx % N;
        const double sl = s_sil[d];
        const double lam = 1.0 / (sl * sl);
        oR[idx] = oq[n] * ((1.0 / (lam + 1.0)) * a.xi[idx]);
    }
    const bool general = (NB == 0) || (NB < N);
    if (general) {
        for (int idx = tid; idx < N * N; idx += kWgtBlock) {
            const int i = idx / N, j = idx % N;
            // xi_i + xi_j + maha(Lam^-1 x_i, -Lam^-1 x_j; (2 Lam^-1 + I)^-1) / 2
            double m2i = 0.0, m2j = 0.0, mij = 0.0;
            for (int d = 0; d < D; ++d) {
                const double il = s_sil[d] * s_sil[d];
                const double v = 1.0 / (il + il + 1.0);
                const double yi = il * a.xi[d * N + i], yj = -(il * a.xi[d * N + j]);
                m2i += (yi * v) * yi;
                m2j += (yj * v) * yj;
                mij += (yi * v) * yj;
            }
            const double mh = (m2i + m2j) - 2.0 * mij;
            const double e = ((0.0 - 0.5 * nrm[i]) + (0.0 - 0.5 * nrm[j])) + 0.5 * mh;
            oQ[idx] = cQ * exp(e);
        }
    }
    }   // front
    bsync();

    if (NB == 0) {
        // ---- GP weights (bq/bqmod.py:495-523) -------------------------------------------------------------------------
        if (front) {
            GEMM(owm, N, oq, N, false, iK, N, false, 1, N, N);          // wm = q iK
            GEMM(oWcc, N, oR, N, false, iK, N, false, D, N, N);         // Wcc = R iK
        }
        if (a.stage == 3) return;                                       // M1, M2: two k_wb_gemm launches
        if (a.stage != 4) {
            GEMM(M1, N, oQ, N, false, iK, N, false, N, N, N);           // M1 = Q iK
            GEMM(M2, N, iK, N, false, M1, N, false, N, N, N);           // M2 = iK Q iK
        }
        for (int idx = tid; idx < N * N; idx += kWgtBlock) {
            const int i = idx / N, j = idx % N;
            oWc[idx] = 0.5 * (M2[i * N + j] + M2[j * N + i]);
        }
        double tr = 0.0, qq = 0.0;
        for (int n = tid; n < N; n += kWgtBlock) { tr += M1[n * N + n]; qq += owm[n] * oq[n]; }
        tr = block_sum(tr, s_red);
        qq = block_sum(qq, s_red);
        if (tid == 0) {
            *o.mv = (alpha * alpha) * (1.0 - tr);
            *o.iv = kbar - qq;
        }
            return;
    }

    // ---- Bayes-Sard weights (bq/bqmod.py:893-992) -----------------------------------------------------------------------
    double *V = w + cv.V, *Z = w + cv.Z, *G = w + cv.G, *iG = w + cv.iG, *kx = w + cv.kx, *T1 = w + cv.T1, *T2 = w + cv.T2,
           *bv = w + cv.bv;
    if (front) {
    for (int idx = tid; idx < N * NB; idx += kWgtBlock)
        bs_basis_entry(D, N, NB, idx / NB, idx % NB, a.xi, a.mulind, s_sil, V[idx], kx[idx]);
    bsync();
    GEMM(Z, N, V, NB, true, iK, N, false, NB, N, N);        // Z = V' iK
    GEMM(G, NB, Z, N, false, V, NB, false, NB, NB, N);      // G = Z V
    if (!a.var_mode)
        for (int i = tid; i < NB; i += kWgtBlock) G[i * NB + i] += 1e-8;
    bsync();
    const bool pd2 = chol_block(G, NB, &s_flag);
    if (!pd2) {
        if (tid == 0) *o.status = 2;
        return;
    }
    chol_inverse(G, iG, NB);                                 // cho_solve(cho_factor(.), I): not symmetrised
    }   // front
    const double ks2 = alpha * alpha;
    if (NB == N && !a.var_mode) {
        // pi-unisolvent points: weights from the inverse Vandermonde matrix only (:952-961)
        double *Vc = T1, *iV = T2;
        for (int idx = tid; idx < N * N; idx += kWgtBlock) Vc[idx] = V[idx];
        bsync();
        if (!lu_inverse(Vc, iV, N, &s_piv, &s_flag)) {
            if (tid == 0) *o.status = 3;
            return;
        }
        GEMM(owm, N, a.px, NB, false, iV, N, false, 1, N, NB);              // wm = iV' px  == px' iV
        GEMM(M1, N, a.pxpx, NB, false, iV, N, false, NB, N, NB);            // pxpx iV
        GEMM(M2, N, iV, N, true, M1, N, false, N, N, NB);                   // iV' pxpx iV
        GEMM(oWcc, N, a.xpx, NB, false, iV, N, false, D, N, NB);            // xpx iV
        for (int idx = tid; idx < N * N; idx += kWgtBlock) {
            const int i = idx / N, j = idx % N;
            oWc[idx] = 0.5 * (M2[i * N + j] + M2[j * N + i]);
        }
        // model_var = ks2 (1 - tr(kxpx' iV' + kxpx iV - pxpx iViKV));  tr(kxpx' iV') = tr(iV kxpx) = tr(kxpx iV)
        GEMM(M1, N, kx, NB, false, iV, N, false, N, N, NB);                 // kxpx iV   (N x N)
        GEMM(M2, NB, a.pxpx, NB, false, iG, NB, false, NB, NB, NB);         // pxpx iViKV
        double tr = 0.0;
        for (int n = tid; n < N; n += kWgtBlock) tr += 2.0 * M1[n * N + n] - M2[n * NB + n];
        tr = block_sum(tr, s_red);
        // integral_var = kbar - q' iV' px - px' iV q + px' iViKV px
        double t1 = 0.0, t3 = 0.0;
        for (int n = tid; n < N; n += kWgtBlock) {
            t1 += owm[n] * oq[n];                                           // (px' iV) q
            double s = 0.0;
            for (int k = 0; k < NB; ++k) s += iG[n * NB + k] * a.px[k];
            t3 += a.px[n] * s;
        }
        t1 = block_sum(t1, s_red);
        t3 = block_sum(t3, s_red);
        if (tid == 0) {
            *o.mv = ks2 * (1.0 - tr);
            *o.iv = kbar - t1 - t1 + t3;
        }
        return;
    }
    // general case NB < N (:963-982)
    double *Am = T1;                     // A = V iViKV          (N x NB)
    double *Bm = T2;                     // B                    (NB x NB)
    double *Dm = w + cv.Dm, *T3 = w + cv.T3, *T4 = w + cv.T4;
    if (front) {
    GEMM(Am, NB, V, NB, false, iG, NB, false, N, NB, NB);
    // b = Z q - px
    for (int i = tid; i < NB; i += kWgtBlock) {
        double s = 0.0;
        for (int n = 0; n < N; ++n) s += Z[i * N + n] * oq[n];
        bv[i] = s - a.px[i];
    }
    bsync();
    // B = Z Q Z' + pxpx - Z kxpx - kxpx' Z'
    GEMM(T3, N, Z, N, false, oQ, N, false, NB, N, N);                       // Z Q         (NB x N)
    GEMM(T4, NB, T3, N, false, Z, N, true, NB, NB, N);                      // Z Q Z'      (NB x NB)
    GEMM(T3, NB, Z, N, false, kx, NB, false, NB, NB, N);                    // Z kxpx      (NB x NB)
    for (int idx = tid; idx < NB * NB; idx += kWgtBlock) {
        const int i = idx / NB, j = idx % NB;
        Bm[idx] = ((T4[idx] + a.pxpx[idx]) - T3[i * NB + j]) - T3[j * NB + i];
    }
    bsync();
    // D = R Z' - xpx
    GEMM(Dm, NB, oR, N, false, Z, N, true, D, NB, N);
    for (int idx = tid; idx < D * NB; idx += kWgtBlock) Dm[idx] -= a.xpx[idx];
    bsync();
    // wm = iK (q - A b)
    for (int n = tid; n < N; n += kWgtBlock) {
        double s = 0.0;
        for (int k = 0; k < NB; ++k) s += Am[n * NB + k] * bv[k];
        nrm[n] = oq[n] - s;
    }
    bsync();
    GEMM(owm, 1, iK, N, false, nrm, 1, false, N, 1, N);
    // Wc = iK (Q - A B A') iK
    GEMM(T3, NB, Am, NB, false, Bm, NB, false, N, NB, NB);                  // A B         (N x NB)
    GEMM(T4, N, T3, NB, false, Am, NB, true, N, N, NB);                     // A B A'      (N x N)
    for (int idx = tid; idx < N * N; idx += kWgtBlock) T4[idx] = oQ[idx] - T4[idx];
    bsync();
    }   // front
    if (a.stage == 3) return;                                // M1 = T4 iK, M2 = iK M1: two k_wb_gemm launches
    if (a.stage != 4) {
        GEMM(M1, N, T4, N, false, iK, N, false, N, N, N);
        GEMM(M2, N, iK, N, false, M1, N, false, N, N, N);
    }
    for (int idx = tid; idx < N * N; idx += kWgtBlock) {
        const int i = idx / N, j = idx % N;
        oWc[idx] = 0.5 * (M2[i * N + j] + M2[j * N + i]);
    }
    // Wcc = (R - D A') iK
    GEMM(T3, N, Dm, NB, false, Am, NB, true, D, N, NB);                     // D A'        (D x N)
    for (int idx = tid; idx < D * N; idx += kWgtBlock) T3[idx] = oR[idx] - T3[idx];
    bsync();
    GEMM(oWcc, N, T3, N, false, iK, N, false, D, N, N);
    // model_var = ks2 (1 - tr(Q iK) + tr(B iViKV)); integral_var = kbar - q' iK q + b' iViKV b
    // tr(Q iK): only the diagonal of the product is needed - the same sums, in the same order, as the full N x N x N product
    // this used to take (iK is symmetric bit for bit, so its row n serves as its column n)
    GEMM(T3, NB, Bm, NB, false, iG, NB, false, NB, NB, NB);
    double tr1 = 0.0, tr2 = 0.0, qq = 0.0, bb = 0.0;
    for (int n = tid; n < N; n += kWgtBlock) {
        double d1 = 0.0;
        for (int k = 0; k < N; ++k) d1 += oQ[n * N + k] * iK[n * N + k];
        tr1 += d1;
        double s = 0.0;
        for (int k = 0; k < N; ++k) s += iK[n * N + k] * oq[k];
        qq += oq[n] * s;
    }
    for (int i = tid; i < NB; i += kWgtBlock) {
        tr2 += T3[i * NB + i];
        double s = 0.0;
        for (int k = 0; k < NB; ++k) s += iG[i * NB + k] * bv[k];
        bb += bv[i] * s;
    }
    tr1 = block_sum(tr1, s_red);
    tr2 = block_sum(tr2, s_red);
    qq = block_sum(qq, s_red);
    bb = block_sum(bb, s_red);
    if (tid == 0) {
        *o.mv = ks2 * (1.0 - tr1 + tr2);
        *o.iv = kbar - qq + bb;
    }
#undef GEMM
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_weights(const WgtArgs a) {
    extern __shared__ __align__(16) double lds[];
    const int64_t p = blockIdx.x, N = a.N, D = a.D;
    const WgtOut o{a.work + p * a.work_stride, a.q + p * N, a.Q + p * N * N, a.R + p * D * N, a.iK + p * N * N, a.wm + p * N,
                   a.Wc + p * N * N, a.Wcc + p * D * N, a.mv + p, a.iv + p, a.status + p};
    weights_body(a, o, (int)p, lds);
}

// ---- host side: polynomial moments under N(0, I) (integer tables; bq/bqmod.py:635-731) -----------------------------
static double dfact(int n) {  // (-1)!! = 0!! = 1  
    double r = 1.0;
    for (; n > 1; n -= 2) r *= n;
    return r;
}

static void poly_moments(int D, int NB, const int32_t *mi, std::vector<double> &px, std::vector<double> &xpx,
                         std::vector<double> &pxpx) {
    px.assign(NB, 0.0);
    xpx.assign((size_t)D * NB, 0.0);
    pxpx.assign((size_t)NB * NB, 0.0);
    for (int q = 0; q < NB; ++q) {
        bool even = true;
        for (int d = 0; d < D; ++d) even = even && (mi[d * NB + q] % 2 == 0);
        if (even) {
            double v = 1.0;
            for (int d = 0; d < D; ++d) v *= dfact(mi[d * NB + q] - 1);
            px[q] = v;
        }
        for (int e = 0; e < D; ++e) {
            bool rest_even = true;
            for (int d = 0; d < D; ++d)
                if (d != e) rest_even = rest_even && (mi[d * NB + q] % 2 == 0);
            if ((mi[e * NB + q] + 1) % 2 == 0 && rest_even) {
                double v = mi[e * NB + q];
                for (int d = 0; d < D; ++d)
                    if (d != e) v *= dfact(mi[d * NB + q] - 1);
                xpx[(size_t)e * NB + q] = v;
            }
        }
        for (int r = 0; r < NB; ++r) {
            bool ev = true;
            for (int d = 0; d < D; ++d) ev = ev && ((mi[d * NB + r] + mi[d * NB + q]) % 2 == 0);
            if (ev) {
                double v = 1.0;
                for (int d = 0; d < D; ++d) v *= dfact(mi[d * NB + r] + mi[d * NB + q] - 1);
                pxpx[(size_t)r * NB + q] = v;
            }
        }
    }
}

namespace {
struct DBuf {
    void *p = nullptr;
    ~DBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { return hip_fail(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc"); }
    double *d() { return (double *)p; }
};
}  // namespace

// Launches the weights computation for a.P parameter rows.  N <= 64: one launch, dense K and inverse resident in LDS
// (256 threads).  Larger point sets: packed factor in LDS (stage 1, 1024 threads) | inverse by column blocks on
// ceil(N / 16) workgroups per row | stage 2 (1024 threads, LDS tiles), while the packed triangle fits the CU's LDS
// (N <= 201) and a column fits its lanes' registers; beyond that one launch with everything in the L2-resident workspace.
static int launch_weights(WgtArgs &a, hipStream_t s) {
    const int N = a.N;
    const size_t nn = (size_t)N * N;
    const size_t packed = sizeof(double) * ((size_t)N * (N + 1) / 2);
    const size_t tiles = sizeof(double) * 2 * kTileK * kTilePitch;
    const size_t lds_cap = 160 * 1024 - 512;     // static __shared__ of the kernels: < 512 bytes
    a.stage = 0;
    if (N <= 64) {
        a.use_lds = 1;
        // up to 8 points every loop of the kernel has at most 64 iterations: one wave per parameter row does the same
        // arithmetic (each thread still owns at most one element of every N x N step) with barriers that cost nothing,
        // and four times as many rows are resident per CU - the theta-batched callers launch 1e4 ... 1e5 rows
        const int threads = (N <= 8 && a.NB <= 8 && !ssmq::sw("SSMQ_WEIGHTS_WIDE_BLOCK")) ? 64 : 256;
        hipLaunchKernelGGL(k_weights<256>, dim3(a.P), dim3(threads), sizeof(double) * 2 * nn, s, a);
        return hip_fail(hipGetLastError(), "k_weights");
    }
    static thread_local unsigned attr_epoch = 0;   // per-device attribute: set again after a device change
    if (attr_epoch != ssmq::device_epoch()) {
        SSMQ_HIP(hipFuncSetAttribute((const void *)k_weights<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        SSMQ_HIP(hipFuncSetAttribute((const void *)k_weights_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        attr_epoch = ssmq::device_epoch();
    }
    const bool staged = packed <= lds_cap && N <= kInvLanes * kInvSlots && !ssmq::sw("SSMQ_WEIGHTS_NO_LDS");
    const bool unisolvent = a.NB == N && !a.var_mode;      // N x N LU inverse in one workgroup: not a large-N case in practice
    if (!staged && !unisolvent && !ssmq::sw("SSMQ_WEIGHTS_ONE_WG")) {
        // factor, inverse and the two N^3 products on many workgroups (k_wb_*), the rest in k_weights<1024> stages 3 and 4
        static thread_local unsigned wb_epoch = 0;
        if (wb_epoch != ssmq::device_epoch()) {
            SSMQ_HIP(hipFuncSetAttribute((const void *)k_wb_chol_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCbLds));
            SSMQ_HIP(hipFuncSetAttribute((const void *)k_wb_chol_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCbLds));
            wb_epoch = ssmq::device_epoch();
        }
        a.use_lds = 0;
        a.tiled = 1;
        const int P = a.P, nblk = (N + kCb - 1) / kCb;
        // the workspace carve-up is checked BEFORE the first kernel that writes into it is queued
        const int NB = a.NB;
        const WgtCarve cv(a.D, N, NB);
        if (cv.end > a.work_stride) {
            set_error("weights: workspace smaller than its carve-up");
            return SSMQ_E_ARG;
        }
        hipLaunchKernelGGL(k_wb_kmatrix, dim3((unsigned)((nn + 255) / 256), P), dim3(256), 0, s, a);
        for (int kb = 0; kb < nblk; ++kb) {
            hipLaunchKernelGGL(k_wb_chol_panel, dim3(nblk - kb, P), dim3(256), kCbLds, s, N, kb, a.work, a.work_stride, a.status);
            const int m = nblk - kb - 1;
            if (m > 0)
                hipLaunchKernelGGL(k_wb_chol_update, dim3(m * (m + 1) / 2, P), dim3(256), kCbLds, s, N, kb, a.work, a.work_stride,
                                   a.status);
        }
        int rc = hip_fail(hipGetLastError(), "k_wb_chol");
        if (rc) return rc;
        const int slots = (N + 63) / 64;
#define SSMQ_WB_INV(S, CPW) \
        hipLaunchKernelGGL((k_wb_inverse<S, CPW>), dim3((N + 4 * CPW - 1) / (4 * CPW), P), dim3(256), 0, s, N, a.work, a.work_stride, \
                           a.status)
        if (slots <= 12) SSMQ_WB_INV(12, 2);
        else if (slots <= 20) SSMQ_WB_INV(20, 2);
        else if (slots <= 32) SSMQ_WB_INV(32, 1);
        else SSMQ_WB_INV(64, 1);
#undef SSMQ_WB_INV
        if ((rc = hip_fail(hipGetLastError(), "k_wb_inverse"))) return rc;
        a.stage = 3;
        hipLaunchKernelGGL(k_weights<1024>, dim3(P), dim3(1024), tiles, s, a);
        const int64_t off_m1 = cv.M1, off_m2 = cv.M2, off_t4 = cv.T4;
        const dim3 gg((N + 63) / 64, (N + 63) / 64, P);
        if (NB == 0)
            hipLaunchKernelGGL(k_wb_gemm, gg, dim3(256), 0, s, a.work + off_m1, a.work_stride, N, a.Q, (int64_t)nn, N, a.iK,
                               (int64_t)nn, N, N, N, N, a.status);
        else
            hipLaunchKernelGGL(k_wb_gemm, gg, dim3(256), 0, s, a.work + off_m1, a.work_stride, N, a.work + off_t4, a.work_stride, N,
                               a.iK, (int64_t)nn, N, N, N, N, a.status);
        hipLaunchKernelGGL(k_wb_gemm, gg, dim3(256), 0, s, a.work + off_m2, a.work_stride, N, a.iK, (int64_t)nn, N, a.work + off_m1,
                           a.work_stride, N, N, N, N, a.status);
        a.stage = 4;
        hipLaunchKernelGGL(k_weights<1024>, dim3(P), dim3(1024), tiles, s, a);
        return hip_fail(hipGetLastError(), "k_weights(large)");
    }
    if (!staged) {
        // K, its factor and its inverse in the (L2-resident) workspace; the products still run through LDS tiles - the same
        // sums in the same order as gemm(), which streamed both operands of every dot product from L2 (N = 1 181: 2.5 s)
        a.use_lds = 0;
        a.tiled = 1;
        hipLaunchKernelGGL(k_weights<1024>, dim3(a.P), dim3(1024), tiles, s, a);
        return hip_fail(hipGetLastError(), "k_weights");
    }
    if (!a.lpack) {
        set_error("weights: no buffer for the packed factors");
        return SSMQ_E_ARG;
    }
    a.use_lds = 2;
    a.stage = 1;
    hipLaunchKernelGGL(k_weights<1024>, dim3(a.P), dim3(1024), packed, s, a);
    int rc = hip_fail(hipGetLastError(), "k_weights(factor)");
    if (!rc) {
        hipLaunchKernelGGL(k_weights_inverse, dim3((N + kInvCols - 1) / kInvCols, a.P), dim3(kInvBlock), packed, s, N, a.lpack,
                           a.status, a.work, a.work_stride);
        rc = hip_fail(hipGetLastError(), "k_weights_inverse");
    }
    if (!rc) {
        a.stage = 2;
        hipLaunchKernelGGL(k_weights<1024>, dim3(a.P), dim3(1024), tiles, s, a);
        rc = hip_fail(hipGetLastError(), "k_weights(algebra)");
    }
    return rc;
}

static int weights_impl(int var_mode, int D, int N, const double *xi, const double *par, int P, double jitter,
                        const int32_t *mulind, int NB, double *wm, double *Wc, double *Wcc, double *iK, double *q, double *Q, double *R,
                        double *model_var, double *integral_var, int32_t *status) {
    if (D < 1 || D > SSMQ_MAX_DIM || N < 1 || N > SSMQ_MAX_PTS || P < 1 || !xi || !par || NB < 0 || NB > N ||
        (NB > 0 && !mulind)) {
        set_error("weights: bad argument");
        return SSMQ_E_ARG;
    }
    if (NB > 0)
        for (int i = 0; i < D * NB; ++i)
            if (mulind[i] < 0 || mulind[i] > 20) {
                set_error("weights: multi-index entries must be in 0..20");
                return SSMQ_E_ARG;
            }
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = stream();
    const size_t nn = (size_t)N * N;
    const int64_t work_stride = (int64_t)(10 * nn + 4 * (size_t)N * std::max(NB, 1) + 2 * (size_t)std::max(NB, 1) * NB +
                                          (size_t)D * (N + NB) + 2 * N + NB + 64);
    // one device allocation, carved up (a call used to spend ~0.5 ms in 17 hipMalloc / hipFree pairs)
    struct Part { void *p = nullptr; double *d() { return (double *)p; } };
    Part dxi, dpar, dmi, dpx, dxpx, dpxpx, dwm, dWc, dWcc, diK, dq, dQ, dR, dmv, div, dst, dwork, dlp;
    const int nb1 = std::max(NB, 1);
    const size_t npk = (N > 64) ? (size_t)N * (N + 1) / 2 : 0;
    struct { Part *part; size_t bytes; } plan[] = {
        {&dxi, sizeof(double) * D * N}, {&dpar, sizeof(double) * P * (1 + D)}, {&dmi, sizeof(int32_t) * D * nb1},
        {&dpx, sizeof(double) * nb1}, {&dxpx, sizeof(double) * D * nb1}, {&dpxpx, sizeof(double) * nb1 * nb1},
        {&dwm, sizeof(double) * P * N}, {&dWc, sizeof(double) * P * nn}, {&dWcc, sizeof(double) * P * D * N},
        {&diK, sizeof(double) * P * nn}, {&dq, sizeof(double) * P * N}, {&dQ, sizeof(double) * P * nn},
        {&dR, sizeof(double) * P * D * N}, {&dmv, sizeof(double) * P}, {&div, sizeof(double) * P},
        {&dst, sizeof(int32_t) * P}, {&dwork, sizeof(double) * (size_t)P * work_stride}, {&dlp, sizeof(double) * P * npk}};
    size_t total = 0;
    for (auto &e : plan) total += (e.bytes + 255) / 256 * 256;
    DBuf arena;
    if ((rc = arena.alloc(total))) return rc;
    {
        char *base = (char *)arena.p;
        for (auto &e : plan) {
            e.part->p = base;
            base += (e.bytes + 255) / 256 * 256;
        }
    }
    SSMQ_HIP(hipMemcpyAsync(dxi.p, xi, sizeof(double) * D * N, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dpar.p, par, sizeof(double) * P * (1 + D), hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemsetAsync(dQ.p, 0, sizeof(double) * P * nn, s));
    std::vector<double> px, xpx, pxpx;
    if (NB > 0) {
        poly_moments(D, NB, mulind, px, xpx, pxpx);
        SSMQ_HIP(hipMemcpyAsync(dmi.p, mulind, sizeof(int32_t) * D * NB, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(dpx.p, px.data(), sizeof(double) * NB, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(dxpx.p, xpx.data(), sizeof(double) * D * NB, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(dpxpx.p, pxpx.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice, s));
    }
    WgtArgs a;
    memset(&a, 0, sizeof(a));
    a.D = D; a.N = N; a.P = P; a.NB = NB; a.jitter = jitter; a.var_mode = var_mode;
    a.xi = dxi.d(); a.par = dpar.d(); a.mulind = (const int32_t *)dmi.p; a.px = dpx.d(); a.xpx = dxpx.d();
    a.pxpx = dpxpx.d(); a.wm = dwm.d(); a.Wc = dWc.d(); a.Wcc = dWcc.d(); a.iK = diK.d(); a.q = dq.d(); a.Q = dQ.d();
    a.R = dR.d(); a.mv = dmv.d(); a.iv = div.d(); a.status = (int32_t *)dst.p; a.work = dwork.d();
    a.work_stride = work_stride;
    a.lpack = npk ? dlp.d() : nullptr;
    if ((rc = launch_weights(a, s))) return rc;
#define SSMQ_D2H(host, dev, count) \
    if (host) SSMQ_HIP(hipMemcpyAsync(host, dev.p, sizeof(*host) * (count), hipMemcpyDeviceToHost, s));
    SSMQ_D2H(wm, dwm, (size_t)P * N)
    SSMQ_D2H(Wc, dWc, (size_t)P * nn)
    SSMQ_D2H(Wcc, dWcc, (size_t)P * D * N)
    SSMQ_D2H(iK, diK, (size_t)P * nn)
    SSMQ_D2H(q, dq, (size_t)P * N)
    SSMQ_D2H(Q, dQ, (size_t)P * nn)
    SSMQ_D2H(R, dR, (size_t)P * D * N)
    SSMQ_D2H(model_var, dmv, (size_t)P)
    SSMQ_D2H(integral_var, div, (size_t)P)
#undef SSMQ_D2H
    std::vector<int32_t> st(P);
    SSMQ_HIP(hipMemcpyAsync(st.data(), dst.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    int first = 0;
    for (int i = 0; i < P; ++i) {
        if (status) status[i] = st[i];
        if (st[i] && !first) first = i + 1;
    }
    return first;
}

// ---- theta-batched GP weights that stay on the device ----------------------------------------------------------------
// consts[p] = WideLayout block of item p: xi' | wm | Wc | Wcc | emv = model_var * I_E | iK   (ssmq_wide.h)
__global__ void k_pack_wide_consts(int D, int E, int N, const double *xi, const double *wm, const double *Wc,
                                   const double *Wcc, const double *iK, const double *mv, double *consts) {
    const WideLayout cl = wide_layout(D, E, N, SSMQ_FORM_BQ);
    const int64_t p = blockIdx.x;
    double *c = consts + p * cl.total;
    for (int i = threadIdx.x; i < D * N; i += blockDim.x) c[cl.xiT + i] = xi[(i % D) * N + i / D];   // [N][D]
    for (int i = threadIdx.x; i < N; i += blockDim.x) c[cl.wm + i] = wm[p * N + i];
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
        c[cl.Wc + i] = Wc[p * N * N + i];
        c[cl.iK + i] = iK[p * N * N + i];
    }
    for (int i = threadIdx.x; i < D * N; i += blockDim.x) c[cl.Wcc + i] = Wcc[p * D * N + i];
    for (int i = threadIdx.x; i < E * E; i += blockDim.x) c[cl.emv + i] = mv[p];   // mode DIAG / BROADCAST picks entries
}

// d_xi [D][N], d_par [P][1+D] device; d_consts [P][wide_layout(D, E, N, BQ).total], d_status [P] device outputs.
// Synchronous (the temporaries are released on return).
// Workspace of gp_weights_wide_consts for P parameter rows (bytes; every part 256-byte aligned)
static size_t wide_consts_part(size_t doubles) { return (sizeof(double) * doubles + 255) / 256 * 256; }
size_t gp_weights_wide_ws_bytes(int D, int N, int64_t P) {
    const size_t nn = (size_t)N * N, p = (size_t)P;
    const size_t work_stride = 10 * nn + 4 * (size_t)N + (size_t)D * N + 2 * N + 64;
    return 2 * wide_consts_part(p * N) + 3 * wide_consts_part(p * nn) + 2 * wide_consts_part(p * D * N) +
           2 * wide_consts_part(p) + wide_consts_part(p * work_stride) + wide_consts_part(D + 2) +
           (N > 64 ? wide_consts_part(p * N * (N + 1) / 2) : 0);
}

// GP weights of P parameter rows, packed as the per-item constant blocks of the generic transform kernel.  Everything is
// enqueued on the library's stream, nothing is allocated and nothing waits: ws (gp_weights_wide_ws_bytes) is the caller's.
int gp_weights_wide_consts(int D, int E, int N, const double *d_xi, const double *d_par, int P, double jitter,
                           double *d_consts, int32_t *d_status, void *ws, size_t ws_bytes) {
    hipStream_t s = stream();
    const size_t nn = (size_t)N * N, p = (size_t)P;
    const int64_t work_stride = (int64_t)(10 * nn + 4 * (size_t)N + (size_t)D * N + 2 * N + 64);
    if (!ws || ws_bytes < gp_weights_wide_ws_bytes(D, N, P)) {
        set_error("gp_weights_wide_consts: workspace too small");
        return SSMQ_E_ARG;
    }
    char *w = (char *)ws;
    auto take = [&](size_t doubles) {
        double *r = (double *)w;
        w += wide_consts_part(doubles);
        return r;
    };
    double *dwm = take(p * N), *dq = take(p * N), *dWc = take(p * nn), *diK = take(p * nn), *dQ = take(p * nn);
    double *dWcc = take(p * D * N), *dR = take(p * D * N), *dmv = take(p), *div = take(p);
    double *dwork = take(p * (size_t)work_stride), *dz = take(D + 2);
    double *dlp = N > 64 ? take(p * N * (N + 1) / 2) : nullptr;   // packed Cholesky factors of the staged large-N path
    int rc;
    WgtArgs a;          // (the GP branch of k_weights writes every element of Q, NaN on failure: nothing to clear)
    memset(&a, 0, sizeof(a));
    a.D = D; a.N = N; a.P = P; a.NB = 0; a.jitter = jitter;
    a.xi = d_xi; a.par = d_par; a.mulind = (const int32_t *)dz; a.px = dz; a.xpx = dz; a.pxpx = dz;
    a.wm = dwm; a.Wc = dWc; a.Wcc = dWcc; a.iK = diK; a.q = dq; a.Q = dQ; a.R = dR;
    a.mv = dmv; a.iv = div; a.status = d_status; a.work = dwork; a.work_stride = work_stride;
    a.lpack = dlp;
    if ((rc = launch_weights(a, s))) return rc;
    hipLaunchKernelGGL(k_pack_wide_consts, dim3(P), dim3(64), 0, s, D, E, N, d_xi, dwm, dWc, dWcc, diK, dmv, d_consts);
    return hip_fail(hipGetLastError(), "k_pack_wide_consts");
}

// ---- both transforms' weights of the theta-batched step in ONE launch, workspace and results in LDS ----------------------
// gp_weights_wide_consts twice was four launches (k_weights, k_pack_wide_consts for the dynamics and for the measurement
// model), each k_weights a chain of ~25 barrier-separated steps over a GLOBAL workspace - at N = 4 ... 21 points that is
// 13 us of write -> barrier -> read round trips for a few thousand flops.  Here blockIdx.y picks the transform, the
// workspace and every intermediate result of weights_body (q, Q, R, iK, wm, Wc, Wcc) live in the workgroup's LDS, and
// the only global writes are the transform kernel's constant block (what k_pack_wide_consts copied) and the status flag.
// Same body, same summation orders: the constants are the same bits as the two-launch route's.
struct ThetaWgtArgs {
    WgtArgs w[2];
    double *consts[2];
    int32_t E[2];
    const int32_t *count;      // null, or the number of items on the device (the grid then covers an upper bound): the
                               // device-resident rounds of the batched marginalised filter (ssmq_marginal.hip)
};

static size_t theta_weights_lds_doubles(int D, int N) {
    const size_t nn = (size_t)N * N;
    return 2 * nn + (4 * nn + (size_t)D * N + N) + (3 * nn + 2 * (size_t)N + 2 * (size_t)D * N + 2);
}

__global__ __launch_bounds__(256) void k_theta_weights(const ThetaWgtArgs t) {
    extern __shared__ __align__(16) double lds[];
    if (t.count && (int)blockIdx.x >= *t.count) return;      // (whole workgroup: before any barrier)
    const bool second = blockIdx.y != 0;
    const WgtArgs a = second ? t.w[1] : t.w[0];
    double *consts = second ? t.consts[1] : t.consts[0];
    const int E = second ? t.E[1] : t.E[0];
    const int p = blockIdx.x, N = a.N, D = a.D, nn = N * N;
    double *w = lds + 2 * nn;           // [0, 2 nn): the kernel matrix and its inverse (weights_body, use_lds == 1)
    WgtOut o;
    o.work = w; w += 4 * nn + D * N + N;
    o.q = w; w += N;
    o.Q = w; w += nn;
    o.R = w; w += D * N;
    o.iK = w; w += nn;
    o.wm = w; w += N;
    o.Wc = w; w += nn;
    o.Wcc = w; w += D * N;
    o.mv = w;
    o.iv = w + 1;
    o.status = a.status + p;
    weights_body(a, o, p, lds);
    bsync();
    const WideLayout cl = wide_layout(D, E, N, SSMQ_FORM_BQ);
    double *c = consts + (int64_t)p * cl.total;
    for (int i = threadIdx.x; i < D * N; i += blockDim.x) c[cl.xiT + i] = a.xi[(i % D) * N + i / D];   // [N][D]
    for (int i = threadIdx.x; i < N; i += blockDim.x) c[cl.wm + i] = o.wm[i];
    for (int i = threadIdx.x; i < nn; i += blockDim.x) {
        c[cl.Wc + i] = o.Wc[i];
        c[cl.iK + i] = o.iK[i];
    }
    for (int i = threadIdx.x; i < D * N; i += blockDim.x) c[cl.Wcc + i] = o.Wcc[i];
    const double mv = *o.mv;
    for (int i = threadIdx.x; i < E * E; i += blockDim.x) c[cl.emv + i] = mv;
}

bool gp_theta_weights_fits(int D0, int N0, int D1, int N1) {
    const size_t cap = 160 * 1024 - 1024;
    return N0 <= 64 && N1 <= 64 && sizeof(double) * theta_weights_lds_doubles(D0, N0) <= cap &&
           sizeof(double) * theta_weights_lds_doubles(D1, N1) <= cap;
}

// transform i (0: dynamics, 1: measurement): D[i] -> E[i] on N[i] points d_xi[i], parameter rows d_par[i] [P][1 + D[i]],
// constant blocks d_consts[i] [P][wide_layout(D, E, N, BQ).total], flags d_status[i] [P].  Enqueued, nothing allocated.
int gp_theta_weights_pair(const int D[2], const int E[2], const int N[2], const double *const d_xi[2],
                          const double *const d_par[2], int P, double jitter, double *const d_consts[2],
                          int32_t *const d_status[2], const int32_t *d_count) {
    hipStream_t s = stream();
    ThetaWgtArgs t;
    memset(&t, 0, sizeof(t));
    t.count = d_count;
    size_t lds = 0;
    int threads = 64;
    for (int i = 0; i < 2; ++i) {
        WgtArgs &a = t.w[i];
        a.D = D[i]; a.N = N[i]; a.P = P; a.NB = 0; a.use_lds = 1; a.stage = 0; a.jitter = jitter;
        a.xi = d_xi[i]; a.par = d_par[i]; a.mulind = (const int32_t *)d_xi[i]; a.px = a.xpx = a.pxpx = d_xi[i];   // unused: NB = 0
        a.status = d_status[i];
        t.consts[i] = d_consts[i];
        t.E[i] = E[i];
        lds = std::max(lds, sizeof(double) * theta_weights_lds_doubles(D[i], N[i]));
        if (N[i] > 8 || ssmq::sw("SSMQ_WEIGHTS_WIDE_BLOCK")) threads = 256;     // as launch_weights
    }
    static thread_local unsigned attr_epoch = 0;
    if (lds > 48 * 1024 && attr_epoch != ssmq::device_epoch()) {
        SSMQ_HIP(hipFuncSetAttribute((const void *)k_theta_weights, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attr_epoch = ssmq::device_epoch();
    }
    hipLaunchKernelGGL(k_theta_weights, dim3(P, 2), dim3(threads), lds, s, t);
    return hip_fail(hipGetLastError(), "k_theta_weights");
}

// ---- the kernel-level methods of the reference as entry points of their own ---------------------------------------------
// RBFGauss.eval (bq/bqkern.py:329-343 with utils.maha, utils.py:385-409), Kernel.eval_chol / eval_inv_dot / _cho_inv
// (:38-64, 96-142) and RBFGauss.exp_x_kxkx for two different parameter rows (:366-415).  The weights kernel above needs
// none of them separately (it builds K, its inverse and Q in one go); they exist so that callers of those methods - the
// reference's tests, hyper-parameter studies - get device results too.

// K[p][i][j] = exp(2 log(alpha) - maha(z1_i, z2_j) / 2), z = Lam^-1/2 x, maha as |a|^2 + |b|^2 - 2 a.b; diag: only
// i == j through the difference form of the reference's `diag=True` branch.  One thread per entry.
__global__ void k_rbf_eval(int D, int N1, int N2, const double *__restrict__ x1, const double *__restrict__ x2,
                           const double *__restrict__ par, int scaling, int diag, double *__restrict__ K) {
    const int p = blockIdx.y;
    const double *pr = par + (int64_t)p * (1 + D);
    const double la = scaling ? 2.0 * log(pr[0]) : 2.0 * log(1.0);
    const int64_t total = diag ? N1 : (int64_t)N1 * N2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = diag ? (int)idx : (int)(idx / N2), j = diag ? (int)idx : (int)(idx % N2);
        double na = 0.0, nb = 0.0, dot = 0.0, dd = 0.0;
        for (int d = 0; d < D; ++d) {
            const double sil = 1.0 / pr[1 + d];
            const double a = sil * x1[d * N1 + i], b = sil * x2[d * N2 + j];
            na += a * a;
            nb += b * b;
            dot += a * b;
            dd += (a - b) * (a - b);
        }
        const double mh = diag ? dd : (na + nb) - 2.0 * dot;
        K[(int64_t)p * total + idx] = exp(la - 0.5 * mh);
    }
}

// X = (L L')^-1 Bm for the lower factor L (n x n) and a square right-hand side: thread per column, forward then backward
// substitution (scipy.linalg.cho_solve)
__device__ void chol_solve(const double *L, const double *Bm, double *X, int n) {
    for (int c = threadIdx.x; c < n; c += kWgtBlock) {
        for (int i = 0; i < n; ++i) {
            double s = Bm[i * n + c];
            for (int k = 0; k < i; ++k) s -= L[i * n + k] * X[k * n + c];
            X[i * n + c] = s / L[i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = X[i * n + c];
            for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * X[k * n + c];
            X[i * n + c] = s / L[i * n + i];
        }
    }
    bsync();
}

// A <- K + jitter I (K as k_rbf_eval, scaling optional), factor, optional inverse / solve: one workgroup per parameter
// row, global workspace (init-time sizes).  chol: lower factor with zeros above the diagonal (numpy.linalg.cholesky); iK
// = sym(A^-1) (rhs null) or sym(A^-1 rhs) for a square rhs - the reference symmetrises whatever it solved for.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_rbf_factor(int D, int N, const double *__restrict__ x, const double *__restrict__ par,
                                                       int scaling, double jitter, double *__restrict__ work, double *chol,
                                                       double *iK, const double *__restrict__ rhs, int32_t *status) {
    __shared__ int s_flag;
    const int p = blockIdx.x, tid = threadIdx.x;
    const double *pr = par + (int64_t)p * (1 + D);
    const double la = scaling ? 2.0 * log(pr[0]) : 2.0 * log(1.0);
    double *A = work + (int64_t)p * 2 * N * N, *X = A + (int64_t)N * N;
    for (int idx = tid; idx < N * N; idx += kWgtBlock) {
        const int i = idx / N, j = idx % N;
        double na = 0.0, nb = 0.0, dot = 0.0;
        for (int d = 0; d < D; ++d) {
            const double sil = 1.0 / pr[1 + d];
            const double a = sil * x[d * N + i], b = sil * x[d * N + j];
            na += a * a;
            nb += b * b;
            dot += a * b;
        }
        A[idx] = exp(la - 0.5 * ((na + nb) - 2.0 * dot)) + (i == j ? jitter : 0.0);
    }
    bsync();
    const bool pd = chol_block(A, N, &s_flag);
    if (tid == 0) status[p] = pd ? 0 : 1;
    const double nan = __builtin_nan("");
    if (chol)
        for (int idx = tid; idx < N * N; idx += kWgtBlock) {
            const int i = idx / N, j = idx % N;
            chol[(int64_t)p * N * N + idx] = pd ? (j <= i ? A[idx] : 0.0) : nan;
        }
    if (!iK) return;
    if (!pd) {
        for (int idx = tid; idx < N * N; idx += kWgtBlock) iK[(int64_t)p * N * N + idx] = nan;
        return;
    }
    if (rhs) chol_solve(A, rhs, X, N);
    else chol_inverse(A, X, N);
    for (int idx = tid; idx < N * N; idx += kWgtBlock) {
        const int i = idx / N, j = idx % N;
        iK[(int64_t)p * N * N + idx] = 0.5 * (X[i * N + j] + X[j * N + i]);
    }
}

// Q[i][j] = det(R)^-1/2 exp(xi_i + xi'_j + maha(Lam0^-1 x_i, -Lam1^-1 x_j; R^-1) / 2), R = Lam0^-1 + Lam1^-1 + I,
// xi = 2 log(alpha0) - |Lam0^-1/2 x_i|^2 / 2, xi' likewise with row 1 (all matrices diagonal)
__global__ void k_rbf_kxkx(int D, int N, const double *__restrict__ x, const double *__restrict__ par0,
                           const double *__restrict__ par1, int scaling, double *__restrict__ Q) {
    const double la0 = scaling ? 2.0 * log(par0[0]) : 2.0 * log(1.0), la1 = scaling ? 2.0 * log(par1[0]) : 2.0 * log(1.0);
    double det = 1.0;
    for (int d = 0; d < D; ++d) {
        const double s0 = 1.0 / par0[1 + d], s1 = 1.0 / par1[1 + d];
        det *= (s0 * s0 + s1 * s1) + 1.0;
    }
    const double c = 1.0 / sqrt(det);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * N; idx += gridDim.x * blockDim.x) {
        const int i = idx / N, j = idx % N;
        double n0 = 0.0, n1 = 0.0, m2i = 0.0, m2j = 0.0, mij = 0.0;
        for (int d = 0; d < D; ++d) {
            const double s0 = 1.0 / par0[1 + d], s1 = 1.0 / par1[1 + d];
            const double il0 = s0 * s0, il1 = s1 * s1;
            const double z0 = s0 * x[d * N + i], z1 = s1 * x[d * N + j];
            n0 += z0 * z0;
            n1 += z1 * z1;
            const double v = 1.0 / ((il0 + il1) + 1.0);
            const double yi = il0 * x[d * N + i], yj = -(il1 * x[d * N + j]);
            m2i += (yi * v) * yi;
            m2j += (yj * v) * yj;
            mij += (yi * v) * yj;
        }
        const double mh = (m2i + m2j) - 2.0 * mij;
        Q[idx] = c * exp(((la0 - 0.5 * n0) + (la1 - 0.5 * n1)) + 0.5 * mh);
    }
}

static bool rbf_args_ok(int D, int N, const double *x, const double *par, int P) {
    return D >= 1 && D <= SSMQ_MAX_DIM && N >= 1 && N <= SSMQ_MAX_PTS && P >= 1 && x && par;
}

}  // namespace ssmq

extern "C" int ssmq_rbf_eval(int D, int N1, const double *x1, int N2, const double *x2, const double *par, int P,
                             int scaling, int diag, double *K) {
    using namespace ssmq;
    if (!x2) { x2 = x1; N2 = N1; }
    if (!rbf_args_ok(D, N1, x1, par, P) || N2 < 1 || N2 > SSMQ_MAX_PTS || !K || (diag && N1 != N2)) {
        set_error("rbf_eval: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = stream();
    const size_t nout = (size_t)P * (diag ? (size_t)N1 : (size_t)N1 * N2);
    DBuf d1, d2, dp, dk;
    if ((rc = d1.alloc(sizeof(double) * D * N1)) || (rc = d2.alloc(sizeof(double) * D * N2)) ||
        (rc = dp.alloc(sizeof(double) * P * (1 + D))) || (rc = dk.alloc(sizeof(double) * nout)))
        return rc;
    SSMQ_HIP(hipMemcpyAsync(d1.p, x1, sizeof(double) * D * N1, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(d2.p, x2, sizeof(double) * D * N2, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dp.p, par, sizeof(double) * P * (1 + D), hipMemcpyHostToDevice, s));
    const size_t per = nout / P;
    hipLaunchKernelGGL(k_rbf_eval, dim3((unsigned)std::min<size_t>((per + 255) / 256, 4096), P), dim3(256), 0, s, D, N1, N2, d1.d(),
                       d2.d(), dp.d(), scaling, diag, dk.d());
    if ((rc = hip_fail(hipGetLastError(), "k_rbf_eval"))) return rc;
    SSMQ_HIP(hipMemcpyAsync(K, dk.p, sizeof(double) * nout, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    return SSMQ_OK;
}

extern "C" int ssmq_rbf_factor(int D, int N, const double *x, const double *par, int P, int scaling, double jitter,
                               const double *rhs, double *chol, double *iK, int32_t *status) {
    using namespace ssmq;
    if (!rbf_args_ok(D, N, x, par, P) || (!chol && !iK)) {
        set_error("rbf_factor: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = stream();
    const size_t nn = (size_t)N * N;
    DBuf dx, dp, dw, dc, di, dst, db;
    if ((rc = db.alloc(sizeof(double) * (rhs ? nn : 1)))) return rc;
    if (rhs) SSMQ_HIP(hipMemcpyAsync(db.p, rhs, sizeof(double) * nn, hipMemcpyHostToDevice, s));
    if ((rc = dx.alloc(sizeof(double) * D * N)) || (rc = dp.alloc(sizeof(double) * P * (1 + D))) ||
        (rc = dw.alloc(sizeof(double) * 2 * nn * P)) || (rc = dc.alloc(sizeof(double) * (chol ? nn * P : 1))) ||
        (rc = di.alloc(sizeof(double) * (iK ? nn * P : 1))) || (rc = dst.alloc(sizeof(int32_t) * P)))
        return rc;
    SSMQ_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * D * N, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dp.p, par, sizeof(double) * P * (1 + D), hipMemcpyHostToDevice, s));
    if (N > 64)
        hipLaunchKernelGGL(k_rbf_factor<1024>, dim3(P), dim3(1024), 0, s, D, N, dx.d(), dp.d(), scaling, jitter, dw.d(),
                           chol ? dc.d() : nullptr, iK ? di.d() : nullptr, rhs ? db.d() : nullptr, (int32_t *)dst.p);
    else
        hipLaunchKernelGGL(k_rbf_factor<256>, dim3(P), dim3(256), 0, s, D, N, dx.d(), dp.d(), scaling, jitter, dw.d(),
                           chol ? dc.d() : nullptr, iK ? di.d() : nullptr, rhs ? db.d() : nullptr, (int32_t *)dst.p);
    if ((rc = hip_fail(hipGetLastError(), "k_rbf_factor"))) return rc;
    if (chol) SSMQ_HIP(hipMemcpyAsync(chol, dc.p, sizeof(double) * nn * P, hipMemcpyDeviceToHost, s));
    if (iK) SSMQ_HIP(hipMemcpyAsync(iK, di.p, sizeof(double) * nn * P, hipMemcpyDeviceToHost, s));
    std::vector<int32_t> st(P);
    SSMQ_HIP(hipMemcpyAsync(st.data(), dst.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    int first = 0;
    for (int i = 0; i < P; ++i) {
        if (status) status[i] = st[i];
        if (st[i] && !first) first = i + 1;
    }
    return first;
}

extern "C" int ssmq_rbf_exp_kxkx(int D, int N, const double *x, const double *par0, const double *par1, int scaling,
                                 double *Q) {
    using namespace ssmq;
    if (!rbf_args_ok(D, N, x, par0, 1) || !par1 || !Q) {
        set_error("rbf_exp_kxkx: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = stream();
    DBuf dx, dp, dq;
    if ((rc = dx.alloc(sizeof(double) * D * N)) || (rc = dp.alloc(sizeof(double) * 2 * (1 + D))) ||
        (rc = dq.alloc(sizeof(double) * (size_t)N * N)))
        return rc;
    SSMQ_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * D * N, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dp.p, par0, sizeof(double) * (1 + D), hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dp.d() + 1 + D, par1, sizeof(double) * (1 + D), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_rbf_kxkx, dim3((unsigned)std::min<size_t>(((size_t)N * N + 255) / 256, 4096)), dim3(256), 0, s, D, N,
                       dx.d(), dp.d(), dp.d() + 1 + D, scaling, dq.d());
    if ((rc = hip_fail(hipGetLastError(), "k_rbf_kxkx"))) return rc;
    SSMQ_HIP(hipMemcpyAsync(Q, dq.p, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    return SSMQ_OK;
}

// BayesSardModel._exp_x_kxpx and utils.vandermonde for an arbitrary point set (the weights kernel forms both inline)
__global__ void k_bs_moments(int D, int N, int NB, const double *__restrict__ x, const double *__restrict__ par,
                             const int32_t *__restrict__ mulind, double *__restrict__ vand, double *__restrict__ kxpx) {
    __shared__ double sil[SSMQ_MAX_DIM];
    if ((int)threadIdx.x < D) sil[threadIdx.x] = 1.0 / par[1 + threadIdx.x];
    __syncthreads();
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)N * NB; idx += (int64_t)gridDim.x * blockDim.x) {
        double v, k;
        ssmq::bs_basis_entry(D, N, NB, (int)(idx / NB), (int)(idx % NB), x, mulind, sil, v, k);
        if (vand) vand[idx] = v;
        if (kxpx) kxpx[idx] = k;
    }
}

extern "C" int ssmq_bs_moments(int D, int N, const double *x, const double *par, const int32_t *mulind, int NB, double *px,
                               double *xpx, double *pxpx, double *kxpx, double *vand) {
    using namespace ssmq;
    if (D < 1 || D > SSMQ_MAX_DIM || NB < 1 || !mulind || N < 0 || ((kxpx || vand) && (!x || N < 1)) || (kxpx && !par)) {
        set_error("bs_moments: bad argument");
        return SSMQ_E_ARG;
    }
    for (int i = 0; i < D * NB; ++i)
        if (mulind[i] < 0) {
            set_error("bs_moments: negative multi-index");
            return SSMQ_E_ARG;
        }
    if (px || xpx || pxpx) {     // integer arithmetic on the multi-indices: host code, as the point sets are
        std::vector<double> a, b, c;
        poly_moments(D, NB, mulind, a, b, c);
        if (px) std::copy(a.begin(), a.end(), px);
        if (xpx) std::copy(b.begin(), b.end(), xpx);
        if (pxpx) std::copy(c.begin(), c.end(), pxpx);
    }
    if (!kxpx && !vand) return SSMQ_OK;
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = stream();
    const size_t nn = (size_t)N * NB;
    DBuf dx, dp, dm, dk, dv;
    const double one_par[1 + SSMQ_MAX_DIM] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    if ((rc = dx.alloc(sizeof(double) * D * N)) || (rc = dp.alloc(sizeof(double) * (1 + D))) ||
        (rc = dm.alloc(sizeof(int32_t) * D * NB)) || (kxpx && (rc = dk.alloc(sizeof(double) * nn))) ||
        (vand && (rc = dv.alloc(sizeof(double) * nn))))
        return rc;
    SSMQ_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * D * N, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dp.p, par ? par : one_par, sizeof(double) * (1 + D), hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dm.p, mulind, sizeof(int32_t) * D * NB, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bs_moments, dim3((unsigned)std::min<size_t>((nn + 255) / 256, 4096)), dim3(256), 0, s, D, N, NB, dx.d(),
                       dp.d(), (const int32_t *)dm.p, vand ? dv.d() : nullptr, kxpx ? dk.d() : nullptr);
    if ((rc = hip_fail(hipGetLastError(), "k_bs_moments"))) return rc;
    if (kxpx) SSMQ_HIP(hipMemcpyAsync(kxpx, dk.p, sizeof(double) * nn, hipMemcpyDeviceToHost, s));
    if (vand) SSMQ_HIP(hipMemcpyAsync(vand, dv.p, sizeof(double) * nn, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    return SSMQ_OK;
}

extern "C" int ssmq_weights_gp(int D, int N, const double *xi, const double *par, int P, double jitter, double *wm,
                               double *Wc, double *Wcc, double *iK, double *q, double *Q, double *R, double *model_var,
                               double *integral_var, int32_t *status) {
    return ssmq::weights_impl(0, D, N, xi, par, P, jitter, nullptr, 0, wm, Wc, Wcc, iK, q, Q, R, model_var, integral_var,
                              status);
}

// The Student-t process model integrates with the SAME weights as the GP model (StudentTProcessModel inherits
// GaussianProcessModel.bq_weights, bq/bqmod.py:1060-1130); only its model / integral variance are rescaled by the data
// (bq/bqmod.py:1132-1190), which the transform kernels do per trajectory (tp_nu, tp_iK of ssmq_transform_create).
extern "C" int ssmq_weights_tp(int D, int N, const double *xi, const double *par, int P, double jitter, double *wm,
                               double *Wc, double *Wcc, double *iK, double *q, double *Q, double *R, double *model_var,
                               double *integral_var, int32_t *status) {
    return ssmq_weights_gp(D, N, xi, par, P, jitter, wm, Wc, Wcc, iK, q, Q, R, model_var, integral_var, status);
}

extern "C" int ssmq_weights_bs(int D, int N, const double *xi, const double *par, int P, double jitter,
                               const int32_t *mulind, int NB, double *wm, double *Wc, double *Wcc, double *iK,
                               double *q, double *Q, double *R, double *model_var, double *integral_var,
                               int32_t *status) {
    if (NB < 1) {
        ssmq::set_error("weights_bs: NB must be >= 1");
        return SSMQ_E_ARG;
    }
    return ssmq::weights_impl(0, D, N, xi, par, P, jitter, mulind, NB, wm, Wc, Wcc, iK, q, Q, R, model_var, integral_var,
                              status);
}

extern "C" int ssmq_variances_bs(int D, int N, const double *xi, const double *par, int P, double jitter,
                                 const int32_t *mulind, int NB, double *model_var, double *integral_var,
                                 int32_t *status) {
    if (NB < 1) {
        ssmq::set_error("variances_bs: NB must be >= 1");
        return SSMQ_E_ARG;
    }
    return ssmq::weights_impl(1, D, N, xi, par, P, jitter, mulind, NB, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, model_var, integral_var, status);
}
