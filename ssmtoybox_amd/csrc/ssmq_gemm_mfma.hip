// T = FX * Wc for a whole batch on the matrix cores: the one GEMM-shaped stage of the moment transform.
//
// bq/bqmtran.py:199 forms fx Wc fx' per trajectory; with N = 201 points (fully-symmetric degree-5 rule at D = 10, the
// Bayes-Sard configuration) the first product is 87 % of the transform's arithmetic and, stacked over the batch, a plain
// GEMM: FX is (B E) x N row-major (the E rows of trajectory b are rows b E .. b E + E - 1), Wc is N x N and shared by
// everyone.  v_mfma_f64_16x16x4_f64 tiles it: a workgroup of 4 waves owns 128 rows and ALL N columns, so Wc is read
// from L2 once per 128 rows instead of once per trajectory (what limits the generic kernel: 2 TFLOP/s at N = 201).
//
//   wave w: RT 16-row tiles of the block (RT = 1 or 2: 64 or 128 rows per workgroup), NT column tiles of 16 -> RT NT
//   accumulator tiles
//   k loop in blocks of 16: the block's 16 x (16 NT) slab of Wc goes through LDS (double-buffered, row pitch
//   16 NT + 4 doubles so that the two 16-lane halves of a ds_read_b64 group fall on disjoint banks); each lane loads 4
//   consecutive doubles of its A row (32 B) and feeds element s in MFMA step s, i.e. the k order inside a block is
//   permuted (k = 4 g + s for lane group g) - consistently for A and B, so the sum is unchanged.
// Operand maps (cdna_hip_programming.md): A lane l -> A[l & 15][l >> 4], B lane l -> B[l >> 4][l & 15],
// C/D register r of lane l -> row (l >> 4) + 4 r, column l & 15.
// Padding: K and the column count are padded to 16 NT with zeros (FX columns / Wc rows and columns), rows beyond M are
// neither loaded nor stored.
#include "ssmq_host.h"

namespace ssmq {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kGemmBlock = 256;

template <int NT, int RT>   // RT row tiles of 16 per wave: a workgroup owns 64 RT rows
__global__ __launch_bounds__(kGemmBlock, RT == 1 ? 2 : 1) void k_fxwc_mfma(const double *__restrict__ A, const double *__restrict__ Bm,
                                                          double *__restrict__ T, int64_t M, int lda, int ldt, int KB) {
    // KB: k blocks of 16 (= NT for the square case).  Large point sets run one launch over column blocks of NP columns:
    // blockIdx.y selects the block, stored contiguously as [KB 16][NP] (host-made, zero-padded), and the NP output columns
    constexpr int NP = NT * 16;          // columns of this block (the square case: padded N = padded K)
    Bm += (int64_t)blockIdx.y * KB * 16 * NP;
    T += (int64_t)blockIdx.y * NP;
    constexpr int LB = NP + 4;           // LDS row pitch (doubles)
    extern __shared__ __align__(16) double lds[];      // [2][16][LB]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (64 * RT) + wave * (16 * RT);
    v4d acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = v4d{0.0, 0.0, 0.0, 0.0};

    // software pipeline: the next block's slab of Wc and A fragments are loaded into registers BEFORE the MFMA loop of
    // the current block and parked in LDS / renamed after it, so the L2 latency hides behind the matrix instructions
    constexpr int PER = 16 * NP / kGemmBlock;     // slab elements per thread: 16 * 16 NT / 256 = NT exactly, no tail
    static_assert(PER * kGemmBlock == 16 * NP, "slab size must be a multiple of the block size");
    double breg[PER], a[RT][4], an[RT][4];
    auto load_b = [&](int kb) {
#pragma unroll
        for (int q = 0; q < PER; ++q)     // rows 16 kb .. 16 kb + 15 are contiguous
            breg[q] = Bm[(int64_t)kb * 16 * NP + threadIdx.x + q * kGemmBlock];
    };
    auto park_b = [&](int buf) {
        double *dst = lds + buf * 16 * LB;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = threadIdx.x + q * kGemmBlock;
            dst[(i / NP) * LB + i % NP] = breg[q];
        }
    };
    auto load_a = [&](int kb, double (&dst)[RT][4]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + li;
            if (row < M) {
                const double2 *p = (const double2 *)(A + row * lda + kb * 16 + 4 * lg);
                const double2 v0 = p[0], v1 = p[1];
                dst[rt][0] = v0.x; dst[rt][1] = v0.y; dst[rt][2] = v1.x; dst[rt][3] = v1.y;
            } else {
                dst[rt][0] = dst[rt][1] = dst[rt][2] = dst[rt][3] = 0.0;
            }
        }
    };
    load_b(0);
    load_a(0, a);
    park_b(0);
    __syncthreads();
    for (int kb = 0; kb < KB; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < KB) {
            load_b(kb + 1);
            load_a(kb + 1, an);
        }
        const double *sb = lds + buf * 16 * LB;
#ifndef SSMQ_GEMM_PARK_AT
#define SSMQ_GEMM_PARK_AT 3      // the next slab goes to LDS before the last of the four k sub-steps: the writes and
                                 // the waits for the global loads hide behind its MFMAs (2-4 % over parking after them)
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s == SSMQ_GEMM_PARK_AT && kb + 1 < KB) park_b(buf ^ 1);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const double b = sb[(4 * lg + s) * LB + ct * 16 + li];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt][s], b, acc[rt][ct], 0, 0, 0);
            }
        }
        if (kb + 1 < KB) {
            if (SSMQ_GEMM_PARK_AT >= 4) park_b(buf ^ 1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[rt][q] = an[rt][q];
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + rt * 16 + lg + 4 * r;
            if (row < M) {
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) T[row * ldt + ct * 16 + li] = acc[rt][ct][r];
            }
        }
}

// ---- fused variant: covariance and cross-covariance of every trajectory straight from the accumulators ------------------
// cov_b = (fx_b Wc) fx_b' - m_b m_b' + emv and ccov_b = (fx_b Wcc') L_b' (bq/bqmtran.py:199, 223) need T = FX Wc and
// G = FX Wcc' only as intermediates: writing T's (B E) x NP rows and re-reading them together with FX in a third pass
// was 2 x 166 MB of scratch traffic for 8 MB of results at B = 1e4 (that pass took 45 % of the whole transform).  Here
// the second product runs on the matrix cores too, on the tile the wave already holds:
//   1. the first GEMM is computed TRANSPOSED, [T G]' = [Wc | Wcc']' FX' - the slab in LDS is a block of rows of the
//      host-made matrix X = [Wc | Wcc'] (NP x (NP + 16)), the operands of the MFMA are swapped - and the columns of X
//      are taken in the order pi(lg + 4 r) = 4 lg + r inside each block of 16.  Accumulator register r of column tile
//      ct then holds T[row0 + li][16 ct + 4 lg + r]: exactly the A-operand fragment (row li, k = lg) of the MFMA step
//      that sums over j in {16 ct + r, + 4, + 8, + 12} - no transposition, no LDS round trip;
//   2. S = T FX2' with FX2 = the 32 rows that start at the first row of the first trajectory intersecting the tile (all
//      trajectories that intersect a 16-row tile lie inside them for the supported E, fxwc_cov_supported); its B
//      fragments are FX[row][16 ct + 4 lg .. + 3], the very 16-byte pairs the main loop loads - now L2 hits;
//   3. lane (lg, li) holds S[row0 + lg + 4 r][s0 + 16 h + li]: entries whose two rows belong to the same trajectory are
//      covariance entries; the lower triangle is finished (- mean mean' + emv, scale, additive term) and stored to both
//      (e, e2) and (e2, e), as the per-trajectory kernels mirror it;
//   4. the last column tile holds G[row0 + li][4 lg + r]: each lane multiplies its four columns into L_b' and the four
//      lane groups are summed with two cross-lane exchanges.
struct CovEpilogue {
    const double *mean_rows;   // [M] transformed means, row b E + e (written by the evaluation pass)
    const double *chol;        // [B][D][D] lower Cholesky factors of the input covariances (evaluation pass)
    const double *emv;         // [E*E]
    const double *cov_add;     // [E*E] or null
    double *cov_f, *cov_fx;    // element idx of trajectory b at cov_f[idx * es + b * bs], cov_fx[idx * es + b * bs_fx]
    int64_t es, bs, bs_fx;
    int32_t E, D, emv_broadcast;
    double cov_scale, ccov_scale;
};

template <int NT>
__global__ __launch_bounds__(kGemmBlock, 2) void k_fxwc_cov_mfma(const double *__restrict__ A, const double *__restrict__ X,
                                                                int64_t M, int lda, const CovEpilogue ep) {
    constexpr int NP = NT * 16;          // padded point count = K
    constexpr int NX = NP + 16;          // columns of X = [Wc | Wcc']
    constexpr int NTX = NT + 1;
    constexpr int LB = NX + 4;           // LDS row pitch (doubles)
    extern __shared__ __align__(16) double lds[];      // [2][16][LB]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int lip = 4 * (li & 3) + (li >> 2);           // pi(li): column of the block this lane feeds as A operand
    const int64_t row0 = (int64_t)blockIdx.x * 64 + wave * 16;
    v4d acc[NTX];
#pragma unroll
    for (int ct = 0; ct < NTX; ++ct) acc[ct] = v4d{0.0, 0.0, 0.0, 0.0};
    constexpr int PER = 16 * NX / kGemmBlock;
    static_assert(PER * kGemmBlock == 16 * NX, "slab size must be a multiple of the block size");
    double breg[PER], a[4], an[4];
    auto load_b = [&](int kb) {
#pragma unroll
        for (int q = 0; q < PER; ++q) breg[q] = X[(int64_t)kb * 16 * NX + threadIdx.x + q * kGemmBlock];
    };
    auto park_b = [&](int buf) {
        double *dst = lds + buf * 16 * LB;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = threadIdx.x + q * kGemmBlock;
            dst[(i / NX) * LB + i % NX] = breg[q];
        }
    };
    auto load_frag = [&](int64_t row, int kb, double (&dst)[4]) {
        if (row < M) {
            const double2 *p = (const double2 *)(A + row * lda + kb * 16 + 4 * lg);
            const double2 v0 = p[0], v1 = p[1];
            dst[0] = v0.x; dst[1] = v0.y; dst[2] = v1.x; dst[3] = v1.y;
        } else {
            dst[0] = dst[1] = dst[2] = dst[3] = 0.0;
        }
    };
    load_b(0);
    load_frag(row0 + li, 0, a);
    park_b(0);
    __syncthreads();
    for (int kb = 0; kb < NT; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < NT) {
            load_b(kb + 1);
            load_frag(row0 + li, kb + 1, an);
        }
        const double *sb = lds + buf * 16 * LB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s == SSMQ_GEMM_PARK_AT && kb + 1 < NT) park_b(buf ^ 1);
#pragma unroll
            for (int ct = 0; ct < NTX; ++ct) {
                const double w = sb[(4 * lg + s) * LB + ct * 16 + lip];    // X[16 kb + 4 lg + s][16 ct + pi(li)]
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, a[s], acc[ct], 0, 0, 0);
            }
        }
        if (kb + 1 < NT) {
            if (SSMQ_GEMM_PARK_AT >= 4) park_b(buf ^ 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = an[q];
        }
        __syncthreads();
    }
    const int E = ep.E, D = ep.D;
    // ---- cross-covariance: (G L')[i][j] = sum_{d <= j} G[i][d] L[j][d], this lane's d = 4 lg + r ---------------------------
    {
        const int64_t i = row0 + li;
        const int64_t b = (i < M ? i : M - 1) / E;
        const int e = (int)(i - b * E);
        const double *Lb = ep.chol + b * D * D;
        for (int j = 0; j < D; ++j) {
            double p = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 4 * lg + r;
                if (d <= j) p += acc[NT][r] * Lb[j * D + d];
            }
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            if (lg == 0 && i < M) ep.cov_fx[(int64_t)(e * D + j) * ep.es + b * ep.bs_fx] = p * ep.ccov_scale;
        }
    }
    // ---- S = T FX2' ------------------------------------------------------------------------------------------------
    const int64_t s0 = (row0 / E) * E;        // first row of the first trajectory that intersects this tile
    v4d acc2[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
    double f0[4], f1[4], g0[4], g1[4];
    load_frag(s0 + li, 0, f0);
    load_frag(s0 + 16 + li, 0, f1);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        if (ct + 1 < NT) {
            load_frag(s0 + li, ct + 1, g0);
            load_frag(s0 + 16 + li, ct + 1, g1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[ct][r], f0[r], acc2[0], 0, 0, 0);
            acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[ct][r], f1[r], acc2[1], 0, 0, 0);
        }
        if (ct + 1 < NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f0[q] = g0[q];
                f1[q] = g1[q];
            }
        }
    }
    // ---- covariance entries ------------------------------------------------------------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t i2 = s0 + 16 * h + li;
        const int64_t b2 = i2 / E;
        const int e2 = (int)(i2 - b2 * E);
        const double m2 = i2 < M ? ep.mean_rows[i2] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = row0 + lg + 4 * r;
            const int64_t b = i / E;
            const int e = (int)(i - b * E);
            if (i < M && i2 < M && b == b2 && e2 <= e) {
                const int idx = e * E + e2, idt = e2 * E + e;
                const bool use = (e == e2) || ep.emv_broadcast;
                const double em = use ? ep.emv[idx] : 0.0;
                const double mi = ep.mean_rows[i];
                double v = (acc2[h][r] - mi * m2 + em) * ep.cov_scale;
                if (ep.cov_add) v += ep.cov_add[idx];
                ep.cov_f[(int64_t)idx * ep.es + b * ep.bs] = v;
                // mirrored entry: same value, as the per-trajectory kernels produce it (additive terms are symmetric)
                if (e2 != e) ep.cov_f[(int64_t)idt * ep.es + b * ep.bs] = v;
            }
        }
    }
}

template <int NT>
hipError_t launch_cov(const double *A, const double *X, int64_t M, int lda, const CovEpilogue &ep, hipStream_t s) {
    constexpr size_t lds = sizeof(double) * 2 * 16 * (NT * 16 + 16 + 4);
    static thread_local unsigned attr_epoch = 0;   // per-device attribute: set again after a device change
    if (attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_fxwc_cov_mfma<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        attr_epoch = device_epoch();
    }
    hipLaunchKernelGGL((k_fxwc_cov_mfma<NT>), dim3((unsigned)((M + 63) / 64)), dim3(kGemmBlock), lds, s, A, X, M, lda, ep);
    return hipGetLastError();
}

template <int NT, int RT>
hipError_t launch_rt(const double *A, const double *Bm, double *T, int64_t M, int lda, int ldt, hipStream_t s, int KB = NT,
                     int ncb = 1) {
    constexpr size_t lds = sizeof(double) * 2 * 16 * (NT * 16 + 4);
    static thread_local unsigned attr_epoch = 0;   // per-device attribute: set again after a device change
    if (attr_epoch != device_epoch()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_fxwc_mfma<NT, RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        attr_epoch = device_epoch();
    }
    constexpr int rows = 64 * RT;
    hipLaunchKernelGGL((k_fxwc_mfma<NT, RT>), dim3((unsigned)((M + rows - 1) / rows), (unsigned)ncb), dim3(kGemmBlock), lds, s,
                       A, Bm, T, M, lda, ldt, KB);
    return hipGetLastError();
}

// One row tile per wave (64-row workgroups, two of them per CU) measured faster than two at every batch size tried
// (B E = 1e4 ... 4e4 rows: 37 / 90 us against 67 / 133 us): the larger tile needs 506 registers and leaves one wave per
// SIMD alone with its barriers.
template <int NT>
hipError_t launch_nt(const double *A, const double *Bm, double *T, int64_t M, int lda, int ldt, hipStream_t s) {
    return launch_rt<NT, 1>(A, Bm, T, M, lda, ldt, s);
}

}  // namespace

// padded point count the matrix-core path works with, or 0 when N has no instantiation (the generic kernel then forms
// fx Wc itself)
int gemm_mfma_padded(int N) {
    const int nt = (N + 15) / 16;
    return (nt == 8 || nt == 13 || nt == 16) ? nt * 16 : 0;
}

// T [M][ldt] = A [M][lda] * Bm [NP][NP]; lda, ldt >= NP, A zero in columns N..NP-1, Bm zero-padded, rows 16-byte aligned
int launch_fxwc_mfma(int NP, const double *A, const double *Bm, double *T, int64_t M, int lda, int ldt, hipStream_t s) {
    hipError_t e;
    switch (NP / 16) {
        case 8: e = launch_nt<8>(A, Bm, T, M, lda, ldt, s); break;
        case 13: e = launch_nt<13>(A, Bm, T, M, lda, ldt, s); break;
        case 16: e = launch_nt<16>(A, Bm, T, M, lda, ldt, s); break;
        default: set_error("fxwc_mfma: no instantiation for this point count"); return SSMQ_E_UNSUPPORTED;
    }
    return hip_fail(e, "k_fxwc_mfma");
}

// Any point count: T [M][ldt] = A [M][lda] * W, W given as `ncb` column blocks of kBigCols columns, block c stored
// contiguously as [KB 16][kBigCols] (zero-padded); lda >= 16 KB, ldt >= ncb kBigCols.  One launch, blockIdx.y = block: the
// workgroups of consecutive row blocks share a block of W (2.4 MB at N = 1161: L2-resident), A is streamed once per block.
int launch_fxwc_blocks(const double *A, const double *Wblk, double *T, int64_t M, int lda, int ldt, int KB, int ncb,
                       hipStream_t s) {
    if (M <= 0) return SSMQ_OK;
    return hip_fail(launch_rt<kBigCols / 16, 1>(A, Wblk, T, M, lda, ldt, s, KB, ncb), "k_fxwc_mfma(blocks)");
}

// mean_rows[i] = FX[i][:] . wm for caller-supplied integrand values (the evaluation pass computes it itself): one wave per
// row, lanes over the points
__global__ __launch_bounds__(256) void k_row_means(const double *__restrict__ A, const double *__restrict__ wm, int64_t M, int lda,
                                                   int N, double *__restrict__ mean_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    double s = 0.0;
    for (int n = lane; n < N; n += 64) s += A[row * lda + n] * wm[n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) mean_rows[row] = s;
}
int launch_row_means(const double *A, const double *wm, int64_t M, int lda, int N, double *mean_rows, hipStream_t s) {
    if (M <= 0) return SSMQ_OK;
    hipLaunchKernelGGL(k_row_means, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, A, wm, M, lda, N, mean_rows);
    return hip_fail(hipGetLastError(), "k_row_means");
}

// The fused route needs every trajectory that intersects a 16-row tile to lie inside the 32 rows that start at the first
// such trajectory: tile rows [16 a, 16 a + 16), window [E floor(16 a / E), + 32).  True for every E <= 16 except where a
// trajectory would end beyond the window; checked exhaustively over one period (lcm(16, E) rows).
bool fxwc_cov_supported(int E) {
    if (E < 1 || E > 16) return false;
    for (int a = 0; a < E; ++a) {                       // 16 a mod E repeats after E tiles
        const int s0 = (16 * a / E) * E;
        const int last = ((16 * a + 15) / E) * E + E;   // one past the last row of the last intersecting trajectory
        if (last - s0 > 32) return false;
    }
    return true;
}

// cov_f and cov_fx of every trajectory from the integrand values A [M][lda] (M = B E rows, columns N..NP-1 zero),
// X = [Wc | Wcc'] padded to [NP][NP + 16], means mean_rows [M], factors chol [B][D][D]: see k_fxwc_cov_mfma
int launch_fxwc_cov_mfma(int NP, const double *A, const double *X, int64_t M, int lda, const double *mean_rows,
                         const double *chol, const double *emv, int emv_broadcast, const double *cov_add,
                         double cov_scale, double ccov_scale, int E, int D, double *cov_f, double *cov_fx, int64_t es,
                         int64_t bs, int64_t bs_fx, hipStream_t s) {
    if (!fxwc_cov_supported(E) || D < 1 || D > 16) {
        set_error("fxwc_cov_mfma: dimensions not supported");
        return SSMQ_E_UNSUPPORTED;
    }
    if (M <= 0) return SSMQ_OK;
    CovEpilogue ep{mean_rows, chol, emv, cov_add, cov_f, cov_fx, es, bs, bs_fx, E, D, emv_broadcast, cov_scale, ccov_scale};
    hipError_t e;
    switch (NP / 16) {
        case 8: e = launch_cov<8>(A, X, M, lda, ep, s); break;
        case 13: e = launch_cov<13>(A, X, M, lda, ep, s); break;
        case 16: e = launch_cov<16>(A, X, M, lda, ep, s); break;
        default: set_error("fxwc_cov_mfma: no instantiation for this point count"); return SSMQ_E_UNSUPPORTED;
    }
    return hip_fail(e, "k_fxwc_cov_mfma");
}

}  // namespace ssmq
