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
                                                          double *__restrict__ T, int64_t M, int lda, int ldt) {
    constexpr int NP = NT * 16;          // padded N = padded K
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
    for (int kb = 0; kb < NT; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < NT) {
            load_b(kb + 1);
            load_a(kb + 1, an);
        }
        const double *sb = lds + buf * 16 * LB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const double b = sb[(4 * lg + s) * LB + ct * 16 + li];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt][s], b, acc[rt][ct], 0, 0, 0);
            }
        }
        if (kb + 1 < NT) {
            park_b(buf ^ 1);
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

template <int NT, int RT>
hipError_t launch_rt(const double *A, const double *Bm, double *T, int64_t M, int lda, int ldt, hipStream_t s) {
    constexpr size_t lds = sizeof(double) * 2 * 16 * (NT * 16 + 4);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_fxwc_mfma<NT, RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    constexpr int rows = 64 * RT;
    hipLaunchKernelGGL((k_fxwc_mfma<NT, RT>), dim3((unsigned)((M + rows - 1) / rows)), dim3(kGemmBlock), lds, s, A, Bm, T, M,
                       lda, ldt);
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

}  // namespace ssmq
