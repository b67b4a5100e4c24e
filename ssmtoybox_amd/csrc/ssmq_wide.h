// Argument block and constant layout of the generic (run-time shape) moment-transform kernel.
#pragma once
#include "ssmq_device.h"

namespace ssmq {

constexpr int kWideBlock = 64;

enum WideMode {
    SSMQ_WIDE_FULL = 0,    // chol + sigma points + built-in integrand + moments
    SSMQ_WIDE_POINTS = 1,  // chol + sigma points, written out for a host-evaluated integrand
    SSMQ_WIDE_FX = 2,      // moments from caller-supplied integrand values
    SSMQ_WIDE_EVAL = 3     // chol + sigma points + built-in integrand, values written out (first pass of the GEMM route)
};

// Second constant block of a transform, natural (row-major, untransposed) layout.
struct WideLayout {
    int32_t xiT, wm, Wc, Wcc, emv, iK, total;
};
__host__ __device__ constexpr inline WideLayout wide_layout(int D, int E, int N, int form) {
    WideLayout c{};
    c.xiT = 0;                                              // [N][D]
    c.wm = c.xiT + D * N;                                   // [N]
    c.Wc = c.wm + N;                                        // [N][N] row-major, or [N] diagonal (SIGMA)
    c.Wcc = c.Wc + (form == SSMQ_FORM_SIGMA ? N : N * N);   // [D][N]
    c.emv = c.Wcc + D * N;                                  // [E][E]
    c.iK = c.emv + E * E;                                   // [N][N] row-major
    c.total = c.iK + N * N;
    return c;
}

struct WideArgs {
    int32_t D, E, N, form, mode, fid, time_stride, emv_mode;
    double tp_nu, cov_scale, ccov_scale;
    const double *consts;   // WideLayout block
    int64_t consts_stride;  // doubles between the blocks of consecutive trajectories (0: one block for the batch)
    const double *cov_add;  // [E*E] or null
    // inputs: element e of trajectory b at ptr[e * es_in + b * bs_*]
    const double *mean, *cov, *time;
    int64_t es_in, bs_mean, bs_cov;
    // outputs
    double *mean_f, *cov_f, *cov_fx;
    int64_t es_out, bs_mf, bs_cf, bs_cfx;
    int32_t *status;
    // split entry points (reference layout, trajectory-major)
    double *x_out, *chol_out;
    const double *chol_in, *fx_in, *x_in;
    // GEMM route (ssmq_gemm_mfma.hip): integrand values as rows of pitch fx_ld (0 = N, unpadded), written by the EVAL
    // pass (fx_out, zero beyond column N) and read back by the FX pass together with t_in = fx Wc
    int64_t fx_ld;
    double *fx_out;
    const double *t_in;
    double *mrow_out;       // k_eval_wave: transformed means as rows b E + e, for the GEMM epilogue
    int32_t wave_k;         // k_apply_wave: trajectories side by side in one wave (set by the launcher)
    // k_eval_wave for k_bq_stream: FX in FRAGMENT order instead of rows - blocks of fx_frag trajectories (64 rows, the last ones
    // unused), each [4 row tiles][fx_ld / 16 k-blocks][64 lanes][4]: value (row lr of the block, point n) at lane (lr & 15) +
    // 16 ((n & 15) >> 2), slot n & 3 of tile (lr >> 4, n >> 4) - what a lane of the f64 matrix instruction takes as ONE 32-byte
    // read, the wave's reads contiguous.  0: rows.
    int32_t fx_frag;
    FPar fp;
};

size_t wide_lds_bytes(int D, int E, int N);
hipError_t launch_apply_wide(const WideArgs &a, int64_t B, hipStream_t s);   // FULL mode, N <= 64: k_apply_wave inside
bool wide_full_uses_wave(int D, int E, int N);
// ... or, with one constant block for the whole batch and 8 < N <= 64, on the matrix cores (ssmq_apply_tile.hip)
bool wide_full_uses_tile(int D, int E, int N);
hipError_t launch_apply_tile(const WideArgs &a, int64_t B, hipStream_t s);
bool tile_pitch_ok(const WideArgs &a);
// ... its condition on the plane pitch alone (what the route / kernel-name query of ssmq_api.hip can know)
inline bool tile_ld_ok(int64_t ld) { return ld < ((int64_t)1 << 29); }
// evaluation pass of the two-pass matrix-core route, one wave per trajectory (fx_out, chol_out, mean_f, mrow_out, status)
hipError_t launch_eval_wave(const WideArgs &a, int64_t B, hipStream_t s);

}  // namespace ssmq
