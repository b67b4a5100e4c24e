// The fused filter time loop by TIME BLOCKS, for the host-array entry point (round 6; ssmq_filter_forward_piped in ssmq_api_study.hip).
//
// forward_pass takes the measurements as a host array and returns host arrays (ssinf.py:66-118).  At BASELINE configs[1] that is
// 8 MB up and 16 MB down around a 34 us kernel: 1.15 ms per call when upload, pass and downloads run back to back.  Here the pass is
// cut into K time blocks; k_filter_range runs the steps [kb, ke) of EVERY trajectory from the state the previous block's launch left
// in the hand-over buffer (the mechanism of the strip schedule, ssmq_filter_chunked.hip: mean, covariance triangle and status word
// as the registers held them, so the filtered moments are the whole-pass kernel's BITS), which lets the host feed block k + 1 and
// drain block k - 1 over PCIe while block k runs.  Launch boundaries order the hand-over; no flags, no spinning.
#include <cstring>
#include "ssmq_filter_fused_kernel.h"

namespace ssmq {
namespace {

template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int OPT, int STU>
__global__ __launch_bounds__(kSmallBlock, (D >= 6 ? 1 : ((D >= 5 && FORM == SSMQ_FORM_SIGMA) ? SSMQ_FUSED_OCC_D5_SIGMA : 2))) void k_filter_range(
    const FusedArgs a, int kb, int ke) {
    if ((int)threadIdx.x >= a.lpw) return;
    fused_pass<D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, STU, true>(a, blockIdx.x, kb, ke, kb == 0, ke == a.T);
}

typedef void (*range_kernel)(const FusedArgs, int, int);
struct RangeEntry {
    int fd, fo, D, Y, ND, NO, form, tp, selo, opt;
    range_kernel k;
    const char *name;
};
// STU as the whole-pass kernel of the shape has it for the Gaussian recursion (0 for the scalar models, -1 decided at run time for
// the others): another value contracts other products into multiply-adds and the last bits differ.
#define SSMQ_RG_ONE(FD, FO, D, Y, N, FORM, TP, SELO, OPT)                                                                \
    {FD, FO, D, Y, N, N, FORM, TP, SELO, OPT, &k_filter_range<D, Y, N, N, FD, FO, FORM, TP, SELO, OPT, (D == 1 ? 0 : -1)>, \
     "k_filter_range<D=" #D ",Y=" #Y ",ND=" #N ",NO=" #N "," #FD "," #FO "," #FORM ",TP=" #TP ",SELO=" #SELO ",OPT=" #OPT ">"}
#define SSMQ_RG(FD, FO, D, Y, N, SELO)                                                                             \
    SSMQ_RG_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 0), SSMQ_RG_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO, 0), \
    SSMQ_RG_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO, 0)
#define SSMQ_RG_FAST(FD, FO, D, Y, N, SELO)                                                                                    \
    SSMQ_RG(FD, FO, D, Y, N, SELO), SSMQ_RG_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 7), SSMQ_RG_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 3), SSMQ_RG_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO, 2), \
    SSMQ_RG_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO, 2)
const RangeEntry kRange[] = {
    // the scalar UNGM filters (BASELINE configs[1] and the six filters of the reference's UNGM studies) ...
    SSMQ_RG(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 1, 1, 2, 0),
    SSMQ_RG(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 1, 1, 3, 0),
    SSMQ_RG(SSMQ_F_UNGM_DYN, SSMQ_F_UNGM_MEAS, 1, 1, 5, 0),
    // ... and the reentry / coordinated-turn shapes of configs[2] and configs[3] with unscented points
    SSMQ_RG_FAST(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0),
    SSMQ_RG_FAST(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0),
    SSMQ_RG_FAST(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 11, 1),
};

}  // namespace

// Doubles of hand-over state per block of 64 trajectories (FusedArgs::hand).
size_t range_hand_doubles(int D) { return ((size_t)D + (size_t)D * (D + 1) / 2 + 1) * 64; }

// 1: the steps [kb, ke) were queued on `s` (dry_run: a kernel exists); 0: no range kernel for this combination; < 0: error.
// Same selection rule as try_launch_fused (the best fast path BOTH handles qualify for, then the dense kernel), Gaussian
// recursion only.  `hand`: range_hand_doubles(D) doubles per block of 64 trajectories.
int try_launch_range(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho, const ssmq_integrand *fo, int sel_obs,
                     int64_t B, int64_t ld, int T, int kb, int ke, const double *d_y, const double *d_m0, const double *d_P0,
                     const double *d_gqg, const double *d_rr, double *d_fm, double *d_fP, int32_t *d_status, double *hand, hipStream_t s,
                     const char **name, bool dry_run, const double *d_ttab_dyn, const double *d_ttab_obs) {
    if (hd->form != ho->form || (hd->tp_nu > 0.0) != (ho->tp_nu > 0.0) || sel_obs < 0 || fd->n_idx > 0) return 0;
    const int tp = hd->tp_nu > 0.0 ? 1 : 0;
    const int both = hd->opt_mask & ho->opt_mask;
    const int plain = !(tp || hd->form == SSMQ_FORM_SIGMA);
    const int want[3] = {(plain && (both & 7) == 7) ? 7 : -1, both & (plain ? 3 : SSMQ_OPT_UT), 0};
    for (int w = 0; w < 3; ++w)
        for (const RangeEntry &e : kRange) {
            if (want[w] < 0) break;
            if (!(e.fd == fd->id && e.fo == fo->id && e.D == hd->D && e.Y == ho->E && e.ND == hd->N && e.NO == ho->N && e.form == hd->form &&
                  e.tp == tp && e.selo == sel_obs && e.opt == want[w]))
                continue;
            if ((fd->id == SSMQ_F_UNGM_DYN || fd->id == SSMQ_F_UNGMNA_DYN) && !d_ttab_dyn && !dry_run) return 0;      // the kernels read the table
            if (name) *name = e.name;
            if (dry_run) return 1;
            if (kb < 0 || ke > T || kb >= ke) return SSMQ_E_ARG;
            FusedArgs a;
            memset(&a, 0, sizeof(a));
            a.y = d_y; a.m0 = d_m0; a.P0 = d_P0; a.fm = d_fm; a.fP = d_fP; a.status = d_status;
            a.c_dyn = hd->d_small; a.c_obs = ho->d_small; a.gqg = d_gqg; a.rr = d_rr; a.B = B; a.ld = ld; a.T = T;
            a.emv_dyn = hd->emv_mode; a.emv_obs = ho->emv_mode; a.nu_dyn = hd->tp_nu; a.nu_obs = ho->tp_nu;
            a.sscale = nullptr; a.student_dof = 0.0;
            a.lpw = 64;
            a.hand = hand;
            fill_fpar(fd, &a.fd);
            fill_fpar(fo, &a.fo);
            a.fd.ttab = d_ttab_dyn;
            a.fo.ttab = d_ttab_obs;
            hipLaunchKernelGGL(e.k, dim3((unsigned)((B + 63) / 64)), dim3(kSmallBlock), 0, s, a, kb, ke);
            const int rc = hip_fail(hipGetLastError(), e.name);
            return rc ? rc : 1;
        }
    return 0;
}

}  // namespace ssmq
