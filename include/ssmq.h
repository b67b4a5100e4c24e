/*
 * ssmq.h - C ABI of the MI355X-native sigma-point / Bayesian-quadrature moment-transform library (libssmq.so).
 *
 * This is the drop-in boundary for ONE path of jacobnzw/SSMToybox: the moment transform that the filters in
 * ssmtoybox/ssinf.py call twice per time step, plus the quadrature-weight construction that feeds it.  The reference
 * is pure Python and has no FFI; the entry points below are what a ctypes binding for that path would bind, and each
 * cites the reference interface (file:line under the reference tree) it replaces.  INTEGRATION.md shows the
 * reference-side stub.
 *
 * Conventions
 *   - plain C types only; all floating point is IEEE fp64 (`double`); integers are int32 unless stated.
 *   - host arrays are C-contiguous in the reference's NumPy layout ("AoS": trajectory-major, e.g. cov[b][i][j]).
 *   - device arrays handed to the *_dev entry points are in the library's HBM layout ("SoA planes"): element e of
 *     trajectory b lives at ptr[e * ld + b]; ld >= B is the plane pitch in doubles.  Matrices are row-major in e
 *     (e = i * ncol + j).  ssmq_aos_to_soa / ssmq_soa_to_aos convert on the device.
 *   - return value: 0 ok; > 0 = 1 + index of the first batch item whose input covariance is not positive definite
 *     (the reference raises numpy.linalg.LinAlgError there: bq/bqmtran.py:98, mtran.py:139); < 0 error
 *     (SSMQ_E_*); ssmq_last_error() gives the text.  No exception crosses the ABI.
 *   - a transform handle is bound to the device that was current when it was created.
 *     Threads (round 5): every calling thread has a CONTEXT of its own - one HIP stream and the caches that belong to a stream
 *     (grow-only workspaces, pinned staging blocks, captured launch graphs).  Calls of different threads on different handles
 *     run concurrently, on the host and - stream by stream - on the device.  A handle is locked for the duration of every
 *     entry point that takes it (two handles in address order), so calls of several threads on the SAME handle are safe and
 *     run one after the other; a thread that picks up a handle last used by another thread first waits for that thread's
 *     stream, so constants uploaded or buffers built there are complete.  The *_dev entry points queue on the CALLING
 *     thread's stream and return; ssmq_sync() / events / copies act on that stream.  Device buffers the caller hands from one
 *     thread to another are the caller's to order (ssmq_sync() in the thread that queued the work), as with any per-thread
 *     stream.  A thread that ends returns its context (stream and caches intact) to a pool for the next new thread.  The
 *     ssmq_comm_* entry points belong to one thread.  One process per GPU is the intended deployment: HIP's
 *     current device is per thread and starts at 0, so a thread's first call moves it to the device of the last
 *     ssmq_set_device(); a context that finds its thread on another device drops its caches and binds there.
 *   - there is NO CPU fallback anywhere behind this ABI: without a usable gfx950 device every compute entry point
 *     returns SSMQ_E_HIP.
 */
#ifndef SSMQ_H
#define SSMQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSMQ_VERSION 102

#define SSMQ_OK 0
#define SSMQ_E_ARG (-1)          /* bad argument (null pointer, size out of range, unknown id) */
#define SSMQ_E_HIP (-2)          /* HIP runtime error / no device */
#define SSMQ_E_UNSUPPORTED (-3)  /* combination not implemented */
#define SSMQ_E_NOMEM (-4)

#define SSMQ_MAX_FPAR 16 /* doubles of integrand constants */
#define SSMQ_MAX_FIDX 16 /* state-index entries (= SSMQ_MAX_DIM: any sub-state of any state) */
#define SSMQ_MAX_DIM 16  /* D, E (input / output dimension of a transform) */
#define SSMQ_MAX_PTS 4096 /* N (sigma points) */

/* Integrands: closed-form dynamics / measurement functions of ssmtoybox/ssmod.py evaluated on the device so sigma
 * points never leave the GPU (bq/bqmtran.py:132-156 evaluates a Python callable column by column). */
enum ssmq_integrand_id {
    SSMQ_F_UNGM_DYN = 1,          /* ssmod.py:268-269   par: -            in 1 out 1, uses time          */
    SSMQ_F_UNGM_MEAS = 2,         /* ssmod.py:1060-1061 par: -            in 1 out 1                     */
    SSMQ_F_UNGMNA_DYN = 3,        /* ssmod.py:299-300   input [x, q]      in 2 out 1, uses time          */
    SSMQ_F_UNGMNA_MEAS = 4,       /* ssmod.py:1085-1086 input [x, r]      in 2 out 1                     */
    SSMQ_F_PENDULUM_DYN = 5,      /* ssmod.py:357-358   par: dt           in 2 out 2                     */
    SSMQ_F_PENDULUM_MEAS = 6,     /* ssmod.py:1114-1115                   in 1 out 1                     */
    SSMQ_F_REENTRY1D_DYN = 7,     /* ssmod.py:424-427   par: dt           in 3 out 3                     */
    SSMQ_F_RANGE_MEAS = 8,        /* ssmod.py:1147-1149                   in 1 out 1                     */
    SSMQ_F_REENTRY2D_DYN = 9,     /* ssmod.py:530-564   par: dt           in 5 out 5                     */
    SSMQ_F_RADAR2D_MEAS = 10,     /* ssmod.py:1227-1252 par: loc_x loc_y  in 2 out 2                     */
    SSMQ_F_CT_DYN = 11,           /* ssmod.py:675-690   par: dt           in 5 out 5                     */
    SSMQ_F_BEARING_MEAS = 12,     /* ssmod.py:1189-1195 par: S x (sx, sy) in 2 out S (S = n_par / 2 <= 8) */
    SSMQ_F_CTRS_DYN = 13,         /* ssmod.py:755-774   par: dt, input [x(5), q(2)]  in 7 out 5          */
    SSMQ_F_CV_DYN = 14,           /* ssmod.py:839-846   par: dt           in 4 out 4                     */
    SSMQ_F_REENTRY2D_BIAS_DYN = 15, /* this build's synthetic 6-D case: reentry-2D + pass-through state, in 6 out 6 */
    SSMQ_F_SMOOTH10D_DYN = 16      /* this build's synthetic 10-D case (the reference has no model above 7 inputs; Bayes-Sard
                                      quadrature at D = 10 is BASELINE config 5): out[i] = sin(x[i]) + x[5+i]^2,
                                      out[5+i] = x[5+i] cos(x[i]), i = 0..4; in 10 out 10; no state index */
};

/* One integrand = id + constants + optional sub-state selection (MeasurementModel.state_index, ssmod.py:990-991):
 * if n_idx > 0 the integrand sees x[idx[0]], x[idx[1]], ... of the D-dimensional sigma point. */
typedef struct ssmq_integrand {
    int32_t id;
    int32_t n_par;
    int32_t n_idx;
    int32_t reserved;
    double par[SSMQ_MAX_FPAR];
    int32_t idx[SSMQ_MAX_FIDX];
} ssmq_integrand;

/* Form of the moment equations. */
enum ssmq_form {
    SSMQ_FORM_BQ = 0,    /* bq/bqmtran.py:158-223: mean = fx wm; cov = fx Wc fx' - mean mean' + emv; ccov = fx Wcc' L' */
    SSMQ_FORM_SIGMA = 1, /* mtran.py:141-149: centred, diagonal Wc: cov = dfx diag(wc) dfx'; ccov = dfx diag(wc) (x-m)' */
    SSMQ_FORM_TAYLOR1 = 2 /* mtran.py:49-59 (LinearizationTransform): mean = f(m); J = df/dx(m); ccov = J cov; cov = ccov J'.
                             Handles of this form come from ssmq_transform_create_linear only */
};

/* How the expected model variance enters the covariance (bq/bqmtran.py:198 `model_var * I_out`). */
enum ssmq_emv_mode {
    SSMQ_EMV_DIAG = 0,      /* I_out = eye(E): only the diagonal of the (E, E) emv matrix is added            */
    SSMQ_EMV_BROADCAST = 1  /* I_out = eye(1) with E > 1 (ssinf.py StudentProcessKalman builds its transforms so):
                               the whole (E, E) matrix is added                                                 */
};

typedef struct ssmq_transform ssmq_transform; /* opaque; replaces the attribute carriers BQTransform / Model
                                                 (bq/bqmtran.py:11-58, bq/bqmod.py:15-106) on the device side */

/* ---- library / device ------------------------------------------------------------------------------------- */
int ssmq_version(void);
const char *ssmq_last_error(void);
int ssmq_device_count(int *n);
int ssmq_set_device(int device);
/* The device the library's stream and caches live on (the calling thread's current device on first use), -1 on error.
 * The HIP current device is PER THREAD: a caller that enters the library from a second thread selects this device there
 * first (ssmq_set_device), otherwise the library would move to that thread's default device. */
int ssmq_current_device(void);
int ssmq_device_name(char *buf, int len);
/* PCI bus id ("0000:c1:00.0", len >= 13) of that device: what tells N ranks of one node apart in a scaling record. */
int ssmq_device_pci_bus_id(char *buf, int len);

/* ---- device memory, layout conversion, timing (plumbing) ---------------------------------------------------- */
int ssmq_malloc(void **dptr, size_t bytes);
int ssmq_free(void *dptr);
int ssmq_memcpy_h2d(void *dst, const void *src, size_t bytes);
int ssmq_memcpy_d2h(void *dst, const void *src, size_t bytes);
int ssmq_memcpy_d2d(void *dst, const void *src, size_t bytes);
int ssmq_memset(void *dptr, int value, size_t bytes);
int ssmq_sync(void);
/* AoS [B][n] (reference layout) <-> SoA planes [n][ld] (HBM layout), both on the device. */
int ssmq_aos_to_soa(const double *d_aos, double *d_soa, int n, int64_t B, int64_t ld);
int ssmq_soa_to_aos(const double *d_soa, double *d_aos, int n, int64_t B, int64_t ld);
/* Host arrays in the reference's study layout (n_elem..., n_outer, B) - dim_y x T x B measurements, D x T x B means,
 * D x D x T x B covariances of forward_pass / the Monte-Carlo loops (ssinf.py:66-118, research/tpq/tpq_base.py:175-192) -
 * to / from the filters' time-major planes [n_outer][n_elem][ld] in HBM.  Transfers go through the library's pinned
 * staging block in chunks; padding lanes (B .. ld) are zero-filled on upload.  Synchronous. */
int ssmq_upload_planes(const double *host, int n_outer, int n_elem, int64_t B, int64_t ld, double *d_planes);
int ssmq_download_planes(const double *d_planes, int n_outer, int n_elem, int64_t B, int64_t ld, double *host);
/* HIP events on the library's stream (bench.py times kernels with these). */
int ssmq_event_create(void **ev);
int ssmq_event_destroy(void *ev);
int ssmq_event_record(void *ev);
int ssmq_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
/* index of the first non-zero entry of a device status vector, or -1 */
int ssmq_status_first(const int32_t *d_status, int64_t B, int64_t *first);

/* ---- quadrature weights (theta-batched, computed on the device) --------------------------------------------- */
/*
 * GP quadrature weights for P kernel-parameter rows at once.
 * Replaces GaussianProcessModel.bq_weights (bq/bqmod.py:495-523) together with RBFGauss.eval / exp_x_kx / exp_x_xkx /
 * exp_x_kxkx / exp_x_kxx / exp_xy_kxy (bq/bqkern.py:329-424), utils.maha (utils.py:385-409) and
 * Kernel.eval_inv_dot / _cho_inv (bq/bqkern.py:38-64, 96-120).
 *   xi  [D*N]      unit sigma points, row-major (D, N)            (host)
 *   par [P*(1+D)]  rows [alpha, ell_1..ell_D]                     (host)
 * outputs (host; any may be NULL): wm [P*N], Wc [P*N*N], Wcc [P*D*N], iK [P*N*N] (scaling=False inverse),
 *   q [P*N], Q [P*N*N], R [P*D*N], model_var [P], integral_var [P], status [P] (1 = K + jitter I not PD).
 */
int ssmq_weights_gp(int D, int N, const double *xi, const double *par, int P, double jitter,
                    double *wm, double *Wc, double *Wcc, double *iK, double *q, double *Q, double *R,
                    double *model_var, double *integral_var, int32_t *status);
/*
 * The kernel-level methods of the reference as entry points of their own (the weights entry points build K, its inverse
 * and the expectations in one kernel and need none of these):
 *   ssmq_rbf_eval      RBFGauss.eval (bq/bqkern.py:329-343, utils.maha utils.py:385-409): K [P][N1][N2] between the
 *                      point sets x1 [D*N1], x2 [D*N2] (NULL: x1); scaling = use alpha; diag: N1 values k(x1_i, x2_i)
 *                      through the difference form of the reference's diag branch.
 *   ssmq_rbf_factor    Kernel.eval_chol (bq/bqkern.py:122-142): chol [P][N][N], lower factor of K + jitter I, and / or
 *                      Kernel.eval_inv_dot / _cho_inv (:38-64, 96-120): iK [P][N][N] = sym((K + jitter I)^-1 rhs), rhs [N][N]
 *                      or NULL (= I; the reference symmetrises whatever it solved for, so a right-hand side is square);
 *                      chol or iK may be NULL; status [P] / return value as ssmq_weights_gp.
 *   ssmq_rbf_exp_kxkx  RBFGauss.exp_x_kxkx for two (possibly different) parameter rows (:366-415): Q [N][N].
 */
int ssmq_rbf_eval(int D, int N1, const double *x1, int N2, const double *x2, const double *par, int P, int scaling,
                  int diag, double *K);
int ssmq_rbf_factor(int D, int N, const double *x, const double *par, int P, int scaling, double jitter,
                    const double *rhs, double *chol, double *iK, int32_t *status);
int ssmq_rbf_exp_kxkx(int D, int N, const double *x, const double *par0, const double *par1, int scaling, double *Q);
/*
 * Student-t process model: the same weights as ssmq_weights_gp (StudentTProcessModel inherits bq_weights,
 * bq/bqmod.py:1060-1130); model_var / integral_var are the GP values, which the t-process rescales with the integrand
 * values at transform time (bq/bqmod.py:1132-1190; tp_nu / tp_iK of ssmq_transform_create).
 */
int ssmq_weights_tp(int D, int N, const double *xi, const double *par, int P, double jitter, double *wm, double *Wc,
                    double *Wcc, double *iK, double *q, double *Q, double *R, double *model_var, double *integral_var,
                    int32_t *status);
/*
 * Bayes-Sard quadrature weights.  Replaces BayesSardModel.bq_weights (bq/bqmod.py:893-992) with _exp_x_px / _exp_x_xpx
 * / _exp_x_pxpx / _exp_x_kxpx (bq/bqmod.py:635-797) and utils.vandermonde (utils.py:478-502).
 *   mulind [D*NB]  multi-indices, row-major (D, NB), NB <= N; NB == N selects the unisolvent branch (:952-961).
 * outputs as ssmq_weights_gp (iK, q, Q, R as there; Q / R are only produced in the NB < N branch).
 */
int ssmq_weights_bs(int D, int N, const double *xi, const double *par, int P, double jitter,
                    const int32_t *mulind, int NB,
                    double *wm, double *Wc, double *Wcc, double *iK, double *q, double *Q, double *R,
                    double *model_var, double *integral_var, int32_t *status);

/*
 * The polynomial expectations behind the Bayes-Sard weights as entry points of their own (ssmq_weights_bs forms them
 * inline), for multi-indices mulind [D*NB] (row-major (D, NB)); any output may be NULL:
 *   px [NB]       BayesSardModel._exp_x_px   (bq/bqmod.py:635-662)   E[p_q(x)]
 *   xpx [D*NB]    BayesSardModel._exp_x_xpx  (:664-698)              E[x_e p_q(x)], with the reference's alpha_e factor
 *   pxpx [NB*NB]  BayesSardModel._exp_x_pxpx (:700-731)              E[p_r(x) p_q(x)]
 *   kxpx [N*NB]   BayesSardModel._exp_x_kxpx (:733-797)              E[k(x, x_n) p_q(x)] for the points x [D*N], kernel
 *                 parameters par [1+D] (the reference's `ell = sqrt_inv_lam ** -2` kept as written)
 *   vand [N*NB]   utils.vandermonde (utils.py:478-502)               p_q(x_n)
 * px / xpx / pxpx are integer arithmetic on the multi-indices (host code, no device needed); kxpx / vand run on the
 * device.
 */
int ssmq_bs_moments(int D, int N, const double *x, const double *par, const int32_t *mulind, int NB, double *px,
                    double *xpx, double *pxpx, double *kxpx, double *vand);

/*
 * Expected model variance and integral variance of the Bayes-Sard model as BayesSardModel.exp_model_variance /
 * integral_variance compute them (bq/bqmod.py:995-1050; the length-scale sweeps of research/bsq/bsq_ungm.py:244-282):
 * the general formulas  alpha^2 (1 - tr(Q iK) + tr(B (V' iK V)^-1)),  kbar - q' iK q + b' (V' iK V)^-1 b  for every
 * point set, with NO jitter on V' iK V - which is not what bq_weights() returns beside the weights (ssmq_weights_bs).
 * theta-batched: par [P][1+D], outputs model_var [P], integral_var [P] (either may be NULL), status [P] as above.
 */
int ssmq_variances_bs(int D, int N, const double *xi, const double *par, int P, double jitter, const int32_t *mulind,
                      int NB, double *model_var, double *integral_var, int32_t *status);

/* ---- transform handle --------------------------------------------------------------------------------------- */
/*
 * Upload the constants of one moment transform (what BQTransform.__init__ / SigmaPointTransform.__init__ keep as
 * tf.wm / tf.Wc / tf.Wcc / tf.model.points / tf.model.model_var / tf.model.iK / tf.model.nu: bq/bqmtran.py:55-58,
 * 306-310, 387-392; mtran.py:165-168, 226-232).
 *   xi [D*N]; wm [N]; Wc: [N*N] (BQ form) or the diagonal [N] (SIGMA form); Wcc [D*N] (BQ form; NULL for SIGMA);
 *   emv [E*E] expected model variance as an (E, E) matrix (scalar model_var -> model_var * ones; NULL = 0);
 *   tp_nu > 0 selects the Student-t process covariance (bq/bqmtran.py:394-415, bq/bqmod.py:1132-1160) and needs
 *   tp_iK [N*N].
 * Returns NULL on error.
 */
ssmq_transform *ssmq_transform_create(int D, int E, int N, int form, const double *xi, const double *wm,
                                      const double *Wc, const double *Wcc, const double *emv, int emv_mode,
                                      double tp_nu, const double *tp_iK);
/* Replace the constants in place (research code overwrites tf.wm / tf.Wc / tf.Wcc / model_var:
 * research/tpq/tpq_ungm.py:114-124, research/bsq/bsq_tracking.py:276-281).  Any pointer may be NULL = keep. */
int ssmq_transform_update(ssmq_transform *h, const double *xi, const double *wm, const double *Wc, const double *Wcc,
                          const double *emv, int emv_mode, double tp_nu, const double *tp_iK);
/* The linearisation transform of ExtendedKalman (mtran.py:49-59: LinearizationTransform.apply; ssinf.py:347-357): no points and
 * no weights - mean_f = f(mean), cov_fx = J cov, cov_f = cov_fx J' with the model's own Jacobian (the seven models whose
 * dyn_fcn_dx / meas_fcn_dx the reference implements: UNGM, UNGM with non-additive noise, pendulum, constant velocity; every
 * other integrand: SSMQ_E_UNSUPPORTED at apply time, where the reference's Jacobian is None).  The handle goes wherever a
 * transform handle goes (ssmq_apply_batch[_dev], the filter / smoother entry points: time loop as a launch loop). */
ssmq_transform *ssmq_transform_create_linear(int D, int E);
void ssmq_transform_destroy(ssmq_transform *h);
int ssmq_transform_dims(const ssmq_transform *h, int *D, int *E, int *N);

/* ---- the moment transform, batched over independent trajectories --------------------------------------------- */
/*
 * B moment transforms in one launch.  Replaces BQTransform.apply (bq/bqmtran.py:60-109) / SigmaPointTransform.apply
 * (mtran.py:105-149) for the built-in integrands: Cholesky of each cov, sigma points x = mean + L xi, integrand,
 * weighted mean / covariance / cross-covariance.
 * Host version, reference layout: mean [B*D], cov [B*D*D], time [B] or [1] (time_stride 1 or 0; the reference passes
 * np.atleast_1d(k - 1): ssinf.py:276-277), outputs mean_f [B*E], cov_f [B*E*E], cov_fx [B*E*D], status [B] (may be NULL).
 * B == 1 serves the drop-in apply().
 */
int ssmq_apply_batch(ssmq_transform *h, const ssmq_integrand *f, int64_t B, const double *mean, const double *cov,
                     const double *time, int time_stride, double *mean_f, double *cov_f, double *cov_fx,
                     int32_t *status);
/* Device version, SoA planes with pitch ld; d_time [B] or [1]; d_status [B] int32 (required).  Asynchronous on the
 * library stream.  Only the lower triangle of each input covariance is read (as LAPACK dpotrf 'L' does). */
int ssmq_apply_batch_dev(ssmq_transform *h, const ssmq_integrand *f, int64_t B, int64_t ld, const double *d_mean,
                         const double *d_cov, const double *d_time, int time_stride, double *d_mean_f,
                         double *d_cov_f, double *d_cov_fx, int32_t *d_status);
/* Which kernel ssmq_apply_batch_dev would run for this (transform, integrand): writes its name (for profiles). */
int ssmq_apply_kernel_name(const ssmq_transform *h, const ssmq_integrand *f, char *buf, int len);

/*
 * Arbitrary Python integrand f: the sigma points are formed on the device, f is evaluated by the caller, and the
 * weighted reductions run on the device (bq/bqmtran.py:97-101 and :104-107 around the `_fcn_eval` call at :102).
 *   ssmq_sigma_points_batch: x [B*D*N] (each (D, N) row-major), chol [B*D*D] lower factors.
 *   ssmq_apply_fx_batch:     fx [B*E*N]; `mean` / `x` are only read for the SIGMA form (centred cross-covariance).
 */
int ssmq_sigma_points_batch(ssmq_transform *h, int64_t B, const double *mean, const double *cov, double *x,
                            double *chol, int32_t *status);
int ssmq_apply_fx_batch(ssmq_transform *h, int64_t B, const double *chol, const double *mean, const double *x,
                        const double *fx, double *mean_f, double *cov_f, double *cov_fx);

/*
 * T = FX Wc for M = B E rows of integrand values that are already on the device: the GEMM-shaped stage of
 * fx Wc fx' (bq/bqmtran.py:199) for large point sets, on the matrix cores (v_mfma_f64_16x16x4_f64).  Row r of d_fx is
 * the integrand output e of trajectory b with r = b E + e, pitch ld_fx >= NP doubles (even), columns N..NP-1 ZERO,
 * NP = N rounded up to 16; d_t gets the same shape with pitch ld_t.  *n_padded (may be NULL) returns NP, 0 when the
 * handle has no instantiation (then SSMQ_E_UNSUPPORTED; ssmq_apply_batch* fall back to the generic kernel themselves).
 * Asynchronous on the library stream.
 */
int ssmq_fxwc_batch_dev(ssmq_transform *h, int64_t M, const double *d_fx, int64_t ld_fx, double *d_t, int64_t ld_t,
                        int *n_padded);

/* ---- callers of the path kept on the device (filter recursion) ------------------------------- */
/*
 * Gaussian measurement update for B trajectories (ssinf.py:297-323): gain = (P_y^-1 P_yx)' by Cholesky,
 * m = m_pr + gain (y - y_mean), P = P_pr - gain P_y gain' (left unsymmetrised as the reference does, :323).
 * SoA planes, pitch ld.  P_yx is (Y, D) as returned by the obs transform.  In-place (m_fi == m_pr etc.) is allowed.
 */
int ssmq_kalman_update_dev(int D, int Y, int64_t B, int64_t ld, const double *d_m_pr, const double *d_P_pr,
                           const double *d_y_mean, const double *d_P_y, const double *d_P_yx, const double *d_y,
                           double *d_m_fi, double *d_P_fi, int32_t *d_status);
/*
 * Forward pass of an additive-noise Gaussian filter for B trajectories and T steps (ssinf.py:66-118, 254-323):
 * per step k = 1..T: dyn transform at time k-1, += GQG', obs transform at time k-1, += R, measurement update.
 *   d_y [T][Y][ld]; d_m0 [D][ld], d_P0 [D*D][ld] (read only); GQG [D*D], R [Y*Y] host;
 *   outputs d_fm [T][D][ld], d_fP [T][D*D][ld]; d_status [ld] (0 ok, else 1 + first failing step).
 * Asynchronous on the library stream.  Common (models, shape, form) combinations run as ONE fused kernel with the
 * filter state in registers; every other combination replays the 3 T launches of the loop as a hipGraph.
 */
int ssmq_filter_forward_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                            const ssmq_integrand *f_obs, int64_t B, int64_t ld, int T, const double *d_y,
                            const double *d_m0, const double *d_P0, const double *GQG, const double *R,
                            double *d_fm, double *d_fP, int32_t *d_status);

/*
 * Forward pass of a Gaussian filter whose transition and / or measurement model takes its noise as an argument
 * (ssinf.py:271-272 and 282-283: mean <- [mean; noise_mean], cov <- blockdiag(cov, noise_cov) before each transform;
 * ssinf.py:294-295: cross-covariances trimmed to the first dim_state columns).
 *   dq > 0: non-additive dynamics, q_mean [dq], q_cov [dq*dq] host, h_dyn is a (dim_state + dq -> dim_state) transform;
 *   dq = 0: additive dynamics, q_cov is G Q G' [dim_state^2] (or NULL for none), q_mean ignored.
 *   dr likewise for the measurement model (h_obs: dim_state + dr -> Y; dr = 0: r_cov is R [Y*Y]).
 * Buffers and status as ssmq_filter_forward_dev.  Synchronous; one fused kernel where this (models, shapes, form)
 * combination has an instantiation (UNGMNA with 4 / 5 points, CTRS + radar with unscented points), else a launch loop
 * of 5 T launches.
 */
int ssmq_filter_forward_aug_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                const ssmq_integrand *f_obs, int dim_state, int64_t B, int64_t ld, int T,
                                const double *d_y, const double *d_m0, const double *d_P0, const double *q_mean,
                                const double *q_cov, int dq, const double *r_mean, const double *r_cov, int dr,
                                double *d_fm, double *d_fP, int32_t *d_status);

/*
 * Forward pass + Rauch-Tung-Striebel backward pass (ssinf.py:120-147, 325-344): as ssmq_filter_forward_dev, and
 * additionally d_sm [T][D][ld], d_sP [T][D*D][ld] smoothed moments.  The reference's indexing is kept: the recursion
 * starts at the last filtered estimate and leaves the last two smoothed steps equal to the filtered ones.  Synchronous.
 * The forward pass that keeps the predictive moments is one kernel for the UNGM / pendulum / reentry (5, 2, 11) shapes,
 * else the launch loop (hipGraph).
 */
/* The backward pass alone (ssinf.py:120-147, 325-344) over moments the caller kept from its own forward pass (the
 * marginalised filter drives its forward pass from the host): filtered d_fm [T][D][ld], d_fP [T][D*D][ld], predictive
 * d_pm, d_pP (element k = the prediction INTO step k), dynamics cross-covariance d_pC [T][D*D][ld]; outputs d_sm, d_sP;
 * d_status [ld] gets bit 30 set where a predictive covariance is not positive definite.  D <= 7.  Synchronous. */
int ssmq_rts_backward_dev(int D, int64_t B, int64_t ld, int T, const double *d_fm, const double *d_fP, const double *d_pm,
                          const double *d_pP, const double *d_pC, double *d_sm, double *d_sP, int32_t *d_status);

int ssmq_filter_smooth_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                           const ssmq_integrand *f_obs, int64_t B, int64_t ld, int T, const double *d_y,
                           const double *d_m0, const double *d_P0, const double *GQG, const double *R, double *d_fm,
                           double *d_fP, double *d_sm, double *d_sP, int32_t *d_status);
/*
 * The same for models that take their noise as an argument (arguments as ssmq_filter_forward_aug_dev): backward_pass of
 * the reference does not look at the model (ssinf.py:120-147, 325-344); the cross-covariance it uses is the one that
 * _time_update cut back to the state columns (ssinf.py:294-295).  Synchronous.
 */
int ssmq_filter_smooth_aug_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                               const ssmq_integrand *f_obs, int dim_state, int64_t B, int64_t ld, int T,
                               const double *d_y, const double *d_m0, const double *d_P0, const double *q_mean,
                               const double *q_cov, int dq, const double *r_mean, const double *r_cov, int dr,
                               double *d_fm, double *d_fP, double *d_sm, double *d_sP, int32_t *d_status);

/*
 * theta-batched filter step for GP-quadrature transforms whose kernel parameters are re-drawn per item - the inner
 * evaluation of the marginalised filter (ssinf.py:1117-1151 _state_posterior_moments, :1153-1198 _param_log_likelihood;
 * re-weighting bq/bqmtran.py:93-95).  For item i = 0..P-1:
 *   weights(par_dyn[i]) -> dyn transform at `time` (+ GQG) -> weights(par_obs[i]) -> obs transform at `time` (+ R)
 *   -> Kalman update with y -> post_mean[i][D], post_cov[i][D*D], loglik[i] = log N(y | y_mean, P_y).
 * h_dyn / h_obs supply shapes, sigma points and the emv mode (their own weights are not used; GP models only).
 * par_dyn: host [P][1+Din], par_obs [P][1+D] = [alpha, ell_1..] (already exponentiated), Din = input dimension of h_dyn.
 * Additive dynamics: Din = D.  Dynamics that take their noise as an argument: h_dyn is a (D + dq) -> D transform and the
 * caller passes the augmented moments [mean; q_mean], blockdiag(cov, Q) (ssinf.py:1174-1176) with GQG = NULL; the
 * measurement model is additive in either case (the reference builds that transform on dim_state inputs, ssinf.py:1288).
 * mean [Din] / cov [Din*Din] when shared_state = 1, else [P][Din] / [P][Din*Din]; y [Y] when shared_y = 1, else [P][Y].
 * GQG [D*D], R [Y*Y] host or NULL.
 * status[i] (may be NULL): bit 0 K_dyn not positive definite, bit 1 K_obs, bit 2 cov (Cholesky in the dyn transform),
 * bit 3 predictive cov, bit 4 P_y.  Returns 0, or 1 + index of the first item with a nonzero status.  Synchronous;
 * the weights never leave the device (k_weights writes per-item constant blocks that the generic transform kernel reads in place).
 */
int ssmq_gp_theta_step(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                       const ssmq_integrand *f_obs, int64_t P, const double *par_dyn, const double *par_obs,
                       double jitter, const double *mean, const double *cov, int shared_state, const double *y,
                       int shared_y, double time, const double *GQG, const double *R, double *post_mean,
                       double *post_cov, double *loglik, int32_t *status);

/* ssmq_gp_theta_step with a time of its own per item (times [P]): items of different time steps in one call. */
int ssmq_gp_theta_step_times(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                             const ssmq_integrand *f_obs, int64_t P, const double *par_dyn, const double *par_obs,
                             double jitter, const double *mean, const double *cov, int shared_state, const double *y,
                             int shared_y, const double *times, const double *GQG, const double *R, double *post_mean,
                             double *post_cov, double *loglik, int32_t *status);

/*
 * Batched Laplace step of the marginalised filter (ssinf.py:1243-1273 _param_posterior_moments: one scipy BFGS run per
 * trajectory and time step in the reference; research/tpq/tpq_base.py:175-192 loops over trajectories): B independent
 * minimisations of  -log N(y_b | theta-conditioned step)  -  log N(theta | prior_mean_b, prior_cov_b)  in lock step - every
 * round ONE ssmq_gp_theta_step of (unfinished trajectories) x (param_dim + 1) items (objective + forward-difference gradient,
 * step fd_step as scipy's default 1.4901161193847656e-08).  The optimiser restates SciPy 1.15's BFGS with its first line
 * search (MINPACK-2 DCSRCH, c1 = 1e-4, c2 = 0.9, gtol 1e-5 on the max-norm, maxiter 200 param_dim).
 * mean [B][Din], cov [B][Din*Din] (augmented as for ssmq_gp_theta_step), y [B][Y], prior_mean [B][P], prior_cov [B][P*P],
 * theta [B][P]: start points in (normally the prior means), minimisers out; hess_inv [B][P*P] the BFGS inverse Hessian.
 * status[b]: 0 converged, SSMQ_BFGS_MAXITER / _PRECISION_LOSS / _NAN = scipy's warnflag 1 / 2 / 3 (theta / hess_inv as scipy
 * leaves them), SSMQ_BFGS_PRIOR_NOT_PD.  (SSMQ_BFGS_FALLBACK is kept for binary compatibility and no longer returned: where
 * scipy switches to its second line search, scalar_search_wolfe2 / _zoom, the state machine follows it.)  An objective point
 * that the reference's objective would RAISE on (kernel matrix or covariance not positive definite: LinAlgError out of
 * scipy.optimize.minimize, ssinf.py:1153-1241) is a value of +inf here and the search continues; NaN is tested before maxiter
 * (scipy: the other way round).  iters[b] (may be NULL): BFGS iterations; *rounds (may be NULL): device calls made.  Host
 * arrays; synchronous.
 */
enum { SSMQ_BFGS_MAXITER = 1, SSMQ_BFGS_PRECISION_LOSS = 2, SSMQ_BFGS_NAN = 3, SSMQ_BFGS_FALLBACK = 100, SSMQ_BFGS_PRIOR_NOT_PD = 101 };
/*
 * The same optimiser on a host objective (no device): fn(ctx, n, P, traj, rows, vals) fills vals[i] with the objective of
 * trajectory traj[i] at the parameter row rows[i][P] and returns 0 (< 0: abort with that code).  What the tests pin against
 * scipy.optimize.minimize(method='BFGS') on the CPU.
 */
typedef int (*ssmq_objective_fn)(void *ctx, int64_t n, int P, const int64_t *traj, const double *rows, double *vals);
int ssmq_bfgs_lockstep_host(ssmq_objective_fn fn, void *ctx, int64_t B, int P, double fd_step, double *theta, double *hess_inv,
                            int32_t *status, int32_t *iters, int64_t *rounds);
int ssmq_gp_marginal_laplace_batch(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                   const ssmq_integrand *f_obs, int64_t B, double jitter, const double *mean,
                                   const double *cov, const double *y, double time, const double *GQG, const double *R,
                                   const double *prior_mean, const double *prior_cov, double fd_step, double *theta,
                                   double *hess_inv, int32_t *status, int32_t *iters, int64_t *rounds);

/*
 * The whole marginalised filter (ssinf.py:66-118 around :1083-1273) for B trajectories, every trajectory at its own pace: each
 * walks Laplace step (BFGS as above) -> mixture over the NP parameter sigma points -> next time step by itself, and every device
 * round (ONE theta step) serves whatever the unfinished trajectories wait for.  Rounds = the longest trajectory's total, not
 * the sum over the time steps of the slowest one's.  Where the theta step has its two-launch route and P <= 16 the trajectories'
 * state machines live ON THE DEVICE (round 5: pack | theta step | advance are five launches per round, nothing is copied or
 * waited for per round; every eighth round one integer comes back); SSMQ_MARGINAL_HOST_ROUNDS=1 keeps them on the host.
 * y [B][T][Y]; x0_mean [D], x0_cov [D*D]; q_mean [dq] / q_cov [dq*dq] for dynamics that take their noise as an argument (h_dyn is
 * then a (D + dq) -> D transform, GQG = NULL), else NULL; prior_mean [P], prior_cov [P*P] of the log-parameters at step 1 (each
 * step's posterior is the next step's prior); upts [P][NP] unit sigma points and uwts [NP] weights of the parameter mixture
 * (the reference: spherical-radial, NP = 2 P); time index of step k is k (ssinf.py:1088-1122 as called from :101-110).
 * fm [B][T][D], fP [B][T][D*D] filtered moments (NaN from the step at which a trajectory failed); failed [B]: 0, or
 * step + 65536 reason (T < 65536) for the step at which the reference would raise numpy.linalg.LinAlgError - reason 1: the
 * step's parameter prior is not positive definite (_param_log_prior); 2 / 3: the Laplace posterior hess_inv + jitter is not
 * finite / not positive definite (the Cholesky factor of _measurement_update's parameter sigma points, ssinf.py:1103-1106);
 * 4: a parameter sigma point's filter step failed (kernel matrix or covariance not positive definite); 5: the mixture moments
 * are not finite.  Reasons 2-3 depend on the path BFGS took through a noisy objective (forward-difference gradients): a
 * trajectory that fails so in one implementation of the same optimiser may pass in another (tests/test_gpu_parity.py:
 * test_marginal_filter_failures_are_the_reference_s_linalg_errors);
 * theta_last [B][P], pcov_last [B][P*P] (may be NULL): the last parameter posterior; stats [3] (may be NULL): device rounds,
 * BFGS iterations, theta items.  Host arrays; synchronous.
 */
int ssmq_gp_marginal_filter_batch(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                  const ssmq_integrand *f_obs, int64_t B, int T, double jitter, const double *y,
                                  const double *x0_mean, const double *x0_cov, const double *q_mean, const double *q_cov,
                                  const double *GQG, const double *R, const double *prior_mean, const double *prior_cov,
                                  const double *upts, const double *uwts, int NP, double fd_step, double param_jitter,
                                  double *fm, double *fP, int32_t *failed, double *theta_last, double *pcov_last,
                                  int64_t *stats);

/*
 * Unit sigma-point sets and classical quadrature weights, host code (no device needed):
 *   SSMQ_PTS_UT  unscented, 2D+1 points    mtran.py:234-293   par = [kappa, alpha, beta]   (NaN / missing: max(3-D,0), 1, 2)
 *   SSMQ_PTS_SR  spherical-radial, 2D      mtran.py:171-204   par = []
 *   SSMQ_PTS_GH  Gauss-Hermite, degree^D   mtran.py:315-360   par = [degree]               (default 3)
 *   SSMQ_PTS_FS  fully symmetric Student   mtran.py:405-578   par = [degree 3|5, kappa, dof] (defaults 3, max(3-D,0), 4);
 *                degree 7: NOT in the reference (mtran.py:392 stops at 5) - this build's rule for BASELINE configs[4],
 *                generators [0], [v1], [v2], [u,u], [u,u,u], 1 + 4D + 2D(D-1) + 4D(D-1)(D-2)/3 points (1181 at D = 10), exact
 *                for all monomials of total degree <= 7 under St(0, I, max(dof, 7)); parity-unpinned, property-tested
 * BQ models use the points only (bq/bqmod.py:340-382).  ssmq_points_count returns N (or < 0); ssmq_points fills
 * xi [D*N] (row-major (D, N), the reference's column order), wm [N] mean weights, wc [N] covariance weights (any may be
 * NULL) and returns N.
 */
enum ssmq_point_kind { SSMQ_PTS_UT = 0, SSMQ_PTS_SR = 1, SSMQ_PTS_GH = 2, SSMQ_PTS_FS = 3 };
int ssmq_points_count(int kind, int D, const double *par, int n_par);
int ssmq_points(int kind, int D, const double *par, int n_par, double *xi, double *wm, double *wc);

/*
 * Synthetic trajectories and measurements generated on the device, in the filter's plane layout
 * (TransitionModel.simulate_discrete ssmod.py:168-199, MeasurementModel.simulate_measurements ssmod.py:1011-1039):
 *   x[0] = x0_mean + x0_chol z;  x[k] = dyn_fcn(x[k-1], q[k-1], k-1);  y[k] = meas_fcn(x[k], r[k], k+1),
 *   q = q_mean + q_chol z, r = r_mean + r_chol z (lower Cholesky factors, host arrays; means may be NULL = 0).
 * Additive dynamics add G q (G [D*dq] host, NULL = eye(D, dq)); non-additive models get the noise as integrand input.
 * Outputs d_x [T][D][ld], d_y [T][Y][ld].  Random numbers: Philox4x32-10 keyed by `seed`, counter = (traj_offset + b,
 * step, purpose, pair) + Box-Muller, so a trajectory depends on its GLOBAL index only (shard with traj_offset).
 * f_obs = NULL: states only (d_y unused); f_dyn = NULL: measurements of the GIVEN states d_x (simulate_measurements(x)).
 * Parity with np.random is statistical; the generator itself is restated in oracle/ (known-answer vectors).  Synchronous.
 */
int ssmq_simulate_dev(const ssmq_integrand *f_dyn, const ssmq_integrand *f_obs, int D, int Y, int dq, int dr,
                      int dyn_additive, int obs_additive, int64_t B, int64_t ld, int T, const double *x0_mean,
                      const double *x0_chol, const double *q_mean, const double *q_chol, const double *G,
                      const double *r_mean, const double *r_chol, uint64_t seed, uint64_t traj_offset, double *d_x,
                      double *d_y);

/*
 * The same with any of the reference's random variables for the initial state, the process noise and the measurement
 * noise, and optionally the continuous-time dynamics:
 *   SSMQ_RV_GAUSS    mean + chol z                                  GaussRV     utils.py:580-625
 *   SSMQ_RV_STUDENT  mean + chol z / sqrt(u), u ~ Gamma(dof/2, 2/dof) StudentRV   utils.py:349-382, 628-674 (chol = factor
 *                    of the SCALE matrix)
 *   SSMQ_RV_MIXTURE  component k with probability alpha[k], then Gaussian (mean[k], chol[k])   utils.py:254-299,
 *                    research/tpq/tpq_base.py:13-32; n_comp <= 8
 * mean [n_comp][dim] (NULL = 0), chol [n_comp][dim*dim] lower factors, alpha [n_comp]: host arrays.
 * continuous != 0: Euler-Maruyama of TransitionModel.simulate_continuous (ssmod.py:201-244) -
 *   x[k] = x[k-1] + dt dyn_fcn_cont(x[k-1], (sqrt(dt)/dt) q[k-1], k-1), T = floor(duration / dt) columns, the initial
 *   state not among them - for the models that define dyn_fcn_cont: SSMQ_F_REENTRY1D_DYN (ssmod.py:429-432),
 *   SSMQ_F_REENTRY2D_DYN (:569-585), SSMQ_F_CTRS_DYN (:779-780); SSMQ_E_UNSUPPORTED otherwise (the reference's other models
 *   return None there).  Measurements (f_obs) are taken of the returned columns, column k at time k + 1.
 * Gamma variates: Marsaglia-Tsang on the same counter-based stream (attempt t of (trajectory, step, purpose) has its own
 * counter), so every draw is a pure function of (seed, global trajectory index, step).
 */
enum ssmq_rv_kind { SSMQ_RV_GAUSS = 0, SSMQ_RV_STUDENT = 1, SSMQ_RV_MIXTURE = 2 };
typedef struct ssmq_rv {
    int32_t kind, dim, n_comp, reserved;
    double dof;
    const double *mean, *chol, *alpha;
} ssmq_rv;
int ssmq_simulate_rv_dev(const ssmq_integrand *f_dyn, const ssmq_integrand *f_obs, int D, int Y, const ssmq_rv *x0,
                         const ssmq_rv *q, const ssmq_rv *r, const double *G, int dyn_additive, int obs_additive,
                         int64_t B, int64_t ld, int T, int continuous, double dt, uint64_t seed, uint64_t traj_offset,
                         double *d_x, double *d_y);

/*
 * Error statistics of B filtered trajectories against the true states, summed over the Monte-Carlo axis on the device
 * (utils.py:18-38 squared_error, :41-64 mse_matrix, :123-148 neg_log_likelihood; aggregated per time step as
 * research/tpq/tpq_base.py:154-160 does).  d_x, d_fm [T][D][ld], d_fP [T][D*D][ld] (the filter's output buffers),
 * d_status [ld] or NULL (nonzero = trajectory excluded).  sums: host [T][W], W = ssmq_error_sums_width(D) = D*D+D+4:
 *   se[D] sum (x-m)^2 | rmse sum ||x-m|| | nll sum | mse[D*D] sum (x-m)(x-m)' | n_ok | n_pd
 * n_ok = trajectories counted (status 0), n_pd = those that entered the nll sum: every trajectory whose P is nonsingular.
 * A P that is not positive definite does not stop the reference (inv(P), sign * logdet of slogdet, utils.py:143-148):
 * such entries are handled by a second pass with an LU factorisation, so n_pd < n_ok only for a singular P (where
 * numpy.linalg.inv raises).  Sums, not means: ranks add them (one all-reduce) before dividing.  Deterministic summation
 * order.  Synchronous.
 */
int ssmq_error_sums_width(int D);
int ssmq_error_sums_dev(int D, int64_t B, int64_t ld, int T, const double *d_x, const double *d_fm, const double *d_fP,
                        const int32_t *d_status, double *sums);
/*
 * Second phase: sums of the log credibility ratio 10 (log10 dx'P^-1 dx - log10 dx'M^-1 dx) (utils.py:66-120) against
 * the GLOBAL per-step MSE matrices mse [T][D*D] (host; regularisation, e.g. + 1e-6 I of research/tpq/tpq_base.py:161,
 * already added).  sums: host [T][2] = lcr sum | n counted (status 0).  A P that is not positive definite goes through
 * the reference's fallback, mat_sqrt = u sqrt(s) of an SVD (utils.py:426-432), i.e. the quadratic form with |P|: a second
 * pass with a Jacobi eigen-decomposition handles those entries.
 */
int ssmq_lcr_sums_dev(int D, int64_t B, int64_t ld, int T, const double *d_x, const double *d_fm, const double *d_fP,
                      const int32_t *d_status, const double *mse, double *sums);

/*
 * Forward pass of a Studentian filter (ssinf.py:555-736: StudentianInference._time_update / _measurement_update) for B
 * trajectories.  Same loop as ssmq_filter_forward_dev with the reference's scale-matrix bookkeeping:
 *   transforms are fed the SCALE matrix; scale[k] * cov_f (+ G q_smat G') and scale[k] * (cov_f, cov_fx) (+ r_smat)
 *   replace the Gaussian predictive covariances; after the Kalman-form update P = S_pr - K S_y K' the next step's scale
 *   matrix is (dof + delta'delta) / (dof + Y) * P with delta = chol(S_y)^-1 (y - y_mean).
 *   d_S0 [D*D][ld] initial scale matrix ((dof - 2) / dof * x0_cov); GqG [D*D], r_smat [Y*Y], scale [T] host arrays
 *   (scale[k] = (dof_pr - 2) / dof_pr of step k - it depends on the degrees of freedom only, not on the data);
 *   outputs as above: d_fm filtered means, d_fP the reference's `x_cov_fi` (ssinf.py:726).
 */
int ssmq_student_filter_forward_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                    const ssmq_integrand *f_obs, int64_t B, int64_t ld, int T, const double *d_y,
                                    const double *d_m0, const double *d_S0, const double *GqG, const double *r_smat,
                                    const double *scale, double dof, double *d_fm, double *d_fP, int32_t *d_status);

/*
 * A independent filters in ONE launch (round 6; ABI 102).  The reference's studies run several filters over the same
 * measurements one after the other (research/bsq/bsq_ungm.py:132-137, research/tpq/tpq_base.py:175-192); a pass of 1e4
 * trajectories occupies a sixth of the device, so the passes of a study fit side by side.  Each job is one call of
 * ssmq_filter_forward_dev (scale == NULL, dof == 0) or of ssmq_student_filter_forward_dev (scale [T] host, dof > 0) with
 * the same argument meaning; jobs may share d_y / d_m0 / d_P0 (read only) but not their outputs.  The calling thread's
 * stream forks into one branch per job inside a captured launch graph and joins again: the jobs run concurrently on the
 * device, every job is its own fused time-loop kernel (the results are the bits of the single calls), and a repeated call
 * with the same job list is one graph launch.  Jobs whose (models, shapes, form) have no fused kernel run after the graph,
 * one by one, through the ordinary path.  Asynchronous on the calling thread's stream; d_status as for the single calls.
 * At most 64 jobs.  The strip schedule (k_filter_chunked) is not used for the jobs of a multi-launch.
 */
typedef struct ssmq_filter_job {
    ssmq_transform *h_dyn;
    const ssmq_integrand *f_dyn;
    ssmq_transform *h_obs;
    const ssmq_integrand *f_obs;
    int64_t B, ld;
    int32_t T, reserved;
    const double *d_y, *d_m0, *d_P0; /* device: [T][Y][ld], [D][ld], [D*D][ld] */
    const double *GQG, *R;           /* host: [D*D], [Y*Y] (NULL = zeros) */
    double *d_fm, *d_fP;             /* device: [T][D][ld], [T][D*D][ld] */
    int32_t *d_status;               /* device: [ld] */
    const double *scale;             /* host [T]: Studentian recursion (with dof > 0), NULL for the Gaussian filters */
    double dof;
} ssmq_filter_job;
int ssmq_filter_forward_multi_dev(int n_jobs, const ssmq_filter_job *jobs);

/*
 * The drop-in forward pass with HOST arrays in the reference's layout (ssinf.py:66-118: forward_pass takes the measurements
 * as an ndarray and returns ndarrays), pipelined (round 6; ABI 102): the pass runs as K launches over consecutive time blocks
 * (k_filter_range: the whole-pass kernel's bits), the measurements of block k + 1 are uploaded and the filtered moments of
 * block k - 1 downloaded while block k runs.
 *   y (Y, T, B) host; m0 (D) and P0 (D, D) host, or (B, D) and (B, D, D) with SSMQ_PIPED_X0_PER_TRAJECTORY; GQG (D, D), R (Y, Y)
 *   host or NULL; outputs fm (D, T, B), fP (D, D, T, B), status [B] host.  With SSMQ_PIPED_OUT_PINNED fm and fP are page-locked
 *   (ssmq_pinned_alloc): the copy engine writes them in place, no staging copy.  n_blocks = 0: the library chooses K.
 * Synchronous.  Gaussian recursion, additive noise; SSMQ_E_UNSUPPORTED when the (models, shapes, form) combination has no
 * time-block kernel (UNGM with 2 / 3 / 5 points, the reentry and coordinated-turn shapes with unscented points have one) -
 * the caller then uses ssmq_upload_planes / ssmq_filter_forward_dev / ssmq_download_planes.
 */
#define SSMQ_PIPED_OUT_PINNED 1
#define SSMQ_PIPED_X0_PER_TRAJECTORY 2
int ssmq_filter_forward_piped(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                              const ssmq_integrand *f_obs, int64_t B, int T, const double *y, const double *m0,
                              const double *P0, const double *GQG, const double *R, double *fm, double *fP,
                              int32_t *status, int flags, int n_blocks);
/* Page-locked host memory from a process-wide pool (blocks are reused; at most 1 GiB is kept idle).  ssmq_pinned_is_block:
 * 1 if p is the start of a live block. */
int ssmq_pinned_alloc(size_t bytes, void **p);
int ssmq_pinned_free(void *p);
int ssmq_pinned_is_block(const void *p);

/* Name of the kernel(s) ssmq_filter_forward_dev would run for this pair of transforms (for profiles). */
int ssmq_filter_kernel_name(const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, const ssmq_transform *h_obs,
                            const ssmq_integrand *f_obs, char *buf, int len);
/* ... for a batch of B trajectories: batches that leave most of the device idle run the time loop with the sigma points of every
 * transform shared out over the waves of a workgroup (k_filter_wsplit, csrc/ssmq_filter_wsplit.hip); B = 0: the kernel of
 * saturated batches. */
int ssmq_filter_kernel_name_batch(const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, const ssmq_transform *h_obs,
                                  const ssmq_integrand *f_obs, int64_t B, char *buf, int len);

/*
 * The path's only collective: independent Monte-Carlo trajectories shard across ranks, one process per
 * GPU, and only the per-time-step error sums are added at the end (the reference loops `for imc in range(mc)` in one
 * process and averages with numpy: research/tpq/tpq_base.py:154-172, utils.py:113-120).  RCCL over xGMI, opened at run
 * time; nothing else in this library depends on it.
 *   ssmq_comm_unique_id  rank 0: 128-byte id (ncclGetUniqueId) to be handed to every rank out of band
 *   ssmq_comm_init       every rank, after ssmq_set_device: joins the communicator (world = 1 with id = NULL: no RCCL)
 *   ssmq_comm_abandon_init   for a caller that ran ssmq_comm_init on a helper thread and stopped waiting for it (a peer
 *                        never arrived): puts the process's stdout back (ssmq_comm_init parks it on stderr while RCCL
 *                        prints its banner); harmless at any other time
 *   ssmq_allreduce_sum / _max   host buffer of n doubles, reduced in place over all ranks (synchronous)
 *   ssmq_comm_barrier    drains this rank's stream, then a one-element all-reduce
 * One communicator per process; calls are collective and must be issued in the same order on every rank.
 */
int ssmq_comm_unique_id(char *id, int len);
int ssmq_comm_init(int rank, int world, const char *id, int len);
int ssmq_comm_abandon_init(void);
int ssmq_comm_rank(void);
int ssmq_comm_world(void);
int ssmq_allreduce_sum(double *buf, int64_t n);
int ssmq_allreduce_max(double *buf, int64_t n);
int ssmq_comm_barrier(void);
int ssmq_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* SSMQ_H */
