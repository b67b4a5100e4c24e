// Host-side internals of libssmq shared between translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>
#include "ssmq_device.h"
#include "ssmq_wide.h"

struct ssmq_transform {
    int D, E, N, form, emv_mode, device;
    double tp_nu;
    // host copies in the reference's natural layout
    std::vector<double> xi, wm, Wc, Wcc, emv, iK;
    // device constant blocks: `small` = transposed layout of ssmq_apply_small.h, `wide` = natural layout
    double *d_small, *d_wide;
    int opt_mask;   // SSMQ_OPT_* fast paths this handle's constants qualify for (decided in upload_consts)
    // matrix-core route for large point sets (ssmq_gemm_mfma.hip): Wc zero-padded to np_pad x np_pad, or null
    double *d_wc_pad = nullptr;
    int np_pad = 0;
    // ... and [Wc | Wcc'] as np_pad x (np_pad + 16) for the route whose GEMM epilogue forms both covariances, or null
    double *d_wcx_pad = nullptr;
    // ... and [S | Wcc' | wm] with S = tril(sym(Wc)), half the diagonal (Wc = S + S'): the one-launch routes skip the zero blocks
    double *d_sx_pad = nullptr;
    // ... and for point sets beyond that route's instantiations (208 < N): S in fragment order by panels of 256 columns, then the G tile
    // (ssmq_bq_stream.hip: bq_stream_pack), or null
    double *d_sx_pan = nullptr;
    // point sets without an instantiation of that route (N > 64): Wc (and iK for the t-process) as column blocks of
    // kBigCols columns, each [big_kb 16][kBigCols] zero-padded (ssmq_apply_big.hip); null = not built
    double *d_wc_blk = nullptr, *d_ik_blk = nullptr;
    int big_kb = 0, big_ncb = 0;   // k blocks of 16 points; column blocks of [Wc | pad to 16 big_kb | Wcc'] (big_ncb)
    uint32_t generation = 0;   // bumped by every upload of constants (create / update)
    // Threads (include/ssmq.h, conventions): every entry point that takes this handle holds `mu` for its duration; `owner` /
    // `owner_epoch` name the thread context (its stream) that used the handle last - another context waits for that stream
    // before it touches the handle's device blocks (ssmq::HandleGuard).
    mutable std::recursive_mutex mu;
    mutable void *owner = nullptr;
    mutable unsigned owner_epoch = 0;
};

namespace ssmq {

void set_error(const std::string &msg);
// Threads.  Every calling thread has its own CONTEXT: a HIP stream and the caches that belong to a stream (grow-only workspaces,
// pinned staging blocks, captured launch graphs).  Calls of different threads on different handles run concurrently - on the
// host and, stream by stream, on the device.  Contexts are pooled: a thread that ends hands its context (stream and caches
// intact) to the next new thread.  A handle is locked for the duration of every entry point that takes it (HandleGuard; two
// handles in address order), and a context that picks up a handle last used by another one waits for that context's stream
// first, so that constants uploaded or buffers built there are complete.  Device buffers the CALLER passes between threads are
// the caller's to order (ssmq_sync() in the thread that queued the work), as with any per-thread stream.  The communicator entry
// points (ssmq_comm_*) belong to one thread.
struct Ctx {
    hipStream_t stream = nullptr;
    int dev = -1;
    unsigned epoch = 0;                      // unique per (context, device binding): per-device function attributes, cached graphs
    void *gemm_ws = nullptr;                 // scratch of the matrix-core routes (ssmq_api.hip)
    size_t gemm_ws_bytes = 0;
    void *stage = nullptr, *fc = nullptr, *theta_graphs = nullptr;      // ssmq_api.hip: StagingArena, FilterCache, theta graphs
    void *pinned_flags = nullptr;            // 64 bytes of pinned host memory the device rounds report through (ssmq_marginal.hip)
    void *strip_buf = nullptr;               // flags + hand-over buffer of the strip schedule (ssmq_filter_chunked.hip), grow-only
    size_t strip_bytes = 0;
    void *multi = nullptr;                   // ssmq_api_study.hip: MultiCache (side streams, events, constants and captured graph of
                                             // ssmq_filter_forward_multi_dev)
    void *pipe = nullptr;                    // ssmq_api_study.hip: PipeCache (copy streams and events of ssmq_filter_forward_piped)
    bool no_strips = false;                  // set while a multi-filter launch is being built: the strip schedule owns ONE buffer
                                             // per context and its jobs run concurrently
};
Ctx &ctx();
struct HandleGuard {
    const ssmq_transform *a, *b;
    explicit HandleGuard(const ssmq_transform *h0, const ssmq_transform *h1 = nullptr);
    ~HandleGuard();
    HandleGuard(const HandleGuard &) = delete;
    HandleGuard &operator=(const HandleGuard &) = delete;
};
#define SSMQ_HANDLE_LOCK(...) ssmq::HandleGuard ssmq_handle_guard_(__VA_ARGS__)
// ... for any number of handles (ssmq_filter_forward_multi_dev): unique handles, locked in address order
struct MultiHandleGuard {
    std::vector<const ssmq_transform *> hs;
    explicit MultiHandleGuard(std::vector<const ssmq_transform *> handles);
    ~MultiHandleGuard();
    MultiHandleGuard(const MultiHandleGuard &) = delete;
    MultiHandleGuard &operator=(const MultiHandleGuard &) = delete;
};
int hip_fail(hipError_t e, const char *what);
hipStream_t stream();
int ensure_device();
// The calling thread's context epoch: a new value whenever the context binds to a device (reset_device_caches); per-function
// attributes (hipFuncAttributeMaxDynamicSharedMemorySize) are per device and are set again when a thread sees a new value
// (the `static thread_local unsigned attr_epoch` of the launchers).
unsigned device_epoch();

// Device arena + pinned staging blocks of the host-buffer entry points (ssmq_api.hip, ssmq_api_study.hip): one per context.
struct StagingArena;
StagingArena &stage_of_ctx();

#define SSMQ_HIP(call)                                        \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return ssmq::hip_fail(e__, #call); \
    } while (0)

// Device arena + pinned staging blocks of the host-buffer entry points that are called in tight loops with small batches
// (ssmq_apply_batch: the drop-in apply(); ssmq_gp_theta_step), grow-only, dropped when the device changes.  The calls are
// synchronous on the library's one stream, so one arena serves them all.
struct StagingArena {
    void *dev = nullptr, *hin = nullptr, *hout = nullptr;
    size_t dev_bytes = 0, hin_bytes = 0, hout_bytes = 0;
    static int grow(void **p, size_t *have, size_t need, bool host) {
        if (*have >= need) return SSMQ_OK;
        if (*p) {
            SSMQ_HIP(hipStreamSynchronize(stream()));
            if (host) hipHostFree(*p); else hipFree(*p);
        }
        *p = nullptr;
        *have = 0;
        const size_t want = need + need / 4;       // a little head room: consecutive calls differ by a few items
        if (host) SSMQ_HIP(hipHostMalloc(p, want, hipHostMallocDefault)); else SSMQ_HIP(hipMalloc(p, want));
        *have = want;
        return SSMQ_OK;
    }
    int reserve(size_t d, size_t hi, size_t ho) {
        int rc;
        if ((rc = grow(&dev, &dev_bytes, d, false)) || (rc = grow(&hin, &hin_bytes, hi, true)) ||
            (rc = grow(&hout, &hout_bytes, ho, true)))
            return rc;
        return SSMQ_OK;
    }
    void drop() {
        if (dev) hipFree(dev);
        if (hin) hipHostFree(hin);
        if (hout) hipHostFree(hout);
        dev = hin = hout = nullptr;
        dev_bytes = hin_bytes = hout_bytes = 0;
    }
};

void fill_fpar(const ssmq_integrand *f, FPar *fp);

// dispatch table of the register-resident kernels (ssmq_apply_small_*.hip)
typedef hipError_t (*small_launch_fn)(const ApplyArgs &, hipStream_t);
struct SmallEntry {
    int fid, D, E, N, form, tp, sel, opt;
    small_launch_fn fn;
    const char *name;
};
const SmallEntry *find_small(int fid, int D, int E, int N, int form, int tp, int sel, int opt);

// batch GEMM fx Wc on the matrix cores (ssmq_gemm_mfma.hip)
int gemm_mfma_padded(int N);
int launch_fxwc_mfma(int NP, const double *A, const double *Bm, double *T, int64_t M, int lda, int ldt, hipStream_t s);
// ... with the covariance of every trajectory formed in the epilogue (no T in memory)
bool fxwc_cov_supported(int E);
// whole BQ transform in one launch, integrand values LDS-resident (ssmq_bq_fused.hip); WideArgs: ssmq_wide.h
struct WideArgs;
bool bq_fused_supported(int D, int E, int N);
int launch_bq_fused(const WideArgs &a, const double *X, const double *emv, int emv_broadcast, int64_t B, hipStream_t s);
// ... the same for any larger point set, the point axis tiled (ssmq_bq_stream.hip)
bool bq_stream_supported(int D, int E, int N);
int bq_stream_tpw(int E);          // trajectories per 64-row block of FX (fragment order, WideArgs::fx_frag)
size_t bq_stream_x_doubles(int N);
void bq_stream_pack(int D, int N, const double *Wc, const double *Wcc, const double *wm, double *X);
// linearisation transform (ssmq_linear.hip): mean_f = f(mean), cov_fx = J cov, cov_f = cov_fx J' with the model's own Jacobian
struct FPar;
int launch_linearize(int D, int E, int din, const ssmq_integrand *f, const FPar &fp, int64_t B, int64_t ld, const double *d_mean,
                     const double *d_cov, const double *d_time, int time_stride, double *d_mean_f, double *d_cov_f, double *d_cov_fx,
                     int32_t *d_status, const double *d_cov_add, double cov_scale, double ccov_scale, hipStream_t s);
size_t bq_stream_parts_doubles(int E, int N, int64_t B, int cus);   // scratch for the panel-wise tail of a batch (0: no tail is cut)
int launch_bq_stream(const WideArgs &a, const double *X, const double *emv, int emv_broadcast, int64_t B, const double *fx,
                     const double *chol, int64_t lda, int cus, double *parts, hipStream_t s);
int launch_fxwc_cov_mfma(int NP, const double *A, const double *X, int64_t M, int lda, const double *mean_rows,
                         const double *chol, const double *emv, int emv_broadcast, const double *cov_add,
                         double cov_scale, double ccov_scale, int E, int D, double *cov_f, double *cov_fx, int64_t es,
                         int64_t bs, int64_t bs_fx, hipStream_t s);
int launch_row_means(const double *A, const double *wm, int64_t M, int lda, int N, double *mean_rows, hipStream_t s);
// ... and for any point count, by column blocks (ssmq_gemm_mfma.hip); the per-trajectory rest (ssmq_apply_big.hip)
constexpr int kBigCols = 256;
int launch_fxwc_blocks(const double *A, const double *Wblk, double *T, int64_t M, int lda, int ldt, int KB, int ncb,
                       hipStream_t s);
struct BigRest {
    int D, E, N, form, emv_mode;
    double tp_nu, cov_scale, ccov_scale;
    const double *consts;          // WideLayout block (wm, Wc diagonal for the centred form, Wcc, xiT, emv)
    const double *fx, *t, *t2;     // rows b E + e: integrand values [lda], fx Wc [ldt] (null: centred form), fx iK [ldt] or null
    int64_t lda, ldt;
    int p_col;                     // BQ: column of T where the D columns fx Wcc' start (the GEMM's extra columns)
    const double *mean_rows;       // [B E]
    const double *chol;            // [B][D][D]
    const double *cov_add;         // [E*E] or null
    double *cov_f, *cov_fx;        // element e of trajectory b at ptr[e * es + b * bs_*]
    int64_t es, bs_cf, bs_cfx;
    const int32_t *status;         // [B] or null: nonzero -> NaN outputs
};
int launch_big_rest(const BigRest &r, int64_t B, hipStream_t s);

// theta-batched step on items that are already on the device, their number read from device memory (ssmq_api.hip; used by the
// device-resident rounds of ssmq_gp_marginal_filter_batch, ssmq_marginal.hip)
struct ThetaDev {
    int Din, D, Y, Nd, No;
    int64_t cap, ld;                                     // items the arena holds; plane pitch
    double *pard, *paro, *mean, *cov, *ysoa, *tt;        // per item, filled by the caller: kernel parameters [cap][1 + Din] / [cap][1 + D],
                                                         // state moments [cap][Din] / [cap][Din Din], measurement planes [Y][ld], time [ld]
    double *m_fi, *P_fi, *ll;                            // results: planes [D][ld], [D D][ld], log-likelihood [ld]
    int32_t *st_all;                                     // merged status flags [ld]
    double *xid, *xio, *gq, *rr, *cd, *co, *mid;         // internal: unit points, noise terms, constant blocks, work planes
    int32_t *st5;
};
bool theta_dev_supported(const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, const ssmq_transform *h_obs, const ssmq_integrand *f_obs);
size_t theta_dev_bytes(const ssmq_transform *h_dyn, const ssmq_transform *h_obs, int64_t cap);
size_t theta_dev_carve(ThetaDev &t, const ssmq_transform *h_dyn, const ssmq_transform *h_obs, int64_t cap, void *base);
int theta_dev_upload_static(const ThetaDev &t, const ssmq_transform *h_dyn, const ssmq_transform *h_obs, const double *GQG, const double *R,
                            hipStream_t s);
int theta_dev_enqueue(const ThetaDev &t, const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, const ssmq_transform *h_obs,
                      const ssmq_integrand *f_obs, double jitter, int64_t bound, const int32_t *d_count, hipStream_t s);

// trajectory / measurement simulator (ssmq_simulate.hip)
struct SimRv {
    int kind, dim, ncomp, off;      // off: doubles into the constants block (alpha | mean | chol)
    double dof;
};
struct SimLaunch {
    int mode, D, Y, dq, dr, dyn_additive, obs_additive, T, continuous, g_off;
    double dt;
    int64_t B, ld;
    uint64_t seed, traj_offset;
    SimRv rv[3];                    // initial state, process noise, measurement noise
    const ssmq_integrand *f_dyn, *f_obs;
    const double *d_consts;
    double *d_x, *d_y;
};
int launch_simulate(const SimLaunch &h, hipStream_t s);
bool has_continuous_dynamics(int fid);

// measurement update (ssmq_filter.hip)
int launch_kalman_update(int D, int Y, int64_t B, int64_t ld, const double *m_pr, const double *P_pr,
                         const double *y_mean, const double *P_y, const double *P_yx, const double *y, double *m_fi,
                         double *P_fi, int32_t *status, hipStream_t s);

}  // namespace ssmq
