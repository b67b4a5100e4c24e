// C ABI of libssmq (include/ssmq.h): device plumbing, transform handles, kernel dispatch.  No CPU fallback exists
// behind these entry points: every compute call ends in a HIP kernel launch or returns an error.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <string>
#include <thread>
#include <unordered_map>
#include <unistd.h>
#include <vector>
#include "ssmq_host.h"
#include "ssmq_update.h"
#include "ssmq_apply_small.h"
#include "ssmq_fused.h"

namespace ssmq {

static thread_local std::string g_err;

void set_error(const std::string &msg) { g_err = msg; }

int hip_fail(hipError_t e, const char *what) {
    if (e == hipSuccess) return SSMQ_OK;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return SSMQ_E_HIP;
}

// ---- per-thread contexts (ssmq_host.h) ------------------------------------------------------------------------------------
namespace {
struct CtxRegistry {
    std::mutex mu;                 // guards the free list, and every creation / destruction of a context's stream
    std::vector<Ctx *> free_list;
};
CtxRegistry &registry() {          // leaked on purpose: threads may end after the static destructors have run
    static CtxRegistry *r = new CtxRegistry;
    return *r;
}
std::atomic<unsigned> g_epoch_counter{0};
std::atomic<int> g_preferred_device{-1};     // the device of the last ssmq_set_device(): where a new thread starts
struct CtxHolder {
    Ctx *c = nullptr;
    ~CtxHolder() {                 // the thread ends: its context (stream, caches) goes back to the pool as it is
        if (!c) return;
        CtxRegistry &r = registry();
        std::lock_guard<std::mutex> l(r.mu);
        r.free_list.push_back(c);
        c = nullptr;
    }
};
thread_local CtxHolder t_holder;
thread_local bool t_first_call = true;
}  // namespace

// ---- switches: one snapshot of the environment, re-read only when the environment changed (ssmq_host.h) --------------------
namespace {
struct SwitchSnapshot {
    std::mutex mu;
    uint64_t sig = 0;
    std::unordered_map<std::string, const char *> vals;      // nullptr = unset
    std::deque<std::string> pool;                            // values are never freed: callers may hold the pointer
};
uint64_t environ_signature() {
    uint64_t h = 1469598103934665603ull;
    for (char **e = ::environ; e && *e; ++e) h = (h ^ (uint64_t)(uintptr_t)*e) * 1099511628211ull;
    return h ^ (uint64_t)(uintptr_t)::environ;
}
}  // namespace
const char *sw(const char *name) {
    static SwitchSnapshot c;
    std::lock_guard<std::mutex> g(c.mu);
    const uint64_t s = environ_signature();
    if (s != c.sig) {
        c.vals.clear();
        c.sig = s;
    }
    auto it = c.vals.find(name);
    if (it == c.vals.end()) {
        const char *v = getenv(name);
        if (v) {
            c.pool.emplace_back(v);
            v = c.pool.back().c_str();
        }
        it = c.vals.emplace(name, v).first;
    }
    return it->second;
}

Ctx &ctx() {
    if (!t_holder.c) {
        CtxRegistry &r = registry();
        std::lock_guard<std::mutex> l(r.mu);
        if (!r.free_list.empty()) {
            t_holder.c = r.free_list.back();
            r.free_list.pop_back();
        } else {
            t_holder.c = new Ctx;
        }
    }
    return *t_holder.c;
}

// Caches that hold memory / graphs of ONE device (the matrix-core scratch, the filter workspace with its captured launch
// loop, staging blocks): the calling thread's context drops them when it binds to another device, so that a workspace or graph
// of the previous device is never used from the new device's stream.
void reset_device_caches();

int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (libssmq has no CPU fallback)");
        return SSMQ_E_HIP;
    }
    int dev = 0;
    SSMQ_HIP(hipGetDevice(&dev));
    if (t_first_call) {            // HIP's current device is per thread and starts at 0: a new thread starts where the library is
        t_first_call = false;
        const int pref = g_preferred_device.load();
        if (pref >= 0 && pref < n && pref != dev) {
            SSMQ_HIP(hipSetDevice(pref));
            dev = pref;
        }
    }
    Ctx &c = ctx();
    if (c.stream == nullptr || c.dev != dev) {
        std::lock_guard<std::mutex> l(registry().mu);
        if (c.stream != nullptr) {          // leaving a device: finish the context's work, release what it cached there
            hipSetDevice(c.dev);
            hipStreamSynchronize(c.stream);
            reset_device_caches();
            hipStreamDestroy(c.stream);
            c.stream = nullptr;
            SSMQ_HIP(hipSetDevice(dev));
        }
        SSMQ_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        c.dev = dev;
        c.epoch = ++g_epoch_counter;
    }
    return SSMQ_OK;
}

hipStream_t stream() { return ctx().stream; }
unsigned device_epoch() { return ctx().epoch; }

// A context that picks up handle `h` (locked by the caller) after another context used it: see HandleGuard.
static void adopt_handle(const ssmq_transform *h, Ctx &me) {
    if (!h || (h->owner == &me && h->owner_epoch == me.epoch)) return;
    if (h->owner) {
        // last used from another context: what that context queued on its stream - uploads of the handle's constants, buffers
        // built on first use, kernels still reading them - precedes whatever this context queues next.  Under the registry lock
        // only an EVENT is recorded on the owner's stream (the lock keeps the owner's stream alive); this context's stream
        // waits for it after the lock is released, so one thread's pending GPU work no longer stalls every other thread's
        // first call / hand-over behind the global mutex.  Everything the library does to a handle's device
        // blocks is stream-ordered (hipMemcpyAsync / kernels on the context's stream; hipFree synchronises the device).
        hipEvent_t ev = nullptr;
        {
            std::lock_guard<std::mutex> l(registry().mu);
            const Ctx *o = (const Ctx *)h->owner;
            if (o->stream && o->epoch == h->owner_epoch && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
                if (o->dev != me.dev) hipSetDevice(o->dev);
                if (hipEventRecord(ev, o->stream) != hipSuccess) {
                    hipStreamSynchronize(o->stream);        // (cannot happen; the old behaviour as the fallback)
                    hipEventDestroy(ev);
                    ev = nullptr;
                }
                if (o->dev != me.dev) hipSetDevice(me.dev);
            }
        }
        if (ev) {
            if (me.stream == nullptr || hipStreamWaitEvent(me.stream, ev, 0) != hipSuccess) hipEventSynchronize(ev);
            hipEventDestroy(ev);         // (released once the event has completed)
        }
    }
    h->owner = &me;
    h->owner_epoch = me.epoch;
}

HandleGuard::HandleGuard(const ssmq_transform *h0, const ssmq_transform *h1) : a(h0), b(h1) {
    if (b == a) b = nullptr;
    if (!a) { a = b; b = nullptr; }
    if (a && b && b < a) std::swap(a, b);          // two handles: always in address order
    if (a) a->mu.lock();
    if (b) b->mu.lock();
    if (!a) return;
    (void)ensure_device();                         // (a failure is reported by the entry point's own call)
    Ctx &me = ctx();
    for (const ssmq_transform *h : {a, b}) adopt_handle(h, me);
}
MultiHandleGuard::MultiHandleGuard(std::vector<const ssmq_transform *> handles) : hs(std::move(handles)) {
    hs.erase(std::remove(hs.begin(), hs.end(), nullptr), hs.end());
    std::sort(hs.begin(), hs.end());
    hs.erase(std::unique(hs.begin(), hs.end()), hs.end());
    for (const ssmq_transform *h : hs) h->mu.lock();
    if (hs.empty()) return;
    (void)ensure_device();
    Ctx &me = ctx();
    for (const ssmq_transform *h : hs) adopt_handle(h, me);
}
MultiHandleGuard::~MultiHandleGuard() {
    for (auto it = hs.rbegin(); it != hs.rend(); ++it) (*it)->mu.unlock();
}
HandleGuard::~HandleGuard() {
    if (b) b->mu.unlock();
    if (a) a->mu.unlock();
}

void fill_fpar(const ssmq_integrand *f, FPar *fp) {
    memset(fp, 0, sizeof(*fp));
    fp->n_par = std::max(0, std::min<int>(f->n_par, SSMQ_MAX_FPAR));
    fp->n_idx = std::max(0, std::min<int>(f->n_idx, SSMQ_MAX_FIDX));
    for (int i = 0; i < fp->n_par; ++i) fp->p[i] = f->par[i];
    for (int i = 0; i < fp->n_idx; ++i) fp->idx[i] = f->idx[i];
    fp->ttab = nullptr;
}

const SmallEntry *small_table_a(int *n);
const SmallEntry *small_table_b(int *n);
const SmallEntry *small_table_c(int *n);
const SmallEntry *small_table_d(int *n);

const SmallEntry *find_small(int fid, int D, int E, int N, int form, int tp, int sel, int opt) {
    typedef const SmallEntry *(*tab_fn)(int *);
    static const tab_fn tabs[] = {small_table_a, small_table_b, small_table_c, small_table_d};
    for (tab_fn t : tabs) {
        int n = 0;
        const SmallEntry *e = t(&n);
        for (int i = 0; i < n; ++i)
            if (e[i].fid == fid && e[i].D == D && e[i].E == E && e[i].N == N && e[i].form == form && e[i].tp == tp &&
                e[i].sel == sel && e[i].opt == opt)
                return &e[i];
    }
    return nullptr;
}

// ---- layout conversion ---------------------------------------------------------------------------------------------
// AoS [B][n] <-> SoA [n][ld] through a padded LDS tile: both the global read and the global write are coalesced.
constexpr int kTile = 64;
__global__ __launch_bounds__(256) void k_aos_to_soa(const double *__restrict__ aos, double *__restrict__ soa, int n,
                                                     int64_t B, int64_t ld) {
    __shared__ double tile[kTile][kTile + 1];
    const int64_t b0 = (int64_t)blockIdx.x * kTile;
    const int e0 = blockIdx.y * kTile;
    const int tx = threadIdx.x % kTile, ty = threadIdx.x / kTile;  // 64 x 4
    for (int r = ty; r < kTile; r += 4) {  // r: trajectory within tile, tx: element within tile
        const int64_t b = b0 + r;
        const int e = e0 + tx;
        if (b < B && e < n) tile[r][tx] = aos[b * n + e];
    }
    __syncthreads();
    for (int r = ty; r < kTile; r += 4) {  // r: element within tile, tx: trajectory
        const int64_t b = b0 + tx;
        const int e = e0 + r;
        if (b < B && e < n) soa[(int64_t)e * ld + b] = tile[tx][r];
    }
}
__global__ __launch_bounds__(256) void k_soa_to_aos(const double *__restrict__ soa, double *__restrict__ aos, int n,
                                                     int64_t B, int64_t ld) {
    __shared__ double tile[kTile][kTile + 1];
    const int64_t b0 = (int64_t)blockIdx.x * kTile;
    const int e0 = blockIdx.y * kTile;
    const int tx = threadIdx.x % kTile, ty = threadIdx.x / kTile;
    for (int r = ty; r < kTile; r += 4) {  // r: element, tx: trajectory
        const int64_t b = b0 + tx;
        const int e = e0 + r;
        if (b < B && e < n) tile[r][tx] = soa[(int64_t)e * ld + b];
    }
    __syncthreads();
    for (int r = ty; r < kTile; r += 4) {  // r: trajectory, tx: element
        const int64_t b = b0 + r;
        const int e = e0 + tx;
        if (b < B && e < n) aos[b * n + e] = tile[tx][r];
    }
}

__global__ void k_status_first(const int32_t *st, int64_t B, unsigned long long *first) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B && st[i] != 0) atomicMin(first, (unsigned long long)i);
}

static int upload_consts(ssmq_transform *h) {
    ++h->generation;   // part of the filter loop's graph key: new constants never replay a graph captured for old ones
    const int D = h->D, E = h->E, N = h->N;
    const bool sigma = h->form == SSMQ_FORM_SIGMA;
    const ConstLayout cs = const_layout(D, E, N, h->form);
    const WideLayout cw = wide_layout(D, E, N, h->form);
    std::vector<double> s(cs.total, 0.0), w(cw.total, 0.0);
    for (int d = 0; d < D; ++d)
        for (int n = 0; n < N; ++n) {
            s[cs.xi + n * D + d] = h->xi[d * N + n];
            w[cw.xiT + n * D + d] = h->xi[d * N + n];
        }
    for (int n = 0; n < N; ++n) s[cs.wm + n] = w[cw.wm + n] = h->wm[n];
    if (sigma) {
        for (int n = 0; n < N; ++n) s[cs.Wc + n] = w[cw.Wc + n] = h->Wc[n];
        // the centred form's cross-covariance sum_n wc_n (fx_n - m)(x_n - m_x)' with x_n - m_x = L xi_n is (fx_c W') L' for
        // W[d][n] = xi[d][n] wc_n: kept in the natural-layout block's Wcc slot for the kernels that form it that way
        for (int d = 0; d < D; ++d)
            for (int n = 0; n < N; ++n) w[cw.Wcc + d * N + n] = h->xi[d * N + n] * h->Wc[n];
    } else {
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                s[cs.Wc + j * N + i] = h->Wc[i * N + j];  // transposed: column j contiguous
                w[cw.Wc + i * N + j] = h->Wc[i * N + j];
            }
        for (int d = 0; d < D; ++d)
            for (int n = 0; n < N; ++n) {
                s[cs.Wcc + d * N + n] = h->Wcc[d * N + n];   // row d contiguous (one body of the ccov stage)
                w[cw.Wcc + d * N + n] = h->Wcc[d * N + n];
            }
    }
    for (int i = 0; i < E * E; ++i) s[cs.emv + i] = w[cw.emv + i] = h->emv[i];
    if (h->tp_nu > 0.0) {
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                s[cs.iK + j * N + i] = h->iK[i * N + j];
                w[cw.iK + i * N + j] = h->iK[i * N + j];
            }
    }
    // per-point records (const_layout: rec / rs); record N stays zero
    for (int n = 0; n < N; ++n) {
        double *r = s.data() + cs.rec + (size_t)n * cs.rs;
        for (int d = 0; d < D; ++d) r[d] = h->xi[d * N + n];
        r[D] = h->wm[n];
        if (sigma) {
            r[D + 1] = h->Wc[n];
        } else {
            for (int d = 0; d < D; ++d) r[D + 1 + d] = h->Wcc[d * N + n];
            for (int i = 0; i < N; ++i) r[2 * D + 1 + i] = h->Wc[i * N + n];
            if (h->tp_nu > 0.0)
                for (int i = 0; i < N; ++i) r[2 * D + 1 + N + i] = h->iK[i * N + n];
        }
    }
    // ---- optional fast paths (ssmq_apply_small.h: SSMQ_OPT_LDL / SSMQ_OPT_UT), each verified before it is offered ----
    h->opt_mask = 0;
    if (!sigma) {
        // Wc = U diag(d) U', unit lower U, no pivoting; accepted only if the factorisation reproduces Wc to 1e-14
        std::vector<double> U((size_t)N * N, 0.0), dd(N, 0.0), A(h->Wc);
        bool ok = true;
        double wmax = 0.0;
        for (double v : A) wmax = std::max(wmax, std::fabs(v));
        for (int j = 0; j < N && ok; ++j) {
            double dj = A[j * N + j];
            for (int k = 0; k < j; ++k) dj -= U[j * N + k] * U[j * N + k] * dd[k];
            if (!(std::fabs(dj) > 1e-13 * wmax)) ok = false;
            dd[j] = dj;
            U[j * N + j] = 1.0;
            for (int i = j + 1; i < N && ok; ++i) {
                double v = 0.5 * (A[i * N + j] + A[j * N + i]);
                for (int k = 0; k < j; ++k) v -= U[i * N + k] * U[j * N + k] * dd[k];
                U[i * N + j] = v / dj;
            }
        }
        double err = 0.0;
        for (int i = 0; i < N && ok; ++i)
            for (int j = 0; j < N; ++j) {
                double v = 0.0;
                for (int k = 0; k <= std::min(i, j); ++k) v += U[i * N + k] * dd[k] * U[j * N + k];
                err = std::max(err, std::fabs(v - A[i * N + j]));
            }
        if (ok && err <= 1e-14 * wmax && !ssmq::sw("SSMQ_NO_FASTPATH")) {
            h->opt_mask |= SSMQ_OPT_LDL;
            for (int j = 0; j < N; ++j) {
                s[cs.ldlD + j] = dd[j];
                for (int i = 0; i < N; ++i) s[cs.ldlU + j * N + i] = U[i * N + j];   // column j contiguous
            }
        }
    }
    if (N == 2 * D + 1 && !ssmq::sw("SSMQ_NO_FASTPATH")) {
        const double cc = h->xi[0 * N + 1];
        bool ut = cc > 0.0;
        for (int d = 0; d < D && ut; ++d)
            for (int n = 0; n < N; ++n) {
                double want = 0.0;
                if (n == 1 + d) want = cc;
                if (n == 1 + D + d) want = -cc;
                if (h->xi[d * N + n] != want) ut = false;
            }
        if (ut) {
            h->opt_mask |= SSMQ_OPT_UT;
            s[cs.utc] = cc;
        }
        // SSMQ_OPT_SYM: weights invariant under every reflection of the point set (swap of points 1 + k and 1 + D + k) up to the
        // round-off of the weight computation: symmetrise, rebuild (wm, Wc, Wcc) from the symmetric parameters, accept if no
        // weight moved by more than 2e-13 of the largest of its array; then the LDL' of the (D + 1) x (D + 1) symmetric block.
        if (ut && !sigma && h->tp_nu <= 0.0) {
            const int M = D + 1;
            const double tol = 2e-13;
            auto W = [&](int i, int j) { return h->Wc[(size_t)i * N + j]; };
            std::vector<double> wms(M), gam(D), beta(D), Mt((size_t)M * M);
            double dev_wm = 0.0, dev_wc = 0.0, dev_cc = 0.0, mx_wm = 0.0, mx_wc = 0.0, mx_cc = 0.0;
            wms[0] = h->wm[0];
            for (int k = 0; k < D; ++k) wms[1 + k] = 0.5 * (h->wm[1 + k] + h->wm[1 + D + k]);
            for (int n = 0; n < N; ++n) {
                mx_wm = std::max(mx_wm, std::fabs(h->wm[n]));
                dev_wm = std::max(dev_wm, std::fabs(h->wm[n] - wms[n == 0 ? 0 : 1 + (n - 1) % D]));
            }
            for (int d = 0; d < D; ++d) {
                gam[d] = 0.5 * (h->Wcc[(size_t)d * N + 1 + d] - h->Wcc[(size_t)d * N + 1 + D + d]);
                for (int n = 0; n < N; ++n) {
                    const double want = n == 1 + d ? gam[d] : (n == 1 + D + d ? -gam[d] : 0.0);
                    mx_cc = std::max(mx_cc, std::fabs(h->Wcc[(size_t)d * N + n]));
                    dev_cc = std::max(dev_cc, std::fabs(h->Wcc[(size_t)d * N + n] - want));
                }
            }
            Mt[0] = W(0, 0);
            for (int k = 0; k < D; ++k) {
                const int p = 1 + k, q = 1 + D + k;
                Mt[1 + k] = Mt[(size_t)(1 + k) * M] = 0.25 * (W(0, p) + W(0, q) + W(p, 0) + W(q, 0));
                beta[k] = 0.25 * (W(p, p) + W(q, q) - W(p, q) - W(q, p));
                for (int j = 0; j < D; ++j) {
                    const int r = 1 + j, t = 1 + D + j;
                    Mt[(size_t)(1 + k) * M + 1 + j] = 0.25 * (W(p, r) + W(p, t) + W(q, r) + W(q, t));
                }
            }
            for (int i = 0; i < M; ++i)          // (symmetric by construction up to the asymmetry of Wc itself)
                for (int j = 0; j < i; ++j) Mt[(size_t)i * M + j] = Mt[(size_t)j * M + i] = 0.5 * (Mt[(size_t)i * M + j] + Mt[(size_t)j * M + i]);
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    const int ci = i == 0 ? 0 : 1 + (i - 1) % D, cj = j == 0 ? 0 : 1 + (j - 1) % D;
                    double want = Mt[(size_t)ci * M + cj];
                    if (i != 0 && ci == cj) want += (i == j) ? beta[ci - 1] : -beta[ci - 1];
                    mx_wc = std::max(mx_wc, std::fabs(W(i, j)));
                    dev_wc = std::max(dev_wc, std::fabs(W(i, j) - want));
                }
            bool ok = dev_wm <= tol * mx_wm && dev_wc <= tol * mx_wc && dev_cc <= tol * mx_cc;
            // Mt = Ut diag(d) Ut' with Ut unit UPPER triangular (the kernel meets the columns of G in increasing order): the LDL' of
            // the index-reversed matrix, reversed back.  No pivoting; accepted only if it reproduces Mt to 1e-14.
            std::vector<double> Lr((size_t)M * M, 0.0), dr(M, 0.0), Ut((size_t)M * M, 0.0), dd(M, 0.0);
            auto Mr = [&](int i, int j) { return Mt[(size_t)(M - 1 - i) * M + (M - 1 - j)]; };
            for (int j = 0; j < M && ok; ++j) {
                double dj = Mr(j, j);
                for (int k = 0; k < j; ++k) dj -= Lr[(size_t)j * M + k] * Lr[(size_t)j * M + k] * dr[k];
                if (!(std::fabs(dj) > 1e-13 * mx_wc)) ok = false;
                dr[j] = dj;
                Lr[(size_t)j * M + j] = 1.0;
                for (int i = j + 1; i < M && ok; ++i) {
                    double v = Mr(i, j);
                    for (int k = 0; k < j; ++k) v -= Lr[(size_t)i * M + k] * Lr[(size_t)j * M + k] * dr[k];
                    Lr[(size_t)i * M + j] = v / dj;
                }
            }
            for (int i = 0; i < M; ++i) {
                dd[i] = dr[M - 1 - i];
                for (int j = 0; j < M; ++j) Ut[(size_t)i * M + j] = Lr[(size_t)(M - 1 - i) * M + (M - 1 - j)];
            }
            double err = 0.0;
            for (int i = 0; i < M && ok; ++i)
                for (int j = 0; j < M; ++j) {
                    double v = 0.0;
                    for (int k = std::max(i, j); k < M; ++k) v += Ut[(size_t)i * M + k] * dd[k] * Ut[(size_t)j * M + k];
                    err = std::max(err, std::fabs(v - Mt[(size_t)i * M + j]));
                }
            if (ok && err <= 1e-14 * mx_wc && (h->opt_mask & SSMQ_OPT_LDL) && !ssmq::sw("SSMQ_NO_SYM")) {
                h->opt_mask |= SSMQ_OPT_SYM;
                for (int j = 0; j < M; ++j) {
                    double *r = s.data() + cs.sym + (size_t)j * cs.sym_rs;
                    r[0] = wms[j];
                    r[1] = dd[j];
                    r[2] = j ? gam[j - 1] : 0.0;
                    r[3] = j ? beta[j - 1] : 0.0;
                    for (int i = 0; i < j; ++i) r[4 + i] = Ut[(size_t)i * M + j];
                }
            }
        }
    }
    std::vector<double> wpad;
    const int np = sigma ? 0 : gemm_mfma_padded(N);
    if (np && !ssmq::sw("SSMQ_NO_MFMA")) {
        wpad.assign((size_t)np * np, 0.0);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) wpad[(size_t)i * np + j] = h->Wc[i * N + j];
        if (!h->d_wc_pad) SSMQ_HIP(hipMalloc(&h->d_wc_pad, sizeof(double) * np * np));
        h->np_pad = np;
        SSMQ_HIP(hipMemcpyAsync(h->d_wc_pad, wpad.data(), sizeof(double) * np * np, hipMemcpyHostToDevice, stream()));
        // X = [Wc | Wcc'] for the route whose GEMM epilogue forms both covariances (row k: row k of Wc, then
        // Wcc[0..D)[k]): T = FX Wc exactly as the other kernels form it, also for a Wc that is not symmetric to the last bit
        const int nx = np + 16;
        std::vector<double> xpad((size_t)np * nx, 0.0);
        for (int k = 0; k < N; ++k) {
            for (int j = 0; j < N; ++j) xpad[(size_t)k * nx + j] = h->Wc[k * N + j];
            for (int d = 0; d < D && d < 16; ++d) xpad[(size_t)k * nx + np + d] = h->Wcc[d * N + k];
            // the tile's last column is free up to D = 15: wm there makes FX wm a by-product of the same GEMM
            // (k_bq_fused reads it; k_fxwc_cov_mfma only looks at columns < D)
            if (D <= 15) xpad[(size_t)k * nx + np + 15] = h->wm[k];
        }
        if (!h->d_wcx_pad) SSMQ_HIP(hipMalloc(&h->d_wcx_pad, sizeof(double) * np * nx));
        SSMQ_HIP(hipMemcpyAsync(h->d_wcx_pad, xpad.data(), sizeof(double) * np * nx, hipMemcpyHostToDevice, stream()));
        // the same with S in place of Wc: S = lower triangle of Wc with half its diagonal, so that Wc = S + S' and
        // fx Wc fx' = C + C', C = (fx S) fx' (k_bq_fused / k_bq_stream: half the matrix instructions of the main product).
        // Only for a Wc that is symmetric to the last bit - every Wc the weight kernels (and the reference, bq/bqmod.py:520-521)
        // produce; an injected non-symmetric one keeps the routes that form (fx Wc) fx' as written.
        bool symmetric = true;
        for (int k = 0; k < N && symmetric; ++k)
            for (int j = 0; j < k; ++j)
                if (h->Wc[k * N + j] != h->Wc[j * N + k]) {
                    symmetric = false;
                    break;
                }
        if (symmetric) {
            std::vector<double> spad(xpad);
            for (int k = 0; k < N; ++k)
                for (int j = 0; j < N; ++j)
                    spad[(size_t)k * nx + j] = j < k ? h->Wc[k * N + j] : (j == k ? 0.5 * h->Wc[k * N + k] : 0.0);
            if (!h->d_sx_pad) SSMQ_HIP(hipMalloc(&h->d_sx_pad, sizeof(double) * np * nx));
            SSMQ_HIP(hipMemcpyAsync(h->d_sx_pad, spad.data(), sizeof(double) * np * nx, hipMemcpyHostToDevice, stream()));
            SSMQ_HIP(hipStreamSynchronize(stream()));   // spad goes out of scope
        } else if (h->d_sx_pad) {
            SSMQ_HIP(hipStreamSynchronize(stream()));
            hipFree(h->d_sx_pad);
            h->d_sx_pad = nullptr;
        }
        SSMQ_HIP(hipStreamSynchronize(stream()));   // xpad goes out of scope
    }
    if (!sigma && bq_stream_supported(D, E, N)) {
        // one-launch route for these sizes (k_bq_stream): S = tril(Wc), half the diagonal, by panels - a Wc symmetric to the last
        // bit only (see d_sx_pad above)
        bool symmetric = bq_stream_supported(D, E, N) && h->tp_nu <= 0.0;
        for (int k = 0; k < N && symmetric; ++k)
            for (int j = 0; j < k; ++j)
                if (h->Wc[(size_t)k * N + j] != h->Wc[(size_t)j * N + k]) {
                    symmetric = false;
                    break;
                }
        if (h->d_sx_pan) {
            SSMQ_HIP(hipStreamSynchronize(stream()));
            hipFree(h->d_sx_pan);
            h->d_sx_pan = nullptr;
        }
        if (symmetric) {
            std::vector<double> xs(bq_stream_x_doubles(N));
            bq_stream_pack(D, N, h->Wc.data(), h->Wcc.data(), h->wm.data(), xs.data());
            SSMQ_HIP(hipMalloc(&h->d_sx_pan, sizeof(double) * xs.size()));
            SSMQ_HIP(hipMemcpyAsync(h->d_sx_pan, xs.data(), sizeof(double) * xs.size(), hipMemcpyHostToDevice, stream()));
            SSMQ_HIP(hipStreamSynchronize(stream()));
        }
    }
    if (!sigma && N > 64 && !np && !ssmq::sw("SSMQ_NO_MFMA")) {
        // any other point count beyond the wave kernels: Wc (and iK for the t-process) as column blocks of kBigCols
        // columns, block c = [kb 16][kBigCols] zero-padded, for the blocked GEMM (launch_fxwc_blocks)
        // the Wc blocks carry D extra columns from column 16 kb on: Wcc', so that fx Wcc' comes out of the same GEMM
        const int kb = (N + 15) / 16, ncols = 16 * kb + D, ncb = (ncols + kBigCols - 1) / kBigCols;
        const size_t per = (size_t)kb * 16 * kBigCols, total = per * ncb;
        auto pack = [&](const std::vector<double> &src, double **dst, bool with_wcc) -> int {
            std::vector<double> blk(total, 0.0);
            auto at = [&](int i, int j) -> double & { return blk[(size_t)(j / kBigCols) * per + (size_t)i * kBigCols + j % kBigCols]; };
            for (int i = 0; i < N; ++i) {
                for (int j = 0; j < N; ++j) at(i, j) = src[(size_t)i * N + j];
                if (with_wcc)
                    for (int d = 0; d < D; ++d) at(i, 16 * kb + d) = h->Wcc[(size_t)d * N + i];
            }
            // on the library's stream, as every other upload of this function: a transform queued there (the entry points ending
            // in _dev are asynchronous) may still be reading the old blocks
            if (*dst && (h->big_kb != kb || h->big_ncb != ncb)) {
                SSMQ_HIP(hipStreamSynchronize(stream()));
                hipFree(*dst);
                *dst = nullptr;
            }
            if (!*dst) SSMQ_HIP(hipMalloc(dst, sizeof(double) * total));
            SSMQ_HIP(hipMemcpyAsync(*dst, blk.data(), sizeof(double) * total, hipMemcpyHostToDevice, stream()));
            SSMQ_HIP(hipStreamSynchronize(stream()));   // blk goes out of scope
            return SSMQ_OK;
        };
        int rcp = pack(h->Wc, &h->d_wc_blk, true);
        if (rcp) return rcp;
        if (h->tp_nu > 0.0 && (int)h->iK.size() == N * N && (rcp = pack(h->iK, &h->d_ik_blk, false))) return rcp;
        h->big_kb = kb;
        h->big_ncb = ncb;
    }
    SSMQ_HIP(hipMemcpyAsync(h->d_small, s.data(), sizeof(double) * cs.total, hipMemcpyHostToDevice, stream()));
    SSMQ_HIP(hipMemcpyAsync(h->d_wide, w.data(), sizeof(double) * cw.total, hipMemcpyHostToDevice, stream()));
    SSMQ_HIP(hipStreamSynchronize(stream()));
    return SSMQ_OK;
}

int sel_pattern(const ssmq_integrand *f, int din) {
    // 0: leading entries, 1: (0, 2, 4, ...), -1: anything else
    if (f->n_idx <= 0) return 0;
    bool lead = true, even = true;
    for (int k = 0; k < din && k < f->n_idx; ++k) {
        lead = lead && f->idx[k] == k;
        even = even && f->idx[k] == 2 * k;
    }
    if (f->n_idx < din) return -1;
    return lead ? 0 : (even ? 1 : -1);
}

static int check_integrand(const ssmq_transform *h, const ssmq_integrand *f, FInfo *fi) {
    if (!f || !integrand_info(f->id, fi)) {
        set_error("unknown integrand id");
        return SSMQ_E_ARG;
    }
    if (f->id == SSMQ_F_BEARING_MEAS) {
        fi->dout = f->n_par / 2;
        if (fi->dout < 1 || fi->dout > SSMQ_MAX_FPAR / 2) {
            set_error("bearing measurement: n_par must be 2 * sensors, 1..8 sensors");
            return SSMQ_E_ARG;
        }
    }
    if (fi->dout != h->E) {
        set_error("integrand output dimension does not match the transform's E");
        return SSMQ_E_ARG;
    }
    if (f->n_idx > SSMQ_MAX_FIDX || f->n_par > SSMQ_MAX_FPAR || f->n_idx < 0 || f->n_par < 0) {
        set_error("integrand: n_idx / n_par out of range");
        return SSMQ_E_ARG;
    }
    if (f->n_idx > 0) {
        if (f->n_idx < fi->din) {
            set_error("integrand: state index shorter than the integrand's input");
            return SSMQ_E_ARG;
        }
        for (int k = 0; k < f->n_idx; ++k)
            if (f->idx[k] < 0 || f->idx[k] >= h->D) {
                set_error("integrand: state index out of range");
                return SSMQ_E_ARG;
            }
    } else if (fi->din > h->D) {
        set_error("integrand reads more inputs than the transform's D");
        return SSMQ_E_ARG;
    }
    return SSMQ_OK;
}

// Grow-only scratch of the matrix-core route (asynchronous callers cannot own temporaries): FX and T = FX Wc as
// (B E) x NP row-major, the Cholesky factors [B][D][D].
constexpr int64_t kGemmMinRows = 256;
#define g_gemm_ws (ssmq::ctx().gemm_ws)                  // (the calling thread's context: ssmq_host.h)
#define g_gemm_ws_bytes (ssmq::ctx().gemm_ws_bytes)
static int gemm_scratch(int64_t M, int NP, int64_t B, int D, double **fx, double **tt, double **chol, bool fused = false) {
    // three-pass route: FX | T | factors; two-pass route: FX | transformed means as rows | factors
    const size_t n_fx = (size_t)M * NP, n_t = fused ? (size_t)M : n_fx,
                 need = sizeof(double) * (n_fx + n_t + (size_t)B * D * D);
    if (g_gemm_ws_bytes < need) {
        if (g_gemm_ws) {
            SSMQ_HIP(hipStreamSynchronize(stream()));
            hipFree(g_gemm_ws);
        }
        g_gemm_ws = nullptr;
        g_gemm_ws_bytes = 0;
        SSMQ_HIP(hipMalloc(&g_gemm_ws, need));
        g_gemm_ws_bytes = need;
    }
    *fx = (double *)g_gemm_ws;
    *tt = *fx + n_fx;
    *chol = *tt + n_t;
    return SSMQ_OK;
}
// scratch of the blocked route: FX [M][lda] | T [M][ldt] x n_t | means [M] | factors [B][D][D]
static int big_scratch(int64_t M, int lda, int ldt, int n_t, int64_t B, int D, double **fx, double **tt, double **mrow,
                       double **chol) {
    const size_t n_fx = (size_t)M * lda, n_tt = (size_t)M * ldt * n_t;
    const size_t need = sizeof(double) * (n_fx + n_tt + (size_t)M + (size_t)B * D * D);
    if (g_gemm_ws_bytes < need) {
        if (g_gemm_ws) {
            SSMQ_HIP(hipStreamSynchronize(stream()));
            hipFree(g_gemm_ws);
        }
        g_gemm_ws = nullptr;
        g_gemm_ws_bytes = 0;
        SSMQ_HIP(hipMalloc(&g_gemm_ws, need));
        g_gemm_ws_bytes = need;
    }
    *fx = (double *)g_gemm_ws;
    *tt = *fx + n_fx;
    *mrow = *tt + n_tt;
    *chol = *mrow + M;
    return SSMQ_OK;
}
static void drop_gemm_scratch() {
    if (g_gemm_ws) hipFree(g_gemm_ws);
    g_gemm_ws = nullptr;
    g_gemm_ws_bytes = 0;
}

int apply_dev_impl(ssmq_transform *h, const ssmq_integrand *f, int64_t B, int64_t ld, const double *d_mean,
                   const double *d_cov, const double *d_time, int time_stride, double *d_mean_f, double *d_cov_f,
                   double *d_cov_fx, int32_t *d_status, const double *d_cov_add, const char **kernel_name,
                   bool dry_run, double cov_scale = 1.0, double ccov_scale = 1.0, const double *ttab = nullptr,
                   bool stream_out = true) {
    FInfo fi;
    int rc = check_integrand(h, f, &fi);
    if (rc) return rc;
    if (h->form == SSMQ_FORM_TAYLOR1) {
        // the linearisation transform (mtran.py:49-59): no points, no weights, one launch (ssmq_linear.hip)
        if (kernel_name) *kernel_name = "k_linearize";
        if (dry_run || B <= 0) return SSMQ_OK;
        if (!d_mean || !d_cov || !d_mean_f || !d_cov_f || !d_cov_fx || !d_status || (fi.uses_time && !d_time) || ld < B) {
            set_error("apply: null pointer or ld < B");
            return SSMQ_E_ARG;
        }
        FPar fp;
        fill_fpar(f, &fp);
        fp.ttab = ttab;
        return launch_linearize(h->D, h->E, fi.din, f, fp, B, ld, d_mean, d_cov, d_time, d_time ? time_stride : 0, d_mean_f, d_cov_f,
                                d_cov_fx, d_status, d_cov_add, cov_scale, ccov_scale, stream());
    }
    const int tp = h->tp_nu > 0.0 ? 1 : 0;
    const int sel = sel_pattern(f, fi.din);
    const SmallEntry *se = nullptr;
    if (sel >= 0) {
        // best available fast path first (TP keeps the dense covariance form; see SSMQ_OPT_* in ssmq_apply_small.h)
        const int want[5] = {(!tp && (h->opt_mask & 7) == 7) ? 7 : -1, h->opt_mask & (tp ? SSMQ_OPT_UT : 3), h->opt_mask & SSMQ_OPT_UT,
                             h->opt_mask & SSMQ_OPT_LDL & (tp ? 0 : 1), 0};
        for (int k = 0; k < 5 && !se; ++k)
            if (want[k] >= 0) se = find_small(f->id, h->D, h->E, h->N, h->form, tp, sel, want[k]);
    }
    const bool wide_fits = wide_lds_bytes(h->D, h->E, h->N) <= 160 * 1024 - 64;
    // point sets beyond the wave kernels without a fused matrix-core instantiation: evaluation pass, blocked GEMM, rest
    // (the kernel-name query runs with B = 0: it reports the route of a large batch)
    const int64_t b_route = dry_run ? ((int64_t)1 << 20) : B;
    const bool big = !se && h->N > 64 && ((h->form == SSMQ_FORM_BQ && h->d_wc_blk && (b_route * h->E >= kGemmMinRows || !wide_fits) &&
                                           (h->tp_nu <= 0.0 || h->d_ik_blk)) ||
                                          (h->form == SSMQ_FORM_SIGMA && !wide_fits));
    const bool streamed = !se && h->form == SSMQ_FORM_BQ && h->tp_nu <= 0.0 && h->d_sx_pan && b_route * h->E >= kGemmMinRows &&
                          bq_stream_supported(h->D, h->E, h->N);
    const bool one_launch = !se && !big && h->form == SSMQ_FORM_BQ && h->d_wc_pad && h->d_sx_pad && h->tp_nu <= 0.0 &&
                            b_route * h->E >= kGemmMinRows && bq_fused_supported(h->D, h->E, h->N);
    if (kernel_name) *kernel_name = se ? se->name : streamed ? "k_bq_stream" : big ? "k_apply_big" : one_launch ? "k_bq_fused" : ((wide_full_uses_tile(h->D, h->E, h->N) && tile_ld_ok(dry_run ? 0 : ld)) ? "k_apply_tile" : wide_full_uses_wave(h->D, h->E, h->N) ? "k_apply_wave" : "k_apply_wide");
    // (the same two conditions launch_apply_wide tests - tile_pitch_ok there, with unit batch strides as set below; the name
    // query has no batch and reports the route of planes shorter than 2^29 doubles)
    if (dry_run) return SSMQ_OK;
    if (B <= 0) return SSMQ_OK;
    if (!d_mean || !d_cov || !d_mean_f || !d_cov_f || !d_cov_fx || !d_status || (fi.uses_time && !d_time) ||
        ld < B) {
        set_error("apply: null pointer or ld < B");
        return SSMQ_E_ARG;
    }
    if (se) {
        ApplyArgs a;
        a.mean = d_mean; a.cov = d_cov; a.time = d_time ? d_time : d_mean; a.mean_f = d_mean_f; a.cov_f = d_cov_f;
        a.cov_fx = d_cov_fx; a.status = d_status; a.consts = h->d_small;
        a.cov_add = d_cov_add ? d_cov_add : h->d_small + const_layout(h->D, h->E, h->N, h->form).zero; a.B = B; a.ld = ld;
        a.time_stride = d_time ? time_stride : 0; a.emv_mode = h->emv_mode; a.tp_nu = h->tp_nu;
        a.cov_scale = cov_scale; a.ccov_scale = ccov_scale;
        a.stream_out = stream_out ? 1 : 0;
        fill_fpar(f, &a.fp);
        a.fp.ttab = ttab;
        return hip_fail(se->fn(a, stream()), se->name);
    }
    if (!big && !wide_fits) {
        set_error("apply: shape too large for the LDS-resident generic kernel");
        return SSMQ_E_UNSUPPORTED;
    }
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.D = h->D; a.E = h->E; a.N = h->N; a.form = h->form; a.mode = SSMQ_WIDE_FULL; a.fid = f->id;
    a.time_stride = d_time ? time_stride : 0; a.emv_mode = h->emv_mode; a.tp_nu = h->tp_nu; a.consts = h->d_wide;
    a.cov_scale = cov_scale; a.ccov_scale = ccov_scale;
    a.cov_add = d_cov_add; a.mean = d_mean; a.cov = d_cov; a.time = d_time; a.es_in = ld; a.bs_mean = 1; a.bs_cov = 1;
    a.mean_f = d_mean_f; a.cov_f = d_cov_f; a.cov_fx = d_cov_fx; a.es_out = ld; a.bs_mf = a.bs_cf = a.bs_cfx = 1;
    a.status = d_status;
    fill_fpar(f, &a.fp);
    a.fp.ttab = ttab;
    if (streamed) {
        // two launches: (1) one wave per trajectory: factor, points, integrand values FX, factors; (2) the streamed product whose
        // epilogues form mean, covariance and cross-covariance (ssmq_bq_stream.hip)
        const int kb = (h->N + 15) / 16, lda = kb * 16;
        // FX in fragment order: blocks of 64 / E trajectories, 64 rows each (ssmq_wide.h: fx_frag)
        const int tpw = bq_stream_tpw(h->E);
        const int64_t M = (B + tpw - 1) / tpw * 64;
        // the last, partly empty round of workgroups is cut by panel (ssmq_bq_stream.hip: bq_stream_split): room for the parts
        static thread_local int cus = 0;
        static thread_local unsigned cus_epoch = ~0u;
        if (cus_epoch != device_epoch()) {
            int dev = 0;
            hipDeviceProp_t prop;
            SSMQ_HIP(hipGetDevice(&dev));
            SSMQ_HIP(hipGetDeviceProperties(&prop, dev));
            cus = prop.multiProcessorCount;
            cus_epoch = device_epoch();
        }
        const size_t parts_n = bq_stream_parts_doubles(h->E, h->N, B, cus);
        const int ldt = (int)((parts_n + (size_t)M - 1) / (size_t)M);
        double *fx, *tt, *mrow, *chol;
        if ((rc = big_scratch(M, lda, ldt, ldt ? 1 : 0, B, h->D, &fx, &tt, &mrow, &chol))) return rc;
        WideArgs e = a;
        e.fx_ld = lda; e.fx_out = fx; e.mrow_out = mrow; e.chol_out = chol; e.fx_frag = tpw;
        if ((rc = hip_fail(launch_eval_wave(e, B, stream()), "k_eval_wave"))) return rc;
        const WideLayout wl = wide_layout(h->D, h->E, h->N, h->form);
        return launch_bq_stream(a, h->d_sx_pan, h->d_wide + wl.emv, h->emv_mode == SSMQ_EMV_BROADCAST ? 1 : 0, B, fx, chol, lda, cus,
                                parts_n ? tt : nullptr, stream());
    }
    if (big) {
        const bool bq = h->form == SSMQ_FORM_BQ, tpb = bq && h->tp_nu > 0.0;
        const int kb = (h->N + 15) / 16, lda = kb * 16, ldt = bq ? h->big_ncb * kBigCols : 0;
        const int64_t M = B * h->E;
        double *fx, *tt, *mrow, *chol;
        if ((rc = big_scratch(M, lda, ldt, bq ? (tpb ? 2 : 1) : 0, B, h->D, &fx, &tt, &mrow, &chol))) return rc;
        WideArgs e = a;
        e.fx_ld = lda; e.fx_out = fx; e.mrow_out = mrow; e.chol_out = chol;
        if ((rc = hip_fail(launch_eval_wave(e, B, stream()), "k_eval_wave"))) return rc;
        if (bq && (rc = launch_fxwc_blocks(fx, h->d_wc_blk, tt, M, lda, ldt, kb, h->big_ncb, stream()))) return rc;
        if (tpb && (rc = launch_fxwc_blocks(fx, h->d_ik_blk, tt + (size_t)M * ldt, M, lda, ldt, kb, h->big_ncb, stream()))) return rc;
        BigRest r;
        memset(&r, 0, sizeof(r));
        r.D = h->D; r.E = h->E; r.N = h->N; r.form = h->form; r.emv_mode = h->emv_mode; r.tp_nu = h->tp_nu;
        r.cov_scale = cov_scale; r.ccov_scale = ccov_scale; r.consts = h->d_wide; r.fx = fx; r.t = bq ? tt : nullptr;
        r.t2 = tpb ? tt + (size_t)M * ldt : nullptr; r.lda = lda; r.ldt = ldt; r.p_col = 16 * kb; r.mean_rows = mrow; r.chol = chol;
        r.cov_add = d_cov_add; r.cov_f = d_cov_f; r.cov_fx = d_cov_fx; r.es = ld; r.bs_cf = 1; r.bs_cfx = 1; r.status = d_status;
        return launch_big_rest(r, B, stream());
    }
    if (h->d_wc_pad && h->form == SSMQ_FORM_BQ && B * h->E >= kGemmMinRows) {
        // large point set: integrand values of the whole batch -> one GEMM on the matrix cores -> per-trajectory rest
        const int NP = h->np_pad;
        const int64_t M = B * h->E;
        double *fx, *tt, *chol;
        if (h->tp_nu <= 0.0 && h->d_sx_pad && bq_fused_supported(h->D, h->E, h->N)) {
            // one launch: the workgroup that owns a block of the GEMM's rows evaluates the integrand into LDS itself
            const WideLayout wl = wide_layout(h->D, h->E, h->N, h->form);
            return launch_bq_fused(a, h->d_sx_pad, h->d_wide + wl.emv, h->emv_mode == SSMQ_EMV_BROADCAST ? 1 : 0, B, stream());
        }
        if (h->tp_nu <= 0.0 && h->d_wcx_pad && fxwc_cov_supported(h->E) && h->D <= 16 && !ssmq::sw("SSMQ_NO_FUSED_COV")) {
            // two passes: (1) one wave per trajectory: factor, points, integrand values, mean; (2) the GEMM whose
            // epilogue forms the covariance and the cross-covariance from its accumulators
            if ((rc = gemm_scratch(M, NP, B, h->D, &fx, &tt, &chol, true))) return rc;
            double *mrow = tt;
            WideArgs e = a;
            e.fx_ld = NP; e.fx_out = fx; e.mrow_out = mrow; e.chol_out = chol;
            if ((rc = hip_fail(launch_eval_wave(e, B, stream()), "k_eval_wave"))) return rc;
            const WideLayout wl = wide_layout(h->D, h->E, h->N, h->form);
            return launch_fxwc_cov_mfma(NP, fx, h->d_wcx_pad, M, NP, mrow, chol, h->d_wide + wl.emv,
                                        h->emv_mode == SSMQ_EMV_BROADCAST ? 1 : 0, d_cov_add, cov_scale, ccov_scale, h->E,
                                        h->D, d_cov_f, d_cov_fx, ld, 1, 1, stream());
        }
        if ((rc = gemm_scratch(M, NP, B, h->D, &fx, &tt, &chol))) return rc;
        WideArgs e = a;
        e.mode = SSMQ_WIDE_EVAL; e.fx_ld = NP; e.fx_out = fx; e.chol_out = chol;
        if ((rc = hip_fail(launch_apply_wide(e, B, stream()), "k_apply_wide(eval)"))) return rc;
        if ((rc = launch_fxwc_mfma(NP, fx, h->d_wc_pad, tt, M, NP, NP, stream()))) return rc;
        a.mode = SSMQ_WIDE_FX; a.fx_ld = NP; a.fx_in = fx; a.t_in = tt; a.chol_in = chol; a.status = nullptr;
        return hip_fail(launch_apply_wide(a, B, stream()), "k_apply_wide(fx + T)");
    }
    return hip_fail(launch_apply_wide(a, B, stream()), "k_apply_wide");
}

}  // namespace ssmq

using namespace ssmq;

namespace {
}  // namespace
namespace ssmq {
StagingArena &stage_of_ctx() {
    Ctx &c = ssmq::ctx();
    if (!c.stage) c.stage = new StagingArena;
    return *(StagingArena *)c.stage;
}
}  // namespace ssmq
namespace {
#define g_stage (stage_of_ctx())

// memcpy between caller memory and the pinned blocks; large blocks on several threads (one core moves ~8 GB/s, which
// would cost more than the PCIe transfer it feeds)
void fast_copy(void *dst, const void *src, size_t bytes) {
    constexpr size_t kChunk = size_t(2) << 20;
    if (bytes < 2 * kChunk) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t nt = std::min<size_t>(8, bytes / kChunk);
    const size_t per = (bytes / nt + 63) / 64 * 64;
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; ++t) {
        const size_t lo = t * per, n = lo < bytes ? std::min(per, bytes - lo) : 0;
        if (n) th.emplace_back([=] { memcpy((char *)dst + lo, (const char *)src + lo, n); });
    }
    memcpy(dst, src, std::min(per, bytes));
    for (auto &t : th) t.join();
}
}  // namespace
namespace ssmq {
void drop_staging_arena() { g_stage.drop(); }
}

extern "C" {

int ssmq_version(void) { return SSMQ_VERSION; }
const char *ssmq_last_error(void) { return g_err.c_str(); }

int ssmq_device_count(int *n) {
    if (!n) return SSMQ_E_ARG;
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) {
        *n = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    return SSMQ_OK;
}
int ssmq_set_device(int device) {
    SSMQ_HIP(hipSetDevice(device));
    t_first_call = false;
    g_preferred_device.store(device);            // threads that make their first call from now on start here
    return ensure_device();
}
int ssmq_current_device(void) {
    if (ensure_device()) return -1;
    return ctx().dev;
}
int ssmq_device_name(char *buf, int len) {
    if (!buf || len <= 0) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0;
    SSMQ_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    SSMQ_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return SSMQ_OK;
}
int ssmq_device_pci_bus_id(char *buf, int len) {
    if (!buf || len < 13) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0;
    SSMQ_HIP(hipGetDevice(&dev));
    SSMQ_HIP(hipDeviceGetPCIBusId(buf, len, dev));
    return SSMQ_OK;
}

int ssmq_malloc(void **dptr, size_t bytes) {
    if (!dptr) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    SSMQ_HIP(hipMalloc(dptr, bytes ? bytes : 8));
    return SSMQ_OK;
}
int ssmq_free(void *dptr) {
    if (!dptr) return SSMQ_OK;
    SSMQ_HIP(hipFree(dptr));
    return SSMQ_OK;
}
int ssmq_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    int rc = ensure_device();
    if (rc) return rc;
    SSMQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream()));
    SSMQ_HIP(hipStreamSynchronize(stream()));
    return SSMQ_OK;
}
int ssmq_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    int rc = ensure_device();
    if (rc) return rc;
    SSMQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream()));
    SSMQ_HIP(hipStreamSynchronize(stream()));
    return SSMQ_OK;
}
int ssmq_memcpy_d2d(void *dst, const void *src, size_t bytes) {
    int rc = ensure_device();
    if (rc) return rc;
    SSMQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream()));
    return SSMQ_OK;
}
int ssmq_memset(void *dptr, int value, size_t bytes) {
    int rc = ensure_device();
    if (rc) return rc;
    SSMQ_HIP(hipMemsetAsync(dptr, value, bytes, stream()));
    return SSMQ_OK;
}
int ssmq_sync(void) {
    int rc = ensure_device();
    if (rc) return rc;
    SSMQ_HIP(hipStreamSynchronize(stream()));
    return SSMQ_OK;
}

int ssmq_aos_to_soa(const double *d_aos, double *d_soa, int n, int64_t B, int64_t ld) {
    if (!d_aos || !d_soa || n <= 0 || B < 0 || ld < B) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    dim3 grid((unsigned)((B + kTile - 1) / kTile), (unsigned)((n + kTile - 1) / kTile));
    hipLaunchKernelGGL(k_aos_to_soa, grid, dim3(256), 0, stream(), d_aos, d_soa, n, B, ld);
    return hip_fail(hipGetLastError(), "k_aos_to_soa");
}
int ssmq_soa_to_aos(const double *d_soa, double *d_aos, int n, int64_t B, int64_t ld) {
    if (!d_aos || !d_soa || n <= 0 || B < 0 || ld < B) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    dim3 grid((unsigned)((B + kTile - 1) / kTile), (unsigned)((n + kTile - 1) / kTile));
    hipLaunchKernelGGL(k_soa_to_aos, grid, dim3(256), 0, stream(), d_soa, d_aos, n, B, ld);
    return hip_fail(hipGetLastError(), "k_soa_to_aos");
}

int ssmq_event_create(void **ev) {
    if (!ev) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    hipEvent_t e;
    SSMQ_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return SSMQ_OK;
}
int ssmq_event_destroy(void *ev) {
    if (!ev) return SSMQ_OK;
    SSMQ_HIP(hipEventDestroy((hipEvent_t)ev));
    return SSMQ_OK;
}
int ssmq_event_record(void *ev) {
    if (!ev) return SSMQ_E_ARG;
    if (int rc = ensure_device()) return rc;        // a thread's FIRST call: its context has no stream yet (the NULL stream would not
                                                    // bracket the work the thread queues afterwards)
    SSMQ_HIP(hipEventRecord((hipEvent_t)ev, stream()));
    return SSMQ_OK;
}
int ssmq_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!start || !stop || !ms) return SSMQ_E_ARG;
    SSMQ_HIP(hipEventSynchronize((hipEvent_t)stop));
    SSMQ_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SSMQ_OK;
}

int ssmq_status_first(const int32_t *d_status, int64_t B, int64_t *first) {
    if (!d_status || !first || B < 0) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    *first = -1;
    if (B == 0) return SSMQ_OK;
    unsigned long long *d_first = nullptr;
    SSMQ_HIP(hipMalloc((void **)&d_first, sizeof(unsigned long long)));
    unsigned long long init = ~0ull;
    SSMQ_HIP(hipMemcpyAsync(d_first, &init, sizeof(init), hipMemcpyHostToDevice, stream()));
    hipLaunchKernelGGL(k_status_first, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream(), d_status, B, d_first);
    unsigned long long res = ~0ull;
    hipError_t e = hipMemcpyAsync(&res, d_first, sizeof(res), hipMemcpyDeviceToHost, stream());
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    hipFree(d_first);
    if (e != hipSuccess) return hip_fail(e, "ssmq_status_first");
    *first = (res == ~0ull) ? -1 : (int64_t)res;
    return SSMQ_OK;
}

// ---- transform handle ----------------------------------------------------------------------------------------------
ssmq_transform *ssmq_transform_create(int D, int E, int N, int form, const double *xi, const double *wm,
                                      const double *Wc, const double *Wcc, const double *emv, int emv_mode,
                                      double tp_nu, const double *tp_iK) {
    if (D < 1 || D > SSMQ_MAX_DIM || E < 1 || E > SSMQ_MAX_DIM || N < 1 || N > SSMQ_MAX_PTS ||
        (form != SSMQ_FORM_BQ && form != SSMQ_FORM_SIGMA) || !xi || !wm || !Wc || (form == SSMQ_FORM_BQ && !Wcc) ||
        (tp_nu > 0.0 && !tp_iK) || (emv_mode != SSMQ_EMV_DIAG && emv_mode != SSMQ_EMV_BROADCAST)) {
        set_error("transform_create: bad argument");
        return nullptr;
    }
    if (ensure_device()) return nullptr;
    ssmq_transform *h = new ssmq_transform();
    h->D = D; h->E = E; h->N = N; h->form = form; h->emv_mode = emv_mode; h->tp_nu = tp_nu;
    hipGetDevice(&h->device);
    h->xi.assign(xi, xi + D * N);
    h->wm.assign(wm, wm + N);
    h->Wc.assign(Wc, Wc + (form == SSMQ_FORM_SIGMA ? N : N * N));
    if (form == SSMQ_FORM_BQ) h->Wcc.assign(Wcc, Wcc + D * N);
    h->emv.assign(E * E, 0.0);
    if (emv) h->emv.assign(emv, emv + E * E);
    if (tp_nu > 0.0) h->iK.assign(tp_iK, tp_iK + N * N);
    h->d_small = h->d_wide = nullptr;
    const ConstLayout cs = const_layout(D, E, N, form);
    const WideLayout cw = wide_layout(D, E, N, form);
    if (hipMalloc((void **)&h->d_small, sizeof(double) * cs.total) != hipSuccess ||
        hipMalloc((void **)&h->d_wide, sizeof(double) * cw.total) != hipSuccess || upload_consts(h) != SSMQ_OK) {
        if (g_err.empty()) set_error("transform_create: device allocation failed");
        ssmq_transform_destroy(h);
        return nullptr;
    }
    return h;
}

// The linearisation transform has neither points nor weights; the handle keeps a one-point placeholder block so that every
// code path that sizes or frees constants finds what it expects.
ssmq_transform *ssmq_transform_create_linear(int D, int E) {
    if (D < 1 || D > SSMQ_MAX_DIM || E < 1 || E > SSMQ_MAX_DIM) {
        set_error("transform_create_linear: bad argument");
        return nullptr;
    }
    std::vector<double> xi((size_t)D, 0.0);
    const double one = 1.0;
    ssmq_transform *h = ssmq_transform_create(D, E, 1, SSMQ_FORM_SIGMA, xi.data(), &one, &one, nullptr, nullptr, SSMQ_EMV_DIAG, 0.0,
                                              nullptr);
    if (h) h->form = SSMQ_FORM_TAYLOR1;
    return h;
}

int ssmq_transform_update(ssmq_transform *h, const double *xi, const double *wm, const double *Wc, const double *Wcc,
                          const double *emv, int emv_mode, double tp_nu, const double *tp_iK) {
    SSMQ_HANDLE_LOCK(h);
    if (h && h->form == SSMQ_FORM_TAYLOR1) {
        set_error("transform_update: the linearisation transform has no constants");
        return SSMQ_E_ARG;
    }
    if (!h) return SSMQ_E_ARG;
    const int D = h->D, E = h->E, N = h->N;
    if (xi) h->xi.assign(xi, xi + D * N);
    if (wm) h->wm.assign(wm, wm + N);
    if (Wc) h->Wc.assign(Wc, Wc + (h->form == SSMQ_FORM_SIGMA ? N : N * N));
    if (Wcc && h->form == SSMQ_FORM_BQ) h->Wcc.assign(Wcc, Wcc + D * N);
    if (emv) h->emv.assign(emv, emv + E * E);
    if (emv_mode == SSMQ_EMV_DIAG || emv_mode == SSMQ_EMV_BROADCAST) h->emv_mode = emv_mode;
    if (tp_iK) h->iK.assign(tp_iK, tp_iK + N * N);
    if (tp_nu > 0.0) {
        if (h->iK.empty()) {
            set_error("transform_update: tp_nu > 0 needs tp_iK");
            return SSMQ_E_ARG;
        }
        h->tp_nu = tp_nu;
    }
    return upload_consts(h);
}

void ssmq_transform_destroy(ssmq_transform *h) {
    if (!h) return;
    {   // whatever context used the handle last has finished with its device blocks (the guard waits for that stream) ...
        SSMQ_HANDLE_LOCK(h);
        if (ssmq::stream()) hipStreamSynchronize(ssmq::stream());
    }   // ... and nobody may hold the handle any more: destroying it while another thread uses it is the caller's error
    if (h->d_small) hipFree(h->d_small);
    if (h->d_wide) hipFree(h->d_wide);
    if (h->d_wc_pad) hipFree(h->d_wc_pad);
    if (h->d_wcx_pad) hipFree(h->d_wcx_pad);
    if (h->d_sx_pad) hipFree(h->d_sx_pad);
    if (h->d_sx_pan) hipFree(h->d_sx_pan);
    if (h->d_wc_blk) hipFree(h->d_wc_blk);
    if (h->d_ik_blk) hipFree(h->d_ik_blk);
    delete h;
}

int ssmq_transform_dims(const ssmq_transform *h, int *D, int *E, int *N) {
    SSMQ_HANDLE_LOCK(h);
    if (!h) return SSMQ_E_ARG;
    if (D) *D = h->D;
    if (E) *E = h->E;
    if (N) *N = h->N;
    return SSMQ_OK;
}

// ---- apply -----------------------------------------------------------------------------------------------------------
int ssmq_apply_batch_dev(ssmq_transform *h, const ssmq_integrand *f, int64_t B, int64_t ld, const double *d_mean,
                         const double *d_cov, const double *d_time, int time_stride, double *d_mean_f,
                         double *d_cov_f, double *d_cov_fx, int32_t *d_status) {
    SSMQ_HANDLE_LOCK(h);
    if (!h || !f) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    return apply_dev_impl(h, f, B, ld, d_mean, d_cov, d_time, time_stride, d_mean_f, d_cov_f, d_cov_fx, d_status,
                          nullptr, nullptr, false);
}

int ssmq_apply_kernel_name(const ssmq_transform *h, const ssmq_integrand *f, char *buf, int len) {
    SSMQ_HANDLE_LOCK(h);
    if (!h || !f || !buf || len <= 0) return SSMQ_E_ARG;
    const char *name = nullptr;
    int rc = apply_dev_impl(const_cast<ssmq_transform *>(h), f, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                            nullptr, nullptr, nullptr, &name, true);
    if (rc) return rc;
    snprintf(buf, len, "%s", name ? name : "");
    return SSMQ_OK;
}

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { return hip_fail(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc"); }
    double *d() { return (double *)p; }
};
}  // namespace

// ---- host arrays in the reference's study layout <-> time-major planes in HBM, through the pinned staging blocks --------
// The filters' device buffers are [n_outer][n_elem][ld] (time step, element, trajectory); the reference's arrays are
// (n_elem..., n_outer, B): dim_y x T x B measurements in, D x T x B means and D x D x T x B covariances out
// (ssinf.py:66-118).  Rows of B doubles are contiguous on both sides, so the permutation is a row copy: done on the host
// between the caller's array and the pinned block (several threads), one contiguous transfer per chunk.
namespace {
void copy_rows(bool to_planes, double *host, double *pinned, int64_t t0, int64_t t1, int n_outer, int n_elem, int64_t B,
               int64_t ld) {
    // planes row (t - t0, e) of the chunk <-> host row (e, t)
    const int64_t rows = (t1 - t0) * n_elem;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, (rows * B) / (256 * 1024)));
    auto work = [=](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t t = t0 + r / n_elem, e = r % n_elem;
            double *pl = pinned + r * ld, *hs = host + (e * n_outer + t) * B;
            if (to_planes) {
                memcpy(pl, hs, sizeof(double) * B);
                if (ld > B) memset(pl + B, 0, sizeof(double) * (ld - B));
            } else {
                memcpy(hs, pl, sizeof(double) * B);
            }
        }
    };
    std::vector<std::thread> th;
    const int64_t per = (rows + nt - 1) / nt;
    for (int k = 1; k < nt; ++k)
        if (k * per < rows) th.emplace_back(work, k * per, std::min(rows, (k + 1) * per));
    work(0, std::min(rows, per));
    for (auto &t : th) t.join();
}
constexpr size_t kPlaneChunkBytes = size_t(128) << 20;
}  // namespace

int ssmq_upload_planes(const double *host, int n_outer, int n_elem, int64_t B, int64_t ld, double *d_planes) {
    if (!host || !d_planes || n_outer < 0 || n_elem < 1 || B < 0 || ld < B) {
        set_error("upload_planes: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (n_outer == 0 || B == 0) return SSMQ_OK;
    hipStream_t s = stream();
    const size_t step_bytes = sizeof(double) * (size_t)n_elem * ld;
    const int64_t tc = std::max<int64_t>(1, std::min<int64_t>(n_outer, (int64_t)(kPlaneChunkBytes / step_bytes)));
    if ((rc = g_stage.reserve(0, step_bytes * tc, 0))) return rc;
    for (int64_t t0 = 0; t0 < n_outer; t0 += tc) {
        const int64_t t1 = std::min<int64_t>(n_outer, t0 + tc);
        copy_rows(true, const_cast<double *>(host), (double *)g_stage.hin, t0, t1, n_outer, n_elem, B, ld);
        SSMQ_HIP(hipMemcpyAsync(d_planes + (size_t)t0 * n_elem * ld, g_stage.hin, step_bytes * (t1 - t0), hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipStreamSynchronize(s));       // the pinned block is refilled for the next chunk
    }
    return SSMQ_OK;
}

int ssmq_download_planes(const double *d_planes, int n_outer, int n_elem, int64_t B, int64_t ld, double *host) {
    if (!host || !d_planes || n_outer < 0 || n_elem < 1 || B < 0 || ld < B) {
        set_error("download_planes: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (n_outer == 0 || B == 0) return SSMQ_OK;
    hipStream_t s = stream();
    const size_t step_bytes = sizeof(double) * (size_t)n_elem * ld;
    const int64_t tc = std::max<int64_t>(1, std::min<int64_t>(n_outer, (int64_t)(kPlaneChunkBytes / step_bytes)));
    if ((rc = g_stage.reserve(0, 0, step_bytes * tc))) return rc;
    for (int64_t t0 = 0; t0 < n_outer; t0 += tc) {
        const int64_t t1 = std::min<int64_t>(n_outer, t0 + tc);
        SSMQ_HIP(hipMemcpyAsync(g_stage.hout, d_planes + (size_t)t0 * n_elem * ld, step_bytes * (t1 - t0), hipMemcpyDeviceToHost, s));
        SSMQ_HIP(hipStreamSynchronize(s));
        copy_rows(false, host, (double *)g_stage.hout, t0, t1, n_outer, n_elem, B, ld);
    }
    return SSMQ_OK;
}

int ssmq_apply_batch(ssmq_transform *h, const ssmq_integrand *f, int64_t B, const double *mean, const double *cov,
                     const double *time, int time_stride, double *mean_f, double *cov_f, double *cov_fx,
                     int32_t *status) {
    SSMQ_HANDLE_LOCK(h);
    if (!h || !f || B < 0 || !mean || !cov || !mean_f || !cov_f || !cov_fx) {
        set_error("apply_batch: null argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    const int D = h->D, E = h->E;
    const int64_t ld = (B + 63) / 64 * 64;
    const size_t n_in = (size_t)D + (size_t)D * D, n_out = (size_t)E + (size_t)E * E + (size_t)E * D;
    const size_t n_time = time && time_stride ? (size_t)B : 1;
    // Small batches - the drop-in apply() is B = 1 - convert the layout on the host and move one pinned block each way:
    // one upload, one kernel, one download (the reference needs 60-120 us per apply(); six allocations, five layout
    // kernels and seven copies per call took longer than that).  Large batches transpose on the device.
    const bool host_layout = (size_t)B * (n_in + n_out) <= 65536;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    // device: [planes in | time] [planes out | status] and, for the device-side conversion, [AoS in | time] [AoS out | status]
    const size_t pin_bytes = sizeof(double) * (n_in * ld + n_time), pout_bytes = sizeof(double) * n_out * ld + sizeof(int32_t) * ld;
    const size_t ain_bytes = sizeof(double) * ((size_t)B * n_in + n_time), aout_bytes = sizeof(double) * B * n_out + sizeof(int32_t) * ld;
    const size_t off_in = 0, off_out = al(pin_bytes), off_ai = off_out + al(pout_bytes),
                 off_ao = off_ai + (host_layout ? 0 : al(ain_bytes)), total = off_ao + (host_layout ? 0 : al(aout_bytes));
    // every transfer goes through the pinned blocks: copies from / to pageable caller memory stalled for 20-30 ms at some
    // sizes (the runtime pins fresh pages on the fly)
    if ((rc = g_stage.reserve(total, host_layout ? pin_bytes : ain_bytes, host_layout ? pout_bytes : aout_bytes))) return rc;
    hipStream_t s = stream();
    char *dev = (char *)g_stage.dev;
    double *soa_in = (double *)(dev + off_in), *d_time = soa_in + n_in * ld;
    double *o_mf = (double *)(dev + off_out), *o_cf = o_mf + ld * E, *o_cfx = o_cf + ld * E * E;
    int32_t *d_st = (int32_t *)(o_cfx + ld * E * D);
    const double tzero = 0.0;
    double *hin = (double *)g_stage.hin;
    if (host_layout) {
        for (int e = 0; e < D; ++e) {
            double *pl = hin + (size_t)e * ld;
            for (int64_t i = 0; i < B; ++i) pl[i] = mean[(size_t)i * D + e];
            for (int64_t i = B; i < ld; ++i) pl[i] = 0.0;
        }
        for (int e = 0; e < D * D; ++e) {
            double *pl = hin + (size_t)(D + e) * ld;
            for (int64_t i = 0; i < B; ++i) pl[i] = cov[(size_t)i * D * D + e];
            for (int64_t i = B; i < ld; ++i) pl[i] = 0.0;
        }
        memcpy(hin + n_in * ld, time ? time : &tzero, sizeof(double) * n_time);
        SSMQ_HIP(hipMemcpyAsync(soa_in, hin, pin_bytes, hipMemcpyHostToDevice, s));
    } else {
        double *aos_in = (double *)(dev + off_ai);
        fast_copy(hin, mean, sizeof(double) * B * D);
        fast_copy(hin + (size_t)B * D, cov, sizeof(double) * B * D * D);
        memcpy(hin + (size_t)B * n_in, time ? time : &tzero, sizeof(double) * n_time);
        SSMQ_HIP(hipMemcpyAsync(aos_in, hin, ain_bytes, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(d_time, aos_in + (size_t)B * n_in, sizeof(double) * n_time, hipMemcpyDeviceToDevice, s));
        if ((rc = ssmq_aos_to_soa(aos_in, soa_in, D, B, ld))) return rc;
        if ((rc = ssmq_aos_to_soa(aos_in + B * D, soa_in + ld * D, D * D, B, ld))) return rc;
    }
    rc = apply_dev_impl(h, f, B, ld, soa_in, soa_in + ld * D, d_time, time && time_stride ? 1 : 0, o_mf, o_cf, o_cfx, d_st,
                        nullptr, nullptr, false);
    if (rc) return rc;
    const int32_t *hst;
    const double *ho = (const double *)g_stage.hout;
    if (host_layout) {
        SSMQ_HIP(hipMemcpyAsync(g_stage.hout, o_mf, pout_bytes, hipMemcpyDeviceToHost, s));
        SSMQ_HIP(hipStreamSynchronize(s));
        for (int e = 0; e < E; ++e)
            for (int64_t i = 0; i < B; ++i) mean_f[(size_t)i * E + e] = ho[(size_t)e * ld + i];
        const double *hc = ho + (size_t)E * ld;
        for (int e = 0; e < E * E; ++e)
            for (int64_t i = 0; i < B; ++i) cov_f[(size_t)i * E * E + e] = hc[(size_t)e * ld + i];
        const double *hx = hc + (size_t)E * E * ld;
        for (int e = 0; e < E * D; ++e)
            for (int64_t i = 0; i < B; ++i) cov_fx[(size_t)i * E * D + e] = hx[(size_t)e * ld + i];
        hst = (const int32_t *)(hx + (size_t)E * D * ld);
    } else {
        double *a_mf = (double *)(dev + off_ao), *a_cf = a_mf + B * E, *a_cfx = a_cf + B * E * E;
        if ((rc = ssmq_soa_to_aos(o_mf, a_mf, E, B, ld))) return rc;
        if ((rc = ssmq_soa_to_aos(o_cf, a_cf, E * E, B, ld))) return rc;
        if ((rc = ssmq_soa_to_aos(o_cfx, a_cfx, E * D, B, ld))) return rc;
        SSMQ_HIP(hipMemcpyAsync(a_mf + (size_t)B * n_out, d_st, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(g_stage.hout, a_mf, aout_bytes, hipMemcpyDeviceToHost, s));
        SSMQ_HIP(hipStreamSynchronize(s));
        fast_copy(mean_f, ho, sizeof(double) * B * E);
        fast_copy(cov_f, ho + (size_t)B * E, sizeof(double) * B * E * E);
        fast_copy(cov_fx, ho + (size_t)B * (E + E * E), sizeof(double) * B * E * D);
        hst = (const int32_t *)(ho + (size_t)B * n_out);
    }
    int first = 0;
    for (int64_t i = 0; i < B; ++i) {
        if (status) status[i] = hst[i];
        if (hst[i] && !first) first = (int)std::min<int64_t>(i + 1, 0x7fffffff);
    }
    return first;
}

int ssmq_sigma_points_batch(ssmq_transform *h, int64_t B, const double *mean, const double *cov, double *x,
                            double *chol, int32_t *status) {
    SSMQ_HANDLE_LOCK(h);
    if (h && h->form == SSMQ_FORM_TAYLOR1) {
        set_error("the linearisation transform has no sigma points");
        return SSMQ_E_UNSUPPORTED;
    }
    if (!h || B < 0 || !mean || !cov || !x || !chol) return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    const int D = h->D, N = h->N;
    if (wide_lds_bytes(D, h->E, N) > 160 * 1024 - 64) return SSMQ_E_UNSUPPORTED;
    // staging arena: [mean | cov] up, [x | chol | status] down, one transfer each way through the pinned blocks
    const size_t nb = (size_t)B, n_in = nb * ((size_t)D + (size_t)D * D), n_x = nb * D * N, n_l = nb * D * D;
    const size_t in_bytes = sizeof(double) * n_in, out_bytes = sizeof(double) * (n_x + n_l) + sizeof(int32_t) * nb;
    const size_t off_out = (in_bytes + 255) / 256 * 256;
    if ((rc = g_stage.reserve(off_out + out_bytes, in_bytes, out_bytes))) return rc;
    hipStream_t s = stream();
    double *dm = (double *)g_stage.dev, *dc = dm + nb * D;
    double *dx = (double *)((char *)g_stage.dev + off_out), *dl = dx + n_x;
    int32_t *ds = (int32_t *)(dl + n_l);
    double *hin = (double *)g_stage.hin;
    fast_copy(hin, mean, sizeof(double) * nb * D);
    fast_copy(hin + nb * D, cov, sizeof(double) * n_l);
    SSMQ_HIP(hipMemcpyAsync(dm, hin, in_bytes, hipMemcpyHostToDevice, s));
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.D = D; a.E = h->E; a.N = N; a.form = h->form; a.mode = SSMQ_WIDE_POINTS; a.consts = h->d_wide;
    a.cov_scale = a.ccov_scale = 1.0;
    a.mean = dm; a.cov = dc; a.es_in = 1; a.bs_mean = D; a.bs_cov = D * D; a.status = ds;
    a.x_out = dx; a.chol_out = dl;
    if ((rc = hip_fail(launch_apply_wide(a, B, s), "k_apply_wide(points)"))) return rc;
    SSMQ_HIP(hipMemcpyAsync(g_stage.hout, dx, out_bytes, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    const double *ho = (const double *)g_stage.hout;
    fast_copy(x, ho, sizeof(double) * n_x);
    fast_copy(chol, ho + n_x, sizeof(double) * n_l);
    const int32_t *st = (const int32_t *)(ho + n_x + n_l);
    int first = 0;
    for (int64_t i = 0; i < B; ++i) {
        if (status) status[i] = st[i];
        if (st[i] && !first) first = (int)std::min<int64_t>(i + 1, 0x7fffffff);
    }
    return first;
}

int ssmq_apply_fx_batch(ssmq_transform *h, int64_t B, const double *chol, const double *mean, const double *x,
                        const double *fx, double *mean_f, double *cov_f, double *cov_fx) {
    SSMQ_HANDLE_LOCK(h);
    if (h && h->form == SSMQ_FORM_TAYLOR1) {
        set_error("the linearisation transform has no sigma points");
        return SSMQ_E_UNSUPPORTED;
    }
    if (!h || B < 0 || !chol || !fx || !mean_f || !cov_f || !cov_fx) return SSMQ_E_ARG;
    if (h->form == SSMQ_FORM_SIGMA && (!mean || !x)) {
        set_error("apply_fx_batch: the centred form needs mean and x");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    const int D = h->D, E = h->E, N = h->N;
    const bool wide_fits = wide_lds_bytes(D, E, N) <= 160 * 1024 - 64;
    const bool centred = h->form == SSMQ_FORM_SIGMA;
    // point sets beyond the wave kernels without a fused matrix-core instantiation (as apply_dev_impl): blocked GEMM + rest
    const bool big = N > 64 && ((!centred && h->d_wc_blk && (B * E >= kGemmMinRows || !wide_fits) && (h->tp_nu <= 0.0 || h->d_ik_blk)) ||
                                (centred && !wide_fits));
    if (!big && !wide_fits) {
        set_error("apply_fx_batch: shape too large for the LDS-resident generic kernel");
        return SSMQ_E_UNSUPPORTED;
    }
    // staging arena: [chol | fx | mean | x] up, [mean_f | cov_f | cov_fx] down through the pinned blocks; the padded copies
    // of the matrix-core route behind them
    const bool gemm = !big && h->d_wc_pad && h->form == SSMQ_FORM_BQ && B * E >= kGemmMinRows;
    const size_t nb = (size_t)B, n_l = nb * D * D, n_fx = nb * E * N, n_m = centred ? nb * D : 0, n_x = centred ? nb * D * N : 0;
    const size_t n_out = nb * ((size_t)E + (size_t)E * E + (size_t)E * D);
    const size_t in_bytes = sizeof(double) * (n_l + n_fx + n_m + n_x), out_bytes = sizeof(double) * n_out;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const int big_kb = (N + 15) / 16, big_lda = big_kb * 16, big_ldt = (big && !centred) ? h->big_ncb * kBigCols : 0;
    const int big_nt = (big && !centred) ? (h->tp_nu > 0.0 ? 2 : 1) : 0;
    const size_t pad_bytes = gemm ? sizeof(double) * nb * E * h->np_pad : big ? sizeof(double) * nb * E * big_lda : 0;
    const size_t t_bytes = gemm ? pad_bytes : sizeof(double) * nb * E * (size_t)big_ldt * big_nt;
    const size_t off_out = al(in_bytes), off_fxp = off_out + al(out_bytes), off_tt = off_fxp + al(pad_bytes);
    if ((rc = g_stage.reserve(off_tt + al(t_bytes), in_bytes, out_bytes))) return rc;
    hipStream_t s = stream();
    char *dev = (char *)g_stage.dev;
    double *dl = (double *)dev, *dfx = dl + n_l, *dm = dfx + n_fx, *dx = dm + n_m;
    double *omf = (double *)(dev + off_out), *ocf = omf + nb * E, *ocfx = ocf + nb * E * E;
    double *hin = (double *)g_stage.hin;
    fast_copy(hin, chol, sizeof(double) * n_l);
    fast_copy(hin + n_l, fx, sizeof(double) * n_fx);
    if (centred) {
        fast_copy(hin + n_l + n_fx, mean, sizeof(double) * n_m);
        fast_copy(hin + n_l + n_fx + n_m, x, sizeof(double) * n_x);
    }
    SSMQ_HIP(hipMemcpyAsync(dl, hin, in_bytes, hipMemcpyHostToDevice, s));
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.D = D; a.E = E; a.N = N; a.form = h->form; a.mode = SSMQ_WIDE_FX; a.emv_mode = h->emv_mode; a.tp_nu = h->tp_nu;
    a.cov_scale = a.ccov_scale = 1.0;
    a.consts = h->d_wide; a.mean = dm; a.chol_in = dl; a.fx_in = dfx; a.x_in = dx;
    a.mean_f = omf; a.cov_f = ocf; a.cov_fx = ocfx; a.es_out = 1; a.bs_mf = E; a.bs_cf = E * E;
    a.bs_cfx = E * D;
    if (big) {
        const int64_t M = B * E;
        double *fxp = (double *)(dev + off_fxp), *ttp = (double *)(dev + off_tt);
        SSMQ_HIP(hipMemsetAsync(fxp, 0, sizeof(double) * M * big_lda, s));
        SSMQ_HIP(hipMemcpy2DAsync(fxp, sizeof(double) * big_lda, dfx, sizeof(double) * N, sizeof(double) * N, M,
                                  hipMemcpyDeviceToDevice, s));
        const WideLayout wl = wide_layout(D, E, N, h->form);
        if ((rc = launch_row_means(fxp, h->d_wide + wl.wm, M, big_lda, N, omf, s))) return rc;
        if (!centred && (rc = launch_fxwc_blocks(fxp, h->d_wc_blk, ttp, M, big_lda, big_ldt, big_kb, h->big_ncb, s))) return rc;
        if (big_nt == 2 && (rc = launch_fxwc_blocks(fxp, h->d_ik_blk, ttp + (size_t)M * big_ldt, M, big_lda, big_ldt, big_kb,
                                                    h->big_ncb, s)))
            return rc;
        BigRest r;
        memset(&r, 0, sizeof(r));
        r.D = D; r.E = E; r.N = N; r.form = h->form; r.emv_mode = h->emv_mode; r.tp_nu = h->tp_nu; r.cov_scale = r.ccov_scale = 1.0;
        r.consts = h->d_wide; r.fx = fxp; r.t = centred ? nullptr : ttp; r.t2 = big_nt == 2 ? ttp + (size_t)M * big_ldt : nullptr;
        r.lda = big_lda; r.ldt = big_ldt; r.p_col = 16 * big_kb; r.mean_rows = omf; r.chol = dl;
        r.cov_f = ocf; r.cov_fx = ocfx; r.es = 1; r.bs_cf = (int64_t)E * E; r.bs_cfx = (int64_t)E * D;
        if ((rc = launch_big_rest(r, B, s))) return rc;
        a.mode = -1;   // done
    } else if (gemm) {
        // matrix-core route: rows re-pitched to the padded column count, T = FX Wc for the whole batch, then the rest
        const int NP = h->np_pad;
        const int64_t M = B * E;
        double *fxp = (double *)(dev + off_fxp), *ttp = (double *)(dev + off_tt);
        SSMQ_HIP(hipMemsetAsync(fxp, 0, sizeof(double) * M * NP, s));
        SSMQ_HIP(hipMemcpy2DAsync(fxp, sizeof(double) * NP, dfx, sizeof(double) * N, sizeof(double) * N, M,
                                  hipMemcpyDeviceToDevice, s));
        if (h->tp_nu <= 0.0 && h->d_wcx_pad && fxwc_cov_supported(E) && D <= 16 && !ssmq::sw("SSMQ_NO_FUSED_COV")) {
            // means of the supplied values, then the GEMM whose epilogue forms both covariances (no T in memory)
            const WideLayout wl = wide_layout(D, E, N, h->form);
            if ((rc = launch_row_means(fxp, h->d_wide + wl.wm, M, NP, N, omf, s))) return rc;
            if ((rc = launch_fxwc_cov_mfma(NP, fxp, h->d_wcx_pad, M, NP, omf, dl, h->d_wide + wl.emv,
                                           h->emv_mode == SSMQ_EMV_BROADCAST ? 1 : 0, nullptr, 1.0, 1.0, E, D, ocf, ocfx, 1,
                                           (int64_t)E * E, (int64_t)E * D, s)))
                return rc;
            a.mode = -1;   // done
        } else {
            if ((rc = launch_fxwc_mfma(NP, fxp, h->d_wc_pad, ttp, M, NP, NP, s))) return rc;
            a.fx_ld = NP; a.fx_in = fxp; a.t_in = ttp;
        }
    }
    if (a.mode != -1 && (rc = hip_fail(launch_apply_wide(a, B, s), "k_apply_wide(fx)"))) return rc;
    SSMQ_HIP(hipMemcpyAsync(g_stage.hout, omf, out_bytes, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    const double *ho = (const double *)g_stage.hout;
    fast_copy(mean_f, ho, sizeof(double) * nb * E);
    fast_copy(cov_f, ho + nb * E, sizeof(double) * nb * E * E);
    fast_copy(cov_fx, ho + nb * (E + (size_t)E * E), sizeof(double) * nb * E * D);
    return SSMQ_OK;
}

// T = FX Wc on the matrix cores for device-resident integrand values (the GEMM-shaped stage of a large-N BQ transform)
int ssmq_fxwc_batch_dev(ssmq_transform *h, int64_t M, const double *d_fx, int64_t ld_fx, double *d_t, int64_t ld_t,
                        int *n_padded) {
    SSMQ_HANDLE_LOCK(h);
    if (h && h->form == SSMQ_FORM_TAYLOR1) {
        set_error("the linearisation transform has no sigma points");
        return SSMQ_E_UNSUPPORTED;
    }
    if (!h || M < 0 || (M > 0 && (!d_fx || !d_t))) {
        set_error("fxwc_batch: bad argument");
        return SSMQ_E_ARG;
    }
    if (n_padded) *n_padded = h->np_pad;
    if (!h->d_wc_pad) {
        set_error("fxwc_batch: this transform has no matrix-core instantiation (BQ form, N in 113..128, 193..208, 241..256)");
        return SSMQ_E_UNSUPPORTED;
    }
    if (M == 0) return SSMQ_OK;
    if (ld_fx < h->np_pad || ld_t < h->np_pad || ld_fx > 0x7fffffff || ld_t > 0x7fffffff || (ld_fx & 1)) {
        set_error("fxwc_batch: row pitches must be even and at least the padded point count");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    return launch_fxwc_mfma(h->np_pad, d_fx, h->d_wc_pad, d_t, M, (int)ld_fx, (int)ld_t, stream());
}

// ---- filter recursion around the path ---------------------------------------------------------------------------------
int ssmq_kalman_update_dev(int D, int Y, int64_t B, int64_t ld, const double *d_m_pr, const double *d_P_pr,
                           const double *d_y_mean, const double *d_P_y, const double *d_P_yx, const double *d_y,
                           double *d_m_fi, double *d_P_fi, int32_t *d_status) {
    if (D < 1 || Y < 1 || B < 0 || ld < B || !d_m_pr || !d_P_pr || !d_y_mean || !d_P_y || !d_P_yx || !d_y || !d_m_fi ||
        !d_P_fi || !d_status)
        return SSMQ_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    SSMQ_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * B, stream()));
    return launch_kalman_update(D, Y, B, ld, d_m_pr, d_P_pr, d_y_mean, d_P_y, d_P_yx, d_y, d_m_fi, d_P_fi, d_status,
                                stream());
}

}  // extern "C"

namespace ssmq {
int launch_kalman_update_ex(int D, int Y, int64_t B, int64_t ld, const double *m_pr, const double *P_pr,
                            const double *y_mean, const double *P_y, const double *P_yx, const double *y,
                            double *m_fi, double *P_fi, int32_t *status, const int32_t *st_a, const int32_t *st_b,
                            int step, hipStream_t s, double student_dof, double *smat_out, int Dx);
int launch_augment(const double *m, const double *P, const double *nmean, const double *ncov, double *ma, double *Pa,
                   int D, int Dn, int64_t B, int64_t ld, hipStream_t s);
}

namespace ssmq {
int try_launch_fused(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho,
                     const ssmq_integrand *fo, int sel_obs, int64_t B, int64_t ld, int T, const double *d_y,
                     const double *d_m0, const double *d_P0, const double *d_gqg, const double *d_rr, double *d_fm,
                     double *d_fP, int32_t *d_status, hipStream_t s, const char **name, bool dry_run,
                     const double *d_sscale, double student_dof, const double *d_ttab_dyn, const double *d_ttab_obs);
int try_launch_fused_aug(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho,
                         const ssmq_integrand *fo, int sel_obs, int D, int dq, int dr, int64_t B, int64_t ld, int T,
                         const double *d_y, const double *d_m0, const double *d_P0, const double *d_add_dyn,
                         const double *d_add_obs, const double *d_noise, double *d_fm, double *d_fP, int32_t *d_status,
                         hipStream_t s, const char **name, bool dry_run, const double *d_ttab_dyn,
                         const double *d_ttab_obs, double *d_pm, double *d_pP, double *d_pC);
}

namespace {
// Grow-only device workspace + captured launch sequence of the filter loop, kept between calls so that a repeated
// forward pass (Monte-Carlo studies, bench.py) neither allocates nor pays 3 T kernel-launch latencies: the whole time
// loop is replayed as one hipGraph while the argument set is unchanged.
struct FilterCache {
    void *ws = nullptr;
    size_t ws_bytes = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<uint64_t> key;
    std::vector<double> gqg, rr, ss;
    int T = -1, fid_dyn = -1, fid_obs = -1, D = -1, Y = -1;
    int64_t ld = -1;          // the constants sit behind the ld-sized planes: a new pitch moves them
    bool consts_ok = false;
    void drop_graph() {
        if (exec) hipGraphExecDestroy(exec);
        if (graph) hipGraphDestroy(graph);
        exec = nullptr;
        graph = nullptr;
        key.clear();
    }
};
FilterCache &fc_of_ctx() {
    Ctx &c = ssmq::ctx();
    if (!c.fc) c.fc = new FilterCache;
    return *(FilterCache *)c.fc;
}
#define g_fc (fc_of_ctx())
// drops the captured loop on every exit of a scope whose temporaries the graph points into
struct GraphDropGuard {
    ~GraphDropGuard() { g_fc.drop_graph(); }
};
}  // namespace

namespace ssmq {
void reset_wide_attributes();
void drop_staging_arena();
void drop_theta_step_graphs();
void drop_multi_cache();
void drop_pipe_cache();
void reset_device_caches() {
    g_fc.drop_graph();
    drop_theta_step_graphs();
    g_fc.consts_ok = false;
    if (g_fc.ws) hipFree(g_fc.ws);
    g_fc.ws = nullptr;
    g_fc.ws_bytes = 0;
    drop_gemm_scratch();
    drop_staging_arena();
    reset_wide_attributes();
    drop_multi_cache();
    drop_pipe_cache();
    Ctx &c = ctx();
    if (c.strip_buf) hipFree(c.strip_buf);
    c.strip_buf = nullptr;
    c.strip_bytes = 0;
}
}  // namespace ssmq

// sscale (host, [T]) / student_dof: Studentian recursion (ssinf.py:634-736); null / 0 for the Gaussian filters.
int filter_forward_impl(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                               const ssmq_integrand *f_obs, int64_t B, int64_t ld, int T, const double *d_y,
                               const double *d_m0, const double *d_P0, const double *GQG, const double *R,
                               double *d_fm, double *d_fP, int32_t *d_status, const double *sscale,
                               double student_dof, double *d_pm = nullptr, double *d_pP = nullptr,
                               double *d_pC = nullptr) {
    if (!h_dyn || !h_obs || !f_dyn || !f_obs || B < 0 || ld < B || T < 0 || !d_y || !d_m0 || !d_P0 || !d_fm || !d_fP ||
        !d_status) {
        set_error("filter_forward: bad argument");
        return SSMQ_E_ARG;
    }
    const int D = h_dyn->D, Y = h_obs->E;
    if (h_dyn->E != D || h_obs->D != D) {
        set_error("filter_forward: additive-noise filter needs dyn (D -> D) and obs (D -> Y) transforms");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    if (T == 0) {   // nothing to filter: every trajectory is trivially fine
        SSMQ_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * ld, stream()));
        return SSMQ_OK;
    }
    hipStream_t s = stream();
    // workspace carve-up (doubles first, then the two int32 status planes)
    const size_t n_dbl = (size_t)ld * (D + 3 * D * D + Y + Y * Y + Y * D) + 4 * (size_t)T + D * D + Y * Y;
    const size_t need = sizeof(double) * n_dbl + 2 * sizeof(int32_t) * (size_t)ld;
    if (g_fc.ws_bytes < need) {
        g_fc.drop_graph();
        g_fc.consts_ok = false;
        if (g_fc.ws) hipFree(g_fc.ws);
        g_fc.ws = nullptr;
        g_fc.ws_bytes = 0;
        SSMQ_HIP(hipMalloc(&g_fc.ws, need));
        g_fc.ws_bytes = need;
    }
    double *w = (double *)g_fc.ws;
    double *m_pr = w; w += (size_t)ld * D;
    double *P_pr = w; w += (size_t)ld * D * D;
    double *C_xx = w; w += (size_t)ld * D * D;
    double *y_mean = w; w += (size_t)ld * Y;
    double *P_y = w; w += (size_t)ld * Y * Y;
    double *P_yx = w; w += (size_t)ld * Y * D;
    double *smat = w; w += (size_t)ld * D * D;   // Studentian: rescaled scale matrix fed to the next time update
    double *tvec = w; w += T;
    double *svec = w; w += T;
    double *ttab_d = w; w += T;   // time tables of the two integrands (UNGM: 8 cos(1.2 k)), see time_table()
    double *ttab_o = w; w += T;
    double *gqg = w; w += D * D;
    double *rr = w; w += Y * Y;
    int32_t *st_a = (int32_t *)w, *st_b = st_a + ld;

    std::vector<double> hg(D * D, 0.0), hr(Y * Y, 0.0);
    if (GQG) hg.assign(GQG, GQG + D * D);
    if (R) hr.assign(R, R + Y * Y);
    std::vector<double> hs(T, 1.0);
    if (sscale) hs.assign(sscale, sscale + T);
    std::vector<uint64_t> key = {(uint64_t)(uintptr_t)h_dyn, (uint64_t)(uintptr_t)h_obs, (uint64_t)B, (uint64_t)ld,
                                 (uint64_t)T, (uint64_t)(uintptr_t)d_y, (uint64_t)(uintptr_t)d_m0,
                                 (uint64_t)(uintptr_t)d_P0, (uint64_t)(uintptr_t)d_fm, (uint64_t)(uintptr_t)d_fP,
                                 (uint64_t)(uintptr_t)d_status, (uint64_t)(uintptr_t)h_dyn->d_small,
                                 (uint64_t)(uintptr_t)g_fc.ws, (uint64_t)(uintptr_t)d_pm, (uint64_t)(uintptr_t)d_pP,
                                 (uint64_t)(uintptr_t)d_pC};
    const unsigned char *fb = (const unsigned char *)f_dyn;
    for (size_t i = 0; i + 8 <= sizeof(ssmq_integrand); i += 8) { uint64_t v; memcpy(&v, fb + i, 8); key.push_back(v); }
    fb = (const unsigned char *)f_obs;
    for (size_t i = 0; i + 8 <= sizeof(ssmq_integrand); i += 8) { uint64_t v; memcpy(&v, fb + i, 8); key.push_back(v); }
    key.push_back((uint64_t)h_dyn->emv_mode * 2 + (uint64_t)h_obs->emv_mode);
    key.push_back(((uint64_t)D << 48) | ((uint64_t)Y << 32) | ((uint64_t)h_dyn->N << 16) | (uint64_t)h_obs->N);
    key.push_back(((uint64_t)h_dyn->form << 2) | (uint64_t)h_obs->form);
    key.push_back((uint64_t)(uintptr_t)h_obs->d_small);
    // which kernel variant apply_dev_impl picks depends on the fast paths the handle's CURRENT constants qualify for:
    // ssmq_transform_update keeps the block addresses but may withdraw SSMQ_OPT_LDL (and zero its factors)
    key.push_back(((uint64_t)(uint32_t)h_dyn->opt_mask << 32) | (uint64_t)(uint32_t)h_obs->opt_mask);
    key.push_back(((uint64_t)(uint32_t)h_dyn->np_pad << 32) | (uint64_t)(uint32_t)h_obs->np_pad);
    key.push_back(((uint64_t)h_dyn->generation << 32) ^ (uint64_t)h_obs->generation);
    { uint64_t v; memcpy(&v, &h_dyn->tp_nu, 8); key.push_back(v); memcpy(&v, &h_obs->tp_nu, 8); key.push_back(v);
      memcpy(&v, &student_dof, 8); key.push_back(v); key.push_back(sscale ? 1 : 0); }

    std::vector<double> htd(T), hto(T);
    const bool has_td = time_table(f_dyn->id, T, htd.data()), has_to = time_table(f_obs->id, T, hto.data());
    if (!(g_fc.consts_ok && g_fc.gqg == hg && g_fc.rr == hr && g_fc.T == T && g_fc.ss == hs &&
          g_fc.fid_dyn == f_dyn->id && g_fc.fid_obs == f_obs->id && g_fc.ld == ld && g_fc.D == D && g_fc.Y == Y)) {
        g_fc.drop_graph();
        std::vector<double> tv(T);
        for (int k = 0; k < T; ++k) tv[k] = (double)k;  // both transforms of step k + 1 use time index k (ssinf.py:104)
        SSMQ_HIP(hipMemcpyAsync(tvec, tv.data(), sizeof(double) * T, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(gqg, hg.data(), sizeof(double) * D * D, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(rr, hr.data(), sizeof(double) * Y * Y, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(svec, hs.data(), sizeof(double) * T, hipMemcpyHostToDevice, s));
        if (has_td) SSMQ_HIP(hipMemcpyAsync(ttab_d, htd.data(), sizeof(double) * T, hipMemcpyHostToDevice, s));
        if (has_to) SSMQ_HIP(hipMemcpyAsync(ttab_o, hto.data(), sizeof(double) * T, hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipStreamSynchronize(s));
        g_fc.fid_dyn = f_dyn->id;
        g_fc.fid_obs = f_obs->id;
        g_fc.ss = hs;
        g_fc.gqg = hg;
        g_fc.rr = hr;
        g_fc.T = T;
        g_fc.ld = ld;
        g_fc.D = D;
        g_fc.Y = Y;
        g_fc.consts_ok = true;
    }
    // one fused kernel for the whole time loop when this (models, shapes, form) combination has one (it does not keep
    // the predictive moments, so a pass that has to store them for the smoother takes the launch loop)
    const bool keep_pred = d_pm && d_pP && d_pC;
    if (!ssmq::sw("SSMQ_NO_FUSED") && !keep_pred) {
        FInfo fio;
        if (!integrand_info(f_obs->id, &fio)) {
            set_error("unknown integrand id");
            return SSMQ_E_ARG;
        }
        rc = try_launch_fused(h_dyn, f_dyn, h_obs, f_obs, sel_pattern(f_obs, fio.din), B, ld, T, d_y, d_m0, d_P0, gqg,
                              rr, d_fm, d_fP, d_status, s, nullptr, false, sscale ? svec : nullptr, student_dof,
                              has_td ? ttab_d : nullptr, has_to ? ttab_o : nullptr);
        if (rc < 0) return rc;
        if (rc == 1) return SSMQ_OK;
    }
    if (!ssmq::sw("SSMQ_NO_FUSED") && keep_pred && !sscale && student_dof == 0.0) {
        // smoother: the time loop in one kernel that also leaves the predictive moments of every step in HBM
        FInfo fio;
        if (!integrand_info(f_obs->id, &fio)) {
            set_error("unknown integrand id");
            return SSMQ_E_ARG;
        }
        rc = try_launch_fused_aug(h_dyn, f_dyn, h_obs, f_obs, sel_pattern(f_obs, fio.din), D, 0, 0, B, ld, T, d_y, d_m0, d_P0,
                                  gqg, rr, gqg, d_fm, d_fP, d_status, s, nullptr, false, has_td ? ttab_d : nullptr,
                                  has_to ? ttab_o : nullptr, d_pm, d_pP, d_pC);
        if (rc < 0) return rc;
        if (rc == 1) return SSMQ_OK;
    }
    if (!(g_fc.exec && g_fc.key == key)) {
        g_fc.drop_graph();
        SSMQ_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        rc = hip_fail(hipMemsetAsync(d_status, 0, sizeof(int32_t) * ld, s), "hipMemsetAsync");
        for (int k = 0; k < T && !rc; ++k) {
            const double *m_in = k == 0 ? d_m0 : d_fm + (int64_t)(k - 1) * D * ld;
            const double *P_in = k == 0 ? d_P0 : (student_dof > 0.0 ? smat : d_fP + (int64_t)(k - 1) * D * D * ld);
            if (keep_pred) {   // predictive moments of every step stay in HBM for the backward pass (ssinf.py:105-107)
                m_pr = d_pm + (int64_t)k * D * ld;
                P_pr = d_pP + (int64_t)k * D * D * ld;
                C_xx = d_pC + (int64_t)k * D * D * ld;
            }
            rc = apply_dev_impl(h_dyn, f_dyn, B, ld, m_in, P_in, tvec + k, 0, m_pr, P_pr, C_xx, st_a, gqg, nullptr, false,
                                hs[k], 1.0, has_td ? ttab_d : nullptr, false);
            if (!rc)
                rc = apply_dev_impl(h_obs, f_obs, B, ld, m_pr, P_pr, tvec + k, 0, y_mean, P_y, P_yx, st_b, rr, nullptr,
                                    false, hs[k], hs[k], has_to ? ttab_o : nullptr, false);
            if (!rc)
                rc = launch_kalman_update_ex(D, Y, B, ld, m_pr, P_pr, y_mean, P_y, P_yx, d_y + (int64_t)k * Y * ld,
                                             d_fm + (int64_t)k * D * ld, d_fP + (int64_t)k * D * D * ld, d_status,
                                             st_a, st_b, k, s, student_dof, smat, 0);
        }
        hipGraph_t g = nullptr;
        hipError_t ce = hipStreamEndCapture(s, &g);
        if (rc) {
            if (g) hipGraphDestroy(g);
            return rc;
        }
        SSMQ_HIP(ce);
        g_fc.graph = g;
        SSMQ_HIP(hipGraphInstantiate(&g_fc.exec, g_fc.graph, nullptr, nullptr, 0));
        g_fc.key = key;
    }
    SSMQ_HIP(hipGraphLaunch(g_fc.exec, s));
    return SSMQ_OK;
}

extern "C" int ssmq_filter_forward_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                       const ssmq_integrand *f_obs, int64_t B, int64_t ld, int T, const double *d_y,
                                       const double *d_m0, const double *d_P0, const double *GQG, const double *R,
                                       double *d_fm, double *d_fP, int32_t *d_status) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    return filter_forward_impl(h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, GQG, R, d_fm, d_fP, d_status,
                               nullptr, 0.0);
}

// Filters whose models take the noise as an argument (ssinf.py:271-272, 282-283, 294-295): the moments are augmented with
// the noise statistics before each transform and the cross-covariance is cut back to the state columns. Plain launch
// loop (augment | apply | augment | apply | update per step); no fused kernel and no graph cache for this path yet.
// d_pm / d_pP / d_pC (all or none): predictive mean [T][D][ld], covariance [T][D*D][ld] and dynamics cross-covariance of
// every step for the RTS pass; *c_cols returns the number of columns stored per row of d_pC (D from the fused kernel,
// D + dq from the launch loop, whose transform writes the full E x (D + dq) block: d_pC must hold T * D * (D + dq) planes).
static int filter_forward_aug_impl(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                   const ssmq_integrand *f_obs, int dim_state, int64_t B, int64_t ld, int T,
                                   const double *d_y, const double *d_m0, const double *d_P0, const double *q_mean,
                                   const double *q_cov, int dq, const double *r_mean, const double *r_cov, int dr,
                                   double *d_fm, double *d_fP, int32_t *d_status, double *d_pm, double *d_pP,
                                   double *d_pC, int *c_cols) {
    if (!h_dyn || !h_obs || !f_dyn || !f_obs || dim_state <= 0 || dq < 0 || dr < 0 || B < 0 || ld < B || T < 0 || !d_y ||
        !d_m0 || !d_P0 || !d_fm || !d_fP || !d_status || (dq > 0 && (!q_mean || !q_cov)) ||
        (dr > 0 && (!r_mean || !r_cov))) {
        set_error("filter_forward_aug: bad argument");
        return SSMQ_E_ARG;
    }
    const int D = dim_state, Da = D + dq, Do = D + dr, Y = h_obs->E;
    if (h_dyn->D != Da || h_dyn->E != D || h_obs->D != Do) {
        set_error("filter_forward_aug: transforms must be (dim_state + dq -> dim_state) and (dim_state + dr -> dim_y)");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    hipStream_t s = stream();
    if (T == 0) {
        SSMQ_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * ld, s));
        SSMQ_HIP(hipStreamSynchronize(s));
        return SSMQ_OK;
    }
    // small constants first (time tables, noise statistics); the plane workspace only if the launch loop is needed
    DevBuf cs;
    const size_t n_noise = (size_t)dq + (dq ? (size_t)dq * dq : (size_t)D * D) + dr + (dr ? (size_t)dr * dr : (size_t)Y * Y);
    if ((rc = cs.alloc(sizeof(double) * (3 * (size_t)T + n_noise)))) return rc;
    double *w = cs.d();
    double *tvec = w; w += T;
    double *ttab_d = w; w += T;
    double *ttab_o = w; w += T;
    double *d_qm = w; w += dq;
    double *d_qc = w; w += dq ? (size_t)dq * dq : (size_t)D * D;
    double *d_rm = w; w += dr;
    double *d_rc = w; w += dr ? (size_t)dr * dr : (size_t)Y * Y;

    std::vector<double> tv(T), htd(T), hto(T);
    for (int k = 0; k < T; ++k) tv[k] = (double)k;   // both transforms of step k + 1 use time index k (ssinf.py:104)
    const bool has_td = time_table(f_dyn->id, T, htd.data()), has_to = time_table(f_obs->id, T, hto.data());
    SSMQ_HIP(hipMemcpyAsync(tvec, tv.data(), sizeof(double) * T, hipMemcpyHostToDevice, s));
    if (has_td) SSMQ_HIP(hipMemcpyAsync(ttab_d, htd.data(), sizeof(double) * T, hipMemcpyHostToDevice, s));
    if (has_to) SSMQ_HIP(hipMemcpyAsync(ttab_o, hto.data(), sizeof(double) * T, hipMemcpyHostToDevice, s));
    std::vector<double> zq((size_t)D * D, 0.0), zr((size_t)Y * Y, 0.0);
    if (dq) SSMQ_HIP(hipMemcpyAsync(d_qm, q_mean, sizeof(double) * dq, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(d_qc, q_cov ? q_cov : zq.data(), sizeof(double) * (dq ? (size_t)dq * dq : (size_t)D * D),
                            hipMemcpyHostToDevice, s));
    if (dr) SSMQ_HIP(hipMemcpyAsync(d_rm, r_mean, sizeof(double) * dr, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(d_rc, r_cov ? r_cov : zr.data(), sizeof(double) * (dr ? (size_t)dr * dr : (size_t)Y * Y),
                            hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * ld, s));
    SSMQ_HIP(hipStreamSynchronize(s));   // the host staging vectors above go out of scope with this call

    // one kernel for the whole time loop when this combination has an instantiation (ssmq_filter_fused.hip)
    if (!ssmq::sw("SSMQ_NO_FUSED")) {
        FInfo fio;
        if (!integrand_info(f_obs->id, &fio)) {
            set_error("unknown integrand id");
            return SSMQ_E_ARG;
        }
        // noise block q_mean | q_cov | r_mean | r_cov and the additive terms (zeros for a non-additive model)
        std::vector<double> hn, ha((size_t)D * D + (size_t)Y * Y, 0.0);
        for (int i = 0; i < dq; ++i) hn.push_back(q_mean[i]);
        for (int i = 0; i < dq * dq; ++i) hn.push_back(q_cov[i]);
        for (int i = 0; i < dr; ++i) hn.push_back(r_mean[i]);
        for (int i = 0; i < dr * dr; ++i) hn.push_back(r_cov[i]);
        hn.push_back(0.0);
        if (!dq && q_cov) for (int i = 0; i < D * D; ++i) ha[i] = q_cov[i];
        if (!dr && r_cov) for (int i = 0; i < Y * Y; ++i) ha[(size_t)D * D + i] = r_cov[i];
        DevBuf dn, da;
        if ((rc = dn.alloc(sizeof(double) * hn.size())) || (rc = da.alloc(sizeof(double) * ha.size()))) return rc;
        SSMQ_HIP(hipMemcpyAsync(dn.p, hn.data(), sizeof(double) * hn.size(), hipMemcpyHostToDevice, s));
        SSMQ_HIP(hipMemcpyAsync(da.p, ha.data(), sizeof(double) * ha.size(), hipMemcpyHostToDevice, s));
        rc = try_launch_fused_aug(h_dyn, f_dyn, h_obs, f_obs, sel_pattern(f_obs, fio.din), D, dq, dr, B, ld, T, d_y, d_m0,
                                  d_P0, da.d(), da.d() + (size_t)D * D, dn.d(), d_fm, d_fP, d_status, s, nullptr, false,
                                  has_td ? ttab_d : nullptr, has_to ? ttab_o : nullptr, d_pm, d_pP, d_pC);
        hipError_t e = hipStreamSynchronize(s);
        if (rc < 0) return rc;
        SSMQ_HIP(e);
        if (rc == 1) {
            if (c_cols) *c_cols = D;
            return SSMQ_OK;
        }
        rc = 0;
    }

    DevBuf ws, st;
    const size_t n_dbl = (size_t)ld * (Da + Da * Da + D + D * D + D * Da + Do + Do * Do + Y + Y * Y + Y * Do);
    if ((rc = ws.alloc(sizeof(double) * n_dbl)) || (rc = st.alloc(2 * sizeof(int32_t) * (size_t)ld))) return rc;
    w = ws.d();
    double *ma = w; w += (size_t)ld * Da;
    double *Pa = w; w += (size_t)ld * Da * Da;
    double *m_pr = w; w += (size_t)ld * D;
    double *P_pr = w; w += (size_t)ld * D * D;
    double *C_xx = w; w += (size_t)ld * D * Da;
    double *mo = w; w += (size_t)ld * Do;
    double *Po = w; w += (size_t)ld * Do * Do;
    double *y_mean = w; w += (size_t)ld * Y;
    double *P_y = w; w += (size_t)ld * Y * Y;
    double *P_yx = w; w += (size_t)ld * Y * Do;
    int32_t *st_a = (int32_t *)st.p, *st_b = st_a + ld;

    if (c_cols) *c_cols = Da;
    for (int k = 0; k < T && !rc; ++k) {
        const double *m_in = k == 0 ? d_m0 : d_fm + (int64_t)(k - 1) * D * ld;
        const double *P_in = k == 0 ? d_P0 : d_fP + (int64_t)(k - 1) * D * D * ld;
        if (d_pm) {      // predictive moments of every step stay in HBM for the backward pass (ssinf.py:105-107)
            m_pr = d_pm + (int64_t)k * D * ld;
            P_pr = d_pP + (int64_t)k * D * D * ld;
            C_xx = d_pC + (int64_t)k * D * Da * ld;
        }
        if (dq) {
            rc = launch_augment(m_in, P_in, d_qm, d_qc, ma, Pa, D, dq, B, ld, s);
            if (!rc)
                rc = apply_dev_impl(h_dyn, f_dyn, B, ld, ma, Pa, tvec + k, 0, m_pr, P_pr, C_xx, st_a, nullptr, nullptr, false,
                                    1.0, 1.0, has_td ? ttab_d : nullptr, false);
        } else {
            rc = apply_dev_impl(h_dyn, f_dyn, B, ld, m_in, P_in, tvec + k, 0, m_pr, P_pr, C_xx, st_a, d_qc, nullptr, false,
                                1.0, 1.0, has_td ? ttab_d : nullptr, false);
        }
        if (rc) break;
        if (dr) {
            rc = launch_augment(m_pr, P_pr, d_rm, d_rc, mo, Po, D, dr, B, ld, s);
            if (!rc)
                rc = apply_dev_impl(h_obs, f_obs, B, ld, mo, Po, tvec + k, 0, y_mean, P_y, P_yx, st_b, nullptr, nullptr, false,
                                    1.0, 1.0, has_to ? ttab_o : nullptr, false);
        } else {
            rc = apply_dev_impl(h_obs, f_obs, B, ld, m_pr, P_pr, tvec + k, 0, y_mean, P_y, P_yx, st_b, d_rc, nullptr, false,
                                1.0, 1.0, has_to ? ttab_o : nullptr, false);
        }
        if (!rc)
            rc = launch_kalman_update_ex(D, Y, B, ld, m_pr, P_pr, y_mean, P_y, P_yx, d_y + (int64_t)k * Y * ld,
                                         d_fm + (int64_t)k * D * ld, d_fP + (int64_t)k * D * D * ld, d_status, st_a, st_b,
                                         k, s, 0.0, nullptr, Do);
    }
    hipError_t se = hipStreamSynchronize(s);   // workspace is released on return
    if (rc) return rc;
    SSMQ_HIP(se);
    return SSMQ_OK;
}

extern "C" int ssmq_filter_forward_aug_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                           const ssmq_integrand *f_obs, int dim_state, int64_t B, int64_t ld, int T,
                                           const double *d_y, const double *d_m0, const double *d_P0,
                                           const double *q_mean, const double *q_cov, int dq, const double *r_mean,
                                           const double *r_cov, int dr, double *d_fm, double *d_fP,
                                           int32_t *d_status) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    return filter_forward_aug_impl(h_dyn, f_dyn, h_obs, f_obs, dim_state, B, ld, T, d_y, d_m0, d_P0, q_mean, q_cov, dq,
                                   r_mean, r_cov, dr, d_fm, d_fP, d_status, nullptr, nullptr, nullptr, nullptr);
}

namespace ssmq {
int launch_rts_backward(int D, int64_t B, int64_t ld, int T, const double *fm, const double *fP, const double *pm,
                        const double *pP, const double *pC, double *sm, double *sP, int32_t *status, hipStream_t s,
                        int c_cols);
}

// Forward pass + RTS smoother for models that take their noise as an argument: backward_pass of the reference is model-
// agnostic (ssinf.py:120-147, 325-344); the cross-covariance it needs is the one _time_update cut back to the state
// columns (:294-295).  Arguments as ssmq_filter_forward_aug_dev, outputs as ssmq_filter_smooth_dev.  Synchronous.
extern "C" int ssmq_filter_smooth_aug_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                          const ssmq_integrand *f_obs, int dim_state, int64_t B, int64_t ld, int T,
                                          const double *d_y, const double *d_m0, const double *d_P0,
                                          const double *q_mean, const double *q_cov, int dq, const double *r_mean,
                                          const double *r_cov, int dr, double *d_fm, double *d_fP, double *d_sm,
                                          double *d_sP, int32_t *d_status) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!h_dyn || !d_sm || !d_sP || dim_state <= 0 || dq < 0 || B < 0 || T < 0 || ld < B) {
        set_error("filter_smooth_aug: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    const int D = dim_state;
    if (T == 0)
        return filter_forward_aug_impl(h_dyn, f_dyn, h_obs, f_obs, D, B, ld, T, d_y, d_m0, d_P0, q_mean, q_cov, dq, r_mean,
                                       r_cov, dr, d_fm, d_fP, d_status, nullptr, nullptr, nullptr, nullptr);
    DevBuf pm, pP, pC;
    if ((rc = pm.alloc(sizeof(double) * (size_t)T * D * ld)) || (rc = pP.alloc(sizeof(double) * (size_t)T * D * D * ld)) ||
        (rc = pC.alloc(sizeof(double) * (size_t)T * D * (D + dq) * ld)))
        return rc;
    int c_cols = D;
    rc = filter_forward_aug_impl(h_dyn, f_dyn, h_obs, f_obs, D, B, ld, T, d_y, d_m0, d_P0, q_mean, q_cov, dq, r_mean, r_cov,
                                 dr, d_fm, d_fP, d_status, pm.d(), pP.d(), pC.d(), &c_cols);
    if (rc) return rc;
    rc = launch_rts_backward(D, B, ld, T, d_fm, d_fP, pm.d(), pP.d(), pC.d(), d_sm, d_sP, d_status, stream(), c_cols);
    hipError_t e = hipStreamSynchronize(stream());
    if (rc) return rc;
    SSMQ_HIP(e);
    return SSMQ_OK;
}

// backward_pass alone (ssinf.py:120-147, 325-344) over moments a caller kept from its own forward pass - the marginalised
// filter, whose forward pass is driven from the host (BFGS per step)
extern "C" int ssmq_rts_backward_dev(int D, int64_t B, int64_t ld, int T, const double *d_fm, const double *d_fP,
                                     const double *d_pm, const double *d_pP, const double *d_pC, double *d_sm, double *d_sP,
                                     int32_t *d_status) {
    if (D < 1 || B < 0 || T < 0 || ld < B || !d_fm || !d_fP || !d_pm || !d_pP || !d_pC || !d_sm || !d_sP || !d_status) {
        set_error("rts_backward: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || T == 0) return SSMQ_OK;
    rc = launch_rts_backward(D, B, ld, T, d_fm, d_fP, d_pm, d_pP, d_pC, d_sm, d_sP, d_status, stream(), D);
    hipError_t e = hipStreamSynchronize(stream());
    if (rc) return rc;
    SSMQ_HIP(e);
    return SSMQ_OK;
}

extern "C" int ssmq_filter_smooth_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                      const ssmq_integrand *f_obs, int64_t B, int64_t ld, int T, const double *d_y,
                                      const double *d_m0, const double *d_P0, const double *GQG, const double *R,
                                      double *d_fm, double *d_fP, double *d_sm, double *d_sP, int32_t *d_status) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!h_dyn || !d_sm || !d_sP || B < 0 || T < 0 || ld < B) {
        set_error("filter_smooth: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return SSMQ_OK;
    if (T == 0) {
        if (d_status) SSMQ_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * ld, stream()));
        SSMQ_HIP(hipStreamSynchronize(stream()));
        return SSMQ_OK;
    }
    const int D = h_dyn->D;
    DevBuf pm, pP, pC;
    if ((rc = pm.alloc(sizeof(double) * (size_t)T * D * ld)) || (rc = pP.alloc(sizeof(double) * (size_t)T * D * D * ld)) ||
        (rc = pC.alloc(sizeof(double) * (size_t)T * D * D * ld)))
        return rc;
    // a captured launch loop points into pm / pP / pC, which are released when this call returns - on every path
    GraphDropGuard drop_on_exit;
    rc = filter_forward_impl(h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_P0, GQG, R, d_fm, d_fP, d_status,
                             nullptr, 0.0, pm.d(), pP.d(), pC.d());
    if (rc) return rc;
    rc = launch_rts_backward(D, B, ld, T, d_fm, d_fP, pm.d(), pP.d(), pC.d(), d_sm, d_sP, d_status, stream(), D);
    if (rc) {
        hipStreamSynchronize(stream());
        return rc;
    }
    SSMQ_HIP(hipStreamSynchronize(stream()));
    return SSMQ_OK;
}

extern "C" int ssmq_student_filter_forward_dev(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn,
                                               ssmq_transform *h_obs, const ssmq_integrand *f_obs, int64_t B,
                                               int64_t ld, int T, const double *d_y, const double *d_m0,
                                               const double *d_S0, const double *GqG, const double *r_smat,
                                               const double *scale, double dof, double *d_fm, double *d_fP,
                                               int32_t *d_status) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!scale || !(dof > 0.0)) {
        set_error("student_filter_forward: scale[T] and dof > 0 are required");
        return SSMQ_E_ARG;
    }
    return filter_forward_impl(h_dyn, f_dyn, h_obs, f_obs, B, ld, T, d_y, d_m0, d_S0, GqG, r_smat, d_fm, d_fP, d_status,
                               scale, dof);
}

namespace ssmq {
int metrics_values_per_step(int D);
int metrics_chunks(int64_t B);
int launch_metrics(int phase, int D, int64_t B, int64_t ld, int T, const double *x, const double *fm, const double *fP,
                   const int32_t *status, const double *mse, double *partial, double *out, hipStream_t s);
}

namespace ssmq {
int launch_metrics_indef(int phase, int D, int64_t B, int64_t ld, int T, const double *x, const double *fm, const double *fP,
                         const int32_t *status, const double *mse, double *partial, double *out, hipStream_t s);
}

static int metrics_impl(int phase, int D, int64_t B, int64_t ld, int T, const double *d_x, const double *d_fm,
                        const double *d_fP, const int32_t *d_status, const double *mse, double *sums) {
    if (D < 1 || D > SSMQ_MAX_DIM || B < 0 || ld < B || T < 0 || !d_x || !d_fm || !d_fP || !sums || (phase == 2 && !mse)) {
        set_error("error_sums: bad argument");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    const int NV = phase == 1 ? metrics_values_per_step(D) : 2;     // values per step handed back
    const int NI = phase == 1 ? NV : 3;                             // ... and reduced on the device
    if (T == 0) return SSMQ_OK;
    if (B == 0) {
        memset(sums, 0, sizeof(double) * (size_t)T * NV);
        return SSMQ_OK;
    }
    hipStream_t s = stream();
    DevBuf partial, out, dm;
    if ((rc = partial.alloc(sizeof(double) * (size_t)T * metrics_chunks(B) * NI)) ||
        (rc = out.alloc(sizeof(double) * (size_t)T * NI)) || (rc = dm.alloc(sizeof(double) * (size_t)T * D * D)))
        return rc;
    if (phase == 2) SSMQ_HIP(hipMemcpyAsync(dm.p, mse, sizeof(double) * (size_t)T * D * D, hipMemcpyHostToDevice, s));
    rc = launch_metrics(phase, D, B, ld, T, d_x, d_fm, d_fP, d_status, dm.d(), partial.d(), out.d(), s);
    if (rc) {
        hipStreamSynchronize(s);
        return rc;
    }
    std::vector<double> h((size_t)T * NI);
    SSMQ_HIP(hipMemcpyAsync(h.data(), out.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    // entries whose covariance is not positive definite were left out by the streaming kernels (they factor P): the
    // reference's formulas do not need a positive-definite P (utils.py:143-148, 426-432) - a second pass adds them
    const int i_ok = D + 2 + D * D, i_cnt = phase == 1 ? i_ok + 1 : 1, i_sum = phase == 1 ? D + 1 : 0;
    bool left_out = false;
    for (int t = 0; t < T && !left_out; ++t)
        left_out = phase == 1 ? h[(size_t)t * NI + i_ok] > h[(size_t)t * NI + i_cnt] : h[(size_t)t * NI + 2] > 0.0;
    if (left_out) {
        rc = launch_metrics_indef(phase, D, B, ld, T, d_x, d_fm, d_fP, d_status, dm.d(), partial.d(), out.d(), s);
        std::vector<double> extra((size_t)T * 2);
        if (!rc) rc = hip_fail(hipMemcpyAsync(extra.data(), out.p, sizeof(double) * extra.size(), hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
        hipError_t e = hipStreamSynchronize(s);
        if (rc) return rc;
        SSMQ_HIP(e);
        for (int t = 0; t < T; ++t) {
            h[(size_t)t * NI + i_sum] += extra[(size_t)t * 2];
            h[(size_t)t * NI + i_cnt] += extra[(size_t)t * 2 + 1];
        }
    }
    for (int t = 0; t < T; ++t)
        for (int v = 0; v < NV; ++v) sums[(size_t)t * NV + v] = h[(size_t)t * NI + v];
    return SSMQ_OK;
}

extern "C" int ssmq_error_sums_width(int D) { return D >= 1 && D <= SSMQ_MAX_DIM ? metrics_values_per_step(D) : SSMQ_E_ARG; }

extern "C" int ssmq_error_sums_dev(int D, int64_t B, int64_t ld, int T, const double *d_x, const double *d_fm,
                                   const double *d_fP, const int32_t *d_status, double *sums) {
    return metrics_impl(1, D, B, ld, T, d_x, d_fm, d_fP, d_status, nullptr, sums);
}

extern "C" int ssmq_lcr_sums_dev(int D, int64_t B, int64_t ld, int T, const double *d_x, const double *d_fm,
                                 const double *d_fP, const int32_t *d_status, const double *mse, double *sums) {
    return metrics_impl(2, D, B, ld, T, d_x, d_fm, d_fP, d_status, mse, sums);
}

namespace ssmq {
size_t gp_weights_wide_ws_bytes(int D, int N, int64_t P);
int gp_weights_wide_consts(int D, int E, int N, const double *d_xi, const double *d_par, int P, double jitter,
                           double *d_consts, int32_t *d_status, void *ws, size_t ws_bytes);
int launch_gauss_logpdf(int Y, int64_t B, int64_t ld, const double *y, const double *y_mean, const double *P_y,
                        double *out, hipStream_t s, const int32_t *merge = nullptr, int32_t *merge_out = nullptr);
// the two-launch route of the theta-batched step (ssmq_weights.hip: k_theta_weights, ssmq_apply_wide.hip: k_theta_chain)
bool gp_theta_weights_fits(int D0, int N0, int D1, int N1);
int gp_theta_weights_pair(const int D[2], const int E[2], const int N[2], const double *const d_xi[2],
                          const double *const d_par[2], int P, double jitter, double *const d_consts[2],
                          int32_t *const d_status[2], const int32_t *d_count = nullptr);
bool theta_chain_supported(int Din, int D, int Y, int Nd, int No);
// ... and for the small systems one lane per item, everything in registers (ssmq_theta_item.hip)
bool theta_item_supported(int Din, int D, int Y, int Nd, int No);
int launch_theta_item(int Din, int D, int Y, int Nd, int No, const ssmq_integrand *f_dyn, const ssmq_integrand *f_obs, int emv_dyn,
                      int emv_obs, const double *xid, const double *xio, const double *pard, const double *paro, const double *mean,
                      const double *cov, int64_t bs_mean, int64_t bs_cov, const double *ysoa, const double *time, int time_stride,
                      const double *gq, const double *rr, double jitter, double *m_fi, double *P_fi, double *ll, int32_t *st_all,
                      int64_t ld, int64_t bound, const int32_t *d_count, hipStream_t s);
hipError_t launch_theta_chain(const WideArgs &dyn, const WideArgs &obs, const UpdArgs &upd, const double *y, double *loglik,
                              const int32_t *merge, int32_t *merge_out, int64_t B, hipStream_t s, const int32_t *d_count = nullptr);
}



namespace {
struct ThetaGraph {
    std::vector<uint64_t> key;
    hipGraph_t graph;
    hipGraphExec_t exec;
};
std::vector<ThetaGraph> &theta_graphs_of_ctx() {
    Ctx &c = ssmq::ctx();
    if (!c.theta_graphs) c.theta_graphs = new std::vector<ThetaGraph>;
    return *(std::vector<ThetaGraph> *)c.theta_graphs;
}
#define g_theta_graphs (theta_graphs_of_ctx())
void drop_theta_graphs() {
    for (auto &g : g_theta_graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    g_theta_graphs.clear();
}
}  // namespace
namespace ssmq {
void drop_theta_step_graphs() { drop_theta_graphs(); }
}

// One filter step per parameter item: weights(theta_dyn) -> dyn transform -> + GQG -> weights(theta_obs) -> obs transform
// -> + R -> measurement update and log N(y | y_mean, P_y).  Everything between the host arrays stays on the device.
static int gp_theta_step_impl(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                              const ssmq_integrand *f_obs, int64_t P, const double *par_dyn, const double *par_obs,
                              double jitter, const double *mean, const double *cov, int shared_state,
                              const double *y, int shared_y, double time, const double *times, const double *GQG, const double *R,
                              double *post_mean, double *post_cov, double *loglik, int32_t *status);

extern "C" int ssmq_gp_theta_step(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                  const ssmq_integrand *f_obs, int64_t P, const double *par_dyn, const double *par_obs,
                                  double jitter, const double *mean, const double *cov, int shared_state,
                                  const double *y, int shared_y, double time, const double *GQG, const double *R,
                                  double *post_mean, double *post_cov, double *loglik, int32_t *status) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    return gp_theta_step_impl(h_dyn, f_dyn, h_obs, f_obs, P, par_dyn, par_obs, jitter, mean, cov, shared_state, y, shared_y, time,
                              nullptr, GQG, R, post_mean, post_cov, loglik, status);
}

// the same with a time of its own per item (times [P]): items of different time steps in one call - the batched marginalised
// filter lets every trajectory run ahead at its own pace (csrc/ssmq_marginal.hip)
extern "C" int ssmq_gp_theta_step_times(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                        const ssmq_integrand *f_obs, int64_t P, const double *par_dyn, const double *par_obs,
                                        double jitter, const double *mean, const double *cov, int shared_state,
                                        const double *y, int shared_y, const double *times, const double *GQG, const double *R,
                                        double *post_mean, double *post_cov, double *loglik, int32_t *status) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!times) {
        set_error("gp_theta_step_times: times is NULL");
        return SSMQ_E_ARG;
    }
    return gp_theta_step_impl(h_dyn, f_dyn, h_obs, f_obs, P, par_dyn, par_obs, jitter, mean, cov, shared_state, y, shared_y, 0.0,
                              times, GQG, R, post_mean, post_cov, loglik, status);
}

static int gp_theta_step_impl(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                              const ssmq_integrand *f_obs, int64_t P, const double *par_dyn, const double *par_obs,
                              double jitter, const double *mean, const double *cov, int shared_state,
                              const double *y, int shared_y, double time, const double *times, const double *GQG, const double *R,
                              double *post_mean, double *post_cov, double *loglik, int32_t *status) {
    if (!h_dyn || !h_obs || !f_dyn || !f_obs || P < 0 || P > 0x7fffffff || !par_dyn || !par_obs || !mean || !cov || !y ||
        !post_mean || !post_cov || !loglik) {
        set_error("gp_theta_step: bad argument");
        return SSMQ_E_ARG;
    }
    // Din > D: dynamics that take their noise as an argument - the caller passes the AUGMENTED moments [m; q_mean],
    // blockdiag(P, Q) (ssinf.py:1174-1176) and GQG = NULL; the measurement model is additive (the reference builds its
    // measurement transform on dim_state inputs, ssinf.py:1288, so it cannot run a non-additive one either)
    const int Din = h_dyn->D, D = h_dyn->E, Y = h_obs->E, Nd = h_dyn->N, No = h_obs->N;
    if (Din < D || h_obs->D != D || h_dyn->form != SSMQ_FORM_BQ || h_obs->form != SSMQ_FORM_BQ ||
        h_dyn->tp_nu > 0.0 || h_obs->tp_nu > 0.0) {
        set_error("gp_theta_step: needs GP-quadrature transforms (D + dq) -> D and D -> Y");
        return SSMQ_E_ARG;
    }
    FInfo fid, fio;
    int rc = check_integrand(h_dyn, f_dyn, &fid);
    if (rc || (rc = check_integrand(h_obs, f_obs, &fio))) return rc;
    if (wide_lds_bytes(Din, D, Nd) > 160 * 1024 - 64 || wide_lds_bytes(D, Y, No) > 160 * 1024 - 64) {
        set_error("gp_theta_step: shape too large for the LDS-resident generic kernel");
        return SSMQ_E_UNSUPPORTED;
    }
    if ((rc = ensure_device())) return rc;
    if (P == 0) return SSMQ_OK;
    hipStream_t s = stream();
    const int64_t ld = (P + 63) / 64 * 64;
    const WideLayout cld = wide_layout(Din, D, Nd, SSMQ_FORM_BQ), clo = wide_layout(D, Y, No, SSMQ_FORM_BQ);
    const int64_t ns = shared_state ? 1 : P;
    // ---- one device arena and two pinned host blocks, kept between calls (the marginalised filter calls this once per
    // BFGS iteration with a handful of items: forty allocations and three synchronisations per call were 310 us) -----------
    // input block, same layout on host and device:  xi_dyn | xi_obs | par_dyn | par_obs | mean | cov | y (planes) | GQG | R | t
    const size_t n_in = (size_t)Din * Nd + (size_t)D * No + (size_t)P * (1 + Din) + (size_t)P * (1 + D) +
                        (size_t)ns * (Din + Din * Din) + (size_t)ld * Y + (size_t)D * D + (size_t)Y * Y + (times ? (size_t)ld : 1);
    // work planes: m_pr D | P_pr D*D | C_xx D*D | y_mean Y | P_y Y*Y | P_yx Y*D, then the output block m_fi D | P_fi D*D |
    // loglik 1 | merged status (int32), then the five partial status vectors
    const size_t n_mid = (size_t)D + (size_t)D * D + (size_t)D * Din + Y + (size_t)Y * Y + (size_t)Y * D;
    const size_t n_out = (size_t)D + (size_t)D * D + 1;
    const size_t out_bytes = sizeof(double) * n_out * ld + sizeof(int32_t) * ld;
    const size_t ws_d = gp_weights_wide_ws_bytes(Din, Nd, P), ws_o = gp_weights_wide_ws_bytes(D, No, P);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t off_in = 0, off_cd = al(sizeof(double) * n_in), off_co = off_cd + al(sizeof(double) * P * cld.total),
                 off_mid = off_co + al(sizeof(double) * P * clo.total), off_out = off_mid + al(sizeof(double) * n_mid * ld),
                 off_st = off_out + al(out_bytes), off_ws = off_st + al(sizeof(int32_t) * 5 * ld),
                 total = off_ws + std::max(ws_d, ws_o);
    if ((rc = g_stage.reserve(total, sizeof(double) * n_in, out_bytes))) return rc;
    char *dev = (char *)g_stage.dev;
    double *hin = (double *)g_stage.hin;
    {
        double *h = hin;
        auto put = [&](const double *src, size_t n) {
            memcpy(h, src, sizeof(double) * n);
            h += n;
        };
        put(h_dyn->xi.data(), (size_t)Din * Nd);
        put(h_obs->xi.data(), (size_t)D * No);
        put(par_dyn, (size_t)P * (1 + Din));
        put(par_obs, (size_t)P * (1 + D));
        put(mean, (size_t)ns * Din);
        put(cov, (size_t)ns * Din * Din);
        for (int k = 0; k < Y; ++k) {                      // measurements straight into the plane layout
            for (int64_t i = 0; i < P; ++i) h[(size_t)k * ld + i] = y[(shared_y ? 0 : (size_t)i * Y) + k];
            for (int64_t i = P; i < ld; ++i) h[(size_t)k * ld + i] = 0.0;
        }
        h += (size_t)ld * Y;
        if (GQG) put(GQG, (size_t)D * D); else { memset(h, 0, sizeof(double) * D * D); h += (size_t)D * D; }
        if (R) put(R, (size_t)Y * Y); else { memset(h, 0, sizeof(double) * Y * Y); h += (size_t)Y * Y; }
        if (times) {
            for (int64_t i = 0; i < ld; ++i) *h++ = i < P ? times[i] : 0.0;
        } else {
            *h++ = time;
        }
    }
    double *in = (double *)(dev + off_in);
    double *xid = in; in += (size_t)Din * Nd;
    double *xio = in; in += (size_t)D * No;
    double *pard = in; in += (size_t)P * (1 + Din);
    double *paro = in; in += (size_t)P * (1 + D);
    double *min_ = in; in += (size_t)ns * Din;
    double *cin = in; in += (size_t)ns * Din * Din;
    double *ysoa = in; in += (size_t)ld * Y;
    double *gq = in; in += (size_t)D * D;
    double *rr = in; in += (size_t)Y * Y;
    double *tt = in;
    double *cd = (double *)(dev + off_cd), *co = (double *)(dev + off_co);
    int32_t *st_wd = (int32_t *)(dev + off_st), *st_wo = st_wd + ld, *st_td = st_wo + ld, *st_to = st_td + ld, *st_up = st_to + ld;
    // everything between the two pinned blocks as ONE unit.  The launch-per-stage route (ten launches + two copies) runs
    // eagerly the first time a (shape, item count) is seen, is captured into a hipGraph the second time and replayed from
    // then on: the marginalised filter calls this hundreds of times with two item counts (gradient: param_dim + 1,
    // marginalisation: 2 param_dim); 89 -> 76 us per call at P = 7
    // two launches - k_theta_weights (both transforms' weights, LDS-resident) and k_theta_chain (transform -> transform ->
    // update -> log-likelihood by the wave that owns the item) - where the point sets fit; else the launch per stage
    // ... or ONE launch with a lane per item for the small systems (k_theta_item, round 5)
    const bool item_route = theta_item_supported(Din, D, Y, Nd, No);
    const bool two_launch = item_route || (theta_chain_supported(Din, D, Y, Nd, No) && gp_theta_weights_fits(Din, Nd, D, No));
    auto enqueue = [&]() -> int {
    int rc;
    SSMQ_HIP(hipMemcpyAsync(dev + off_in, hin, sizeof(double) * n_in, hipMemcpyHostToDevice, s));
    if (item_route) {
        double *wo_ = (double *)(dev + off_out);
        double *m_fi = wo_, *P_fi = wo_ + ld * D, *ll = P_fi + ld * D * D;
        if ((rc = launch_theta_item(Din, D, Y, Nd, No, f_dyn, f_obs, h_dyn->emv_mode, h_obs->emv_mode, xid, xio, pard, paro, min_, cin,
                                    shared_state ? 0 : Din, shared_state ? 0 : (int64_t)Din * Din, ysoa, tt, times ? 1 : 0, gq, rr, jitter,
                                    m_fi, P_fi, ll, (int32_t *)(ll + ld), ld, P, nullptr, s)))
            return rc;
        SSMQ_HIP(hipMemcpyAsync(g_stage.hout, dev + off_out, out_bytes, hipMemcpyDeviceToHost, s));
        return SSMQ_OK;
    }
    if (!two_launch) {
        SSMQ_HIP(hipMemsetAsync(st_wd, 0, sizeof(int32_t) * 5 * ld, s));
        if ((rc = gp_weights_wide_consts(Din, D, Nd, xid, pard, (int)P, jitter, cd, st_wd, dev + off_ws, ws_d))) return rc;
        if ((rc = gp_weights_wide_consts(D, Y, No, xio, paro, (int)P, jitter, co, st_wo, dev + off_ws, ws_o))) return rc;
    } else {
        // every flag vector is written by the two kernels themselves (padding items beyond P are never read back)
        const int dd[2] = {Din, D}, ee[2] = {D, Y}, nn[2] = {Nd, No};
        const double *const xx[2] = {xid, xio}, *const pp[2] = {pard, paro};
        double *const cc[2] = {cd, co};
        int32_t *const ss[2] = {st_wd, st_wo};
        if ((rc = gp_theta_weights_pair(dd, ee, nn, xx, pp, (int)P, jitter, cc, ss))) return rc;
    }
    double *w = (double *)(dev + off_mid);
    double *m_pr = w; w += ld * D;
    double *P_pr = w; w += ld * D * D;
    double *C_xx = w; w += ld * D * Din;
    double *y_mean = w; w += ld * Y;
    double *P_y = w; w += ld * Y * Y;
    double *P_yx = w;
    w = (double *)(dev + off_out);
    double *m_fi = w; w += ld * D;
    double *P_fi = w; w += ld * D * D;
    double *ll = w; w += ld;
    int32_t *st_all = (int32_t *)w;
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.D = Din; a.E = D; a.N = Nd; a.form = SSMQ_FORM_BQ; a.mode = SSMQ_WIDE_FULL; a.fid = f_dyn->id; a.time_stride = times ? 1 : 0;
    a.emv_mode = h_dyn->emv_mode; a.tp_nu = 0.0; a.cov_scale = a.ccov_scale = 1.0;
    a.consts = cd; a.consts_stride = cld.total; a.cov_add = gq;
    a.mean = min_; a.cov = cin; a.time = tt; a.es_in = 1; a.bs_mean = shared_state ? 0 : Din;
    a.bs_cov = shared_state ? 0 : (int64_t)Din * Din;
    a.mean_f = m_pr; a.cov_f = P_pr; a.cov_fx = C_xx; a.es_out = ld; a.bs_mf = a.bs_cf = a.bs_cfx = 1; a.status = st_td;
    fill_fpar(f_dyn, &a.fp);
    const WideArgs a_dyn = a;
    if (!two_launch && (rc = hip_fail(launch_apply_wide(a, P, s), "k_apply_wide(theta, dyn)"))) return rc;
    a.D = D; a.E = Y; a.N = No; a.fid = f_obs->id; a.emv_mode = h_obs->emv_mode; a.consts = co; a.consts_stride = clo.total;
    a.cov_add = rr; a.mean = m_pr; a.cov = P_pr; a.es_in = ld; a.bs_mean = a.bs_cov = 1;
    a.mean_f = y_mean; a.cov_f = P_y; a.cov_fx = P_yx; a.status = st_to;
    fill_fpar(f_obs, &a.fp);
    if (two_launch) {
        const UpdArgs u{m_pr, P_pr, y_mean, P_y, P_yx, ysoa, m_fi, P_fi, st_up, nullptr, nullptr, P, ld, 0, D, Y, 0.0, nullptr, D};
        if ((rc = hip_fail(launch_theta_chain(a_dyn, a, u, ysoa, ll, st_wd, st_all, P, s), "k_theta_chain"))) return rc;
    } else {
    if ((rc = hip_fail(launch_apply_wide(a, P, s), "k_apply_wide(theta, obs)"))) return rc;
    if ((rc = launch_kalman_update(D, Y, P, ld, m_pr, P_pr, y_mean, P_y, P_yx, ysoa, m_fi, P_fi, st_up, s))) return rc;
    // log-likelihood and the merged status flags in one launch (the five partial vectors are contiguous, pitch ld)
    if ((rc = launch_gauss_logpdf(Y, P, ld, ysoa, y_mean, P_y, ll, s, st_wd, st_all))) return rc;
    }
    SSMQ_HIP(hipMemcpyAsync(g_stage.hout, dev + off_out, out_bytes, hipMemcpyDeviceToHost, s));
    return SSMQ_OK;
    };
    {
        std::vector<uint64_t> key = {(uint64_t)(uintptr_t)dev, (uint64_t)(uintptr_t)hin, (uint64_t)(uintptr_t)g_stage.hout, (uint64_t)P,
                                     (uint64_t)Din, (uint64_t)D, (uint64_t)Y, (uint64_t)Nd, (uint64_t)No, (uint64_t)shared_state,
                                     (uint64_t)h_dyn->emv_mode, (uint64_t)h_obs->emv_mode, (uint64_t)total, (uint64_t)two_launch, (uint64_t)(times ? 1 : 0)};
        uint64_t jb;
        memcpy(&jb, &jitter, sizeof(jb));
        key.push_back(jb);
        for (const ssmq_integrand *f : {f_dyn, f_obs}) {
            // what the kernels read of the descriptor: id, counts and the USED parameter / index slots (not the padding or the
            // unused tail, which a caller may leave uninitialised: every call would then look like a new graph)
            uint64_t hsh = 1469598103934665603ull;
            auto mix = [&](const void *p, size_t n) {
                const unsigned char *pb = (const unsigned char *)p;
                for (size_t i = 0; i < n; ++i) hsh = (hsh ^ pb[i]) * 1099511628211ull;
            };
            const int np_ = f->n_par < 0 ? 0 : (f->n_par > SSMQ_MAX_FPAR ? SSMQ_MAX_FPAR : f->n_par);
            const int ni_ = f->n_idx < 0 ? 0 : (f->n_idx > SSMQ_MAX_FIDX ? SSMQ_MAX_FIDX : f->n_idx);
            mix(&f->id, sizeof(f->id)); mix(&f->n_par, sizeof(f->n_par)); mix(&f->n_idx, sizeof(f->n_idx));
            mix(f->par, sizeof(f->par[0]) * np_); mix(f->idx, sizeof(f->idx[0]) * ni_);
            key.push_back(hsh);
        }
        ThetaGraph *tg = nullptr;
        for (auto &g : g_theta_graphs)
            if (g.key == key) tg = &g;
        // two launches + two copies: replay measured no faster (52 us either way).  Per-item times = the batched marginalised
        // filter, whose item count changes from round to round: every new count would be captured, instantiated and evict an
        // older entry of the six-slot cache without ever being replayed.
        if (two_launch || times || ssmq::sw("SSMQ_NO_THETA_GRAPH")) {
            if ((rc = enqueue())) return rc;
        } else if (!tg) {
            if (g_theta_graphs.size() >= 6) {       // the oldest entry goes, not all of them (a filter alternates between two item counts)
                if (g_theta_graphs.front().exec) hipGraphExecDestroy(g_theta_graphs.front().exec);
                if (g_theta_graphs.front().graph) hipGraphDestroy(g_theta_graphs.front().graph);
                g_theta_graphs.erase(g_theta_graphs.begin());
            }
            g_theta_graphs.push_back(ThetaGraph{key, nullptr, nullptr});
            if ((rc = enqueue())) return rc;
        } else {
            if (!tg->exec) {
                SSMQ_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                rc = enqueue();
                hipGraph_t g = nullptr;
                hipError_t ce = hipStreamEndCapture(s, &g);
                if (rc || ce != hipSuccess) {
                    if (g) hipGraphDestroy(g);
                    return rc ? rc : hip_fail(ce, "hipStreamEndCapture(theta step)");
                }
                const hipError_t ie = hipGraphInstantiate(&tg->exec, g, nullptr, nullptr, 0);
                if (ie != hipSuccess) {
                    hipGraphDestroy(g);
                    tg->exec = nullptr;
                    return hip_fail(ie, "hipGraphInstantiate(theta step)");
                }
                tg->graph = g;
            }
            SSMQ_HIP(hipGraphLaunch(tg->exec, s));
        }
    }
    SSMQ_HIP(hipStreamSynchronize(s));
    // planes -> the caller's item-major arrays
    const double *ho = (const double *)g_stage.hout;
    const int32_t *hst = (const int32_t *)(ho + n_out * ld);
    for (int e = 0; e < D; ++e)
        for (int64_t i = 0; i < P; ++i) post_mean[(size_t)i * D + e] = ho[(size_t)e * ld + i];
    const double *hp = ho + (size_t)D * ld;
    for (int e = 0; e < D * D; ++e)
        for (int64_t i = 0; i < P; ++i) post_cov[(size_t)i * D * D + e] = hp[(size_t)e * ld + i];
    memcpy(loglik, hp + (size_t)D * D * ld, sizeof(double) * P);
    int first = 0;
    for (int64_t i = 0; i < P; ++i) {
        if (status) status[i] = hst[i];
        if (hst[i] && !first) first = (int)std::min<int64_t>(i + 1, 0x7fffffff);
    }
    return first;
}

// ---- the theta-batched step with its items ALREADY on the device and their number in device memory --------------------------
// (the device-resident rounds of the batched marginalised filter, ssmq_marginal.hip: no host copy and no synchronisation per
// round; the kernels are the two of gp_theta_step_impl's two-launch route, launched on an upper bound of the item count.)
namespace ssmq {
bool theta_dev_supported(const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, const ssmq_transform *h_obs, const ssmq_integrand *f_obs) {
    if (!h_dyn || !h_obs || !f_dyn || !f_obs) return false;
    const int Din = h_dyn->D, D = h_dyn->E, Y = h_obs->E, Nd = h_dyn->N, No = h_obs->N;
    if (Din < D || h_obs->D != D || h_dyn->form != SSMQ_FORM_BQ || h_obs->form != SSMQ_FORM_BQ || h_dyn->tp_nu > 0.0 || h_obs->tp_nu > 0.0)
        return false;
    FInfo fi;
    if (check_integrand(h_dyn, f_dyn, &fi) || check_integrand(h_obs, f_obs, &fi))
        return false;
    return theta_item_supported(Din, D, Y, Nd, No) || (theta_chain_supported(Din, D, Y, Nd, No) && gp_theta_weights_fits(Din, Nd, D, No));
}

size_t theta_dev_bytes(const ssmq_transform *h_dyn, const ssmq_transform *h_obs, int64_t cap) {
    ThetaDev t;
    return theta_dev_carve(t, h_dyn, h_obs, cap, nullptr);
}

// lays the arena out (base may be null: size only); returns its size in bytes
size_t theta_dev_carve(ThetaDev &t, const ssmq_transform *h_dyn, const ssmq_transform *h_obs, int64_t cap, void *base) {
    const int Din = h_dyn->D, D = h_dyn->E, Y = h_obs->E, Nd = h_dyn->N, No = h_obs->N;
    const int64_t ld = (cap + 63) / 64 * 64;
    const WideLayout cld = wide_layout(Din, D, Nd, SSMQ_FORM_BQ), clo = wide_layout(D, Y, No, SSMQ_FORM_BQ);
    t.Din = Din; t.D = D; t.Y = Y; t.Nd = Nd; t.No = No; t.cap = cap; t.ld = ld;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? (char *)base + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return p;
    };
    t.xid = (double *)take(sizeof(double) * Din * Nd);
    t.xio = (double *)take(sizeof(double) * D * No);
    t.gq = (double *)take(sizeof(double) * D * D);
    t.rr = (double *)take(sizeof(double) * Y * Y);
    t.pard = (double *)take(sizeof(double) * cap * (1 + Din));
    t.paro = (double *)take(sizeof(double) * cap * (1 + D));
    t.mean = (double *)take(sizeof(double) * cap * Din);
    t.cov = (double *)take(sizeof(double) * cap * Din * Din);
    t.ysoa = (double *)take(sizeof(double) * ld * Y);
    t.tt = (double *)take(sizeof(double) * ld);
    t.cd = (double *)take(sizeof(double) * cap * cld.total);
    t.co = (double *)take(sizeof(double) * cap * clo.total);
    t.mid = (double *)take(sizeof(double) * ld * ((size_t)D + (size_t)D * D + (size_t)D * Din + Y + (size_t)Y * Y + (size_t)Y * D));
    t.m_fi = (double *)take(sizeof(double) * ld * D);
    t.P_fi = (double *)take(sizeof(double) * ld * D * D);
    t.ll = (double *)take(sizeof(double) * ld);
    t.st_all = (int32_t *)take(sizeof(int32_t) * ld);
    t.st5 = (int32_t *)take(sizeof(int32_t) * 5 * ld);
    return off;
}

// uploads what does not change between rounds: the unit points of both transforms, G Q G' (zeros for dynamics that take their
// noise as an argument) and R
int theta_dev_upload_static(const ThetaDev &t, const ssmq_transform *h_dyn, const ssmq_transform *h_obs, const double *GQG, const double *R,
                            hipStream_t s) {
    SSMQ_HIP(hipMemcpyAsync(t.xid, h_dyn->xi.data(), sizeof(double) * t.Din * t.Nd, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(t.xio, h_obs->xi.data(), sizeof(double) * t.D * t.No, hipMemcpyHostToDevice, s));
    if (GQG) SSMQ_HIP(hipMemcpyAsync(t.gq, GQG, sizeof(double) * t.D * t.D, hipMemcpyHostToDevice, s));
    else SSMQ_HIP(hipMemsetAsync(t.gq, 0, sizeof(double) * t.D * t.D, s));
    if (R) SSMQ_HIP(hipMemcpyAsync(t.rr, R, sizeof(double) * t.Y * t.Y, hipMemcpyHostToDevice, s));
    else SSMQ_HIP(hipMemsetAsync(t.rr, 0, sizeof(double) * t.Y * t.Y, s));
    SSMQ_HIP(hipStreamSynchronize(s));        // (the host vectors may go away)
    return SSMQ_OK;
}

// weights of both transforms, then transform -> transform -> update -> log-likelihood per item: two launches covering
// `bound` items (>= the count the kernels read from *d_count; 0 < bound <= t.cap)
int theta_dev_enqueue(const ThetaDev &t, const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, const ssmq_transform *h_obs,
                      const ssmq_integrand *f_obs, double jitter, int64_t bound, const int32_t *d_count, hipStream_t s) {
    if (bound < 1 || bound > t.cap) {
        set_error("theta_dev_enqueue: bad bound");
        return SSMQ_E_ARG;
    }
    const int Din = t.Din, D = t.D, Y = t.Y, Nd = t.Nd, No = t.No;
    const int64_t ld = t.ld;
    if (theta_item_supported(Din, D, Y, Nd, No))
        return launch_theta_item(Din, D, Y, Nd, No, f_dyn, f_obs, h_dyn->emv_mode, h_obs->emv_mode, t.xid, t.xio, t.pard, t.paro, t.mean, t.cov,
                                 Din, (int64_t)Din * Din, t.ysoa, t.tt, 1, t.gq, t.rr, jitter, t.m_fi, t.P_fi, t.ll, t.st_all, ld, bound,
                                 d_count, s);
    const WideLayout cld = wide_layout(Din, D, Nd, SSMQ_FORM_BQ), clo = wide_layout(D, Y, No, SSMQ_FORM_BQ);
    int32_t *st_wd = t.st5, *st_wo = st_wd + ld, *st_td = st_wo + ld, *st_to = st_td + ld, *st_up = st_to + ld;
    int rc;
    {
        const int dd[2] = {Din, D}, ee[2] = {D, Y}, nn[2] = {Nd, No};
        const double *const xx[2] = {t.xid, t.xio}, *const pp[2] = {t.pard, t.paro};
        double *const cc[2] = {t.cd, t.co};
        int32_t *const ss[2] = {st_wd, st_wo};
        if ((rc = gp_theta_weights_pair(dd, ee, nn, xx, pp, (int)bound, jitter, cc, ss, d_count))) return rc;
    }
    double *w = t.mid;
    double *m_pr = w; w += ld * D;
    double *P_pr = w; w += ld * D * D;
    double *C_xx = w; w += ld * D * Din;
    double *y_mean = w; w += ld * Y;
    double *P_y = w; w += ld * Y * Y;
    double *P_yx = w;
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.D = Din; a.E = D; a.N = Nd; a.form = SSMQ_FORM_BQ; a.mode = SSMQ_WIDE_FULL; a.fid = f_dyn->id; a.time_stride = 1;
    a.emv_mode = h_dyn->emv_mode; a.tp_nu = 0.0; a.cov_scale = a.ccov_scale = 1.0;
    a.consts = t.cd; a.consts_stride = cld.total; a.cov_add = t.gq;
    a.mean = t.mean; a.cov = t.cov; a.time = t.tt; a.es_in = 1; a.bs_mean = Din; a.bs_cov = (int64_t)Din * Din;
    a.mean_f = m_pr; a.cov_f = P_pr; a.cov_fx = C_xx; a.es_out = ld; a.bs_mf = a.bs_cf = a.bs_cfx = 1; a.status = st_td;
    fill_fpar(f_dyn, &a.fp);
    const WideArgs a_dyn = a;
    a.D = D; a.E = Y; a.N = No; a.fid = f_obs->id; a.emv_mode = h_obs->emv_mode; a.consts = t.co; a.consts_stride = clo.total;
    a.cov_add = t.rr; a.mean = m_pr; a.cov = P_pr; a.es_in = ld; a.bs_mean = a.bs_cov = 1;
    a.mean_f = y_mean; a.cov_f = P_y; a.cov_fx = P_yx; a.status = st_to;
    fill_fpar(f_obs, &a.fp);
    const UpdArgs u{m_pr, P_pr, y_mean, P_y, P_yx, t.ysoa, t.m_fi, t.P_fi, st_up, nullptr, nullptr, bound, ld, 0, D, Y, 0.0, nullptr, D};
    return hip_fail(launch_theta_chain(a_dyn, a, u, t.ysoa, t.ll, st_wd, t.st_all, bound, s, d_count), "k_theta_chain(device rounds)");
}
}  // namespace ssmq

extern "C" int ssmq_simulate_rv_dev(const ssmq_integrand *f_dyn, const ssmq_integrand *f_obs, int D, int Y, const ssmq_rv *x0,
                                    const ssmq_rv *q, const ssmq_rv *r, const double *G, int dyn_additive, int obs_additive,
                                    int64_t B, int64_t ld, int T, int continuous, double dt, uint64_t seed,
                                    uint64_t traj_offset, double *d_x, double *d_y) {
    const int mode = (f_dyn ? 1 : 0) | (f_obs ? 2 : 0);
    auto rv_ok = [](const ssmq_rv *v, int dim) {
        return v && v->dim == dim && v->chol && v->kind >= SSMQ_RV_GAUSS && v->kind <= SSMQ_RV_MIXTURE &&
               (v->kind != SSMQ_RV_STUDENT || v->dof > 2.0) &&
               (v->kind == SSMQ_RV_MIXTURE ? (v->n_comp >= 1 && v->n_comp <= 8 && v->alpha) : v->n_comp <= 1);
    };
    int dq = (f_dyn && q) ? q->dim : 0, dr = (f_obs && r) ? r->dim : 0;
    if (!mode || D < 1 || D > SSMQ_MAX_DIM || B < 0 || ld < B || T < 0 || !d_x ||
        (f_dyn && (!rv_ok(x0, D) || dq < 1 || dq > SSMQ_MAX_DIM || !rv_ok(q, dq))) ||
        (f_obs && (!d_y || Y < 1 || Y > SSMQ_MAX_DIM || dr < 1 || dr > SSMQ_MAX_DIM || !rv_ok(r, dr))) ||
        (continuous && (!f_dyn || !(dt > 0.0)))) {
        set_error("simulate: bad argument");
        return SSMQ_E_ARG;
    }
    if (!f_obs) Y = 0;
    FInfo fid, fio;
    if (f_dyn) {
        const int in_dyn = D + (dyn_additive ? 0 : dq);
        if (!integrand_info(f_dyn->id, &fid) || fid.dout != D || f_dyn->n_idx != 0 ||
            (!continuous && (fid.din > in_dyn || in_dyn > kMaxIntegrandIn))) {
            set_error("simulate: transition integrand / dimension mismatch (state + noise inputs: at most 16)");
            return SSMQ_E_ARG;
        }
        if (continuous && !has_continuous_dynamics(f_dyn->id)) {
            set_error("simulate: this model has no continuous-time dynamics (dyn_fcn_cont is defined for the reentry-1D, "
                      "reentry-2D and constant-turn-rate-and-speed models only, ssmod.py:429-432, 569-585, 779-780)");
            return SSMQ_E_UNSUPPORTED;
        }
        if (continuous && dq < (f_dyn->id == SSMQ_F_CTRS_DYN ? 1 : 3)) {
            set_error("simulate: the continuous-time dynamics read three noise components");
            return SSMQ_E_ARG;
        }
    }
    if (f_obs) {
        const int in_obs = D + (obs_additive ? 0 : dr);
        if (!integrand_info(f_obs->id, &fio) || (fio.dout ? fio.dout : Y) != Y || (obs_additive && dr != Y) ||
            f_obs->n_idx > SSMQ_MAX_FIDX || f_obs->n_idx < 0 || (f_obs->n_idx == 0 && (fio.din > in_obs || in_obs > kMaxIntegrandIn))) {
            set_error("simulate: measurement integrand / dimension mismatch");
            return SSMQ_E_ARG;
        }
        for (int k = 0; k < f_obs->n_idx; ++k)
            if (f_obs->idx[k] < 0 || f_obs->idx[k] >= in_obs) {
                set_error("simulate: measurement state index out of range");
                return SSMQ_E_ARG;
            }
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || T == 0) return SSMQ_OK;
    SimLaunch h;
    memset(&h, 0, sizeof(h));
    std::vector<double> hc;
    auto put_rv = [&](const ssmq_rv *v, SimRv *out) {
        out->off = (int)hc.size();
        if (!v) {
            out->kind = SSMQ_RV_GAUSS; out->dim = 0; out->ncomp = 1; out->dof = 0.0;
            hc.push_back(1.0);
            return;
        }
        const int nc = v->kind == SSMQ_RV_MIXTURE ? v->n_comp : 1, n = v->dim;
        out->kind = v->kind; out->dim = n; out->ncomp = nc; out->dof = v->dof;
        for (int k = 0; k < nc; ++k) hc.push_back(v->kind == SSMQ_RV_MIXTURE ? v->alpha[k] : 1.0);
        for (int i = 0; i < nc * n; ++i) hc.push_back(v->mean ? v->mean[i] : 0.0);
        for (int i = 0; i < nc * n * n; ++i) hc.push_back(v->chol[i]);
    };
    put_rv(f_dyn ? x0 : nullptr, &h.rv[0]);
    put_rv(f_dyn ? q : nullptr, &h.rv[1]);
    put_rv(f_obs ? r : nullptr, &h.rv[2]);
    h.g_off = (int)hc.size();
    for (int i = 0; i < D * dq; ++i)        // default noise gain eye(D, dq)  (ssmod.py:52)
        hc.push_back(G ? G[i] : ((i / dq) == (i % dq) ? 1.0 : 0.0));
    DevBuf dc;
    if ((rc = dc.alloc(sizeof(double) * std::max<size_t>(hc.size(), 1)))) return rc;
    hipStream_t s = stream();
    SSMQ_HIP(hipMemcpyAsync(dc.p, hc.data(), sizeof(double) * hc.size(), hipMemcpyHostToDevice, s));
    h.mode = mode; h.D = D; h.Y = Y; h.dq = dq; h.dr = dr; h.dyn_additive = dyn_additive; h.obs_additive = obs_additive; h.T = T;
    h.continuous = continuous ? 1 : 0; h.dt = dt; h.B = B; h.ld = ld; h.seed = seed; h.traj_offset = traj_offset;
    h.f_dyn = f_dyn; h.f_obs = f_obs; h.d_consts = dc.d(); h.d_x = d_x; h.d_y = d_y;
    rc = launch_simulate(h, s);
    hipError_t e = hipStreamSynchronize(s);
    if (rc) return rc;
    SSMQ_HIP(e);
    return SSMQ_OK;
}

// the Gaussian case with plain arrays (the round-1 entry point)
extern "C" int ssmq_simulate_dev(const ssmq_integrand *f_dyn, const ssmq_integrand *f_obs, int D, int Y, int dq, int dr,
                                 int dyn_additive, int obs_additive, int64_t B, int64_t ld, int T,
                                 const double *x0_mean, const double *x0_chol, const double *q_mean,
                                 const double *q_chol, const double *G, const double *r_mean, const double *r_chol,
                                 uint64_t seed, uint64_t traj_offset, double *d_x, double *d_y) {
    ssmq_rv x0{SSMQ_RV_GAUSS, D, 1, 0, 0.0, x0_mean, x0_chol, nullptr};
    ssmq_rv q{SSMQ_RV_GAUSS, dq, 1, 0, 0.0, q_mean, q_chol, nullptr};
    ssmq_rv r{SSMQ_RV_GAUSS, dr, 1, 0, 0.0, r_mean, r_chol, nullptr};
    if (f_dyn && (!x0_mean || !x0_chol || !q_chol)) {
        set_error("simulate: bad argument");
        return SSMQ_E_ARG;
    }
    return ssmq_simulate_rv_dev(f_dyn, f_obs, D, Y, &x0, &q, &r, G, dyn_additive, obs_additive, B, ld, T, 0, 0.0, seed,
                                traj_offset, d_x, d_y);
}

extern "C" int ssmq_filter_kernel_name(const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn,
                                       const ssmq_transform *h_obs, const ssmq_integrand *f_obs, char *buf, int len) {
    return ssmq_filter_kernel_name_batch(h_dyn, f_dyn, h_obs, f_obs, 0, buf, len);
}

extern "C" int ssmq_filter_kernel_name_batch(const ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, const ssmq_transform *h_obs,
                                             const ssmq_integrand *f_obs, int64_t B, char *buf, int len) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!h_dyn || !h_obs || !f_dyn || !f_obs || !buf || len <= 0 || B < 0) return SSMQ_E_ARG;
    FInfo fio;
    if (!integrand_info(f_obs->id, &fio)) return SSMQ_E_ARG;
    const char *name = nullptr;
    int rc = ssmq::sw("SSMQ_NO_FUSED") ? 0
                                     : try_launch_fused(h_dyn, f_dyn, h_obs, f_obs, sel_pattern(f_obs, fio.din), B, 0, 0,
                                                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                        nullptr, nullptr, &name, true, nullptr, 0.0, nullptr, nullptr);
    if (rc < 0) return rc;
    snprintf(buf, len, "%s", rc == 1 ? name : "hipGraph of 3 T launches (apply dyn | apply obs | k_kalman_update)");
    return SSMQ_OK;
}
