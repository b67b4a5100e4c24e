// Study-level entry points (round 6): several filters in one launch (ssmq_filter_forward_multi_dev) and the forward pass with host
// arrays as a pipeline of time blocks (ssmq_filter_forward_piped, with its pooled page-locked blocks).  Split from ssmq_api.hip,
// which holds the per-transform and per-filter entry points these build on.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "ssmq_host.h"
#include "ssmq_fused.h"

using namespace ssmq;

int filter_forward_impl(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs, const ssmq_integrand *f_obs, int64_t B,
                        int64_t ld, int T, const double *d_y, const double *d_m0, const double *d_P0, const double *GQG, const double *R,
                        double *d_fm, double *d_fP, int32_t *d_status, const double *sscale, double student_dof, double *d_pm,
                        double *d_pP, double *d_pC);
namespace ssmq {
int sel_pattern(const ssmq_integrand *f, int din);
int try_launch_fused(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho,
                     const ssmq_integrand *fo, int sel_obs, int64_t B, int64_t ld, int T, const double *d_y,
                     const double *d_m0, const double *d_P0, const double *d_gqg, const double *d_rr, double *d_fm,
                     double *d_fP, int32_t *d_status, hipStream_t s, const char **name, bool dry_run,
                     const double *d_sscale, double student_dof, const double *d_ttab_dyn, const double *d_ttab_obs);
}
#define g_stage (ssmq::stage_of_ctx())

// ---- A independent filters as ONE launch (round 6) -------------------------------------------------------------------------
// The reference's studies run several filters over the same data, one after the other (research/bsq/bsq_ungm.py:132-137,
// research/tpq/tpq_base.py:175-192).  A configs[1]-sized pass occupies 157 of the chip's 1 024 SIMDs, so A of them fit side by
// side; what stood in the way was the launch path (round 5: six host threads reached 1.8 x).  Here the calling context's stream
// forks into one branch per job inside a captured graph - each branch is the job's own fused time-loop kernel, so the RESULTS ARE
// THE BITS of ssmq_filter_forward_dev / ssmq_student_filter_forward_dev - and joins again; a repeated call with the same jobs is one
// hipGraphLaunch.  Jobs without a fused kernel run after the graph, one by one, through the ordinary path.
namespace ssmq {
int multi_family_table(int n, const ssmq_transform *const *hd, const ssmq_integrand *const *fd, const ssmq_transform *const *ho,
                       const ssmq_integrand *const *fo, const FusedArgs *args, std::vector<char> *table, int *blocks);
int multi_family_launch(const char *d_table, int blocks, hipStream_t s);
}
namespace {
struct MultiCache {
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> joined;
    hipEvent_t fork = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<uint64_t> key;
    std::vector<char> fused;          // per job of the cached key: 1 = in the graph
    std::vector<char> htable;         // one-kernel route (every job of one model family): the kernel's argument block
    int table_blocks = 0;             // > 0: the cached key runs as ONE launch of the family kernel
    void drop_graph() {
        if (exec) hipGraphExecDestroy(exec);
        if (graph) hipGraphDestroy(graph);
        exec = nullptr;
        graph = nullptr;
        key.clear();
        table_blocks = 0;
    }
    void drop_all() {
        drop_graph();
        for (hipStream_t s : side) hipStreamDestroy(s);
        for (hipEvent_t e : joined) hipEventDestroy(e);
        if (fork) hipEventDestroy(fork);
        side.clear();
        joined.clear();
        fork = nullptr;
        if (ws) hipFree(ws);
        ws = nullptr;
        ws_bytes = 0;
    }
};
MultiCache &multi_of_ctx() {
    Ctx &c = ssmq::ctx();
    if (!c.multi) c.multi = new MultiCache;
    return *(MultiCache *)c.multi;
}
void key_bytes(std::vector<uint64_t> &key, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, b + i, 8); key.push_back(v); }
    if (i < n) { uint64_t v = 0; memcpy(&v, b + i, n - i); key.push_back(v); }
}
}  // namespace
namespace ssmq {
void drop_multi_cache() {
    Ctx &c = ctx();
    if (c.multi) ((MultiCache *)c.multi)->drop_all();
}
}  // namespace ssmq

extern "C" int ssmq_filter_forward_multi_dev(int n_jobs, const ssmq_filter_job *jobs) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs) || n_jobs > 64) {
        set_error("filter_forward_multi: bad job list (0 .. 64 jobs)");
        return SSMQ_E_ARG;
    }
    if (n_jobs == 0) return SSMQ_OK;
    std::vector<const ssmq_transform *> hs;
    for (int i = 0; i < n_jobs; ++i) {
        const ssmq_filter_job &j = jobs[i];
        if (!j.h_dyn || !j.h_obs || !j.f_dyn || !j.f_obs || j.B < 0 || j.ld < j.B || j.T < 0 || !j.d_y || !j.d_m0 || !j.d_P0 || !j.d_fm ||
            !j.d_fP || !j.d_status || (j.scale != nullptr) != (j.dof > 0.0)) {
            set_error("filter_forward_multi: bad argument in job " + std::to_string(i));
            return SSMQ_E_ARG;
        }
        if (j.h_dyn->E != j.h_dyn->D || j.h_obs->D != j.h_dyn->D) {
            set_error("filter_forward_multi: additive-noise filter needs dyn (D -> D) and obs (D -> Y) transforms (job " + std::to_string(i) + ")");
            return SSMQ_E_ARG;
        }
        for (int k = 0; k < i; ++k)
            if (jobs[k].d_fm == j.d_fm || jobs[k].d_fP == j.d_fP || jobs[k].d_status == j.d_status) {
                set_error("filter_forward_multi: jobs " + std::to_string(k) + " and " + std::to_string(i) + " share an output buffer");
                return SSMQ_E_ARG;
            }
        hs.push_back(j.h_dyn);
        hs.push_back(j.h_obs);
    }
    MultiHandleGuard guard(hs);
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = stream();
    MultiCache &mc = multi_of_ctx();
    // ---- the key: everything a captured launch depends on --------------------------------------------------------------------
    std::vector<uint64_t> key = {(uint64_t)n_jobs};
    for (int i = 0; i < n_jobs; ++i) {
        const ssmq_filter_job &j = jobs[i];
        const int D = j.h_dyn->D, Y = j.h_obs->E;
        for (const void *p : {(const void *)j.h_dyn, (const void *)j.h_obs, (const void *)j.d_y, (const void *)j.d_m0, (const void *)j.d_P0,
                              (const void *)j.d_fm, (const void *)j.d_fP, (const void *)j.d_status, (const void *)j.h_dyn->d_small,
                              (const void *)j.h_obs->d_small})
            key.push_back((uint64_t)(uintptr_t)p);
        key.push_back((uint64_t)j.B); key.push_back((uint64_t)j.ld); key.push_back((uint64_t)j.T);
        key_bytes(key, j.f_dyn, sizeof(ssmq_integrand));
        key_bytes(key, j.f_obs, sizeof(ssmq_integrand));
        key.push_back(((uint64_t)j.h_dyn->generation << 32) ^ (uint64_t)j.h_obs->generation);
        key.push_back(((uint64_t)(uint32_t)j.h_dyn->opt_mask << 32) | (uint64_t)(uint32_t)j.h_obs->opt_mask);
        key.push_back((uint64_t)j.h_dyn->emv_mode * 2 + (uint64_t)j.h_obs->emv_mode);
        key_bytes(key, &j.h_dyn->tp_nu, 8); key_bytes(key, &j.h_obs->tp_nu, 8); key_bytes(key, &j.dof, 8);
        if (j.GQG) key_bytes(key, j.GQG, sizeof(double) * D * D); else key.push_back(0);
        if (j.R) key_bytes(key, j.R, sizeof(double) * Y * Y); else key.push_back(0);
        if (j.scale) key_bytes(key, j.scale, sizeof(double) * j.T); else key.push_back(0);
    }
    key.push_back(ssmq::sw("SSMQ_NO_FUSED") ? 1 : 0);
    key.push_back(ssmq::sw("SSMQ_MULTI_NO_GRAPH") ? 1 : 0);
    key.push_back(ssmq::sw("SSMQ_MULTI_NO_FAMILY") ? 1 : 0);
    auto run_rest = [&]() -> int {       // the jobs that are not in the graph, through the ordinary path, one by one
        for (int i = 0; i < n_jobs; ++i) {
            if (mc.fused[i]) continue;
            const ssmq_filter_job &j = jobs[i];
            const int r = filter_forward_impl(j.h_dyn, j.f_dyn, j.h_obs, j.f_obs, j.B, j.ld, j.T, j.d_y, j.d_m0, j.d_P0, j.GQG, j.R, j.d_fm,
                                              j.d_fP, j.d_status, j.scale, j.dof, nullptr, nullptr, nullptr);
            if (r) return r;
        }
        return SSMQ_OK;
    };
    if (mc.table_blocks > 0 && mc.key == key) return multi_family_launch(mc.htable.data(), mc.table_blocks, s);
    if (mc.exec && mc.key == key) {
        SSMQ_HIP(hipGraphLaunch(mc.exec, s));
        return run_rest();
    }
    mc.drop_graph();
    // ---- per-job constants behind one allocation: G Q G', R, scale [T], the two time tables [T] ---------------------------------
    size_t n_dbl = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const int D = jobs[i].h_dyn->D, Y = jobs[i].h_obs->E;
        n_dbl += (size_t)D * D + (size_t)Y * Y + 3 * (size_t)jobs[i].T + 8;
    }
    if (mc.ws_bytes < sizeof(double) * n_dbl) {
        SSMQ_HIP(hipStreamSynchronize(s));
        if (mc.ws) hipFree(mc.ws);
        mc.ws = nullptr;
        mc.ws_bytes = 0;
        SSMQ_HIP(hipMalloc(&mc.ws, sizeof(double) * n_dbl * 2));
        mc.ws_bytes = sizeof(double) * n_dbl * 2;
    }
    std::vector<double> host(n_dbl, 0.0);
    struct JobConsts { const double *gqg, *rr, *svec, *ttd, *tto; };
    std::vector<JobConsts> jc(n_jobs);
    {
        size_t o = 0;
        double *dev = (double *)mc.ws;
        for (int i = 0; i < n_jobs; ++i) {
            const ssmq_filter_job &j = jobs[i];
            const int D = j.h_dyn->D, Y = j.h_obs->E, T = j.T;
            jc[i].gqg = dev + o; if (j.GQG) memcpy(&host[o], j.GQG, sizeof(double) * D * D); o += (size_t)D * D;
            jc[i].rr = dev + o; if (j.R) memcpy(&host[o], j.R, sizeof(double) * Y * Y); o += (size_t)Y * Y;
            jc[i].svec = j.scale ? dev + o : nullptr; if (j.scale) memcpy(&host[o], j.scale, sizeof(double) * T); o += T;
            const bool td = T > 0 && time_table(j.f_dyn->id, T, &host[o]);
            jc[i].ttd = td ? dev + o : nullptr; o += T;
            const bool to = T > 0 && time_table(j.f_obs->id, T, &host[o]);
            jc[i].tto = to ? dev + o : nullptr; o += T;
            o = (o + 7) / 8 * 8;
        }
    }
    SSMQ_HIP(hipMemcpyAsync(mc.ws, host.data(), sizeof(double) * n_dbl, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipStreamSynchronize(s));          // `host` goes out of scope; only when the job list changed
    // ---- side streams and events -------------------------------------------------------------------------------------------
    while ((int)mc.side.size() < n_jobs) {
        hipStream_t ns = nullptr;
        hipEvent_t ne = nullptr;
        SSMQ_HIP(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        mc.side.push_back(ns);
        SSMQ_HIP(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
        mc.joined.push_back(ne);
    }
    if (!mc.fork) SSMQ_HIP(hipEventCreateWithFlags(&mc.fork, hipEventDisableTiming));
    // ---- which jobs have a fused kernel (dry run), then the fork / launch / join sequence, captured unless told otherwise -------
    mc.fused.assign(n_jobs, 0);
    std::vector<int> sel(n_jobs, -1);
    const bool no_fused = ssmq::sw("SSMQ_NO_FUSED") != nullptr;
    for (int i = 0; i < n_jobs && !no_fused; ++i) {
        const ssmq_filter_job &j = jobs[i];
        FInfo fio;
        if (!integrand_info(j.f_obs->id, &fio)) {
            set_error("unknown integrand id");
            return SSMQ_E_ARG;
        }
        sel[i] = sel_pattern(j.f_obs, fio.din);
        if (j.B == 0 || j.T == 0) continue;
        const int r = try_launch_fused(j.h_dyn, j.f_dyn, j.h_obs, j.f_obs, sel[i], 0, j.ld, j.T, j.d_y, j.d_m0, j.d_P0, jc[i].gqg, jc[i].rr, j.d_fm,
                                       j.d_fP, j.d_status, s, nullptr, true, jc[i].svec, j.dof, jc[i].ttd, jc[i].tto);
        if (r < 0) return r;
        mc.fused[i] = r == 1;
    }
    int n_fused = 0;
    for (int i = 0; i < n_jobs; ++i) n_fused += mc.fused[i];
    if (n_fused == n_jobs && !ssmq::sw("SSMQ_MULTI_NO_FAMILY")) {
        // every job a filter of one model family with a common kernel: ONE launch, the jobs' blocks side by side
        std::vector<FusedArgs> fa(n_jobs);
        std::vector<const ssmq_transform *> vhd(n_jobs), vho(n_jobs);
        std::vector<const ssmq_integrand *> vfd(n_jobs), vfo(n_jobs);
        for (int i = 0; i < n_jobs; ++i) {
            const ssmq_filter_job &j = jobs[i];
            FusedArgs &a = fa[i];
            memset(&a, 0, sizeof(a));
            a.y = j.d_y; a.m0 = j.d_m0; a.P0 = j.d_P0; a.fm = j.d_fm; a.fP = j.d_fP; a.status = j.d_status;
            a.c_dyn = j.h_dyn->d_small; a.c_obs = j.h_obs->d_small; a.gqg = jc[i].gqg; a.rr = jc[i].rr; a.B = j.B; a.ld = j.ld; a.T = j.T;
            a.emv_dyn = j.h_dyn->emv_mode; a.emv_obs = j.h_obs->emv_mode; a.nu_dyn = j.h_dyn->tp_nu; a.nu_obs = j.h_obs->tp_nu;
            a.sscale = jc[i].svec; a.student_dof = j.dof; a.lpw = 64;
            fill_fpar(j.f_dyn, &a.fd);
            fill_fpar(j.f_obs, &a.fo);
            a.fd.ttab = jc[i].ttd;
            a.fo.ttab = jc[i].tto;
            vhd[i] = j.h_dyn; vho[i] = j.h_obs; vfd[i] = j.f_dyn; vfo[i] = j.f_obs;
        }
        std::vector<char> table;
        int blocks = 0;
        if (multi_family_table(n_jobs, vhd.data(), vfd.data(), vho.data(), vfo.data(), fa.data(), &table, &blocks) == 1) {
            mc.htable = table;
            rc = multi_family_launch(mc.htable.data(), blocks, s);
            if (rc) return rc;
            mc.table_blocks = blocks;
            mc.key = key;
            return SSMQ_OK;
        }
    }
    const bool capture = n_fused > 0 && !ssmq::sw("SSMQ_MULTI_NO_GRAPH");
    Ctx &cx = ctx();
    struct StripsOff {
        Ctx &c;
        explicit StripsOff(Ctx &c_) : c(c_) { c.no_strips = true; }
        ~StripsOff() { c.no_strips = false; }
    } strips_off(cx);
    if (n_fused > 0) {
        if (capture) SSMQ_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        rc = hip_fail(hipEventRecord(mc.fork, s), "hipEventRecord");
        int b = 0;
        for (int i = 0; i < n_jobs && !rc; ++i) {
            if (!mc.fused[i]) continue;
            const ssmq_filter_job &j = jobs[i];
            hipStream_t bs = mc.side[b];
            rc = hip_fail(hipStreamWaitEvent(bs, mc.fork, 0), "hipStreamWaitEvent");
            if (!rc) {
                const int r = try_launch_fused(j.h_dyn, j.f_dyn, j.h_obs, j.f_obs, sel[i], j.B, j.ld, j.T, j.d_y, j.d_m0, j.d_P0, jc[i].gqg, jc[i].rr,
                                               j.d_fm, j.d_fP, j.d_status, bs, nullptr, false, jc[i].svec, j.dof, jc[i].ttd, jc[i].tto);
                rc = r < 0 ? r : (r == 1 ? 0 : SSMQ_E_UNSUPPORTED);
            }
            if (!rc) rc = hip_fail(hipEventRecord(mc.joined[b], bs), "hipEventRecord");
            if (!rc) rc = hip_fail(hipStreamWaitEvent(s, mc.joined[b], 0), "hipStreamWaitEvent");
            ++b;
        }
        if (capture) {
            hipGraph_t g = nullptr;
            const hipError_t ce = hipStreamEndCapture(s, &g);
            if (rc) {
                if (g) hipGraphDestroy(g);
                return rc;
            }
            SSMQ_HIP(ce);
            mc.graph = g;
            SSMQ_HIP(hipGraphInstantiate(&mc.exec, mc.graph, nullptr, nullptr, 0));
            mc.key = key;
            SSMQ_HIP(hipGraphLaunch(mc.exec, s));
        } else if (rc) {
            return rc;
        }
    }
    return run_rest();
}

// ---- the host-array forward pass as a pipeline of time blocks (round 6) --------------------------------------------------------
// forward_pass returns host arrays (ssinf.py:66-118): at configs[1] 8 MB of measurements go up and 16 MB of filtered moments come
// down around a 34 us kernel.  Upload, pass and downloads back to back cost 1.15 ms per call (round 5).  Here the pass runs as K
// launches of k_filter_range (ssmq_filter_piped.hip: steps [kb, ke) of every trajectory, state handed from launch to launch, the
// whole-pass kernel's bits), and three queues overlap: a copy-in stream feeds the measurements of block k + 1, the context's stream
// runs block k, a copy-out stream brings block k - 1 back.  Outputs that live in page-locked memory (ssmq_pinned_alloc: what
// ssmtoybox_amd hands out as the returned ndarrays) are written by the copy engine in the reference's (D, T, B) layout directly -
// strided 2-D copies, no staging and no host-side memcpy of the 16 MB; pageable outputs go through the pinned staging block and a
// pool of copy threads.
namespace ssmq {
int try_launch_range(const ssmq_transform *hd, const ssmq_integrand *fd, const ssmq_transform *ho, const ssmq_integrand *fo, int sel_obs,
                     int64_t B, int64_t ld, int T, int kb, int ke, const double *d_y, const double *d_m0, const double *d_P0,
                     const double *d_gqg, const double *d_rr, double *d_fm, double *d_fP, int32_t *d_status, double *hand, hipStream_t s,
                     const char **name, bool dry_run, const double *d_ttab_dyn, const double *d_ttab_obs);
size_t range_hand_doubles(int D);
}
namespace {
// page-locked host blocks, pooled per process: hipHostMalloc of 16 MB costs milliseconds, a pooled block nothing
struct PinnedPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;                 // block -> its (rounded) size
    std::unordered_map<size_t, std::vector<void *>> idle;    // size -> free blocks
    size_t idle_bytes = 0;
    static size_t round(size_t b) { return (std::max<size_t>(b, 8) + 65535) / 65536 * 65536; }
};
PinnedPool &pinned_pool() {
    static PinnedPool *p = new PinnedPool;      // (never destroyed: blocks may outlive static destruction order)
    return *p;
}
constexpr size_t kPinnedIdleCap = size_t(1) << 30;

std::atomic<bool> g_forked{false};        // a forked child has the pool object but none of its threads: it copies by itself
// a few persistent threads for the row copies between caller memory and the pinned blocks (std::thread per call costs 30-50 us)
struct CopyPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> th;
    std::function<void(int64_t, int64_t)> fn;
    int64_t n = 0, chunk = 0;
    std::atomic<int64_t> next{0};
    int pending = 0;
    uint64_t gen = 0;
    bool stop = false;
    explicit CopyPool(int workers) {
        for (int i = 0; i < workers; ++i)
            th.emplace_back([this] {
                uint64_t seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> l(mu);
                        cv_work.wait(l, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                    }
                    drain();
                    std::lock_guard<std::mutex> l(mu);
                    if (--pending == 0) cv_done.notify_all();
                }
            });
    }
    void drain() {
        for (;;) {
            const int64_t lo = next.fetch_add(chunk);
            if (lo >= n) return;
            fn(lo, std::min(n, lo + chunk));
        }
    }
    // fn(lo, hi) over [0, n) in pieces of `chunk`; the caller works too; returns when everything is done
    void run(int64_t n_, int64_t chunk_, std::function<void(int64_t, int64_t)> f) {
        if (n_ <= chunk_ || th.empty() || g_forked.load()) {
            f(0, n_);
            return;
        }
        {
            std::lock_guard<std::mutex> l(mu);
            fn = std::move(f);
            n = n_;
            chunk = chunk_;
            next = 0;
            pending = (int)th.size();
            ++gen;
        }
        cv_work.notify_all();
        drain();
        std::unique_lock<std::mutex> l(mu);
        cv_done.wait(l, [&] { return pending == 0; });
    }
};
CopyPool &copy_pool() {
    static CopyPool *p = [] {
        pthread_atfork(nullptr, nullptr, [] { g_forked.store(true); });
        return new CopyPool((int)std::max(1u, std::min(6u, std::thread::hardware_concurrency() / 2)));
    }();
    return *p;
}
std::mutex g_copy_pool_mu;       // one user of the pool at a time (calls from several threads take turns; the copies are short)

// rows of B doubles: planes row r = (t - t0) * n_elem + e of a block <-> host row (e, t) of an (n_elem, n_outer, B) array
void copy_rows_pool(bool to_planes, double *host, double *pinned, int64_t t0, int64_t t1, int n_outer, int n_elem, int64_t B, int64_t ld) {
    const int64_t rows = (t1 - t0) * n_elem;
    const int64_t per = std::max<int64_t>(1, (128 * 1024) / std::max<int64_t>(B, 1));        // ~1 MB of doubles per piece
    std::lock_guard<std::mutex> g(g_copy_pool_mu);
    copy_pool().run(rows, per, [=](int64_t r0, int64_t r1) {
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
    });
}

struct PipeCache {
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev_in, ev_run, ev_out;
    hipEvent_t ev_done = nullptr;       // the previous call's last use of the device block / pinned blocks
    void drop() {
        if (s_in) hipStreamDestroy(s_in);
        if (s_out) hipStreamDestroy(s_out);
        for (auto *v : {&ev_in, &ev_run, &ev_out}) {
            for (hipEvent_t e : *v) hipEventDestroy(e);
            v->clear();
        }
        s_in = s_out = nullptr;
    }
};
}  // namespace
namespace ssmq {
void drop_pipe_cache() {
    Ctx &c = ctx();
    if (c.pipe) ((PipeCache *)c.pipe)->drop();
}
}  // namespace ssmq

extern "C" int ssmq_pinned_alloc(size_t bytes, void **p) {
    if (!p) return SSMQ_E_ARG;
    *p = nullptr;
    int rc = ensure_device();
    if (rc) return rc;
    PinnedPool &pp = pinned_pool();
    const size_t sz = PinnedPool::round(bytes);
    {
        std::lock_guard<std::mutex> l(pp.mu);
        auto it = pp.idle.find(sz);
        if (it != pp.idle.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            pp.idle_bytes -= sz;
            pp.live[*p] = sz;
            return SSMQ_OK;
        }
    }
    void *q = nullptr;
    SSMQ_HIP(hipHostMalloc(&q, sz, hipHostMallocPortable));
    std::lock_guard<std::mutex> l(pp.mu);
    pp.live[q] = sz;
    *p = q;
    return SSMQ_OK;
}
extern "C" int ssmq_pinned_free(void *p) {
    if (!p) return SSMQ_OK;
    PinnedPool &pp = pinned_pool();
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> l(pp.mu);
        auto it = pp.live.find(p);
        if (it == pp.live.end()) {
            set_error("pinned_free: not a block of ssmq_pinned_alloc");
            return SSMQ_E_ARG;
        }
        sz = it->second;
        pp.live.erase(it);
        if (pp.idle_bytes + sz <= kPinnedIdleCap) {
            pp.idle[sz].push_back(p);
            pp.idle_bytes += sz;
            return SSMQ_OK;
        }
    }
    hipHostFree(p);
    return SSMQ_OK;
}
extern "C" int ssmq_pinned_is_block(const void *p) {
    PinnedPool &pp = pinned_pool();
    std::lock_guard<std::mutex> l(pp.mu);
    return pp.live.count(const_cast<void *>(p)) ? 1 : 0;
}

extern "C" int ssmq_filter_forward_piped(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs, const ssmq_integrand *f_obs,
                                         int64_t B, int T, const double *y, const double *m0, const double *P0, const double *GQG,
                                         const double *R, double *fm, double *fP, int32_t *status, int flags, int n_blocks) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!h_dyn || !h_obs || !f_dyn || !f_obs || B < 0 || T < 0 || !y || !m0 || !P0 || !fm || !fP || !status || n_blocks < 0) {
        set_error("filter_forward_piped: bad argument");
        return SSMQ_E_ARG;
    }
    const int D = h_dyn->D, Y = h_obs->E;
    if (h_dyn->E != D || h_obs->D != D) {
        set_error("filter_forward_piped: additive-noise filter needs dyn (D -> D) and obs (D -> Y) transforms");
        return SSMQ_E_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0 || T == 0) {
        for (int64_t b = 0; b < B; ++b) status[b] = 0;
        return SSMQ_OK;
    }
    FInfo fio;
    if (!integrand_info(f_obs->id, &fio)) {
        set_error("unknown integrand id");
        return SSMQ_E_ARG;
    }
    const int sel = sel_pattern(f_obs, fio.din);
    const char *kname = nullptr;
    // (a forced route - wave split, quad, strips, lanes per wave - means the caller wants THAT kernel: not pipelined)
    if (ssmq::sw("SSMQ_NO_FUSED") || ssmq::sw("SSMQ_NO_PIPED") || ssmq::sw("SSMQ_FUSED_WSPLIT") || ssmq::sw("SSMQ_FUSED_QUAD") ||
        ssmq::sw("SSMQ_FUSED_CHUNKED") || ssmq::sw("SSMQ_FUSED_LPW") ||
        try_launch_range(h_dyn, f_dyn, h_obs, f_obs, sel, B, 0, T, 0, T, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                         nullptr, &kname, true, nullptr, nullptr) != 1) {
        set_error("filter_forward_piped: no time-block kernel for this (models, shapes, form) combination");
        return SSMQ_E_UNSUPPORTED;
    }
    const bool out_pinned = (flags & SSMQ_PIPED_OUT_PINNED) != 0, per_traj = (flags & SSMQ_PIPED_X0_PER_TRAJECTORY) != 0;
    const int64_t ld = (B + 63) / 64 * 64, nblk = ld / 64;
    hipStream_t s = stream();
    Ctx &cx = ctx();
    if (!cx.pipe) cx.pipe = new PipeCache;
    PipeCache &pc = *(PipeCache *)cx.pipe;
    if (!pc.s_in) SSMQ_HIP(hipStreamCreateWithFlags(&pc.s_in, hipStreamNonBlocking));
    if (!pc.s_out) SSMQ_HIP(hipStreamCreateWithFlags(&pc.s_out, hipStreamNonBlocking));
    // ---- time blocks: ~8 MB of output per block, at most 16 (measured at configs[1], 16 MB of output: 0.62 ms with one block,
    // 0.55 with two or three, 0.59 with five, 0.66 with seven, 0.92 with sixteen - every block costs its launches and events) ----
    const size_t out_step = sizeof(double) * ((size_t)D + (size_t)D * D) * ld;
    int K = n_blocks;
    if (K == 0) K = (int)std::max<size_t>(1, std::min<size_t>(16, (out_step * (size_t)T + (size_t(8) << 20) - 1) / (size_t(8) << 20)));
    K = std::max(1, std::min(K, T));
    while ((int)pc.ev_in.size() < K) {
        hipEvent_t a = nullptr, b = nullptr, c = nullptr;
        SSMQ_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        SSMQ_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        SSMQ_HIP(hipEventCreateWithFlags(&c, hipEventDisableTiming));
        pc.ev_in.push_back(a); pc.ev_run.push_back(b); pc.ev_out.push_back(c);
    }
    // ---- device block and pinned staging (grow-only, per context) ---------------------------------------------------------------
    const size_t n_y = (size_t)T * Y * ld, n_m = (size_t)D * ld, n_P = (size_t)D * D * ld, n_fm = (size_t)T * D * ld, n_fP = (size_t)T * D * D * ld;
    const size_t n_c = ((size_t)D * D + (size_t)Y * Y + 2 * (size_t)T + 7) / 8 * 8, n_hand = (size_t)nblk * range_hand_doubles(D);
    const size_t d_dbl = n_y + n_m + n_P + n_fm + n_fP + n_c + n_hand;
    // pageable results are staged block by block through TWO slots (block k + 1 lands while block k - 1's slot is free again)
    auto t_of = [&](int k) { return (int)((int64_t)T * k / K); };
    size_t blk_steps = 0;
    for (int k = 0; k < K; ++k) blk_steps = std::max<size_t>(blk_steps, (size_t)(t_of(k + 1) - t_of(k)));
    const size_t slot_dbl = blk_steps * ((size_t)D + (size_t)D * D) * ld;
    const size_t hin_dbl = n_y + n_m + n_P + n_c, hout_dbl = (out_pinned ? 0 : 2 * slot_dbl);
    // (the staging arena may be resized: nothing of an earlier call is in flight - every call ends with its last copy complete)
    if ((rc = g_stage.reserve(sizeof(double) * d_dbl + sizeof(int32_t) * ld, sizeof(double) * hin_dbl, sizeof(double) * hout_dbl + sizeof(int32_t) * ld)))
        return rc;
    double *dv = (double *)g_stage.dev;
    double *d_y = dv; dv += n_y;
    double *d_m0 = dv; dv += n_m;
    double *d_P0 = dv; dv += n_P;
    double *d_c = dv; dv += n_c;             // (m0 | P0 | constants: ONE transfer, same order as in the pinned block)
    double *d_fm = dv; dv += n_fm;
    double *d_fP = dv; dv += n_fP;
    double *d_hand = dv; dv += n_hand;
    int32_t *d_st = (int32_t *)dv;
    double *hin = (double *)g_stage.hin;
    double *h_y = hin, *h_m0 = hin + n_y, *h_P0 = h_m0 + n_m, *h_c = h_P0 + n_P;
    double *h_out = (double *)g_stage.hout;       // [2][slot_dbl]: a block's means, then its covariances
    int32_t *h_st = (int32_t *)((double *)g_stage.hout + hout_dbl);
    // ---- constants and initial moments, then the first block of measurements: one transfer ------------------------------------
    double *c_gqg = h_c, *c_rr = c_gqg + D * D, *c_ttd = c_rr + Y * Y, *c_tto = c_ttd + T;
    for (int i = 0; i < D * D; ++i) c_gqg[i] = GQG ? GQG[i] : 0.0;
    for (int i = 0; i < Y * Y; ++i) c_rr[i] = R ? R[i] : 0.0;
    const bool has_td = time_table(f_dyn->id, T, c_ttd), has_to = time_table(f_obs->id, T, c_tto);
    if (per_traj) {
        for (int d = 0; d < D; ++d)
            for (int64_t b = 0; b < ld; ++b) h_m0[(size_t)d * ld + b] = b < B ? m0[b * D + d] : 0.0;
        for (int i = 0; i < D * D; ++i)
            for (int64_t b = 0; b < ld; ++b) h_P0[(size_t)i * ld + b] = b < B ? P0[b * D * D + i] : ((i / D == i % D) ? 1.0 : 0.0);
    } else {
        for (int d = 0; d < D; ++d) std::fill(h_m0 + (size_t)d * ld, h_m0 + (size_t)(d + 1) * ld, m0[d]);
        for (int i = 0; i < D * D; ++i) std::fill(h_P0 + (size_t)i * ld, h_P0 + (size_t)(i + 1) * ld, P0[i]);
    }
    SSMQ_HIP(hipMemcpyAsync(d_m0, h_m0, sizeof(double) * (n_m + n_P + n_c), hipMemcpyHostToDevice, pc.s_in));      // m0 | P0 | consts are adjacent
    const double *dc_gqg = d_c, *dc_rr = d_c + D * D, *dc_ttd = has_td ? d_c + D * D + Y * Y : nullptr, *dc_tto = has_to ? d_c + D * D + Y * Y + T : nullptr;
    auto drain = [&](int k) -> int {        // block k's outputs are in host memory: bring them into the caller's arrays if staged
        SSMQ_HIP(hipEventSynchronize(pc.ev_out[k]));
        if (!out_pinned) {
            const int kb = t_of(k), ke = t_of(k + 1);
            double *slot = h_out + (size_t)(k & 1) * slot_dbl;
            copy_rows_pool(false, fm, slot, kb, ke, T, D, B, ld);
            copy_rows_pool(false, fP, slot + (size_t)(ke - kb) * D * ld, kb, ke, T, D * D, B, ld);
        }
        return SSMQ_OK;
    };
    for (int k = 0; k < K; ++k) {
        const int kb = t_of(k), ke = t_of(k + 1);
        copy_rows_pool(true, const_cast<double *>(y), h_y + (size_t)kb * Y * ld, kb, ke, T, Y, B, ld);
        SSMQ_HIP(hipMemcpyAsync(d_y + (size_t)kb * Y * ld, h_y + (size_t)kb * Y * ld, sizeof(double) * (size_t)(ke - kb) * Y * ld, hipMemcpyHostToDevice, pc.s_in));
        SSMQ_HIP(hipEventRecord(pc.ev_in[k], pc.s_in));
        SSMQ_HIP(hipStreamWaitEvent(s, pc.ev_in[k], 0));
        rc = try_launch_range(h_dyn, f_dyn, h_obs, f_obs, sel, B, ld, T, kb, ke, d_y, d_m0, d_P0, dc_gqg, dc_rr, d_fm, d_fP, d_st, d_hand, s, nullptr, false,
                              dc_ttd, dc_tto);
        if (rc != 1) return rc < 0 ? rc : SSMQ_E_UNSUPPORTED;
        SSMQ_HIP(hipEventRecord(pc.ev_run[k], s));
        SSMQ_HIP(hipStreamWaitEvent(pc.s_out, pc.ev_run[k], 0));
        if (out_pinned) {
            // the copy engine writes the reference's layout: host row (e, t) <- plane row (t, e), rows of B doubles
            for (int e = 0; e < D; ++e)
                SSMQ_HIP(hipMemcpy2DAsync(fm + ((size_t)e * T + kb) * B, sizeof(double) * B, d_fm + ((size_t)kb * D + e) * ld, sizeof(double) * D * ld,
                                          sizeof(double) * B, ke - kb, hipMemcpyDeviceToHost, pc.s_out));
            for (int e = 0; e < D * D; ++e)
                SSMQ_HIP(hipMemcpy2DAsync(fP + ((size_t)e * T + kb) * B, sizeof(double) * B, d_fP + ((size_t)kb * D * D + e) * ld,
                                          sizeof(double) * D * D * ld, sizeof(double) * B, ke - kb, hipMemcpyDeviceToHost, pc.s_out));
        } else {
            double *slot = h_out + (size_t)(k & 1) * slot_dbl;
            SSMQ_HIP(hipMemcpyAsync(slot, d_fm + (size_t)kb * D * ld, sizeof(double) * (size_t)(ke - kb) * D * ld, hipMemcpyDeviceToHost, pc.s_out));
            SSMQ_HIP(hipMemcpyAsync(slot + (size_t)(ke - kb) * D * ld, d_fP + (size_t)kb * D * D * ld, sizeof(double) * (size_t)(ke - kb) * D * D * ld,
                                    hipMemcpyDeviceToHost, pc.s_out));
        }
        if (k == K - 1) SSMQ_HIP(hipMemcpyAsync(h_st, d_st, sizeof(int32_t) * ld, hipMemcpyDeviceToHost, pc.s_out));
        SSMQ_HIP(hipEventRecord(pc.ev_out[k], pc.s_out));
        if (k >= 1 && (rc = drain(k - 1))) return rc;
    }
    if ((rc = drain(K - 1))) return rc;
    memcpy(status, h_st, sizeof(int32_t) * B);
    // the context's stream has nothing pending that uses the staging blocks (the last block's copies waited for its kernel)
    return SSMQ_OK;
}

