// Aggregate rate of K host threads that each run their OWN filter through the C ABI (include/ssmq.h) - configs[1]: GPQ-Kalman on
// UNGM (unscented points, N = 3), B = 1e4, T = 100, everything device-resident.  Every thread has its own context (stream,
// workspace), so the passes of different threads overlap on the device: one pass occupies 157 of the chip's 1 024 SIMDs.
// Python cannot show this: the interpreter's ~15 us per call are serial under the GIL (tools/threads_rate.py: 1.36 x at two
// threads, then down).  Build + run: tools/micro/run_threads_rate.sh
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "ssmq.h"

#define CHECK(x) do { int rc__ = (x); if (rc__ < 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc__, ssmq_last_error()); exit(1); } } while (0)

struct Filter {
    ssmq_transform *hd = nullptr, *ho = nullptr;
    ssmq_integrand fd, fo;
    void *y = nullptr, *m0 = nullptr, *P0 = nullptr, *fm = nullptr, *fP = nullptr, *st = nullptr;
    int64_t B, ld;
    int T;
    double gqg[1] = {10.0}, rr[1] = {1.0};
    void build(int64_t B_, int T_, unsigned seed) {
        B = B_; T = T_; ld = (B + 63) / 64 * 64;
        double xi[3], wm[3], wc[3], par[2] = {1.0, 3.0}, upar[3] = {NAN, NAN, NAN};
        CHECK(ssmq_points(SSMQ_PTS_UT, 1, upar, 3, xi, wm, wc));
        double gw[3], gWc[9], gWcc[3], mv[1];
        CHECK(ssmq_weights_gp(1, 3, xi, par, 1, 1e-8, gw, gWc, gWcc, nullptr, nullptr, nullptr, nullptr, mv, nullptr, nullptr));
        hd = ssmq_transform_create(1, 1, 3, SSMQ_FORM_BQ, xi, gw, gWc, gWcc, mv, SSMQ_EMV_DIAG, 0.0, nullptr);
        ho = ssmq_transform_create(1, 1, 3, SSMQ_FORM_BQ, xi, gw, gWc, gWcc, mv, SSMQ_EMV_DIAG, 0.0, nullptr);
        if (!hd || !ho) { fprintf(stderr, "transform_create: %s\n", ssmq_last_error()); exit(1); }
        memset(&fd, 0, sizeof(fd)); memset(&fo, 0, sizeof(fo));
        fd.id = SSMQ_F_UNGM_DYN; fo.id = SSMQ_F_UNGM_MEAS;
        // measurements of simulated UNGM trajectories (ssmod.py:268-269, 1060-1061), planes [T][1][ld]
        std::vector<double> yh((size_t)T * ld, 0.0), mh(ld, 0.0), Ph(ld, 1.0);
        unsigned long long lcg = seed;               // (a generator of its own per filter: rand() is one state for the process)
        auto uni = [&lcg] { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return ((lcg >> 11) + 1.0) / 9007199254740994.0; };
        auto gauss = [&uni] { const double u = uni(), v = uni(); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
        for (int64_t b = 0; b < B; ++b) {
            double x = gauss();
            for (int k = 0; k < T; ++k) {
                x = 0.5 * x + 25 * x / (1 + x * x) + 8 * cos(1.2 * k) + sqrt(10.0) * gauss();
                yh[(size_t)k * ld + b] = 0.05 * x * x + gauss();
            }
        }
        CHECK(ssmq_malloc(&y, 8 * yh.size())); CHECK(ssmq_malloc(&m0, 8 * ld)); CHECK(ssmq_malloc(&P0, 8 * ld));
        CHECK(ssmq_malloc(&fm, 8 * (size_t)T * ld)); CHECK(ssmq_malloc(&fP, 8 * (size_t)T * ld)); CHECK(ssmq_malloc(&st, 4 * ld));
        CHECK(ssmq_memcpy_h2d(y, yh.data(), 8 * yh.size())); CHECK(ssmq_memcpy_h2d(m0, mh.data(), 8 * ld)); CHECK(ssmq_memcpy_h2d(P0, Ph.data(), 8 * ld));
    }
    void pass() {
        CHECK(ssmq_filter_forward_dev(hd, &fd, ho, &fo, B, ld, T, (const double *)y, (const double *)m0, (const double *)P0, gqg, rr,
                                      (double *)fm, (double *)fP, (int32_t *)st));
    }
    double checksum() {
        std::vector<double> h((size_t)T * ld);
        CHECK(ssmq_sync());
        CHECK(ssmq_memcpy_d2h(h.data(), fm, 8 * h.size()));
        double s = 0;
        for (int k = 0; k < T; ++k) for (int64_t b = 0; b < B; ++b) s += h[(size_t)k * ld + b];
        return s;
    }
};

int main(int argc, char **argv) {
    const int64_t B = 10000;
    const int T = 100, passes = 400;
    CHECK(ssmq_set_device(0));
    // the reference result: one thread, one filter
    Filter ref; ref.build(B, T, 7); ref.pass(); const double want = ref.checksum();
    std::vector<int> ks;
    for (int i = 1; i < argc; ++i) ks.push_back(atoi(argv[i]));
    if (ks.empty()) ks = {1, 2, 4, 6, 8};
    for (int K : ks) {
        std::vector<std::thread> th;
        std::vector<double> sums(K);
        std::atomic<int> ready{0};
        std::atomic<bool> go{false};
        std::vector<std::chrono::steady_clock::time_point> t0(K), t1(K);
        for (int k = 0; k < K; ++k)
            th.emplace_back([&, k] {
                Filter f; f.build(B, T, 7);                // (every thread its own handles and buffers, the same data)
                for (int i = 0; i < 20; ++i) f.pass();
                CHECK(ssmq_sync());
                ++ready;
                while (!go.load()) std::this_thread::yield();
                t0[k] = std::chrono::steady_clock::now();
                for (int i = 0; i < passes; ++i) f.pass();
                CHECK(ssmq_sync());
                t1[k] = std::chrono::steady_clock::now();
                sums[k] = f.checksum();
            });
        while (ready.load() < K) std::this_thread::yield();
        go.store(true);
        for (auto &t : th) t.join();
        auto a = t0[0], b = t1[0];
        for (int k = 1; k < K; ++k) { if (t0[k] < a) a = t0[k]; if (t1[k] > b) b = t1[k]; }
        const double sec = std::chrono::duration<double>(b - a).count();
        bool same = true;
        for (int k = 0; k < K; ++k) same = same && sums[k] == want;
        printf("%d thread(s): %6.1f us per pass and thread, %.3e filter steps/s in aggregate, results %s the single-thread ones\n", K,
               1e6 * sec / passes, (double)K * passes * B * T / sec, same ? "EQUAL" : "DIFFER FROM");
        fflush(stdout);
    }
    return 0;
}
