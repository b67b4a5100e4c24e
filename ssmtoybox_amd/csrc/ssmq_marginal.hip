// Batched Laplace step of the marginalised GP-quadrature filter: B independent BFGS runs in lock step (host code).
//
// Reference: MarginalInference._param_posterior_moments (ssinf.py:1243-1273) runs scipy.optimize.minimize(method='BFGS')
// on the negative log posterior of the kernel parameters, ONE trajectory at a time, once per time step; its callers loop
// over Monte-Carlo trajectories in Python (research/tpq/tpq_base.py:175-192).  The objective of trajectory b at theta is
//   - log N(y_b | moments of the theta-conditioned filter step)  -  log N(theta | prior mean_b, prior cov_b)
// (ssinf.py:1153-1241) and costs one theta step on the device (ssmq_gp_theta_step); a forward-difference gradient costs
// param_dim more.  Here every trajectory keeps its own optimiser state and each ROUND sends the points all unfinished
// trajectories are waiting for - (param_dim + 1) per trajectory - to the device in ONE ssmq_gp_theta_step call.
//
// The optimiser is a restatement of what SciPy 1.15.3 (the pinned version of this image; not part of the reference tree) runs
// for method='BFGS' with jac=True and default options: _minimize_bfgs (gtol 1e-5 on the max-norm, maxiter 200 n, initial
// inverse Hessian I, "old_old_fval = f0 + |g0| / 2"), line_search_wolfe1 -> scalar_search_wolfe1 (c1 1e-4, c2 0.9, amin 1e-100,
// amax 1e100, xtol 1e-14, at most 100 trial steps) -> MINPACK-2's DCSRCH / DCSTEP (More' & Thuente; SciPy's _dcsrch.py), written
// as a per-trajectory state machine because the function values arrive a round later.  Where SciPy falls back to its second
// line search (line_search_wolfe2: DCSRCH ended in an ERROR or WARNING task, or refused to start) the state machine goes on
// with scalar_search_wolfe2 / _zoom as well (PH_LINE2 below); the status SSMQ_BFGS_FALLBACK of round 3's first version
// ("the caller finishes this trajectory with SciPy") is no longer produced.
// Two deliberate differences from the reference's serial path, both where its objective RAISES: an objective point whose
// kernel matrix / covariance is not positive definite (or whose value is not finite) counts as +inf here and the search goes
// on - numpy.linalg.LinAlgError propagates out of scipy.optimize.minimize in the reference and ends that trajectory's
// forward_pass (ssinf.py:1088-1122); the batch instead reports it through `failed` only if the Laplace covariance or a
// marginalisation point then fails.  And NaN is tested before maxiter (SciPy: warnflag 1 before 3).
#include "ssmq_host.h"
#include "ssmq_bfgs.h"
#include "ssmq_theta_item.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

using namespace ssmq;

namespace {

using namespace ssmq_bfgs;
using Run = RunT<kMaxPar>;

// why a trajectory left the batch (failed[b] = step + 65536 reason; include/ssmq.h)
enum { WHY_PRIOR_NOT_PD = 1, WHY_LAPLACE_NOT_FINITE = 2, WHY_LAPLACE_NOT_PD = 3, WHY_MIXTURE_ITEM = 4, WHY_MIXTURE_NOT_FINITE = 5 };
__host__ __device__ inline int32_t why(int k, int reason) { return k < 65536 ? k + 65536 * reason : k; }

// values of the objective at n rows of parameters: rows [n][P], `traj[i]` = the trajectory row i belongs to; vals [n]
struct Evaluator {
    virtual int eval(int64_t n, const int64_t *traj, const double *rows, double *vals) = 0;
    virtual ~Evaluator() {}
};

// One trajectory's optimiser takes the objective values at its pending point r.xt (vals[0]) and at the forward-difference
// points (vals[1 + i]: r.xt + fd_step e_i) and either finishes (r.phase = PH_DONE, r.status) or leaves the next point in r.xt.
// B BFGS runs in lock step.  theta [B][P] start points in / minimisers out; skip[b] != 0: trajectory b is not run (status kept).
int bfgs_lockstep(int64_t B, int P, double fd_step, Evaluator &ev, double *theta, double *hess_inv, int32_t *status, int32_t *iters,
                  int64_t *rounds_out) {
    std::vector<Run> run((size_t)B);
    for (int64_t b = 0; b < B; ++b) {
        Run &r = run[b];
        bfgs_start(r, P, theta + (size_t)b * P);
        if (status[b] != 0) {
            r.phase = PH_DONE;
            r.status = status[b];
        }
    }
    std::vector<int64_t> want, traj;
    std::vector<double> rows, vals;
    int64_t rounds = 0;
    const int per = P + 1;
    for (;;) {
        want.clear();
        for (int64_t b = 0; b < B; ++b)
            if (run[b].phase != PH_DONE) want.push_back(b);
        if (want.empty()) break;
        const int64_t nw = (int64_t)want.size(), items = nw * per;
        rows.resize((size_t)items * P); vals.resize((size_t)items); traj.resize((size_t)items);
        // objective and forward-difference gradient at xt: rows [xt; xt + h e_i]
        for (int64_t w = 0; w < nw; ++w) {
            const Run &r = run[want[w]];
            for (int j = 0; j < per; ++j) {
                traj[(size_t)(w * per + j)] = want[w];
                for (int i = 0; i < P; ++i) rows[(size_t)(w * per + j) * P + i] = r.xt[i] + ((j == i + 1) ? fd_step : 0.0);
            }
        }
        const int rc = ev.eval(items, traj.data(), rows.data(), vals.data());
        if (rc < 0) return rc;
        ++rounds;
        for (int64_t w = 0; w < nw; ++w) bfgs_advance(run[want[w]], P, fd_step, &vals[(size_t)(w * per)]);
    }
    for (int64_t b = 0; b < B; ++b) {
        const Run &r = run[b];
        for (int i = 0; i < P; ++i) theta[(size_t)b * P + i] = r.x[i];
        std::memcpy(hess_inv + (size_t)b * P * P, r.H, sizeof(double) * P * P);
        status[b] = r.status;
        if (iters) iters[b] = r.k;
    }
    if (rounds_out) *rounds_out = rounds;
    return SSMQ_OK;
}

// objective of the marginalised filter: -log N(y_b | theta-conditioned step) - log N(theta | prior_b)   (ssinf.py:1153-1241)
struct MarginalObjective : Evaluator {
    ssmq_transform *h_dyn, *h_obs;
    const ssmq_integrand *f_dyn, *f_obs;
    int Din, D, Y, Pd, Po, P;
    double jitter, time;
    const double *mean, *cov, *y, *GQG, *R, *prior_mean;
    std::vector<double> Lp, logdet2;                 // per trajectory: Cholesky factor of the prior covariance, 2 sum log diag
    std::vector<double> pd, po, mm, cc, yy, ll, om, oc;
    std::vector<int32_t> st;
    // log N(theta | m, C) = -(v'v + 2 sum log diag L + P log 2 pi) / 2, v = L^-1 (theta - m)   (ssinf.py:1200-1218)
    double log_prior(int64_t b, const double *th) const {
        const double *L = &Lp[(size_t)b * P * P], *m = prior_mean + (size_t)b * P;
        double v[kMaxPar], q = 0.0;
        for (int i = 0; i < P; ++i) {
            double s = th[i] - m[i];
            for (int k = 0; k < i; ++k) s -= L[i * P + k] * v[k];
            v[i] = s / L[i * P + i];
            q += v[i] * v[i];
        }
        return -0.5 * (q + logdet2[(size_t)b] + P * std::log(2.0 * M_PI));
    }
    int eval(int64_t items, const int64_t *traj, const double *rows, double *vals) override {
        pd.resize((size_t)items * Pd); po.resize((size_t)items * Po);
        mm.resize((size_t)items * Din); cc.resize((size_t)items * Din * Din); yy.resize((size_t)items * Y);
        ll.resize((size_t)items); om.resize((size_t)items * D); oc.resize((size_t)items * D * D); st.assign((size_t)items, 0);
        for (int64_t it = 0; it < items; ++it) {
            const int64_t b = traj[it];
            for (int i = 0; i < P; ++i) {
                const double e = std::exp(rows[(size_t)it * P + i]);        // the kernel parameters are exp(theta)
                if (i < Pd) pd[(size_t)it * Pd + i] = e;
                else po[(size_t)it * Po + (i - Pd)] = e;
            }
            std::memcpy(&mm[(size_t)it * Din], mean + (size_t)b * Din, sizeof(double) * Din);
            std::memcpy(&cc[(size_t)it * Din * Din], cov + (size_t)b * Din * Din, sizeof(double) * Din * Din);
            std::memcpy(&yy[(size_t)it * Y], y + (size_t)b * Y, sizeof(double) * Y);
        }
        const int rc = ssmq_gp_theta_step(h_dyn, f_dyn, h_obs, f_obs, items, pd.data(), po.data(), jitter, mm.data(), cc.data(), 0,
                                          yy.data(), 0, time, GQG, R, om.data(), oc.data(), ll.data(), st.data());
        if (rc < 0) return rc;          // argument / device error; rc > 0 only reports items that are not positive definite
        for (int64_t it = 0; it < items; ++it) vals[it] = -ll[(size_t)it] - log_prior(traj[it], rows + (size_t)it * P);
        return 0;
    }
};

// a host function as the objective: the optimiser's restatement is pinned against scipy.optimize.minimize on the CPU with it
struct CallbackObjective : Evaluator {
    ssmq_objective_fn fn;
    void *ctx;
    int P;
    int eval(int64_t items, const int64_t *traj, const double *rows, double *vals) override { return fn(ctx, items, P, traj, rows, vals); }
};

}  // namespace

extern "C" int ssmq_bfgs_lockstep_host(ssmq_objective_fn fn, void *ctx, int64_t B, int P, double fd_step, double *theta,
                                       double *hess_inv, int32_t *status, int32_t *iters, int64_t *rounds) {
    if (!fn || B < 0 || P < 1 || P > kMaxPar || (B > 0 && (!theta || !hess_inv || !status))) {
        set_error("bfgs_lockstep_host: bad argument");
        return SSMQ_E_ARG;
    }
    for (int64_t b = 0; b < B; ++b) status[b] = 0;
    CallbackObjective ev;
    ev.fn = fn; ev.ctx = ctx; ev.P = P;
    return bfgs_lockstep(B, P, fd_step, ev, theta, hess_inv, status, iters, rounds);
}

extern "C" int ssmq_gp_marginal_laplace_batch(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                              const ssmq_integrand *f_obs, int64_t B, double jitter, const double *mean,
                                              const double *cov, const double *y, double time, const double *GQG, const double *R,
                                              const double *prior_mean, const double *prior_cov, double fd_step, double *theta,
                                              double *hess_inv, int32_t *status, int32_t *iters, int64_t *rounds_out) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!h_dyn || !h_obs || !f_dyn || !f_obs || B < 0 || (B > 0 && (!mean || !cov || !y || !prior_mean || !prior_cov || !theta ||
                                                                     !hess_inv || !status))) {
        set_error("marginal_laplace_batch: null argument");
        return SSMQ_E_ARG;
    }
    MarginalObjective ev;
    ev.h_dyn = h_dyn; ev.h_obs = h_obs; ev.f_dyn = f_dyn; ev.f_obs = f_obs;
    ev.Din = h_dyn->D; ev.D = h_dyn->E; ev.Y = h_obs->E; ev.Pd = ev.Din + 1; ev.Po = h_obs->D + 1; ev.P = ev.Pd + ev.Po;
    ev.jitter = jitter; ev.time = time; ev.mean = mean; ev.cov = cov; ev.y = y; ev.GQG = GQG; ev.R = R; ev.prior_mean = prior_mean;
    const int P = ev.P;
    if (P > kMaxPar) {
        set_error("marginal_laplace_batch: too many kernel parameters");
        return SSMQ_E_UNSUPPORTED;
    }
    if (rounds_out) *rounds_out = 0;
    if (B == 0) return SSMQ_OK;
    ev.Lp.assign((size_t)B * P * P, 0.0);
    ev.logdet2.assign((size_t)B, 0.0);
    for (int64_t b = 0; b < B; ++b) {
        const double *C = prior_cov + (size_t)b * P * P;
        double *L = &ev.Lp[(size_t)b * P * P];
        status[b] = 0;
        for (int j = 0; j < P && status[b] == 0; ++j) {
            double s = C[j * P + j];
            for (int k = 0; k < j; ++k) s -= L[j * P + k] * L[j * P + k];
            if (!(s > 0.0)) {           // numpy.linalg.cholesky would raise in _param_log_prior
                status[b] = SSMQ_BFGS_PRIOR_NOT_PD;
                break;
            }
            const double ljj = std::sqrt(s);
            L[j * P + j] = ljj;
            ev.logdet2[(size_t)b] += 2.0 * std::log(ljj);
            for (int i = j + 1; i < P; ++i) {
                double t = C[i * P + j];
                for (int k = 0; k < j; ++k) t -= L[i * P + k] * L[j * P + k];
                L[i * P + j] = t / ljj;
            }
        }
    }
    return bfgs_lockstep(B, P, fd_step, ev, theta, hess_inv, status, iters, rounds_out);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The whole marginalised filter for B trajectories, every trajectory at its own pace.
//
// ssmq_gp_marginal_laplace_batch keeps the trajectories in lock step PER TIME STEP: a step costs as many device rounds as its
// slowest trajectory needs (a few run 50 BFGS iterations on a noisy objective), and the others wait.  Trajectories are
// independent across time steps as well (research/tpq/tpq_base.py:175-192 loops over them), so here each one walks
// ssinf.py:66-118 / 1083-1273 by itself - Laplace step (BFGS), marginalisation over the parameter sigma points, next time
// step - and a round sends whatever every unfinished trajectory is waiting for, with its own time index
// (ssmq_gp_theta_step_times): (param_dim + 1) objective points or NP marginalisation points.  The number of rounds is the
// LONGEST trajectory's total, not the sum over the steps of the slowest one's.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// The host side of a round (parameter rows in, optimiser steps out) is independent per trajectory: with a thousand trajectories
// in flight it was 40 % of the call (17 + 23 of 101 ms at B = 1 024), so rounds with many active trajectories are cut into
// chunks for a few worker threads that live for the duration of the call.  Small rounds (the long tail of a few slow
// trajectories) run inline: waking the workers costs more than they would save.
class Workers {
public:
    explicit Workers(int n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // fn(begin, end) over [0, n) in chunks, the calling thread included; returns when all of it is done
    void run(size_t n, const std::function<void(size_t, size_t)> &fn) {
        const size_t parts = th_.size() + 1;
        if (th_.empty() || n < 256) {
            fn(0, n);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            chunk_ = (n + parts - 1) / parts;
            pending_ = (int)th_.size();
            ++gen_;
        }
        cv_.notify_all();
        fn(0, std::min(n, chunk_));
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

private:
    void loop(int i) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t)> *fn;
            size_t b, e;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
                b = std::min(n_, chunk_ * (size_t)(i + 1));
                e = std::min(n_, chunk_ * (size_t)(i + 2));
            }
            if (b < e) (*fn)(b, e);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t, size_t)> *fn_ = nullptr;
    size_t n_ = 0, chunk_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct Traj {
    int k = 1;                      // time step being worked on (1 .. T)
    int mode = 0;                   // 0: optimising (BFGS), 1: waiting for the marginalisation points, 2: finished / failed
    Run run;
    double xm[SSMQ_MAX_DIM], xP[SSMQ_MAX_DIM * SSMQ_MAX_DIM];          // filtered state moments
    double pm[kMaxPar], pc[kMaxPar * kMaxPar], Lp[kMaxPar * kMaxPar], logdet2 = 0;   // parameter prior of this step, its factor
    double pts[kMaxPar * 2 * kMaxPar];                                  // [NP][P] marginalisation points of this step
};

// ---- device-resident rounds ------------------------------------------------------------------------------------------------------
// The same filter with the per-trajectory state machines ON THE DEVICE: one thread per trajectory packs the points it is
// waiting for straight into the theta step's device arena (k_mg_scan: item offsets by a block-wide scan; k_mg_fill), the theta
// step runs on them with the item count read from device memory (theta_dev_enqueue: k_theta_weights, k_theta_chain), and the same
// thread takes the values and advances its optimiser / mixture / time step (k_mg_advance: bfgs_advance of ssmq_bfgs.h, the code the
// host rounds run).  The host only queues rounds - four or five launches each, no copy, no synchronisation - a few ahead of the
// progress the device reports through two integers in pinned host memory (unfinished trajectories - also the bound of the next
// launches' grids - and scans done).
// Round 4's host rounds cost ~85 us each (55 us of which copies, synchronisation and host turn-around: DESIGN.md 3.13) and
// the number of rounds is set by the ONE longest trajectory.
constexpr int kRoundsAhead = 12;       // rounds queued ahead of the device's progress

template <int PM>
struct TrajD {
    int k, mode;                     // time step being worked on (1 .. T); 0: optimising, 1: waiting for the mixture points, 2: done / failed
    RunT<PM> run;
    double xm[SSMQ_MAX_DIM], xP[SSMQ_MAX_DIM * SSMQ_MAX_DIM];
    double pm[PM], pc[PM * PM], Lp[PM * PM], logdet2;
    double pts[PM * 2 * PM];         // [NP][P]
};

struct MgArgs {
    void *traj;
    int64_t B;
    int32_t T, P, Pd, Po, NP, D, Din, Y, dq;
    const double *y;                 // [B][T][Y]
    const double *x0_mean, *x0_cov, *prior_mean, *prior_cov, *q_mean, *q_cov, *upts, *uwts;
    double fd_step, param_jitter;
    int32_t *first;                  // [B] item offset of every trajectory in this round
    signed char *modes;              // [B] TrajD::mode of every trajectory, compact (what the scan reads)
    int32_t *count;                  // [0] items of this round, [1] unfinished trajectories, [2] rounds that had items, [3] scans done
    volatile int32_t *hflag;         // pinned host memory the device writes after every scan: [0] unfinished trajectories, [1] scans done
    unsigned long long *totals;      // [0] items, [1] BFGS iterations
    ThetaDev th;
    double *fm, *fP;                 // [B][T][D], [B][T][D D], NaN where nothing was produced
    int32_t *failed;                 // [B]
};

template <int PM>
__device__ void mg_begin_step(TrajD<PM> &t, const MgArgs &a, int64_t b) {
    if (!chol_lower(t.pc, a.P, t.Lp, &t.logdet2)) {     // numpy.linalg.cholesky would raise in _param_log_prior
        t.mode = 2;
        a.failed[b] = why(t.k, WHY_PRIOR_NOT_PD);
        return;
    }
    bfgs_start(t.run, a.P, t.pm);
    t.mode = 0;
}

// what the host steers by, written to pinned host memory after every scan (ONE thread): the number of unfinished trajectories, then -
// behind a system-scope fence - the number of scans done.  The host never waits for a round: it keeps a few rounds queued ahead
// of the scan count it sees and stops queueing when a scan has found nothing unfinished.
__device__ __forceinline__ void mg_publish(const MgArgs &a, int unfinished) {
    const int seq = a.count[3] + 1;
    a.count[3] = seq;
    if (a.hflag) {
        a.hflag[0] = unfinished;
        __threadfence_system();
        a.hflag[1] = seq;
    }
}

template <int PM>
__global__ void k_mg_init(const MgArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    TrajD<PM> &t = ((TrajD<PM> *)a.traj)[b];
    t.k = 1;
    for (int i = 0; i < a.D; ++i) t.xm[i] = a.x0_mean[i];
    for (int i = 0; i < a.D * a.D; ++i) t.xP[i] = a.x0_cov[i];
    for (int i = 0; i < a.P; ++i) t.pm[i] = a.prior_mean[i];
    for (int i = 0; i < a.P * a.P; ++i) t.pc[i] = a.prior_cov[i];
    a.failed[b] = 0;
    mg_begin_step(t, a, b);
    a.modes[b] = (signed char)t.mode;
}

// item offsets of this round (one workgroup; trajectories in order, so the item order is the host rounds'): every thread takes
// four consecutive trajectories (their modes from the compact mirror a.modes), wave prefix sums by shuffles, the four wave totals
// through LDS
template <int PM>
__global__ __launch_bounds__(256) void k_mg_scan(const MgArgs a) {
    __shared__ int32_t wtot[4], wact[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0, act = 0;
    for (int64_t base = 0; base < a.B; base += 1024) {
        int n[4], mine = 0, alive = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t b = base + 4 * (int64_t)threadIdx.x + q;
            const int mode = b < a.B ? (int)a.modes[b] : 2;
            n[q] = mode == 0 ? a.P + 1 : (mode == 1 ? a.NP : 0);
            mine += n[q];
            alive += mode != 2;
        }
        int incl = mine, asum = alive;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
            asum += __shfl_xor(asum, off, 64);
        }
        if (lane == 63) wtot[wave] = incl;
        if (lane == 0) wact[wave] = asum;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        int run = before + incl - mine;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t b = base + 4 * (int64_t)threadIdx.x + q;
            if (b < a.B) a.first[b] = run;
            run += n[q];
        }
        carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        act += wact[0] + wact[1] + wact[2] + wact[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.count[0] = carry;
        a.count[1] = act;
        if (carry > 0) {
            a.count[2] += 1;
            a.totals[0] += (unsigned long long)carry;
        }
        mg_publish(a, act);
    }
}

// the points trajectory b waits for, as items of the theta step (what the host rounds pack into rows / pd / po / mm / cc / yy / tt);
// one thread per (trajectory, item slot): `per` = max(P + 1, NP) slots per trajectory
// SCAN: the item offsets are formed HERE, by every workgroup for its own trajectories (256 / per of them) from the compact mode
// mirror - a sweep over B bytes per workgroup instead of a kernel of its own (k_mg_scan: 4.4 us + a launch gap per round); used
// while that sweep is short (B <= 8 192).  Workgroup 0 also leaves the round's item and trajectory counts.
template <int PM, bool SCAN>
__global__ __launch_bounds__(256) void k_mg_fill(const MgArgs a, int per) {
    int64_t b;
    int j;
    int32_t first_b = 0;
    if constexpr (SCAN) {
        __shared__ int32_t red[3][4], nloc[64];
        const int tpb = 256 / per;                                  // trajectories of this workgroup (per <= 32: >= 8)
        const int lb = threadIdx.x / per;
        j = threadIdx.x - lb * per;
        const int64_t b_first = (int64_t)blockIdx.x * tpb;
        b = b_first + lb;
        int pre = 0, tot = 0, act = 0;
        for (int64_t i = threadIdx.x; i < a.B; i += 256) {
            const int mode = (int)a.modes[i];
            const int n = mode == 0 ? a.P + 1 : (mode == 1 ? a.NP : 0);
            tot += n;
            act += mode != 2;
            if (i < b_first) pre += n;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            pre += __shfl_xor(pre, off, 64);
            tot += __shfl_xor(tot, off, 64);
            act += __shfl_xor(act, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = pre; red[1][threadIdx.x >> 6] = tot; red[2][threadIdx.x >> 6] = act;
        }
        if (threadIdx.x < 64) {
            const int64_t bb = b_first + threadIdx.x;
            const int mode = (threadIdx.x < tpb && bb < a.B) ? (int)a.modes[bb] : 2;
            nloc[threadIdx.x] = mode == 0 ? a.P + 1 : (mode == 1 ? a.NP : 0);
        }
        __syncthreads();
        first_b = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        for (int l = 0; l < lb && l < tpb; ++l) first_b += nloc[l];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const int total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            a.count[0] = total;
            a.count[1] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
            if (total > 0) {
                a.count[2] += 1;
                a.totals[0] += (unsigned long long)total;
            }
            mg_publish(a, a.count[1]);
        }
        if (lb >= tpb || b >= a.B) return;
        if (j == 0) a.first[b] = first_b;
    } else {
        const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        b = tid / per;
        j = (int)(tid - b * per);
        if (b >= a.B) return;
        first_b = a.first[b];
    }
    const TrajD<PM> &t = ((const TrajD<PM> *)a.traj)[b];
    if (t.mode == 2) return;
    const int P = a.P, Pd = a.Pd, Po = a.Po, D = a.D, Din = a.Din, Y = a.Y, dq = a.dq;
    const int n = t.mode == 0 ? P + 1 : a.NP;
    if (j >= n) return;
    const int64_t ld = a.th.ld;
    const int64_t it = (int64_t)first_b + j;
    for (int i = 0; i < P; ++i) {
        const double row = t.mode == 0 ? t.run.xt[i] + ((j == i + 1) ? a.fd_step : 0.0) : t.pts[(size_t)j * P + i];
        const double e = exp(row);                         // the kernel parameters are exp(theta)
        if (i < Pd) a.th.pard[(size_t)it * Pd + i] = e;
        else a.th.paro[(size_t)it * Po + (i - Pd)] = e;
    }
    // [mean; q_mean], blockdiag(cov, Q) for dynamics that take their noise as an argument (ssinf.py:1174-1176)
    double *m = a.th.mean + (size_t)it * Din, *c = a.th.cov + (size_t)it * Din * Din;
    for (int i = 0; i < Din * Din; ++i) c[i] = 0.0;
    for (int i = 0; i < D; ++i) {
        m[i] = t.xm[i];
        for (int k = 0; k < D; ++k) c[i * Din + k] = t.xP[i * D + k];
    }
    for (int i = 0; i < dq; ++i) {
        m[D + i] = a.q_mean[i];
        for (int k = 0; k < dq; ++k) c[(D + i) * Din + D + k] = a.q_cov[i * dq + k];
    }
    for (int k = 0; k < Y; ++k) a.th.ysoa[(size_t)k * ld + it] = a.y[((size_t)b * a.T + (t.k - 1)) * Y + k];
    a.th.tt[it] = (double)t.k;
}

// PX: the parameter count at compile time (= PM), or 0 = a.P at run time.  With PX every loop of the optimiser has a constant trip
// count: unrolled, its small arrays in registers - at run-time bounds they are indexed private memory and the kernel took
// 40 us per round for 1 024 trajectories (a wave walks the union of its lanes' branches, a few thousand dependent instructions).
// where a trajectory's item results of this round are: the theta step's device arena (rounds route) ...
struct ArenaResults {
    const ThetaDev &th;
    int64_t f0;
    __device__ __forceinline__ double ll(int j) const { return th.ll[f0 + j]; }
    __device__ __forceinline__ int32_t st(int j) const { return th.st_all[f0 + j]; }
    __device__ __forceinline__ double m(int i, int j) const { return th.m_fi[(size_t)i * th.ld + f0 + j]; }
    __device__ __forceinline__ double P(int i, int j) const { return th.P_fi[(size_t)i * th.ld + f0 + j]; }
};

template <int PM, int PX, class Res, bool IN_PLACE = false>
__device__ __forceinline__ void mg_advance_one(TrajD<PM> &t, const MgArgs &a, int64_t b, const Res &res);

// kAdvPerWave trajectories per wave (every (64 / kAdvPerWave)-th lane): a wave walks the union of its lanes' branches, fewer
// lanes = fewer of them
#ifndef SSMQ_MG_ADV_PER_WAVE
#define SSMQ_MG_ADV_PER_WAVE 8
#endif
constexpr int kAdvPerWave = SSMQ_MG_ADV_PER_WAVE;
template <int PM, int PX>
__global__ __launch_bounds__(64) void k_mg_advance(const MgArgs a) {
    const int64_t b = (int64_t)blockIdx.x * kAdvPerWave + threadIdx.x / (64 / kAdvPerWave);
    if (threadIdx.x % (64 / kAdvPerWave) != 0 || b >= a.B) return;
    if (a.modes[b] == 2) return;
    mg_advance_one<PM, PX>(((TrajD<PM> *)a.traj)[b], a, b, ArenaResults{a.th, (int64_t)a.first[b]});
    a.modes[b] = (signed char)((const TrajD<PM> *)a.traj)[b].mode;
}

template <int PM, int PX, class Res, bool IN_PLACE>
__device__ __forceinline__ void mg_advance_one(TrajD<PM> &t, const MgArgs &a, int64_t b, const Res &res) {
    const int P = PX ? PX : a.P, D = a.D, NP = a.NP, T = a.T;
    const double inf = __builtin_huge_val();
    if (t.mode == 0) {
        // The optimiser works on a LOCAL copy of its state (private memory: lane-interleaved and cached) and writes it back
        // once: on the 3 KB-strided structs themselves every one of its few hundred dependent accesses was a cache miss of its
        // own (38 us per round for 1 024 trajectories).
        // (IN_PLACE: the state is in LDS - k_mg_persistent - and the optimiser works on it where it is)
        RunT<PM> run_copy;
        if constexpr (!IN_PLACE) run_copy = t.run;
        RunT<PM> &run = IN_PLACE ? t.run : run_copy;
        double vals[PM + 1];
        for (int j = 0; j <= P; ++j) {
            // log N(theta | prior) at the row as it was evaluated (ssinf.py:1200-1218)
            double v[PM], q = 0.0;
            for (int i = 0; i < P; ++i) {
                const double th = run.xt[i] + ((j == i + 1) ? a.fd_step : 0.0);
                double s = th - t.pm[i];
                for (int k = 0; k < i; ++k) s -= t.Lp[i * P + k] * v[k];
                v[i] = s / t.Lp[i * P + i];
                q += v[i] * v[i];
            }
            const double lp = -0.5 * (q + t.logdet2 + P * log(2.0 * M_PI));
            const double val = -res.ll(j) - lp;
            vals[j] = __builtin_isfinite(val) ? val : inf;
        }
        bfgs_advance(run, P, a.fd_step, vals);
        if constexpr (!IN_PLACE) t.run = run;
        if (run.phase != PH_DONE) return;
        atomicAdd(&a.totals[1], (unsigned long long)run.k);
        // Laplace posterior (ssinf.py:1272-1273) and its sigma points (:1103-1106)
        double pcn[PM * PM], L[PM * PM];
        bool fin = true;
        for (int i = 0; i < P; ++i) {
            t.pm[i] = run.x[i];
            fin = fin && __builtin_isfinite(t.pm[i]);
            for (int k = 0; k < P; ++k) {
                pcn[i * P + k] = run.H[i * P + k] + (i == k ? a.param_jitter : 0.0);
                fin = fin && __builtin_isfinite(pcn[i * P + k]);
            }
        }
        if (!fin || !chol_lower(pcn, P, L, nullptr)) {
            t.mode = 2;
            a.failed[b] = why(t.k, fin ? WHY_LAPLACE_NOT_PD : WHY_LAPLACE_NOT_FINITE);
            return;
        }
        for (int i = 0; i < P * P; ++i) t.pc[i] = pcn[i];
        for (int j = 0; j < NP; ++j)
            for (int i = 0; i < P; ++i) {
                double s = t.pm[i];
                for (int k = 0; k <= i; ++k) s += L[i * P + k] * a.upts[(size_t)k * NP + j];
                t.pts[(size_t)j * P + i] = s;
            }
        t.mode = 1;
    } else {
        // mixture over the parameter points (ssinf.py:1108-1115): plain weighted sums of the conditional moments
        bool ok = true;
        for (int j = 0; j < NP; ++j) ok = ok && res.st(j) == 0;
        const bool items_ok = ok;
        double xm[SSMQ_MAX_DIM], xP[SSMQ_MAX_DIM * SSMQ_MAX_DIM];
        for (int i = 0; i < D; ++i) xm[i] = 0.0;
        for (int i = 0; i < D * D; ++i) xP[i] = 0.0;
        for (int j = 0; j < NP; ++j) {
            const double w = a.uwts[j];
            for (int i = 0; i < D; ++i) xm[i] += res.m(i, j) * w;
            for (int i = 0; i < D * D; ++i) xP[i] += res.P(i, j) * w;
        }
        for (int i = 0; i < D; ++i) ok = ok && __builtin_isfinite(xm[i]);
        for (int i = 0; i < D * D; ++i) ok = ok && __builtin_isfinite(xP[i]);
        if (!ok) {                               // where forward_pass raises LinAlgError for this trajectory
            t.mode = 2;
            a.failed[b] = why(t.k, items_ok ? WHY_MIXTURE_NOT_FINITE : WHY_MIXTURE_ITEM);
            return;
        }
        for (int i = 0; i < D; ++i) {
            t.xm[i] = xm[i];
            a.fm[((size_t)b * T + (t.k - 1)) * D + i] = xm[i];
        }
        for (int i = 0; i < D * D; ++i) {
            t.xP[i] = xP[i];
            a.fP[((size_t)b * T + (t.k - 1)) * D * D + i] = xP[i];
        }
        if (t.k == T) {
            t.mode = 2;
        } else {
            ++t.k;
            mg_begin_step(t, a, b);
        }
    }
}

template <int PM>
__global__ void k_mg_finish(const MgArgs a, double *theta_last, double *pcov_last) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const TrajD<PM> &t = ((const TrajD<PM> *)a.traj)[b];
    for (int i = 0; i < a.P; ++i) theta_last[(size_t)b * a.P + i] = t.pm[i];
    for (int i = 0; i < a.P * a.P; ++i) pcov_last[(size_t)b * a.P * a.P + i] = t.pc[i];
}

// ---- the whole filter in ONE launch (small systems) --------------------------------------------------------------------------------
// Trajectories never interact, and with the theta step of an item a per-lane device function (ssmq_theta_item.h) nothing in a
// round needs another kernel: a group of PER = max(P + 1, NP) lanes owns a trajectory, every lane of the group evaluates ONE of
// the points the trajectory waits for, the group's first lane takes the values and advances the trajectory's state machine, and
// the wave (64 / PER trajectories) loops until all of its trajectories are through their T steps or have failed.  No scan, no
// packing, no kernel boundary and no host between two evaluations; the exit condition is per wave and every path of the state
// machine is bounded (BFGS: 200 P iterations of at most 100 + 10 + 10 line-search evaluations).  Item inputs and results cross
// lanes through a few hundred bytes of LDS per trajectory.  Same arithmetic as the rounds route - same device functions, same
// exp / log - so the two agree bit for bit (tests/test_gpu_parity.py::test_marginal_filter_one_launch_matches_device_rounds).
struct MgItem {
    int32_t fid_dyn, fid_obs, emv_dyn, emv_obs;
    FPar fpd, fpo;
    double jitter;
};

template <int TPW, int PER, int D>
struct LdsResults {
    const double (*ll_)[PER];
    const double (*m_)[PER][D];
    const double (*P_)[PER][D * D];
    const int32_t (*st_)[PER];
    int g;
    __device__ __forceinline__ double ll(int j) const { return ll_[g][j]; }
    __device__ __forceinline__ int32_t st(int j) const { return st_[g][j]; }
    __device__ __forceinline__ double m(int i, int j) const { return m_[g][j][i]; }
    __device__ __forceinline__ double P(int i, int j) const { return P_[g][j][i]; }
};

template <int PX, int DIN, int D, int Y, int ND, int NO>
__global__ __launch_bounds__(64) void k_mg_persistent(const MgArgs a, const MgItem it) {
    constexpr int PER = 2 * PX, TPW = 64 / PER, PM = PX, Pd = DIN + 1, dq = DIN - D;
    __shared__ int32_t s_n[TPW];
    __shared__ double o_ll[TPW][PER], o_m[TPW][PER][D], o_P[TPW][PER][D * D];
    __shared__ int32_t o_st[TPW][PER];
    const int lane = threadIdx.x, g = lane / PER, j = lane - g * PER;
    const int64_t b = (int64_t)blockIdx.x * TPW + g;
    const bool in_group = g < TPW;
    const bool member = in_group && b < a.B;
    const bool leader = member && j == 0;
    // The trajectories' states live in LDS for the length of the kernel (3 KB each): the optimiser reads and writes its state
    // every round - on the arena's 3 KB-strided structs each of those accesses is an L2 round trip on the wave's critical path -
    // and the lanes of the group read the point they are to evaluate straight from it.
    static_assert(sizeof(TrajD<PM>) % sizeof(double) == 0, "TrajD: a whole number of doubles");
    __shared__ double s_traj[TPW][sizeof(TrajD<PM>) / sizeof(double)];      // (raw: the struct has member initialisers)
    TrajD<PM> *tp = reinterpret_cast<TrajD<PM> *>(s_traj[in_group ? g : 0]);
    if (leader) {
        TrajD<PM> &t = *tp;
        t.k = 1;
        for (int i = 0; i < D; ++i) t.xm[i] = a.x0_mean[i];
        for (int i = 0; i < D * D; ++i) t.xP[i] = a.x0_cov[i];
        for (int i = 0; i < PX; ++i) t.pm[i] = a.prior_mean[i];
        for (int i = 0; i < PX * PX; ++i) t.pc[i] = a.prior_cov[i];
        a.failed[b] = 0;
        mg_begin_step(t, a, b);
    }
    if (in_group && j == 0) s_n[g] = leader ? (tp->mode == 0 ? PX + 1 : (tp->mode == 1 ? a.NP : 0)) : 0;
    __syncthreads();
    int32_t rounds = 0;
    unsigned long long items = 0;
    for (;;) {
        int any = 0;
#pragma unroll
        for (int gg = 0; gg < TPW; ++gg) any += s_n[gg];
        if (any == 0) break;                                   // (the same for every lane of the wave)
        ++rounds;
        items += (unsigned long long)any;
        // ---- one point per lane: the theta-conditioned filter step ------------------------------------------------------------------
        if (member && j < s_n[g]) {
            const TrajD<PM> &t = *tp;
            const int mode = t.mode;
            double par_d[Pd], par_o[D + 1], m[DIN], cv[DIN][DIN], yv[Y], m_fi[D], P_fi[D][D], ll;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const double row = mode == 0 ? t.run.xt[i] + ((j == i + 1) ? a.fd_step : 0.0) : t.pts[(size_t)j * PX + i];
                const double e = exp(row);                     // the kernel parameters are exp(theta)
                if (i < Pd) par_d[i] = e;
                else par_o[i - Pd] = e;
            }
            // [mean; q_mean], blockdiag(cov, Q) for dynamics that take their noise as an argument (ssinf.py:1174-1176)
#pragma unroll
            for (int i = 0; i < DIN; ++i)
#pragma unroll
                for (int k = 0; k < DIN; ++k) cv[i][k] = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                m[i] = t.xm[i];
#pragma unroll
                for (int k = 0; k < D; ++k) cv[i][k] = t.xP[i * D + k];
            }
#pragma unroll
            for (int i = 0; i < dq; ++i) {
                m[D + i] = a.q_mean[i];
#pragma unroll
                for (int k = 0; k < dq; ++k) cv[D + i][D + k] = a.q_cov[i * dq + k];
            }
#pragma unroll
            for (int i = 0; i < Y; ++i) yv[i] = a.y[((size_t)b * a.T + (t.k - 1)) * Y + i];
            const int32_t st = theta_item::theta_item_core<DIN, D, Y, ND, NO>(it.fid_dyn, it.fid_obs, it.fpd, it.fpo, it.emv_dyn, it.emv_obs,
                                                                            a.th.xid, a.th.xio, par_d, par_o, m, cv, yv, (double)t.k, a.th.gq,
                                                                            a.th.rr, it.jitter, m_fi, P_fi, ll);
            o_ll[g][j] = ll;
            o_st[g][j] = st;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                o_m[g][j][d] = m_fi[d];
#pragma unroll
                for (int d2 = 0; d2 < D; ++d2) o_P[g][j][d * D + d2] = P_fi[d][d2];
            }
        }
        __syncthreads();
        // ---- the group's first lane advances: optimiser step, or Laplace posterior and its sigma points, or the mixture and the next
        // time step - and says what the trajectory waits for next
        if (leader && s_n[g] > 0) {
            mg_advance_one<PM, PX, LdsResults<TPW, PER, D>, true>(*tp, a, b, LdsResults<TPW, PER, D>{o_ll, o_m, o_P, o_st, g});
            s_n[g] = tp->mode == 0 ? PX + 1 : (tp->mode == 1 ? a.NP : 0);
        }
        __syncthreads();
    }
    if (leader) {                                              // what k_mg_finish reads: the last step's parameter posterior
        TrajD<PM> &o = ((TrajD<PM> *)a.traj)[b];
        for (int i = 0; i < PX; ++i) o.pm[i] = tp->pm[i];
        for (int i = 0; i < PX * PX; ++i) o.pc[i] = tp->pc[i];
    }
    if (lane == 0) {
        atomicMax(&a.count[2], rounds);
        atomicAdd(&a.totals[0], items);
    }
}

typedef void (*mg_persistent_kernel)(const MgArgs, const MgItem);
struct MgPersistentEntry {
    int P, Din, D, Y, Nd, No;
    mg_persistent_kernel k;
};
#define SSMQ_MGP(PX, DIN, D, Y, ND, NO) {PX, DIN, D, Y, ND, NO, &k_mg_persistent<PX, DIN, D, Y, ND, NO>}
// the shapes of k_theta_item (ssmq_theta_item.hip): P = Din + D + 2 log-parameters
const MgPersistentEntry kMgPersistent[] = {
    SSMQ_MGP(4, 1, 1, 1, 2, 2), SSMQ_MGP(4, 1, 1, 1, 3, 3), SSMQ_MGP(5, 2, 1, 1, 4, 2), SSMQ_MGP(5, 2, 1, 1, 5, 3), SSMQ_MGP(6, 2, 2, 1, 4, 4),
    SSMQ_MGP(6, 2, 2, 1, 5, 5),
};

// Returns SSMQ_OK having produced everything, SSMQ_E_UNSUPPORTED if this shape has no device-resident route (the caller then runs
// the host rounds), or an error.
template <int PM, int PX>
int marginal_filter_batch_device(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs, const ssmq_integrand *f_obs,
                                 int64_t B, int T, double jitter, const double *y, const double *x0_mean, const double *x0_cov,
                                 const double *q_mean, const double *q_cov, const double *GQG, const double *R, const double *prior_mean,
                                 const double *prior_cov, const double *upts, const double *uwts, int NP, double fd_step,
                                 double param_jitter, double *fm, double *fP, int32_t *failed, double *theta_last, double *pcov_last,
                                 int64_t *stats) {
    const int Din = h_dyn->D, D = h_dyn->E, Y = h_obs->E, dq = Din - D;
    const int Pd = Din + 1, Po = h_obs->D + 1, P = Pd + Po;
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = stream();
    const int per = std::max(P + 1, NP);
    const int64_t cap = B * per;
    if (cap > 0x7fffffff / 2) return SSMQ_E_UNSUPPORTED;
    // one arena: theta step | trajectory states | offsets and counters | inputs | outputs
    auto al = [](size_t n) { return (n + 255) / 256 * 256; };
    const size_t th_bytes = theta_dev_bytes(h_dyn, h_obs, cap);
    const size_t n_fm = (size_t)B * T * D, n_fP = (size_t)B * T * D * D;
    const size_t statics = (size_t)D + (size_t)D * D + P + (size_t)P * P + dq + (size_t)dq * dq + (size_t)P * NP + NP;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_th = take(th_bytes), o_tr = take(sizeof(TrajD<PM>) * (size_t)B), o_first = take(sizeof(int32_t) * (size_t)B), o_modes = take((size_t)B),
                 o_count = take(sizeof(int32_t) * 4), o_tot = take(sizeof(unsigned long long) * 8), o_y = take(sizeof(double) * (size_t)B * T * Y),
                 o_st = take(sizeof(double) * statics), o_fm = take(sizeof(double) * n_fm), o_fP = take(sizeof(double) * n_fP),
                 o_failed = take(sizeof(int32_t) * (size_t)B), o_tl = take(sizeof(double) * (size_t)B * P),
                 o_pl = take(sizeof(double) * (size_t)B * P * P);
    char *dev = nullptr;
    SSMQ_HIP(hipMalloc((void **)&dev, off));
    struct Free { char *p; ~Free() { if (p) hipFree(p); } } guard{dev};
    MgArgs a;
    memset(&a, 0, sizeof(a));
    theta_dev_carve(a.th, h_dyn, h_obs, cap, dev + o_th);
    a.traj = dev + o_tr; a.B = B; a.T = T; a.P = P; a.Pd = Pd; a.Po = Po; a.NP = NP; a.D = D; a.Din = Din; a.Y = Y; a.dq = dq;
    a.first = (int32_t *)(dev + o_first); a.modes = (signed char *)(dev + o_modes); a.count = (int32_t *)(dev + o_count); a.totals = (unsigned long long *)(dev + o_tot);
    a.y = (const double *)(dev + o_y);
    a.fd_step = fd_step; a.param_jitter = param_jitter;
    a.fm = (double *)(dev + o_fm); a.fP = (double *)(dev + o_fP); a.failed = (int32_t *)(dev + o_failed);
    // what the trajectories share: one host block, one copy
    std::vector<double> hs(statics);
    {
        double *h = hs.data(), *d = (double *)(dev + o_st);
        auto put = [&](const double *src, size_t n, const double **dst) {
            if (n) std::memcpy(h, src, sizeof(double) * n);
            *dst = d;
            h += n; d += n;
        };
        put(x0_mean, D, &a.x0_mean); put(x0_cov, (size_t)D * D, &a.x0_cov); put(prior_mean, P, &a.prior_mean);
        put(prior_cov, (size_t)P * P, &a.prior_cov); put(q_mean, dq, &a.q_mean); put(q_cov, (size_t)dq * dq, &a.q_cov);
        put(upts, (size_t)P * NP, &a.upts); put(uwts, NP, &a.uwts);
    }
    SSMQ_HIP(hipMemcpyAsync(dev + o_st, hs.data(), sizeof(double) * statics, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemcpyAsync(dev + o_y, y, sizeof(double) * (size_t)B * T * Y, hipMemcpyHostToDevice, s));
    SSMQ_HIP(hipMemsetAsync(dev + o_count, 0, sizeof(int32_t) * 4, s));
    SSMQ_HIP(hipMemsetAsync(dev + o_tot, 0, sizeof(unsigned long long) * 8, s));
    SSMQ_HIP(hipMemsetAsync(dev + o_fm, 0xff, sizeof(double) * (n_fm + 0), s));      // all-ones bit pattern: a NaN
    SSMQ_HIP(hipMemsetAsync(dev + o_fP, 0xff, sizeof(double) * n_fP, s));
    if ((rc = theta_dev_upload_static(a.th, h_dyn, h_obs, GQG, R, s))) return rc;
    const unsigned tb = 64, tg = (unsigned)((B + tb - 1) / tb);
    Ctx &cx = ctx();                                // 64 bytes of pinned, device-visible host memory, kept with the thread's context
    if (!cx.pinned_flags) SSMQ_HIP(hipHostMalloc(&cx.pinned_flags, 64, hipHostMallocPortable | hipHostMallocMapped));
    volatile int32_t *hf = (volatile int32_t *)cx.pinned_flags;
    hf[0] = (int32_t)std::min<int64_t>(B, 0x7fffffff);
    hf[1] = 0;
    a.hflag = hf;
    // the whole filter in one launch where the item step is a per-lane device function (k_mg_persistent); SSMQ_MARGINAL_ROUNDS=1
    // keeps the rounds below (the route of every other shape)
    int32_t hc[4] = {0, 0, 0, 0};
    const MgPersistentEntry *pe = nullptr;
    if (!ssmq::sw("SSMQ_MARGINAL_ROUNDS") && !ssmq::sw("SSMQ_NO_THETA_ITEM") && NP == 2 * P && PX == P)
        for (const MgPersistentEntry &e : kMgPersistent)
            if (e.P == P && e.Din == Din && e.D == D && e.Y == Y && e.Nd == h_dyn->N && e.No == h_obs->N) pe = &e;
    if (pe) {
        MgItem it;
        memset(&it, 0, sizeof(it));
        it.fid_dyn = f_dyn->id; it.fid_obs = f_obs->id; it.emv_dyn = h_dyn->emv_mode; it.emv_obs = h_obs->emv_mode; it.jitter = jitter;
        fill_fpar(f_dyn, &it.fpd);
        fill_fpar(f_obs, &it.fpo);
        const int tpw = 64 / (2 * P);
        hipLaunchKernelGGL(pe->k, dim3((unsigned)((B + tpw - 1) / tpw)), dim3(64), 0, s, a, it);
        if ((rc = hip_fail(hipGetLastError(), "k_mg_persistent"))) return rc;
    } else {
    hipLaunchKernelGGL(k_mg_init<PM>, dim3(tg), dim3(tb), 0, s, a);
    const bool fused_scan = B <= 8192 && per <= 32 && !ssmq::sw("SSMQ_MARGINAL_SCAN_KERNEL");
    const int64_t tpb_fill = 256 / per;
    // Rounds are queued kRoundsAhead ahead of the scan count the device reports through pinned host memory; nothing in this loop
    // waits for the device (round 5's first version synchronised every eighth round: a bubble of a copy and a launch each time).
    // The number of unfinished trajectories only falls, so the latest value seen bounds the grids of every round queued after it.
    int64_t launched = 0;
    int32_t last_seen = -1;
    auto last_progress = std::chrono::steady_clock::now();
    for (;;) {
        const int32_t seen = hf[1];
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const int32_t unfinished = hf[0];
        if (seen > 0 && unfinished == 0) break;               // a scan found every trajectory done: what is queued finds nothing to do
        if (launched - seen >= kRoundsAhead) {
            // every wait in this library has an end: a device that reports no scan for a minute is asked for its error
            if (seen != last_seen) {
                last_seen = seen;
                last_progress = std::chrono::steady_clock::now();
            } else if (std::chrono::steady_clock::now() - last_progress > std::chrono::seconds(60)) {
                SSMQ_HIP(hipStreamSynchronize(s));
                if (hf[1] == seen) {
                    set_error("marginal_filter_batch: the device rounds made no progress");
                    return SSMQ_E_HIP;
                }
            }
            std::this_thread::yield();
            continue;
        }
        const int64_t bound = std::max<int64_t>(1, unfinished) * per;
        if (fused_scan) {
            hipLaunchKernelGGL((k_mg_fill<PM, true>), dim3((unsigned)((B + tpb_fill - 1) / tpb_fill)), dim3(256), 0, s, a, per);
        } else {
            hipLaunchKernelGGL(k_mg_scan<PM>, dim3(1), dim3(256), 0, s, a);
            hipLaunchKernelGGL((k_mg_fill<PM, false>), dim3((unsigned)((B * per + 255) / 256)), dim3(256), 0, s, a, per);
        }
        if ((rc = theta_dev_enqueue(a.th, h_dyn, f_dyn, h_obs, f_obs, jitter, bound, a.count, s))) return rc;
        hipLaunchKernelGGL((k_mg_advance<PM, PX>), dim3((unsigned)((B + kAdvPerWave - 1) / kAdvPerWave)), dim3(64), 0, s, a);
        if ((rc = hip_fail(hipGetLastError(), "marginal filter: device rounds"))) return rc;
        ++launched;
    }
    }   // rounds route
    hipLaunchKernelGGL(k_mg_finish<PM>, dim3(tg), dim3(tb), 0, s, a, (double *)(dev + o_tl), (double *)(dev + o_pl));
    SSMQ_HIP(hipMemcpyAsync(fm, a.fm, sizeof(double) * n_fm, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipMemcpyAsync(fP, a.fP, sizeof(double) * n_fP, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipMemcpyAsync(failed, a.failed, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    if (theta_last) SSMQ_HIP(hipMemcpyAsync(theta_last, dev + o_tl, sizeof(double) * (size_t)B * P, hipMemcpyDeviceToHost, s));
    if (pcov_last) SSMQ_HIP(hipMemcpyAsync(pcov_last, dev + o_pl, sizeof(double) * (size_t)B * P * P, hipMemcpyDeviceToHost, s));
    unsigned long long tot[8] = {0, 0};
    SSMQ_HIP(hipMemcpyAsync(hc, a.count, sizeof(hc), hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipMemcpyAsync(tot, a.totals, sizeof(tot), hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    if (stats) {
        stats[0] = hc[2]; stats[1] = (int64_t)tot[1]; stats[2] = (int64_t)tot[0];
    }
    return SSMQ_OK;
}

}  // namespace

extern "C" int ssmq_gp_marginal_filter_batch(ssmq_transform *h_dyn, const ssmq_integrand *f_dyn, ssmq_transform *h_obs,
                                             const ssmq_integrand *f_obs, int64_t B, int T, double jitter, const double *y,
                                             const double *x0_mean, const double *x0_cov, const double *q_mean,
                                             const double *q_cov, const double *GQG, const double *R, const double *prior_mean,
                                             const double *prior_cov, const double *upts, const double *uwts, int NP,
                                             double fd_step, double param_jitter, double *fm, double *fP, int32_t *failed,
                                             double *theta_last, double *pcov_last, int64_t *stats) {
    SSMQ_HANDLE_LOCK(h_dyn, h_obs);
    if (!h_dyn || !h_obs || !f_dyn || !f_obs || B < 0 || T < 0 || (B > 0 && T > 0 && (!y || !x0_mean || !x0_cov || !prior_mean ||
        !prior_cov || !upts || !uwts || !fm || !fP || !failed))) {
        set_error("marginal_filter_batch: null argument");
        return SSMQ_E_ARG;
    }
    const int Din = h_dyn->D, D = h_dyn->E, Y = h_obs->E, dq = Din - D;
    const int Pd = Din + 1, Po = h_obs->D + 1, P = Pd + Po;
    if (P > kMaxPar || NP < 1 || NP > 2 * kMaxPar || dq < 0 || (dq > 0 && (!q_mean || !q_cov))) {
        set_error("marginal_filter_batch: bad shape (parameters, parameter points, or noise moments of augmented dynamics missing)");
        return SSMQ_E_ARG;
    }
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    if (B == 0 || T == 0) return SSMQ_OK;
    // the state machines on the device where the theta step has its two-launch route and the parameter count an instantiation
    // (SSMQ_MARGINAL_HOST_ROUNDS=1: the host rounds below, round 4's route, kept as the second implementation of the same filter)
    if (!ssmq::sw("SSMQ_MARGINAL_HOST_ROUNDS") && P <= 16 && NP <= 32 && theta_dev_supported(h_dyn, f_dyn, h_obs, f_obs)) {
#define SSMQ_MG_DEV(PM, PX) marginal_filter_batch_device<PM, PX>(h_dyn, f_dyn, h_obs, f_obs, B, T, jitter, y, x0_mean, x0_cov, q_mean, q_cov, GQG, R, \
                                                              prior_mean, prior_cov, upts, uwts, NP, fd_step, param_jitter, fm, fP, failed,      \
                                                              theta_last, pcov_last, stats)
        // (P = D + Din + 2: 4 scalar state, 5 scalar state with its noise as an argument, 6 / 8 two / three states)
        const int rc = P == 4 ? SSMQ_MG_DEV(4, 4) : P == 5 ? SSMQ_MG_DEV(5, 5) : P == 6 ? SSMQ_MG_DEV(6, 6) : P == 8 ? SSMQ_MG_DEV(8, 8)
                                                                                                       : SSMQ_MG_DEV(16, 0);
#undef SSMQ_MG_DEV
        if (rc != SSMQ_E_UNSUPPORTED) return rc;
    }
    const double nan = std::numeric_limits<double>::quiet_NaN();
    std::vector<Traj> tr((size_t)B);
    for (int64_t i = 0; i < (int64_t)B * T * D; ++i) fm[i] = nan;
    for (int64_t i = 0; i < (int64_t)B * T * D * D; ++i) fP[i] = nan;
    auto begin_step = [&](Traj &t, int64_t b) {        // the Laplace step of time step t.k starts from the prior (t.pm, t.pc)
        if (!chol_lower(t.pc, P, t.Lp, &t.logdet2)) {  // numpy.linalg.cholesky would raise in _param_log_prior
            t.mode = 2;
            failed[b] = why(t.k, WHY_PRIOR_NOT_PD);
            return;
        }
        bfgs_start(t.run, P, t.pm);
        t.mode = 0;
    };
    for (int64_t b = 0; b < B; ++b) {
        Traj &t = tr[b];
        std::memcpy(t.xm, x0_mean, sizeof(double) * D);
        std::memcpy(t.xP, x0_cov, sizeof(double) * D * D);
        std::memcpy(t.pm, prior_mean, sizeof(double) * P);
        std::memcpy(t.pc, prior_cov, sizeof(double) * P * P);
        failed[b] = 0;
        begin_step(t, b);
    }
    auto log_prior = [&](const Traj &t, const double *th) {
        double v[kMaxPar], q = 0.0;
        for (int i = 0; i < P; ++i) {
            double s = th[i] - t.pm[i];
            for (int k = 0; k < i; ++k) s -= t.Lp[i * P + k] * v[k];
            v[i] = s / t.Lp[i * P + i];
            q += v[i] * v[i];
        }
        return -0.5 * (q + t.logdet2 + P * std::log(2.0 * M_PI));
    };
    std::vector<int64_t> who, first;
    std::vector<double> rows, pd, po, mm, cc, yy, tt, ll, om, oc;
    std::vector<int32_t> st;
    const double inf = std::numeric_limits<double>::infinity();
    int64_t rounds = 0, iters = 0, total_items = 0;
    // worker threads: SSMQ_MARGINAL_THREADS (0 = none), else up to 8 and never more than the trajectories could use
    int n_workers = 7;
    if (const char *e = ssmq::sw("SSMQ_MARGINAL_THREADS")) n_workers = std::max(0, atoi(e) - 1);
    n_workers = (int)std::min<int64_t>(std::min<unsigned>((unsigned)n_workers, std::max(1u, std::thread::hardware_concurrency()) - 1), B / 256);
    Workers pool(n_workers);
    const bool timing = ssmq::sw("SSMQ_MARGINAL_TIMING") != nullptr;      // host-side budget of the rounds, printed at the end
    double t_pack = 0.0, t_call = 0.0, t_adv = 0.0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
        const double t0 = timing ? now() : 0.0;
        who.clear(); first.clear();
        int64_t items = 0;
        for (int64_t b = 0; b < B; ++b)
            if (tr[b].mode != 2) {
                who.push_back(b);
                first.push_back(items);
                items += tr[b].mode == 0 ? P + 1 : NP;
            }
        if (who.empty()) break;
        rows.resize((size_t)items * P); pd.resize((size_t)items * Pd); po.resize((size_t)items * Po);
        mm.resize((size_t)items * Din); cc.assign((size_t)items * Din * Din, 0.0); yy.resize((size_t)items * Y); tt.resize((size_t)items);
        ll.resize((size_t)items); om.resize((size_t)items * D); oc.resize((size_t)items * D * D); st.assign((size_t)items, 0);
        pool.run(who.size(), [&](size_t w0, size_t w1) {
        for (size_t w = w0; w < w1; ++w) {
            const int64_t b = who[w];
            const Traj &t = tr[b];
            const int n = t.mode == 0 ? P + 1 : NP;
            for (int j = 0; j < n; ++j) {
                const int64_t it = first[w] + j;
                double *row = &rows[(size_t)it * P];
                if (t.mode == 0)
                    for (int i = 0; i < P; ++i) row[i] = t.run.xt[i] + ((j == i + 1) ? fd_step : 0.0);
                else
                    for (int i = 0; i < P; ++i) row[i] = t.pts[(size_t)j * P + i];
                for (int i = 0; i < P; ++i) {
                    const double e = std::exp(row[i]);            // the kernel parameters are exp(theta)
                    if (i < Pd) pd[(size_t)it * Pd + i] = e;
                    else po[(size_t)it * Po + (i - Pd)] = e;
                }
                // [mean; q_mean], blockdiag(cov, Q) for dynamics that take their noise as an argument (ssinf.py:1174-1176)
                double *m = &mm[(size_t)it * Din], *c = &cc[(size_t)it * Din * Din];
                for (int i = 0; i < D; ++i) {
                    m[i] = t.xm[i];
                    for (int k = 0; k < D; ++k) c[i * Din + k] = t.xP[i * D + k];
                }
                for (int i = 0; i < dq; ++i) {
                    m[D + i] = q_mean[i];
                    for (int k = 0; k < dq; ++k) c[(D + i) * Din + D + k] = q_cov[i * dq + k];
                }
                std::memcpy(&yy[(size_t)it * Y], y + ((size_t)b * T + (t.k - 1)) * Y, sizeof(double) * Y);
                tt[(size_t)it] = (double)t.k;
            }
        }
        });
        const double t1 = timing ? now() : 0.0;
        const int rc = ssmq_gp_theta_step_times(h_dyn, f_dyn, h_obs, f_obs, items, pd.data(), po.data(), jitter, mm.data(), cc.data(), 0,
                                                yy.data(), 0, tt.data(), GQG, R, om.data(), oc.data(), ll.data(), st.data());
        if (rc < 0) return rc;
        const double t2 = timing ? now() : 0.0;
        ++rounds;
        total_items += items;
        std::atomic<int64_t> iters_round{0};
        pool.run(who.size(), [&](size_t w0, size_t w1) {
        int64_t iters_mine = 0;
        for (size_t w = w0; w < w1; ++w) {
            const int64_t b = who[w];
            Traj &t = tr[b];
            if (t.mode == 0) {
                double vals[kMaxPar + 1];
                for (int j = 0; j <= P; ++j) {
                    const int64_t it = first[w] + j;
                    const double v = -ll[(size_t)it] - log_prior(t, &rows[(size_t)it * P]);
                    vals[j] = std::isfinite(v) ? v : inf;
                }
                bfgs_advance(t.run, P, fd_step, vals);
                if (t.run.phase != PH_DONE) continue;
                iters_mine += t.run.k;
                // Laplace posterior (ssinf.py:1272-1273) and its sigma points (:1103-1106)
                double pcn[kMaxPar * kMaxPar], L[kMaxPar * kMaxPar];
                bool fin = true;
                for (int i = 0; i < P; ++i) {
                    t.pm[i] = t.run.x[i];
                    fin = fin && std::isfinite(t.pm[i]);
                    for (int k = 0; k < P; ++k) {
                        pcn[i * P + k] = t.run.H[i * P + k] + (i == k ? param_jitter : 0.0);
                        fin = fin && std::isfinite(pcn[i * P + k]);
                    }
                }
                if (!fin || !chol_lower(pcn, P, L, nullptr)) {
                    t.mode = 2;
                    failed[b] = why(t.k, fin ? WHY_LAPLACE_NOT_PD : WHY_LAPLACE_NOT_FINITE);
                    continue;
                }
                std::memcpy(t.pc, pcn, sizeof(double) * P * P);
                for (int j = 0; j < NP; ++j)
                    for (int i = 0; i < P; ++i) {
                        double s = t.pm[i];
                        for (int k = 0; k <= i; ++k) s += L[i * P + k] * upts[(size_t)k * NP + j];
                        t.pts[(size_t)j * P + i] = s;
                    }
                t.mode = 1;
            } else {
                // mixture over the parameter points (ssinf.py:1108-1115): plain weighted sums of the conditional moments
                bool ok = true;
                for (int j = 0; j < NP; ++j) ok = ok && st[(size_t)(first[w] + j)] == 0;
                const bool items_ok = ok;
                double xm[SSMQ_MAX_DIM], xP[SSMQ_MAX_DIM * SSMQ_MAX_DIM];
                for (int i = 0; i < D; ++i) xm[i] = 0.0;
                for (int i = 0; i < D * D; ++i) xP[i] = 0.0;
                for (int j = 0; j < NP; ++j) {
                    const int64_t it = first[w] + j;
                    for (int i = 0; i < D; ++i) xm[i] += om[(size_t)it * D + i] * uwts[j];
                    for (int i = 0; i < D * D; ++i) xP[i] += oc[(size_t)it * D * D + i] * uwts[j];
                }
                for (int i = 0; i < D; ++i) ok = ok && std::isfinite(xm[i]);
                for (int i = 0; i < D * D; ++i) ok = ok && std::isfinite(xP[i]);
                if (!ok) {                               // where forward_pass raises LinAlgError for this trajectory
                    t.mode = 2;
                    failed[b] = why(t.k, items_ok ? WHY_MIXTURE_NOT_FINITE : WHY_MIXTURE_ITEM);
                    continue;
                }
                std::memcpy(t.xm, xm, sizeof(double) * D);
                std::memcpy(t.xP, xP, sizeof(double) * D * D);
                std::memcpy(fm + ((size_t)b * T + (t.k - 1)) * D, xm, sizeof(double) * D);
                std::memcpy(fP + ((size_t)b * T + (t.k - 1)) * D * D, xP, sizeof(double) * D * D);
                if (t.k == T) {
                    t.mode = 2;
                } else {
                    ++t.k;
                    begin_step(t, b);
                }
            }
        }
        iters_round += iters_mine;
        });
        iters += iters_round.load();
        if (timing) {
            const double t3 = now();
            t_pack += t1 - t0; t_call += t2 - t1; t_adv += t3 - t2;
        }
    }
    if (timing)
        fprintf(stderr, "marginal_filter_batch: %lld rounds, %lld items: pack %.2f ms, theta step %.2f ms, optimiser / mixtures %.2f ms\n",
                (long long)rounds, (long long)total_items, 1e3 * t_pack, 1e3 * t_call, 1e3 * t_adv);
    for (int64_t b = 0; b < B; ++b) {
        if (theta_last) std::memcpy(theta_last + (size_t)b * P, tr[b].pm, sizeof(double) * P);
        if (pcov_last) std::memcpy(pcov_last + (size_t)b * P * P, tr[b].pc, sizeof(double) * P * P);
    }
    if (stats) {
        stats[0] = rounds; stats[1] = iters; stats[2] = total_items;
    }
    return SSMQ_OK;
}
