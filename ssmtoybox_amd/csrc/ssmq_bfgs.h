// SciPy's BFGS (scipy.optimize._minimize_bfgs with both of its line searches) as a per-run state machine that takes the
// objective values of its pending point a ROUND later - the numeric core of the batched Laplace step of the marginalised filter
// (ssmq_marginal.hip has the reference citations and the drivers).  One definition for the host rounds (ssmq_bfgs_lockstep_host,
// ssmq_gp_marginal_laplace_batch, the host-round route of ssmq_gp_marginal_filter_batch; pinned against SciPy in
// tests/test_bfgs_lockstep.py) and for the device-resident rounds (ssmq_marginal_dev.hip: one thread per trajectory).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/ssmq.h"

#define SSMQ_BFGS_HD __host__ __device__

namespace ssmq_bfgs {

constexpr int kMaxPar = 2 * (SSMQ_MAX_DIM + 1);

struct Dcsrch {            // scipy/optimize/_dcsrch.py: class DCSRCH (state), _iterate
    int stage = 0;
    bool brackt = false;
    double ginit = 0, gtest = 0, gx = 0, gy = 0, finit = 0, fx = 0, fy = 0, stx = 0, sty = 0, stmin = 0, stmax = 0, width = 0, width1 = 0;
    double ftol = 1e-4, gtol = 0.9, xtol = 1e-14, stpmin = 1e-100, stpmax = 1e100;
};
enum Task { T_START, T_FG, T_CONV, T_WARN, T_ERROR };

SSMQ_BFGS_HD inline double sgn(double v) { return (v > 0) - (v < 0); }
SSMQ_BFGS_HD inline double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// _dcsrch.py: dcstep
SSMQ_BFGS_HD inline void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp, double fp, double dp, bool &brackt,
            double stpmin, double stpmax) {
    const double sgnd = sgn(dp) * sgn(dx);
    double stpf, stpc, stpq;
    if (fp > fx) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = __builtin_fmax(__builtin_fabs(theta), __builtin_fmax(__builtin_fabs(dx), __builtin_fabs(dp)));
        double gamma = s * __builtin_sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma *= -1;
        const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
        stpc = stx + r * (stp - stx);
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        stpf = __builtin_fabs(stpc - stx) <= __builtin_fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (sgnd < 0.0) {
        const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = __builtin_fmax(__builtin_fabs(theta), __builtin_fmax(__builtin_fabs(dx), __builtin_fabs(dp)));
        double gamma = s * __builtin_sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma *= -1;
        const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
        stpc = stp + r * (stx - stp);
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        stpf = __builtin_fabs(stpc - stp) > __builtin_fabs(stpq - stp) ? stpc : stpq;
        brackt = true;
    } else if (__builtin_fabs(dp) < __builtin_fabs(dx)) {
        const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = __builtin_fmax(__builtin_fabs(theta), __builtin_fmax(__builtin_fabs(dx), __builtin_fabs(dp)));
        double gamma = s * __builtin_sqrt(__builtin_fmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
        if (r < 0 && gamma != 0) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            stpf = __builtin_fabs(stpc - stp) < __builtin_fabs(stpq - stp) ? stpc : stpq;
            if (stp > stx) stpf = __builtin_fmin(stp + 0.66 * (sty - stp), stpf);
            else stpf = __builtin_fmax(stp + 0.66 * (sty - stp), stpf);
        } else {
            stpf = __builtin_fabs(stpc - stp) > __builtin_fabs(stpq - stp) ? stpc : stpq;
            stpf = clipd(stpf, stpmin, stpmax);
        }
    } else {
        if (brackt) {
            const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            const double s = __builtin_fmax(__builtin_fabs(theta), __builtin_fmax(__builtin_fabs(dy), __builtin_fabs(dp)));
            double gamma = s * __builtin_sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
            stpc = stp + r * (sty - stp);
            stpf = stpc;
        } else if (stp > stx) {
            stpf = stpmax;
        } else {
            stpf = stpmin;
        }
    }
    if (fp > fx) {
        sty = stp; fy = fp; dy = dp;
    } else {
        if (sgnd < 0) {
            sty = stx; fy = fx; dy = dx;
        }
        stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
}

// _dcsrch.py: DCSRCH._iterate.  Returns the next task; stp is updated in place.
SSMQ_BFGS_HD inline Task dcsrch_iterate(Dcsrch &d, double &stp, double f, double g, Task task) {
    const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    if (task == T_START) {
        if (stp < d.stpmin || stp > d.stpmax || g >= 0) return T_ERROR;
        d.brackt = false;
        d.stage = 1;
        d.finit = f; d.ginit = g; d.gtest = d.ftol * d.ginit;
        d.width = d.stpmax - d.stpmin; d.width1 = d.width / p5;
        d.stx = 0.0; d.fx = d.finit; d.gx = d.ginit;
        d.sty = 0.0; d.fy = d.finit; d.gy = d.ginit;
        d.stmin = 0; d.stmax = stp + xtrapu * stp;
        return T_FG;
    }
    const double ftest = d.finit + stp * d.gtest;
    if (d.stage == 1 && f <= ftest && g >= 0) d.stage = 2;
    Task out = T_FG;
    if (d.brackt && (stp <= d.stmin || stp >= d.stmax)) out = T_WARN;
    if (d.brackt && d.stmax - d.stmin <= d.xtol * d.stmax) out = T_WARN;
    if (stp == d.stpmax && f <= ftest && g <= d.gtest) out = T_WARN;
    if (stp == d.stpmin && (f > ftest || g >= d.gtest)) out = T_WARN;
    if (f <= ftest && __builtin_fabs(g) <= d.gtol * -d.ginit) out = T_CONV;
    if (out == T_WARN || out == T_CONV) return out;
    if (d.stage == 1 && f <= d.fx && f > ftest) {
        const double fm = f - stp * d.gtest;
        double fxm = d.fx - d.stx * d.gtest, fym = d.fy - d.sty * d.gtest;
        const double gm = g - d.gtest;
        double gxm = d.gx - d.gtest, gym = d.gy - d.gtest;
        dcstep(d.stx, fxm, gxm, d.sty, fym, gym, stp, fm, gm, d.brackt, d.stmin, d.stmax);
        d.fx = fxm + d.stx * d.gtest; d.fy = fym + d.sty * d.gtest;
        d.gx = gxm + d.gtest; d.gy = gym + d.gtest;
    } else {
        dcstep(d.stx, d.fx, d.gx, d.sty, d.fy, d.gy, stp, f, g, d.brackt, d.stmin, d.stmax);
    }
    if (d.brackt) {
        if (__builtin_fabs(d.sty - d.stx) >= p66 * d.width1) stp = d.stx + p5 * (d.sty - d.stx);
        d.width1 = d.width;
        d.width = __builtin_fabs(d.sty - d.stx);
    }
    if (d.brackt) {
        d.stmin = __builtin_fmin(d.stx, d.sty);
        d.stmax = __builtin_fmax(d.stx, d.sty);
    } else {
        d.stmin = stp + xtrapl * (stp - d.stx);
        d.stmax = stp + xtrapu * (stp - d.stx);
    }
    stp = clipd(stp, d.stpmin, d.stpmax);
    if ((d.brackt && (stp <= d.stmin || stp >= d.stmax)) || (d.brackt && d.stmax - d.stmin <= d.xtol * d.stmax)) stp = d.stx;
    return T_FG;
}

// _linesearch.py: _cubicmin / _quadmin (None = false: an arithmetic error or a non-finite result)
SSMQ_BFGS_HD inline bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double *xmin) {
    const double C = fpa, db = b - a, dc = c - a;
    const double denom = (db * dc) * (db * dc) * (db - dc);
    const double r0 = fb - fa - C * db, r1 = fc - fa - C * dc;
    double A = dc * dc * r0 + -(db * db) * r1, Bq = -(dc * dc * dc) * r0 + db * db * db * r1;
    if (denom == 0.0 || !__builtin_isfinite(denom) || !__builtin_isfinite(A) || !__builtin_isfinite(Bq)) return false;
    A /= denom;
    Bq /= denom;
    const double radical = Bq * Bq - 3 * A * C;
    if (!(radical >= 0.0) || A == 0.0 || !__builtin_isfinite(radical)) return false;
    const double x = a + (-Bq + __builtin_sqrt(radical)) / (3 * A);
    if (!__builtin_isfinite(x)) return false;
    *xmin = x;
    return true;
}
SSMQ_BFGS_HD inline bool quadmin(double a, double fa, double fpa, double b, double fb, double *xmin) {
    const double D = fa, C = fpa, db = b - a * 1.0;
    if (db * db == 0.0) return false;
    const double Bq = (fb - D - C * db) / (db * db);
    if (Bq == 0.0 || !__builtin_isfinite(Bq)) return false;
    const double x = a - C / (2.0 * Bq);
    if (!__builtin_isfinite(x)) return false;
    *xmin = x;
    return true;
}

enum Phase { PH_INIT, PH_LINE, PH_LINE2, PH_DONE };

template <int PM>
struct RunT {            // _minimize_bfgs' locals of one trajectory; PM: compile-time bound on the number of parameters
    Phase phase = PH_INIT;
    int k = 0, ls_iter = 0, status = 0;
    double x[PM], g[PM], pk[PM], xt[PM], H[PM * PM];
    double old_fval = 0, old_old_fval = 0, derphi0 = 0, stp = 0;
    Dcsrch ls;
    Task task = T_START;
    // second line search (scipy _linesearch.py: scalar_search_wolfe2 / _zoom), entered where the first one gives up
    int w2_i = 0, z_i = 0;
    bool zoom = false;
    double w2_alpha0 = 0, w2_phi_a0 = 0, w2_derphi_a0 = 0;
    double a_lo = 0, a_hi = 0, phi_lo = 0, phi_hi = 0, derphi_lo = 0, phi_rec = 0, a_rec = 0;
};

template <int PM>
SSMQ_BFGS_HD inline __attribute__((always_inline)) void bfgs_advance(RunT<PM> &r, int P, double fd_step, const double *vals) {
    const double gtol = 1e-5, inf = __builtin_huge_val();
    const int maxiter = 200 * P, per = P + 1;
    // value and gradient at xt (non-finite -> +inf as the Python objective, ssmtoybox_amd/ssinf.py)
    double val[PM + 1], gt[PM];
    for (int j = 0; j < per; ++j) {
        const double v = vals[j];
        val[j] = __builtin_isfinite(v) ? v : inf;
    }
    for (int i = 0; i < P; ++i) gt[i] = (val[i + 1] - val[0]) / ((r.xt[i] + fd_step) - r.xt[i]);
    const double ft = val[0];
    bool start_iteration = false;
    if (r.phase == PH_INIT) {
        r.old_fval = ft;
        double n2 = 0.0, gmax = 0.0;
        for (int i = 0; i < P; ++i) {
            r.g[i] = gt[i];
            n2 += gt[i] * gt[i];
            gmax = (__builtin_isnan(gt[i]) || __builtin_isnan(gmax)) ? __builtin_nan("") : __builtin_fmax(gmax, __builtin_fabs(gt[i]));   // numpy's max keeps NaN
        }
        r.old_old_fval = r.old_fval + __builtin_sqrt(n2) / 2;
        if (!(gmax > gtol)) {            // (a NaN gradient ends the loop as in SciPy: `while gnorm > gtol`)
            r.phase = PH_DONE;
            r.status = (__builtin_isnan(gmax) || __builtin_isnan(ft)) ? SSMQ_BFGS_NAN : 0;
            return;
        }
        start_iteration = true;
    } else {                             // PH_LINE / PH_LINE2: a trial step has been evaluated
        double dphi = 0.0;
        for (int i = 0; i < P; ++i) dphi += gt[i] * r.pk[i];
        const double c1 = 1e-4, c2 = 0.9, amax = 1e100;
        const double phi0 = r.old_fval, derphi0 = r.derphi0;
        bool accepted = false, to_second = false, failed = false;
        double next = 0.0;                 // the next trial step, if neither
        if (r.phase == PH_LINE) {
            double stp = r.stp;
            const Task t = dcsrch_iterate(r.ls, stp, ft, dphi, T_FG);
            if (t == T_FG) {
                ++r.ls_iter;
                if (!__builtin_isfinite(stp) || r.ls_iter >= 100) to_second = true;
                else next = stp;
            } else if (t == T_CONV) {
                accepted = true;
            } else {                       // WARNING / ERROR: SciPy goes on with line_search_wolfe2
                to_second = true;
            }
        } else if (!r.zoom) {              // scalar_search_wolfe2, iteration w2_i, alpha1 = r.stp evaluated
            const double alpha1 = r.stp, phi_a1 = ft, derphi_a1 = dphi;
            auto start_zoom = [&](double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo) {
                r.zoom = true; r.z_i = 0;
                r.a_lo = a_lo; r.a_hi = a_hi; r.phi_lo = phi_lo; r.phi_hi = phi_hi; r.derphi_lo = derphi_lo;
                r.phi_rec = phi0; r.a_rec = 0.0;
            };
            if (r.w2_i >= 10) {            // for ... else: maxiter reached; the last evaluated step is returned
                accepted = true;
            } else if (alpha1 == 0.0) {
                failed = true;
            } else if ((phi_a1 > phi0 + c1 * alpha1 * derphi0) || ((phi_a1 >= r.w2_phi_a0) && r.w2_i > 0)) {
                start_zoom(r.w2_alpha0, alpha1, r.w2_phi_a0, phi_a1, r.w2_derphi_a0);
            } else if (__builtin_fabs(derphi_a1) <= -c2 * derphi0) {
                accepted = true;
            } else if (derphi_a1 >= 0) {
                start_zoom(alpha1, r.w2_alpha0, phi_a1, r.w2_phi_a0, derphi_a1);
            } else {
                const double alpha2 = __builtin_fmin(2 * alpha1, amax);
                r.w2_alpha0 = alpha1; r.w2_phi_a0 = phi_a1; r.w2_derphi_a0 = derphi_a1;
                ++r.w2_i;
                next = alpha2;
            }
        } else {                           // _zoom: a_j = r.stp evaluated
            const double a_j = r.stp, phi_aj = ft, derphi_aj = dphi;
            if ((phi_aj > phi0 + c1 * a_j * derphi0) || (phi_aj >= r.phi_lo)) {
                r.phi_rec = r.phi_hi; r.a_rec = r.a_hi; r.a_hi = a_j; r.phi_hi = phi_aj;
            } else {
                if (__builtin_fabs(derphi_aj) <= -c2 * derphi0) {
                    accepted = true;
                } else {
                    if (derphi_aj * (r.a_hi - r.a_lo) >= 0) {
                        r.phi_rec = r.phi_hi; r.a_rec = r.a_hi; r.a_hi = r.a_lo; r.phi_hi = r.phi_lo;
                    } else {
                        r.phi_rec = r.phi_lo; r.a_rec = r.a_lo;
                    }
                    r.a_lo = a_j; r.phi_lo = phi_aj; r.derphi_lo = derphi_aj;
                }
            }
            if (!accepted) {
                ++r.z_i;
                if (r.z_i > 10) failed = true;
            }
        }
        if (to_second) {
            // scalar_search_wolfe2 from the same point and direction: first trial step as for the first search
            double alpha1 = 1.0;
            if (derphi0 != 0) alpha1 = __builtin_fmin(1.0, 1.01 * 2 * (phi0 - r.old_old_fval) / derphi0);
            if (alpha1 < 0) alpha1 = 1.0;
            alpha1 = __builtin_fmin(alpha1, amax);
            r.phase = PH_LINE2;
            r.zoom = false; r.w2_i = 0;
            r.w2_alpha0 = 0.0; r.w2_phi_a0 = phi0; r.w2_derphi_a0 = derphi0;
            next = alpha1;
        }
        if (failed) {                      // _LineSearchError: "Desired error not necessarily achieved due to precision loss"
            r.phase = PH_DONE;
            r.status = SSMQ_BFGS_PRECISION_LOSS;
            return;
        }
        if (!accepted) {
            if (r.phase == PH_LINE2 && r.zoom) {
                // the next trial step of _zoom: cubic, else quadratic interpolation, else bisection
                const double dalpha = r.a_hi - r.a_lo;
                const double a = dalpha < 0 ? r.a_hi : r.a_lo, b = dalpha < 0 ? r.a_lo : r.a_hi;
                double a_j = 0.0;
                bool have = false;
                const double cchk = 0.2 * dalpha;
                if (r.z_i > 0) have = cubicmin(r.a_lo, r.phi_lo, r.derphi_lo, r.a_hi, r.phi_hi, r.a_rec, r.phi_rec, &a_j);
                if (r.z_i == 0 || !have || a_j > b - cchk || a_j < a + cchk) {
                    const double qchk = 0.1 * dalpha;
                    have = quadmin(r.a_lo, r.phi_lo, r.derphi_lo, r.a_hi, r.phi_hi, &a_j);
                    if (!have || a_j > b - qchk || a_j < a + qchk) a_j = r.a_lo + 0.5 * dalpha;
                }
                next = a_j;
            }
            r.stp = next;
            for (int i = 0; i < P; ++i) r.xt[i] = r.x[i] + next * r.pk[i];
            return;
        }
        // accepted: alpha_k = stp, the last evaluated step
        const double alpha = r.stp;
        double sk[PM], yk[PM], pn = 0.0, gmax = 0.0;
        for (int i = 0; i < P; ++i) {
            sk[i] = alpha * r.pk[i];
            r.x[i] = r.x[i] + sk[i];
            yk[i] = gt[i] - r.g[i];
            r.g[i] = gt[i];
            pn += r.pk[i] * r.pk[i];
            gmax = (__builtin_isnan(gt[i]) || __builtin_isnan(gmax)) ? __builtin_nan("") : __builtin_fmax(gmax, __builtin_fabs(gt[i]));
        }
        r.old_old_fval = r.old_fval;
        r.old_fval = ft;
        ++r.k;
        if (!(gmax > gtol) && !__builtin_isnan(gmax)) {
            r.phase = PH_DONE;
            r.status = 0;
            return;
        }
        if (alpha * __builtin_sqrt(pn) <= 0.0) {      // xrtol = 0
            r.phase = PH_DONE;
            r.status = __builtin_isnan(gmax) ? SSMQ_BFGS_NAN : 0;
            return;
        }
        if (!__builtin_isfinite(r.old_fval)) {
            r.phase = PH_DONE;
            r.status = SSMQ_BFGS_PRECISION_LOSS;
            return;
        }
        double rho_inv = 0.0;
        for (int i = 0; i < P; ++i) rho_inv += yk[i] * sk[i];
        const double rho = rho_inv == 0.0 ? 1000.0 : 1.0 / rho_inv;
        // Hk = (I - sk yk' rho) Hk (I - yk sk' rho) + rho sk sk'
        double A2[PM * PM], HA[PM * PM], Hn[PM * PM];
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) A2[i * P + j] = (i == j ? 1.0 : 0.0) - yk[i] * sk[j] * rho;
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) {
                double s = 0.0;
                for (int k = 0; k < P; ++k) s += r.H[i * P + k] * A2[k * P + j];
                HA[i * P + j] = s;
            }
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) {
                double s = 0.0;
                for (int k = 0; k < P; ++k) s += ((i == k ? 1.0 : 0.0) - sk[i] * yk[k] * rho) * HA[k * P + j];
                Hn[i * P + j] = s + rho * sk[i] * sk[j];
            }
        for (int i = 0; i < P * P; ++i) r.H[i] = Hn[i];
        if (__builtin_isnan(gmax)) {                  // `while gnorm > gtol` ends on NaN
            r.phase = PH_DONE;
            r.status = SSMQ_BFGS_NAN;
            return;
        }
        if (r.k >= maxiter) {
            r.phase = PH_DONE;
            r.status = SSMQ_BFGS_MAXITER;
            return;
        }
        start_iteration = true;
    }
    if (start_iteration) {
        // pk = -Hk gfk; scalar_search_wolfe1's first trial step; DCSRCH "START"
        double dphi0 = 0.0;
        for (int i = 0; i < P; ++i) {
            double s = 0.0;
            for (int j = 0; j < P; ++j) s += r.H[i * P + j] * r.g[j];
            r.pk[i] = -s;
        }
        for (int i = 0; i < P; ++i) dphi0 += r.g[i] * r.pk[i];
        r.derphi0 = dphi0;
        double alpha1 = 1.0;
        if (dphi0 != 0) {
            alpha1 = __builtin_fmin(1.0, 1.01 * 2 * (r.old_fval - r.old_old_fval) / dphi0);
            if (alpha1 < 0) alpha1 = 1.0;
        }
        r.ls = Dcsrch();
        double stp = alpha1;
        const Task t = dcsrch_iterate(r.ls, stp, r.old_fval, dphi0, T_START);
        if (t != T_FG || !__builtin_isfinite(stp)) {
            // the first search refuses to start (e.g. not a descent direction): scalar_search_wolfe2 from its first step
            r.phase = PH_LINE2;
            r.zoom = false; r.w2_i = 0;
            r.w2_alpha0 = 0.0; r.w2_phi_a0 = r.old_fval; r.w2_derphi_a0 = dphi0;
            stp = __builtin_fmin(alpha1, 1e100);
        } else {
            r.ls_iter = 1;
            r.phase = PH_LINE;
        }
        r.stp = stp;
        for (int i = 0; i < P; ++i) r.xt[i] = r.x[i] + stp * r.pk[i];
    }
}

template <int PM>
SSMQ_BFGS_HD void bfgs_start(RunT<PM> &r, int P, const double *x0) {
    r = RunT<PM>();
    for (int i = 0; i < P; ++i) r.x[i] = r.xt[i] = x0[i];
    for (int i = 0; i < P * P; ++i) r.H[i] = 0.0;
    for (int i = 0; i < P; ++i) r.H[i * P + i] = 1.0;
}

// lower Cholesky factor of a P x P matrix (row-major, pitch P); false where numpy.linalg.cholesky raises
SSMQ_BFGS_HD inline bool chol_lower(const double *C, int P, double *L, double *logdet2) {
    double ld = 0.0;
    for (int i = 0; i < P * P; ++i) L[i] = 0.0;
    for (int j = 0; j < P; ++j) {
        double s = C[j * P + j];
        for (int k = 0; k < j; ++k) s -= L[j * P + k] * L[j * P + k];
        if (!(s > 0.0)) return false;
        const double ljj = __builtin_sqrt(s);
        L[j * P + j] = ljj;
        ld += 2.0 * log(ljj);
        for (int i = j + 1; i < P; ++i) {
            double t = C[i * P + j];
            for (int k = 0; k < j; ++k) t -= L[i * P + k] * L[j * P + k];
            L[i * P + j] = t / ljj;
        }
    }
    if (logdet2) *logdet2 = ld;
    return true;
}

}  // namespace ssmq_bfgs
