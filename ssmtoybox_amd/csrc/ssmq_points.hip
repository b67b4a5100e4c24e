// Unit sigma-point sets and classical quadrature weights (host code; init-time, a few hundred numbers).
//   unscented            mtran.py:234-293      xi = [0 | c I | -c I], c = sqrt(D + lambda)
//   spherical-radial     mtran.py:171-204      xi = sqrt(D) [I | -I], w = 1 / (2 D)
//   Gauss-Hermite        mtran.py:315-360      product grid of the roots of He_deg, w = deg! / (deg^2 He_{deg-1}(x)^2)
//   fully symmetric (t)  mtran.py:405-578      degree 3 and 5 rules for Student-t densities
// Column order of every set is the reference's (the quadrature weights computed elsewhere are tied to it).
#include <cmath>
#include <string>
#include <vector>
#include "ssmq_host.h"

namespace {

using ssmq::set_error;

double he(int n, double x, double *prev = nullptr) {   // probabilists' Hermite He_n(x) (and He_{n-1})
    double p0 = 1.0, p1 = x;
    if (n == 0) {
        if (prev) *prev = 0.0;
        return 1.0;
    }
    for (int k = 1; k < n; ++k) {
        const double p2 = x * p1 - k * p0;
        p0 = p1;
        p1 = p2;
    }
    if (prev) *prev = p0;
    return p1;
}

// roots of He_n, ascending: bracketed by the interlacing roots of He_{n-1}, bisection then Newton, then made symmetric
std::vector<double> hermite_roots(int n) {
    std::vector<double> r;
    if (n == 1) return {0.0};
    std::vector<double> inner = hermite_roots(n - 1);
    const double big = 2.0 * std::sqrt((double)n) + 2.0;
    std::vector<double> edges = {-big};
    edges.insert(edges.end(), inner.begin(), inner.end());
    edges.push_back(big);
    for (int i = 0; i < n; ++i) {
        double lo = edges[i], hi = edges[i + 1];
        const bool up = he(n, lo) < 0.0;
        for (int it = 0; it < 200 && hi - lo > 1e-14 * (1.0 + std::fabs(lo)); ++it) {
            const double mid = 0.5 * (lo + hi);
            if ((he(n, mid) < 0.0) == up) lo = mid; else hi = mid;
        }
        double x = 0.5 * (lo + hi);
        for (int it = 0; it < 3; ++it) {          // He_n' = n He_{n-1}
            double pm1;
            const double p = he(n, x, &pm1);
            x -= p / (n * pm1);
        }
        r.push_back(x);
    }
    for (int i = 0; i < n / 2; ++i) {
        const double s = 0.5 * (r[n - 1 - i] - r[i]);
        r[i] = -s;
        r[n - 1 - i] = s;
    }
    if (n % 2) r[n / 2] = 0.0;
    return r;
}

// fully symmetric set of a generator with equal entries (mtran.py:522-578): leading index ascending, each sub-point as
// +u then -u; columns appended to `cols` (each of length dim)
void symmetric_set(int dim, const std::vector<double> &gen, std::vector<std::vector<double>> &cols) {
    if (gen.empty()) {
        cols.push_back(std::vector<double>(dim, 0.0));
        return;
    }
    for (int i = 0; i < dim; ++i) {
        std::vector<std::vector<double>> tails;
        if (gen.size() == 1) {
            tails.push_back(std::vector<double>(dim - i - 1, 0.0));
        } else if (dim - i - 1 > 0) {
            symmetric_set(dim - i - 1, std::vector<double>(gen.begin() + 1, gen.end()), tails);
        }
        for (const auto &tail : tails) {
            std::vector<double> u(dim, 0.0);
            u[i] = gen[0];
            for (size_t k = 0; k < tail.size(); ++k) u[i + 1 + k] = tail[k];
            cols.push_back(u);
            for (auto &v : u) v = -v;
            cols.push_back(u);
        }
    }
}

struct Rule {
    std::vector<std::vector<double>> cols;   // N columns of length D
    std::vector<double> wm, wc;
};

double par_or(const double *par, int n_par, int i, double dflt) {
    return (par && i < n_par && !std::isnan(par[i])) ? par[i] : dflt;
}

int build(int kind, int D, const double *par, int n_par, Rule &r) {
    if (D < 1 || D > SSMQ_MAX_DIM) {
        set_error("points: dimension out of range");
        return SSMQ_E_ARG;
    }
    auto axis = [&](int i, double v) {
        std::vector<double> c(D, 0.0);
        c[i] = v;
        return c;
    };
    switch (kind) {
        case SSMQ_PTS_UT: {
            const double kappa = par_or(par, n_par, 0, std::fmax(3.0 - D, 0.0)), alpha = par_or(par, n_par, 1, 1.0),
                         beta = par_or(par, n_par, 2, 2.0);
            const double lam = alpha * alpha * (D + kappa) - D, c = std::sqrt(D + lam);
            r.cols.push_back(std::vector<double>(D, 0.0));
            for (int i = 0; i < D; ++i) r.cols.push_back(axis(i, c));
            for (int i = 0; i < D; ++i) r.cols.push_back(axis(i, -c));
            r.wm.assign(2 * D + 1, 1.0 / (2.0 * (D + lam)));
            r.wc = r.wm;
            r.wm[0] = lam / (D + lam);
            r.wc[0] = r.wm[0] + (1.0 - alpha * alpha + beta);
            return SSMQ_OK;
        }
        case SSMQ_PTS_SR: {
            const double c = std::sqrt((double)D);
            for (int i = 0; i < D; ++i) r.cols.push_back(axis(i, c));
            for (int i = 0; i < D; ++i) r.cols.push_back(axis(i, -c));
            r.wm.assign(2 * D, 1.0 / (2.0 * D));
            r.wc = r.wm;
            return SSMQ_OK;
        }
        case SSMQ_PTS_GH: {
            const int deg = (int)par_or(par, n_par, 0, 3.0);
            if (deg < 1 || deg > 20 || std::pow((double)deg, D) > SSMQ_MAX_PTS) {
                set_error("points: Gauss-Hermite degree out of range (1..20, degree^D <= SSMQ_MAX_PTS)");
                return SSMQ_E_ARG;
            }
            const std::vector<double> x = hermite_roots(deg);
            std::vector<double> w(deg);
            double fact = 1.0;
            for (int k = 2; k <= deg; ++k) fact *= k;
            for (int i = 0; i < deg; ++i) {
                const double h = he(deg - 1, x[i]);
                w[i] = fact / ((double)deg * deg * h * h);
            }
            int N = 1;
            for (int d = 0; d < D; ++d) N *= deg;
            for (int n = 0; n < N; ++n) {        // last coordinate fastest (sklearn's cartesian, mtran.py:355-357)
                std::vector<double> c(D);
                int idx[SSMQ_MAX_DIM], rem = n;
                for (int d = D - 1; d >= 0; --d) {
                    idx[d] = rem % deg;
                    c[d] = x[idx[d]];
                    rem /= deg;
                }
                double wn = 1.0;                 // product formed coordinate 0 first, as np.prod over a row does
                for (int d = 0; d < D; ++d) wn *= w[idx[d]];
                r.cols.push_back(c);
                r.wm.push_back(wn);
            }
            r.wc = r.wm;
            return SSMQ_OK;
        }
        case SSMQ_PTS_FS: {
            int deg = (int)par_or(par, n_par, 0, 3.0);
            if (deg != 3 && deg != 5 && deg != 7) deg = 3;            // the reference prints a note and defaults to 3
            const double kappa = par_or(par, n_par, 1, std::fmax(3.0 - D, 0.0));
            const double dof = std::fmax(par_or(par, n_par, 2, 4.0), (double)deg);
            const double i2 = dof / (dof - 2.0);
            if (deg == 3) {
                const double u = std::sqrt(i2 * (D + kappa));
                r.cols.push_back(std::vector<double>(D, 0.0));
                for (int i = 0; i < D; ++i) r.cols.push_back(axis(i, u));
                for (int i = 0; i < D; ++i) r.cols.push_back(axis(i, -u));
                r.wm.assign(2 * D + 1, 1.0 / (2.0 * (D + kappa)));
                r.wm[0] = kappa / (D + kappa);
            } else if (deg == 7) {
                // NOT in the reference (its rules stop at degree 5, mtran.py:392): this build's degree-7 rule for BASELINE
                // configs[4], generators [0], [v1], [v2], [u, u], [u, u, u]; derivation in ssmtoybox_amd/mtran.py
                // (FullySymmetricStudentTransform.degree7_rule), exactness for all monomials of degree <= 7 is tested
                const int n = D;
                const double nu = dof, m2 = i2, m22 = nu * nu / ((nu - 2.0) * (nu - 4.0)), m4 = 3.0 * m22;
                const double m222 = nu * nu * nu / ((nu - 2.0) * (nu - 4.0) * (nu - 6.0)), m42 = 3.0 * m222, m6 = 15.0 * m222;
                const double sq = m42 / m22;
                const double d3 = n >= 3 ? m222 / (8.0 * sq * sq * sq) : 0.0;
                const double c2 = n >= 2 ? (m22 / (sq * sq) - 8.0 * (n - 2) * d3) / 4.0 : 0.0;
                const double t = 4.0 * (n - 1) * c2 + 4.0 * (n - 1) * (n - 2) * d3;
                const double r1 = m2 - t * sq, r2 = m4 - t * sq * sq, r3 = m6 - t * sq * sq * sq;
                const double e2 = r2 / r1, e1 = (r3 + e2 * r1) / r2, disc = e1 * e1 - 4.0 * e2;
                if (!(disc > 0.0 && e1 > 0.0 && e2 > 0.0)) {
                    set_error("points: degree-7 rule has no real axis generators for this dimension / dof");
                    return SSMQ_E_ARG;
                }
                const double p = 0.5 * (e1 + std::sqrt(disc)), q = 0.5 * (e1 - std::sqrt(disc));
                const double a = 0.5 * (r2 - r1 * q) / (p * (p - q)), b = 0.5 * (r1 * p - r2) / (q * (p - q));
                const int n_pair = 2 * n * (n - 1), n_trip = 4 * n * (n - 1) * (n - 2) / 3;
                const double u = std::sqrt(sq);
                symmetric_set(D, {}, r.cols);
                symmetric_set(D, {std::sqrt(p)}, r.cols);
                symmetric_set(D, {std::sqrt(q)}, r.cols);
                if (D > 1) symmetric_set(D, {u, u}, r.cols);
                if (D > 2) symmetric_set(D, {u, u, u}, r.cols);
                r.wm.push_back(1.0 - (2.0 * n * a + 2.0 * n * b + n_pair * c2 + n_trip * d3));
                for (int i = 0; i < 2 * n; ++i) r.wm.push_back(a);
                for (int i = 0; i < 2 * n; ++i) r.wm.push_back(b);
                for (int i = 0; i < n_pair; ++i) r.wm.push_back(c2);
                for (int i = 0; i < n_trip; ++i) r.wm.push_back(d3);
            } else {
                const double i22 = dof * dof / ((dof - 2.0) * (dof - 4.0)), i4 = 3.0 * i22, u = std::sqrt(i4 / i2);
                symmetric_set(D, {}, r.cols);
                symmetric_set(D, {u}, r.cols);
                if (D > 1) symmetric_set(D, {u, u}, r.cols);
                const double q = (i2 / i4) * (i2 / i4);
                const double a0 = 1.0 - D * q * (i4 - 0.5 * (D - 1) * i22), a1 = 0.5 * q * (i4 - (D - 1) * i22),
                             a11 = 0.25 * q * i22;
                r.wm.push_back(a0);
                for (int i = 0; i < 2 * D; ++i) r.wm.push_back(a1);
                for (int i = 0; i < 2 * D * (D - 1); ++i) r.wm.push_back(a11);
            }
            r.wc = r.wm;
            return SSMQ_OK;
        }
        default:
            set_error("points: unknown kind");
            return SSMQ_E_ARG;
    }
}

}  // namespace

extern "C" int ssmq_points_count(int kind, int D, const double *par, int n_par) {
    Rule r;
    const int rc = build(kind, D, par, n_par, r);
    return rc ? rc : (int)r.cols.size();
}

extern "C" int ssmq_points(int kind, int D, const double *par, int n_par, double *xi, double *wm, double *wc) {
    Rule r;
    const int rc = build(kind, D, par, n_par, r);
    if (rc) return rc;
    const int N = (int)r.cols.size();
    if ((int)r.wm.size() != N) {
        set_error("points: internal size mismatch");
        return SSMQ_E_ARG;
    }
    if (xi)
        for (int d = 0; d < D; ++d)
            for (int n = 0; n < N; ++n) xi[d * N + n] = r.cols[n][d];
    if (wm)
        for (int n = 0; n < N; ++n) wm[n] = r.wm[n];
    if (wc)
        for (int n = 0; n < N; ++n) wc[n] = r.wc[n];
    return N;
}
