// Synthetic state and measurement trajectories generated on the device (reference: TransitionModel.simulate_discrete
// ssmod.py:168-199, simulate_continuous :201-244, MeasurementModel.simulate_measurements ssmod.py:1011-1039):
//     x[0] ~ init_rv;   x[k] = dyn_fcn(x[k-1], q[k-1], k-1);   y[k] = meas_fcn(x[k], r[k], k+1)
//     continuous: x[k] = x[k-1] + dt dyn_fcn_cont(x[k-1], (sqrt(dt) / dt) q[k-1], k-1), the initial state not returned
// The three random variables are Gaussian, Student-t (utils.py:349-382 multivariate_t: mean + n / sqrt(u),
// n ~ N(0, scale), u ~ Gamma(nu / 2, 2 / nu)) or Gaussian mixtures (utils.py:254-299 gauss_mixture).
// One trajectory per lane, the state in registers / scratch, outputs written as planes [T][D][ld], [T][Y][ld] - the
// layout the filter kernels read, so a Monte-Carlo study never touches the host.
//
// Random numbers: Philox4x32-10 (Salmon et al., SC'11; counter-based, no state), keyed by the seed, counter =
// (global trajectory index lo, hi, time step, purpose << 16 | pair).  A trajectory's numbers depend on its GLOBAL index
// only, so a batch sharded over ranks reproduces the single-GPU data bit for bit whatever the world size.  Normals by
// Box-Muller on two 53-bit uniforms.  Parity with the reference's np.random streams can only be statistical; parity
// with the oracle's restatement of THIS generator is to rounding (tests/test_gpu_parity.py).
#include <cstring>
#include "ssmq_host.h"

namespace ssmq {
namespace {

constexpr int kSimBlock = 64;
constexpr int kSimMaxAug = kMaxIntegrandIn;   // state + noise inputs an integrand can read

// one random variable in the constants block: alpha[ncomp] | mean[ncomp][dim] | chol[ncomp][dim*dim] from `off`
struct RvDev {
    int32_t kind, dim, ncomp, off;
    double dof;
};

struct SimArgs {
    int32_t D, Y, dq, dr, dyn_additive, obs_additive, T, fid_dyn, fid_obs;
    int32_t mode;   // bit 0: generate the states (else read them from x), bit 1: generate the measurements
    int32_t continuous, g_off;      // Euler-Maruyama with the continuous-time dynamics; offset of G[D*dq] in the block
    double dt, qscale;              // continuous: step and sqrt(dt) / dt
    int64_t B, ld;
    uint64_t seed, traj_offset;
    RvDev x0, q, r;
    const double *c;                // device constants: the three random variables' blocks, G
    double *x, *y;
    FPar fd, fo;
};

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// two independent standard normals for (trajectory, step, purpose, pair)
__device__ __forceinline__ void normal_pair(uint64_t seed, uint64_t traj, uint32_t step, uint32_t tag, double *z0,
                                            double *z1) {
    uint32_t c[4] = {(uint32_t)traj, (uint32_t)(traj >> 32), step, tag};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // 53-bit uniforms in (0, 1): 27 + 26 bits, offset by half a step
    const double u1 = ((double)(((uint64_t)(c[0] >> 5) << 26) | (uint64_t)(c[1] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(((uint64_t)(c[2] >> 5) << 26) | (uint64_t)(c[3] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    *z0 = r * cs;
    *z1 = r * sn;
}

// one 53-bit uniform in (0, 1) for (trajectory, step, tag)
__device__ __forceinline__ double uniform_one(uint64_t seed, uint64_t traj, uint32_t step, uint32_t tag) {
    uint32_t c[4] = {(uint32_t)traj, (uint32_t)(traj >> 32), step, tag};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return ((double)(((uint64_t)(c[0] >> 5) << 26) | (uint64_t)(c[1] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
}

// Gamma(shape, 1), shape >= 1 (Marsaglia & Tsang 2000): attempt t draws its normal from tag base + 0x100 + t and its
// uniform from tag base + 0x180 + t - counter-based, so the result is a pure function of (seed, trajectory, step, purpose).
// 16 attempts fail together with probability < 1e-20; the last candidate is then taken.
__device__ __forceinline__ double gamma_mt(uint64_t seed, uint64_t traj, uint32_t step, uint32_t base, double shape) {
    const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double v = 1.0;
    for (uint32_t t = 0; t < 16; ++t) {
        double x, unused;
        normal_pair(seed, traj, step, base | (0x100u + t), &x, &unused);
        const double u = uniform_one(seed, traj, step, base | (0x180u + t));
        const double w = 1.0 + c * x;
        v = w * w * w;
        if (v > 0.0 && log(u) < 0.5 * x * x + d - d * v + d * log(v)) break;
        v = fabs(v) > 0.0 ? fabs(v) : 1.0;
    }
    return d * v;
}

// v ~ rv for (traj, step, purpose): Gaussian mean + L z; Student-t mean + L z / sqrt(u), u ~ Gamma(nu / 2, 2 / nu);
// mixture: component by one uniform against the cumulative proportions, then Gaussian
__device__ __forceinline__ void sample_rv(const SimArgs &a, const RvDev &rv, uint64_t traj, uint32_t step, uint32_t purpose,
                                          double *v) {
    const int n = rv.dim;
    const double *alpha = a.c + rv.off, *mean = alpha + rv.ncomp, *L = mean + rv.ncomp * n;
    if (rv.kind == SSMQ_RV_MIXTURE) {
        const double u = uniform_one(a.seed, traj, step, (purpose << 16) | 0x200u);
        int comp = rv.ncomp - 1;
        double acc = 0.0;
        for (int k = 0; k < rv.ncomp; ++k) {
            acc += alpha[k];
            if (u < acc) { comp = k; break; }
        }
        mean += comp * n;
        L += comp * n * n;
    }
    double z[SSMQ_MAX_DIM];
    for (int j = 0; j < n; j += 2) {
        double z0, z1;
        normal_pair(a.seed, traj, step, (purpose << 16) | (uint32_t)(j >> 1), &z0, &z1);
        z[j] = z0;
        if (j + 1 < n) z[j + 1] = z1;
    }
    double scale = 1.0;
    if (rv.kind == SSMQ_RV_STUDENT) {
        const double g = gamma_mt(a.seed, traj, step, purpose << 16, 0.5 * rv.dof) * (2.0 / rv.dof);
        scale = 1.0 / sqrt(g);
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s += L[i * n + k] * z[k];
        v[i] = mean[i] + s * scale;
    }
}

// continuous-time dynamics dx/dt = f(x, q) of the models that have one (ssmod.py:429-432, 569-585, 779-780)
__device__ __forceinline__ void eval_integrand_cont(int id, const double *x, const double *q, double *o) {
    if (id == SSMQ_F_REENTRY1D_DYN) {
        const double gam = 1.0 / 6.096;
        o[0] = -x[1] + q[0];
        o[1] = -exp(-gam * x[0]) * (x[1] * x[1]) * x[2] + q[1];
        o[2] = q[2];
    } else if (id == SSMQ_F_REENTRY2D_DYN) {
        const double r0 = 6374.0, h0 = 13.406, gm0 = 3.9860e5, b0 = -0.59783;
        const double b = b0 * exp(x[4]);
        const double rr = sqrt(x[0] * x[0] + x[1] * x[1]), vv = sqrt(x[2] * x[2] + x[3] * x[3]);
        const double dr = b * exp((r0 - rr) / h0) * vv, gr = -gm0 / (rr * rr * rr);
        o[0] = x[2];
        o[1] = x[3];
        o[2] = dr * x[2] + gr * x[0] + q[0];
        o[3] = dr * x[3] + gr * x[1] + q[1];
        o[4] = q[2];
    } else if (id == SSMQ_F_CTRS_DYN) {
        double sn, cs;
        sincos(x[3], &sn, &cs);
        o[0] = x[2] * cs;
        o[1] = x[2] * sn;
        o[2] = 0.0;
        o[3] = x[4];
        o[4] = 0.0;
    }
}

__device__ __forceinline__ void gather_inputs(const FPar &fp, const double *aug, int n_aug, double *xs) {
#pragma unroll
    for (int k = 0; k < kSimMaxAug; ++k) {
        const int src = fp.n_idx > 0 ? (k < fp.n_idx ? fp.idx[k] : 0) : (k < n_aug ? k : 0);
        xs[k] = aug[src];
    }
}

__global__ __launch_bounds__(kSimBlock) void k_simulate(const SimArgs a) {
    const uint32_t b = blockIdx.x * kSimBlock + threadIdx.x;
    if ((int64_t)b >= a.B) return;
    const uint64_t traj = a.traj_offset + b;
    const int D = a.D, Y = a.Y, dq = a.dq, dr = a.dr;
    const double *G = a.c + a.g_off;
    double aug[SSMQ_MAX_DIM + SSMQ_MAX_DIM], xs[kSimMaxAug], o[SSMQ_MAX_DIM], nz[SSMQ_MAX_DIM];
    const bool gen_x = a.mode & 1, gen_y = a.mode & 2;
    if (gen_x) sample_rv(a, a.x0, traj, 0u, 0u, aug);
    for (int k = 0; k < a.T; ++k) {
        if (gen_x && a.continuous) {
            // Euler-Maruyama step into column k (ssmod.py:236-243): the noise scaled by sqrt(dt) / dt, time index k
            sample_rv(a, a.q, traj, (uint32_t)k, 1u, nz);
            for (int i = 0; i < dq; ++i) nz[i] *= a.qscale;
#pragma unroll
            for (int e = 0; e < SSMQ_MAX_DIM; ++e) o[e] = 0.0;
            eval_integrand_cont(a.fid_dyn, aug, nz, o);
            for (int d = 0; d < D; ++d) aug[d] += a.dt * o[d];
        }
        for (int d = 0; d < D; ++d) {
            if (gen_x) __builtin_nontemporal_store(aug[d], &a.x[((int64_t)k * D + d) * a.ld + b]);   // written once, streamed
            else aug[d] = a.x[((int64_t)k * D + d) * a.ld + b];
        }
        if (gen_y) {
            // measurement of x[k], taken at time k + 1 (ssmod.py:1036-1038)
            sample_rv(a, a.r, traj, (uint32_t)k, 2u, nz);
            for (int i = 0; i < dr; ++i) aug[D + i] = a.obs_additive ? 0.0 : nz[i];
            gather_inputs(a.fo, aug, D + (a.obs_additive ? 0 : dr), xs);
#pragma unroll
            for (int e = 0; e < SSMQ_MAX_DIM; ++e) o[e] = 0.0;
            eval_integrand(a.fid_obs, xs, (double)(k + 1), a.fo, o);
            for (int e = 0; e < Y; ++e)
                __builtin_nontemporal_store(o[e] + ((a.obs_additive && e < dr) ? nz[e] : 0.0),
                                            &a.y[((int64_t)k * Y + e) * a.ld + b]);
        }
        if (k + 1 == a.T || !gen_x || a.continuous) continue;
        // next state from x[k] with noise q[k] at time k (ssmod.py:196-198)
        sample_rv(a, a.q, traj, (uint32_t)k, 1u, nz);
        for (int i = 0; i < dq; ++i) aug[D + i] = a.dyn_additive ? 0.0 : nz[i];
        gather_inputs(a.fd, aug, D + (a.dyn_additive ? 0 : dq), xs);
#pragma unroll
        for (int e = 0; e < SSMQ_MAX_DIM; ++e) o[e] = 0.0;
        eval_integrand(a.fid_dyn, xs, (double)k, a.fd, o);
        for (int d = 0; d < D; ++d) {
            double s = o[d];
            if (a.dyn_additive)
                for (int j = 0; j < dq; ++j) s += G[d * dq + j] * nz[j];
            aug[d] = s;
        }
    }
}

}  // namespace

bool has_continuous_dynamics(int fid) {
    return fid == SSMQ_F_REENTRY1D_DYN || fid == SSMQ_F_REENTRY2D_DYN || fid == SSMQ_F_CTRS_DYN;
}

int launch_simulate(const SimLaunch &h, hipStream_t s) {
    SimArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = h.mode; a.D = h.D; a.Y = h.Y; a.dq = h.dq; a.dr = h.dr; a.dyn_additive = h.dyn_additive;
    a.obs_additive = h.obs_additive; a.T = h.T; a.B = h.B; a.ld = h.ld; a.seed = h.seed; a.traj_offset = h.traj_offset;
    a.continuous = h.continuous; a.dt = h.dt; a.qscale = h.continuous ? sqrt(h.dt) / h.dt : 1.0; a.g_off = h.g_off;
    const RvDev *dst[3] = {&a.x0, &a.q, &a.r};
    for (int i = 0; i < 3; ++i) {
        RvDev &d = *const_cast<RvDev *>(dst[i]);
        d.kind = h.rv[i].kind; d.dim = h.rv[i].dim; d.ncomp = h.rv[i].ncomp; d.off = h.rv[i].off; d.dof = h.rv[i].dof;
    }
    a.c = h.d_consts; a.x = h.d_x; a.y = h.d_y;
    if (h.f_dyn) { a.fid_dyn = h.f_dyn->id; fill_fpar(h.f_dyn, &a.fd); }
    if (h.f_obs) { a.fid_obs = h.f_obs->id; fill_fpar(h.f_obs, &a.fo); }
    hipLaunchKernelGGL(k_simulate, dim3((unsigned)((h.B + kSimBlock - 1) / kSimBlock)), dim3(kSimBlock), 0, s, a);
    return hip_fail(hipGetLastError(), "k_simulate");
}

}  // namespace ssmq
