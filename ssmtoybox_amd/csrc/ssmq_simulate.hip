// Synthetic state and measurement trajectories generated on the device (reference: TransitionModel.simulate_discrete
// ssmod.py:168-199, MeasurementModel.simulate_measurements ssmod.py:1011-1039):
//     x[0] ~ N(m0, P0);   x[k] = dyn_fcn(x[k-1], q[k-1], k-1);   y[k] = meas_fcn(x[k], r[k], k+1)
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

struct SimArgs {
    int32_t D, Y, dq, dr, dyn_additive, obs_additive, T, fid_dyn, fid_obs;
    int32_t mode;   // bit 0: generate the states (else read them from x), bit 1: generate the measurements
    int64_t B, ld;
    uint64_t seed, traj_offset;
    // device constants: x0_mean[D] | x0_chol[D*D] | q_mean[dq] | q_chol[dq*dq] | G[D*dq] | r_mean[dr] | r_chol[dr*dr]
    const double *c;
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

// v = mean + L z with z ~ N(0, I_n) drawn for (traj, step, purpose)
__device__ __forceinline__ void gauss_vector(const SimArgs &a, uint64_t traj, uint32_t step, uint32_t purpose, int n,
                                             const double *mean, const double *L, double *v) {
    double z[SSMQ_MAX_DIM];
    for (int j = 0; j < n; j += 2) {
        double z0, z1;
        normal_pair(a.seed, traj, step, (purpose << 16) | (uint32_t)(j >> 1), &z0, &z1);
        z[j] = z0;
        if (j + 1 < n) z[j + 1] = z1;
    }
    for (int i = 0; i < n; ++i) {
        double s = mean[i];
        for (int k = 0; k <= i; ++k) s += L[i * n + k] * z[k];
        v[i] = s;
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
    const double *m0 = a.c, *L0 = m0 + D, *qm = L0 + D * D, *Lq = qm + dq, *G = Lq + dq * dq, *rm = G + D * dq,
                 *Lr = rm + dr;
    double aug[SSMQ_MAX_DIM + SSMQ_MAX_DIM], xs[kSimMaxAug], o[SSMQ_MAX_DIM], nz[SSMQ_MAX_DIM];
    const bool gen_x = a.mode & 1, gen_y = a.mode & 2;
    if (gen_x) gauss_vector(a, traj, 0u, 0u, D, m0, L0, aug);
    for (int k = 0; k < a.T; ++k) {
        for (int d = 0; d < D; ++d) {
            if (gen_x) __builtin_nontemporal_store(aug[d], &a.x[((int64_t)k * D + d) * a.ld + b]);   // written once, streamed
            else aug[d] = a.x[((int64_t)k * D + d) * a.ld + b];
        }
        if (gen_y) {
            // measurement of x[k], taken at time k + 1 (ssmod.py:1036-1038)
            gauss_vector(a, traj, (uint32_t)k, 2u, dr, rm, Lr, nz);
            for (int i = 0; i < dr; ++i) aug[D + i] = a.obs_additive ? 0.0 : nz[i];
            gather_inputs(a.fo, aug, D + (a.obs_additive ? 0 : dr), xs);
#pragma unroll
            for (int e = 0; e < SSMQ_MAX_DIM; ++e) o[e] = 0.0;
            eval_integrand(a.fid_obs, xs, (double)(k + 1), a.fo, o);
            for (int e = 0; e < Y; ++e)
                __builtin_nontemporal_store(o[e] + ((a.obs_additive && e < dr) ? nz[e] : 0.0),
                                            &a.y[((int64_t)k * Y + e) * a.ld + b]);
        }
        if (k + 1 == a.T || !gen_x) continue;
        // next state from x[k] with noise q[k] at time k (ssmod.py:196-198)
        gauss_vector(a, traj, (uint32_t)k, 1u, dq, qm, Lq, nz);
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

int launch_simulate(int mode, int D, int Y, int dq, int dr, int dyn_additive, int obs_additive, int T, int64_t B, int64_t ld,
                    uint64_t seed, uint64_t traj_offset, const ssmq_integrand *f_dyn, const ssmq_integrand *f_obs,
                    const double *d_consts, double *d_x, double *d_y, hipStream_t s) {
    SimArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = mode; a.D = D; a.Y = Y; a.dq = dq; a.dr = dr; a.dyn_additive = dyn_additive; a.obs_additive = obs_additive; a.T = T;
    a.B = B; a.ld = ld; a.seed = seed; a.traj_offset = traj_offset;
    a.c = d_consts; a.x = d_x; a.y = d_y;
    if (f_dyn) { a.fid_dyn = f_dyn->id; fill_fpar(f_dyn, &a.fd); }
    if (f_obs) { a.fid_obs = f_obs->id; fill_fpar(f_obs, &a.fo); }
    hipLaunchKernelGGL(k_simulate, dim3((unsigned)((B + kSimBlock - 1) / kSimBlock)), dim3(kSimBlock), 0, s, a);
    return hip_fail(hipGetLastError(), "k_simulate");
}

}  // namespace ssmq
