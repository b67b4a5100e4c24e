// Cross-lane primitives for a "one sigma point per lane" layout, gfx950.  Findings (profiles/r05_lane_prims.txt):
//   * fp64 VALU has ONE DPP form, row_newbcast (broadcast inside a row of 16 lanes), on v_mov_b64 and v_fmac_f64; inline asm
//     is outside the compiler's hazard recogniser: a VALU write of the DPP source needs two wait states before the DPP read
//     (k_sem reads garbage without the s_nop) - the intrinsic llvm.amdgcn.update.dpp.f64 (callable through an asm label) is
//     hazard-safe but is never folded into the FMA (v_fma_f64 is VOP3 when the combine pass runs).
//   * v_mfma_f64_4x4x4_4b contracts over lane / 16 (the ROW index); its four blocks are the quads (lane % 16) / 4 of every row:
//     A[i][k] at lane 16 k + 4 b + i, B[k][j] at 16 k + 4 b + j, D[i][j] at 16 i + 4 b + j.  Two of them all-reduce over the 16
//     lanes {16 k + 4 b + i}: quad b of all four rows - NOT a DPP row, so the two primitives want different trajectory groups.
//   * cost per SIMD with four waves: fmac_dpp = a plain FMA (2.3 ns); MFMA 4x4x4 7.0 ns; all-reduce of ONE double over 16 lanes
//     16.9 ns by two MFMAs, 32.6 ns by the 32-bit DPP butterfly (12 instructions) - six to thirteen FMA slots per reduced value.
// Primitives timed:
//   1. v_fmac_f64_dpp row_newbcast:k    acc += (src0 of lane k of the row) * src1        (the only DPP form fp64 VALU has)
//   2. v_mov_b64_dpp row_newbcast:k     broadcast of a double inside a row
//   3. v_mfma_f64_4x4x4_4b_f64          four 4x4x4 products, one per row: lane layouts of A, B and D found by probing,
//                                       and its use as an all-reduce over the 16 lanes of a row (two instructions)
//   4. the butterfly all-reduce on 32-bit DPP moves (row_mirror / row_half_mirror / quad_perm) + v_add_f64
// Part 1 prints semantics / layouts; part 2 the cost per instruction of each for one wave alone on its SIMD and for
// four waves per SIMD (s_memtime would be fine too; HIP events over a long loop are used, as in fp64_lat.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int K>
__device__ __forceinline__ double bcast(double v) {
    double d;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(K));
    return d;
}
template <int K>
__device__ __forceinline__ void fmac_bcast(double &acc, double a, double b) {   // acc += a[lane K of the row] * b
    // s_nop 1: two wait states between a VALU write of `a` and its DPP read (inline asm is not covered by the hazard recogniser)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(K));
}

template <int K>
__device__ __forceinline__ void fmac_bcast_raw(double &acc, double a, double b) {   // timing only: no wait states (values are not checked)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(K));
}

__global__ void k_sem(double *out) {
    const int l = threadIdx.x;
    double a = 100.0 + l, b = 1.0 + 0.001 * l, acc = 0.5;
    fmac_bcast<3>(acc, a, b);
    out[l] = acc;                       // expect 0.5 + (100 + 16 (l / 16) + 3) * (1 + 0.001 l)
    out[64 + l] = bcast<5>(a);          // expect 100 + 16 (l / 16) + 5
}

// MFMA probe: one wave per (p, q): A = unit at lane p, B = unit at lane q, C = 0; D[lane] recorded.
__global__ void k_probe(double *out) {
    const int l = threadIdx.x, p = blockIdx.x / 64, q = blockIdx.x % 64;
    const double a = l == p ? 1.0 : 0.0, b = l == q ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[(size_t)blockIdx.x * 64 + l] = d;
}

// all-reduce over the 16 lanes {16 k + 4 b + i : k, i = 0..3} (quad b of all four rows) with two MFMAs
__device__ __forceinline__ double row_allreduce_mfma(double v) {
    const double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);   // D[i][j] = sum_k A[i][k]  (B = ones)
    return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, s1, 0.0, 0, 0, 0);              // D[i][j] = sum_k B[k][j]  (A = ones)
}

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// DPP controls: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_mirror = 0x140, row_half_mirror = 0x141
__device__ __forceinline__ double row_allreduce_dpp(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v;
}

__global__ void k_allreduce(double *out) {
    const int l = threadIdx.x;
    const double v = 1.0 + l + 0.25 * (l % 7);
    out[l] = row_allreduce_mfma(v);
    out[64 + l] = row_allreduce_dpp(v);
}

// ---- cost ----------------------------------------------------------------------------------------------------------------
// OP 0: dependent v_fma_f64 chain (reference); 1: dependent fmac_bcast chain; 2: 4 independent fmac_bcast chains;
// 3: dependent MFMA 4x4x4 chain (through C); 4: 4 independent MFMA chains; 5: row_allreduce_mfma, dependent;
// 6: row_allreduce_dpp, dependent; 7: 4 independent row_allreduce_mfma; 8: 4 independent row_allreduce_dpp;
// 9: bcast (v_mov_b64_dpp) dependent; 10: fma + MFMA interleaved (1 MFMA chain + 4 fma per MFMA)
template <int OP>
__global__ void k_cost(double *out, int iters, double a, double b) {
    double v0 = 1.0 + 1e-3 * threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) v0 = __builtin_fma(v0, a, b);
            if (OP == 1) fmac_bcast_raw<3>(v0, v0, a);
            if (OP == 2) { fmac_bcast_raw<3>(v0, v1, a); fmac_bcast_raw<5>(v1, v2, a); fmac_bcast_raw<7>(v2, v3, a); fmac_bcast_raw<9>(v3, v0, a); }
            if (OP == 3) v0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v0, 0, 0, 0);
            if (OP == 4) {
                v0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v0, 0, 0, 0);
                v1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v1, 0, 0, 0);
                v2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v2, 0, 0, 0);
                v3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v3, 0, 0, 0);
            }
            if (OP == 5) v0 = row_allreduce_mfma(v0) * a;
            if (OP == 6) v0 = row_allreduce_dpp(v0) * a;
            if (OP == 7) { v0 = row_allreduce_mfma(v0) * a; v1 = row_allreduce_mfma(v1) * a; v2 = row_allreduce_mfma(v2) * a; v3 = row_allreduce_mfma(v3) * a; }
            if (OP == 8) { v0 = row_allreduce_dpp(v0) * a; v1 = row_allreduce_dpp(v1) * a; v2 = row_allreduce_dpp(v2) * a; v3 = row_allreduce_dpp(v3) * a; }
            if (OP == 9) v0 = bcast<3>(v0) * a;
            if (OP == 10) {
                v0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v0, 0, 0, 0);
                v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b); v1 = __builtin_fma(v1, a, b);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
}

template <int OP>
int cost(const char *name, double ops_per_unroll) {
    double *d;
    CHECK(hipMalloc(&d, sizeof(double) * 1024 * 1024));
    const int iters = 1 << 14;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg == 0 ? 1 : 256, threads = cfg == 0 ? 64 : 1024;      // one wave alone | 4 waves on every SIMD of every CU
        k_cost<OP><<<blocks, threads>>>(d, 64, 0.999999, 1e-7);
        CHECK(hipDeviceSynchronize());
        hipEventRecord(e0);
        k_cost<OP><<<blocks, threads>>>(d, iters, 0.999999, 1e-7);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 8 * ops_per_unroll;
        if (cfg == 0) printf("%-44s  lone wave: %7.2f ns per op", name, ms * 1e6 / n);
        else printf("   | 4 waves/SIMD: %7.2f ns per op per wave, %6.2f ns per op per SIMD\n", ms * 1e6 / n, ms * 1e6 / n / 4);
    }
    hipFree(d);
    return 0;
}

int main() {
    double *d;
    CHECK(hipMalloc(&d, sizeof(double) * 4096 * 64));
    std::vector<double> h(4096 * 64);
    k_sem<<<1, 64>>>(d);
    CHECK(hipMemcpy(h.data(), d, 128 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const double e0 = 0.5 + (100.0 + 16 * (l / 16) + 3) * (1.0 + 0.001 * l), e1 = 100.0 + 16 * (l / 16) + 5;
        if (fabs(h[l] - e0) > 1e-12 || h[64 + l] != e1) ++bad;
    }
    printf("v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast:k = source lane k of the SAME row of 16: %s (lane 17: %.6f, %.1f)\n",
           bad ? "NO" : "yes", h[17], h[64 + 17]);

    k_probe<<<4096, 64>>>(d);
    CHECK(hipMemcpy(h.data(), d, sizeof(double) * 4096 * 64, hipMemcpyDeviceToHost));
    // for each output lane: which (p, q) pairs contribute
    printf("v_mfma_f64_4x4x4_4b_f64: output lane <- sum over (A lane, B lane) pairs\n");
    for (int l = 0; l < 64; ++l) {
        if (l % 16 >= 6 && l != 63 && l != 21) continue;
        printf("  D lane %2d:", l);
        for (int p = 0; p < 64; ++p)
            for (int q = 0; q < 64; ++q)
                if (h[(size_t)(p * 64 + q) * 64 + l] != 0.0) printf(" (%d,%d)", p, q);
        printf("\n");
    }
    // hypothesis test: A[i][k] at lane 16 b + 4 k + i ... try all four index conventions
    for (int ha = 0; ha < 2; ++ha) for (int hb = 0; hb < 2; ++hb) for (int hd = 0; hd < 2; ++hd) {
        int mism = 0;
        for (int p = 0; p < 64 && !mism; ++p) for (int q = 0; q < 64 && !mism; ++q) for (int l = 0; l < 64; ++l) {
            const int bp = p / 16, bq = q / 16, bl = l / 16;
            const int ai = ha ? (p % 16) / 4 : p % 4, ak = ha ? p % 4 : (p % 16) / 4;
            const int bk = hb ? q % 4 : (q % 16) / 4, bj = hb ? (q % 16) / 4 : q % 4;
            const int di = hd ? l % 4 : (l % 16) / 4, dj = hd ? (l % 16) / 4 : l % 4;
            const double want = (bp == bl && bq == bl && ai == di && bj == dj && ak == bk) ? 1.0 : 0.0;
            if (h[(size_t)(p * 64 + q) * 64 + l] != want) { mism = 1; break; }
        }
        if (!mism) printf("  layout: A[i][k] at lane 16 b + %s; B[k][j] at lane 16 b + %s; D[i][j] at lane 16 b + %s\n",
                          ha ? "4 i + k" : "4 k + i", hb ? "4 j + k" : "4 k + j", hd ? "4 j + i" : "4 i + j");
    }
    k_allreduce<<<1, 64>>>(d);
    CHECK(hipMemcpy(h.data(), d, 128 * 8, hipMemcpyDeviceToHost));
    int okm = 1, okd = 1;
    for (int l = 0; l < 64; ++l) {
        double s = 0, sm = 0;
        for (int m = 16 * (l / 16); m < 16 * (l / 16) + 16; ++m) s += 1.0 + m + 0.25 * (m % 7);            // the DPP row of lane l
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 4; ++i) { const int m = 16 * k + 4 * ((l % 16) / 4) + i; sm += 1.0 + m + 0.25 * (m % 7); }   // its MFMA group
        if (fabs(h[l] - sm) > 1e-9) okm = 0;
        if (fabs(h[64 + l] - s) > 1e-9) okd = 0;
    }
    printf("all-reduce over quad b of all four rows by two MFMAs: %s (lane 0: %.3f, lane 20: %.3f); over a row of 16 by the DPP butterfly: %s\n", okm ? "correct" : "WRONG", h[0], h[20],
           okd ? "correct" : "WRONG");

    cost<0>("v_fma_f64, dependent", 1);
    cost<1>("v_fmac_f64_dpp row_newbcast, dependent", 1);
    cost<2>("v_fmac_f64_dpp row_newbcast, 4 chains", 4);
    cost<9>("v_mov_b64_dpp row_newbcast + v_mul, dependent", 1);
    cost<3>("v_mfma_f64_4x4x4_4b, dependent (C)", 1);
    cost<4>("v_mfma_f64_4x4x4_4b, 4 chains", 4);
    cost<10>("1 MFMA + 4 v_fma_f64 per group (per group)", 1);
    cost<5>("row all-reduce, 2 MFMA (+ mul), dependent", 1);
    cost<7>("row all-reduce, 2 MFMA (+ mul), 4 chains", 4);
    cost<6>("row all-reduce, DPP butterfly (+ mul), dep.", 1);
    cost<8>("row all-reduce, DPP butterfly (+ mul), 4 ch.", 4);
    hipFree(d);
    return 0;
}
