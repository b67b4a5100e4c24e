// The fused filter time loop for batches that do not fill the chip evenly (round 5).
//
// k_filter_fused gives a wave its 64 trajectories for all T steps.  A batch of 1e5 trajectories is 1 563 waves on 1 024 SIMDs:
// 539 SIMDs hold two waves, 485 hold one and are idle for the second half of the launch - two co-resident waves of these
// kernels (2 300 vector instructions per step, 250+ registers) share one issue port and each runs at half the speed of a wave that
// has the SIMD to itself: reentry UKF, B = 12 500 (every wave alone) 0.221 ms, B = 1e5 0.456 ms = 2.06 x for 1.53 x the waves
// per SIMD.  BASELINE configs[2] is exactly that batch.
//
// Here the T steps of a block of 64 trajectories are cut into chunks, and the chunks that are READY (their predecessor is complete)
// wait in a FIFO for whichever wave is free - the kernel is launched with every wave slot of the chip (two per SIMD where the
// registers allow, else one).  With more slots than blocks a block that finishes a chunk on a SIMD it shares is picked up by an
// idle wave on a SIMD with a free issue port; the wave that ran it goes idle: the chains of chunks rotate through the fast and the
// slow positions and all end together, at total work / total issue rate instead of at the pace of the doubly occupied SIMDs.
// With fewer slots than blocks (one wave per SIMD) every slot simply stays busy until the queue is empty.
//   * FIFO: q[0] = head (items taken), q[1] = continuations pushed, slots[i] = block | chunk << 24 of the i-th continuation
//     (-1: not pushed yet).  Item idx < n_blocks is chunk 0 of block idx; item idx >= n_blocks waits for
//     slots[idx - n_blocks].  n_items = n_chunks n_blocks is known, so nothing wraps and a wave whose index is beyond it ends.
//     No wait is circular: if every wave waited, every taken item would be complete and every continuation pushed - then the
//     indices the waves hold are beyond the end.  Every wait is bounded all the same (~seconds; then the block is marked failed).
//   * State: a chunk leaves (mean, lower triangle of the covariance, status word) - what its registers hold, so the RESULTS ARE THE
//     BITS OF THE WHOLE-PASS KERNEL - in a hand-over buffer, and its successor, possibly on another XCD, reads it from there.  The
//     eight L2 caches of the chip are not coherent with each other: an agent-scope release is an L2 write-back per chunk, measured
//     at ~0.5 us EACH and serial per XCD (a first version: 1.2 ms per pass against 0.45).  So the hand-over (and the queue words)
//     go through SYSTEM-scope atomic accesses, which bypass the non-coherent levels, ordered by the wave's own store counter
//     (workgroup-scope release = s_waitcnt): state stores complete -> slot.  The filter outputs keep their streaming stores.
#include "ssmq_filter_fused_kernel.h"

namespace ssmq {
namespace {

template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int OPT>
__global__ __launch_bounds__(kSmallBlock, (D >= 6 ? 1 : ((D >= 5 && FORM == SSMQ_FORM_SIGMA) ? SSMQ_FUSED_OCC_D5_SIGMA : 2))) void k_filter_chunked(const FusedArgs a) {
    __shared__ int32_t s_item;
    const int n_chunks = (a.T + a.t_chunk - 1) / a.t_chunk;
    const int n_items = n_chunks * a.n_blocks;
    int32_t *slots = a.queue + 16;        // [n_items - n_blocks]: block | chunk << 24 of the i-th continuation, -1 until it is pushed
    // (a wave takes at most every item once: the bound makes the loop finite whatever the queue holds)
    for (int taken = 0; taken <= n_items; ++taken) {
        if (threadIdx.x == 0) s_item = atomicAdd(a.queue, 1);
        __syncthreads();
        const int idx = __builtin_amdgcn_readfirstlane(s_item);
        __syncthreads();
        if (idx >= n_items) break;
        int blk = idx, c = 0;
        if (idx >= a.n_blocks) {
            int spins = 0, v;
            for (;;) {
                v = __hip_atomic_load(&slots[idx - a.n_blocks], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v >= 0 || ++spins > (1 << 22)) break;
                __builtin_amdgcn_s_sleep(32);
            }
            if (v < 0) break;              // (never seen: a continuation that did not arrive within seconds ends the wave)
            v = __builtin_amdgcn_readfirstlane(v);
            blk = v & 0xffffff;
            c = v >> 24;
        }
        const int k0 = c * a.t_chunk, k1 = k0 + a.t_chunk < a.T ? k0 + a.t_chunk : a.T;
        const bool last = k1 == a.T;
        if ((int)threadIdx.x < a.lpw)   // (STU = -1: the instantiation of the whole-pass kernel of these shapes - with the recursion
            // type fixed at compile time other products are contracted into multiply-adds and the last bits differ)
            fused_pass<D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, -1, true>(a, (uint32_t)blk, k0, k1, c == 0, last);
        if (!last) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // the state stores have completed (s_waitcnt) ...
            __syncthreads();
            if (threadIdx.x == 0) {                                      // ... then the continuation is published
                const int t = atomicAdd(a.queue + 1, 1);
                __hip_atomic_store(&slots[t], blk | ((c + 1) << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

typedef void (*chunked_kernel)(const FusedArgs);
struct ChunkedEntry {
    int fd, fo, D, Y, ND, NO, form, tp, selo, opt;
    chunked_kernel k;
    const char *name;
};
#define SSMQ_CH_ONE(FD, FO, D, Y, N, FORM, TP, SELO, OPT)                                                  \
    {FD, FO, D, Y, N, N, FORM, TP, SELO, OPT, &k_filter_chunked<D, Y, N, N, FD, FO, FORM, TP, SELO, OPT>, \
     "k_filter_chunked<D=" #D ",Y=" #Y ",ND=" #N ",NO=" #N "," #FD "," #FO "," #FORM ",TP=" #TP ",SELO=" #SELO ",OPT=" #OPT ">"}
// the shapes of ssmq_filter_fused.hip's table with five or six states (what a step of them costs makes a chunk of a few steps long
// against the hand-over: a queue atomic, D + D (D + 1) / 2 loads); unscented and spherical-radial point sets
#define SSMQ_CH(FD, FO, D, Y, N, SELO)                                  \
    SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 0), SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO, 0), \
    SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO, 0)
#define SSMQ_CH_FAST(FD, FO, D, Y, N, SELO)                             \
    SSMQ_CH(FD, FO, D, Y, N, SELO), SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 3), SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO, 2), \
    SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO, 2)
const ChunkedEntry kChunked[] = {
    SSMQ_CH_FAST(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0),
    SSMQ_CH_FAST(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0),
    SSMQ_CH_FAST(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 11, 1),
    SSMQ_CH(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 10, 0),
    SSMQ_CH(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 12, 0),
    SSMQ_CH(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 10, 1),
};

struct QueueBuf {          // per thread context, grow-only: queue words + hand-over buffer
    char *p = nullptr;
    size_t n = 0;          // bytes
    unsigned epoch = 0;
};
thread_local QueueBuf t_queue;

}  // namespace

// 1: launched; 0: this shape / batch keeps the whole-pass kernel; < 0: error.
// Default: taken for the kernels that hold ONE wave per SIMD (the centred 5-D forms, every 6-D shape) when the batch has more blocks
// than SIMDs and whole passes would cost at least 10 % more wave-times (ceil(x) against x, x = blocks per SIMD).  Measured
// (tools/chunked_time.py, B = 1e5, T = 50; profiles/r05_chunked.txt): reentry UKF 5-D 0.459 -> 0.432 ms, 6-D 0.594 -> 0.547; the
// kernels that hold two waves per SIMD LOSE 7-11 % (Bayes-Sard / GPQ 5-D) and keep the whole pass.  Why not the ideal (total work at
// full issue rate: 0.34 ms for the 5-D UKF): a hand-over is four dependent round trips to memory (state stores acknowledged,
// queue atomic, slot store; slot load, state loads: ~12 us per chunk of 40 us) that a wave alone on its SIMD cannot hide, and with
// two waves per SIMD (256 registers, 24 spilled) the pass is slower than the two rounds it replaces (0.548 ms).
// SSMQ_FUSED_CHUNKED=0 never; =1 wherever a kernel exists (same batch condition); = n > 1: chunks of n steps, any batch.  The
// results ARE the whole-pass kernel's bits in every variant (test_chunked_time_loop_is_bitwise_the_whole_pass).
int try_launch_chunked(const FusedArgs &a0, int fd, int fo, int D, int Y, int ND, int NO, int form, int tp, int selo, int opt, int cus,
                       hipStream_t s, bool dry_run, const char **name) {
    const char *ev = getenv("SSMQ_FUSED_CHUNKED");
    const int force = ev ? atoi(ev) : -1;
    if (force == 0 || a0.sscale != nullptr || a0.student_dof > 0.0 || (!dry_run && a0.T < 4)) return 0;
    const ChunkedEntry *e = nullptr;
    for (const ChunkedEntry &c : kChunked)
        if (c.fd == fd && c.fo == fo && c.D == D && c.Y == Y && c.ND == ND && c.NO == NO && c.form == form && c.tp == tp && c.selo == selo &&
            c.opt == opt)
            e = &c;
    if (!e) return 0;
    const int64_t n_blocks = (a0.B + a0.lpw - 1) / a0.lpw, simds = 4 * (int64_t)cus;
    if (n_blocks >= (int64_t)1 << 24) return 0;
    if (force <= 1) {
        const double x = (double)n_blocks / (double)simds;
        if (x <= 1.0 || std::ceil(x) < 1.1 * x) return 0;
    }
    // (register allocation as the whole-pass kernel's launch bounds: one wave per SIMD = four 64-thread blocks per CU; the
    // occupancy query needs a device, the name query has none)
    const bool one_per_simd = D >= 6 || (D >= 5 && form == SSMQ_FORM_SIGMA && SSMQ_FUSED_OCC_D5_SIGMA == 1);
    if (force < 0 && !one_per_simd) return 0;
    if (name) *name = e->name;
    if (dry_run) return 1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)e->k, kSmallBlock, 0) != hipSuccess || per_cu < 1) return 0;
    FusedArgs a = a0;
    a.n_blocks = (int32_t)n_blocks;
    a.t_chunk = force > 1 ? force : std::max(4, (a.T + 5) / 6);
    const int64_t n_chunks = (a.T + a.t_chunk - 1) / a.t_chunk, n_items = n_chunks * n_blocks;
    // queue words: [16] head, continuations pushed | slots [n_items - n_blocks];  hand-over: [n_blocks][NS + 1][64]
    if (n_chunks > 127) return 0;
    const size_t q_words = 16 + (size_t)(n_items - n_blocks), ns = (size_t)D + (size_t)D * (D + 1) / 2 + 1;
    const size_t q_bytes = (sizeof(int32_t) * q_words + 255) / 256 * 256, need = q_bytes + sizeof(double) * (size_t)n_blocks * ns * 64;
    if (t_queue.epoch != device_epoch() || t_queue.n < need) {
        if (t_queue.p && t_queue.epoch == device_epoch()) {
            SSMQ_HIP(hipStreamSynchronize(s));
            hipFree(t_queue.p);
        }
        t_queue = QueueBuf{};
        SSMQ_HIP(hipMalloc((void **)&t_queue.p, need + need / 4));
        t_queue.n = need + need / 4;
        t_queue.epoch = device_epoch();
    }
    a.queue = (int32_t *)t_queue.p;
    a.hand = (double *)(t_queue.p + q_bytes);
    SSMQ_HIP(hipMemsetAsync(a.queue, 0, sizeof(int32_t) * 16, s));
    if (n_items > n_blocks) SSMQ_HIP(hipMemsetAsync(a.queue + 16, 0xff, sizeof(int32_t) * (size_t)(n_items - n_blocks), s));
    const int64_t resident = (int64_t)per_cu * cus;
    const unsigned grid = (unsigned)std::min<int64_t>(resident, n_items);
    hipLaunchKernelGGL(e->k, dim3(grid), dim3(kSmallBlock), 0, s, a);
    const int rc = hip_fail(hipGetLastError(), "k_filter_chunked");
    return rc ? rc : 1;
}

}  // namespace ssmq
