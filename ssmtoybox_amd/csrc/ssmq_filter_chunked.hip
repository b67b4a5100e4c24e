// The fused filter time loop for batches that do not fill the chip evenly (round 5).
//
// k_filter_fused gives a wave its 64 trajectories for all T steps.  A batch of 1e5 trajectories is 1 563 waves on 1 024 SIMDs:
// 539 SIMDs hold two waves, 485 hold one and are idle for the second half of the launch - two co-resident waves of these
// kernels (2 300 vector instructions per step, 250+ registers) share one issue port and each runs at half the speed of a wave that
// has the SIMD to itself: reentry UKF, B = 12 500 (every wave alone) 0.221 ms, B = 1e5 0.456 ms = 2.06 x for 1.53 x the waves
// per SIMD.  BASELINE configs[2] is exactly that batch.
//
// Here the n_blocks x T block-steps of a batch are laid end to end (block-major) and cut into as many equal STRIPS as the chip has
// SIMDs; one wave per strip.  A strip of L = T n_blocks / n_strips > T block-steps is: the tail of a block its left neighbour began,
// some whole blocks, the head of a block its right neighbour will finish.  Every wave runs its HEAD piece first and leaves the
// state for the neighbour, then its whole blocks, then the TAIL piece - whose predecessor state was published by the neighbour as
// the first thing it did, a whole strip ago: one hand-over per wave, nobody waits, and every SIMD carries the same number of steps
// to within one: the launch lasts 1.53 wave-times instead of 2.
//   * State: a piece leaves (mean, lower triangle of the covariance, status word) - what its registers hold, so the RESULTS ARE THE
//     BITS OF THE WHOLE-PASS KERNEL - in a hand-over buffer, and the piece that continues the block, on another CU and possibly
//     another XCD, reads it from there.  The eight L2 caches of the chip are not coherent with each other: an agent-scope release
//     is an L2 write-back, measured at ~0.5 us EACH and serial per XCD.  So the hand-over and its flag go through SYSTEM-scope
//     atomic accesses, which bypass the non-coherent levels, ordered by the wave's own store counter (workgroup-scope release =
//     s_waitcnt): state stores complete -> flag.  The filter outputs keep their streaming stores.
//   * A first design dealt chunks of a few steps from a ready FIFO to free waves (profiles/r05_chunked.txt): correct, same bits, but
//     every chunk boundary costs ~6 us of exposed memory round trips and equal chunks quantise again (9.16 rounds of chunks are 10):
//     0.459 -> 0.432 ms.  The strips: 0.455 -> 0.405 ms (reentry UKF 5-D, B = 1e5), 0.584 -> 0.514 (6-D), 0.448 -> 0.285 at B = 7e4
//     (1 094 blocks: 1.07 rounds instead of 2) - the balanced figure at the ~2.07 GHz the chip holds with every SIMD busy.
//     Ordering, stated once: producer = [state: system-scope atomic stores, sc0 sc1 = write-through past the
//     per-XCD L2] -> release fence, workgroup scope (the compiler barrier) + an EXPLICIT s_waitcnt vmcnt(0) (every one of those
//     stores has been ACKNOWLEDGED: the fence alone emits no such wait for a one-CU workgroup, see the kernel) -> __syncthreads ->
//     [flag: system-scope atomic store].  Consumer = [spin on a system-scope
//     atomic load of the flag] -> acquire fence, workgroup scope (compiler barrier: the loads below stay below) -> [state:
//     system-scope atomic loads, sc0 sc1 = served from memory, never from this XCD's L2].  Every access of the shared words is an
//     atomic of system scope, so there is no data race in the language's sense; what the fences do NOT do is write back / invalidate
//     L2 (agent- or system-scope fences would: buffer_wbl2 / buffer_inv, ~0.5 us each) - and they need not, because no access of
//     these words is ever cached: that is the code sequence LLVM's AMDGPU memory model documents for system-scope atomics on
//     gfx942 / gfx950 (AMDGPUUsage, "Memory Model gfx942": "sc0=1 sc1=1" on loads, stores and RMWs of system scope), not an
//     observed accident.  Stress test: tests/test_gpu_parity.py::test_chunked_many_small_strips_across_xcds (SSMQ_FUSED_CHUNKED=n,
//     hundreds of hand-overs per launch, bitwise against the whole pass).
//   * Every wait is bounded (~seconds; then the block's trajectories are marked failed at step 0) although none is ever long.
#include "ssmq_filter_fused_kernel.h"

namespace ssmq {
namespace {

// STU: the recursion-type parameter of the whole-pass kernel of the shape (ssmq_filter_fused.hip: -1 decided at run time for the
// larger shapes, 0 = Gaussian fixed at compile time for the scalar ones) - with another value other products are contracted into
// multiply-adds and the last bits differ.
template <int D, int Y, int ND, int NO, int FD, int FO, int FORM, int TP, int SELO, int OPT, int STU>
__global__ __launch_bounds__(kSmallBlock, (D >= 6 ? 1 : ((D >= 5 && FORM == SSMQ_FORM_SIGMA) ? SSMQ_FUSED_OCC_D5_SIGMA : 2))) void k_filter_chunked(const FusedArgs a) {
    // The strip index is TAKEN (one atomic per wave), not read off blockIdx: a strip waits only for its left neighbour, and a wave
    // that holds index s took it after s - 1 was taken by a wave that is running - whatever order the hardware starts workgroups in,
    // and whatever else keeps part of the chip busy (the ordered-block-id rule of decoupled look-back scans).
    __shared__ int32_t s_strip;
    if (threadIdx.x == 0) s_strip = atomicAdd(a.queue + a.n_blocks, 1);
    __syncthreads();
    const int64_t strip = __builtin_amdgcn_readfirstlane(s_strip);
    // block-steps [p0, p1) of the block-major order
    const int64_t total = (int64_t)a.n_blocks * a.T;
    const int64_t p0 = total * strip / gridDim.x, p1 = total * (strip + 1) / gridDim.x;
    if (p1 <= p0) return;
    const int b0 = (int)(p0 / a.T), k0 = (int)(p0 - (int64_t)b0 * a.T);            // first block of the strip, its first step here
    const int b1 = (int)((p1 - 1) / a.T), k1 = (int)(p1 - (int64_t)b1 * a.T);      // last block, one past its last step here
    int32_t *flag = a.queue;                                                      // [n_blocks]: 1 = the head piece's state is in a.hand
    // the pieces in the order they are run: 1. the head of the last block (the right neighbour finishes it; not if the strip ends
    // on a block boundary), 2. the whole blocks, 3. the tail of the first block, from the state the left neighbour left as ITS
    // first piece.  (One call site for the pass: it is the whole time loop, inlined.)
    const bool head = k1 < a.T && (b1 > b0 || k0 == 0), tail = k0 > 0;
    const int w0 = b0 + (tail ? 1 : 0), w1 = b1 + ((k1 == a.T) ? 1 : 0);          // whole blocks [w0, w1)
    const int n_pieces = (head ? 1 : 0) + (w1 > w0 ? w1 - w0 : 0) + (tail ? 1 : 0);
    for (int i = 0; i < n_pieces; ++i) {
        const int j = i - (head ? 1 : 0);
        const bool is_head = head && i == 0, is_tail = tail && i == n_pieces - 1;
        const int blk = is_head ? b1 : (is_tail ? b0 : w0 + j);
        const int kb = is_tail ? k0 : 0, ke = is_head ? k1 : ((is_tail && b1 == b0) ? k1 : a.T);
        if (is_tail) {
            int spins = 0;
            while (__hip_atomic_load(&flag[blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
                if (++spins > (1 << 22)) break;
                __builtin_amdgcn_s_sleep(32);
            }
            if (spins > (1 << 22)) {        // (never seen)
                const int64_t b = (int64_t)blk * a.lpw + threadIdx.x;
                if ((int)threadIdx.x < a.lpw && b < a.B) a.status[b] = 1;
                break;
            }
            // ACQUIRE side of the hand-over (see the ordering note above the kernel): nothing below - in particular the system-scope
            // loads of a.hand in fused_pass - may be moved above the flag observation by the compiler, and the wave's own counters
            // are drained before it goes on.  Workgroup scope on purpose: it emits no cache invalidate, and none is needed, because
            // the state is READ with sc0 sc1 loads that do not look at the non-coherent levels.
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if ((int)threadIdx.x < a.lpw) fused_pass<D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, STU, true>(a, (uint32_t)blk, kb, ke, kb == 0, ke == a.T);
        if (is_head) {
            // RELEASE side: every state store of this wave has been ACKNOWLEDGED before the flag goes out.  The wait is spelled out:
            // round 5 relied on a workgroup-scope release fence for it, and hipcc lowers that fence to NO vmcnt wait when a
            // workgroup cannot span compute units (non-tgsplit mode; checked in the disassembly in round 6: the flag store followed
            // the state stores with only an lgkmcnt wait between them) - the flag, on another memory channel, could then overtake
            // the state it announces: one bitwise mismatch in ~10 suite runs (test_chunked_time_loop_is_bitwise_the_whole_pass,
            // ct / tpqkf, round 6).  The fence stays as the compiler-level barrier.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&flag[blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // ... then the flag
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
    {FD, FO, D, Y, N, N, FORM, TP, SELO, OPT, &k_filter_chunked<D, Y, N, N, FD, FO, FORM, TP, SELO, OPT, (D == 1 ? 0 : -1)>, \
     "k_filter_chunked<D=" #D ",Y=" #Y ",ND=" #N ",NO=" #N "," #FD "," #FO "," #FORM ",TP=" #TP ",SELO=" #SELO ",OPT=" #OPT ">"}
// the shapes of ssmq_filter_fused.hip's table with five or six states (2 000+ vector instructions per step: one wave keeps a SIMD's
// issue port busy by itself); unscented and spherical-radial point sets
#define SSMQ_CH(FD, FO, D, Y, N, SELO)                                  \
    SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 0), SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO, 0), \
    SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO, 0)
#define SSMQ_CH_FAST(FD, FO, D, Y, N, SELO)                             \
    SSMQ_CH(FD, FO, D, Y, N, SELO), SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 7), SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 0, SELO, 3), SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_BQ, 1, SELO, 2), \
    SSMQ_CH_ONE(FD, FO, D, Y, N, SSMQ_FORM_SIGMA, 0, SELO, 2)
const ChunkedEntry kChunked[] = {
    // (NOT the scalar UNGM filters: a 36-register kernel with a dependent chain of 125 instructions per step lives on many waves per
    // SIMD - one strip per SIMD ran the bench headline's kernel at 0.083 ms against 0.063 at B = 1e5 and 0.25 against 0.13 at 2e5,
    // 4 096 strips break even; profiles/r05_chunked.txt)
    SSMQ_CH_FAST(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0),
    SSMQ_CH_FAST(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0),
    SSMQ_CH_FAST(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 11, 1),
    SSMQ_CH(SSMQ_F_REENTRY2D_DYN, SSMQ_F_RADAR2D_MEAS, 5, 2, 10, 0),
    SSMQ_CH(SSMQ_F_REENTRY2D_BIAS_DYN, SSMQ_F_RADAR2D_MEAS, 6, 2, 12, 0),
    SSMQ_CH(SSMQ_F_CT_DYN, SSMQ_F_BEARING_MEAS, 5, 4, 10, 1),
};

}  // namespace

// 1: launched; 0: this shape / batch keeps the whole-pass kernel; < 0: error.
// Default: taken when the batch has more blocks than the chip has SIMDs and whole passes would cost at least 5 % more wave-times
// (ceil(x) against x, x = blocks per SIMD).  SSMQ_FUSED_CHUNKED=0 never; = n > 1: n strips (tests: any batch).
// The results ARE the whole-pass kernel's bits (test_chunked_time_loop_is_bitwise_the_whole_pass).
int try_launch_chunked(const FusedArgs &a0, int fd, int fo, int D, int Y, int ND, int NO, int form, int tp, int selo, int opt, int cus,
                       hipStream_t s, bool dry_run, const char **name) {
    const char *ev = ssmq::sw("SSMQ_FUSED_CHUNKED");
    const int force = ev ? atoi(ev) : -1;
    if (!dry_run && ctx().no_strips) return 0;      // a job of a multi-filter launch (one hand-over buffer per context)
    if (force == 0 || a0.sscale != nullptr || a0.student_dof > 0.0 || (!dry_run && a0.T < 2)) return 0;
    const ChunkedEntry *e = nullptr;
    for (const ChunkedEntry &c : kChunked)
        if (c.fd == fd && c.fo == fo && c.D == D && c.Y == Y && c.ND == ND && c.NO == NO && c.form == form && c.tp == tp && c.selo == selo &&
            c.opt == opt)
            e = &c;
    if (!e) return 0;
    const int64_t n_blocks = (a0.B + a0.lpw - 1) / a0.lpw, simds = 4 * (int64_t)cus;
    if (n_blocks >= (int64_t)1 << 24) return 0;
    int64_t strips = simds;
    if (force > 1) {
        strips = force;
    } else {
        const double x = (double)n_blocks / (double)simds;
        if (x <= 1.0 || std::ceil(x) < 1.05 * x) return 0;
    }
    if (strips >= n_blocks) return 0;              // (a strip must be longer than a block: one open piece at either end)
    if (name) *name = e->name;
    if (dry_run) return 1;
    FusedArgs a = a0;
    a.n_blocks = (int32_t)n_blocks;
    a.t_chunk = 0;
    // flags [n_blocks], strip counter | hand-over [n_blocks][NS + 1][64]
    const size_t ns = (size_t)D + (size_t)D * (D + 1) / 2 + 1;
    const size_t q_bytes = (sizeof(int32_t) * ((size_t)n_blocks + 1) + 255) / 256 * 256, need = q_bytes + sizeof(double) * (size_t)n_blocks * ns * 64;
    Ctx &cx = ctx();       // the buffer belongs to the thread's context (pooled; dropped with the context's other caches on a device change)
    if (cx.strip_bytes < need) {
        if (cx.strip_buf) {
            SSMQ_HIP(hipStreamSynchronize(s));
            hipFree(cx.strip_buf);
        }
        cx.strip_buf = nullptr;
        cx.strip_bytes = 0;
        SSMQ_HIP(hipMalloc(&cx.strip_buf, need + need / 4));
        cx.strip_bytes = need + need / 4;
    }
    a.queue = (int32_t *)cx.strip_buf;
    a.hand = (double *)((char *)cx.strip_buf + q_bytes);
    SSMQ_HIP(hipMemsetAsync(a.queue, 0, sizeof(int32_t) * ((size_t)n_blocks + 1), s));
    hipLaunchKernelGGL(e->k, dim3((unsigned)strips), dim3(kSmallBlock), 0, s, a);
    const int rc = hip_fail(hipGetLastError(), "k_filter_chunked");
    return rc ? rc : 1;
}

}  // namespace ssmq
