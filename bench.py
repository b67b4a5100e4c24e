#!/usr/bin/env python3
"""
bench.py - headline measurement: filter steps/s of the batched Monte-Carlo GPQ-Kalman filter on the UNGM model
(BASELINE.json configs[1]: GaussianProcessTransform (RBF, UT points), D = 1, N = 3, 1e4 MC trajectories per GPU,
T = 100 time steps), plus the achieved-bandwidth figure of the batched GPQ moment transform at state-dim 6 / 1e5
trajectories (BASELINE.json north_star target).

One "step" of this bench = one forward pass of the filter over the whole batch = B * T filter steps (each: two moment
transforms + one measurement update).  Inputs (measurements, initial moments, weights) are resident in HBM before the
timed region; filtered means / covariances of every step are written to HBM (forward_pass returns all of them).

Launch:  python bench.py [--gpus N --steps K --warmup W]          (N = 1)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)
Independent MC trajectories shard across ranks with no data-path collective (weak scaling: B per GPU fixed); the only
collective is the final all-reduce of the per-step error sums: RCCL behind the C ABI (ssmq_allreduce_sum), rendezvous
from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT).  No PyTorch in this process; compute
goes python -> ctypes -> libssmq.so (HIP).  (SSMQ_BENCH_BACKEND=gloo swaps in a torch.distributed gloo group for
rehearsals with several ranks on one GPU.)

OUTPUT.  The LAST line of stdout is the result record: one line of strict JSON of at most 4 kB with scalars only (the
contract keys, `roofline` with the north-star transform as `target_*`, `cpu_baseline`, one number per secondary leg under
`legs`; benchlib/record.py).  The full record of every leg goes to bench_detail.json next to this file (--detail PATH) and
to stderr.  The legs themselves live in benchlib/ (workloads.py, legs.py, launch.py, common.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SETTLE_S = 0.06          # continuous untimed passes before the warm-up passes (clock ramp-up after the idle set-up)

from benchlib.common import (HBM_PEAK_GBS, CLOCK_HZ, F64_MFMA_PEAK_TF, pmc_traffic, pmc_traffic_named, pmc_traffic_source, pmc_issue,     # noqa: E402,F401
                             issue_block, settle, timed_passes, cpu_port_info, host_cores, c_port_transforms,
                             cpu_baseline_filter, cpu_baseline_apply)
from benchlib.workloads import simulate_ungm, simulate_reentry, synthetic_reentry6, FilterBench                        # noqa: E402,F401
from benchlib.legs import (Mt6Bench, Study6Bench, measure_study6, register_kernel_ms, C5GemmBench, measure_c5_unisolvent, measure_c5_degree7, measure_linearize,        # noqa: E402,F401
                           measure_theta_step, filter_leg, saturated_sweep, measure_api_rate, c5_full_record)
from benchlib.launch import make_comm, final_aggregation, free_port, child_env, needs_launcher, launch_ranks, rank_devices  # noqa: E402,F401
from benchlib.record import compact_record, result_line, write_detail                                                  # noqa: E402,F401


def headline_blocks(wl, steps, blocks=5):
    """The headline pass once more, as `blocks` blocks of `steps` passes each, HIP events around every block: the contract's
    ms_per_step is ONE timed region (0.65 ms at the driver's --steps 20), the median over blocks says how stable it is."""
    from ssmtoybox_amd import _lib
    per = max(int(steps), 20)
    out = []
    for _ in range(blocks):
        e0, e1 = _lib.Event(), _lib.Event()
        e0.record()
        for _ in range(per):
            wl.step()
        e1.record()
        out.append(e0.elapsed_ms(e1) / per)
    return [float(v) for v in out], per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=10000, help='MC trajectories per GPU (weak scaling, the default)')
    ap.add_argument('--total-batch', type=int, default=0,
                    help='strong scaling: this many MC trajectories in total, split over the ranks in contiguous slices '
                         '(mcshard.shard_bounds); e.g. BASELINE configs[2]: --workload reentry6 --filter ukf '
                         '--total-batch 100000 --time-steps 50')
    ap.add_argument('--time-steps', type=int, default=100)
    ap.add_argument('--workload', default='ungm', choices=['ungm', 'reentry5', 'reentry6', 'ct'],
                    help="'ungm' is the headline (BASELINE configs[1]); the others are extra measurements")
    ap.add_argument('--filter', default='gpqkf', choices=['gpqkf', 'ukf', 'tpqkf', 'bsqkf'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-mt6', action='store_true', help='skip the single-kernel / other-config legs of the N = 1 run')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'bench_detail.json'),
                    help='where the full record of every leg goes (the stdout line keeps scalars only)')
    args = ap.parse_args()

    if needs_launcher(args.gpus):
        # one rank per GPU, started from here; nothing above or in this branch initialises the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # the result line goes to the descriptor stdout had at start-up, whatever a library does to fd 1 later
    result_out = os.fdopen(os.dup(1), 'w')
    import ssmtoybox_amd as amd
    if amd.device_count() < 1:
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    from ssmtoybox_amd import _lib, mcshard
    comm, rank, world, local_rank = make_comm()
    if world != args.gpus and rank == 0:
        sys.stderr.write('bench.py: --gpus {} but the launcher started {} rank(s); reporting n_gpus = {}\n'.format(
            args.gpus, world, world))
    devices = rank_devices(comm, rank, world)          # every rank's device name + PCI bus id (one small all-reduce)

    B, T = args.batch, args.time_steps
    strong = args.total_batch > 0
    if strong:
        lo, hi = mcshard.shard_bounds(args.total_batch, rank, world)
        B = hi - lo
        if B < 1:
            raise SystemExit('bench.py: --total-batch {} leaves rank {} of {} without trajectories'.format(
                args.total_batch, rank, world))
    wl = FilterBench(amd, B, T, seed=1 + rank, workload=args.workload, filt=args.filter)
    # The device has idled through the set-up above (uploads, weights, host-side simulation) and needs ~50 ms of continuous work
    # before its clocks are back up (tools/thermal_check.py): at the driver's --warmup 5 --steps 20 the contract's 0.65 ms region
    # otherwise starts on a device that is still ramping (34.4 us per pass against 32.9 for the blocks timed right after it).
    # Untimed, before the W warm-up passes, and said in the record (config.settle_s).
    settle(wl.step, _lib.sync, SETTLE_S)
    for _ in range(args.warmup):
        wl.step()
    comm.barrier()                      # common start: device synchronisation + barrier on every rank
    ev0, ev1 = _lib.Event(), _lib.Event()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        wl.step()
    ev1.record()
    _lib.sync()                         # this rank's K steps are complete ...
    elapsed = time.perf_counter() - t0
    pass_ms_dev = ev0.elapsed_ms(ev1) / max(args.steps, 1)
    comm.barrier()                      # ... closing barrier; the job's time is the slowest rank's
    elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])
    block_ms, block_len = headline_blocks(wl, args.steps)      # outside the contract's region: stability of the figure

    # final aggregation: per-time-step error sums -> RMSE / NLL (the path's only collective)
    fa = final_aggregation(
        comm, rank, world, mcshard.device_error_sums(wl.D, B, wl.ld, T, wl.d_x, wl.d_fm, wl.d_fP, wl.d_st),
        lambda mse: mcshard.device_lcr_sums(wl.D, B, wl.ld, T, wl.d_x, wl.d_fm, wl.d_fP, mse, wl.d_st), pass_ms_dev, B)
    agg, lcr, slot, allreduce_us, n_packed = fa['agg'], fa['lcr'], fa['slot'], fa['allreduce_us'], fa['n_packed']
    rmse, nll = agg['rmse_total'], float(agg['nll_avg'].mean())

    out = None
    headline = args.workload == 'ungm' and args.filter == 'gpqkf'
    if rank == 0:
        b_total = int(round(slot[world:].sum()))          # trajectories of all ranks (world x B when weak)
        steps_total = b_total * T * args.steps
        value = steps_total / elapsed
        bytes_pass = wl.bytes_per_pass()
        ach = bytes_pass / (pass_ms_dev * 1e-3) / 1e9
        out = {
            'metric': 'filter steps/sec (batched MC) for GPQ-Kalman UNGM' if headline else
            'filter steps/sec (batched MC), {} {}'.format(args.filter, args.workload),
            'value': value, 'unit': 'filter steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('GPQ-Kalman (GaussianProcessTransform, RBF, UT points) on UNGM, D=1, N=3, {} MC trajectories '
                                    'per GPU x T={} per pass (BASELINE configs[1])'.format(B, T))
                       if headline else
                       '{} on {} (D={}, Y={}), {} MC trajectories per GPU x T={}'.format(args.filter, args.workload, wl.D,
                                                                                       wl.Y, B, T),
                       'mc_per_gpu': B, 'mc_total': b_total, 'time_steps': T, 'parallelism': 'mc-shard x{}'.format(world),
                       'settle_s': SETTLE_S,
                       'per_rank_kernel_ms': [float(v) for v in slot[:world]],
                       'per_rank_trajectories': [int(round(v)) for v in slot[world:]],
                       'allreduce_us': allreduce_us, 'allreduce_bytes': 8 * n_packed,
                       'collective': type(comm).__name__ + (
                           ' (gloo fallback: ' + comm.fallback_reason + ')' if getattr(comm, 'fallback_reason', '') else ''),
                       'devices': devices['compact'], 'distinct_devices': devices['distinct'], 'per_rank_devices': devices['per_rank'],
                       'comm_world': devices['comm_world']},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': pmc_traffic(wl.kernel, wl.ld),
                         'traffic_source': pmc_traffic_source(), 'kernel': wl.kernel,
                         'bytes_per_launch': bytes_pass, 'ms_per_launch': pass_ms_dev,
                         'note': 'the contract\'s HBM figure for the headline filter pass (a serial recursion per trajectory: '
                                 'roofline_issue and DESIGN.md 3.4); the north-star transform is target_*'},
            'ms_per_step_median': float(np.median(block_ms)), 'ms_per_step_min_block': min(block_ms),
            'ms_per_step_max_block': max(block_ms), 'timing_blocks': '{} x {} passes, HIP events'.format(len(block_ms), block_len),
            'ms_per_step_blocks': block_ms,
            'rmse': rmse, 'nll': nll, 'inclination_indicator': float(np.mean(lcr)),
            'trajectories_aggregated': int(agg['count']),
            # what the averages above leave out (summed over ranks, worst time step): failed filters / singular covariances
            'excluded_failed_trajectories': int(agg['excluded_failed'].max()) if T else 0,
            'excluded_singular_covariances': int(agg['excluded_not_pd'].max()) if T else 0,
        }
        if world > 1 and type(comm).__name__ == 'RcclComm' and devices['distinct'] != world:
            raise SystemExit('bench.py: {} ranks over RCCL on {} distinct device(s): {}'.format(world, devices['distinct'],
                                                                                              devices['per_rank']))
        ib = issue_block(wl.kernel, T, pass_ms_dev)
        if ib:
            out['roofline_issue'] = ib
    single = world == 1      # the single-kernel legs and the CPU baselines belong to the N = 1 run only
    with_cpu = not args.no_cpu_baseline
    if rank == 0 and single and with_cpu and headline:
        cb, cpu_fm, _, cpu_st = cpu_baseline_filter(wl, B, 8.0, 'UNGM GPQ-Kalman (configs[1])')
        out['cpu_baseline'] = cb
        # the GPU pass and the CPU port ran the same trajectories: cross-check them
        fm, _, st = wl.results()
        good = (st == 0) & (cpu_st == 0)
        rel = np.abs(fm[0][:, good] - cpu_fm[0][:, good]) / np.max(np.abs(cpu_fm[0][:, good]))
        # identical weights and measurements; the UNGM recursion amplifies rounding differences along a trajectory
        # (uncentred covariance, bq/bqmtran.py:199), hence median and max over the 1e6 filtered means
        out['rel_diff_vs_cpu_port'] = {'median': float(np.median(rel)), 'p99': float(np.quantile(rel, 0.99)),
                                       'max': float(rel.max())}
        out['rel_diff_vs_cpu_port_median'] = out['rel_diff_vs_cpu_port']['median']
    if rank == 0 and single and headline and not args.no_mt6:
        # the same pass through the drop-in entry point: host arrays in and out (forward_pass returns host arrays, ssinf.py:118)
        out.update(measure_api_rate(B, T))
    if rank == 0 and single and not args.no_mt6:
        mt = Mt6Bench(amd, 100000, seed=2)
        err = mt.check()
        ms, b_alg, b_mov = mt.measure()
        ach = b_alg / (ms * 1e-3) / 1e9
        r6 = out['roofline_mt6'] = {
            'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': ach / HBM_PEAK_GBS, 'traffic': pmc_traffic(mt.kernel, mt.ld), 'kernel': mt.kernel,
            'bytes_per_launch': b_alg, 'bytes_moved_per_launch': b_mov, 'ms_per_launch': ms,
            'transforms_per_s': mt.B / (ms * 1e-3), 'max_scaled_err_vs_oracle': err, 'block_ms': mt.block_ms,
            'workload': 'batched GPQ moment transform, D=E=6, N=13, B=1e5, 4 rotating buffer sets '
                        '(north_star target; BASELINE configs[2] transform shape)'}
        # the north-star figure as SCALARS of `roofline`, the block the driver's record keeps (>= 0.40 asked on this kernel)
        out['roofline'].update({
            'target_kernel': r6['kernel'], 'target_frac': r6['frac'], 'target_achieved_gbs': r6['achieved'],
            'target_ms_per_launch': r6['ms_per_launch'], 'target_bytes_per_launch': r6['bytes_per_launch'],
            'target_bytes_moved_per_launch': r6['bytes_moved_per_launch'], 'target_traffic': r6['traffic'],
            'target_max_scaled_err_vs_oracle': r6['max_scaled_err_vs_oracle'],
            'target_timing': 'median of {} blocks of {} launches, HIP events'.format(len(mt.block_ms), max(1, 100 // len(mt.block_ms))),
            'target_ms_min_block': min(mt.block_ms), 'target_ms_max_block': max(mt.block_ms)})
        if with_cpu:
            means, covs = mt.host
            r6['cpu_baseline'] = cpu_baseline_apply(
                mt.tf, mt.model._fid, (0.1,), 6, 6, means[:50000], covs[:50000], 3.0, 'the D=E=6 GPQ transform')
        mt.free()
        # the same kernel with ten generations of waves instead of one (B = 1e6): how close it gets to HBM when the
        # load / compute / store phases of different waves overlap
        mt = Mt6Bench(amd, 1000000, seed=12, nsets=2)
        ms, b_alg, _ = mt.measure(warmup=3, iters=30)
        r6['at_1e6_trajectories'] = {'ms_per_launch': ms, 'achieved': b_alg / (ms * 1e-3) / 1e9, 'unit': 'GB/s',
                                     'frac': b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'bytes_per_launch': b_alg}
        mt.free()
        out['roofline']['target_frac_at_1e6'] = r6['at_1e6_trajectories']['frac']
        if headline:
            out['roofline']['saturated'] = saturated_sweep(amd, T, (10000, 100000, 1000000), wl.kernel, pass_ms_dev, B)
            for row in out['roofline']['saturated']:
                tag = {10000: '1e4', 100000: '1e5', 1000000: '1e6'}.get(row['mc'], str(row['mc']))
                out['roofline']['saturated_frac_' + tag] = row['frac']
                if 'issue_frac_chip' in row:
                    out['roofline']['saturated_issue_frac_chip_' + tag] = row['issue_frac_chip']
    if rank == 0 and single and not args.no_mt6:
        # BASELINE configs[2]: the filters that are stable on the reentry model (the GPQ-Kalman recursion itself fails
        # within three steps on every trajectory, in the reference as here: tests/test_gpu_parity.py::test_config3_gpqkf_*)
        out['roofline_c3'] = {
            'ukf_reentry5': filter_leg(amd, 'reentry5', 'ukf', 100000, 50, 31, 4000, 3.0,
                                       'UKF, reentry 5-D + radar (the reference\'s model), B=1e5 x T=50', with_cpu),
            'bsqkf_reentry5': filter_leg(amd, 'reentry5', 'bsqkf', 100000, 50, 32, 4000, 3.0,
                                         'Bayes-Sard Kalman (research/bsq/bsq_tracking.py set-up), reentry 5-D + radar, '
                                         'B=1e5 x T=50', with_cpu),
            'ukf_reentry6': filter_leg(amd, 'reentry6', 'ukf', 100000, 50, 33, 4000, 3.0,
                                       'UKF, reentry-shaped 6-D + radar (BASELINE state-dim 6), B=1e5 x T=50', with_cpu),
            # one GPU's share of configs[2] on an 8-GPU node: 12 500 trajectories
            'ukf_reentry5_gpu_share': filter_leg(amd, 'reentry5', 'ukf', 12500, 50, 35, 0, 0.0,
                                                 'UKF, reentry 5-D + radar, B=12500 (1e5 over 8 GPUs) x T=50', False),
        }
        out['roofline_c3']['ukf_reentry5_gpu_share']['register_kernel_ms'] = register_kernel_ms(amd, 'reentry5', 'ukf', 12500, 50, 35)
        # BASELINE configs[3]: t-process quadrature Kalman filter, 5-D coordinated turn + four bearing sensors
        out['roofline_c4'] = filter_leg(amd, 'ct', 'tpqkf', 10000, 20, 34, 2000, 3.0,
                                        'TPQ-Kalman (StudentProcessKalman), coordinated turn 5-D + 4 bearings, B=1e4 x T=20',
                                        with_cpu)
    if rank == 0 and single and not args.no_mt6:
        c5 = C5GemmBench(amd, 10000, seed=5)
        out['roofline_c5'] = c5_full_record(c5, with_cpu)
        out['roofline_c5']['unisolvent_n21'] = measure_c5_unisolvent(amd)
        out['roofline_c5']['degree7_as_worded'] = measure_c5_degree7(amd, with_cpu=with_cpu)
    if rank == 0 and single and headline and not args.no_mt6:
        out['study6'] = measure_study6(amd, B, T)
    if rank == 0 and single and not args.no_mt6:
        out['theta_step'] = measure_theta_step()
        out['roofline_linear'] = measure_linearize()
    if rank == 0:
        try:
            write_detail(out, args.detail)
            out['detail_file'] = os.path.basename(args.detail)
        except OSError as e:
            sys.stderr.write('bench.py: could not write {}: {}\n'.format(args.detail, e))
        sys.stderr.write('bench.py: full record\n' + json.dumps(out, default=str) + '\n')
        sys.stderr.flush()
        result_out.write(result_line(out) + '\n')          # the LAST line of stdout: scalars only, <= 4 kB
        result_out.flush()
    wl.free()
    comm.close()
    if getattr(comm, 'abandoned_rccl_thread', False):
        sys.stdout.flush()
        os._exit(0)          # a helper thread is still inside ncclCommInitRank: no atexit handler may wait for it


if __name__ == '__main__':
    main()
