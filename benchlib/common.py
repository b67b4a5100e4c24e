"""Shared pieces of bench.py's legs: constants, the committed PMC summaries, clock settling, the CPU baselines
(the C oracle timed on the host cores - kind "port"; only bench.py's cpu_baseline leg and its cross-checks call it)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9          # MI355X peak engine clock
F64_MFMA_PEAK_TF = 78.6  # MI355X fp64 matrix peak = 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (equals the fp64 vector peak)


def pmc_traffic(kernel_name, grid=None):
    """HBM bytes per launch of `kernel_name` at grid size `grid` (threads) from the committed PMC summary
    (profiles/pmc_traffic.json, produced by profiles/collect_pmc.sh + profiles/pmc_summary.py from separate rocprofv3
    --pmc passes, FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes).  Entries are keyed on
    (kernel, grid): the same kernel launched at two batch sizes has two entries.  None if there is no entry for this
    pair - never the figure of another grid."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None
    import re
    base = kernel_name.split('<')[0]
    nums = [int(v) for v in re.findall(r'=(\d+)', kernel_name)]
    chunked = 'chunked' in base      # k_filter_chunked (csrc/ssmq_filter_chunked.hip): as many waves as the chip holds, not ld threads
    if 'fused' in base or chunked:
        # reported name: <D=,Y=,ND=,NO=,F_DYN,F_OBS,FORM,TP=,SELO=,OPT=>; profile: <D,Y,ND,NO,FD,FO,FORM,TP,SELO,OPT,STU>
        want = nums[:4] + [1 if 'SSMQ_FORM_SIGMA' in kernel_name else 0] + nums[4:7]
        pick = lambda t: t[:4] + t[6:10]
    else:
        want = nums[:3]                                # (D, E, N) identify the shape
        pick = lambda t: t[:3]
    hits = []
    for key, rec in table.items():
        if key.startswith('_') or key.split('<')[0] != base:
            continue
        have = [int(v) for v in re.findall(r'-?\d+', key.split('<', 1)[1].split('>')[0])]
        if pick(have) == want:
            hits.append(rec)
    if grid is not None and not chunked:
        hits = [r for r in hits if int(r.get('grid', -1)) == int(grid)]
    return hits[0].get('hbm_bytes_per_launch') if len(hits) == 1 else None


def pmc_traffic_source():
    """Where `roofline.traffic` / `target_traffic` come from: the committed PMC summary (NOT counters of this run - rocprofv3
    --pmc cannot run inside the timed command), with the commit and date that summary was collected on."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None
    return 'profiles/pmc_traffic.json (separate rocprofv3 --pmc passes; tree {}, collected {})'.format(
        table.get('_commit', 'unknown'), table.get('_collected', 'unknown'))


def pmc_traffic_named(prefix):
    """HBM bytes per launch of the ONE entry of profiles/pmc_traffic.json whose kernel name (what stands before its template
    arguments and the grid) is `prefix`."""
    try:
        table = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    except (OSError, ValueError):
        return None
    hits = [rec for key, rec in table.items() if key.split('@')[0].split('<')[0] == prefix]
    return hits[0].get('hbm_bytes_per_launch') if len(hits) == 1 else None


_CPU_PORT = {}


def cpu_port_info():
    """Switch the C port to its -O3 -march=native build, compiled on THIS host when the first baseline leg runs
    (oracle/Makefile: native), and name the host: every cpu_baseline record carries `cpu_model` and `flags`."""
    if not _CPU_PORT:
        from oracle import c_oracle as co
        _CPU_PORT['flags'] = co.use_native()
        _CPU_PORT['cpu_model'] = co.cpu_model()
    return dict(_CPU_PORT)


def cgroup_cpu_share():
    """CPUs this process is ALLOWED to use at once when a cgroup quota is set (cpu.max / cfs_quota_us), else None: a GPU box shows
    256 hardware threads in its affinity mask and grants a one-GPU job 16 of them."""
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: (t.split()[0], t.split()[1])),):
        try:
            quota, period = parse(open(path).read())
            if quota != 'max' and float(period) > 0:
                return max(1, int(round(float(quota) / float(period))))
        except (OSError, ValueError, IndexError):
            pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and p > 0:
            return max(1, int(round(q / p)))
    except (OSError, ValueError):
        pass
    return None


def host_cores(max_threads=None):
    """Host threads the CPU baseline may use: ALL the cores this process may run on ("all host cores") - the
    affinity mask, cut to the cgroup's CPU quota where there is one.  Rounds 1-5 capped this at 16 outright; with no cap at all
    the port ran its OpenMP loop on 256 threads inside a 16-CPU quota and came out TEN TIMES slower (5.9e6 against 6.0e7 filter
    steps/s, round 6) - so where no quota can be read, `cpu_threads_for()` times both candidates and keeps the faster."""
    from oracle import c_oracle as co
    cpu_port_info()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    share = cgroup_cpu_share()
    if share:
        cores = min(cores, share)
    cores = min(cores, co.max_threads())
    return min(cores, max_threads) if max_threads else cores


def cpu_threads_for(run):
    """Thread count for a CPU-baseline leg: `run(threads)` executes one pass.  With a readable cgroup quota: host_cores().  Without
    one, the faster of {16, every core of the affinity mask} on a single probe pass each - the record's `cores` is what was used."""
    cores = host_cores()
    if cgroup_cpu_share() or cores <= 16:
        return cores
    best, best_t = cores, None
    for n in (16, cores):
        t0 = time.perf_counter()
        run(n)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = n, dt
    return best


def c_port_transforms(wl):
    """The filter of a FilterBench as transform blocks of the C oracle (oracle/ssmq_oracle.c), with the very weights
    the device run uses (BQ: tf.wm / Wc / Wcc / model_var as the HIP weights kernel produced them): what is compared and
    timed is the filter arithmetic, not two evaluations of an ill-conditioned inverse."""
    from oracle import c_oracle as co
    from ssmtoybox_amd.mtran import SigmaPointTransform
    out = []
    for tf, integ, E in ((wl.alg.tf_dyn, wl.f_dyn, wl.D), (wl.alg.tf_obs, wl.f_obs, wl.Y)):
        ci = co.Integrand.make(integ.id, [integ.par[i] for i in range(integ.n_par)],
                               [integ.idx[i] for i in range(integ.n_idx)] if integ.n_idx else None)
        if isinstance(tf, SigmaPointTransform):
            out.append(co.make_transform(1, tf.unit_sp.shape[0], E, tf.unit_sp, tf.wm, np.diag(tf.Wc).copy(),
                                         integrand=ci))
        else:
            mv = tf.model.model_var
            bc = 1 if tf.I_out.shape[0] != E else 0            # dim_out = 1 transforms broadcast the model variance
            emv = (np.asarray(mv, dtype=float) * np.ones((E, E))) if np.ndim(mv) == 0 else np.asarray(mv, dtype=float)
            nu = float(getattr(tf.model, 'nu', 0.0) or 0.0) if type(tf).__name__.startswith('StudentT') else 0.0
            out.append(co.make_transform(0, tf.model.points.shape[0], E, tf.model.points, tf.wm, tf.Wc, tf.Wcc, emv, bc,
                                         nu, tf.model.iK if nu > 0 else None, ci))
    return out


def cpu_baseline_filter(wl, B_sample, budget_s, what):
    """The C oracle's restatement of the same filter pass on the host cores (kind "port"), OpenMP over trajectories, on
    the first B_sample trajectories of the device run, repeated for ~budget_s.  Returns (record, fm (D, T, b), status)."""
    from oracle import c_oracle as co
    (td, k1), (to, k2) = c_port_transforms(wl)
    T = wl.T
    yb = np.ascontiguousarray(wl.y_host[:, :, :B_sample].transpose(2, 1, 0))
    GQG = wl.alg.G.dot(wl.alg.q_cov).dot(wl.alg.G.T)
    cores = cpu_threads_for(lambda n: co.filter_forward(td, to, yb, wl.m0, wl.P0, GQG, wl.alg.r_cov, threads=n))
    t0 = time.perf_counter()
    fm, fP, st = co.filter_forward(td, to, yb, wl.m0, wl.P0, GQG, wl.alg.r_cov, threads=cores)
    dt = time.perf_counter() - t0
    passes, total = 1, dt
    while total + dt < budget_s and passes < 2000:
        t0 = time.perf_counter()
        co.filter_forward(td, to, yb, wl.m0, wl.P0, GQG, wl.alg.r_cov, threads=cores)
        total += time.perf_counter() - t0
        passes += 1
    rec = {'value': passes * B_sample * T / total, 'unit': 'filter steps/s', 'cores': cores, 'kind': 'port', **cpu_port_info(),
           'sample': '{} passes of the first {} trajectories x T={} of {}, oracle/ssmq_oracle.c, OpenMP over '
                     'trajectories, {:.1f} s'.format(passes, B_sample, T, what, total)}
    return rec, fm.transpose(2, 1, 0), fP.transpose(2, 3, 1, 0), st


def cpu_baseline_apply(tf, integ_id, integ_par, D, E, means, covs, budget_s, what):
    """One batched moment transform in the C oracle (same weights as the device handle), on the host cores."""
    from oracle import c_oracle as co
    mv = tf.model.model_var
    emv = (np.asarray(mv, dtype=float) * np.ones((E, E))) if np.ndim(mv) == 0 else np.asarray(mv, dtype=float)
    t, keep = co.make_transform(0, D, E, tf.model.points, tf.wm, tf.Wc, tf.Wcc, emv,
                                integrand=co.Integrand.make(integ_id, integ_par))
    cores = cpu_threads_for(lambda n: co.apply_batch(t, means, covs, 0.0, threads=n))
    t0 = time.perf_counter()
    co.apply_batch(t, means, covs, 0.0, threads=cores)
    dt = time.perf_counter() - t0
    passes, total = 1, dt
    while total + dt < budget_s and passes < 2000:
        t0 = time.perf_counter()
        co.apply_batch(t, means, covs, 0.0, threads=cores)
        total += time.perf_counter() - t0
        passes += 1
    return {'value': passes * means.shape[0] / total, 'unit': 'transforms/s', 'cores': cores, 'kind': 'port', **cpu_port_info(),
            'sample': '{} passes of {} transforms of {}, oracle/ssmq_oracle.c, OpenMP over trajectories, {:.1f} s'.format(
                passes, means.shape[0], what, total)}


def pmc_issue(kernel):
    """SQ counters of a fused filter kernel from the committed summary (profiles/r02_fused_sq.csv: the rocprofv3 --pmc passes
    of tools/pmc_fused.sh over this bench; SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles,
    MI355X_MICROARCH.md).  `kernel`: the name bench.py reports (k_filter_fused<D=..,Y=..,ND=..,NO=..,..,FORM,TP=..,SELO=..,
    OPT=..>); matched against the template arguments <D, Y, ND, NO, FD, FO, FORM, TP, SELO, OPT, STU> of the profile."""
    import csv
    import re
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_fused_sq.csv')))      # the latest round's summary
    path = found[-1] if found else os.path.join(ROOT, 'profiles', 'r02_fused_sq.csv')
    base = kernel.split('<')[0]
    if base not in ('k_filter_fused', 'k_filter_chunked'):
        return None
    nums = [int(v) for v in re.findall(r'=(\d+)', kernel)]
    if len(nums) < 7:
        return None
    want = nums[:4] + [1 if 'SSMQ_FORM_SIGMA' in kernel else 0] + nums[4:7]      # D Y ND NO | FORM | TP SELO OPT
    rows = {}
    try:
        for r in csv.DictReader(open(path)):
            if base + '<' not in r['kernel']:
                continue
            t = [int(v) for v in re.findall(r'-?\d+', r['kernel'].split('<', 1)[1].split('>')[0])]
            if len(t) >= 10 and t[:4] + t[6:10] == want:
                rows[r['counter']] = float(r['mean_per_launch'])
    except (OSError, KeyError, ValueError):
        return None
    need = ('SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAVES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
            'SQ_INSTS_SALU')
    return rows if all(k in rows for k in need) else None


def issue_block(kernel, T, ms_per_launch):
    """The fused time loops are bound by fp64 VALU issue, not by HBM: instructions of the committed PMC pass against this
    run's HIP-event launch time, per wave."""
    pm = pmc_issue(kernel)
    if not pm:
        return None
    waves = pm['SQ_WAVES']
    valu_wave = pm['SQ_INSTS_VALU'] / waves
    peak = CLOCK_HZ / 4.0       # one fp64 VALU instruction per 4 cycles per SIMD
    achieved = valu_wave / (ms_per_launch * 1e-3)
    chunked = kernel.startswith('k_filter_chunked')     # a wave per wave SLOT there: each runs the chunks of several blocks
    return {'bound': 'fp64-issue', 'unit': 'VALU instructions/s per wave', 'achieved': achieved, 'peak': peak,
            'frac': achieved / peak, 'kernel': kernel,
            'valu_instructions_per_wave_per_step': None if chunked else valu_wave / T,
            'salu_instructions_per_wave_per_step': None if chunked else pm['SQ_INSTS_SALU'] / waves / T,
            'valu_instructions_per_wave': valu_wave,
            'pmc': {'source': 'profiles/r0*_fused_sq.csv, latest (rocprofv3 --pmc, tools/pmc_fused.sh)',
                    'frac_valu_x4_over_wave_cycles': pm['SQ_INSTS_VALU'] / pm['SQ_WAVE_CYCLES'],
                    'active_inst_valu_over_wave_cycles': pm['SQ_ACTIVE_INST_VALU'] / pm['SQ_WAVE_CYCLES'],
                    'wait_any_over_wave_cycles': pm['SQ_WAIT_ANY'] / pm['SQ_WAVE_CYCLES'],
                    'wait_inst_any_over_wave_cycles': pm['SQ_WAIT_INST_ANY'] / pm['SQ_WAVE_CYCLES'],
                    'waves': waves, 'simds': 1024},
            'note': 'a wave issues one fp64 VALU instruction per 4 cycles at best; frac = this kernel\'s instructions per '
                    'wave x 4 cycles / its launch time.  Waves beyond one per SIMD share the issue slots: with 1563 waves '
                    'on 1024 SIMDs (B = 1e5) the SIMDs that host two set the time, frac per wave is then at most 0.5'}


def timed_passes(wl, warmup, iters):
    for _ in range(warmup):
        wl.step()
    wl._lib.sync()
    e0, e1 = wl._lib.Event(), wl._lib.Event()
    e0.record()
    for _ in range(iters):
        wl.step()
    e1.record()
    return e0.elapsed_ms(e1) / iters


def settle(step, sync, seconds=0.06):
    """Run `step` untimed for about `seconds`: after the idle gaps between the legs of this script (set-up, host-side
    checks, the CPU baselines) the device needs some 20-50 ms of continuous work before its clocks are back up - a 0.5 ms
    kernel timed right after three warm-up launches read 15-25 % slow (tools/thermal_check.py: 577 / 512 / 482 us for
    consecutive groups of ten passes from idle, 455 us once warm, 572 us again after 2 s of idle).  The headline pass is
    not affected (32.1-32.4 us with 10, 500 or 3000 warm-up steps) and keeps exactly the --warmup it is given."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            step()
        sync()

