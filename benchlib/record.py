"""The result line of bench.py.

The driver keeps the LAST line of stdout and only a few kB of it (round 4 lost its whole record to a 20.7 kB line), so the
line carries scalars only: the contract keys, `roofline` (the headline kernel plus the north-star transform as `target_*`),
`cpu_baseline`, and one number per secondary leg under `legs`.  Everything else - per-leg records, notes, workloads spelled
out, SQ counters, block timings - goes to `bench_detail.json` next to bench.py and to stderr.

`compact_record(detail)` is pure (no GPU, no files): tests/test_bench_record.py runs it on canned records.
"""
import json
import re

MAX_LINE = 4000          # bytes; the driver's window is 8 kB, VERDICT r04 asks for <= 4 kB

CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                 'vs_baseline', 'dtype', 'data')
REQUIRED_KEYS = CONTRACT_KEYS + ('config', 'roofline')


def short_kernel(name, limit=72):
    """k_filter_fused<D=1,Y=1,ND=3,NO=3,SSMQ_F_UNGM_DYN,SSMQ_F_UNGM_MEAS,SSMQ_FORM_BQ,TP=0,...> -> the same without the
    SSMQ_ prefixes, cut to `limit` characters (the full name is in bench_detail.json and in the rocprof stats)."""
    if not isinstance(name, str):
        return name
    s = re.sub(r'SSMQ_(F_|FORM_)?', '', name)
    return s if len(s) <= limit else s[:limit - 1] + '~'


def _num(v, digits=6):
    """Floats to `digits` significant digits (the detail file keeps full precision)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        f = float(v)
    except (TypeError, ValueError):
        return None
    if f != f or f in (float('inf'), float('-inf')):
        return None
    return float('%.*g' % (digits, f))


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_record(detail):
    """The scalars-only record printed as bench.py's last stdout line, from the full record `detail`."""
    out = {k: _num(detail.get(k), 10) for k in CONTRACT_KEYS if k in detail}
    cfg = detail.get('config', {})
    out['config'] = {k: _num(cfg[k]) for k in ('workload', 'mc_per_gpu', 'mc_total', 'time_steps', 'parallelism', 'collective',
                                               'allreduce_us', 'allreduce_bytes', 'settle_s', 'launcher', 'devices', 'distinct_devices',
                                               'comm_world') if k in cfg}
    for k, v in out['config'].items():
        if isinstance(v, str) and len(v) > 200:
            out['config'][k] = v[:199] + '~'
    rf = detail.get('roofline', {})
    r = {k: _num(rf.get(k)) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'ms_per_launch', 'bytes_per_launch')
         if k in rf}
    r['kernel'] = short_kernel(rf.get('kernel'))
    if isinstance(rf.get('traffic_source'), str):        # where `traffic` / `target_traffic` come from (committed PMC passes: file, tree, date)
        r['traffic_source'] = rf['traffic_source'][:160]
    for k in ('target_kernel', 'target_frac', 'target_achieved_gbs', 'target_ms_per_launch', 'target_bytes_per_launch',
              'target_bytes_moved_per_launch', 'target_traffic', 'target_max_scaled_err_vs_oracle', 'target_ms_min_block',
              'target_ms_max_block', 'target_frac_at_1e6', 'saturated_frac_1e4', 'saturated_frac_1e5', 'saturated_frac_1e6',
              'lane_kernel', 'register_kernel_ms'):
        if k in rf:
            r[k] = short_kernel(rf[k]) if k.endswith('kernel') else _num(rf[k])
    out['roofline'] = r
    cb = detail.get('cpu_baseline')
    if cb:
        c = {k: _num(cb.get(k)) for k in ('value', 'unit', 'cores', 'kind', 'cpu_model') if k in cb}
        if isinstance(cb.get('flags'), str):
            c['flags'] = cb['flags'][:100]
        if isinstance(cb.get('sample'), str):
            c['sample'] = cb['sample'][:140]
        out['cpu_baseline'] = c
    for k in ('ms_per_step_median', 'ms_per_step_min_block', 'ms_per_step_max_block', 'timing_blocks', 'api_steps_per_s',
              'api_ms_per_call', 'rmse', 'nll', 'trajectories_aggregated', 'excluded_failed_trajectories',
              'rel_diff_vs_cpu_port_median'):
        if k in detail:
            out[k] = _num(detail[k])
    legs = {}

    def leg(name, *path, scale=1.0):
        v = _get(detail, *path)
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            legs[name] = _num(v * scale)

    leg('mt6_ms', 'roofline_mt6', 'ms_per_launch')
    leg('mt6_frac', 'roofline_mt6', 'frac')
    leg('mt6_cpu_transforms_per_s', 'roofline_mt6', 'cpu_baseline', 'value')
    leg('c3_ukf5_ms', 'roofline_c3', 'ukf_reentry5', 'ms_per_launch')
    leg('c3_ukf5_frac', 'roofline_c3', 'ukf_reentry5', 'frac')
    leg('c3_bsq5_ms', 'roofline_c3', 'bsqkf_reentry5', 'ms_per_launch')
    leg('c3_ukf6_ms', 'roofline_c3', 'ukf_reentry6', 'ms_per_launch')
    leg('c3_ukf6_frac', 'roofline_c3', 'ukf_reentry6', 'frac')
    leg('c3_ukf5_share_ms', 'roofline_c3', 'ukf_reentry5_gpu_share', 'ms_per_launch')
    leg('c3_ukf5_share_register_kernel_ms', 'roofline_c3', 'ukf_reentry5_gpu_share', 'register_kernel_ms')
    leg('study6_steps_per_s', 'study6', 'steps_per_s')
    leg('study6_x_one_pass', 'study6', 'x_one_pass')
    leg('study6_serial_x_one_pass', 'study6', 'serial_x_one_pass')
    leg('c4_tpq_ms', 'roofline_c4', 'ms_per_launch')
    leg('c4_tpq_valu_per_wave_step', 'roofline_c4', 'issue', 'valu_instructions_per_wave_per_step')
    leg('c5_gemm_frac', 'roofline_c5', 'frac')
    leg('c5_n201_ms', 'roofline_c5', 'full_transform', 'ms_per_launch')
    leg('c5_n201_exec_frac', 'roofline_c5', 'full_transform', 'executed_frac')
    leg('c5_n21_ms', 'roofline_c5', 'unisolvent_n21', 'ms_per_launch')
    leg('c5_n21_frac', 'roofline_c5', 'unisolvent_n21', 'frac')
    leg('c5_deg7_ms', 'roofline_c5', 'degree7_as_worded', 'ms_per_launch')
    leg('c5_deg7_exec_frac', 'roofline_c5', 'degree7_as_worded', 'executed_frac')
    leg('c5_deg7_traffic_gb', 'roofline_c5', 'degree7_as_worded', 'traffic', scale=1e-9)
    leg('theta_us', 'theta_step', 'us_per_call')
    leg('marginal_us_per_traj_step', 'theta_step', 'marginal_filter_batch', 'us_per_trajectory_step')
    leg('marginal_failed', 'theta_step', 'marginal_filter_batch', 'failed_trajectories')
    leg('linear_frac', 'roofline_linear', 'frac')
    if legs:
        out['legs'] = legs
    if detail.get('detail_file'):
        out['detail'] = detail['detail_file']
    return out


def result_line(detail):
    """One line of strict JSON, at most MAX_LINE bytes.  Keys are dropped from the back of the optional groups, never from the
    contract, if a record should still come out too long."""
    rec = compact_record(detail)
    line = json.dumps(rec, allow_nan=False, separators=(',', ':'))
    for group in ('legs', 'detail'):
        if len(line) <= MAX_LINE:
            break
        if isinstance(rec.get(group), dict):
            while rec[group] and len(line) > MAX_LINE:
                rec[group].popitem()
                line = json.dumps(rec, allow_nan=False, separators=(',', ':'))
        else:
            rec.pop(group, None)
            line = json.dumps(rec, allow_nan=False, separators=(',', ':'))
    if len(line) > MAX_LINE:
        raise ValueError('bench.py: result line of {} bytes exceeds {}'.format(len(line), MAX_LINE))
    return line


def _clean(o):
    """NaN / inf -> None, NumPy scalars -> Python, for the detail file (strict JSON as well)."""
    if isinstance(o, dict):
        return {str(k): _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    if isinstance(o, (str, bool)) or o is None:
        return o
    if isinstance(o, int):
        return o
    try:
        f = float(o)
    except (TypeError, ValueError):
        return str(o)
    if f != f or f in (float('inf'), float('-inf')):
        return None
    return int(f) if hasattr(o, 'dtype') and 'int' in str(o.dtype) else f


def write_detail(detail, path):
    with open(path, 'w') as fh:
        json.dump(_clean(detail), fh, indent=1, allow_nan=False)
        fh.write('\n')
