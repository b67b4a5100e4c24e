#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/collect_pmc.sh) into HBM bytes per launch.

Units and gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; WRITE_SIZE is
exact for streaming stores; FETCH_SIZE under-reports wide coalesced reads by 2x and is "uncalibrated" for other access
widths, so the read-side factor is CALIBRATED here on a kernel of this library with a known byte count and the same
8-byte-per-lane access pattern: the k_aos_to_soa dispatches of bench.py's setup (134.4 MB read, 134.4 MB written).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(pass_dir, counter):
    per_kernel = defaultdict(list)
    for path in glob.glob(os.path.join(pass_dir, '**', '*counter_collection.csv'), recursive=True):
        by_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(path)):
            if r.get('Counter_Name') != counter:
                continue
            did = r['Dispatch_Id']
            by_dispatch[did] += float(r['Counter_Value'])
            names[did] = (r['Kernel_Name'], int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0))
        for did, v in by_dispatch.items():
            per_kernel[names[did]].append(v)
    return per_kernel


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
    fetch = load(os.path.join(root, 'fetch'), 'FETCH_SIZE')
    write = load(os.path.join(root, 'write'), 'WRITE_SIZE')
    # calibration: bench.py's setup runs k_aos_to_soa on 4 x [(1e5, 6) means + (1e5, 36) covariances]: every dispatch reads
    # and writes exactly B * n * 8 bytes with the same 8-byte-per-lane pattern as the kernels of interest
    cal_bytes = 4 * (100000 * 6 * 8 + 100000 * 36 * 8)
    cal_r = [sum(v) for (name, grid), v in fetch.items() if 'k_aos_to_soa' in name and len(v) == 8]
    cal_w = [sum(v) for (name, grid), v in write.items() if 'k_aos_to_soa' in name and len(v) == 8]
    fr = cal_bytes / (cal_r[0] * 1024) if cal_r else None
    fw = cal_bytes / (cal_w[0] * 1024) if cal_w else None
    out = {'_calibration': {'kernel': 'k_aos_to_soa, 8 dispatches of bench.py setup', 'known_bytes_each_way': cal_bytes,
                            'fetch_factor': fr, 'write_factor': fw,
                            'note': 'bytes = counter_KiB * 1024 * factor; factor ~2 on the read side is the gfx950 '
                                    'FETCH_SIZE under-count'}}
    out['_commit'] = os.environ.get('SSMQ_COMMIT', 'unknown')     # the tree these counters were collected on
    import datetime
    out['_collected'] = datetime.datetime.utcnow().strftime('%Y-%m-%d %H:%M UTC')
    for (name, grid), v in fetch.items():
        if ('k_apply_small<6' in name.replace(' ', '') or 'k_filter_fused' in name or 'k_filter_chunked' in name or 'k_filter_quad' in name or 'k_filter_multi' in name or 'k_filter_range' in name or 'k_fxwc' in name or
                'k_eval_wave' in name or 'k_apply_tile' in name or 'k_big_rest' in name or 'k_bq_fused' in name or 'k_bq_stream' in name):
            w = write.get((name, grid), [0.0])
            rd = sum(v) / len(v) * 1024 * (fr or 2.0)
            wr = sum(w) / len(w) * 1024 * (fw or 1.0)
            key = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ssmq::', '').replace('ssmq::', '').replace(' ', '')
            key = '{}@{}'.format(key, grid)          # the same kernel at two batch sizes: two entries
            out[key] = {'launches': len(v), 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                        'hbm_bytes_per_launch': rd + wr, 'grid': grid}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
