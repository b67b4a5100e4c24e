#!/usr/bin/env python3
"""Performance gate: compares two `*_kernel_stats_by_grid.csv` files (rocprofv3 --kernel-trace of `python bench.py`, split by
grid size: tools/final_profiles.sh) per (kernel, grid) and FAILS when a kernel of the new file is more than --tolerance slower
than in the old one.  Round 4 shipped a 49 % slowdown of the configs[3] filter under a README that listed only what had
improved; parity has its bars in tests/_cases.py, kernel times have this.

  tools/perf_gate.py profiles/r04_h_bench_kernel_stats_by_grid.csv profiles/r05_x_bench_kernel_stats_by_grid.csv [--markdown]

Compared figures: MinNs AND AverageNs when both files have >= 5 calls of the pair; a pair fails only when BOTH are more than
--tolerance slower (a code regression shows in both; the two rounds' traces come from different boxes of the pool, and either
figure alone moves by 3-6 % between boxes for the same code object: the D = 6 transform's minimum read 15.1 / 15.6 / 16.0 /
16.2 us over four boxes with identical ISA).  The table shows the smaller of the two ratios.  Pairs with fewer launches are
listed with their averages and not gated.  Kernels whose template arguments changed between rounds are
matched by --alias OLD=NEW (substring of the name).  Exit code 1 on a regression, 0 otherwise; pairs present in only one file
are listed, not failed.
"""
import argparse
import csv
import re
import sys


def load(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r'^void\s+', '', r['Name'])
        name = re.sub(r'\s*\[clone.*', '', name)
        rows[(name, int(r['GridSize']))] = (int(r['Calls']), float(r['AverageNs']), float(r['MinNs']))
    return rows


def short(name, n=86):
    name = name.replace('ssmq::', '').replace(', ', ',')
    name = re.sub(r'\((FusedArgs|ApplyArgs|[A-Za-z:]*Args[^)]*)\)$', '', name)
    return name if len(name) <= n else name[:n - 1] + '~'


def compare(old, new, tolerance, min_us, aliases=()):
    """-> (rows, regressions).  rows: (kernel, grid, old_us, new_us, ratio, verdict)."""
    def alias(name):
        for a, b in aliases:
            if a in name:
                return name.replace(a, b)
        return name
    old = {(alias(k[0]), k[1]): v for k, v in old.items()}
    rows, bad = [], []
    for key in sorted(set(old) | set(new), key=lambda k: -(new.get(k, old.get(k))[1])):
        o, n = old.get(key), new.get(key)
        if o is None or n is None:
            rows.append((key[0], key[1], o and o[1] / 1e3, n and n[1] / 1e3, None, 'only in ' + ('old' if n is None else 'new')))
            continue
        use_min = o[0] >= 5 and n[0] >= 5
        a, b = (o[1], n[1])
        ratio = b / a if a > 0 else float('inf')
        if use_min and o[2] > 0 and n[2] / o[2] < ratio:       # the smaller of the two ratios, and its figures
            a, b = o[2], n[2]
            ratio = b / a
        verdict = 'ok'
        if max(a, b) / 1e3 < min_us:
            verdict = 'below {} us: not gated'.format(min_us)
        elif not use_min:
            # one to four launches: the average carries first-launch and clock-state effects of +-15 % (the set-up kernels of the
            # bench); reported, not gated
            verdict = 'few launches: not gated' + (' (slower)' if ratio > 1.0 + tolerance else '')
        elif ratio > 1.0 + tolerance:
            verdict = 'SLOWER'
            bad.append(key)
        elif ratio < 1.0 - tolerance:
            verdict = 'faster'
        rows.append((key[0], key[1], a / 1e3, b / 1e3, ratio, verdict + (' (min)' if (use_min and (a, b) == (o[2], n[2])) else ' (avg)')))
    return rows, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('old')
    ap.add_argument('new')
    ap.add_argument('--tolerance', type=float, default=0.05)
    ap.add_argument('--min-us', type=float, default=8.0, help='pairs faster than this in both files are reported, not gated')
    ap.add_argument('--alias', action='append', default=[], help='OLD=NEW substring replacement applied to names of the old file')
    ap.add_argument('--allow', action='append', default=[], help='substring of a kernel name whose slowdown is accepted (say why in profiles/README.md)')
    ap.add_argument('--markdown', action='store_true')
    ap.add_argument('--matched-only', action='store_true', help='leave out the pairs present in only one file')
    a = ap.parse_args()
    aliases = [tuple(s.split('=', 1)) for s in a.alias]
    rows, bad = compare(load(a.old), load(a.new), a.tolerance, a.min_us, aliases)
    bad = [k for k in bad if not any(s in k[0] for s in a.allow)]
    if a.markdown:
        print('| kernel | grid | old us | new us | new / old | |')
        print('|---|---|---|---|---|---|')
    for name, grid, o, n, ratio, verdict in rows:
        if a.matched_only and ratio is None:
            continue
        f = lambda v, fmt='%.1f': '-' if v is None else fmt % v      # noqa: E731
        if a.markdown:
            print('| `{}` | {} | {} | {} | {} | {} |'.format(short(name), grid, f(o), f(n), f(ratio, '%.3f'), verdict))
        else:
            print('{:<88} {:>8} {:>10} {:>10} {:>7}  {}'.format(short(name), grid, f(o), f(n), f(ratio, '%.3f'), verdict))
    if bad:
        sys.stderr.write('perf_gate: {} kernel(s) more than {:.0f} % slower: {}\n'.format(
            len(bad), 100 * a.tolerance, '; '.join('{} @ {}'.format(short(k[0], 60), k[1]) for k in bad)))
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
