#!/usr/bin/env python3
"""keep_larger_json.py SRC DST: copy the parity statistics SRC over DST unless DST already holds MORE records (a partial rerun must
not replace the full run's file - round 5 committed 40 triples where round 4 had ~1 050)."""
import json
import shutil
import sys


def count(path):
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return -1
    return len(d.get('records', d)) if isinstance(d, dict) else len(d)


src, dst = sys.argv[1:3]
a, b = count(src), count(dst)
if a >= b:
    shutil.copyfile(src, dst)
    print('{}: {} records (was {})'.format(dst, a, b))
else:
    print('{} kept: {} records, the new file has only {}'.format(dst, b, a))
