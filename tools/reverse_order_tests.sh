#!/bin/bash
# GPU tests in reverse definition order, in ONE process: a check for state carried between library calls (workspace /
# graph / constant caches) that the usual order would not expose
set -e
python -m pytest tests/test_gpu_parity.py -m gpu --collect-only -q 2>/dev/null | grep "::" | tac > /tmp/ids.txt
wc -l /tmp/ids.txt
timeout -k 10 600 python -m pytest -m gpu -q -p no:cacheprovider $(cat /tmp/ids.txt | tr '\n' ' ') 2>&1 | tail -6
