#!/bin/bash
for b in 16384 32768 65536 81920 100000 131072 262144 1000000; do MT6_B=$b MT6_NOCHECK=1 MT6_ROUNDS=5 python tools/mt6_time.py | sed "s/^/B=$b /"; done
