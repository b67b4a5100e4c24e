#!/bin/bash
# k_apply_tile: parity against the other two generic kernels, timing at D = 10 / N = 21 against k_apply_wave, SQ counters
export TMPDIR=/tmp
out=gpurun_out/r03_tile
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wave_kernel or bsq_d10 or generic_kernel or apply_golden" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest.log
timeout -k 10 120 python tools/c5_n21.py 100000 > $out/n21_tile.txt 2>&1; cat $out/n21_tile.txt
if [ "$1" = "pmc" ]; then timeout -k 10 400 bash tools/pmc_n21.sh > $out/pmc_n21.txt 2>&1; cat $out/pmc_n21.txt; fi
