#!/bin/bash
# builds tools/micro/threads_rate.cpp against the C ABI and runs it (gpurun: bash tools/micro/run_threads_rate.sh > gpurun_out/threads_rate.txt)
set -e
cd "$(dirname "$0")/../.."
g++ -O2 -std=c++17 -Iinclude tools/micro/threads_rate.cpp -o /tmp/threads_rate -Lssmtoybox_amd -lssmq -lpthread -Wl,-rpath,$PWD/ssmtoybox_amd
timeout -k 10 300 /tmp/threads_rate "$@"
