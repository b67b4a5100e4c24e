#!/bin/bash
# A/B timing of library variants on one device: tools/ab.sh variants/libssmq_a.so variants/libssmq_b.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    SSMQ_LIBRARY=$lib python tools/mt6_time.py
  done
done
