#!/bin/bash
# launch-loop (hipGraph) filter path with library variants: do non-temporal stores of the transform outputs hurt when the
# next kernel of the loop reads them right back?
libs="$@"
for lib in $libs; do
  for wl in "reentry5 ukf 100000 50" "ungm gpqkf 10000 100"; do
    set -- $wl
    SSMQ_LIBRARY=$lib SSMQ_NO_FUSED=1 python bench.py --no-mt6 --no-cpu-baseline --workload $1 --filter $2 --batch $3 --time-steps $4 --steps 10 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$1', d['value'], d['ms_per_step'])"
  done
done
