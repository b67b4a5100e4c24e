#!/bin/bash
# The GPU parity suite once per alternative code path of the library (environment toggles read by libssmq): launch loop
# instead of the fused filter kernels, dense instead of fast-path transform kernels, three-pass instead of two-pass
# matrix-core route, generic kernel instead of the matrix cores, L2-resident instead of LDS-resident large-N weights,
# one wave per trajectory in the generic kernel, the wave kernel instead of the matrix-core tile kernel, the tile kernel's
# D = 16 mean path, the two-pass matrix-core route instead of the one-launch kernel, the launch-per-stage theta step, the blocked route instead of k_bq_stream, the single-workgroup large-N weights, the run-time-size linearisation kernel, k_bq_stream without the panel-wise tail.  The size tests are left out (they take the default paths anyway).
# Round 5: the two-launch theta step instead of k_theta_item, the device rounds / the host rounds of the batched marginalised filter instead
# of the one-launch kernel, no wave split, the fused time loop never / wherever possible as strips
# (SSMQ_FUSED_CHUNKED=64: 64 strips for every batch of more than 64 blocks; "=1" only meant "default" and is no longer in the list).
# Round 6: the LDL' kernels instead of the reflection-symmetric ones, no quad kernel, upload / pass / download instead of the pipelined
# forward pass, the forked graph instead of the one-kernel multi-filter launch (twenty-seven runs; SSMQ_TILE_NO_EXACT: the run-time-shape body of k_apply_tile for the shape that has an exact one).
# Exit status: non-zero when any run failed.
# usage: tools/alt_paths.sh [VAR=value ...]   (default: every toggle)
list="$*"
[ -z "$list" ] && list="SSMQ_NO_FUSED=1 SSMQ_NO_FASTPATH=1 SSMQ_NO_FUSED_COV=1 SSMQ_NO_MFMA=1 SSMQ_WEIGHTS_NO_LDS=1 SSMQ_WIDE_ONE_WAVE=1 SSMQ_NO_WAVE=1 SSMQ_NO_TILE=1 SSMQ_TILE_NO_MROW=1 SSMQ_NO_BQ_FUSED=1 SSMQ_NO_THETA_FUSED=1 SSMQ_NO_BQ_STREAM=1 SSMQ_WEIGHTS_ONE_WG=1 SSMQ_LINEAR_GENERIC=1 SSMQ_BQ_STREAM_NO_SPLIT=1 SSMQ_NO_THETA_ITEM=1 SSMQ_MARGINAL_ROUNDS=1 SSMQ_MARGINAL_HOST_ROUNDS=1 SSMQ_FUSED_WSPLIT=0 SSMQ_FUSED_CHUNKED=0 SSMQ_FUSED_CHUNKED=64 SSMQ_NO_SYM=1 SSMQ_FUSED_QUAD=0 SSMQ_NO_PIPED=1 SSMQ_MULTI_NO_FAMILY=1 SSMQ_TILE_NO_EXACT=1"
fail=0
for kv in $list; do
  v=${kv%%=*}
  echo "== $kv"
  env $kv timeout -k 10 300 python -m pytest tests -m gpu -q -k "not sixteen_million and not at_scale and not config3 and not full_batch" > gpurun_out/pytest_env_${v}_${kv##*=}.log 2>&1
  rc=$?
  tail -2 gpurun_out/pytest_env_${v}_${kv##*=}.log
  if [ $rc -ne 0 ]; then echo "   FAILED (exit $rc): $kv"; fail=1; fi
done
exit $fail
