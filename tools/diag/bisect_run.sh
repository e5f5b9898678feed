#!/bin/bash
# runs pose_bench (callbacks, 4096 poses) on each library given; stops at a result that is neither a pass nor a clean abort
export PYTHONPATH=$PWD
for v in "$@"; do
  POSE_LIB=tools/diag/_build/$v POSE_WHAT=${POSE_WHAT:-callbacks} POSE_BATCHES=4096 timeout -k 10 90 python tools/diag/pose_bench.py > gpurun_out/bis_$v.txt 2>&1
  rc=$?
  echo "$v rc=$rc" >> gpurun_out/bisect_log.txt
  if [ $rc -ne 0 ] && [ $rc -ne 134 ]; then exit 1; fi
done
