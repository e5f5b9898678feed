#!/bin/bash
# confidence run of the shipped library: the 4096-pose launches of both pose kernels in fresh processes, several batch sizes; stops at the first failure
export PYTHONPATH=$PWD
OUT=gpurun_out/pose_soak.txt; : > $OUT
for i in 1 2 3 4 5 6; do
  for B in 4096 1000,4096 4096,8192; do
    POSE_BATCHES=$B timeout -k 10 120 python tools/diag/pose_bench.py > gpurun_out/pose_soak_last.txt 2>&1; rc=$?
    echo "run $i batches $B rc=$rc $(grep -c workload gpurun_out/pose_soak_last.txt) lines" >> $OUT
    if [ $rc -ne 0 ]; then tail -5 gpurun_out/pose_soak_last.txt >> $OUT; exit 1; fi
  done
done
echo done >> $OUT
