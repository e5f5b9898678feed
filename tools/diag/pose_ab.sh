#!/bin/bash
# pose kernels: product against another build in one session (wall clock per launch, tools/diag/pose_bench.py).  usage: pose_ab.sh ALT.so
export PYTHONPATH=$PWD
timeout -k 10 400 python -m pytest tests/test_gpu_pose.py -x -q > gpurun_out/pose_ab_tests.log 2>&1 || exit 1
for rep in 1 2 3; do
  echo product; POSE_BATCHES=4096,16384 timeout -k 10 200 python tools/diag/pose_bench.py 2>/dev/null | cut -c60-330
  echo alt; POSE_LIB=$1 POSE_BATCHES=4096,16384 timeout -k 10 200 python tools/diag/pose_bench.py 2>/dev/null | cut -c60-330
done > gpurun_out/pose_ab.txt
