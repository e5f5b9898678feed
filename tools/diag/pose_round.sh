#!/bin/bash
# pose kernels: parity tests, barrier timeline (diagnostic build), wall-clock bench.  usage: pose_round.sh TAG
export PYTHONPATH=$PWD
T=${1:-x}
timeout -k 10 400 python -m pytest tests/test_gpu_pose.py -x -q > gpurun_out/pose_tests_$T.log 2>&1 || exit 1
(for w in callbacks hessian; do echo "=== $w B=4096"; timeout -k 10 120 python tools/diag/pose_stamps.py 4096 $w || exit 1; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/pose_stamps_$T.txt
POSE_BATCHES=1,4096 timeout -k 10 200 python tools/diag/pose_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/pose_bench_$T.txt
