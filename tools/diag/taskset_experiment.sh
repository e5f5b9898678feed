#!/bin/bash
# VERDICT r05 item 2, the cheap half first: what would each kind of workgroup be left with if the knot program were split into a kinematic and
# a model-free workgroup inside one launch?  Timing-only builds (wrong values) of the SAME kernels with one half of the task groups compiled out:
#   libhipnlp_konly.so   -DHIPNLP_DIAG_TASKSET=1   the groups that need the robot model
#   libhipnlp_monly.so   -DHIPNLP_DIAG_TASKSET=2   the model-free groups
#   libhipnlp_plain.so   the shipped sources through the same one-command build (the reference of the two)
# Measured: the 100-knot launch (the headline: eight-wave kernel) and the x 64 varying-first batch launch (four-wave VARY kernel).
#   here (no GPU):  tools/diag/taskset_experiment.sh build
#   GPU box:        tools/diag/taskset_experiment.sh run     -> gpurun_out/r06_taskset_experiment.txt
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/diag/_build
if [ "${1:-}" = build ]; then
  mkdir -p $B
  for v in "konly -DHIPNLP_DIAG_TASKSET=1" "monly -DHIPNLP_DIAG_TASKSET=2" "plain -DHIPNLP_PLAIN_REFERENCE"; do
    set -- $v
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 $2 \
      "-DHIPNLP_BUILD_VARIANT=\"diagnostic, timing only: $1\"" -fPIC -shared -I $ROOT/include -o $B/libhipnlp_$1.so \
      $ROOT/hippopt_amd/csrc/hipnlp.hip $ROOT/hippopt_amd/csrc/hipnlp_pose.hip $ROOT/hippopt_amd/csrc/hipnlp_ipopt.cpp 2>/dev/null &
  done
  wait
  ls -la $B/libhipnlp_konly.so $B/libhipnlp_monly.so $B/libhipnlp_plain.so
  exit 0
fi
cd $ROOT
OUT=gpurun_out/r06_taskset_experiment.txt
: > $OUT
for rep in 1 2; do
  for L in plain konly monly; do
    for cfg in "--batch 1" "--batch 64 --varying-first" "--workload stairs --horizon 200 --batch 16 --varying-first"; do
      HIPNLP_LIB_PATH=$B/libhipnlp_$L.so python3 bench.py $cfg --no-cpu-baseline --no-hessian --no-host --no-throughput --steps 1000 --warmup 100 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-6s %-58s rep $rep  ms_per_step %.5f  kernel_ms %.5f  (%.4g knots/s)' % ('$L', '$cfg', d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" >> $OUT
    done
  done
done
cat $OUT
