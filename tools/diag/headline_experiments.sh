#!/bin/bash
# The 100-knot launch (the bench line's `value`: the eight-wave latency kernel, one workgroup per knot on 100 of 256 CUs), VERDICT r04 item 6:
# what would two workgroups per knot buy, and what a prologue that stages half the tables?  Timing-only builds (wrong values):
#   libhipnlp_halflate.so   the task groups of the last two phases on half their lanes (the share of one of two workgroups per knot)
#   (no output stores at all, -DHIPNLP_DIAG_SKIP=7, was measured in round 2: 7.63 - 7.81 against 7.82 - 7.94 us)
#   libhipnlp_litestage.so  the eight-wave kernel stages the lite tables only
#   here (no GPU):  tools/diag/headline_experiments.sh build
#   GPU box:        tools/diag/headline_experiments.sh run     -> gpurun_out/headline_experiments.txt
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/diag/_build
if [ "${1:-}" = build ]; then
  mkdir -p $B
  for v in "halflate -DHIPNLP_DIAG_HALF_LATE" "litestage -DHIPNLP_DIAG_LITE_STAGE"; do
    set -- $v
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 $2 \
      "-DHIPNLP_BUILD_VARIANT=\"diagnostic, timing only: $1\"" -fPIC -shared -I $ROOT/include -o $B/libhipnlp_$1.so \
      $ROOT/hippopt_amd/csrc/hipnlp.hip $ROOT/hippopt_amd/csrc/hipnlp_pose.hip $ROOT/hippopt_amd/csrc/hipnlp_ipopt.cpp 2>/dev/null &
  done
  # the shipped sources through the same one-command build (no assembly patch), as the reference of the three
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 \
      '-DHIPNLP_BUILD_VARIANT="plain hipcc build of the shipped sources"' -fPIC -shared -I $ROOT/include -o $B/libhipnlp_plain.so \
      $ROOT/hippopt_amd/csrc/hipnlp.hip $ROOT/hippopt_amd/csrc/hipnlp_pose.hip $ROOT/hippopt_amd/csrc/hipnlp_ipopt.cpp 2>/dev/null &
  wait
  ls -la $B/libhipnlp_halflate.so $B/libhipnlp_litestage.so $B/libhipnlp_plain.so
  exit 0
fi
cd $ROOT
OUT=gpurun_out/headline_experiments.txt
: > $OUT
for rep in 1 2 3; do
  for L in plain halflate litestage; do
    HIPNLP_LIB_PATH=$B/libhipnlp_$L.so python3 bench.py --no-cpu-baseline --no-hessian --no-host --no-throughput --steps 2000 --warmup 200 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-10s rep $rep  ms_per_step %.5f  kernel_ms %.5f  (%.4g knots/s)' % ('$L', d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" >> $OUT
  done
done
cat $OUT
