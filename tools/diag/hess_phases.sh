#!/bin/bash
# Where the exact-Hessian kernel's time goes: diagnostic builds that run only the first n of the program's six phases
# (-DHIPNLP_HESS_DIAG_PHASES=n; staging, copy-out and launch are in every build; the values of a truncated build are wrong),
# timed with tools/diag/hess_bench.py.
#   here (no GPU):  tools/diag/hess_phases.sh build            -> tools/diag/_build/libhipnlp_hp<n>.so, n = 0..5
#   GPU box:        tools/diag/hess_phases.sh run [batches]    -> gpurun_out/hess_phases.txt
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/diag/_build
if [ "${1:-}" = build ]; then
  mkdir -p $B
  for n in 0 1 2 3 4 5; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -DHIPNLP_HESS_DIAG_PHASES=$n \
      '-DHIPNLP_BUILD_VARIANT="diagnostic: Hessian program truncated"' -fPIC -shared -I $ROOT/include -o $B/libhipnlp_hp$n.so \
      $ROOT/hippopt_amd/csrc/hipnlp.hip $ROOT/hippopt_amd/csrc/hipnlp_pose.hip $ROOT/hippopt_amd/csrc/hipnlp_ipopt.cpp 2>/dev/null &
    if [ $((n % 3)) = 2 ]; then wait; fi
  done
  wait
  ls -la $B/libhipnlp_hp*.so
  exit 0
fi
shift || true
BATCHES=${1:-1,64}
OUT=$ROOT/gpurun_out/hess_phases.txt
: > $OUT
for w in periodic stairs; do
  for n in 0 1 2 3 4 5 6; do
    if [ $n = 6 ]; then unset HIPNLP_LIB_PATH; else export HIPNLP_LIB_PATH=$B/libhipnlp_hp$n.so; fi
    HESS_WORKLOAD=$w HESS_BATCHES=$BATCHES python3 $ROOT/tools/diag/hess_bench.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('phases 0..%s  %-70s %9.2f us' % ('$n', d['workload'], d['ms_per_eval'] * 1e3))" >> $OUT
  done
done
cat $OUT
