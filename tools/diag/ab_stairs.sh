#!/bin/bash
# GPU box: A/B of two builds of libhipnlp.so on the stairs workloads in ONE session.   usage: tools/diag/ab_stairs.sh tools/diag/_build/libX.so
set -u
ALT=$1
OUT=gpurun_out/ab_stairs_$(basename $ALT .so).txt
: > $OUT
one() {
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export HIPNLP_LIB_PATH=$lib; else unset HIPNLP_LIB_PATH; fi
  python3 bench.py --no-cpu-baseline --no-hessian --no-host --workload stairs --no-throughput "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s %-40s value %.4g knots/s  kernel_ms %.5f' % ('$label', ' '.join(sys.argv[1:]), d['value'], d['roofline']['kernel_ms']))" "$@" >> $OUT
}
for rep in 1 2 3; do
  for CFG in "--horizon 200 --batch 16 --steps 600" "--horizon 100 --batch 64 --steps 300" "--horizon 100 --batch 1024 --steps 40"; do
    one product "" $CFG --warmup 30
    one alt "$ALT" $CFG --warmup 30
  done
done
cat $OUT
