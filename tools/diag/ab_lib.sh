#!/bin/bash
# GPU box: A/B of two BUILDS of libhipnlp.so in ONE session (kernel times differ by a few percent from box to box): bench.py's
# event-timed knot kernel for the product library and for $1 (HIPNLP_LIB_PATH), at the batches given (default 64 1024) and on
# the stairs configuration 200 x 16.     usage: tools/diag/ab_lib.sh tools/diag/_build/libX.so [batches...]   -> gpurun_out/ab_<name>.txt
set -u
ALT=$1; shift
BATCHES=${@:-"64 1024"}
NAME=$(basename $ALT .so)
OUT=gpurun_out/ab_$NAME.txt
: > $OUT
one() {  # label, extra env, bench args
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export HIPNLP_LIB_PATH=$lib; else unset HIPNLP_LIB_PATH; fi
  python3 bench.py --no-cpu-baseline --no-hessian --no-host --no-throughput ${AB_FLAGS:-} "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s %-34s value %.4g knots/s  ms_per_step %.5f  kernel_ms %.5f' % ('$label', ' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" "$@" >> $OUT
}
for rep in 1 2; do
  for B in $BATCHES; do
    STEPS=$([ $B -ge 1024 ] && echo 60 || echo 600)
    one product "" --batch $B --steps $STEPS --warmup 30
    one alt "$ALT" --batch $B --steps $STEPS --warmup 30
  done
  one product "" --workload stairs --horizon 200 --batch 16 --steps 600 --warmup 30
  one alt "$ALT" --workload stairs --horizon 200 --batch 16 --steps 600 --warmup 30
done
cat $OUT
