#!/bin/bash
# knot kernels: GPU parity tests of the callback path, then the throughput legs (event-timed kernel).  usage: knot_round.sh TAG [alt-lib]
export PYTHONPATH=$PWD
T=${1:-x}; ALT=${2:-}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mode_matrix.py -x -q -m gpu > gpurun_out/knot_tests_$T.log 2>&1 || exit 1
OUT=gpurun_out/knot_bench_$T.txt; : > $OUT
one() { local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export HIPNLP_LIB_PATH=$lib; else unset HIPNLP_LIB_PATH; fi
  python3 bench.py --no-cpu-baseline --no-hessian --no-host --no-throughput "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s %-60s value %.4g knots/s  ms_per_step %.5f  kernel_ms %.5f' % ('$label', ' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" "$@" >> $OUT
}
for rep in 1 2; do
  for cfg in "--batch 64 --steps 600 --varying-first" "--batch 1024 --steps 60 --varying-first" "--workload stairs --horizon 200 --batch 16 --steps 600 --varying-first" "--batch 64 --steps 600" "--batch 1 --steps 2000"; do
    one product "" $cfg --warmup 30
    if [ -n "$ALT" ]; then one alt "$ALT" $cfg --warmup 30; fi
  done
done
cat $OUT
