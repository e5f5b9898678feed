#!/bin/bash
# Where does the two-workgroups-per-knot regime end?  The diagnostic build (tests/_build/libhipnlp_diag.so) with HIPNLP_SPLIT_MAX = 0 (never
# split), 256 (the shipped rule: one workgroup per CU) and 512 (two per CU), device-resident launches of 100 knots x batch 1 .. 4 and of the
# stairs 50 / 100 / 200 knots.   GPU box:  bash tools/diag/split_regime.sh  -> gpurun_out/r06_split_regime.txt
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r06_split_regime.txt
: > $OUT
for cfg in "--batch 1" "--batch 2" "--batch 3" "--batch 4" "--workload stairs --horizon 50" "--workload stairs --horizon 100" "--workload stairs --horizon 200" "--horizon 200" "--horizon 250"; do
  for M in 0 256 512; do
    HIPNLP_LIB_PATH=tests/_build/libhipnlp_diag.so HIPNLP_SPLIT_MAX=$M python3 bench.py $cfg --no-cpu-baseline --no-hessian --no-host --no-throughput --steps 2000 --warmup 200 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-40s split_max %-4s  ms_per_step %.5f  kernel_ms %.5f  (%.4g knots/s)' % ('$cfg', '$M', d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" >> $OUT
  done
done
cat $OUT
