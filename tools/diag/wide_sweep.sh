# 8-wave (two workgroups per CU) against 4-wave on launches of 257..512 knots
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-hessian --no-host --steps 1000 --warmup 50"
for B in 1 2 3 4 5 6 8; do
  for W in 4 8; do
    HIPNLP_LIB_PATH=$PWD/tests/_build/libhipnlp_diag.so HIPNLP_WAVES=$W timeout -k 10 120 python bench.py $C --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B waves=$W  %.4g knots/s  %.5f ms' % (d['value'], d['ms_per_step']))"
  done
done
