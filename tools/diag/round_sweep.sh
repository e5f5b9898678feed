# How a launch's duration steps with the number of workgroup "rounds" (1024 resident four-wave workgroups): stairs N=200 x B and planar N=100 x B
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-hessian --no-host --steps 300 --warmup 30"
for B in 5 10 14 15 16 17 20 21 26; do
  timeout -k 10 120 python bench.py $C --workload stairs --horizon 200 --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stairs 200 x %-3d %5d wg  %.4g knots/s  %.5f ms' % ($B, 200*$B, d['value'], d['ms_per_step']))"
done
for B in 10 20 30 40 41 50 61 64; do
  timeout -k 10 120 python bench.py $C --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('planar 100 x %-3d %5d wg  %.4g knots/s  %.5f ms' % ($B, 100*$B, d['value'], d['ms_per_step']))"
done
