set -e
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-hessian --no-host"
timeout -k 10 120 python bench.py $C --steps 2000 --warmup 100 --details-out gpurun_out/bs_head.json > /dev/null 2>&1
timeout -k 10 120 python bench.py $C --steps 600 --warmup 30 --batch 64 --details-out gpurun_out/bs_b64.json > /dev/null 2>&1
timeout -k 10 120 python bench.py $C --steps 60 --warmup 30 --batch 1024 --details-out gpurun_out/bs_b1024.json > /dev/null 2>&1
timeout -k 10 120 python bench.py $C --steps 600 --warmup 30 --workload stairs --horizon 200 --batch 16 --details-out gpurun_out/bs_stairs.json > /dev/null 2>&1
timeout -k 10 120 python bench.py $C --steps 2000 --warmup 100 --workload stairs --details-out gpurun_out/bs_stairs1.json > /dev/null 2>&1
python - <<'PY'
import json
for f in ('head','b64','b1024','stairs','stairs1'):
    d=json.loads(open('gpurun_out/bs_%s.json'%f).read().strip().splitlines()[-1])
    print(f, '%.4g knots/s  %.5f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
PY
