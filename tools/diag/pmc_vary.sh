#!/bin/bash
# GPU box: the unit counters of the four-wave callback kernel at batch, CCS handle (every entry staged and stored, four workgroups per CU)
# against varying-first handle (VARY kernel, five per CU).  Separate --pmc passes, no trace domains.   -> gpurun_out/pmc_vary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_vary
rm -rf $OUT; mkdir -p $OUT
BATCH=${1:-64}
for V in ccs vf; do
  FLAG=$([ $V = vf ] && echo --varying-first || echo "")
  C="--no-cpu-baseline --no-hessian --no-host --no-throughput --steps 20 --warmup 5 --batch=$BATCH $FLAG"
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/a_$V -- python3 bench.py $C > $OUT/a_$V.log 2>&1; echo rc=$?
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/b_$V -- python3 bench.py $C > $OUT/b_$V.log 2>&1; echo rc=$?
done
python3 - $BATCH <<'PY' | tee gpurun_out/pmc_vary.txt
import csv, glob, collections, sys
B = int(sys.argv[1]); knots = 100 * B
for v in ('ccs', 'vf'):
    tot = {}
    for d in ('a', 'b'):
        for f in glob.glob('gpurun_out/prof_vary/%s_%s/**/*counter_collection.csv' % (d, v), recursive=True):
            acc = collections.defaultdict(float); n = collections.defaultdict(int)
            for r in csv.DictReader(open(f)):
                if 'knot_kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
            for k in acc: tot[k] = acc[k] / n[k]
    print('== %s, N=100 x %d: per knot' % (v, B))
    for k in sorted(tot): print('   %-26s %12.1f' % (k, tot[k] / knots))
    if 'SQ_LDS_IDX_ACTIVE' in tot and 'SQ_BUSY_CU_CYCLES' in tot:
        print('   LDS array busy / CU busy     %.3f' % (tot['SQ_LDS_IDX_ACTIVE'] / tot['SQ_BUSY_CU_CYCLES']))
    if 'SQ_THREAD_CYCLES_VALU' in tot and 'SQ_ACTIVE_INST_VALU' in tot:
        print('   live lanes per VALU cycle    %.3f' % (tot['SQ_THREAD_CYCLES_VALU'] / (64 * tot['SQ_ACTIVE_INST_VALU'])))
PY
