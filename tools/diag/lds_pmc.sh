cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_lds
rm -rf $OUT; mkdir -p $OUT
C="--no-cpu-baseline --no-hessian --no-host --steps 20 --warmup 5 --batch=64"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 bench.py $C > $OUT/a.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 bench.py $C > $OUT/b.log 2>&1; echo rc=$?
python3 - <<'PY'
import csv, glob, collections
for d in ('a','b'):
    for f in glob.glob('gpurun_out/prof_lds/%s/**/*counter_collection.csv' % d, recursive=True):
        acc=collections.defaultdict(float); n=collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if 'knot_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
        for k in acc: print(d, k, acc[k]/n[k], 'per launch (%d launches)' % n[k])
PY
