#!/bin/bash
# GPU box: the unit counters of the four-wave VARY callback kernel (varying-first handle) at batch for SEVERAL BUILDS of the library in one
# session: VALU / LDS instructions and LDS-array cycles per knot, live lanes.  Separate --pmc passes, no trace domains.
#   usage: tools/diag/pmc_libs.sh <batch> <workload: periodic|stairs> lib1.so lib2.so ...   ("product" = the shipped library)  -> gpurun_out/pmc_libs.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
BATCH=$1; WL=$2; shift 2
OUT=gpurun_out/prof_libs
rm -rf $OUT; mkdir -p $OUT
N=100; [ $WL = stairs ] && N=200
for L in "$@"; do
  NAME=$(basename $L .so)
  if [ "$L" = product ]; then unset HIPNLP_LIB_PATH; else export HIPNLP_LIB_PATH=$PWD/$L; fi
  C="--no-cpu-baseline --no-hessian --no-host --no-throughput --steps 20 --warmup 5 --batch=$BATCH --workload $WL --horizon $N --varying-first"
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/a_$NAME -- python3 bench.py $C > $OUT/a_$NAME.log 2>&1; echo "$NAME pass a rc=$?"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/b_$NAME -- python3 bench.py $C > $OUT/b_$NAME.log 2>&1; echo "$NAME pass b rc=$?"
done
unset HIPNLP_LIB_PATH
python3 - $BATCH $N "$@" <<'PY' | tee -a gpurun_out/pmc_libs.txt
import csv, glob, collections, os, sys
B, N = int(sys.argv[1]), int(sys.argv[2]); knots = N * B
for lib in sys.argv[3:]:
    name = os.path.basename(lib).replace('.so', '')
    tot = {}
    for d in ('a', 'b'):
        for f in glob.glob('gpurun_out/prof_libs/%s_%s/**/*counter_collection.csv' % (d, name), recursive=True):
            acc = collections.defaultdict(float); n = collections.defaultdict(int)
            for r in csv.DictReader(open(f)):
                if 'knot_kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
            for k in acc: tot[k] = acc[k] / n[k]
    print('== %s, N=%d x %d (varying-first handle): per knot' % (name, N, B))
    for k in sorted(tot): print('   %-26s %12.1f' % (k, tot[k] / knots))
    if 'SQ_LDS_IDX_ACTIVE' in tot and 'SQ_BUSY_CU_CYCLES' in tot:
        print('   LDS array busy / CU busy     %.3f' % (tot['SQ_LDS_IDX_ACTIVE'] / tot['SQ_BUSY_CU_CYCLES']))
        if 'SQ_INSTS_VALU' in tot: print('   VALU issue / CU busy         %.3f' % (tot['SQ_INSTS_VALU'] / tot['SQ_BUSY_CU_CYCLES']))
    if 'SQ_THREAD_CYCLES_VALU' in tot and 'SQ_ACTIVE_INST_VALU' in tot:
        print('   live lanes per VALU cycle    %.3f' % (tot['SQ_THREAD_CYCLES_VALU'] / (64 * tot['SQ_ACTIVE_INST_VALU'])))
PY
