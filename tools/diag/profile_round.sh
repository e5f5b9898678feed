#!/bin/bash
# Runs on the GPU box (via gpurun): for every workload DESIGN.md quotes — kernel-trace stats + separate PMC passes (FETCH_SIZE /
# WRITE_SIZE / SQ counters; never combined with a trace, guides/MI355X_MICROARCH.md) of the SAME bench.py command, the exact-Hessian
# kernels on both terrains, and the 8-B-per-lane calibration of the TCC byte counters.  The program goes directly after `--`.
# Output: gpurun_out/prof_<tag>/ ; tools/diag/summarize_profiles.py <tag> turns it into profiles/.
# usage: tools/diag/profile_round.sh <tag> [workload ...]      workload = name:bench-args, default list below
set -u
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
WORKLOADS=${@:-"periodic_N100_B1:--batch=1 periodic_N100_B64:--batch=64 periodic_N100_B1024:--batch=1024 stairs_N200_B16:--workload=stairs,--horizon=200,--batch=16"}
run() { name=$1; shift; echo "== $name" ; rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "   rc=$?"; }
for W in $WORKLOADS; do
  NAME=${W%%:*}; ARGS=$(echo ${W#*:} | tr ',' ' ')
  B=$(echo $ARGS | sed -n 's/.*--batch=\([0-9]*\).*/\1/p'); B=${B:-1}
  STEPS=$([ $B -ge 1024 ] && echo 30 || echo 300)
  COMMON="--no-cpu-baseline --no-hessian --no-host"
  run trace_$NAME --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python3 bench.py --steps $STEPS --warmup 20 $ARGS $COMMON
  run fetch_$NAME --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  run write_$NAME --pmc WRITE_SIZE --output-format csv -d $OUT/write_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  run sq_$NAME --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  run lds_$NAME --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/lds_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  python3 bench.py --steps $([ $B -ge 1024 ] && echo 50 || echo 1000) --warmup 50 $ARGS $COMMON > $OUT/bench_$NAME.json 2>/dev/null
done
for HW in periodic stairs; do
  echo "== hess $HW"
  HESS_WORKLOAD=$HW HESS_BATCHES=1,16,64 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hess_$HW -- python3 tools/diag/hess_bench.py > $OUT/hess_bench_$HW.jsonl 2> $OUT/trace_hess_$HW.log
  HESS_WORKLOAD=$HW HESS_BATCHES=64 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_hess_$HW -- python3 tools/diag/hess_bench.py > /dev/null 2> $OUT/fetch_hess_$HW.log
  HESS_WORKLOAD=$HW HESS_BATCHES=64 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_hess_$HW -- python3 tools/diag/hess_bench.py > /dev/null 2> $OUT/write_hess_$HW.log
done
run calib_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- tools/diag/_build/calib
run calib_write --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- tools/diag/_build/calib
ls $OUT | head -60
