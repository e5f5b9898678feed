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
WORKLOADS=${@:-"periodic_N100_B64:--batch=64 periodic_N100_B1:--batch=1 periodic_N100_B1024:--batch=1024 stairs_N200_B16:--workload=stairs,--horizon=200,--batch=16 periodic_N100_B1_vf:--batch=1,--varying-first periodic_N100_B64_vf:--batch=64,--varying-first periodic_N100_B1024_vf:--batch=1024,--varying-first stairs_N200_B16_vf:--workload=stairs,--horizon=200,--batch=16,--varying-first periodic_N100_B64_ccsv:--batch=64,--ccs-constants-in-place periodic_N100_B1024_ccsv:--batch=1024,--ccs-constants-in-place"}
PARTS=${PROFILE_PARTS:-"bench hess pose calib"}   # which parts run (a whole round does not fit one gpurun call of 20 minutes)
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
run() { name=$1; shift; echo "== $name" ; rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "   rc=$?"; }
has bench && for W in $WORKLOADS; do
  NAME=${W%%:*}; ARGS=$(echo ${W#*:} | tr ',' ' ')
  B=$(echo $ARGS | sed -n 's/.*--batch=\([0-9]*\).*/\1/p'); B=${B:-1}
  STEPS=$([ $B -ge 1024 ] && echo 30 || echo 300)
  COMMON="--no-cpu-baseline --no-hessian --no-host --no-throughput"
  run trace_$NAME --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python3 bench.py --steps $STEPS --warmup 20 $ARGS $COMMON
  run fetch_$NAME --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  run write_$NAME --pmc WRITE_SIZE --output-format csv -d $OUT/write_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  run sq_$NAME --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  run lds_$NAME --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/lds_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  # lane utilisation of the VALU issue slots: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) = share of live lanes per issued VALU cycle
  run lane_$NAME --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/lane_$NAME -- python3 bench.py --steps 20 --warmup 5 $ARGS $COMMON
  python3 bench.py --steps $([ $B -ge 1024 ] && echo 50 || echo 1000) --warmup 50 $ARGS $COMMON --details-out $OUT/bench_$NAME.json > /dev/null 2>&1
done
# exact Hessian and pose finder: ONE (kernel, N, batch) per trace, so that every CSV average is the duration of one configuration
has hess && for CFG in periodic:100:1 periodic:100:64 periodic:100:256 stairs:200:16; do
  HW=${CFG%%:*}; R=${CFG#*:}; HN=${R%%:*}; HB=${R#*:}
  NAME=hess_${HW}_N${HN}_B${HB}
  echo "== $NAME"
  HESS_WORKLOAD=$HW HESS_N=$HN HESS_BATCHES=$HB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python3 tools/diag/hess_bench.py > $OUT/bench_$NAME.jsonl 2> $OUT/trace_$NAME.log
  HESS_WORKLOAD=$HW HESS_N=$HN HESS_BATCHES=$HB rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$NAME -- python3 tools/diag/hess_bench.py > /dev/null 2> $OUT/fetch_$NAME.log
  HESS_WORKLOAD=$HW HESS_N=$HN HESS_BATCHES=$HB rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$NAME -- python3 tools/diag/hess_bench.py > /dev/null 2> $OUT/write_$NAME.log
  HESS_WORKLOAD=$HW HESS_N=$HN HESS_BATCHES=$HB rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/lane_$NAME -- python3 tools/diag/hess_bench.py > /dev/null 2> $OUT/lane_$NAME.log
  HESS_WORKLOAD=$HW HESS_N=$HN HESS_BATCHES=$HB rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/lds_$NAME -- python3 tools/diag/hess_bench.py > /dev/null 2> $OUT/lds_$NAME.log
done
has pose && for CFG in callbacks:1 callbacks:4096 hessian:1 hessian:4096; do
  WHAT=${CFG%%:*}; PB=${CFG#*:}
  NAME=pose_${WHAT}_B${PB}
  echo "== $NAME"
  POSE_WHAT=$WHAT POSE_BATCHES=$PB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python3 tools/diag/pose_bench.py > $OUT/bench_$NAME.jsonl 2> $OUT/trace_$NAME.log
  POSE_WHAT=$WHAT POSE_BATCHES=$PB rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$NAME -- python3 tools/diag/pose_bench.py > /dev/null 2> $OUT/fetch_$NAME.log
  POSE_WHAT=$WHAT POSE_BATCHES=$PB rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$NAME -- python3 tools/diag/pose_bench.py > /dev/null 2> $OUT/write_$NAME.log
  # what bounds the batch launches: live lanes per issued VALU instruction, VALU / LDS instruction counts, LDS array cycles against CU-busy cycles
  POSE_WHAT=$WHAT POSE_BATCHES=$PB rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/lane_$NAME -- python3 tools/diag/pose_bench.py > /dev/null 2> $OUT/lane_$NAME.log
  POSE_WHAT=$WHAT POSE_BATCHES=$PB rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/lds_$NAME -- python3 tools/diag/pose_bench.py > /dev/null 2> $OUT/lds_$NAME.log
done
has calib && run calib_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- tools/diag/_build/calib
has calib && run calib_write --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- tools/diag/_build/calib
ls $OUT | head -60
