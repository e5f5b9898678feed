#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) for the
# bench workloads, and the 8-B-per-lane calibration of the TCC byte counters.  Output: gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 "$@" > $OUT/$name.log 2>&1; }
for B in 1 64 1024; do
  STEPS=$([ $B = 1024 ] && echo 30 || echo 300)
  run trace_B$B --kernel-trace --stats --output-format csv -d $OUT/trace_B$B -- python3 bench.py --steps $STEPS --warmup 20 --batch $B --no-cpu-baseline
  run fetch_B$B --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_B$B -- python3 bench.py --steps 20 --warmup 5 --batch $B --no-cpu-baseline
  run write_B$B --pmc WRITE_SIZE --output-format csv -d $OUT/write_B$B -- python3 bench.py --steps 20 --warmup 5 --batch $B --no-cpu-baseline
  run sq_B$B --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_B$B -- python3 bench.py --steps 20 --warmup 5 --batch $B --no-cpu-baseline
done
HESS_BATCHES=1,64 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hess -- python3 tools/diag/hess_bench.py > $OUT/hess_bench.jsonl 2> $OUT/trace_hess.log
run calib_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- tools/diag/_build/calib
run calib_write --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- tools/diag/_build/calib
for B in 1 64 1024; do python3 bench.py --steps $([ $B = 1024 ] && echo 50 || echo 1000) --warmup 50 --batch $B --no-cpu-baseline > $OUT/bench_B$B.json 2>/dev/null; done
ls $OUT | head -40
