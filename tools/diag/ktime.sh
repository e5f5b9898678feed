#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace average duration of the knot kernel for the bench workloads (A/B measurements).
# usage: tools/diag/ktime.sh <tag> [batches...]     output: gpurun_out/ktime_<tag>.txt
set -u
TAG=${1:-x}; shift
BATCHES=${@:-"1 64 1024"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ktime_$TAG
mkdir -p $OUT
: > gpurun_out/ktime_$TAG.txt
for B in $BATCHES; do
  STEPS=$([ $B -ge 1024 ] && echo 30 || echo 300)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/B$B -- python3 bench.py --steps $STEPS --warmup 20 --batch $B --no-cpu-baseline ${EXTRA:-} > $OUT/B$B.log 2>&1
  f=$(find $OUT/B$B -name '*kernel_stats.csv' | head -1)
  echo "B=$B $(grep -E 'knot_kernel|reduce_kernel' $f | awk -F, '{gsub(/"/,""); printf "%s calls=%s avg_ns=%s | ", substr($1,1,60), $(NF-6), $(NF-4)}')" >> gpurun_out/ktime_$TAG.txt
  grep -h "^{" $OUT/B$B.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench: value %.4g knots/s  ms_per_step %.5f  event_kernel_ms %.5f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> gpurun_out/ktime_$TAG.txt
done
cat gpurun_out/ktime_$TAG.txt
