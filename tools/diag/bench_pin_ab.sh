#!/bin/bash
# GPU box: the default bench's host-path figures with the process pinned to the card's NUMA node (the default) and left where the scheduler
# puts it (--no-numa-pin), alternating, two repetitions.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/bench_pin_ab.txt
: > $OUT
for rep in 1 2; do
  for MODE in "" "--no-numa-pin"; do
    python bench.py --steps 20 --warmup 5 --no-throughput $MODE 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-14s numa: %s | value %.4g  hessian host %s  host_visible %s  ratios %s' % ('$MODE' or 'pinned', d['config'].get('numa'), d['value'], d['exact_hessian'], d['host_visible'], {k: d['cpu_baseline']['gpu_over_cpu'][k] for k in ('host_all_vs_1t', 'host_all_raw_vs_1t')}))" >> $OUT
  done
done
cat $OUT
