#!/bin/bash
# GPU box, round 6: the one-caller path (hipnlp_multi_create) — its GPU tests, the default bench line with the one_caller block, the 2- and
# 4-rank rehearsals on one GPU (BENCH_REHEARSAL=1: gloo, every rank on device 0 — plumbing and bytes per step, never a measurement), in which
# every exchange leg now starts from x in rank 0's host memory.
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r06_build.log 2>&1 || { tail -20 gpurun_out/r06_build.log; exit 1; }
python -m pytest tests/test_gpu_multi.py -m gpu -q -x > gpurun_out/r06_gputest_multi.log 2>&1; echo "multi tests rc=$?"; tail -3 gpurun_out/r06_gputest_multi.log
timeout -k 10 600 python bench.py --details-out gpurun_out/r06_bench_default.json > gpurun_out/r06_bench_default.log 2>&1; echo "bench rc=$?"; tail -c 3000 gpurun_out/r06_bench_default.log
for R in 2 4; do
  BENCH_REHEARSAL=1 timeout -k 10 500 python bench.py --gpus $R --steps 30 --warmup 5 --no-cpu-baseline --details-out gpurun_out/r06_rehearsal_${R}_ranks_on_one_gpu.json > gpurun_out/r06_rehearsal_$R.log 2>&1
  echo "rehearsal $R rc=$?"; tail -c 2500 gpurun_out/r06_rehearsal_$R.log
done
