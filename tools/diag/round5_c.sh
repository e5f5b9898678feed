#!/bin/bash
# GPU box, round 5, last part: the whole GPU suite, the default bench line of the final build, and the 2- / 4-rank rehearsals on one GPU
# (BENCH_REHEARSAL=1: gloo, every rank on device 0 — plumbing and bytes per step, never a measurement).
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_final.log 2>&1; tail -4 gpurun_out/r05_gputest_final.log
python bench.py --steps 20 --warmup 5 --details-out gpurun_out/r05_bench_default_steps20.json > gpurun_out/r05_bench_default.log 2>&1; tail -1 gpurun_out/r05_bench_default.log > gpurun_out/r05_bench_default_final_line.json; wc -c gpurun_out/r05_bench_default_final_line.json
for R in 2 4; do
  BENCH_REHEARSAL=1 python bench.py --gpus $R --steps 30 --warmup 5 --no-cpu-baseline --details-out gpurun_out/r05_rehearsal_${R}_ranks_on_one_gpu.json > gpurun_out/r05_rehearsal_$R.log 2>&1
  tail -1 gpurun_out/r05_rehearsal_$R.log > gpurun_out/r05_rehearsal_${R}_final_line.json; tail -c 1400 gpurun_out/r05_rehearsal_$R.log
done
