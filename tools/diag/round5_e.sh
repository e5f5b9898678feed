#!/bin/bash
# GPU box, round 5, final build: the default bench line, the 2- / 4-rank rehearsals on one GPU (BENCH_REHEARSAL=1: gloo, every rank on
# device 0 — plumbing and bytes per step, never a measurement) and the smoke.
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 20 --warmup 5 --details-out gpurun_out/r05_bench_default_steps20.json > gpurun_out/r05_bench_default.log 2>&1; tail -1 gpurun_out/r05_bench_default.log > gpurun_out/r05_bench_default_final_line.json; wc -c gpurun_out/r05_bench_default_final_line.json
for R in 2 4; do
  BENCH_REHEARSAL=1 python bench.py --gpus $R --steps 30 --warmup 5 --no-cpu-baseline --details-out gpurun_out/r05_rehearsal_${R}_ranks_on_one_gpu.json > gpurun_out/r05_rehearsal_$R.log 2>&1
  tail -1 gpurun_out/r05_rehearsal_$R.log > gpurun_out/r05_rehearsal_${R}_final_line.json; tail -c 600 gpurun_out/r05_rehearsal_$R.log; echo
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r05_smoke.txt 2>&1; tail -2 gpurun_out/r05_smoke.txt
