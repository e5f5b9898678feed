#!/bin/bash
# GPU box, round 5: second part of the profile round (CCS / x 1024 / stairs workloads, Hessian and pose kernels, counter calibration) and
# the default bench's host-path figures with and without the early copy-outs (diagnostic build, HIPNLP_EARLY_STORE=0) in ONE session.
cd "$GRAFT_REPO_ROOT"
PROFILE_PARTS="bench hess pose calib" bash tools/diag/profile_round.sh r05 "periodic_N100_B1024:--batch=1024" "stairs_N200_B16:--workload=stairs,--horizon=200,--batch=16" "periodic_N100_B1_vf:--batch=1,--varying-first" "periodic_N100_B64_ccsv:--batch=64,--ccs-constants-in-place" "periodic_N100_B1024_ccsv:--batch=1024,--ccs-constants-in-place" > gpurun_out/profile_round_r05_b.log 2>&1
tail -3 gpurun_out/profile_round_r05_b.log
OUT=gpurun_out/r05_bench_early_ab.txt
: > $OUT
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-throughput --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product    ', d['exact_hessian'], d['host_visible'])" >> $OUT
  HIPNLP_LIB_PATH=$PWD/tests/_build/libhipnlp_diag.so HIPNLP_EARLY_STORE=0 python bench.py --steps 20 --warmup 5 --no-throughput --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('early off  ', d['exact_hessian'], d['host_visible'])" >> $OUT
done
cat $OUT
