#!/bin/bash
# GPU box, round 5: the final build once more through the GPU suite; the 100-knot workloads traced on a WARM device (the first trace of a
# session runs on cold clocks: part A of this round's profile traced periodic_N100_B1 first — 9.14 us average, 7.04 minimum); the counters
# of the x 1024 launch with its reduction kernel (the in-launch sample of the constants alternates its positions since this build).
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_solver_order.py > gpurun_out/r05_gputest_final2.log 2>&1; tail -3 gpurun_out/r05_gputest_final2.log
PROFILE_PARTS=bench bash tools/diag/profile_round.sh r05 "periodic_N100_B1024_vf:--batch=1024,--varying-first" "periodic_N100_B1:--batch=1" "periodic_N100_B1_vf:--batch=1,--varying-first" > gpurun_out/profile_round_r05_d.log 2>&1
tail -2 gpurun_out/profile_round_r05_d.log
