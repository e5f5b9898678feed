#!/bin/bash
# GPU box: are the early copy-outs (callback path: the early third of jac g and five rows of g in six behind the second barrier; exact Hessian:
# the run at the start of a knot's block) a gain on THIS box?  Same-session A/Bs of both, with where the process and the GPU sit.
#   -> gpurun_out/early_boxes_<host-tag>.txt   (run on several boxes: the answer was not the same on all of them)
OUT=gpurun_out/early_boxes_$(date +%H%M%S).txt
{
  echo "cpu: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)  allowed cpus: $(grep Cpus_allowed_list /proc/self/status | cut -f2)"
  for d in /sys/class/drm/card*/device; do [ -e $d/numa_node ] && echo "$d numa_node $(cat $d/numa_node) $(cat $d/current_link_speed 2>/dev/null) x$(cat $d/current_link_width 2>/dev/null)"; done
  echo "numa nodes: $(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l); cpus of node0: $(cat /sys/devices/system/node/node0/cpulist 2>/dev/null)"
  echo "--- callback path (hipnlp_eval, all four outputs, varying-first): early store 1 / 0"
  timeout -k 10 200 python3 tools/diag/early_store_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
  echo "--- exact Hessian, planar"
  timeout -k 10 200 python3 tools/diag/hess_early_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-160
  echo "--- exact Hessian, smooth steps"
  HESS_WORKLOAD=stairs timeout -k 10 200 python3 tools/diag/hess_early_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-160
} > $OUT 2>&1
cat $OUT
