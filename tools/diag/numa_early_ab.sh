#!/bin/bash
# GPU box: does the side of the host the calling thread (and its first-touched pinned memory) sits on decide whether the Hessian's early run
# helps?  hess_early_ab.py (forced off / forced on / decided by the handle) and the callback path, pinned to the CPUs of NUMA node 0 and of
# node 1 in ONE session, with the node of the card the process sees.
OUT=gpurun_out/numa_early_ab_$(date +%H%M%S).txt
{
  python3 - <<'PY'
import ctypes, glob, os
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64)
hip.hipDeviceGetPCIBusId(buf, 64, 0)
bus = buf.value.decode().lower()
node = open("/sys/bus/pci/devices/%s/numa_node" % bus).read().strip() if os.path.exists("/sys/bus/pci/devices/%s/numa_node" % bus) else "?"
print("device 0: PCI %s, NUMA node %s" % (bus, node))
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    print(os.path.basename(n), "cpus", open(n + "/cpulist").read().strip())
PY
  for PIN in 0-63 64-127; do
    echo "=== pinned to CPUs $PIN"
    echo "--- exact Hessian, planar"; taskset -c $PIN timeout -k 10 200 python3 tools/diag/hess_early_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-170
    echo "--- exact Hessian, smooth steps"; HESS_WORKLOAD=stairs taskset -c $PIN timeout -k 10 200 python3 tools/diag/hess_early_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-170
    echo "--- callback path, early store 1 / 0"; taskset -c $PIN timeout -k 10 200 python3 tools/diag/early_store_time.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-170
  done
} > $OUT 2>&1
cat $OUT
