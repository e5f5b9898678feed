#!/bin/bash
# The 100-knot launch under rocprofv3 --kernel-trace --stats, two workgroups per knot (the shipped rule) against one (HIPNLP_SPLIT=0 on the
# diagnostic build: same device code objects), in ONE session: the tracer perturbs a launch this short, so the two are compared under it.
#   GPU box: bash tools/diag/split_trace_ab.sh   -> gpurun_out/r06_split_trace_ab.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_split_ab
rm -rf $OUT; mkdir -p $OUT
export HIPNLP_LIB_PATH=$PWD/tests/_build/libhipnlp_diag.so
C="--no-cpu-baseline --no-hessian --no-host --no-throughput --steps 300 --warmup 20"
for rep in 1 2; do
  for S in 1 0; do
    HIPNLP_SPLIT=$S rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s${S}_$rep -- python3 bench.py $C > $OUT/s${S}_$rep.log 2>&1
  done
done
python3 - <<'PY' | tee gpurun_out/r06_split_trace_ab.txt
import csv, glob, numpy as np
for rep in (1, 2):
    for s in (1, 0):
        f = glob.glob("gpurun_out/prof_split_ab/s%d_%d/*/*_kernel_trace.csv" % (s, rep))[0]
        rows = [r for r in csv.DictReader(open(f)) if "knot_kernel" in r["Kernel_Name"]]
        d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows])
        st = np.array([int(r["Start_Timestamp"]) for r in rows])
        print("rep %d  %-28s dispatches %d  duration ns: mean %.0f median %.0f p10 %.0f p90 %.0f   period median %.0f   %s" % (
            rep, "two workgroups per knot" if s else "one workgroup per knot", len(d), d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90),
            np.median(st[1:] - st[:-1]), rows[0]["Kernel_Name"][41:78]))
PY
