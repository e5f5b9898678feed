#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes) of the knot kernel for several library builds
# usage: tools/diag/traffic_ab.sh "<batch list>" lib_a.so lib_b.so ...      output: gpurun_out/traffic_ab/
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
BATCHES=$1; shift
OUT=gpurun_out/traffic_ab
rm -rf $OUT; mkdir -p $OUT
for lib in "$@"; do
  name=$(basename $lib .so)
  for B in $BATCHES; do
    for C in FETCH_SIZE WRITE_SIZE; do
      HIPNLP_LIB_PATH=$lib rocprofv3 --pmc $C --output-format csv -d $OUT/${name}_B${B}_$C -- python3 bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline > $OUT/${name}_B${B}_$C.log 2>&1 || exit 1
    done
  done
done
python3 - <<'PY'
import csv, glob, os
out = "gpurun_out/traffic_ab"
for d in sorted(glob.glob(out + "/*/")):
    acc = {}
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "knot_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(os.path.basename(d[:-1]), k, "KiB/launch %.1f" % (sum(v) / len(v)), "n", len(v))
PY
