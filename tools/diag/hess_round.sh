#!/bin/bash
# knot Hessian kernels: GPU parity tests, batch timing (product vs an alternative build), stamps.  usage: hess_round.sh TAG [alt-lib]
export PYTHONPATH=$PWD
T=${1:-x}; ALT=${2:-}
timeout -k 10 600 python -m pytest tests/test_gpu_hessian_direct.py tests/test_gpu_parity.py -x -q -m gpu -k "hess or Hess" > gpurun_out/hess_tests_$T.log 2>&1 || exit 1
OUT=gpurun_out/hess_bench_$T.txt; : > $OUT
for rep in 1 2; do
  echo "product" >> $OUT; HESS_BATCHES=1,64,256 timeout -k 10 200 python tools/diag/hess_bench.py 2>/dev/null | cut -c1-300 >> $OUT
  if [ -n "$ALT" ]; then echo "alt" >> $OUT; HIPNLP_LIB_PATH=$ALT HESS_BATCHES=1,64,256 timeout -k 10 200 python tools/diag/hess_bench.py 2>/dev/null | cut -c1-300 >> $OUT; fi
done
timeout -k 10 120 python tools/diag/hess_stamps.py 64 2>&1 | grep -v amdgpu.ids | head -12 > gpurun_out/hess_stamps_$T.txt
