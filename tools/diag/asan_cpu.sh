#!/bin/bash
# CPU sanitizer pass (GPU AddressSanitizer is not available on this pool): the knot / pose / Hessian bodies through the test-only host
# emulation and the oracle, both rebuilt with -fsanitize=address,undefined, under the hostemu / oracle / golden test files.
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o $T/hostemu.so tests/hostemu/hostemu.cpp
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o $T/oracle.so oracle/kinodyn_oracle.cpp oracle/pose_oracle.cpp
python -c "import sys; sys.path.insert(0,'tests'); import hostemu_lib, oracle_lib; hostemu_lib.build(); oracle_lib.build()"
cp tests/_build/libhipnlp_hostemu.so $T/hostemu_orig.so; cp oracle/_build/libkinodyn_oracle.so $T/oracle_orig.so
restore() { cp $T/hostemu_orig.so tests/_build/libhipnlp_hostemu.so; cp $T/oracle_orig.so oracle/_build/libkinodyn_oracle.so; touch tests/_build/libhipnlp_hostemu.so oracle/_build/libkinodyn_oracle.so; }
trap restore EXIT
cp $T/hostemu.so tests/_build/libhipnlp_hostemu.so; cp $T/oracle.so oracle/_build/libkinodyn_oracle.so
touch tests/_build/libhipnlp_hostemu.so oracle/_build/libkinodyn_oracle.so
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 \
  python -m pytest tests/test_kernel_body_hostemu.py tests/test_pose_body_hostemu.py tests/test_oracle.py tests/test_golden_pose.py tests/test_golden_planner.py -x -q
